"""Tiny functional stand-in for py3ode's broadphase (TEST INFRASTRUCTURE ONLY).

The reference only uses ODE to enumerate candidate geometry pairs
(`lcp_physics/physics/world.py:69-72,399`) and to store per-geom position / rotation.
ODE's HashSpace callback order is not reproducible without ODE, so parity is defined
on the canonical order: all pairs ``(i < j)`` in body order, ``geom1 = bodies[i]``
(SURVEY.md §7 "bit-exact contact-pair indices").  No AABB cull is applied: the
narrow phase rejects far pairs itself, so the contact set is a superset-safe identity.
"""


class _Geom:
    def __init__(self, space=None, *args):
        self._pos = (0.0, 0.0, 0.0)
        self._quat = (1.0, 0.0, 0.0, 0.0)
        self.args = args

    def setPosition(self, p):
        self._pos = tuple(float(x) for x in p)

    def getPosition(self):
        return self._pos

    def setQuaternion(self, q):
        self._quat = tuple(float(x) for x in q)

    def getQuaternion(self):
        return self._quat


class GeomSphere(_Geom):
    pass


class GeomBox(_Geom):
    pass


class HashSpace:
    def __init__(self):
        self._geoms = []

    def add(self, g):
        self._geoms.append(g)

    def collide(self, arg, callback):
        n = len(self._geoms)
        for i in range(n):
            for j in range(i + 1, n):
                callback(arg, self._geoms[i], self._geoms[j])


def collide(g1, g2):
    return []
