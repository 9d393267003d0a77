"""ctypes front end of the whole-step C oracle (TEST INFRASTRUCTURE; see step_oracle.c for the citations).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

``World(spec, s)`` is scene ``s`` of a BatchEngine spec (the dict `diffsdfsim_amd.scenes` and `tests/rollout_helpers`
build) stepped by the reference's algorithm on the CPU.  ``hull="scipy"`` routes the convex hull of every contact
cluster to scipy.spatial.ConvexHull -- the very function the reference calls (contacts.py:132) -- through a callback;
``hull="own"`` uses the C file's own hull (no Python in the loop: the timing leg).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_HULL_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int))


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libstep_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.so_world_create.restype = ctypes.c_void_p
        L.so_world_time.restype = ctypes.c_double
        for f in ("so_world_set_mesh", "so_world_share_mesh", "so_set_hull_callback", "so_set_lcp_backward", "so_world_init", "so_world_step",
                  "so_world_time", "so_world_nsub", "so_world_counters", "so_world_state", "so_world_ncontacts", "so_world_contacts",
                  "so_world_substep", "so_world_substep_contacts", "so_world_free"):
            getattr(L, f).argtypes = None
        _LIB = L
    return _LIB


def set_lu_threads(n):
    """Threads ONE dense factorisation may share its trailing update among (bit-identical results).  Off by default and in
    every timing; the tests that follow a single scene over hundreds of steps switch it on to finish sooner."""
    lib().lcp_oracle_set_lu_threads(ctypes.c_int(int(n)))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@_HULL_CB
def _scipy_hull(pts, m, dim, out):
    """scipy.spatial.ConvexHull on the cluster's points, as contacts.py:131-134; -1 where it raises QhullError."""
    from scipy.spatial import ConvexHull, QhullError
    a = np.ctypeslib.as_array(pts, shape=(m * dim,)).reshape(m, dim).copy()
    try:
        v = ConvexHull(a).vertices
    except (QhullError, ValueError):
        return -1
    for i, x in enumerate(v):
        out[i] = int(x)
    return len(v)


class World:
    def __init__(self, spec, s=0, dt=1.0 / 30, eps=1e-3, tol=1e-8, fric_dirs=8, strict_no_pen=True, toc_diff=True, max_iter=10,
                 hull="scipy", lcp_backward=False, shared=None):
        """`shared`: dict mesh_id -> (verts, faces) contiguous arrays kept alive by the caller and borrowed by the C side
        (the 176 000-face floor of the benchmark scenes is shared by every world of a batch)."""
        L = lib()
        pose = np.ascontiguousarray(spec["pose"][s], np.float64)
        nb = self.nb = pose.shape[0]
        c = lambda k, dt_=np.float64: np.ascontiguousarray(np.asarray(spec[k])[s], dt_)
        Je = np.asarray(spec.get("Je", np.zeros((len(spec["pose"]), 0, 6 * nb))))[s]
        fixed = np.zeros(nb, np.int32)
        for b in range(nb):      # TotalConstraint3D rows: an identity block on the body's six velocities
            blk = Je[:, 6 * b:6 * b + 6]
            if np.abs(blk).sum() > 0:
                rows = np.nonzero(np.abs(blk).sum(axis=1))[0]
                assert len(rows) == 6 and np.array_equal(blk[rows], np.eye(6)), "only TotalConstraint3D is restated"
                fixed[b] = 1
        assert int(fixed.sum()) * 6 == Je.shape[0]
        nocon = np.ascontiguousarray(spec.get("no_contact", np.zeros((nb, nb))), np.uint8)
        st = c("shape_type", np.int32)
        assert (st <= 2).all(), "step_oracle.c restates box, sphere and cylinder"
        self._keep = [pose, st, nocon, fixed]
        args = [c("shape_prm"), pose, c("vel"), c("mass"), np.ascontiguousarray(np.asarray(spec["inertia"])[s].reshape(nb, 9), np.float64),
                c("restitution"), c("fric"), c("fext")]
        self._keep += args
        self.h = ctypes.c_void_p(L.so_world_create(nb, _p(st), *[_p(a) for a in args], _p(fixed), _p(nocon), ctypes.c_double(dt), ctypes.c_double(eps),
                                                   ctypes.c_double(tol), fric_dirs, int(strict_no_pen), int(toc_diff), max_iter))
        mid = np.asarray(spec["mesh_id"])[s]
        for b in range(nb):
            m = int(mid[b])
            if shared is not None and m in shared:
                v, f = shared[m]
                L.so_world_share_mesh(self.h, b, _p(v), len(v), _p(f), len(f))
            else:
                v = np.ascontiguousarray(spec["meshes"][m][0], np.float64); f = np.ascontiguousarray(spec["meshes"][m][1], np.int32)
                L.so_world_set_mesh(self.h, b, _p(v), len(v), _p(f), len(f))
        if hull == "scipy":
            L.so_set_hull_callback(self.h, _scipy_hull)
        L.so_set_lcp_backward(self.h, int(lcp_backward))
        n0 = L.so_world_init(self.h)
        if n0 == -1:
            raise AssertionError("Interpenetration at start")
        if n0 < -1:
            raise RuntimeError("step oracle: contact detection failed (%d)" % n0)

    def step(self, n=1):
        for _ in range(n):
            rc = lib().so_world_step(self.h)
            if rc:
                raise RuntimeError("step oracle: step failed with code %d" % rc)

    @property
    def t(self):
        return lib().so_world_time(self.h)

    @property
    def nsub(self):
        return lib().so_world_nsub(self.h)

    def counters(self):
        out = np.zeros(4, np.int64)
        lib().so_world_counters(self.h, _p(out))
        return dict(attempts=int(out[0]), lcp_solves=int(out[1]), lcp_rows=int(out[2]), fw_candidates=int(out[3]))

    def timers(self):
        """seconds spent in solve_dynamics (assembly + LCP [+ its backward]) and in find_contacts so far"""
        out = np.zeros(2)
        lib().so_world_timers(self.h, _p(out))
        return dict(solve=float(out[0]), detect=float(out[1]))

    def state(self):
        pose, vel = np.zeros((self.nb, 7)), np.zeros((self.nb, 6))
        lib().so_world_state(self.h, _p(pose), _p(vel))
        return pose, vel

    def _contacts(self, n, fn, *pre):
        body, geom, st, lap = np.zeros((n, 2), np.int32), np.zeros((n, 10)), np.zeros(n, np.int32), np.zeros((n, 2))
        fn(self.h, *pre, _p(body), _p(geom), _p(st), _p(lap))
        return body, geom, st, lap

    def contacts(self):
        """(body [n, 2], geom [n, 10] = normal, p1, p2, penetration, stable_mask [n], |laplacians| [n, 2])"""
        return self._contacts(lib().so_world_ncontacts(self.h), lib().so_world_contacts)

    def substep(self, k):
        """trajectory record k as the reference appends it (world.py:373-377): (t before, poses, velocities after, contacts after)"""
        pose, vel, t = np.zeros((self.nb, 7)), np.zeros((self.nb, 6)), ctypes.c_double()
        n = lib().so_world_substep(self.h, k, ctypes.byref(t), _p(pose), _p(vel))
        assert n >= 0
        return t.value, pose, vel, self._contacts(n, lib().so_world_substep_contacts, k)

    def close(self):
        if self.h:
            lib().so_world_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_many(worlds, nsteps, nthreads):
    """`nsteps` outer steps of every world, worlds distributed over `nthreads` OpenMP threads (hull = own only)."""
    arr = (ctypes.c_void_p * len(worlds))(*[w.h for w in worlds])
    rc = lib().so_worlds_run(arr, len(worlds), nsteps, nthreads)
    if rc:
        raise RuntimeError("step oracle: run failed with code %d" % rc)
