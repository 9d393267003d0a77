"""Engine plug-in seam (boundary B2): mirrors lcp_physics/physics/engines.py:16-83.

The reference resolves ``World(engine='PdipmEngine')`` by name through ``get_instance`` (world.py:50-52).  Here the
stepping engine of ``World3D`` is always the HIP one; ``HipPdipmEngine.solve_dynamics(world, dt)`` exposes the
reference's engine call -- assemble the mixed LCP from the world's current state and contacts, solve it, return
the new velocities -- through ``dss_solve_dynamics`` without advancing the world."""
import ctypes

import torch


class Engine:
    def solve_dynamics(self, world, dt):
        raise NotImplementedError


class HipPdipmEngine(Engine):
    def __init__(self, max_iter=10):
        self.max_iter = max_iter

    def solve_dynamics(self, world, dt):
        """-> new_v, 1-D tensor of length 6 * nbodies (engines.py:82-83), values only."""
        E = world.engine
        E.arr["pose"].copy_(world.pose.detach())
        E.arr["vel"].copy_(world.vel.detach())
        E.arr["dt_try"].fill_(float(dt))
        E.arr["active"].fill_(1)
        rc = E.be.lib.dss_solve_dynamics(ctypes.byref(E.W), ctypes.c_void_p(E.be.ptr(E.lcp_ws)),
                                         ctypes.c_size_t(E.lcp_ws_bytes), E.be.stream())
        E.arr["active"].fill_(0)
        if rc != 0:
            raise RuntimeError("dss_solve_dynamics failed with code %d" % rc)
        return (-E.arr["x"][0]).clone()


PdipmEngine = HipPdipmEngine   # the name experiments pass as engine='PdipmEngine'
