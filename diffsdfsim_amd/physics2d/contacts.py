"""Analytic 2-D contacts on the device, behind the reference's contact-handler seam.

``lcp_physics.physics.world.World(bodies, ..., contact_callback=DiffContactHandler)`` instantiates the class it is given
(`lcp_physics/physics/utils.py:167-175`) and calls it once per candidate pair with ``(args, geom1, geom2)``
(`world.py:396-399`); the handler appends ``((normal, p1, p2, penetration), body1, body2)`` to ``world.contacts``
(`contacts.py:208-209`).  This module is that handler with the geometry computed by `csrc/contacts2d.hip`
(``dss_contacts2d_forward`` / ``_backward``, include/diffsdfsim_hip.h §R18): circles and convex polygons, gradients to
positions, radii and vertices.  `contacts2d` is the batched operator (P pairs per launch) for callers that hold many scenes.
"""
import ctypes

import torch

from .. import _lib

MAXV = 8      # DSS_C2D_MAXV


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)


def _device_forward(kind, nv, pos, rad, verts, sat_in, eps):
    _lib.require_device(pos, rad, verts, kind, nv, sat_in)
    L = _lib.lib()
    P, maxv = pos.shape[1], verts.shape[2]
    sat_out = torch.empty_like(sat_in)
    count = torch.empty(P, dtype=torch.int32, device=pos.device)
    out = torch.empty(P, 2, 7, dtype=torch.float64, device=pos.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
    rc = L.dss_contacts2d_forward(P, maxv, _ptr(kind), _ptr(nv), _ptr(pos), _ptr(rad), _ptr(verts), _ptr(sat_in),
                                  ctypes.c_double(eps), _ptr(sat_out), _ptr(count), _ptr(out), stream)
    if rc != 0:
        raise _lib.HipLibraryError("dss_contacts2d_forward failed (%d)" % rc)
    return out, count, sat_out


def _device_backward(kind, nv, pos, rad, verts, sat_in, eps, gout):
    L = _lib.lib()
    P, maxv = pos.shape[1], verts.shape[2]
    g_pos, g_rad, g_verts = torch.zeros_like(pos), torch.zeros_like(rad), torch.zeros_like(verts)
    stream = ctypes.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
    rc = L.dss_contacts2d_backward(P, maxv, _ptr(kind), _ptr(nv), _ptr(pos), _ptr(rad), _ptr(verts), _ptr(sat_in),
                                   ctypes.c_double(eps), _ptr(gout), _ptr(g_pos), _ptr(g_rad), _ptr(g_verts), stream)
    if rc != 0:
        raise _lib.HipLibraryError("dss_contacts2d_backward failed (%d)" % rc)
    return g_pos, g_rad, g_verts


DEVICE_KERNELS = (_device_forward, _device_backward)


class _Contacts2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, rad, verts, kind, nv, sat_in, eps, kernels):
        pos, rad, verts = pos.contiguous(), rad.contiguous(), verts.contiguous()
        out, count, sat_out = kernels[0](kind, nv, pos, rad, verts, sat_in, eps)
        ctx.save_for_backward(pos, rad, verts, kind, nv, sat_in)
        ctx.eps, ctx.kernels = eps, kernels
        ctx.mark_non_differentiable(count, sat_out)
        return out, count, sat_out

    @staticmethod
    def backward(ctx, gout, _gc, _gs):
        pos, rad, verts, kind, nv, sat_in = ctx.saved_tensors
        g_pos, g_rad, g_verts = ctx.kernels[1](kind, nv, pos, rad, verts, sat_in, ctx.eps, gout.contiguous())
        return g_pos, g_rad, g_verts, None, None, None, None, None


def contacts2d(pos, rad, verts, kind, nv, sat_in, eps, kernels=DEVICE_KERNELS):
    """Contacts of P body pairs.  pos [2, P, 2], rad [2, P], verts [2, P, maxv, 2] (float64, differentiable), kind / nv /
    sat_in [2, P] (int32: 0 circle / 1 polygon, vertex count, `last_sat_idx`), all on one HIP device.
    Returns out [P, 2, 7] = (normal, p1, p2, penetration) per contact, count [P] in {0, 1, 2}, sat_out [2, P]."""
    return _Contacts2D.apply(pos, rad, verts, kind, nv, sat_in, float(eps), kernels)


def _is_circle(body):
    return hasattr(body, "rad") and not hasattr(body, "verts")


def make_handler(kernels=DEVICE_KERNELS, device=None):
    """The handler class for a given pair of kernel entry points (the product's are the device library's)."""

    class Handler:
        def __call__(self, args, geom1, geom2):
            if geom1 in geom2.no_contact:
                return
            world = args[0]
            bodies = (world.bodies[geom1.body], world.bodies[geom2.body])
            base = bodies[0].pos
            dev = base.device if device is None else torch.device(device)
            mk = lambda v: torch.tensor(v, dtype=torch.int32, device=dev).reshape(2, 1)   # noqa: E731
            kind = mk([0 if _is_circle(b) else 1 for b in bodies])
            nvs = [0 if _is_circle(b) else len(b.verts) for b in bodies]
            if max(nvs) > MAXV:
                raise ValueError("polygons of more than %d vertices are outside the compiled limit" % MAXV)
            maxv = max(max(nvs), 1)
            pos = torch.stack([b.pos for b in bodies]).reshape(2, 1, 2).to(dev)
            rad = torch.stack([b.rad.reshape(()) if _is_circle(b) else base.new_zeros(()) for b in bodies]).reshape(2, 1).to(dev)
            rows = []
            for b, n in zip(bodies, nvs):
                vs = list(b.verts) if n else []
                rows.append(torch.stack(vs + [base.new_zeros(2)] * (maxv - n)))
            verts = torch.stack(rows).reshape(2, 1, maxv, 2).to(dev)
            sat_in = mk([0 if _is_circle(b) else int(b.last_sat_idx) for b in bodies])
            out, count, sat_out = contacts2d(pos, rad, verts, kind, mk(nvs), sat_in, float(world.eps), kernels)
            sat_out = sat_out.cpu()
            for s, b in enumerate(bodies):
                if not _is_circle(b):
                    b.last_sat_idx = int(sat_out[s, 0])
            out = out.to(base.device)
            for q in range(int(count[0])):
                c = out[0, q]
                world.contacts.append(((c[0:2], c[2:4], c[4:6], c[6]), geom1.body, geom2.body))

    Handler.__name__ = Handler.__qualname__ = "DiffContactHandler"
    return Handler


DiffContactHandler = make_handler()
