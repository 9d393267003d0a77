"""World-construction operators on the device: SDF point queries and polyhedral mass properties.

``sdf_query`` mirrors ``SDF3D.query_sdfs`` (`sdf_physics/physics3d/bodies.py:721-760`) for the analytic primitives,
``mesh_inertia`` mirrors ``get_ang_inertia`` (`bodies.py:260-395`).  Both call the C ABI (dss_sdf_query,
dss_mesh_inertia); there is no CPU path.
"""
import numpy as np
import torch

from . import _lib


def _dev(t, dtype=torch.float64):
    t = torch.as_tensor(t, dtype=dtype)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("a HIP device is required (no CPU fallback for the product path)")
        t = t.cuda()
    return t.contiguous()


def sdf_query(shape_type, prm, pts, return_grads=True, return_overlapmask=False):
    """pts [n,3] body-frame points -> sdf [n] (, normalised grad [n,3]) (, overlap mask [n] bool).
    prm: up to three shape parameters (+ the corner radius of a rounded box / brick as a fourth entry)."""
    pts = _dev(pts)
    prm_h = np.zeros(4)
    p = torch.as_tensor(prm, dtype=torch.float64).detach().cpu().numpy().reshape(-1)
    prm_h[: len(p)] = p
    n = pts.shape[0]
    sdf = torch.empty(n, dtype=torch.float64, device=pts.device)
    grad = torch.empty(n, 3, dtype=torch.float64, device=pts.device) if return_grads else None
    mask = torch.empty(n, dtype=torch.uint8, device=pts.device) if return_overlapmask else None
    if n:
        rc = _lib.lib().dss_sdf_query(int(shape_type), prm_h.ctypes.data_as(_lib.ctypes.c_void_p), _lib.ptr(pts), int(n),
                                      _lib.ptr(sdf), _lib.ptr(grad) if return_grads else None,
                                      _lib.ptr(mask) if return_overlapmask else None, _lib.stream_ptr(pts.device))
        _lib.check(rc, "dss_sdf_query")
    out = (sdf,) + ((grad,) if return_grads else ()) + ((mask.bool(),) if return_overlapmask else ())
    return out if len(out) > 1 else out[0]


def grid_sdf_query(grid, scale, pts, return_grads=True, return_overlapmask=False):
    """`SDF3D.query_sdfs` of an SDFGrid3D body (bodies.py:203-241, 763-775): grid [n0,n1,n2] over the unit cube."""
    pts, grid = _dev(pts), _dev(grid)
    n = pts.shape[0]
    sdf = torch.empty(n, dtype=torch.float64, device=pts.device)
    grad = torch.empty(n, 3, dtype=torch.float64, device=pts.device) if return_grads else None
    mask = torch.empty(n, dtype=torch.uint8, device=pts.device) if return_overlapmask else None
    if n:
        rc = _lib.lib().dss_grid_sdf_query(_lib.ptr(grid), int(grid.shape[0]), int(grid.shape[1]), int(grid.shape[2]),
                                           _lib.ctypes.c_double(float(scale)), _lib.ptr(pts), int(n), _lib.ptr(sdf),
                                           _lib.ptr(grad) if return_grads else None,
                                           _lib.ptr(mask) if return_overlapmask else None, _lib.stream_ptr(pts.device))
        _lib.check(rc, "dss_grid_sdf_query")
    out = (sdf,) + ((grad,) if return_grads else ()) + ((mask.bool(),) if return_overlapmask else ())
    return out if len(out) > 1 else out[0]


def mesh_inertia(verts, faces, mass, return_volume=False):
    """One closed triangle mesh (verts [V,3], faces [F,3]) or a list of them -> inertia tensor(s) [3,3] about the
    mesh origin for the given mass(es) at uniform density."""
    single = not isinstance(verts, (list, tuple))
    vs = [verts] if single else list(verts)
    fs = [faces] if single else list(faces)
    ms = [mass] if single else list(mass)
    voff = np.cumsum([0] + [len(v) for v in vs[:-1]]).astype(np.int32)
    foff = np.cumsum([0] + [len(f) for f in fs[:-1]]).astype(np.int32)
    nf = np.array([len(f) for f in fs], np.int32)
    def pooled(xs, dtype):      # one pooled table on the device; tensors that are already there are not taken through the host
        ts = [torch.as_tensor(x).detach().to(dtype) for x in xs]
        if len(ts) == 1:
            return _dev(ts[0], dtype)
        dev = next((t.device for t in ts if t.is_cuda), None)
        return _dev(torch.cat([t.to(dev) if dev is not None else t for t in ts]), dtype)
    V, F = pooled(vs, torch.float64), pooled(fs, torch.int32)
    M = _dev(torch.tensor([float(m) for m in ms], dtype=torch.float64))
    d = V.device
    ti = lambda a: torch.as_tensor(a, dtype=torch.int32).to(d)
    voff_t, foff_t, nf_t = ti(voff), ti(foff), ti(nf)
    J = torch.empty(len(vs), 9, dtype=torch.float64, device=d)
    vol = torch.empty(len(vs), dtype=torch.float64, device=d)
    rc = _lib.lib().dss_mesh_inertia(_lib.ptr(V), _lib.ptr(F), _lib.ptr(voff_t), _lib.ptr(foff_t), _lib.ptr(nf_t), len(vs),
                                     _lib.ptr(M), _lib.ptr(J), _lib.ptr(vol), _lib.stream_ptr(d))
    _lib.check(rc, "dss_mesh_inertia")
    J = J.reshape(-1, 3, 3)
    if single:
        return (J[0], vol[0]) if return_volume else J[0]
    return (J, vol) if return_volume else J


class _MeshInertiaFn(torch.autograd.Function):
    """J(verts, mass) of one closed triangle mesh, differentiable like the reference's get_ang_inertia
    (`bodies.py:380-395`, autograd there): backward = dss_mesh_inertia_backward."""

    @staticmethod
    def forward(ctx, verts, faces, mass):
        V = _dev(verts.detach())
        F = _dev(faces, torch.int32)
        J = mesh_inertia(V, F, float(mass))
        ctx.save_for_backward(V, F, J)
        ctx.mass, ctx.vdev, ctx.mdev = float(mass), verts.device, mass.device if torch.is_tensor(mass) else None
        return J.to(verts.device)

    @staticmethod
    def backward(ctx, gJ):
        V, F, J = ctx.saved_tensors
        g = _dev(gJ).reshape(9)
        gv = torch.empty_like(V)
        rc = _lib.lib().dss_mesh_inertia_backward(_lib.ptr(V), _lib.ptr(F), int(V.shape[0]), int(F.shape[0]),
                                                  _lib.ctypes.c_double(ctx.mass), _lib.ptr(g), _lib.ptr(gv), _lib.stream_ptr(V.device))
        _lib.check(rc, "dss_mesh_inertia_backward")
        gm = None if ctx.mdev is None else ((g.reshape(3, 3) * J).sum() / ctx.mass).to(ctx.mdev)     # J is linear in the mass
        return gv.to(ctx.vdev), None, gm


def mesh_inertia_diff(verts, faces, mass):
    """Differentiable (w.r.t. verts and mass) inertia tensor of one closed mesh; verts a torch tensor [V,3]."""
    return _MeshInertiaFn.apply(verts, faces, mass if torch.is_tensor(mass) else torch.tensor(float(mass), dtype=torch.float64))
