"""Differentiable marching cubes (MeshSDF) on the device -- mirror of ``SDF3D._diff_marching_cubes``
(`sdf_physics/physics3d/bodies.py:653-704`).

``marching_cubes(phi, iso)``            grid of SDF samples -> indexed triangle mesh (dss_mc_count / dss_mc_emit)
``primitive_mesh(shape_type, prm_unit)`` the reference's ``MeshSDF.apply(*params)`` for box / sphere / cylinder:
                                        samples the unit SDF on linspace(-1, 1, res)^3 (dss_sdf_query), meshes it and
                                        returns vertices in [-1, 1]^3; an autograd Function whose backward is the
                                        MeshSDF formula (dss_meshsdf_backward)
``igr_mesh(latent, weights)``           the same for the IGR network (forward only), samples through dss_igr_query.

The case tables come from ``mc_tables`` (generated, not transcribed).  Vertex / face order is this library's own:
the reference's extension (ev_sdf_utils) is un-vendored and its order implementation-defined, so
parity is held on properties (closed, oriented, on the level set) and against the numpy restatement of the same tables.
"""
import numpy as np
import torch

from . import _lib, mc_tables
from .mass_properties import _dev, sdf_query

_TAB = {}


def _tables(device):
    key = str(device)
    if key not in _TAB:
        ntri, tri, _ = mc_tables.tables()
        _TAB[key] = (torch.tensor(ntri, dtype=torch.int32, device=device), torch.tensor(tri, dtype=torch.int8, device=device))
    return _TAB[key]


def marching_cubes(phi, iso=0.0):
    """phi [n0,n1,n2] float64 (x slowest) -> verts [V,3] in grid-index units, faces [F,3] int32 (device tensors)."""
    phi = _dev(phi)
    n0, n1, n2 = phi.shape
    L = _lib.lib()
    L.dss_mc_workspace_bytes.restype = _lib.ctypes.c_size_t
    nbytes = L.dss_mc_workspace_bytes(n0, n1, n2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=phi.device)
    tot = torch.zeros(2, dtype=torch.int32, device=phi.device)
    ntri, tri = _tables(phi.device)
    st = _lib.stream_ptr(phi.device)
    _lib.check(L.dss_mc_count(_lib.ptr(phi), n0, n1, n2, _lib.ctypes.c_double(iso), _lib.ptr(ntri), _lib.ptr(ws),
                              _lib.ctypes.c_size_t(nbytes), _lib.ptr(tot), st), "dss_mc_count")
    nv, nf = [int(x) for x in tot.cpu()]          # the one host read: sizes of the outputs
    verts = torch.empty(max(nv, 1), 3, dtype=torch.float64, device=phi.device)
    faces = torch.empty(max(nf, 1), 3, dtype=torch.int32, device=phi.device)
    _lib.check(L.dss_mc_emit(_lib.ptr(phi), n0, n1, n2, _lib.ctypes.c_double(iso), _lib.ptr(ntri), _lib.ptr(tri),
                             mc_tables.MAX_TRI, _lib.ptr(ws), _lib.ptr(verts), _lib.ptr(faces), st), "dss_mc_emit")
    return verts[:nv], faces[:nf]


def _grid(res, device):
    g = torch.linspace(-1.0, 1.0, res, dtype=torch.float64, device=device)
    return torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=3).reshape(-1, 3)


_GRIDS = {}
GRID_EVENTS = None      # a list: _IgrMeshSDF.forward appends (start event, stop event, points) per network grid it evaluates


def _grid_cached(res, device):
    key = (int(res), str(device))
    if key not in _GRIDS:
        _GRIDS[key] = _grid(res, device)
    return _GRIDS[key]


class _PrimitiveMeshSDF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prm_unit, shape_type, res):
        p = prm_unit.detach().to(torch.float64).cpu()
        dev = torch.device("cuda")
        # the unit parameters of a body span the unit cube (their own scale is 1 up to rounding), so the body-frame
        # query of a shape with these parameters at the unit samples is the reference's sdf_func(samples, *params)
        sdf = sdf_query(shape_type, p, _grid(res, dev), return_grads=False)
        verts, faces = marching_cubes(sdf.reshape(res, res, res), 0.0)
        verts = verts / (res - 1) * 2.0 - 1.0
        ctx.save_for_backward(verts, p)
        ctx.shape_type = shape_type
        ctx.prm_device = prm_unit.device
        ctx.mark_non_differentiable(faces)
        return verts, faces

    @staticmethod
    def backward(ctx, grad_v, _grad_f):
        verts, p = ctx.saved_tensors
        out = torch.empty(4, dtype=torch.float64, device=verts.device)
        prm_h = np.zeros(4); prm_h[: p.numel()] = p.numpy().reshape(-1)
        rc = _lib.lib().dss_meshsdf_backward(int(ctx.shape_type), prm_h.ctypes.data_as(_lib.ctypes.c_void_p), _lib.ptr(verts),
                                             _lib.ptr(grad_v.contiguous()), int(verts.shape[0]), _lib.ptr(out),
                                             _lib.stream_ptr(verts.device))
        _lib.check(rc, "dss_meshsdf_backward")
        return out[: p.numel()].reshape(p.shape).to(ctx.prm_device), None, None


def primitive_mesh(shape_type, prm_unit, res=128):
    """Unit-frame mesh of an analytic primitive, differentiable w.r.t. its unit parameters (dims/scale; rad/scale;
    rad/scale, height/scale; r/scale, d/scale; for a rounded box / brick dims/scale and, fourth, the corner radius/scale), as the reference's `self._diff_marching_cubes(self.sdf_func)(*self.params)`."""
    return _PrimitiveMeshSDF.apply(torch.as_tensor(prm_unit, dtype=torch.float64), int(shape_type), int(res))


class _IgrMeshSDF(torch.autograd.Function):
    """MeshSDF for the IGR network: forward = res^3 evaluations on the fp64 matrix cores + marching cubes; backward =
    dL/dlatent = sum_v -(dL/dv . n_v) d phi / d latent (v)  (bodies.py:680-702), normals and latent derivatives from
    two more network evaluations at the vertices (dss_igr_query, dss_igr_query_latent_grad)."""

    @staticmethod
    def forward(ctx, latent, packed_weights, res):
        from .igr import igr_values
        dev = packed_weights["W0"].device
        lat = _dev(latent.detach())
        ev = GRID_EVENTS
        if ev is not None:      # (bench.py: HIP events around the grid evaluation, the matrix-core share of a mesh build)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        sdf = igr_values(_grid_cached(res, dev), lat, packed_weights)      # values only: a quarter of the matrix work of value + gradient
        if ev is not None:
            b.record()
            ev.append((a, b, res ** 3))
        verts, faces = marching_cubes(sdf.reshape(res, res, res), 0.0)
        verts = verts / (res - 1) * 2.0 - 1.0
        ctx.save_for_backward(verts, lat)
        ctx.P, ctx.ldev = packed_weights, latent.device
        ctx.mark_non_differentiable(faces)
        return verts, faces

    @staticmethod
    def backward(ctx, grad_v, _grad_f):
        from .igr import igr_query
        verts, lat = ctx.saved_tensors
        _, gx = igr_query(verts, lat, ctx.P)                    # d phi / d xyz at the vertices
        _, gl = igr_query(verts, lat, ctx.P, wrt="latent")      # d phi / d latent at the vertices
        nrm = gx / gx.norm(dim=1, keepdim=True).clamp_min(1e-12)
        dl_ds = -(grad_v * nrm).sum(1)
        return (dl_ds[:, None] * gl[:, :2]).sum(0).to(ctx.ldev), None, None


def igr_mesh(latent, packed_weights, res=128):
    """Mesh of the IGR level set in [-1,1]^3, differentiable w.r.t. the latent code (MeshSDF)."""
    return _IgrMeshSDF.apply(torch.as_tensor(latent, dtype=torch.float64), packed_weights, int(res))
