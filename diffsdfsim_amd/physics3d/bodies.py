"""SDF rigid bodies of the 3-D layer (host-side mirror of sdf_physics/physics3d/bodies.py:402-1009).

Same constructor arguments and attributes as the reference (``p``, ``pos``, ``rot``, ``v``, ``mass``, ``verts``,
``faces``, ``dims``/``rad``, ``restitution``, ``fric_coeff``, ``add_force``, ``add_no_contact``, ``query_sdfs``).
Parameters are torch tensors and may require grad; geometry kernels run on the HIP device.
``custom_mesh=True`` gives the reference's analytic meshes; ``custom_mesh=False`` meshes the unit SDF with
marching cubes on the device like ``SDF3D._create_mesh`` (bodies.py:706-711); the mesh stays differentiable w.r.t. the
body's parameters (MeshSDF backward) and the stepper returns the adjoint of its vertices.  ``custom_inertia=False`` integrates the mesh (dss_mesh_inertia).
"""
import math

import numpy as np
import torch

from .. import mass_properties, meshes, meshsdf, world_abi as abi
from .utils import Defaults3D, get_tensor


def _quat_from_euler(v):
    p, t, s = [0.5 * float(x) for x in v]
    return [math.cos(p) * math.cos(t) * math.cos(s) + math.sin(p) * math.sin(t) * math.sin(s),
            math.sin(p) * math.cos(t) * math.cos(s) - math.cos(p) * math.sin(t) * math.sin(s),
            math.cos(p) * math.sin(t) * math.cos(s) + math.sin(p) * math.cos(t) * math.sin(s),
            math.cos(p) * math.cos(t) * math.sin(s) - math.sin(p) * math.sin(t) * math.cos(s)]


class Body3D:
    shape_type = None

    def __init__(self, pos, vel=(0, 0, 0, 0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION,
                 fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, **_render_kwargs):
        pos = get_tensor(pos)
        if pos.numel() == 3:
            self.p = torch.cat([pos.new_tensor([1.0, 0, 0, 0]), pos])
        elif pos.numel() == 6:
            self.p = torch.cat([pos.new_tensor(_quat_from_euler(pos[:3])), pos[3:]])
        else:
            self.p = pos
        vel = get_tensor(vel)
        self.v = torch.cat([vel.new_zeros(3), vel]) if vel.numel() == 3 else vel
        self.mass = get_tensor(mass)
        self.restitution = get_tensor(restitution)
        self.fric_coeff = get_tensor(fric_coeff)
        self.eps = eps
        self.forces = []
        self.no_contact = set()
        self.ang_inertia = self._get_ang_inertia(self.mass)

    rot = property(lambda self: self.p[:4])
    pos = property(lambda self: self.p[4:])

    def set_p(self, new_p, update_geom_rotation=True, update_ang_inertia=True):
        """`Body3D.set_p` (sdf_physics/physics3d/bodies.py:498-511): replace the pose (quaternion wxyz + position).  A
        world that owns the body picks the new tensor up at its next step."""
        self.p = get_tensor(new_p)

    @property
    def M(self):
        """6x6 generalized mass matrix blockdiag(R I R^T, m 1) at the current pose (bodies.py:431-435, 509-511)."""
        q = self.p[:4]
        two_s = 2.0 / (q * q).sum()
        r, i, j, k = q[0], q[1], q[2], q[3]
        R = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                         two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                         two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)]).reshape(3, 3)
        Iw = R @ self.ang_inertia.to(R) @ R.t()
        return torch.block_diag(Iw, torch.eye(3, dtype=Iw.dtype, device=Iw.device) * self.mass.to(Iw))

    def add_force(self, f):
        self.forces.append(f)
        f.set_body(self)

    def add_no_contact(self, other):
        self.no_contact.add(other)
        other.no_contact.add(self)

    def apply_forces(self, t):
        if not self.forces:
            return torch.zeros(6, dtype=torch.float64)
        return sum(f.force(t) for f in self.forces)

    def query_sdfs(self, pts_loc, return_grads=True, return_overlapmask=False):
        """`SDF3D.query_sdfs` (sdf_physics/physics3d/bodies.py:721-760) on the device: body-frame points ->
        sdf (, normalised gradient) (, overlap mask).  Values only: the stepper differentiates contact geometry in
        its own backward kernels, not through this query."""
        prm = torch.cat([self.shape_prm().detach().reshape(-1).to(torch.float64), torch.tensor([self.shape_aux()], dtype=torch.float64)])
        return mass_properties.sdf_query(self.shape_type, prm, pts_loc, return_grads, return_overlapmask)

    def shape_aux(self):
        """Constant fourth shape parameter (corner radius of SDFBoxRounded / SDFBrick); 0 for the other bodies."""
        return 0.0

    def _marching_cubes_mesh(self):
        """`SDF3D._create_mesh` (bodies.py:706-711): unit SDF on 128^3 -> marching cubes -> vertices * scale."""
        # kept differentiable: unit parameters and scale are functions of the body's parameters, the mesher's backward is
        # the MeshSDF rule (bodies.py:680-702); `verts_t` is what a world hands to the stepper as a differentiable input
        prm = self.shape_prm().reshape(-1).to(torch.float64)
        unit = torch.cat([prm, torch.tensor([self.shape_aux()], dtype=torch.float64)]) / self.scale
        v, f = meshsdf.primitive_mesh(self.shape_type, unit, res=128)
        self.verts_t = v * self.scale.to(v.device)
        v = self.verts_t.detach().cpu().numpy()
        return v, f.cpu().numpy().astype(np.int64), np.zeros_like(v)

    def _mesh_ang_inertia(self, mass):
        """`SDF3D._get_ang_inertia` (bodies.py:713-714): volume integrals of the body's own mesh (values only)."""
        vt = getattr(self, "verts_t", None)
        if vt is not None and vt.requires_grad:       # differentiable like the reference's get_ang_inertia (autograd there)
            return mass_properties.mesh_inertia_diff(vt, torch.as_tensor(self.faces_np), mass).cpu()
        m = torch.as_tensor(mass)
        if m.requires_grad:      # the integrals are linear in the mass (system identification: d / d mass through the inertia)
            return mass_properties.mesh_inertia(self.verts_np, self.faces_np, 1.0).cpu() * m.cpu().to(torch.float64)
        return mass_properties.mesh_inertia(self.verts_np, self.faces_np, float(m.detach())).cpu()


class SDFBox(Body3D):
    shape_type = abi.SHAPE_BOX

    def __init__(self, pos, dims, vel=(0, 0, 0, 0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION,
                 fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, custom_mesh=Defaults3D.CUSTOM_MESH,
                 custom_inertia=Defaults3D.CUSTOM_INERTIA, **kw):
        self.custom_inertia = custom_inertia
        self.dims = get_tensor(dims)
        self.scale = torch.max(self.dims) * 1.5 / 2
        if custom_mesh:
            v, f, tie = meshes.box_mesh(self.dims.detach().cpu().numpy())
            self.verts_np, self.faces_np, self.vgrad_np = v, f, 0.5 * tie
        else:
            self.verts_np, self.faces_np, self.vgrad_np = self._marching_cubes_mesh()
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return self.dims

    def _get_ang_inertia(self, mass):   # bodies.py:796-797
        if not self.custom_inertia:
            return self._mesh_ang_inertia(mass)
        d = self.dims
        return mass * torch.diag(torch.stack([d[1] ** 2 + d[2] ** 2, d[0] ** 2 + d[2] ** 2, d[0] ** 2 + d[1] ** 2])) / 12


class SDFSphere(Body3D):
    shape_type = abi.SHAPE_SPHERE

    def __init__(self, pos, rad, vel=(0, 0, 0, 0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION,
                 fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, custom_mesh=Defaults3D.CUSTOM_MESH,
                 custom_inertia=Defaults3D.CUSTOM_INERTIA, **kw):
        self.custom_inertia = custom_inertia
        self.rad = get_tensor(rad)
        self.scale = self.rad * 1.5
        if custom_mesh:
            uv, uf = meshes.icosphere(4)
            self.verts_np, self.faces_np, self.vgrad_np = uv * float(self.rad.detach()), uf, uv
        else:
            self.verts_np, self.faces_np, self.vgrad_np = self._marching_cubes_mesh()
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return torch.cat([self.rad.reshape(1), self.rad.new_zeros(2)])

    def _get_ang_inertia(self, mass):   # bodies.py:993-994
        if not self.custom_inertia:
            return self._mesh_ang_inertia(mass)
        return 2.0 / 5.0 * mass * self.rad ** 2 * torch.eye(3, dtype=torch.float64)


class SDFCylinder(Body3D):
    """`sdf_physics/physics3d/bodies.py:907-976`: cylinder along the body z axis (custom mesh / inertia)."""
    shape_type = abi.SHAPE_CYLINDER

    def __init__(self, pos, rad, height, vel=(0, 0, 0, 0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION,
                 fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, custom_mesh=Defaults3D.CUSTOM_MESH,
                 custom_inertia=Defaults3D.CUSTOM_INERTIA, **kw):
        self.custom_inertia = custom_inertia
        self.rad, self.height = get_tensor(rad), get_tensor(height)
        self.scale = torch.max(self.rad, self.height / 2) * 1.5
        if custom_mesh:
            self.verts_np, self.faces_np, self.vgrad_np = meshes.cylinder_mesh(float(self.rad.detach()), float(self.height.detach()))
        else:
            self.verts_np, self.faces_np, self.vgrad_np = self._marching_cubes_mesh()
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return torch.stack([self.rad.reshape(()), self.height.reshape(()), self.rad.new_zeros(())])

    def _get_ang_inertia(self, mass):   # bodies.py:925-927
        if not self.custom_inertia:
            return self._mesh_ang_inertia(mass)
        a = (3 * self.rad ** 2 + self.height ** 2) / 12
        return mass * torch.diag(torch.stack([a, a, self.rad ** 2 / 2]))


class _LevelSetBox(Body3D):
    """Box-family bodies the reference always meshes with marching cubes and integrates numerically (no custom mesh /
    inertia options, bodies.py:857-885)."""

    def __init__(self, pos, dims, r, vel=(0, 0, 0, 0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION,
                 fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, **kw):
        self.dims = get_tensor(dims)
        self.r = get_tensor(r)
        self.scale = torch.max(self.dims) * 1.5 / 2
        self.verts_np, self.faces_np, self.vgrad_np = self._marching_cubes_mesh()
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return self.dims

    def shape_aux(self):
        return float(self.r.detach())

    def _get_ang_inertia(self, mass):
        return self._mesh_ang_inertia(mass)


class SDFBoxRounded(_LevelSetBox):
    """`sdf_physics/physics3d/bodies.py:857-870`: box of outer size ``dims`` whose edges and corners are rounded with
    radius ``r`` (box of dims - 2 r, offset by r)."""
    shape_type = abi.SHAPE_BOX_ROUNDED


class SDFBrick(_LevelSetBox):
    """`sdf_physics/physics3d/bodies.py:873-885`: box whose four edges along z are rounded with radius ``r``."""
    shape_type = abi.SHAPE_BRICK


class SDFBowl(Body3D):
    """`sdf_physics/physics3d/bodies.py:1013-1065`: hemispherical shell (mid radius ``r``, half thickness ``d``) opening
    towards the body's +z; inertia from the mesh."""
    shape_type = abi.SHAPE_BOWL

    def __init__(self, pos, r, d, vel=(0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION, fric_coeff=Defaults3D.FRIC_COEFF,
                 eps=Defaults3D.EPSILON, custom_mesh=Defaults3D.CUSTOM_MESH, **kw):
        self.r, self.d = get_tensor(r), get_tensor(d)
        self.scale = (self.r + self.d) * 1.3333
        if custom_mesh:
            v, f = meshes.bowl_mesh(float(self.r.detach()), float(self.d.detach()))
            self.verts_np, self.faces_np, self.vgrad_np = v, f, np.zeros_like(v)
        else:
            self.verts_np, self.faces_np, self.vgrad_np = self._marching_cubes_mesh()
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return torch.stack([self.r.reshape(()), self.d.reshape(()), self.r.new_zeros(())])

    def _get_ang_inertia(self, mass):
        return self._mesh_ang_inertia(mass)


class SDF3D(Body3D):
    """`sdf_physics/physics3d/bodies.py:627-760`: a body given by an SDF function over its unit cube, ``scale`` mapping
    the cube to world units -- ``SDF3D(pos, scale, sdf_func=decode_igr(network), params=[latent])`` as in
    demos/demo_meshsdf.py:136.  On the device path the function must be a neural SDF made by ``decode_igr`` (the network
    runs on the matrix cores inside the stepper); the analytic primitives have their own classes.  Mesh: level set of the
    network on 128^3 (MeshSDF, differentiable w.r.t. the latent code); inertia from the mesh."""
    shape_type = abi.SHAPE_IGR

    def __init__(self, pos, scale, sdf_func, params, grad_func=None, vel=(0, 0, 0, 0, 0, 0), mass=1,
                 restitution=Defaults3D.RESTITUTION, fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, res=128, **kw):
        net = getattr(sdf_func, "igr", None)
        if net is None or grad_func is not None:
            raise NotImplementedError("SDF3D on the device path takes sdf_func=decode_igr(network) (a neural SDF the kernels "
                                      "can evaluate); arbitrary Python SDF functions cannot run inside the HIP stepper")
        if len(params) != 1 or torch.as_tensor(params[0]).numel() != 2:
            raise NotImplementedError("decode_igr networks with a 2-dimensional latent code (bob_spot_setup) are built")
        self.igr = net
        self.sdf_func, self.params = sdf_func, params
        self.latent = get_tensor(params[0])
        self.scale = get_tensor(scale)
        v, f = meshsdf.igr_mesh(self.latent, net.packed, res=res)
        self.verts_t = v * self.scale.to(v.device)
        self.verts_np = self.verts_t.detach().cpu().numpy()
        self.faces_np = f.cpu().numpy().astype(np.int64)
        self.vgrad_np = np.zeros_like(self.verts_np)
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return torch.cat([self.latent.reshape(2).to(torch.float64), self.latent.new_zeros(1).to(torch.float64)])

    def shape_aux(self):
        return float(self.scale.detach())

    def query_sdfs(self, pts_loc, return_grads=True, return_overlapmask=False):
        """bodies.py:721-760 with the network on the device: values scaled, gradients normalised, outside the query cube
        phi = scale and grad = 0."""
        from ..igr import igr_query
        dev = self.igr.packed["W0"].device
        pts = torch.as_tensor(pts_loc, dtype=torch.float64).to(dev)
        sc = float(self.scale.detach())
        mask = (pts.abs() <= sc).all(dim=1)
        sdf = torch.full((pts.shape[0],), sc, dtype=torch.float64, device=dev)
        grad = torch.zeros_like(pts)
        if bool(mask.any()):
            phi, g = igr_query((pts[mask] / sc).contiguous(), self.latent.detach().to(dev), self.igr.packed)
            sdf[mask] = phi * sc
            grad[mask] = g / g.norm(dim=1, keepdim=True).clamp_min(1e-12)
        out = (sdf,) + ((grad,) if return_grads else ()) + ((mask,) if return_overlapmask else ())
        return out if len(out) > 1 else out[0]

    def _get_ang_inertia(self, mass):
        return self._mesh_ang_inertia(mass)


class SDFGrid3D(Body3D):
    """`sdf_physics/physics3d/bodies.py:763-775`: the SDF is a voxel grid ``sdf`` [n,n,n] over the body's unit cube
    (``scale`` maps it to world units).  Mesh: marching cubes of the grid itself at its own resolution; inertia from
    the mesh; queries through dss_grid_sdf_query; inside the stepper the kernels interpolate the same grid (geom.h: SHAPE_GRID)."""
    shape_type = abi.SHAPE_GRID

    def __init__(self, pos, scale, sdf, vel=(0, 0, 0), mass=1, restitution=Defaults3D.RESTITUTION,
                 fric_coeff=Defaults3D.FRIC_COEFF, eps=Defaults3D.EPSILON, **kw):
        self.scale = get_tensor(scale)
        self.sdf = torch.as_tensor(sdf, dtype=torch.float64)
        res = self.sdf.shape[0]
        v, f = meshsdf.marching_cubes(self.sdf, 0.0)
        v = (v / (res - 1) * 2.0 - 1.0) * float(self.scale.detach())
        self.verts_np, self.faces_np = v.cpu().numpy(), f.cpu().numpy().astype(np.int64)
        self.vgrad_np = np.zeros_like(self.verts_np)
        super().__init__(pos, vel, mass, restitution, fric_coeff, eps, **kw)

    verts = property(lambda self: torch.as_tensor(self.verts_np))
    faces = property(lambda self: torch.as_tensor(self.faces_np))

    def shape_prm(self):
        return torch.zeros(3, dtype=torch.float64)      # the shape is the grid itself (World3D hands it to the engine)

    def shape_aux(self):
        return float(self.scale.detach())

    def query_sdfs(self, pts_loc, return_grads=True, return_overlapmask=False):
        return mass_properties.grid_sdf_query(self.sdf, float(self.scale.detach()), pts_loc, return_grads, return_overlapmask)

    def _get_ang_inertia(self, mass):
        return self._mesh_ang_inertia(mass)
