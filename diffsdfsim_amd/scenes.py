"""Synthetic scene batches of BASELINE.json's configs (SURVEY.md §8d recipes), as BatchEngine specs.

config 2: sphere drop  -- floor SDFBox([0,-.5,0],[20,1,20]) pinned, SDFSphere r~U(.4,.6) from y~U(.7,1.2),
          v_x~U(0,1), gravity 10, restitution .5, mu .25
config 3: box stack    -- floor + 7 SDFBox dims~U(.9,1.1)^3 stacked with gap 5e-4 (< eps), lateral offsets
          ~U(-.05,.05), small yaw, mu .5, restitution 0
All bodies use the analytic (custom_mesh / custom_inertia) meshes and inertias of the reference
(bodies.py:796-854, 993-1009).  Data is synthetic and seeded; there is nothing to download.
"""
import math

import numpy as np

from . import meshes
from . import world_abi as abi


def _quat_from_euler(phi, the, psi):
    """utils.py:207-221 (`quat`, wxyz)."""
    p, t, s = 0.5 * phi, 0.5 * the, 0.5 * psi
    return np.array([math.cos(p) * math.cos(t) * math.cos(s) + math.sin(p) * math.sin(t) * math.sin(s),
                     math.sin(p) * math.cos(t) * math.cos(s) - math.cos(p) * math.sin(t) * math.sin(s),
                     math.cos(p) * math.sin(t) * math.cos(s) + math.sin(p) * math.cos(t) * math.sin(s),
                     math.cos(p) * math.cos(t) * math.sin(s) - math.sin(p) * math.sin(t) * math.cos(s)])


def box_inertia(m, d):
    return m * np.diag([d[1] ** 2 + d[2] ** 2, d[0] ** 2 + d[2] ** 2, d[0] ** 2 + d[1] ** 2]) / 12.0


def _base(B, nb):
    z = lambda *s: np.zeros((B, nb) + s)
    spec = dict(pose=z(7), vel=z(6), mass=np.ones((B, nb)), inertia=z(3, 3), restitution=z(), fric=z(), fext=z(6),
                shape_type=np.zeros((B, nb), np.int32), shape_prm=z(3), mesh_id=np.zeros((B, nb), np.int32), meshes=[],
                mesh_vgrad=[])
    spec["pose"][:, :, 0] = 1.0
    Je = np.zeros((B, 6, 6 * nb))
    Je[:, :, :6] = np.eye(6)          # TotalConstraint3D on body 0 (the floor)
    spec["Je"] = Je
    return spec


def _add_mesh(spec, cache, key, make):
    if key not in cache:
        verts, faces, vgrad = make()
        cache[key] = len(spec["meshes"])
        spec["meshes"].append((verts, faces))
        spec["mesh_vgrad"].append(vgrad)
    return cache[key]


def _floor(spec, cache, floor_dims, mu, rest):
    fd = np.asarray(floor_dims, np.float64)
    def make():
        v, f, tie = meshes.box_mesh(fd)
        return v, f, 0.5 * tie
    mid = _add_mesh(spec, cache, ("box",) + tuple(fd), make)
    spec["pose"][:, 0, 4:] = (0.0, -fd[1] / 2, 0.0)
    spec["shape_prm"][:, 0] = fd
    spec["mesh_id"][:, 0] = mid
    spec["inertia"][:, 0] = box_inertia(1.0, fd)
    spec["fric"][:, 0] = mu
    spec["restitution"][:, 0] = rest


def box_stack(B, nbox=7, seed=0, floor_dims=(20.0, 1.0, 20.0), mu=0.5, rest=0.0, gap=5e-4, g=10.0, push=0.0):
    """config 3.  Returns a BatchEngine spec (numpy)."""
    r = np.random.default_rng(seed)
    nb = nbox + 1
    spec, cache = _base(B, nb), {}
    _floor(spec, cache, floor_dims, mu, rest)
    for s in range(B):
        y = 0.0
        for k in range(1, nb):
            dims = 0.9 + 0.2 * r.random(3)
            off = -0.05 + 0.1 * r.random(2)
            yaw = 0.2 * r.random()
            def make(dims=dims):
                v, f, tie = meshes.box_mesh(dims)
                return v, f, 0.5 * tie
            spec["mesh_id"][s, k] = _add_mesh(spec, cache, ("box",) + tuple(dims), make)
            spec["pose"][s, k, :4] = _quat_from_euler(0.0, yaw, 0.0)
            spec["pose"][s, k, 4:] = (off[0], y + gap + dims[1] / 2, off[1])
            spec["vel"][s, k, 3] = push * (r.random() - 0.5)
            spec["shape_prm"][s, k] = dims
            spec["inertia"][s, k] = box_inertia(1.0, dims)
            spec["fric"][s, k] = mu
            spec["restitution"][s, k] = rest
            spec["fext"][s, k, 4] = -g * spec["mass"][s, k]      # Gravity3D (forces.py:69-85)
            y += gap + dims[1]
    return spec


def sphere_drop(B, seed=0, floor_dims=(20.0, 1.0, 20.0), mu=0.25, rest=0.5, g=10.0):
    """config 2."""
    r = np.random.default_rng(seed)
    spec, cache = _base(B, 2), {}
    _floor(spec, cache, floor_dims, mu, rest)
    uv, uf = meshes.icosphere(4)
    for s in range(B):
        rad = 0.4 + 0.2 * r.random()
        y0 = 0.7 + 0.5 * r.random()
        vx = r.random()
        spec["mesh_id"][s, 1] = _add_mesh(spec, cache, ("sphere", rad), lambda rad=rad: (uv * rad, uf, uv.copy()))
        spec["shape_type"][s, 1] = abi.SHAPE_SPHERE
        spec["shape_prm"][s, 1, 0] = rad
        spec["pose"][s, 1, 4:] = (0.0, y0, 0.0)
        spec["vel"][s, 1, 3] = vx
        spec["inertia"][s, 1] = 0.4 * rad * rad * np.eye(3)
        spec["fric"][s, 1] = mu
        spec["restitution"][s, 1] = rest
        spec["fext"][s, 1, 4] = -g
    return spec


def igr_pole(B, seed=0, packed=None, radius_init=0.5, latent_sigma=0.1, y0=6.0, res=128, mu=0.15, g=10.0, floor_dims=(50.0, 1.0, 50.0)):
    """config 4 (BASELINE configs[3]): ``demos/demo_meshsdf.make_world`` (demo_meshsdf.py:121-142) for a batch -- level-set floor
    50 x 1 x 50, pinned cylinder pole (r 0.2, h 2, upright at x = 0.35, no contact with the floor), one neural-SDF body of scale
    2 per scene dropped from y = 6 with its own latent code ~ N(0, sigma^2).  The trained IGR weights are not available
    offline: ``packed`` defaults to seeded geometric-init weights of the bob_spot_setup shape (synthetic, like all data here).
    Needs the HIP device: the level-set meshes and their inertias are built by the device operators."""
    import torch
    from . import igr, mass_properties, meshsdf
    r = np.random.default_rng(seed)
    if packed is None:
        packed = igr.pack_weights(*geometric_init_weights(seed, radius_init))
    nb = 3
    spec, cache = _base(B, nb), {}
    spec["Je"] = np.zeros((B, 12, 6 * nb))
    spec["Je"][:, :6, :6] = np.eye(6)            # TotalConstraint3D(floor), TotalConstraint3D(pole)
    spec["Je"][:, 6:, 6:12] = np.eye(6)
    nocon = np.zeros((nb, nb), np.uint8); nocon[0, 1] = nocon[1, 0] = 1
    spec["no_contact"] = nocon
    spec["shape_aux"] = np.zeros((B, nb))
    spec["igr_net"] = packed
    fd = np.asarray(floor_dims, np.float64)

    def level_set(kind, prm, scale):
        v, f = meshsdf.primitive_mesh(kind, np.concatenate([prm, [0.0]]) / scale, res=res)
        v = (v * scale).cpu().numpy(); f = f.cpu().numpy()
        return v, f, np.asarray(mass_properties.mesh_inertia(v, f, 1.0).cpu())
    fs = fd.max() * 1.5 / 2
    fv, ff, fJ = level_set(abi.SHAPE_BOX, fd, fs)
    prm_p = np.array([0.2, 2.0, 0.0]); ps = max(prm_p[0], prm_p[1] / 2) * 1.5
    pv, pf, pJ = level_set(abi.SHAPE_CYLINDER, prm_p, ps)
    spec["meshes"] += [(fv, ff), (pv, pf)]
    spec["mesh_vgrad"] += [np.zeros_like(fv), np.zeros_like(pv)]
    spec["pose"][:, 0, 4:] = (0.0, -fd[1] / 2, 0.0)
    spec["shape_prm"][:, 0] = fd; spec["inertia"][:, 0] = fJ; spec["fric"][:, 0] = mu; spec["restitution"][:, 0] = 0.0
    spec["shape_type"][:, 1] = abi.SHAPE_CYLINDER
    spec["pose"][:, 1, :4] = _quat_from_euler(math.pi / 2, 0.0, 0.0)
    spec["pose"][:, 1, 4:] = (0.35, 1.0, 0.0)
    spec["shape_prm"][:, 1] = prm_p; spec["inertia"][:, 1] = pJ; spec["fric"][:, 1] = mu
    spec["restitution"][:, 1] = 0.5              # Defaults3D.RESTITUTION (the demo does not set it for the pole)
    spec["mesh_id"][:, 1] = 1
    spec["shape_type"][:, 2] = abi.SHAPE_IGR
    spec["shape_aux"][:, 2] = 2.0
    for s in range(B):
        lat = latent_sigma * r.standard_normal(2)
        v, f = meshsdf.igr_mesh(torch.tensor(lat, dtype=torch.float64), packed, res=res)
        v = (v * 2.0).cpu().numpy(); f = f.cpu().numpy()
        spec["meshes"].append((v, f)); spec["mesh_vgrad"].append(np.zeros_like(v))
        spec["mesh_id"][s, 2] = 2 + s
        spec["shape_prm"][s, 2, :2] = lat
        spec["inertia"][s, 2] = np.asarray(mass_properties.mesh_inertia(v, f, 1.0).cpu())
        spec["pose"][s, 2, 4:] = (0.0, y0, 0.0)
        spec["fric"][s, 2] = mu
        spec["fext"][s, 2, 4] = -g
    return spec


def geometric_init_weights(seed=0, radius_init=0.5):
    """IGR's geometric initialisation (Atzmon & Lipman 2020; the network constructor of the external IGR repository) for the
    bob_spot_setup shape (IGR_data/train_configs/bob_spot_setup.conf:38-45): every hidden layer N(0, 2/out), the last layer
    mean sqrt(pi)/sqrt(128) and bias -radius, seeded numpy.  Synthetic stand-in for the trained weights (README.md:41-42:
    a download, unavailable offline)."""
    r = np.random.default_rng(seed)
    dims = [5] + [128] * 8 + [1]
    Ws, bs = [], []
    for l in range(9):
        out = dims[l + 1] - 5 if l + 1 == 4 else dims[l + 1]
        if l == 8:
            Ws.append(r.normal(np.sqrt(np.pi) / np.sqrt(dims[l]), 1e-5, (out, dims[l]))); bs.append(np.full(out, -radius_init))
        else:
            Ws.append(r.normal(0.0, np.sqrt(2) / np.sqrt(out), (out, dims[l]))); bs.append(np.zeros(out))
    return Ws, bs


def inertia_spin(B, seed=0):
    """config 5 shape (BASELINE configs[4], experiments/inertia_fitting/optim_shapespace.py:71-92): one body per scene, its
    translation locked by X/Y/Z constraints, a constant torque about a random axis, no contacts (the engine's linear-solve
    branch, engines.py:40-54).  The body's inertia is the fitted quantity; an icosphere stands in for the shape mesh (the
    stepping cost does not depend on it: nothing collides)."""
    r = np.random.default_rng(seed)
    v, f = meshes.icosphere(3)
    dirs = r.standard_normal((B, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    I0 = np.stack([np.diag(0.1 + 0.2 * r.random(3)) for _ in range(B)])[:, None]
    one = lambda a: np.tile(np.asarray(a, np.float64), (B, 1, 1))
    return dict(pose=one([1.0, 0, 0, 0, 0, 0, 0]), vel=one(np.zeros(6)), mass=np.ones((B, 1)), inertia=I0,
                restitution=np.zeros((B, 1)), fric=np.zeros((B, 1)), fext=np.concatenate([0.5 * dirs, np.zeros((B, 3))], 1)[:, None],
                shape_type=np.ones((B, 1), np.int32), shape_prm=one([0.6, 0, 0]), mesh_id=np.zeros((B, 1), np.int32),
                meshes=[(0.6 * v, f)], mesh_vgrad=[v],
                Je=np.tile(np.concatenate([np.zeros((3, 3)), np.eye(3)], 1), (B, 1, 1)), no_contact=np.zeros((1, 1), np.uint8))
