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
