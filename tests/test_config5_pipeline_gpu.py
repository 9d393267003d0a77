"""GPU: the inertia-fitting pipeline of config 5 end to end on the device (`optim_shapespace.make_world`):
latent code -> IGR level set (fp64 MFMA) -> marching cubes -> mesh inertia -> contact-free constrained rollout under
a torque -> loss on the final angular velocity -> gradient back to the latent code through the stepper's adjoint,
the mesh-inertia adjoint and the MeshSDF formula.  Checked against central differences of the whole pipeline.
(The IGR weights are synthetic geometric-init weights: the reference's checkpoints are not available offline.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def pipeline(latents, P, res=48, nsteps=8):
    from diffsdfsim_amd.mass_properties import mesh_inertia_diff
    from diffsdfsim_amd.meshsdf import igr_mesh
    from diffsdfsim_amd.physics3d import BatchWorld3D
    B = latents.shape[0]
    inert, meshes = [], []
    for s in range(B):
        v, f = igr_mesh(latents[s], P, res=res)
        inert.append(mesh_inertia_diff(v.cpu(), f, 1.0).cpu())
        meshes.append((v.detach().cpu().numpy(), f.cpu().numpy().astype(np.int64)))
    inertia = torch.stack(inert)[:, None]                                   # [B, 1, 3, 3], differentiable w.r.t. the latents
    one = lambda a: np.tile(np.asarray(a, np.float64), (B, 1, 1))
    tq = np.array([0.6, -0.3, 0.74]); tq = 0.5 * tq / np.linalg.norm(tq)
    spec = dict(pose=one([1.0, 0, 0, 0, 0, 0, 0]), vel=one(np.zeros(6)), mass=np.ones((B, 1)), inertia=inertia.detach().numpy(),
                restitution=np.zeros((B, 1)), fric=np.zeros((B, 1)), fext=one(np.concatenate([tq, np.zeros(3)])),
                shape_type=np.ones((B, 1), np.int32), shape_prm=one([0.6, 0, 0]), mesh_id=np.arange(B, dtype=np.int32)[:, None],
                meshes=meshes, mesh_vgrad=[np.zeros_like(m[0]) for m in meshes],
                Je=np.tile(np.concatenate([np.zeros((3, 3)), np.eye(3)], 1), (B, 1, 1)), no_contact=np.zeros((1, 1), np.uint8))
    w = BatchWorld3D(spec, params=dict(inertia=inertia), max_substeps=4 * nsteps)
    for _ in range(nsteps):
        w.step()
    return (w.vel[:, 0, :3] ** 2).sum(dim=1)            # per-scene loss: |omega_T|^2


def test_gradient_of_the_rollout_reaches_the_latent_code():
    from diffsdfsim_amd.igr import pack_weights
    from oracle import igr_oracle as IO
    P = pack_weights(*IO.geometric_init(seed=4, radius_init=0.6))
    lat = torch.tensor([[0.05, -0.08], [-0.03, 0.06], [0.0, 0.1]], dtype=torch.float64, requires_grad=True)
    loss = pipeline(lat, P)
    loss.sum().backward()
    g = lat.grad.clone()
    assert torch.isfinite(g).all() and (g.abs() > 0).all()
    h = 2e-3
    for (s, k) in [(0, 0), (0, 1), (2, 1)]:
        e = torch.zeros_like(lat); e[s, k] = h
        fd = (pipeline((lat.detach() + e), P)[s] - pipeline((lat.detach() - e), P)[s]) / (2 * h)
        assert abs(g[s, k] - fd) < 0.08 * abs(fd) + 1e-5, (s, k, float(g[s, k]), float(fd))
