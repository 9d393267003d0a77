"""Dev aid: BASELINE configs[4] shape on one GPU: N contact-free single-body scenes (X/Y/Z constraints, torque),
200 steps forward + backward through BatchWorld3D; plus the per-body world-construction cost at 128^3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsdfsim_amd import meshes
from diffsdfsim_amd.physics3d import BatchWorld3D

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 200
r = np.random.default_rng(0)
v, f = meshes.icosphere(3)
dirs = r.standard_normal((B, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
I0 = np.stack([np.diag(0.1 + 0.2 * r.random(3)) for _ in range(B)])[:, None]
one = lambda a: np.tile(np.asarray(a, np.float64), (B, 1, 1))
spec = dict(pose=one([1.0, 0, 0, 0, 0, 0, 0]), vel=one(np.zeros(6)), mass=np.ones((B, 1)), inertia=I0,
            restitution=np.zeros((B, 1)), fric=np.zeros((B, 1)), fext=np.concatenate([0.5 * dirs, np.zeros((B, 3))], 1)[:, None],
            shape_type=np.ones((B, 1), np.int32), shape_prm=one([0.6, 0, 0]), mesh_id=np.zeros((B, 1), np.int32), meshes=[(0.6 * v, f)],
            mesh_vgrad=[v], Je=np.tile(np.concatenate([np.zeros((3, 3)), np.eye(3)], 1), (B, 1, 1)), no_contact=np.zeros((1, 1), np.uint8))
inertia = torch.tensor(I0, requires_grad=True)
w = BatchWorld3D(spec, params=dict(inertia=inertia), max_substeps=T + 16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(T):
    w.step()
(w.vel[:, 0, :3] ** 2).sum().backward()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("config 5 stepping: B=%d T=%d  %.0f steps/s (fwd+bwd), %.2f M scene-steps/s, grad finite %s" % (
    B, T, T / dt, B * T / dt / 1e6, bool(torch.isfinite(inertia.grad).all())))
