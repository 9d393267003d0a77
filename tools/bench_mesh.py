"""Dev aid: world-construction operators at the reference's 128^3 resolution (grid SDF, marching cubes, mesh inertia)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsdfsim_amd.mass_properties import sdf_query, mesh_inertia
from diffsdfsim_amd.meshsdf import marching_cubes, _grid

def timed(fn, reps=10):
    for _ in range(2): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out

res = 128
P = _grid(res, torch.device("cuda"))
prm = np.array([0.9, 1.1, 1.3]) / (1.3 * 0.75)
ms, sdf = timed(lambda: sdf_query(0, prm, P, return_grads=False))
print("box SDF on %d^3 samples: %.3f ms  (%.1f GB/s of 32 B/sample)" % (res, ms, P.shape[0] * 32 / ms / 1e6))
ms, (v, f) = timed(lambda: marching_cubes(sdf.reshape(res, res, res)))
print("marching cubes incl. the size read-back: %.3f ms -> %d verts, %d faces  (%.1f GB/s of 8 B/sample x 2 passes)" % (ms, len(v), len(f), 2 * P.shape[0] * 8 / ms / 1e6))
vn, fn = v.cpu().numpy() / (res - 1) * 2 - 1, f.cpu().numpy()
ms, J = timed(lambda: mesh_inertia(vn, fn, 1.0))
print("mesh inertia (incl. host->device of the mesh): %.3f ms" % ms)
