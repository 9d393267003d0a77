import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine
E = BatchEngine(scenes.sphere_drop(256, seed=1000), maxc=64, max_sub=900)
E.run(3)
t0 = time.perf_counter(); r = E.run(200); torch.cuda.synchronize(); t1 = time.perf_counter()
adj = E._adjoint()
for k in ("a_pose", "a_vel", "a_geom", "g_mass", "g_inertia", "g_rest", "g_fric", "g_fext", "g_prm"): adj[k].zero_()
adj["a_pose"][:, :, 4:] = 2.0 * E.arr["pose"][:, :, 4:]
adj["cur_slot"].copy_(E.arr["nsub"] - 1)
E.adj["lo_slot"].zero_()
torch.cuda.synchronize(); t2 = time.perf_counter()
n = int(E.arr["nsub"].max().item())
E.backward_sweep(n)
t3 = time.perf_counter()
torch.cuda.synchronize(); t4 = time.perf_counter()
print("fwd %.3f s (%d rounds)  adjoint setup %.4f  sweeps %d: launch %.4f s, drain %.4f s" % (t1 - t0, r, t2 - t1, n, t3 - t2, t4 - t3))
