import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B = 1024
E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1000), maxc=128, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
for _ in range(5): E.step()
st = E.get("pc_stats").reshape(B, -1, 2)
act = st[:, :, 0] > 0
nm = st[:, :, 1][act]
print("items", act.sum(), "movers mean %.2f  hist" % nm.mean(), np.bincount(np.minimum(nm, 20))[:21])
