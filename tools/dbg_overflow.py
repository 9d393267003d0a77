import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B = 256
E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1000), maxc=192, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
ov = E.get("overflow"); pc = E.get("pc_count")
print("overflow scenes", np.nonzero(ov)[0][:10], "bits", np.unique(ov))
s = int(np.nonzero(ov)[0][0]) if ov.any() else 0
print("scene", s, "pc_count", pc[s][pc[s] > 0], "pairs", np.nonzero(pc[s] > 0)[0])
dp = int(np.argmax(pc[s]))
g = E.get("pc_geom")[s, dp][:, :pc[s, dp]]
np.set_printoptions(precision=4, suppress=True, linewidth=200)
print("dp", dp, "n", pc[s, dp]); print("normals\n", g[0:3].T[:12]); print("p1\n", g[3:6].T[:60])
print("pose", E.get("pose")[s])
