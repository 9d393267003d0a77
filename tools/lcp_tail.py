"""Dev aid: distribution of IPM iterations / contacts per scene in the bench workload (how much of the LCP launch is tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B = 1024
E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1, push=0.0), maxc=128, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
for s in range(6):
    E.step()
    it = E.get("lcp_iters"); nc = E.get("nc")
    work = it * (1 + nc / 64.0)
    print("step", s, "iters hist", np.bincount(it)[5:], "mean", it.mean(), "max", it.max(), "nc mean/max", nc.mean(), nc.max(),
          "mean/max iters ratio %.2f" % (it.mean() / it.max()))
