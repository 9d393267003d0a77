"""Diagnostic: per-phase time of narrowphase_kernel workgroups (wall_clock64 stamps, 100 MHz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
KIND = sys.argv[2] if len(sys.argv) > 2 else "stack"      # "sphere": configs[1]'s scenes after 150 steps (most spheres on the floor)
if KIND == "sphere":
    E = BatchEngine(scenes.sphere_drop(B, seed=1000), maxc=64, max_sub=900, backend=TorchBackend("cuda"))
    E.run(150)
else:
    spec = scenes.box_stack(B, nbox=7, seed=1)
    E = BatchEngine(spec, maxc=128, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
np_ = E.nb * (E.nb - 1)
dbg = torch.zeros(B * np_ * 8, dtype=torch.int64, device="cuda")
E.W.dbg_stamps = dbg.data_ptr()
E.step()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(B, np_, 8)
names = ["scan", "fw", "project", "geom", "filter", "final"]
act = d[:, :, 6] > 0
print("active WGs per scene", act.sum() / B, "of", np_)
for s_ in range(B):
    pass
dd = np.diff(d[:, :, :7], axis=2)[act] / 100.0   # us
print("phase us mean:", dict(zip(names, dd.mean(0).round(1))), "total mean", dd.sum(1).mean().round(1), "max", dd.sum(1).max().round(1))
a, b = np.nonzero(act)
tot = dd.sum(1)
for dp in sorted(set(b)):
    m = b == dp
    print("dp", dp, "(a,b)=", (dp // (E.nb - 1), (lambda r, a_: r if r < a_ else r + 1)(dp % (E.nb - 1), dp // (E.nb - 1))), "n", m.sum(), "us", dd[m].mean(0).round(0), "tot", tot[m].mean().round(0))
span = (d[:, :, 6][act].max() - d[:, :, 0][d[:, :, 0] > 0].min()) / 100.0
print("kernel span us", span)
