"""Dev aid (GPU box): sub-phase times inside stage 1 of the narrow phase (experiment build with DSS_NP_SCAN_STAMPS)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B = 1024
E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1000), maxc=128, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
np_ = E.nb * (E.nb - 1)
dbg = torch.zeros(B * np_ * 8, dtype=torch.int64, device="cuda")
E.W.dbg_stamps = dbg.data_ptr()
E._set_active(1)
E._check(E.be.lib.dss_find_contacts(ctypes.byref(E.W), E.be.stream()), "find")
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(B, np_, 8)
act = d[:, :, 4] > 0
dd = np.diff(d[:, :, :5], axis=2)[act] / 100.0
print("items", act.sum(), "us mean: setup %.1f  boxes %.1f  centroids %.1f  full tests %.1f   total %.1f" % (*dd.mean(0), dd.sum(1).mean()))
st = E.get("pc_stats").reshape(B, np_, 2)[act]
print("runs passed mean %.1f  candidates mean %.1f" % (st[:, 0].mean(), st[:, 1].mean()))
