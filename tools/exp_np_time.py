"""Dev aid (GPU box): time of the detection launch group of config 3 for an experimental library (DSS_LIB_PATH)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spec = scenes.box_stack(B, nbox=7, seed=1000)
E = BatchEngine(spec, maxc=128, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
L = E.be.lib
E._set_active(1)
def detect():
    E._check(L.dss_find_contacts(ctypes.byref(E.W), E.be.stream()), "dss_find_contacts")
for _ in range(3):
    detect()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    detect()
e1.record(); torch.cuda.synchronize()
print("%s  detection %.3f ms  (contacts per scene %.1f)" % (os.environ.get("DSS_LIB_PATH", "product"), e0.elapsed_time(e1) / 10, E.get("nc").mean()), flush=True)
