"""Diagnostic: phase shares of lcp_contact_forward_kernel (needs the -DDSS_DIAG build: tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, glob, os, subprocess, sys
import numpy as np, torch
from diffsdfsim_amd import _lib, scenes
src = sorted(glob.glob(os.path.join(_lib.CSRC, "*.hip")))
diag = os.path.join(_lib.CSRC, "libdiffsdfsim_hip_diag.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDSS_DIAG", "-o", diag] + os.environ.get("DSS_DIAG_FLAGS", "").split() + src)
_lib.LIB_PATH = diag
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
L0 = ctypes.CDLL(diag)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
KIND = sys.argv[2] if len(sys.argv) > 2 else "stack"      # "sphere": configs[1]'s scenes, stepped until most spheres rest on the floor
if KIND == "sphere":
    E = BatchEngine(scenes.sphere_drop(B, seed=1000), maxc=64, max_sub=900, backend=TorchBackend("cuda"))
    E.run(150)
    print("scenes in contact:", int((E.get("nc") > 0).sum()), "of", B)
else:
    E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1), maxc=128, max_cand=1024, max_pc=48, strict_no_pen=False, backend=TorchBackend("cuda"))
st = torch.zeros(B * 16, dtype=torch.int64, device="cuda")
E.be.lib.dss_diag_set_lcp_stamps(ctypes.c_void_p(st.data_ptr()), E.be.stream())
try:
    E.step()
except RuntimeError as e:
    print("(step raised:", str(e)[:60], ")")
torch.cuda.synchronize()
d = st.cpu().numpy().reshape(B, 16) / 100.0
d = d[E.get("nc") > 0] if KIND == "sphere" else d
names = ["pre", "P1 resid+Cmat pass", "gather2", "resid/best", "C to LDS", "assemble_K", "-", "factor+solve aff", "P4 pass", "sigma+P5 pass", "gather1+solve cor", "P6 pass", "update"]
tot = d.sum(1).mean()
print("iters", E.get("lcp_iters").mean(), "nc mean", E.get("nc").mean(), "total us/scene", round(tot, 1))
for i, n in enumerate(names):
    print("%-22s %8.1f us  %5.1f%%" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot))
