"""Dev aid: raw dependent-issue latencies of gfx950 for one wavefront per SIMD (needs the -DDSS_DIAG build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, glob, os, subprocess, sys
import numpy as np, torch
from diffsdfsim_amd import _lib
src = sorted(glob.glob(os.path.join(_lib.CSRC, "*.hip")))
diag = os.path.join(_lib.CSRC, "libdiffsdfsim_hip_diag.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDSS_DIAG", "-o", diag] + src)
L = ctypes.CDLL(diag)
names = ["dep fp64 fma", "4 indep fp64 fma chains (per 4)", "2 readlane + fma", "dep LDS read", "dep global read (L2)", "dep fp64 div", "dep fp64 sqrt+add", "dep shuffle+add"]
n = 20000
chase = torch.tensor((np.arange(1 << 16) * 37 + 11) & 0xFFFF, dtype=torch.int32, device="cuda")
for grid in (1, 1024, 4096):
    sink = torch.zeros(grid * 64, dtype=torch.float64, device="cuda")
    out = torch.zeros(grid * 2, dtype=torch.int64, device="cuda")
    print("grid", grid)
    for mode, nm in enumerate(names):
        for _ in range(2):
            L.dss_diag_latency(mode, n, grid, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, ctypes.c_void_p(chase.data_ptr()), None)
            torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(grid, 2).astype(np.float64)
        cyc, wall = o[:, 0].mean(), o[:, 1].mean()
        print("  %-34s %7.1f clk/iter  %7.2f ns/iter   clock64 rate %.0f MHz" % (nm, cyc / n, wall * 10.0 / n, cyc / (wall * 10.0) * 1000))
