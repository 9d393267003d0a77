"""Dev aid (GPU box): the network kernel on a long VALUE-only list (the candidate scan of the neural narrow phase)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsdfsim_amd import _lib, scenes
from diffsdfsim_amd.igr import pack_weights
P = pack_weights(*scenes.geometric_init_weights(0, 0.5))
L = _lib.lib()
class Net(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("W0", "b0", "Wp", "bh", "W8", "b8")]
net = Net(*[P[k].data_ptr() for k in ("W0", "b0", "Wp", "bh", "W8", "b8")])
for n in (240_000, 60_000, 1_000_000):
    pts = torch.rand(n, 3, dtype=torch.float64, device="cuda") * 1.6 - 0.8
    lat = torch.zeros(1, 3, dtype=torch.float64, device="cuda")
    sdf = torch.empty(n, dtype=torch.float64, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go():
        rc = L.dss_igr_query_list(ctypes.byref(net), ctypes.c_void_p(pts.data_ptr()), None, ctypes.c_void_p(lat.data_ptr()), 3, None, n, 2,
                                  ctypes.c_void_p(sdf.data_ptr()), None, st)
        assert rc == 0, rc
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s  VALUE n=%d  %.3f ms  %.1f TFLOP/s   checksum %.12e" % (os.environ.get("DSS_LIB_PATH", "product")[-16:], n, ms, 2 * 115456 * n / ms / 1e9, float(sdf.sum())), flush=True)
