"""Dev aid: time dss_igr_query (fp64 MFMA IGR MLP) on a 128^3 grid; report TFLOP/s against the fp64 matrix peak."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsdfsim_amd.igr import igr_query, pack_weights
from oracle import igr_oracle as IO

P = pack_weights(*IO.geometric_init(seed=4))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128 ** 3
pts = torch.rand(n, 3, dtype=torch.float64, device="cuda") * 2 - 1
lat = torch.zeros(2, dtype=torch.float64, device="cuda")
from diffsdfsim_amd import _lib
ref = None
for variant in (0,):
    if hasattr(_lib.lib(), "dss_diag_set_igr_variant"): _lib.lib().dss_diag_set_igr_variant(variant)
    for _ in range(3): out = igr_query(pts, lat, P)
    torch.cuda.synchronize()
    if ref is None: ref = out
    assert torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): igr_query(pts, lat, P)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    mac = 4 * (5 * 128 + 7 * 128 * 128 + 128)     # value + 3 tangents
    print(f"variant {variant}: n={n} {ms:.3f} ms  {n / ms / 1e3:.2f} Mpts/s  {2 * mac * n / ms / 1e9:.2f} TFLOP/s fp64 (algorithmic)")
