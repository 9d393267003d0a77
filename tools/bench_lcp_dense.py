"""Dev aid (GPU box): the dense LCP kernel (boundary B1) in the regime where an LCP could be HBM-bound -- many small systems.
For each size: algorithmic bytes (operands in + results out), launch time, GB/s against 8 TB/s, and the arithmetic intensity
that decides the bound (flops ~ iterations x (2/3 n^3 + 2 n^2 m), n = nz + nineq + neq).  -> stdout (JSON); --quick: the two smallest sizes, one launch each (for counter passes)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as H
from diffsdfsim_amd.lcp.lcp import lcp_dense_forward

rows = []
QUICK = "--quick" in sys.argv
SIZES = [(65536, 3, 4, 0), (65536, 6, 8, 3)] if QUICK else [(65536, 3, 4, 0), (65536, 6, 8, 3), (65536, 6, 16, 3), (65536, 12, 24, 3), (16384, 12, 48, 6), (1024, 48, 560, 6)]
for (B, nz, nineq, neq) in SIZES:
    nb = min(B, 2048)
    Q, p, G, h, A, b, F = H.random_lcp(7, nb, nz, nineq, neq)
    rep = B // nb
    t = lambda a: torch.tensor(np.ascontiguousarray(np.tile(a, (rep,) + (1,) * (a.ndim - 1))), dtype=torch.float64, device="cuda")
    ops = [t(x) for x in (Q, p, G, h, A, b, F)]
    for _ in range(1 if QUICK else 2):
        out = lcp_dense_forward(*ops, 1e-12, 3, 20, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_rep = 3 if QUICK else 5
    e0.record()
    for _ in range(n_rep):
        out = lcp_dense_forward(*ops, 1e-12, 3, 20, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n_rep
    iters = float(out[4].float().mean())
    algo = 8.0 * B * (nz * nz + nz + nineq * nz + nineq + neq * nz + neq + nineq * nineq + nz + 2 * nineq + neq)
    n = nz + nineq + neq
    flops = B * iters * (2.0 / 3.0 * nineq ** 3 + 2.0 * nineq * nineq * nz + 2.0 / 3.0 * (nz + neq) ** 3 + 4.0 * n * n)
    rows.append(dict(kernel="lcp_dense_group_forward_kernel (8 lanes per system)" if max(nz, nineq, neq) <= 8 else "lcp_dense_forward_kernel (wavefront per system)",
                     B=B, nz=nz, nineq=nineq, neq=neq, ms=ms, iters_mean=iters, algorithmic_bytes=algo,
                     GBps=algo / ms / 1e6, frac_of_8TBps=algo / ms / 1e6 / 8000.0, est_flops=flops,
                     est_TFLOPs=flops / ms / 1e9, flop_per_byte=flops / algo, machine_balance_flop_per_byte=78.6e12 / 8e12))
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps(dict(rows=rows), indent=1))
