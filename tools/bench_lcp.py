"""Micro-benchmark of the two LCP kernels (development aid; bench.py is the contract benchmark)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time
import numpy as np, torch
sys.path.insert(0, "tests")
import structured as S
from diffsdfsim_amd.lcp.contact import lcp_contact_forward, lcp_contact_backward

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    P = S.random_problem(seed=31, B=B, nb=8, maxc=128, fd=8, nc_lo=90, nc_hi=120)
    t = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    d = dict(Mblk=t(P["Mblk"]), pvec=t(P["pvec"]), A=t(P["A"]), bvec=t(P["bvec"]), cop=t(P["cop"]),
             cbody=t(P["cbody"], torch.int32), nc=t(P["nc"], torch.int32))
    ws = torch.empty(B * 5 * 10 * 128 * 8, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        out = lcp_contact_forward(d["Mblk"], d["pvec"], d["A"], d["bvec"], d["cop"], d["cbody"], d["nc"], 8, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = lcp_contact_forward(d["Mblk"], d["pvec"], d["A"], d["bvec"], d["cop"], d["cbody"], d["nc"], 8, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    x, lam, slack, nu, it, st = out
    print("contact fwd  B=%d: %.3f ms/launch  iters mean %.1f" % (B, e0.elapsed_time(e1) / 10, it.float().mean().item()))
    dl = torch.randn_like(x)
    e0.record()
    for _ in range(10):
        g = lcp_contact_backward(d["Mblk"], d["A"], d["cop"], d["cbody"], d["nc"], 8, x, lam, slack, nu, dl)
    e1.record(); torch.cuda.synchronize()
    print("contact bwd  B=%d: %.3f ms/launch" % (B, e0.elapsed_time(e1) / 10))

if __name__ == "__main__":
    main()
