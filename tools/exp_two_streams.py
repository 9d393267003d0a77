"""Dev aid (GPU box): config-3 stepping as ONE batch of 1024 scenes against TWO half batches on two HIP streams (two host threads):
does the contact LCP of one half (one wavefront per SIMD, 40 % VALU busy) overlap the narrow phase of the other?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NSPLIT = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def make(B, seed):
    return BatchEngine(scenes.box_stack(B, seed=seed), maxc=128, max_cand=1024, max_pc=48, max_sub=int(1.5 * (STEPS + 3)) + 16)


def run(engines, streams):
    def work(E, s):
        with torch.cuda.stream(s):
            for _ in range(STEPS):
                E.step()
            s.synchronize()
    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=work, args=(E, s)) for E, s in zip(engines, streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.time() - t0


one = [make(1024, 1000)]
for E in one:
    for _ in range(3): E.step()
t1 = run(one, [torch.cuda.current_stream()])
print("one batch of 1024: %.3f ms per step" % (1e3 * t1 / STEPS), flush=True)
del one
streams = [torch.cuda.Stream() for _ in range(NSPLIT)]
parts = []
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        E = make(1024 // NSPLIT, 1000 + i)
        for _ in range(3): E.step()
        parts.append(E)
torch.cuda.synchronize()
t2 = run(parts, streams)
print("%d batches of %d on %d streams: %.3f ms per step" % (NSPLIT, 1024 // NSPLIT, NSPLIT, 1e3 * t2 / STEPS), flush=True)
