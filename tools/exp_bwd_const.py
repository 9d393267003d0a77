"""Dev aid (GPU box): where the constant part of the reverse sweep's time goes (bench.py's sequence, 20 steps)."""
import sys
import time
import torch
sys.path.insert(0, ".")
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B, K = 1024, 20
E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1000), maxc=128, max_cand=1024, max_pc=48, max_sub=int(1.5 * (K + 2)) + 16, strict_no_pen=False,
                backend=TorchBackend(torch.device("cuda:0")))


def tick(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (label, 1e3 * (time.perf_counter() - t)))
    return r


def loss_adjoint():
    adj = E._adjoint()
    for k in ("a_pose", "a_vel", "a_geom", "g_mass", "g_inertia", "g_rest", "g_fric", "g_fext", "g_prm"):
        adj[k].zero_()
    adj["a_pose"][:, :, 4:] = 2.0 * E.arr["pose"][:, :, 4:]
    adj["cur_slot"].copy_(E.arr["nsub"] - 1)


for rep in range(2):
    att = 0
    for _ in range(2 if rep == 0 else K):
        att += E.step()
    lo = E.arr["nsub"].clone() - att
    print("rep", rep, "attempts", att)
    tick("loss_adjoint", loss_adjoint)
    tick("lo_slot", lambda: E.adj["lo_slot"].copy_(lo))
    tick("first sweep", lambda: E.backward_sweep(1))
    tick("second sweep", lambda: E.backward_sweep(1))
    tick("remaining sweeps (%d)" % (att - 2), lambda: E.backward_sweep(att - 2))
    tick("final cat", lambda: torch.cat([E.arr["pose"].reshape(B, -1), E.adj["g_prm"].reshape(B, -1)], dim=1))
