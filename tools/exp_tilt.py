"""Dev aid (GPU box): relative tilt of neighbouring boxes in the config-3 stack after N steps (is the contact normal of body 1
the same as that of body 2 to rounding?)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from diffsdfsim_amd import scenes
from diffsdfsim_amd.engine import BatchEngine, TorchBackend
B, K = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 100
E = BatchEngine(scenes.box_stack(B, nbox=7, seed=1000), maxc=128, max_cand=1024, max_pc=48, max_sub=2 * K + 16, strict_no_pen=False,
                backend=TorchBackend(torch.device("cuda:0")))
for k in range(K):
    E.step()
    if k + 1 in (1, 5, 20, 50, K):
        q = E.get("pose")[:, :, :4]                       # [B, nb, 4] wxyz
        w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        up = np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], -1)      # R e_y
        c = (up[:, :-1] * up[:, 1:]).sum(-1)
        ang = np.sqrt(np.maximum(0.0, 2 * (1 - c)))
        print("step %3d: tilt between neighbours: median %.2e, 90%% %.2e, max %.2e; share below 1.4e-6 rad: %.3f" %
              (k + 1, np.median(ang), np.quantile(ang, 0.9), ang.max(), (ang < 1.4e-6).mean()))
