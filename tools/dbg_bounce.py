import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffsdfsim_amd import experiments as X
np.set_printoptions(precision=4, linewidth=200, suppress=True)
tgt = torch.tensor([0.8, 1.1], dtype=torch.float64)
T = X.run_world_fixed_dt(X.bounce_world(tgt, run_time=1.0), 1.0)
rad = torch.tensor([0.95, 1.0], dtype=torch.float64, requires_grad=True)
W = X.run_world_fixed_dt(X.bounce_world(rad, run_time=1.0), 1.0, detach_2nd_bounce=True)
print("target t", T["t"][:, 0].cpu().numpy()); print("target pos", T["pose"][:, 0, -1, 4:].cpu().numpy())
print("world t", W["t"][:, 0].cpu().numpy(), "valid", W["valid"][:, 0].cpu().numpy().astype(int)); print("world pos", W["pose"][:, 0, -1, 4:].detach().cpu().numpy())
L = X.trajectory_loss(W, T); print("loss", L)
L.sum().backward(); print("grad", rad.grad)
