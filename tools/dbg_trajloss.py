import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsdfsim_amd import experiments as X
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "trajectory_loss_bounce.npz"))
rt = float(g["run_time"])
with torch.no_grad():
    tgt = X.run_world_fixed_dt(X.bounce_world(torch.tensor([float(g["r_target"])], dtype=torch.float64), run_time=rt), rt)
rad = torch.tensor([float(g["r_start"])], dtype=torch.float64, requires_grad=True)
tr = X.run_world_fixed_dt(X.bounce_world(rad, run_time=rt), rt)
print("gpu loss", float(X.trajectory_loss(tr, tgt)[0]), "golden", float(g["plain_loss"]))
cpu = lambda d: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in d.items()}
print("same tensors on the cpu", float(X.trajectory_loss(cpu(tr), cpu(tgt))[0]))
v = tr["valid"][:, 0].cpu().numpy(); tv = tgt["valid"][:, 0].cpu().numpy()
print("Kw", len(v), int(v.sum()), "Kt", len(tv), int(tv.sum()))
print("world t (all)", tr["t"][:, 0].detach().cpu().numpy())
print("valid", v.astype(int))
print("target t (all)", tgt["t"][:, 0].cpu().numpy()); print("tvalid", tv.astype(int))
wp = tr["pose"][:, 0].detach().cpu().numpy(); tp = tgt["pose"][:, 0].cpu().numpy()
print("world pose vs golden", np.abs(wp - g["plain_p"]).max(), "target pose vs golden", np.abs(tp - g["target_p"]).max())
print("per entry world err", np.abs(wp - g["plain_p"]).reshape(len(wp), -1).max(1))
print("per entry target err", np.abs(tp - g["target_p"]).reshape(len(tp), -1).max(1))
