"""Dev aid (GPU box): autograd vs central finite differences of the sys-id loss w.r.t. push, mass and friction (push_world)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsdfsim_amd import experiments as X, igr, scenes
packed = igr.pack_weights(*scenes.geometric_init_weights(0, 0.5))
lat = np.array([[0.05, -0.03], [-0.04, 0.06]])
base = dict(force=np.array([[3.0, 2.5], [4.0, 2.0]]), mass=np.array([1.0, 0.95]), fric=np.array([0.1, 0.2]))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cache = {}
with torch.no_grad():
    wt = X.push_world(lat, packed, torch.tensor(base["force"]) * 1.1, torch.tensor(base["mass"]) * 1.05, torch.tensor(base["fric"]) * 0.8, steps, res=48, mesh_cache=cache)
    pos_t = X.rollout(wt, steps)[0][:, :, 1, 4:].clone()
def loss_of(p, grad=False):
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=grad) for k, v in p.items()}
    w = X.push_world(lat, packed, t["force"], t["mass"], t["fric"], steps, res=48, mesh_cache=cache)
    pos = X.rollout(w, steps)[0][:, :, 1, 4:]
    l = ((pos - pos_t.to(pos)) ** 2).sum(dim=(0, 2))
    if grad:
        l.sum().backward()
        return l.detach().cpu().numpy(), {k: v.grad.numpy().copy() for k, v in t.items()}
    return l.detach().cpu().numpy()
l0, g = loss_of(base, True)
print("loss", l0)
for k in ("force", "mass", "fric"):
    for idx in np.ndindex(base[k].shape):
        h = 1e-6
        pp = {a: b.copy() for a, b in base.items()}; pm = {a: b.copy() for a, b in base.items()}
        pp[k][idx] += h; pm[k][idx] -= h
        fd = (loss_of(pp)[idx[0]] - loss_of(pm)[idx[0]]) / (2 * h)
        print("%-6s %s  autograd % .6e  fd % .6e  rel %.2e" % (k, idx, g[k][idx], fd, abs(g[k][idx] - fd) / max(abs(fd), 1e-12)))
