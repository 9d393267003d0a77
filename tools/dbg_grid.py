import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rollout_helpers as R
from diffsdfsim_amd.engine import BatchEngine
g = R.load_rollout("rollout_grid_body")
E = BatchEngine(R.spec_from_golden(g, 1), **R.engine_kwargs(g, max_sub=64, maxc=64, max_cand=4096, max_pc=64))
for _ in range(12): E.step()
np.set_printoptions(precision=9, linewidth=200, suppress=True)
j = 23
n = int(E.get("tp_nc")[j, 0]); nr = int(g["traj_nc"][j - 1])
a = E.get("tp_geom")[j, 0][:, :n].T; b = g["traj_geom"][j - 1][:nr]
ba = E.get("tp_body")[j, 0][:, :n].T
print("mine", n); print(np.c_[ba, a[:, 3:6], a[:, :3]][np.lexsort(a[:, 3:6].T[::-1])])
print("ref", nr); print(np.c_[g["traj_body"][j - 1][:nr], b[:, 3:6], b[:, :3]][np.lexsort(b[:, 3:6].T[::-1])])
