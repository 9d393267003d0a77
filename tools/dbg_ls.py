import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, rollout_helpers as R, test_primitives_gpu as T
from diffsdfsim_amd.engine import BatchEngine
g = R.load_rollout("rollout_levelset_box")
E = BatchEngine(R.spec_from_golden(g, 1, T.level_set_mesh(g)), **R.engine_kwargs(g, max_sub=16, maxc=64, max_cand=32768, max_pc=32))
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for step in range(2):
    nc = int(E.get("nc")[0]); body = E.get("c_body")[0][:, :nc].T; geom = E.get("c_geom")[0][:, :nc].T
    print("after", step, "steps: nc", nc, "pc_stats", E.get("pc_stats")[0].tolist())
    print(np.concatenate([body, geom[:, [3, 4, 5, 9]]], 1))
    E.step()
