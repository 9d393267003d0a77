import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rollout_helpers as R
from diffsdfsim_amd.engine import BatchEngine
g = R.load_rollout(sys.argv[1])
E = BatchEngine(R.spec_from_golden(g, 1), **R.engine_kwargs(g, max_sub=64, maxc=128))
n = int(E.get("nc")[0])
f = E.get("c_face")[0][:n]; body = E.get("c_body")[0][:, :n].T; geom = E.get("c_geom")[0][:, :n].T
np.set_printoptions(precision=6, linewidth=220, suppress=True)
print("mine flags", ((f >> 30) & 1).tolist())
print("body", body.tolist())
print("ref  stable", g["init_stable"][:n].tolist())
print("ref body", g["init_body"][:n].tolist())
print("ref lap", g["init_lap"][:n])
print("mine n", geom[:, :3]); print("ref n", g["init_geom"][:n, :3])
