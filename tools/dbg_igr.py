"""Step a neural-SDF golden scene and print where it leaves the reference's trajectory (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import igr_helpers as H  # noqa: E402
import rollout_helpers as R  # noqa: E402
from diffsdfsim_amd.engine import BatchEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rollout_igr_demo"
g = R.load_rollout(name)
E = BatchEngine(H.spec_from_golden(g, 1), **H.engine_kwargs(g, max_sub=128))
print("init nc", E.get("nc"), "ref", len(g["init_body"]))
j = 0
while float(E.get("t")[0]) < float(g["run_time"]) and j < 60:
    att = E.step_once()
    t = float(E.get("t")[0])
    nc = int(E.get("nc")[0])
    ref_t = g["traj_t"][j + 1] if j + 1 < len(g["traj_t"]) else float(g["t_final"])
    dp = np.abs(E.get("pose")[0] - g["traj_p"][j]).max() if j < len(g["traj_p"]) else -1
    dv = np.abs(E.get("vel")[0] - g["traj_v"][j]).max() if j < len(g["traj_v"]) else -1
    print("step %2d attempts %d t=%.7f (ref %.7f) nc=%d (ref %d) |dpose|=%.2e |dvel|=%.2e ov=%d" %
          (j, att, t, ref_t, nc, int(g["traj_nc"][j]) if j < len(g["traj_nc"]) else -1, dp, dv, int(E.get("overflow")[0])))
    j += 1
    if j - 1 < len(g["traj_nc"]) and nc != int(g["traj_nc"][j - 1]) and not globals().get("_shown"):
        _shown = True
        np.set_printoptions(precision=6, suppress=True, linewidth=200)
        print("OURS body\n", E.get("c_body")[0][:, :nc].T, "\ngeom\n", E.get("c_geom")[0][:, :nc].T)
        n = int(g["traj_nc"][j - 1])
        print("REF body\n", g["traj_body"][j - 1][:n], "\ngeom\n", g["traj_geom"][j - 1][:n], "\nstable", g["traj_stable"][j - 1][:n])
        print("pc_count", E.get("pc_count")[0], "igr hdr", E.get("igr_hdr")[:, :7], "n_pairs", E.get("n_pairs"))
