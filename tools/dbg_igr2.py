"""Contact detection of a neural-SDF golden scene at one recorded pose, with the query rounds cut short (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import igr_helpers as H  # noqa: E402
import rollout_helpers as R  # noqa: E402
from diffsdfsim_amd.engine import BatchEngine  # noqa: E402

name, step = sys.argv[1], int(sys.argv[2])
g = R.load_rollout(name)
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for rounds in [int(a) for a in sys.argv[3:]] or [0]:
    spec = H.spec_from_golden(g, 1)
    spec["pose"] = g["traj_p"][step][None].copy()
    spec["igr_rounds"] = rounds
    try:
        E = BatchEngine(spec, **H.engine_kwargs(g, max_sub=0))
    except Exception as e:
        print("rounds", rounds, "raised", str(e)[:80])
        continue
    print("rounds", rounds, "nc", E.get("nc"), "ref", int(g["traj_nc"][step]), "pc_count", E.get("pc_count")[0])
    print(" hdr", E.get("igr_hdr")[:, :7].tolist(), "qn", E.get("igr_qn")[:12].tolist(), "n_pairs", E.get("n_pairs").tolist())
    if rounds == 0:
        from oracle import igr_oracle
        Ws, bs = H.seeded_weights(g)
        n = int(E.get("igr_qn")[2])
        pts = E.get("igr_qpts")[2][:n]; got = E.get("igr_qsdf")[2][:n]; tag = E.get("igr_qtag")[2][:n]; lat = E.get("igr_qlat")[2][:n]
        want, _ = igr_oracle.query(pts, g["latent"], Ws, bs)
        print("value list of round 1: n", n, "lat idx", sorted(set(lat.tolist())), "max |kernel - oracle|", np.abs(got - want).max())
        print(" phi*scale min", (want * 2).min(), "kernel min", (got * 2).min(), "tags", tag[:8], "pts", pts[:3])
