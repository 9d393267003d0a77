"""Where does the build's normal choice differ from the reference's recorded one (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rollout_helpers as R
from diffsdfsim_amd.engine import BatchEngine
from diffsdfsim_amd import mass_properties
name, nsteps = sys.argv[1], int(sys.argv[2])
g = R.load_rollout(name)
E = BatchEngine(R.spec_from_golden(g, 1), **R.engine_kwargs(g, max_sub=64, maxc=128))
R.rollout_and_sweep(E, nsteps)
tp, tnc, tb, tg = E.get("tp_pose"), E.get("tp_nc"), E.get("tp_body"), E.get("tp_geom")
np.set_printoptions(precision=12, linewidth=200)
k = len(g["traj_t"]) - 1
for j in range(1, k + 1):
    n = int(tnc[j, 0])
    if n != int(g["traj_nc"][j - 1]): continue
    pose, body, geom = tp[j, 0], tb[j, 0], tg[j, 0]
    mine = R.contact_branches(E.get("tp_face")[j, 0], n)
    gb, gg = g["traj_body"][j - 1][:n], g["traj_geom"][j - 1][:n]
    for c in range(n):
        cand = [r for r in range(n) if tuple(gb[r]) == (int(body[0, c]), int(body[1, c])) and np.abs(gg[r, 3:6] - geom[3:6, c]).max() < 1e-6]
        if len(cand) != 1: continue
        r = cand[0]
        ref, lap = int(g["traj_stable"][j - 1][r]), g["traj_lap"][j - 1][r]
        if ref >= 0 and mine[c] != ref and abs(lap[1] - lap[0]) > 1e-9:
            b1, b2 = int(body[0, c]), int(body[1, c])
            R1, R2 = R._quat_to_mat(pose[b1, :4]), R._quat_to_mat(pose[b2, :4])
            cp2 = R2.T @ (geom[3:6, c] + pose[b1, 4:] - pose[b2, 4:]); cp1 = R1.T @ geom[3:6, c]
            q = lambda b, p: mass_properties.sdf_query(int(g["shape_type"][b]), np.concatenate([g["shape_prm"][b], [0.0]]), p[None], return_grads=True)
            d2, n2 = q(b2, cp2); d1, n1 = q(b1, cp1)
            print("step", j, "contact", c, "bodies", b1, b2, "ref stable", ref, "mine", mine[c], "lap", lap)
            print("  n kernel ", geom[0:3, c], " n ref ", gg[r, 0:3])
            print("  R2 n2    ", R2 @ n2.cpu().numpy()[0], " -R1 n1 ", -(R1 @ n1.cpu().numpy()[0]))
            print("  cp1 (b1 frame)", cp1, "dims1", g["shape_prm"][b1], " cp2", cp2, "dims2", g["shape_prm"][b2], "d1", float(d1), "d2", float(d2))
