"""Dev aid (GPU box): step a rollout golden and print, per accepted sub-step, the deviation from the reference."""
import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rollout_helpers as R
from diffsdfsim_amd.engine import BatchEngine

name, nsteps = sys.argv[1], int(sys.argv[2])
g = R.load_rollout(name)
lsm = None
if any(k.startswith("meshsize_") for k in g):
    import test_primitives_gpu as T
    lsm = T.level_set_mesh(g)
E = BatchEngine(R.spec_from_golden(g, 1, lsm), **R.engine_kwargs(g, max_sub=128, maxc=int(os.environ.get("MAXC", 128)), max_cand=16384, max_pc=int(os.environ.get("MAXC", 128))))
print("init nc", int(E.get("nc")[0]), "ref", len(g["init_body"]))
for _ in range(nsteps):
    E.step()
n = int(E.get("nsub")[0])
tp, tdt, tnc = E.get("tp_pose"), E.get("tp_dt"), E.get("tp_nc")
t = 0.0
for j in range(min(n, len(g["traj_t"]))):
    t += tdt[j, 0]
    ref_prev = g["pose0"] if j == 0 else g["traj_p"][j - 1]
    print("sub %2d  dt %.6f t %.6f ref_t %.6f  start-pose dev %.2e  nc(start) %d ref nc(prev end) %d" % (
        j, tdt[j, 0], t, g["traj_t"][j], np.abs(tp[j, 0] - ref_prev).max(), tnc[j, 0], g["traj_nc"][j - 1] if j else len(g["init_body"])))
print("nsub", n, "ref", len(g["traj_t"]), "final dev", np.abs(E.get("pose")[0] - g["traj_p"][-1]).max())
if len(sys.argv) > 3:
    j = int(sys.argv[3])
    tg, tb = E.get("tp_geom"), E.get("tp_body")
    nc = tnc[j, 0]
    np.set_printoptions(precision=5, suppress=True, linewidth=200)
    ours = tg[j, 0][:, :nc].T
    ref = g["traj_geom"][j - 1][:g["traj_nc"][j - 1]]
    print("ours (n, p1, pen) sorted by p1.x"); print(ours[np.argsort(ours[:, 3])][:, [0, 1, 2, 3, 4, 5, 9]])
    print("bodies", tb[j, 0][:, :nc].T.tolist())
    print("ref"); print(ref[np.argsort(ref[:, 3])][:, [0, 1, 2, 3, 4, 5, 9]])
    print("ref bodies", g["traj_body"][j - 1][:g["traj_nc"][j - 1]].tolist())
    print("pc_stats", E.get("pc_stats")[0].tolist())
if len(sys.argv) > 4:      # per-pair contact counts at sub-step j: ours vs reference
    j = int(sys.argv[3])
    tb = E.get("tp_body")
    from collections import Counter
    ours = Counter(map(tuple, tb[j, 0][:, :tnc[j, 0]].T.tolist()))
    ref = Counter(map(tuple, g["traj_body"][j - 1][:g["traj_nc"][j - 1]].tolist()))
    for k in sorted(set(ours) | set(ref)):
        if ours[k] != ref[k]:
            print("pair", k, "ours", ours[k], "ref", ref[k])
            a = tg[j, 0][:, :tnc[j, 0]].T[[i for i, p in enumerate(tb[j, 0][:, :tnc[j, 0]].T.tolist()) if tuple(p) == k]]
            b = g["traj_geom"][j - 1][:g["traj_nc"][j - 1]][[i for i, p in enumerate(g["traj_body"][j - 1][:g["traj_nc"][j - 1]].tolist()) if tuple(p) == k]]
            print("ours p1,pen\n", a[np.lexsort(a[:, 3:6].T[::-1])][:, [0, 1, 2, 3, 4, 5, 9]]); print("ref p1,pen\n", b[np.lexsort(b[:, 3:6].T[::-1])][:, [0, 1, 2, 3, 4, 5, 9]])
