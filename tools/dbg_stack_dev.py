"""Dev aid (GPU box): per-step contact-set identity and LCP status on a stack golden."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, rollout_helpers as R
from diffsdfsim_amd.engine import BatchEngine
name, nsteps = sys.argv[1], int(sys.argv[2])
g = R.load_rollout(name)
E = BatchEngine(R.spec_from_golden(g, 1), **R.engine_kwargs(g, max_sub=16, maxc=160, max_pc=32))
for k in range(nsteps):
    E.step()
    try:
        R.check_contacts(E, 0, g["traj_body"][k], g["traj_geom"][k], int(g["traj_nc"][k]), tol=1e-7)
        same = "contacts identical (1e-7)"
    except AssertionError as e:
        same = "contacts differ: %s" % str(e)[:80]
    print("step", k + 1, same, "| pose dev %.2e vel dev %.2e" % (np.abs(E.get("pose")[0] - g["traj_p"][k]).max(), np.abs(E.get("vel")[0] - g["traj_v"][k]).max()),
          "| lcp iters", int(E.get("lcp_iters")[0]), "status", int(E.get("lcp_status")[0]))
if len(sys.argv) > 3:
    pair = (int(sys.argv[3]), int(sys.argv[4])); k = int(sys.argv[5]) - 1
    E2 = BatchEngine(R.spec_from_golden(g, 1), **R.engine_kwargs(g, max_sub=16, maxc=160, max_pc=32))
    for _ in range(k + 1):
        E2.step()
    nc = int(E2.get("nc")[0]); body = E2.get("c_body")[0][:, :nc].T; geom = E2.get("c_geom")[0][:, :nc].T
    a = geom[(body == pair).all(1)]; n_ref = int(g["traj_nc"][k])
    b = g["traj_geom"][k][:n_ref][(g["traj_body"][k][:n_ref] == pair).all(1)]
    np.set_printoptions(precision=6, suppress=True, linewidth=200)
    print("ours\n", a[np.lexsort(np.round(a[:, 3:6], 6).T[::-1])][:, [0, 1, 2, 3, 4, 5, 9]]); print("ref\n", b[np.lexsort(np.round(b[:, 3:6], 6).T[::-1])][:, [0, 1, 2, 3, 4, 5, 9]])
    dp = pair[0] * (E2.nb - 1) + (pair[1] - 1 if pair[1] > pair[0] else pair[1])
    print("pc_stats", E2.get("pc_stats")[0][dp])
