"""Dev aid (GPU box): run every tests/golden/tmp_*.npz rollout and report where it leaves the reference."""
import glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, rollout_helpers as R
from diffsdfsim_amd.engine import BatchEngine
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "tmp_*.npz"))):
    name = os.path.basename(f)[:-4]
    g = R.load_rollout(name)
    nsteps = int(round(float(g["t_final"]) / float(g["dt"])))
    lsm = None
    if any(k.startswith("meshsize_") for k in g):
        import test_primitives_gpu as T
        lsm = T.level_set_mesh(g)
    kw = dict(max_sub=256, maxc=256, max_pc=256, max_cand=32768) if lsm else dict(max_sub=256, maxc=160, max_pc=32)
    E = BatchEngine(R.spec_from_golden(g, 1, lsm), **R.engine_kwargs(g, **kw))
    for _ in range(nsteps):
        E.step()
    n = int(E.get("nsub")[0]); T = len(g["traj_t"])
    tp, tnc = E.get("tp_pose"), E.get("tp_nc")
    first = None
    for j in range(1, min(n, T)):
        dev = np.abs(tp[j, 0] - g["traj_p"][j - 1]).max()
        if dev > 1e-9 or int(tnc[j, 0]) != int(g["traj_nc"][j - 1]):
            first = (j, dev, int(tnc[j, 0]), int(g["traj_nc"][j - 1])); break
    fin = np.abs(E.get("pose")[0] - g["traj_p"][-1]).max() if n == T else float("nan")
    gerr = float("nan")
    if "grad_0" in g and n == T:
        E2 = BatchEngine(R.spec_from_golden(g, 1), **R.engine_kwargs(g, max_sub=256, maxc=160, max_pc=32))
        R.rollout_and_sweep(E2, nsteps)
        got = R.param_grads(E2, g, 0)
        errs = []
        for key in ("grad_%d", "gradB_%d"):
            errs.append(max(np.abs(gi - g[key % i]).max() / max(np.abs(g[key % i]).max(), 1e-300) for i, gi in enumerate(got)))
        gerr = min(errs)
    print("%-18s nsub %3d ref %3d  final pose dev %.1e  grad rel err %.1e  first deviation %s" % (name, n, T, fin, gerr, first))
