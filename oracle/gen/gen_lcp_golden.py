"""Generate G1 golden vectors for the LCP boundary from the reference's own CPU path.

Run in the build container only:  python -m oracle.gen.gen_lcp_golden
Writes tests/golden/lcp_*.npz: inputs (Q,p,G,h,A,b,F), outputs (zhat, lam, slack, nu) and
the seven input gradients for the upstream gradient ``dl_dz`` stored alongside
(reference: `lcp_physics/lcp/lcp.py:48-213`, `lcp_physics/lcp/solvers/batch.py:70-231`).
Operands are *engine-assembled* (captured at `lcp_physics/physics/engines.py:81`) from
small sphere-drop and box-stack scenes, plus seeded random dense problems.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
import lcp_physics.physics.engines as engines  # noqa: E402
from lcp_physics.lcp.lcp import LCPFunction  # noqa: E402
from oracle.gen import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run_case(name, Q, p, G, h, A, b, F, max_iter, seed, meta=None):
    ins = [t.clone().detach().requires_grad_(t.nelement() > 0) for t in (Q, p, G, h, A, b, F)]
    captured = {}
    import lcp_physics.lcp.solvers.batch as B
    orig = B.forward

    def fwd(*a, **k):
        out = orig(*a, **k)
        captured["x"], captured["y"], captured["z"], captured["s"] = [None if o is None else o.clone() for o in out]
        return out

    B.forward = fwd
    try:
        z = LCPFunction(max_iter=max_iter, verbose=-1)(*ins)
    finally:
        B.forward = orig
    gen = torch.Generator().manual_seed(seed)
    dl = torch.randn(z.shape, generator=gen, dtype=torch.double)
    (z * dl).sum().backward()
    d = {"max_iter": np.int64(max_iter), "zhat": z.detach().numpy(), "dl_dz": dl.numpy(),
         "lam": captured["z"].numpy(), "slack": captured["s"].numpy(),
         "nu": captured["y"].numpy() if captured["y"] is not None else np.zeros((z.shape[0], 0))}
    for n, t in zip("QpGhAbF", ins):
        d[n] = t.detach().numpy()
        d["d" + n] = t.grad.numpy() if t.grad is not None else np.zeros(t.shape)
    if d["F"].shape[-1] > 200:
        # dF = dlam (x) lam = -dh (x) lam is rank one (`lcp.py:186`): verified here, not stored
        assert np.allclose(d["dF"], -d["dh"][:, :, None] * d["lam"][:, None, :], rtol=0, atol=1e-300)
        del d["dF"]
    if meta:
        d.update(meta)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items() if hasattr(v, "shape") and k in "QGAF"},
          "max|z|", np.abs(d["zhat"]).max())


def capture_engine_lcps(bodies, joints, nsteps, pick):
    """Step a reference World3D and capture the operands of selected LCP calls."""
    from sdf_physics.physics3d.world import World3D
    caught = []
    orig = engines.LCPFunction

    def spy(**kw):
        fn = orig(**kw)

        def call(*ops):
            caught.append((kw, [o.detach().clone() for o in ops]))
            return fn(*ops)
        return call

    w = World3D(bodies, joints)
    w.engine.lcp_solver = spy
    for _ in range(nsteps):
        w.step(fixed_dt=True)
    print('captured', len(caught), 'LCP calls, sizes', [c[1][2].shape[1] for c in caught])
    return [caught[i] for i in pick if i < len(caught)], w


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    # random dense QPs with a non-zero F (general B1 use)
    for k, (nB, nz, nineq, neq) in enumerate([(3, 6, 4, 3), (2, 12, 10, 6), (2, 10, 7, 0)]):
        g = torch.Generator().manual_seed(100 + k)
        L = torch.randn(nB, nz, nz, generator=g, dtype=torch.double)
        Q = L @ L.transpose(1, 2) + torch.eye(nz, dtype=torch.double)
        p = torch.randn(nB, nz, generator=g, dtype=torch.double)
        G = torch.randn(nB, nineq, nz, generator=g, dtype=torch.double)
        h = torch.rand(nB, nineq, generator=g, dtype=torch.double)
        A = torch.randn(nB, neq, nz, generator=g, dtype=torch.double) if neq else torch.tensor([])
        b = torch.randn(nB, neq, generator=g, dtype=torch.double) if neq else torch.tensor([])
        Fh = torch.randn(nB, nineq, nineq, generator=g, dtype=torch.double) * 0.1
        F = Fh @ Fh.transpose(1, 2) + 0.05 * torch.randn(nB, nineq, nineq, generator=g, dtype=torch.double)
        run_case("lcp_dense_%d" % k, Q, p, G, h, A, b, F, max_iter=10 if k else 20, seed=k)

    # engine-assembled operands: sphere on floor (config 2 sizes) and box stacks (config 3 structure)
    bodies, joints, _ = scenes.sphere_drop(seed=1, requires_grad=False)
    calls, _ = capture_engine_lcps(bodies, joints, 45, pick=[0, 2])
    for i, (kw, ops) in enumerate(calls):
        run_case("lcp_sphere_%d" % i, *ops, max_iter=kw["max_iter"], seed=10 + i)
    for nbox, tag, pick in ((1, "stack1", [0, 2]), (3, "stack3", [1])):
        bodies, joints, _ = scenes.box_stack(nbox=nbox, seed=2, requires_grad=False, vel_scale=1.0, push=3.0)
        calls, _ = capture_engine_lcps(bodies, joints, 3, pick=pick)
        for i, (kw, ops) in enumerate(calls):
            run_case("lcp_%s_%d" % (tag, i), *ops, max_iter=kw["max_iter"], seed=20 + i)
    # BASELINE configs[2] sizes: 8 bodies (nz = 48, neq = 6).  Seven aligned unit boxes stacked on
    # the floor make 4 contacts per directed pair = 56 contacts, nineq = 560; four stacked + three resting apart 40 contacts, nineq = 400.
    bodies, joints, _ = scenes.box_stack(nbox=7, seed=5, requires_grad=False, vel_scale=0.2, push=0.5, aligned=True)
    calls, _ = capture_engine_lcps(bodies, joints, 1, pick=[0])
    for i, (kw, ops) in enumerate(calls):
        run_case("lcp_stack7_%d" % i, *ops, max_iter=kw["max_iter"], seed=30 + i)
    bodies, joints, _ = scenes.box_stack(nbox=7, seed=6, requires_grad=False, vel_scale=0.2, push=0.5, aligned=True, stacked=4)
    calls, _ = capture_engine_lcps(bodies, joints, 1, pick=[0])
    for i, (kw, ops) in enumerate(calls):
        run_case("lcp_stack4p3_%d" % i, *ops, max_iter=kw["max_iter"], seed=40 + i)
    # an infeasible problem: rows g x <= -1 and -g x <= -1 cannot both hold.  The reference returns its best iterate
    # silently (verbose = -1) where verbose >= 0 would print INACC_ERR (batch.py:165-167, 229-230): residual > 1
    g = torch.Generator().manual_seed(77)
    nB, nz, m = 2, 6, 4
    L = torch.randn(nB, nz, nz, generator=g, dtype=torch.double)
    Q = L @ L.transpose(1, 2) + torch.eye(nz, dtype=torch.double)
    p = torch.randn(nB, nz, generator=g, dtype=torch.double)
    G0 = torch.randn(nB, m, nz, generator=g, dtype=torch.double)
    G = torch.cat([G0, -G0], dim=1)
    h = -torch.ones(nB, 2 * m, dtype=torch.double)
    F = torch.zeros(nB, 2 * m, 2 * m, dtype=torch.double)
    run_case("lcp_infeasible", Q, p, G, h, torch.tensor([]), torch.tensor([]), F, max_iter=20, seed=77, meta={"inaccurate": np.int64(1)})


if __name__ == "__main__":
    main()
