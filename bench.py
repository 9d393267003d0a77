#!/usr/bin/env python3
"""bench.py -- sim steps/s (fwd+bwd) of the batched differentiable SDF rigid-body stepper on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU (torch.distributed.run, or this script as its own launcher); scenes are independent, so every
  rank steps its own shard of B scenes (weak scaling, no collective inside a step, one trivial gather of the final poses and
  per-scene gradients at the end of the timed region).
Workloads (--config, numbered like BASELINE.json's list; 3 is the one the metric is quoted on and the default):
  2  configs[1]  256 sphere-drop scenes, dt halving and time-of-contact events
  3  configs[2]  1024 scenes x 8 SDF bodies (floor + 7-box stack with friction)
  4  configs[3]  512 scenes of demos/demo_meshsdf.py: floor, pole, a NEURAL SDF body (MFMA MLP inside the narrow phase)
  5  configs[4]  shape: 1024 contact-free single-body scenes per GPU (inertia fitting)
A "step" = one outer simulation step (World.step(fixed_dt=True)) of the whole batch, forward, plus its share of the
backward sweep of sum |pos_T|^2; the timed region is K forward steps followed by the full reverse sweep through them.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP64_PEAK_TFLOPS = 78.6      # vector and matrix fp64 alike: 256 CU x 128 flop/clk x 2.4 GHz
IGR_MAC_PER_POINT = 115456   # 5*128 + 6*128*128 + 123*128 + 128: one network evaluation (igr_mlp.hip)

DEFAULTS = {1: dict(batch=1, steps=50), 2: dict(batch=256, steps=200), 3: dict(batch=1024, steps=200), 4: dict(batch=512, steps=100), 5: dict(batch=1024, steps=200)}
WORKLOAD = {1: "configs[0]: 2-D ball on a pinned slab (Circle + Rect, analytic contact), batch %d, %d steps fwd + backward",
            2: "configs[1]: %d sphere-drop scenes (floor + SDF sphere, TOC on), %d steps fwd + reverse sweep",
            3: "configs[2]: floor + 7-box SDF stack with friction, %d scenes per GPU, %d steps fwd + reverse sweep",
            4: "configs[3]: demo_meshsdf scene (level-set floor, pole, neural-SDF body on the fp64 matrix cores), %d scenes per GPU, %d steps fwd + reverse sweep",
            5: "configs[4] shape: %d contact-free single-body scenes per GPU (X/Y/Z constraints, torque), %d steps fwd + reverse sweep"}


def lcp_algorithmic_bytes(nb, neq, fd, nc_per_scene):
    """Operands in + results out of one contact-LCP launch, bytes."""
    ND, NR = fd // 2, fd + 2
    NF = 3 * (1 + ND) + 8
    per_scene_fixed = 8 * (36 * nb + 6 * nb + neq * 6 * nb + neq) + 8 * (6 * nb + neq) + 4
    per_contact = 8 * NF + 8 + 8 * 2 * NR
    return float(np.sum(per_scene_fixed + per_contact * np.asarray(nc_per_scene, np.float64)))


def detect_algorithmic_bytes(E):
    """Compulsory traffic of one contact-detection launch group (overlap + narrow phase + compaction), bytes:
    per active directed pair  culling boxes of mesh a (48 B/run) + centroid/radius of the faces in the runs that
    pass (32 B/face) + triangle of every candidate (3 x 24 B + 12 B) + contacts out (14 x 8 B), plus both bodies' state."""
    st = E.get("pc_stats").astype(np.float64)
    cnt = E.get("pc_count").astype(np.float64)
    mesh_nf = E.get("mesh_nf")[E.get("mesh_id")].astype(np.float64)          # [B, nb]
    nb = E.nb
    if nb < 2:
        return 0.0
    a_of = np.repeat(np.arange(nb), nb - 1)                                   # mesh body of directed pair dp
    nch = np.ceil(mesh_nf[:, a_of] / 256.0)
    active = st[:, :, 0] > 0
    per_pair = nch * 48 + st[:, :, 0] * 256 * 32 + st[:, :, 1] * 84 + cnt * 112 + 2 * 20 * 8
    return float((per_pair * active).sum())


def lib_hash():
    from diffsdfsim_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]


def load_pmc(config):
    """Counter totals per launch from rocprofv3 --pmc passes of this same command (tools/pmc.sh -> profiles/r3_pmc_config<N>.json),
    valid only for the library build they were taken on (the file carries its hash)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_config%d.json" % config)))
    except Exception:
        return {}, "no PMC file for this config"
    if d.get("lib_sha256") != lib_hash():
        return {}, "PMC file is from another library build (%s): counters not reported" % d.get("lib_sha256")
    return d.get("kernels", {}), None


def pmc_sum(pmc, key, *frags):
    tot, hit = 0.0, False
    for k, v in pmc.items():
        if any(f in k for f in frags) and key in v:
            tot += v[key]; hit = True
    return tot if hit else None


def pmc_ratio(pmc, num, den, den_scale, *frags):
    """sum(num) / (den_scale * sum(den)) over all dispatches of the kernels that match (per-kernel averages x dispatch counts)."""
    a = b = 0.0
    for k, v in pmc.items():
        if any(f in k for f in frags) and num in v and den in v:
            n = v.get("dispatches", 1)
            a += v[num] * n; b += v[den] * n
    return a / (den_scale * b) if b > 0 else None


def cpu_baseline_step(spec, engine_kw, n_scenes, n_steps, threads):
    """The reference's WHOLE time step on the host cores: oracle/step_oracle.c (broad phase, Frank-Wolfe search, contact
    geometry, clustering + hull, assembly, dense PDIPM LCP forward, integration, accept / halve; plus the LCP's implicit
    backward once per solve -- a C restatement of the reference's Python, pinned by its goldens) on the first `n_scenes`
    scenes of THIS batch for `n_steps` steps from the start state: once on one thread (2 scenes), once with `threads` scenes
    in flight.  Returns seconds per scene-step (1 thread), wall seconds per scene-step (all threads), the solver's share."""
    from oracle import step_oracle as SO
    SO.build()
    floor = (np.ascontiguousarray(spec["meshes"][0][0], np.float64), np.ascontiguousarray(spec["meshes"][0][1], np.int32))
    kw = dict(dt=engine_kw["dt"], eps=engine_kw["eps"], tol=engine_kw["tol"], fric_dirs=engine_kw["fric_dirs"],
              strict_no_pen=engine_kw["strict_no_pen"], toc_diff=engine_kw["toc_diff"], hull="own", lcp_backward=True, shared={0: floor})

    def run(nthreads, scenes):
        ws = [SO.World(spec, s, **kw) for s in scenes]
        t0 = time.time()
        SO.run_many(ws, n_steps, nthreads)
        wall = time.time() - t0
        tm = [w.timers() for w in ws]
        cn = [w.counters() for w in ws]
        for w in ws:
            w.close()
        return wall / (len(ws) * n_steps), sum(t["solve"] for t in tm) / max(1e-30, sum(t["solve"] + t["detect"] for t in tm)), \
            sum(c["lcp_rows"] for c in cn) / max(1, sum(c["lcp_solves"] for c in cn)), sum(c["attempts"] for c in cn) / (len(ws) * n_steps)
    one, share, rows, att = run(1, list(range(min(2, n_scenes))))
    par, _, _, _ = run(threads, list(range(n_scenes)))
    return dict(one=one, par=par, lcp_share=share, lcp_rows=rows, attempts_per_step=att)


def visible_gpus():
    """Number of GPUs this process may use, WITHOUT touching HIP (the launcher parent must stay uninitialised: it starts
    children).  ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES if set, else the GPU nodes of the KFD topology
    (nodes with SIMDs; CPU nodes have simd_count 0)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n, top = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(top):
            try:
                txt = open(os.path.join(top, d, "properties")).read()
            except OSError:
                continue
            for line in txt.splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return 0
    return n


def self_launch(n):
    """One child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same variables
    torch.distributed.run sets), same argv.  The parent never imports torch and never touches HIP.  With fewer visible GPUs
    than ranks the ranks share devices and the collectives go through gloo (a rehearsal of the same code path, flagged in the
    JSON line).  The children are polled: when one dies the others are terminated instead of waiting in a rendezvous or a
    barrier for the process-group timeout."""
    import socket
    import subprocess
    ndev = visible_gpus()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        if ndev < n:
            env.setdefault("DSS_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code)
                for q in live:          # exactly the children started above
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def build_engine(args, rank, dev):
    from diffsdfsim_amd import scenes
    from diffsdfsim_amd.engine import BatchEngine, TorchBackend
    B, K, Wm, cfg = args.batch, args.steps, args.warmup, args.config
    be = TorchBackend(dev)
    def keep(E, spec, strict):      # what the CPU baseline needs to step the same scenes (bench.py: cpu_baseline_step)
        E.spec, E.spec_kw = spec, dict(dt=1.0 / 30, eps=1e-3, tol=1e-8, fric_dirs=8, strict_no_pen=strict, toc_diff=True)
        return E
    if cfg == 2:
        spec = scenes.sphere_drop(B, seed=1000 + rank)
        return keep(BatchEngine(spec, maxc=64, max_cand=1024, max_pc=32, max_sub=4 * (K + Wm) + 64, backend=be), spec, True)
    if cfg == 3:
        spec = scenes.box_stack(B, nbox=args.nbox, seed=1000 + rank, push=args.push)
        # strict_no_pen=False as in the reference's own long-running experiments (optim_sysid.py:104-131): a scene
        # whose penetration cannot be resolved by halving dt proceeds once dt < dt/2^10 (world.py:345-347) instead of
        # retrying forever, which with strict=True stalls the reference as well.
        # (--push: boxes that slide and tip collect more contacts than the stack at rest; the larger capacity takes the
        # streaming variant of the LCP kernel, so that option is not the configuration the metric is quoted on)
        return keep(BatchEngine(spec, maxc=128 if args.push == 0.0 else 256, max_cand=1024, max_pc=48, max_sub=int(1.5 * (K + Wm)) + 16,
                                strict_no_pen=False, backend=be), spec, False)
    if cfg == 4:
        spec = scenes.igr_pole(B, seed=1000 + rank)
        return BatchEngine(spec, maxc=256, max_cand=8192, max_pc=128, max_sub=3 * (K + Wm) + 64, backend=be)
    spec = scenes.inertia_spin(B, seed=1000 + rank)
    return BatchEngine(spec, maxc=8, max_cand=64, max_pc=8, max_sub=(K + Wm) + 16, backend=be)


def bench_config1(args):
    """BASELINE configs[0], the reference's own CPU case, on the device library (diffsdfsim_amd.physics2d): batch 1, a step is
    a handful of tiny launches with host decisions in between, so the number says what the latency of that chain is and
    nothing about the hardware.  W untimed runs, then one timed run of K steps forward + backward."""
    import torch
    from diffsdfsim_amd import _lib
    from diffsdfsim_amd.physics2d import Circle, Gravity, Rect, TotalConstraint, World
    _lib.lib()
    torch.cuda.set_device(0)
    K, Wm = args.steps, args.warmup

    def run():
        rad = torch.tensor(20.0, dtype=torch.double, requires_grad=True)
        floor = Rect([500, 600], [1000, 50], restitution=0.5, fric_coeff=0.9)
        ball = Circle([500, 480], rad, vel=[0, 30, 0], restitution=0.5, fric_coeff=0.9)
        ball.add_force(Gravity(g=100))
        w = World([floor, ball], [TotalConstraint(floor)], dt=1.0 / 30)
        for _ in range(K):
            w.step()
        (ball.pos ** 2).sum().backward()
        return w, float(rad.grad)
    for _ in range(max(1, min(Wm, 2))):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w, g = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "sim steps/sec (fwd+bwd), BASELINE configs[0]", "value": K / dt, "unit": "steps/s", "n_gpus": 1, "steps": K, "warmup": Wm,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOAD[1] % (1, K), "lcp_solves": w.lcp_calls, "substeps": len(w.trajectory), "d_loss_d_rad": g,
                   "lib_sha256": lib_hash()},
        "roofline": {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                     "note": "one scene, a few 64-lane launches per step and a host decision after each: no roofline applies; the case is "
                             "in the suite for parity (tests/test_contacts2d_gpu.py), not for throughput"},
        "cpu_baseline": {"value": None, "quoted_value": 234.0, "measured": False, "unit": "steps/s", "cores": 1, "kind": "reference",
                         "sample": "the reference itself on one core of the 8-vCPU build container (SURVEY.md section 6: 50 steps fwd+bwd of this "
                                   "scene); not timed on this host -- the reference cannot travel"}}))


def bench_config5(args, dist, rank, world, dev, cdev, backend):
    """BASELINE configs[4] as the reference runs it (experiments/inertia_fitting/optim_shapespace.py:71-92, 136-250): EVERY
    optimisation iteration rebuilds the world from the latent codes -- latent -> 128^3 samples of the network (fp64 matrix cores)
    -> marching cubes -> inertia by volume integrals over the mesh -> single-body scene with X/Y/Z constraints -> 200 steps with a
    torque along a random direction for t < 0.3 -> loss on the final angular velocity -> backward to the latent (reverse sweep,
    inertia adjoint, MeshSDF).  One timed "iteration" = that whole chain for B scenes; a "step" = one of its K simulation steps.
    Trajectories (K, B, 13) are gathered at the end, as north_star says."""
    import torch
    from diffsdfsim_amd import experiments as X
    from diffsdfsim_amd import igr, meshsdf, scenes, sharding
    B, K = args.batch, args.steps
    torch.cuda.set_device(dev)
    packed = igr.pack_weights(*scenes.geometric_init_weights(0, 0.5), device=dev)
    r = np.random.default_rng(1000 + rank)
    lat0 = 0.1 * r.standard_normal((B, 2))
    dirs = r.standard_normal((B, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dt = 1.0 / 30

    def iteration(nsc, record):
        lat = torch.tensor(lat0[:nsc], dtype=torch.float64, requires_grad=True)
        meshsdf.GRID_EVENTS = [] if record else None
        t = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w = X.spin_world(lat, dirs[:nsc], packed, scale=1.0, mass=1.0, res=args.res, steps=K, device=dev, keep_meshes=False)
        torch.cuda.synchronize(); t["build_s"] = time.perf_counter() - t0
        tq = torch.cat([torch.as_tensor(0.5 * dirs[:nsc]), torch.zeros(nsc, 3, dtype=torch.float64)], 1)[:, None]
        traj = []
        for k in range(K):
            w.params["fext"] = tq if k * dt < 0.3 else torch.zeros_like(tq)      # the torque stops at t = 0.3 (optim_shapespace.py:83-88)
            w.step(keep_undo=False)
            traj.append(torch.cat([w.pose[:, 0], w.vel[:, 0]], dim=1))
        torch.cuda.synchronize(); t["spin_s"] = time.perf_counter() - t0 - t["build_s"]
        loss = (w.vel[:, 0, :3] ** 2).sum()
        loss.backward()
        torch.cuda.synchronize(); t["backward_s"] = time.perf_counter() - t0 - t["build_s"] - t["spin_s"]
        tr = torch.stack(traj, dim=1).reshape(nsc, -1)                            # [B, K * 13]
        if dist is not None:
            tr = sharding.gather_scenes(tr.to(cdev), nsc * world, dist)
        torch.cuda.synchronize(); t["total_s"] = time.perf_counter() - t0
        ev, meshsdf.GRID_EVENTS = meshsdf.GRID_EVENTS, None
        return t, ev, lat.grad, [m[1] if isinstance(m[1], int) else len(m[1]) for m in w.meshes]
    for _ in range(max(1, min(args.warmup, 1))):
        iteration(min(B, 8), False)                  # warm-up: the same chain on eight scenes (kernels loaded, allocator primed)
    if dist is not None:
        dist.barrier()
    t, ev, grad, nfaces = iteration(B, True)
    tot = t["total_s"]
    if dist is not None:
        tt = torch.tensor([tot], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tot = float(tt.item())
    if rank != 0:
        return
    ms = np.array([a.elapsed_time(b) for a, b, _n in ev])
    pts = float(sum(n for _a, _b, n in ev))
    flops = 2.0 * IGR_MAC_PER_POINT * pts
    ach = flops / (ms.sum() * 1e-3) / 1e12
    print(json.dumps({
        "metric": "sim steps/sec (fwd+bwd), BASELINE configs[4]", "value": world * K / tot, "unit": "steps/s", "n_gpus": world, "steps": K,
        "warmup": args.warmup, "ms_per_step": 1e3 * tot / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "configs[4]: inertia fitting in shape space, %d scenes per GPU: per iteration latent -> %d^3 network grid -> marching cubes "
                               "-> mesh inertia -> %d-step spin (torque for t < 0.3, no contacts) -> reverse sweep to the latent" % (B, args.res, K),
                   "scenes_per_gpu": B, "grid_res": args.res, "mesh_faces_mean": float(np.mean(nfaces)),
                   "world_build_s": t["build_s"], "spin_forward_s": t["spin_s"], "backward_s": t["backward_s"], "iteration_s": tot,
                   "scene_steps_per_s": world * B * K / tot, "latent_grad_finite": bool(torch.isfinite(grad).all()), "lib_sha256": lib_hash(),
                   "parallelism": "scene-sharded x%d; one all_gather (%s) of the (K, B, 13) trajectories" % (world, backend or "none")},
        "roofline": {"bound": "mfma", "kernel": "igr_query_kernel, value-only (fp64 v_mfma_f64_16x16x4): the %d^3 grid of every scene's level-set mesh" % args.res,
                     "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                     "avg_launch_ms": float(ms.mean()), "launches": len(ms), "algorithmic_flops_per_launch": flops / len(ms),
                     "share_of_iteration": float(ms.sum() * 1e-3 / tot),
                     "note": "flops = 2 x 115456 MAC per grid point; HIP events around every grid evaluation of the timed iteration"},
        "roofline_second_kernel": {"bound": "latency", "kernel": "stepper, contact-free branch (assemble / 9x9 solve / integrate / decide)", "achieved": None,
                                   "peak": None, "unit": None, "frac": None, "traffic": None, "ms_per_step": 1e3 * t["spin_s"] / K,
                                   "note": "six small launches per step: launch-bound (python bench.py --config 5 --spin-only times it alone)"},
        "cpu_baseline": {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "measured": False,
                         "sample": "no CPU leg: the C port of the reference's step has no neural SDF, marching cubes or X/Y/Z constraints"}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(DEFAULTS))
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU")
    ap.add_argument("--nbox", type=int, default=7)
    ap.add_argument("--cpu-sample", type=int, default=32, help="scenes of the batch the CPU baseline steps (at least one per core)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="steps per sampled scene in the CPU baseline")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (capped by the affinity mask)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lockstep", action="store_true", help="one BatchEngine.step() per outer step for the whole batch (every scene waits for the slowest of each step) instead of BatchEngine.run(K)")
    ap.add_argument("--push", type=float, default=0.0, help="random lateral start velocity (0 = BASELINE config 3 as specified)")
    ap.add_argument("--spin-only", action="store_true", help="config 5: time only the 200-step spin of given inertias (round 2's line), "
                    "not the per-iteration rebuild latent -> 128^3 network grid -> marching cubes -> inertia")
    ap.add_argument("--res", type=int, default=128, help="config 5: grid resolution of the level-set mesh (the reference: 128)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULTS[args.config]["steps"]
    if args.batch is None:
        args.batch = DEFAULTS[args.config]["batch"]

    if args.config == 1:
        return bench_config1(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N children, one rank per GPU, started BEFORE anything in
        # this process touches the GPU (no exec of an initialised process); rank 0's JSON line is the children's stdout
        sys.exit(self_launch(args.gpus))

    if os.environ.get("DSS_BENCH_FAIL_RANK") == os.environ.get("RANK", "0") and "WORLD_SIZE" in os.environ:
        sys.exit(7)      # tests/test_bench_launcher.py: a rank that dies before the rendezvous
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm) over xGMI on a real node; DSS_DIST_BACKEND=gloo rehearses the same code path with
        # several ranks sharing one GPU (collectives then go through host copies)
        backend = os.environ.get("DSS_DIST_BACKEND", "nccl")
        if "DSS_DIST_BACKEND" not in os.environ and 0 < torch.cuda.device_count() < world:
            backend = "gloo"      # more ranks than devices (a rehearsal under torchrun on a small box): RCCL refuses two ranks per GPU
        kw = {}
        if not os.environ.get("DSS_BENCH_DRYRUN") and torch.cuda.device_count() > 0:
            torch.cuda.set_device(local % torch.cuda.device_count())     # RCCL binds its communicator to the current device
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local % torch.cuda.device_count())
        import datetime
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("DSS_DIST_TIMEOUT", 180))), **kw)
        if os.environ.get("DSS_BENCH_DRYRUN"):   # launcher + rendezvous check without a GPU (tests/test_bench_launcher.py)
            mine = torch.tensor([rank], dtype=torch.int64)
            got = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)
            if rank == 0:
                print(json.dumps({"dryrun": True, "n_gpus": world, "ranks": [int(g) for g in got], "backend": backend}))
            dist.destroy_process_group()
            return
    else:
        dist = None
        backend = None
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if backend == "gloo" else dev      # where collective buffers live

    from diffsdfsim_amd import _lib
    from diffsdfsim_amd import world_abi as abi
    _lib.lib()   # fail loudly if the HIP library is missing

    if args.config == 5 and not args.spin_only:
        return bench_config5(args, dist, rank, world, dev, cdev, backend)
    B, K, Wm, cfg = args.batch, args.steps, args.warmup, args.config
    t_build = time.time()
    E = build_engine(args, rank, dev)
    t_build = time.time() - t_build
    neural = E.igr_items_cap > 0

    def loss_adjoint():
        adj = E._adjoint()
        for k in ("a_pose", "a_vel", "a_geom", "g_mass", "g_inertia", "g_rest", "g_fric", "g_fext", "g_prm"):
            adj[k].zero_()
        adj["a_pose"][:, :, 4:] = 2.0 * E.arr["pose"][:, :, 4:]
        adj["cur_slot"].copy_(E.arr["nsub"] - 1)

    # warm-up: W steps forward + their backward (untimed)
    att = 0
    if args.lockstep:
        for _ in range(Wm):
            att += E.step()
    else:
        att = E.run(Wm)
    loss_adjoint()
    E.adj["lo_slot"].zero_()
    E.backward_sweep(int(E.arr["nsub"].max().item()))

    def gather_results():
        # "final trivial gather" of the shard results: final poses and the per-scene parameter gradients,
        # through the same diffsdfsim_amd.sharding.gather_scenes that tests/test_shard_cpu.py checks over gloo
        from diffsdfsim_amd import sharding
        final = torch.cat([E.arr["pose"].reshape(B, -1), E.adj["g_prm"].reshape(B, -1), E.adj["g_mass"].reshape(B, -1),
                           E.adj["g_fric"].reshape(B, -1)], dim=1)
        if dist is not None:
            return sharding.gather_scenes(final.to(cdev), B * world, dist)
        return final
    gather_results()      # part of the warm-up: the first concatenation loads its kernel, the first collective sets up its channels
    torch.cuda.synchronize()

    # event pairs around every LCP launch and every detection launch group of the timed region
    ev = []

    def fresh_pair():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        return a, b

    est = {2: 12, 3: 2, 4: 3, 5: 2}[cfg]
    pool = [fresh_pair() for _ in range(2 * est * K + 256)]
    # neural narrow phase: event pairs around the network evaluations of one attempt in IGR_EV_EVERY (the matrix-core kernel)
    IGR_EV_EVERY = 4
    nev = 4 * (abi.IGR_ROUNDS + 1)
    igr_events = [torch.cuda.Event(enable_timing=True) for _ in range(nev)] if neural else []
    for e in igr_events:
        e.record()
    import ctypes
    igr_ev_arr = (ctypes.c_void_p * nev)(*[e.cuda_event for e in igr_events]) if neural else None
    igr_attempts = []
    igr_ms, igr_pts, igr_est = [], [], []           # per sampled launch: duration, (points, list), the grid-size hint it ran with
    qn_total = torch.zeros(2 * (abi.IGR_ROUNDS + 2), dtype=torch.int64, device=dev) if neural else None
    torch.cuda.synchronize()
    lo = E.arr["nsub"].clone()

    L = E.be.lib

    def timed_steps(nsteps):
        # BatchEngine.run(nsteps) (every scene goes through its own outer steps: DssWorld.steps_left) -- or, --lockstep,
        # BatchEngine.step() -- with an event pair per attempt
        if nsteps > 1:
            if "steps_left" not in E.arr:
                E.arr["steps_left"] = E.be.zeros((E.B,), np.int32)
            E.arr["steps_left"][...] = nsteps
            E.W.steps_left = E.be.ptr(E.arr["steps_left"])
        E._check(L.dss_step_begin(ctypes.byref(E.W), E.be.stream()), "dss_step_begin")
        n, k = E.B, 0
        while n > 0:
            if 2 * len(ev) + 1 >= len(pool):
                pool.extend(fresh_pair() for _ in range(256))
            a, b = pool[2 * len(ev)]
            c, d = pool[2 * len(ev) + 1]
            E.W.ev_lcp_start, E.W.ev_lcp_stop = a.cuda_event, b.cuda_event
            E.W.ev_np_start, E.W.ev_np_stop = c.cuda_event, d.cuda_event
            sample = neural and len(ev) % IGR_EV_EVERY == 0
            hint_was = E.igr_hint.copy() if (sample and E.igr_hint is not None) else None
            E.W.igr_ev = ctypes.cast(igr_ev_arr, ctypes.c_void_p) if sample else None
            E._check(L.dss_step_attempt(ctypes.byref(E.W), ctypes.c_void_p(E.be.ptr(E.lcp_ws)),
                                        ctypes.c_size_t(E.lcp_ws_bytes), E.be.stream()), "dss_step_attempt")
            ev.append((a, b, c, d))
            if neural:
                qn_total.add_(E.arr["igr_qn"])
            n = E.be.read_int(E.arr["n_active"])        # (the one host read of an attempt; the stream is idle afterwards)
            if n & (1 << 30):
                E._raise_overflow()
            E._update_igr_hint(k == 0, n == 0)
            if sample:
                igr_attempts.append((len(ev) - 1, len(igr_ms)))
                qn = E.get("igr_qn")
                for r in range(1, abi.IGR_ROUNDS + 1):      # one launch per round serves the value and the gradient list
                    if qn[2 * r] + qn[2 * r + 1] > 0:
                        igr_ms.append(igr_events[4 * r].elapsed_time(igr_events[4 * r + 1]))
                        igr_pts.append((int(qn[2 * r]), int(qn[2 * r + 1])))
                        igr_est.append((int(hint_was[2 * r]), int(hint_was[2 * r + 1])) if hint_was is not None else (-1, -1))
            k += 1
        E.W.steps_left = None
        return k

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    att = 0
    if args.lockstep:
        for _ in range(K):
            att += timed_steps(1)
    else:
        att = timed_steps(K)
    E.W.ev_lcp_start, E.W.ev_lcp_stop, E.W.ev_np_start, E.W.ev_np_stop = None, None, None, None
    E.W.igr_ev = None
    t_fwd = time.perf_counter()
    loss_adjoint()
    E.adj["lo_slot"].copy_(lo)
    # one reverse sweep iteration undoes one recorded sub-step per scene: as many as the longest tape segment of the timed region
    # (rejected attempts left nothing on the tape)
    n_sweeps = int((E.arr["nsub"] - lo).max().item())
    E.backward_sweep(n_sweeps)
    gather_results()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    fwd_s = t_fwd - t0
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        return
    lcp_ms = np.array([a.elapsed_time(b) for a, b, c, d in ev])
    det_ms = np.array([c.elapsed_time(d) for a, b, c, d in ev])
    nc = E.get("nc")
    overflow = int(E.get("overflow").max())
    pmc, pmc_note = load_pmc(cfg)

    def traffic(*frags):
        f, w = pmc_sum(pmc, "FETCH_SIZE_KB_avg", *frags), pmc_sum(pmc, "WRITE_SIZE_KB_avg", *frags)
        return None if f is None or w is None else (f + w) * 1024.0

    def fp64_flops(*frags):
        """fp64 flops per launch from the SQ instruction counters (wave instructions x 64 lanes; FMA = 2)."""
        a, m, f = (pmc_sum(pmc, "SQ_INSTS_VALU_%s_F64_avg" % k, *frags) for k in ("ADD", "MUL", "FMA"))
        return None if a is None or m is None or f is None else 64.0 * (a + m + 2.0 * f)

    def roof_valu(kernel, ms, algo_bytes, note, frags):
        """A kernel that is fp64-VALU / issue bound, not HBM bound: fp64 flops (PMC, if taken on this build) over the measured
        launch time against the 78.6 TFLOP/s vector peak; the HBM view (algorithmic bytes / time) rides along."""
        fl = fp64_flops(*frags)
        t = ms.mean() * 1e-3
        hbm = algo_bytes / t / 1e9
        r = {"bound": "fp64_valu", "kernel": kernel, "avg_launch_ms": float(ms.mean()), "launches": len(ms),
             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": traffic(*frags),
             "algorithmic_bytes_per_launch": algo_bytes, "hbm_view": {"achieved_GBps": hbm, "frac_of_8TBps": hbm / HBM_PEAK_GBS},
             "note": note}
        if fl is None:
            r.update(achieved=None, frac=None, flops_per_launch=None, flops_note=pmc_note or "no fp64 instruction counters in the PMC file")
        else:
            r.update(achieved=fl / t / 1e12, frac=fl / t / 1e12 / FP64_PEAK_TFLOPS, flops_per_launch=fl,
                     flops_source="counter upper bound: SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 wave instructions x 64 lanes (masked-off lanes "
                                  "count), averaged over the dispatches of a rocprofv3 --pmc pass of this library (profiles/, "
                                  "lib_sha256), divided by the launch time measured live in this run")
            vb = pmc_sum(pmc, "valu_util_per_simd", frags[0])     # of the group's main kernel
            if vb is not None:
                r["valu_util_per_simd"] = vb     # cycles in which a SIMD issues a VALU instruction / kernel cycles, chip average
        return r

    r_lcp = roof_valu("lcp_contact_forward_reg_kernel", lcp_ms, lcp_algorithmic_bytes(E.nb, E.neq, E.fd, nc),
                      "one scene per wavefront, reduced KKT and interior-point state in registers: bound by fp64 issue of a "
                      "serial chain, not by HBM (operands are 35 KB per scene)", ("lcp_contact_forward",))
    r_det = roof_valu("narrowphase_kernel (+overlap_kernel, compact_contacts_kernel)", det_ms, detect_algorithmic_bytes(E),
                      "Frank-Wolfe / SDF evaluation: fp64 VALU bound (IEEE div/sqrt sequences), not HBM",
                      ("narrowphase_kernel<false>", "overlap_kernel", "compact_contacts"))
    extra = {}
    if neural and igr_ms:
        ms = np.array(igr_ms)
        flops = np.array([2.0 * IGR_MAC_PER_POINT * (nv_ + 4 * ng_) for nv_, ng_ in igr_pts])
        tot = qn_total.cpu().numpy()
        nv, ng = int(tot[2::2].sum()), int(tot[3::2].sum())
        ach = flops.sum() / (ms.sum() * 1e-3) / 1e12
        big = flops >= np.percentile(flops, 90)
        r_igr = {"bound": "mfma", "kernel": "igr_query2_kernel (fp64 v_mfma_f64_16x16x4)", "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic("igr_query2_kernel"),
                 "avg_launch_ms": float(ms.mean()), "launches_sampled": len(ms),
                 "algorithmic_flops_per_launch": float(flops.mean()),
                 "largest_decile_launches": {"TFLOPs": float(flops[big].sum() / (ms[big].sum() * 1e-3) / 1e12),
                                             "avg_points_value_list": float(np.mean([a for (a, g_), b in zip(igr_pts, big) if b])),
                                             "avg_points_gradient_list": float(np.mean([g_ for (a, g_), b in zip(igr_pts, big) if b]))},
                 "points_evaluated": {"value_only": nv, "with_gradient": ng, "per_step": (nv + ng) / K,
                                      "value_list_by_round": [int(x) for x in tot[2::2][:12]], "gradient_list_by_round": [int(x) for x in tot[3::2][:12]]},
                 "mfma_busy_frac": pmc_ratio(pmc, "SQ_VALU_MFMA_BUSY_CYCLES_avg", "GRBM_GUI_ACTIVE_avg", 1024.0 / 8.0, "igr_query2_kernel"),
                 "sampled_attempts": {"detection_ms_mean": float(np.mean([det_ms[a] for a, _ in igr_attempts])),
                                      "network_ms_mean": float(ms.sum() / max(1, len(igr_attempts))),
                                      "detection_ms_all_attempts_mean": float(det_ms.mean())},
                 "detection_ms_first_attempts": [round(float(x), 2) for x in det_ms[:48]],
                 "slowest_launches": [{"ms": float(ms[i]), "points_value_list": igr_pts[i][0], "points_gradient_list": igr_pts[i][1], "hints": list(igr_est[i])}
                                      for i in np.argsort(-ms)[:6]],
                 "note": "network evaluations of the neural narrow phase, one launch per query round (value list + gradient list); flops = "
                         "2 x 115456 MAC per point (x4 with the three xyz tangents); events around one attempt in %d" % IGR_EV_EVERY}
        dominant, other = r_igr, r_det
        extra["roofline_third_kernel"] = r_lcp
    else:
        dominant, other = (r_det, r_lcp) if det_ms.mean() >= lcp_ms.mean() else (r_lcp, r_det)
    res = {
        "metric": "sim steps/sec (fwd+bwd), 1024 batched 3D scenes x 8 SDF bodies" if cfg == 3 else
                  "sim steps/sec (fwd+bwd), BASELINE configs[%d]" % (cfg - 1),
        "value": world * K / dt, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOAD[cfg] % (B, K),
                   "scenes_per_gpu": B, "bodies": E.nb, "contacts_per_scene_mean": float(nc.mean()),
                   "contacts_per_scene_max": int(nc.max()), "attempts": att, "reverse_sweeps": n_sweeps, "stepping": "lock-step (one step() per outer step)" if args.lockstep else "free-running (run(K): DssWorld.steps_left)", "lcp_iters_mean": float(E.get("lcp_iters").mean()),
                   "substeps_mean": float((E.get("nsub") - E.be.to_numpy(lo)).mean()),
                   "scene_steps_per_s": world * B * K / dt, "capacity_overflow": overflow,
                   "forward_s": fwd_s, "backward_s": dt - fwd_s, "world_build_s": t_build,
                   "lib_sha256": lib_hash(),
                   "parallelism": "scene-sharded x%d, no collective in step; one all_gather (%s) of final poses + per-scene "
                                  "gradients; %d physical device(s)" % (world, backend or "none", torch.cuda.device_count())},
        "roofline": dominant, "roofline_second_kernel": other,
    }
    res.update(extra)
    if pmc_note:
        res["config"]["pmc_note"] = pmc_note
    if not args.no_cpu:
        # host cores: this process's share of the box (a one-GPU box is given 16 cores' worth, whatever the affinity mask says)
        threads = max(1, min(args.cpu_threads, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        if cfg in (2, 3):
            n_sc = min(B, max(threads, args.cpu_sample))
            n_st = args.cpu_steps if cfg == 3 else 20 * args.cpu_steps      # (config 2: the spheres land after 4-12 steps; cover the bounces)
            c = cpu_baseline_step(E.spec, E.spec_kw, n_sc, n_st, threads)
            res["cpu_baseline"] = {
                "value": 1.0 / (B * c["par"]), "unit": "steps/s", "cores": threads, "kind": "port", "measured": True,
                "scope": "whole step forward (detection + assembly + LCP + integration + accept/halve) + the LCP's implicit backward per solve",
                "not_included": "the adjoint of the contact geometry / integration (torch autograd in the reference): the CPU figure is therefore an "
                                "UPPER bound on a full fwd+bwd CPU rate, the GPU/CPU ratio a lower bound",
                "s_per_scene_step_1_thread": c["one"], "wall_s_per_scene_step_at_%d_threads" % threads: c["par"],
                "scene_steps_per_s_all_cores": 1.0 / c["par"], "lcp_share_of_cpu_time": c["lcp_share"], "lcp_rows_mean": c["lcp_rows"],
                "attempts_per_step": c["attempts_per_step"],
                "gpu_over_cpu_all_cores": (world * K / dt) * B * c["par"] / world,
                "reference_python_s_per_scene_step": {"value": 0.45, "where": "the reference itself (imported, torch CPU + LAPACK, float64) on a floor + 7-box "
                                                      "stack, forward + backward, one core of the 8-vCPU build container (SURVEY.md section 6); not timed "
                                                      "on this host -- the reference cannot travel"},
                "sample": "oracle/step_oracle.c + lcp_oracle.c (C port of the reference's step, blocked LU, OpenMP over scenes): the first %d scenes of "
                          "this batch x %d steps from the start state, scaled to %d scenes (scenes are independent)" % (n_sc, n_st, B)}
        else:
            res["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": threads, "kind": "port", "measured": False,
                                   "sample": "the C port of the reference's step knows box / sphere / cylinder bodies pinned by TotalConstraint3D "
                                             "(configs[1], configs[2]); it has no neural SDF body and no X/Y/Z constraint: no CPU leg for this config"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
