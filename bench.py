#!/usr/bin/env python3
"""bench.py -- sim steps/s (fwd+bwd) of the batched differentiable SDF rigid-body stepper on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU; scenes are independent so every rank
  steps its own shard of B scenes (weak scaling, no collective inside a step, one trivial gather of the
  final poses at the end of the timed region).
Workload = BASELINE.json configs[2]: 1024 scenes x 8 SDF bodies (floor + 7-box stack with friction).
A "step" = one outer simulation step (World.step(fixed_dt=True)) of the whole batch, forward, plus its
share of the backward sweep of sum |pos_T|^2; the timed region is K forward steps followed by the full
reverse sweep through them.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


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
    pass (32 B/face) + triangle of every candidate (3 x 24 B + 12 B) + its Frank-Wolfe record written and read once
    (15 x 8 B x 2) + contacts out (14 x 8 B), plus both bodies' state."""
    st = E.get("pc_stats").astype(np.float64)
    cnt = E.get("pc_count").astype(np.float64)
    mesh_nf = E.get("mesh_nf")[E.get("mesh_id")].astype(np.float64)          # [B, nb]
    nb = E.nb
    a_of = np.repeat(np.arange(nb), nb - 1)                                   # mesh body of directed pair dp
    nch = np.ceil(mesh_nf[:, a_of] / 256.0)
    active = st[:, :, 0] > 0
    per_pair = nch * 48 + st[:, :, 0] * 256 * 32 + st[:, :, 1] * (84 + 240) + cnt * 112 + 2 * 20 * 8
    return float((per_pair * active).sum())


def cpu_baseline(E, n_sample, threads):
    """Reference algorithm on the host: dense PDIPM LCP fwd+bwd (oracle/lcp_oracle.c, a port of
    batch.py / lcp.py) on the operands the GPU just solved, for a bounded sample of scenes."""
    from oracle import lcp_expand as X
    from oracle import lcp_oracle as O
    O.build()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    P = dict(Mblk=E.get("Mblk"), pvec=E.get("pvec"), A=E.get("Je"), bvec=np.zeros((E.B, E.neq)), cop=E.get("cop"),
             cbody=E.get("cop_body"), nc=E.be.to_numpy(E.adj["bw_nc"]).copy(), nb=E.nb, neq=E.neq, maxc=E.maxc, fd=E.fd)
    ncs = P["nc"]
    pick = np.argsort(ncs)[len(ncs) // 2 - n_sample // 2: len(ncs) // 2 + (n_sample + 1) // 2]   # median-sized scenes
    nineq_max = int(ncs[pick].max()) * (E.fd + 2)
    ops = []
    for s in pick:   # pad to a common nineq with inert rows (h = 1, G = 0) so one batched call covers them
        Q, p, G, h, A, b, F = X.expand_dense(P, int(s))
        k = nineq_max - len(h)
        G = np.vstack([G, np.zeros((k, G.shape[1]))]); h = np.concatenate([h, np.ones(k)])
        F = np.pad(F, ((0, k), (0, k)))
        ops.append((Q, p, G, h, A, b, F))
    Q, p, G, h, A, b, F = (np.stack(o) for o in zip(*ops))
    t0 = time.time()
    z, lam, sl, nu, it, st = O.forward(Q, p, G, h, A, b, F, max_iter=10, check_spd=True)
    O.backward(Q, G, A, F, z, lam, sl, nu, np.ones_like(z))
    dt = time.time() - t0
    return dt / len(pick), len(pick), nineq_max


def self_launch(n):
    """One child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same variables
    torch.distributed.run sets), same argv.  With fewer visible GPUs than ranks the ranks share devices and the
    collectives go through gloo (a rehearsal of the same code path, flagged in the JSON line)."""
    import socket
    import subprocess
    import torch   # importing torch and counting devices does not initialise the GPU
    ndev = torch.cuda.device_count()
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
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="scenes per GPU")
    ap.add_argument("--nbox", type=int, default=7)
    ap.add_argument("--cpu-sample", type=int, default=16)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--push", type=float, default=0.0, help="random lateral start velocity (0 = BASELINE config 3 as specified)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N children, one rank per GPU, started BEFORE anything in
        # this process touches the GPU (no exec of an initialised process); rank 0's JSON line is the children's stdout
        sys.exit(self_launch(args.gpus))

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
        dist.init_process_group(backend, rank=rank, world_size=world)
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

    from diffsdfsim_amd import _lib, scenes
    from diffsdfsim_amd.engine import BatchEngine, TorchBackend
    _lib.lib()   # fail loudly if the HIP library is missing

    B, K, Wm = args.batch, args.steps, args.warmup
    spec = scenes.box_stack(B, nbox=args.nbox, seed=1000 + rank, push=args.push)
    # strict_no_pen=False as in the reference's own long-running experiments (optim_sysid.py:104-131): a scene
    # whose penetration cannot be resolved by halving dt proceeds once dt < dt/2^10 (world.py:345-347) instead of
    # retrying forever, which with strict=True stalls the reference as well.
    E = BatchEngine(spec, maxc=128, max_cand=1024, max_pc=48, max_sub=int(1.5 * (K + Wm)) + 16,
                    strict_no_pen=False, backend=TorchBackend(dev))

    def loss_adjoint():
        adj = E._adjoint()
        for k in ("a_pose", "a_vel", "a_geom", "g_mass", "g_inertia", "g_rest", "g_fric", "g_fext", "g_prm"):
            adj[k].zero_()
        adj["a_pose"][:, :, 4:] = 2.0 * E.arr["pose"][:, :, 4:]
        adj["cur_slot"].copy_(E.arr["nsub"] - 1)

    # warm-up: W steps forward + their backward (untimed)
    att = 0
    for _ in range(Wm):
        att += E.step()
    loss_adjoint()
    E.adj["lo_slot"].zero_()
    E.backward_sweep(att)
    torch.cuda.synchronize()

    # event pairs around every LCP launch of the timed region
    ev = []

    def fresh_pair():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        return a, b

    pool = [fresh_pair() for _ in range(8 * K + 128)]
    torch.cuda.synchronize()
    lo = E.arr["nsub"].clone()
    nc_hist = []

    import ctypes
    L = E.be.lib

    def timed_step():
        # BatchEngine.step with an event pair per attempt
        E._check(L.dss_step_begin(ctypes.byref(E.W), E.be.stream()), "dss_step_begin")
        n, k = E.B, 0
        while n > 0:
            a, b = pool[2 * len(ev)]
            c, d = pool[2 * len(ev) + 1]
            E.W.ev_lcp_start, E.W.ev_lcp_stop = a.cuda_event, b.cuda_event
            E.W.ev_np_start, E.W.ev_np_stop = c.cuda_event, d.cuda_event
            E._check(L.dss_step_attempt(ctypes.byref(E.W), ctypes.c_void_p(E.be.ptr(E.lcp_ws)),
                                        ctypes.c_size_t(E.lcp_ws_bytes), E.be.stream()), "dss_step_attempt")
            ev.append((a, b, c, d))
            n = E.be.read_int(E.arr["n_active"])
            if n & (1 << 30):
                E._raise_overflow()
            k += 1
        return k

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    att = 0
    for _ in range(K):
        att += timed_step()
    E.W.ev_lcp_start, E.W.ev_lcp_stop, E.W.ev_np_start, E.W.ev_np_stop = None, None, None, None
    loss_adjoint()
    E.adj["lo_slot"].copy_(lo)
    E.backward_sweep(att)
    # "final trivial gather" of the shard results: final poses and the per-scene parameter gradients
    final = torch.cat([E.arr["pose"].reshape(B, -1), E.adj["g_prm"].reshape(B, -1), E.adj["g_mass"].reshape(B, -1),
                       E.adj["g_fric"].reshape(B, -1)], dim=1)
    if dist is not None:
        fin = final.to(cdev)
        out = [torch.empty_like(fin) for _ in range(world)]
        dist.all_gather(out, fin)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
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

    # HBM traffic per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE collected in
    # separate passes, profiles/r1_pmc_traffic.json); KB -> bytes, no access-width correction applied (MI355X guide:
    # FETCH_SIZE under-reports wide streaming reads by 2x; these kernels read 8-byte strided fields, uncalibrated).
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
    except Exception:
        pass

    def traffic(*frags):
        tot = 0.0
        for k, v in pmc.items():
            if any(f in k for f in frags):
                tot += (v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024.0
        return tot or None

    def roof(kernel, ms, algo, note, tr):
        ach = algo / (ms.mean() * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": tr, "avg_launch_ms": float(ms.mean()), "launches": len(ms),
                "algorithmic_bytes_per_launch": algo, "note": note}

    r_lcp = roof("lcp_contact_forward_reg_kernel<4>", lcp_ms, lcp_algorithmic_bytes(E.nb, E.neq, E.fd, nc),
                 "one wavefront per scene and SIMD, KKT and IPM state in registers: bound by the serial fp64 issue latency "
                 "of one wavefront, not by HBM; traffic above the algorithmic bytes is register spill around the "
                 "factorisation (DESIGN.md section 5)",
                 traffic("lcp_contact_forward"))
    r_det = roof("narrowphase_kernel (+overlap_kernel, compact_contacts_kernel)", det_ms, detect_algorithmic_bytes(E),
                 "Frank-Wolfe / SDF evaluation: fp64 VALU bound (61 % VALU-busy, SQ_ACTIVE_INST_VALU; IEEE div/sqrt "
                 "sequences), not HBM (DESIGN.md section 5)",
                 traffic("narrowphase_kernel", "overlap_kernel", "compact_contacts"))
    dominant, other = (r_det, r_lcp) if det_ms.mean() >= lcp_ms.mean() else (r_lcp, r_det)
    res = {
        "metric": "sim steps/sec (fwd+bwd), 1024 batched 3D scenes x 8 SDF bodies",
        "value": world * K / dt, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: floor + %d-box SDF stack with friction, %d scenes per GPU, %d steps fwd + reverse sweep"
                               % (args.nbox, B, K),
                   "scenes_per_gpu": B, "bodies": E.nb, "contacts_per_scene_mean": float(nc.mean()),
                   "contacts_per_scene_max": int(nc.max()), "attempts": att, "lcp_iters_mean": float(E.get("lcp_iters").mean()),
                   "substeps_mean": float((E.get("nsub") - E.be.to_numpy(lo)).mean()),
                   "scene_steps_per_s": world * B * K / dt, "capacity_overflow": overflow,
                   "parallelism": "scene-sharded x%d, no collective in step; one all_gather (%s) of final poses + per-scene "
                                  "gradients; %d physical device(s)" % (world, backend or "none", torch.cuda.device_count())},
        "roofline": dominant, "roofline_second_kernel": other,
    }
    if not args.no_cpu:
        threads = min(os.cpu_count() or 1, args.cpu_sample)
        per_scene, ns, nineq = cpu_baseline(E, args.cpu_sample, threads)
        # per_scene is wall/scene with `threads` scenes in flight; a 1024-scene step needs 1024 * per_scene seconds
        res["cpu_baseline"] = {"value": 1.0 / (B * per_scene), "unit": "steps/s", "cores": threads, "kind": "port",
                               "sample": "dense PDIPM LCP fwd+bwd only (oracle/lcp_oracle.c, the reference's algorithm) on the "
                                         "operands of %d median scenes of this batch (nineq=%d), OpenMP over scenes, scaled to %d "
                                         "scenes; contact detection not included, so this over-states the CPU" % (ns, nineq, B)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
