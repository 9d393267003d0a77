"""CPU: `python bench.py --gpus N` without torchrun becomes its own launcher (one child per rank, rendezvous on
127.0.0.1); checked here with the dry-run switch that stops after the process group's first all_gather."""
import json
import os
import subprocess
import sys

from helpers import ROOT


def test_bench_self_launch_two_ranks():
    env = dict(os.environ, DSS_DIST_BACKEND="gloo", DSS_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["ranks"] == [0, 1]


def test_bench_self_launch_eight_ranks_dry_run():
    """The 8-rank launch the driver makes on a whole node, rehearsed without GPUs: eight children, one rendezvous, rank order."""
    env = dict(os.environ, DSS_DIST_BACKEND="gloo", DSS_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["ranks"] == list(range(8))


def test_launcher_parent_does_not_import_torch_and_reaps_a_failed_rank():
    """The launcher parent must not touch HIP (it spawns ranks): it does not even import torch.  And when a rank dies the others
    are terminated instead of waiting out the process-group timeout."""
    import time
    code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '3']; sys.path.insert(0, %r); import bench\n"
            "os.environ['DSS_BENCH_DRYRUN'] = '1'; os.environ['DSS_DIST_BACKEND'] = 'gloo'; os.environ['DSS_BENCH_FAIL_RANK'] = '1'\n"
            "rc = bench.self_launch(3)\n"
            "assert 'torch' not in sys.modules, 'launcher parent imported torch'\n"
            "print('RC', rc)" % ROOT)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RC" in out.stdout and int(out.stdout.split("RC")[-1]) != 0
    assert time.time() - t0 < 120
