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
