#!/bin/bash
# Dev aid (GPU box): config 4 (40 steps) under scratch-memory settings of the ROCm runtime (igr_advance_kernel uses 1.4 KB of scratch per lane)
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 400 python3 bench.py --config 4 --no-cpu --steps 40 --warmup 1 > gpurun_out/c4.json 2> gpurun_out/c4.err || { tail -3 gpurun_out/c4.err; return 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/c4.json').read().strip().splitlines()[-1])
print('steps/s %.2f attempts %d fwd %.3f s bwd %.3f s' % (r['value'], r['config']['attempts'], r['config']['forward_s'], r['config']['backward_s']))"; }
echo "== default"; run || exit 1
echo "== HSA_NO_SCRATCH_RECLAIM=1"; HSA_NO_SCRATCH_RECLAIM=1 run || exit 1
echo "== HSA_SCRATCH_SINGLE_LIMIT=4G"; HSA_SCRATCH_SINGLE_LIMIT=4294967296 run || exit 1
echo "== HSA_SCRATCH_SINGLE_LIMIT_ASYNC=8G"; HSA_SCRATCH_SINGLE_LIMIT_ASYNC=8589934592 run || exit 1
