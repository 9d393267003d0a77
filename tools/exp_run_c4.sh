#!/bin/bash
# Dev aid (GPU box): config 4 (40 steps) for the product library and every exp_libs/lib_*.so
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 400 python3 bench.py --config 4 --no-cpu --steps 40 --warmup 1 > gpurun_out/c4.json 2> gpurun_out/c4.err || { tail -3 gpurun_out/c4.err; return 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/c4.json').read().strip().splitlines()[-1])
print('steps/s %.2f attempts %d fwd %.3f s bwd %.3f s' % (r['value'], r['config']['attempts'], r['config']['forward_s'], r['config']['backward_s']))"; }
echo "== product"; run || exit 1
for f in exp_libs/lib_*.so; do echo "== $f"; DSS_LIB_PATH=$PWD/$f run || exit 1; done
