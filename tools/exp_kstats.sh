#!/bin/bash
# Dev aid (GPU box): average duration of the kernels matching $1 under the product library and every exp_libs/lib_*.so (config 3, 30 steps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pat=$1
run() { rm -rf gpurun_out/ks_exp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_exp -- python3 bench.py --config 3 --steps 30 --warmup 2 --no-cpu > gpurun_out/ks_exp.log 2>&1
  python3 - "$pat" <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob("gpurun_out/ks_exp/*/*kernel_stats.csv")[0])):
    if sys.argv[1] in r["Name"]: print("   %-50s calls %s avg %.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/ks_exp; }
echo "== product"; run
for f in exp_libs/lib_*.so; do echo "== $f"; DSS_LIB_PATH=$PWD/$f run; done
