#!/bin/bash
# Dev aid (GPU box): kernel-time table of a short bench run.  Usage: tools/kstats.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --steps 20 --warmup 2 --no-cpu "$@" > gpurun_out/ks.log 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/ks/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print("%-70s calls %5s avg %9.1f us" % (r['Name'].replace("(anonymous namespace)::","")[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/ks
