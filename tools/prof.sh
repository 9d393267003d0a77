#!/bin/bash
# Dev aid (GPU box): rocprofv3 kernel stats of a short bench run -> gpurun_out/prof_last ; prints the top kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_last
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_last -- python bench.py --steps ${1:-20} --warmup 3 > gpurun_out/prof_last.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_last/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-62s %4s calls  avg %9.1f us  %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep "^{" gpurun_out/prof_last.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"ms_per_step\", d[\"ms_per_step\"])"
