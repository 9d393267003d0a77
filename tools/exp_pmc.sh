#!/bin/bash
# Dev aid (GPU box): VALU / SALU / LDS / VMEM instruction counts of the narrow phase cut off after each stage
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/exp_pmc; rm -rf $OUT; mkdir -p $OUT
run() {  # tag, lib
  if [ -n "$2" ]; then export DSS_LIB_PATH=$PWD/$2; else unset DSS_LIB_PATH; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/$1 -- python3 tools/exp_np_time.py > $OUT/$1.log 2>&1 || echo "fail $1"
}
run full ""
for st in 1 2 3 4 5; do run stop$st exp_libs/lib_stop${st}w3.so; done
python3 - <<'PY'
import csv, glob, collections
for tag in ["stop1","stop2","stop3","stop4","stop5","full"]:
    fs = glob.glob("gpurun_out/exp_pmc/%s/*/*_counter_collection.csv" % tag)
    if not fs: print(tag, "no data"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "narrowphase_kernel<false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())})
PY
rm -rf $OUT/*/
