#!/bin/bash
# Dev aid (GPU box): rocprofv3 evidence for the IGR MLP kernel: kernel stats + fp64-MFMA counters (separate pass).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/igr_stats gpurun_out/igr_pmc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/igr_stats -- python tools/bench_igr.py > gpurun_out/igr_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d gpurun_out/igr_pmc -- python tools/bench_igr.py > gpurun_out/igr_pmc.log 2>&1
python - <<PY
import csv, glob, collections, json
f = glob.glob("gpurun_out/igr_stats/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "igr_query" in r["Name"]]
out = {"kernel_stats": [{"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in rows]}
f = glob.glob("gpurun_out/igr_pmc/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "igr_query" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_avg_per_dispatch"] = {k: sum(v) / len(v) for k, v in acc.items()}
n, mac = 128 ** 3, 4 * (5 * 128 + 7 * 128 * 128 + 128)
avg_us = min(r["avg_us"] for r in out["kernel_stats"] if r["calls"] >= 5) if out["kernel_stats"] else None
out["points"] = n
out["algorithmic_tflops"] = 2 * mac * n / (avg_us * 1e-6) / 1e12 if avg_us else None
out["fp64_matrix_peak_tflops"] = 78.6
json.dump(out, open("gpurun_out/r1_igr_mfma.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -2 gpurun_out/igr_stats.log | head -1
