#!/bin/bash
# Dev aid (GPU box): HBM-side traffic of the bench kernels from rocprofv3 PMC counters, FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (MI355X guide, HBM section), plus a kernel-stats pass of the same command.
#   -> gpurun_out/r1_pmc_traffic.json, gpurun_out/r1_kernel_stats.csv   (copy into profiles/ to have them judged)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=${1:-5}
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/kstats
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps $STEPS --warmup 1 > gpurun_out/pmc_f.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps $STEPS --warmup 1 > gpurun_out/pmc_w.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python bench.py --steps 20 --warmup 3 > gpurun_out/kstats.log 2>&1
python - <<PY
import csv, glob, json, collections, shutil
out = collections.OrderedDict()
for tag, d in (("FETCH_SIZE", "pmc_f"), ("WRITE_SIZE", "pmc_w")):
    f = glob.glob("gpurun_out/%s/*/*_counter_collection.csv" % d)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == tag:
            acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "anonymous" not in k: continue
        e = out.setdefault(k, {})
        e[tag + "_KB_avg"] = sum(v) / len(v)
        e["dispatches"] = len(v)
    shutil.copy(f, "gpurun_out/r1_pmc_%s_sample.csv" % ("fetch" if tag == "FETCH_SIZE" else "write"))
json.dump(out, open("gpurun_out/r1_pmc_traffic.json", "w"), indent=1)
shutil.copy(glob.glob("gpurun_out/kstats/*/*kernel_stats.csv")[0], "gpurun_out/r1_kernel_stats.csv")
for k, v in out.items(): print(k, v)
PY
tail -1 gpurun_out/kstats.log | cut -c1-400
