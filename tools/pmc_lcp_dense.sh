#!/bin/bash
# Dev aid (GPU box): counters of the dense-LCP kernels under tools/bench_lcp_dense.py --quick (B = 65536; (3,4,0) and (6,8,3)):
# instruction mix, VALU busy / wave cycles, FETCH_SIZE / WRITE_SIZE in separate passes (MI355X guide, HBM section).
#   -> gpurun_out/r3_lcp_dense_counters.json   (copy into profiles/ to have it judged)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_dense; rm -rf $OUT; mkdir -p $OUT
pass() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/bench_lcp_dense.py --quick > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
pass busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pass fetch FETCH_SIZE
pass write WRITE_SIZE
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python3 tools/bench_lcp_dense.py --quick > $OUT/kstats.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = collections.OrderedDict()
for d in ("insts", "busy", "f64", "fetch", "write"):
    fs = glob.glob("%s/%s/*/*_counter_collection.csv" % (out, d))
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "lcp_dense" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        e = res.setdefault(k, {})
        for c, v in cs.items():
            # dispatches alternate between the two sizes of --quick: report per size (even / odd dispatch index would need the
            # order; the sizes differ by > 2x in every counter, so split at the midpoint between min and max)
            lo = [x for x in v if x <= (min(v) + max(v)) / 2]; hi = [x for x in v if x > (min(v) + max(v)) / 2]
            e[c] = {"small(3,4,0)": sum(lo) / max(len(lo), 1), "large(6,8,3)": sum(hi) / max(len(hi), 1), "dispatches": len(v)}
fs = glob.glob("%s/kstats/*/*kernel_stats.csv" % out)
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "lcp_dense" in r["Name"]:
            res.setdefault(r["Name"].replace("(anonymous namespace)::", "").split("(")[0], {})["kernel_stats"] = {k: r[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs")}
json.dump(res, open("gpurun_out/r3_lcp_dense_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/*/
