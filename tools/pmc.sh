#!/bin/bash
# Dev aid (GPU box): per-kernel counters of the bench command from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes, MI355X guide, HBM section; fp64 instruction counts; VALU / MFMA busy cycles), plus a kernel-stats pass of
# the same command.  Usage: tools/pmc.sh <config> <steps> [batch]
#   -> gpurun_out/r3_pmc_config<N>.json (stamped with the library's hash: bench.py reports its numbers only for that build)
#      gpurun_out/r3_kernel_stats_config<N>.csv        (copy both into profiles/ to have them judged)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-3}
STEPS=${2:-5}
BATCH=${3:+--batch $3}
OUT=gpurun_out/pmc_c$CFG
rm -rf $OUT && mkdir -p $OUT
pass() {   # name, counters...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu $BATCH > $OUT/$name.log 2>&1 || echo "pass $name failed (see $OUT/$name.log)"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pass busy SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python3 bench.py --config $CFG --steps $((STEPS * 2)) --warmup 2 --no-cpu $BATCH > $OUT/kstats.log 2>&1
python3 - "$CFG" "$OUT" <<'PY'
import collections, csv, glob, hashlib, json, shutil, sys
cfg, out_dir = sys.argv[1], sys.argv[2]
kern = collections.OrderedDict()
def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
for d in ("fetch", "write", "f64", "busy", "mfma"):
    fs = glob.glob("%s/%s/*/*_counter_collection.csv" % (out_dir, d))
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        e = kern.setdefault(k, {})
        for c, v in cs.items():
            e[("%s_KB_avg" if c in ("FETCH_SIZE", "WRITE_SIZE") else "%s_avg") % c] = sum(v) / len(v)
            e["dispatches"] = len(v)
for k, e in kern.items():
    # SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD summed over the chip; SQ_BUSY_CYCLES per shader engine: report the plain
    # ratio the round-1 profiles used (VALU-issue cycles / wave-resident cycles) as an indication, not an absolute
    if "SQ_ACTIVE_INST_VALU_avg" in e and e.get("SQ_WAVE_CYCLES_avg"):
        e["valu_busy_frac"] = e["SQ_ACTIVE_INST_VALU_avg"] / e["SQ_WAVE_CYCLES_avg"]       # per resident wave
    if "SQ_ACTIVE_INST_VALU_avg" in e and e.get("GRBM_GUI_ACTIVE_avg"):
        # per SIMD: quad-cycles of VALU issue summed over all waves x 4 / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs
        e["valu_util_per_simd"] = e["SQ_ACTIVE_INST_VALU_avg"] * 4.0 / (e["GRBM_GUI_ACTIVE_avg"] / 8.0 * 1024.0)
    if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in e and e.get("GRBM_GUI_ACTIVE_avg"):
        # busy cycles are summed over the 1024 SIMDs of the chip (MI355X guide): fraction of the kernel's cycles the matrix
        # pipes were busy, averaged over SIMDs
        # (GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = that / 8)
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (e["GRBM_GUI_ACTIVE_avg"] / 8.0 * 1024.0)
lib = "diffsdfsim_amd/csrc/libdiffsdfsim_hip.so"
res = {"config": int(cfg), "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16], "kernels": kern}
json.dump(res, open("gpurun_out/r3_pmc_config%s.json" % cfg, "w"), indent=1)
ks = glob.glob("%s/kstats/*/*kernel_stats.csv" % out_dir)
if ks:
    shutil.copy(ks[0], "gpurun_out/r3_kernel_stats_config%s.csv" % cfg)
for k, v in kern.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
PY
tail -1 $OUT/kstats.log | cut -c1-600
# the raw per-dispatch tables are large (gpurun merges at most 64 MiB back): keep the logs and the summaries only
for d in fetch write f64 busy mfma kstats; do rm -rf $OUT/$d; done
