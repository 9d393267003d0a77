#!/bin/bash
# Dev aid (GPU box): everything profiles/ holds for this round, for the library that is in the tree.
#   part 1: counters + kernel stats of config 3 (tools/pmc.sh), bench lines of configs 3, 2, 1
#   part 2: kernel stats of configs 2 and 5, bench line of config 5, dense-LCP HBM table with counters
#   part 3: counters + kernel stats + bench line of config 4
# Usage: tools/collect_profiles.sh <part>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p profiles
bench() { timeout -k 10 900 python3 bench.py --config $1 > gpurun_out/r3_bench_config$1.json 2> gpurun_out/r3_bench_config$1.err || echo "config $1 failed"; cut -c1-260 gpurun_out/r3_bench_config$1.json; }
kstats() {   # config, steps, extra args
    local c=$1 s=$2; shift 2
    rm -rf gpurun_out/ks$c
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks$c -- python3 bench.py --config $c --steps $s --warmup 1 --no-cpu "$@" > gpurun_out/ks$c.log 2>&1
    cp gpurun_out/ks$c/*/*kernel_stats.csv gpurun_out/r3_kernel_stats_config$c.csv 2>/dev/null && rm -rf gpurun_out/ks$c
}
case "${1:-1}" in
1)  bash tools/pmc.sh 3 5 > gpurun_out/pmc3_run.log 2>&1; tail -1 gpurun_out/pmc3_run.log | cut -c1-200
    cp gpurun_out/r3_pmc_config3.json profiles/
    bench 3; bench 2; bench 1 ;;
2)  kstats 2 40; kstats 5 200 --batch 128
    bench 5
    timeout -k 10 600 python3 tools/bench_lcp_dense.py > gpurun_out/r3_lcp_dense_hbm.json 2> gpurun_out/r3_lcp_dense_hbm.err; tail -c 600 gpurun_out/r3_lcp_dense_hbm.json
    bash tools/pmc_lcp_dense.sh > gpurun_out/pmc_dense.log 2>&1; tail -c 300 gpurun_out/pmc_dense.log ;;
3)  bash tools/pmc.sh 4 40 > gpurun_out/pmc4_run.log 2>&1; tail -1 gpurun_out/pmc4_run.log | cut -c1-200
    cp gpurun_out/r3_pmc_config4.json profiles/
    bench 4 ;;
esac
