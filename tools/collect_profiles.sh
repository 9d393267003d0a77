#!/bin/bash
# Dev aid (GPU box): everything profiles/ holds for this round, for the library that is in the tree:
# counters + kernel stats of configs 3 and 4 (tools/pmc.sh), then the bench line of every config (with the counters just taken)
cd "$GRAFT_REPO_ROOT"
bash tools/pmc.sh 3 5 > gpurun_out/pmc3_run.log 2>&1; tail -2 gpurun_out/pmc3_run.log | cut -c1-200
bash tools/pmc.sh 4 40 > gpurun_out/pmc4_run.log 2>&1; tail -2 gpurun_out/pmc4_run.log | cut -c1-200
mkdir -p profiles
cp gpurun_out/r3_pmc_config3.json gpurun_out/r3_pmc_config4.json profiles/ 2>/dev/null
bash tools/bench_all.sh
