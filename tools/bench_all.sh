#!/bin/bash
# Dev aid (GPU box): the bench line of every BASELINE config -> gpurun_out/r3_bench_config<N>.json
cd "$GRAFT_REPO_ROOT"
for c in 3 2 5 4 1; do
    timeout -k 10 500 python3 bench.py --config $c > gpurun_out/r3_bench_config$c.json 2> gpurun_out/r3_bench_config$c.err || echo "config $c failed"
    cut -c1-400 gpurun_out/r3_bench_config$c.json
done
