#!/bin/bash
# Dev aid (GPU box): the two smallest rows of tools/bench_lcp_dense.py for the product library and every exp_libs/lib_*.so
cd "$GRAFT_REPO_ROOT"
run() { python3 tools/bench_lcp_dense.py --quick 2>&1 >/dev/null | grep "'B'" | sed -E "s/.*'nineq': ([0-9]+).*'ms': ([0-9.]+).*'GBps': ([0-9.]+).*/nineq \1  ms \2  GB\/s \3/"; }
echo "== product"; run
for f in exp_libs/lib_*.so; do echo "== $f"; DSS_LIB_PATH=$PWD/$f run; done
