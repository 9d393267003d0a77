#!/bin/bash
# Dev aid (GPU box): run a python tool for the product library and every exp_libs/lib_*.so.  Usage: tools/exp_run_py.sh <script>
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 "$1" 2>&1 | grep -v Warn | tail -3 || exit 1
for f in exp_libs/lib_*.so; do
  DSS_LIB_PATH=$PWD/$f timeout -k 10 200 python3 "$1" 2>&1 | grep -v Warn | tail -3 || exit 1
done
