#!/bin/bash
# Dev aid (GPU box): the tuning summary of the bench for the product library and every exp_libs/lib_*.so
cd "$GRAFT_REPO_ROOT"
echo "== product"; bash tools/bench_brief.sh "$@" || exit 1
for f in exp_libs/lib_*.so; do
  echo "== $f"; DSS_LIB_PATH=$PWD/$f bash tools/bench_brief.sh "$@" || exit 1
done
