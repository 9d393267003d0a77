#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for f in "" exp_libs/lib_*.so; do
  if [ -n "$f" ]; then export DSS_LIB_PATH=$PWD/$f; fi
  echo "== ${f:-product}"; bash tools/kstats.sh 2>&1 | grep "bwd_pre\|bwd_post"
done
