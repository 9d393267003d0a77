#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 python3 tools/exp_np_time.py 2>&1 | tail -1 || exit 1
for f in exp_libs/lib_*.so; do
  w=$(echo $f | sed 's/.*w\([0-9]*\)\.so/\1/')
  DSS_NP_WAVES=$w DSS_LIB_PATH=$PWD/$f timeout -k 10 120 python3 tools/exp_np_time.py 2>&1 | tail -1 || exit 1
done
