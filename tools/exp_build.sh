#!/bin/bash
# Dev aid: build variants of ONE translation unit with extra -D flags into exp_libs/lib_<tag>.so (the other objects are
# the product's).  Usage: tools/exp_build.sh <file.hip> <tag> <flags...>
set -e
cd "$(dirname "$0")/.."
src=$1; tag=$2; shift 2
base=$(basename $src .hip)
mkdir -p exp_libs/obj
lsr="-mllvm -disable-lsr"; [ "$base" = lcp_contact ] && lsr="-mllvm -amdgpu-load-store-vectorizer=0"      # as diffsdfsim_amd/_lib.py: file_flags
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -cuid=dss_$base $lsr "$@" -c -o exp_libs/obj/${base}_$tag.o diffsdfsim_amd/csrc/$src
objs=$(ls diffsdfsim_amd/csrc/_obj/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_libs/lib_$tag.so $objs exp_libs/obj/${base}_$tag.o
echo built exp_libs/lib_$tag.so
