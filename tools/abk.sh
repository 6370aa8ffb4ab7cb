#!/bin/bash
# usage (via gpurun): tools/abk.sh <tag> ab/x.so ...   -- per-kernel averages of library variants on one box (no parity tests)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for lib in "$@"; do
    cp $root/$lib ffmpeg-heaac_amd/libheaac_amd.so
    echo "== $lib"
    tools/kprof.sh ${tag}_$(basename $lib .so) --steps 8 --warmup 2 | grep -v "^k_ps<true"
done
