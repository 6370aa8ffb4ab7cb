#!/bin/bash
# usage (via gpurun): tools/abw.sh <tag> "<bench args>" ab/x.so ...  -- per-kernel averages of library variants for any workload
tag=$1; shift; args=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for lib in "$@"; do
    cp $root/$lib ffmpeg-heaac_amd/libheaac_amd.so
    echo "== $lib"
    tools/kprof.sh ${tag}_$(basename $lib .so) $args --steps 8 --warmup 2 | grep -v "^k_ps<true\|copyBuffer"
    cd $root
done
