#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab.sh "<bench args>" ab/libA.so ab/libB.so ...
# runs tools/kprof.sh once per library variant on the SAME box (box-to-box clocks differ by ~5 %)
args=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
    cp $root/$lib $root/ffmpeg-heaac_amd/libheaac_amd.so
    echo "== $lib"
    $root/tools/kprof.sh ab_$(basename $lib .so) $args
done
