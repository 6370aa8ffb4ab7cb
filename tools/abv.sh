#!/bin/bash
# usage (via gpurun): tools/abv.sh <tag> "<bench args>" ab/x.so ... [product]
# Per-kernel averages (rocprofv3 --kernel-trace --stats) of library variants on ONE box (boxes differ by
# ~5 % in clocks).  A variant is selected through HEAAC_LIB_PATH, which the Python binding honours: the
# product library is never overwritten.  `product` = ffmpeg-heaac_amd/libheaac_amd.so.
tag=$1; shift; args=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for lib in "$@"; do
    if [ "$lib" = product ]; then unset HEAAC_LIB_PATH; name=product
    else export HEAAC_LIB_PATH=$root/$lib; name=$(basename $lib .so); fi
    echo "== $lib"
    tools/kprof.sh ${tag}_$name $args --steps 8 --warmup 2 | grep -v "^k_ps<true\|copyBuffer\|^k_check"
    cd $root
done
unset HEAAC_LIB_PATH
