#!/bin/bash
# usage (via gpurun): tools/ab_s16.sh <tag> ab/x.so ...  -- int16 parity tests with the product library, then per-kernel
# averages of `bench.py --pcm s16` for each variant and the product on the same box
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_he_gpu.py tests/test_golden.py tests/test_shim_gpu.py -m gpu -x -q -k "s16 or S16 or golden or codec" > gpurun_out/$tag.tests.log 2>&1
tail -1 gpurun_out/$tag.tests.log
cp ffmpeg-heaac_amd/libheaac_amd.so /tmp/lib_keep.so
for lib in "$@" /tmp/lib_keep.so; do
    [ "$lib" = /tmp/lib_keep.so ] || cp $root/$lib ffmpeg-heaac_amd/libheaac_amd.so
    [ "$lib" = /tmp/lib_keep.so ] && cp /tmp/lib_keep.so ffmpeg-heaac_amd/libheaac_amd.so
    echo "== $lib"
    tools/kprof.sh ${tag}_$(basename $lib .so) --pcm s16 --steps 8 --warmup 2 | grep -v "^k_ps<true"
    cd $root
done
