#!/bin/bash
# One measurement round on the GPU box: HE parity tests with the product library, then per-kernel
# averages (rocprofv3 --kernel-trace --stats) for each library variant on the SAME box, then phase stamps.
# usage (via gpurun): tools/iter.sh <tag> [ab/variant.so ...]     (the product library is always measured)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python -m pytest tests/test_he_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/$tag.tests.log 2>&1
tail -1 gpurun_out/$tag.tests.log
cp ffmpeg-heaac_amd/libheaac_amd.so /tmp/lib_keep.so
for lib in "$@" /tmp/lib_keep.so; do
    [ "$lib" = /tmp/lib_keep.so ] || cp $root/$lib ffmpeg-heaac_amd/libheaac_amd.so
    [ "$lib" = /tmp/lib_keep.so ] && cp /tmp/lib_keep.so ffmpeg-heaac_amd/libheaac_amd.so
    echo "== $lib"
    tools/kprof.sh ${tag}_$(basename $lib .so) --steps 8 --warmup 2 | grep -v "^k_ps<true"
done
if [ -f ab/libstamps.so ]; then
  cp ab/libstamps.so ffmpeg-heaac_amd/libheaac_amd.so
  N=65536 python3 tools/hfps_stamps.py 2>/dev/null | tee gpurun_out/$tag.stamps.txt | tr '\n' ';'
  cp /tmp/lib_keep.so ffmpeg-heaac_amd/libheaac_amd.so
  echo
fi
