#!/bin/bash
# needs a -DHEAAC_TUNING build (tools/build_variants.sh tuning "-DHEAAC_TUNING"; HEAAC_LIB_PATH=ab/libtuning.so)
# usage (via gpurun): tools/chunk_sweep.sh <tag> <lanes>:<chunk> ...  -- headline bench per (HEAAC_LANES, HEAAC_CHUNK_FRAMES)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for lc in "$@"; do
    l=${lc%%:*}; c=${lc##*:}
    HEAAC_LANES=$l HEAAC_CHUNK_FRAMES=$c python bench.py $SWEEP_ARGS --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes %d chunk %7d  frames/s %.3fM frac %.4f' % ($l, $c, d['value']/1e6, d['roofline']['frac']))"
done | tee gpurun_out/$tag.sweep.txt
