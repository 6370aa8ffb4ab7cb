#!/bin/bash
# usage (on the GPU box): tools/traffic.sh <workload> <frames_per_launch> [f32|s16]   -> gpurun_out/traffic_new.json
# HBM bytes per frame of one workload: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (they do not fit
# one), per kernel and summed; the entry is stamped with the hash of the device sources it was measured on
# (pkg.kernel_source_sha()).  Back in the build container tools/traffic_stamp.py merges it into profiles/traffic.json
# and adds the git head.
wl=$1; fpl=$2; pcm=${3:-f32}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $root/gpurun_out/tr_${wl}_${pcm}_$c
    rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/tr_${wl}_${pcm}_$c -- python3 $root/bench.py --workload $wl --pcm $pcm --steps 3 --warmup 1 --no-cpu-baseline > $root/gpurun_out/tr_${wl}_${pcm}_$c.log 2>&1 || exit 1
done
python3 $root/tools/traffic.py $root/gpurun_out/tr_${wl}_${pcm}_FETCH_SIZE $root/gpurun_out/tr_${wl}_${pcm}_WRITE_SIZE $fpl ${wl}_${pcm} $root/gpurun_out/traffic_new.json
