#!/bin/bash
# usage (on the GPU box): tools/traffic.sh <workload> <frames_per_launch>   -> gpurun_out/traffic_<workload>.json
wl=$1; fpl=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $root/gpurun_out/tr_${wl}_$c
    rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/tr_${wl}_$c -- python3 $root/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $root/gpurun_out/tr_${wl}_$c.log 2>&1 || exit 1
done
python3 $root/tools/traffic.py $root/gpurun_out/tr_${wl}_FETCH_SIZE $root/gpurun_out/tr_${wl}_WRITE_SIZE $fpl ${wl}_f32 $root/gpurun_out/traffic_new.json
