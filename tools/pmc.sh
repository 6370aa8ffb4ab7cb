#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc.sh <tag> "<counters>" <bench args...>
# one rocprofv3 --pmc pass (counters only, no trace domains) + per-wave summary
tag=$1; shift; ctr=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/$tag
rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/$tag -- python3 $root/bench.py "$@" --no-cpu-baseline > $root/gpurun_out/$tag.log 2>&1
python3 $root/tools/pmc_summary.py $root/gpurun_out/$tag
