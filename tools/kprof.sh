#!/bin/bash
# usage (on the GPU box, via gpurun): tools/kprof.sh <tag> <bench args...>
# kernel-trace + stats of one bench run; prints per-kernel averages.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- python3 $root/bench.py "$@" --no-cpu-baseline > $root/gpurun_out/$tag.log 2>&1
python3 - "$root/gpurun_out/$tag" "$root/gpurun_out/$tag.log" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:7]:
    if r['Name'].startswith(('void at::', 'at::')): continue
    print('%-44s calls %5s avg_us %8.1f  %5s%%' % (r['Name'].split('(')[0].replace('void ', '')[:44], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
m = re.search(r'"value": ([0-9.]+)', open(sys.argv[2]).read())
print('frames/s', m.group(1) if m else 'n/a')
PY
