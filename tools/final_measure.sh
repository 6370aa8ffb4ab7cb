#!/bin/bash
# The round's closing measurement on ONE box: the whole GPU suite, one bench line per workload, per-kernel
# averages (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and
# three PMC passes of the headline workload.  Everything lands in gpurun_out/<tag>.*; copy what is to be
# judged into profiles/.
# usage (via gpurun): tools/final_measure.sh <tag> a|b     (two calls: a box call is limited to 20 minutes)
#   a: GPU suite, bench lines, per-kernel averages        b: HBM traffic and the PMC passes
tag=${1:-final}
part=${2:-a}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
if [ $part = a ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/$tag.tests.log 2>&1 || { tail -30 gpurun_out/$tag.tests.log; exit 1; }
tail -1 gpurun_out/$tag.tests.log
: > gpurun_out/$tag.bench.jsonl
python bench.py --steps 10 --warmup 3 | tail -1 >> gpurun_out/$tag.bench.jsonl || exit 1
for wl in hev1 lc_stereo hev2_34; do
    python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline | tail -1 >> gpurun_out/$tag.bench.jsonl || exit 1
done
python bench.py --pcm s16 --steps 10 --warmup 3 --no-cpu-baseline | tail -1 >> gpurun_out/$tag.bench.jsonl || exit 1
cut -c1-400 gpurun_out/$tag.bench.jsonl
for wl in hev2 hev1 lc_stereo hev2_34; do
    echo "== $wl"
    tools/kprof.sh ${tag}_k_$wl --workload $wl --steps 8 --warmup 2 || exit 1
done
exit 0
fi
cd $root
rm -f gpurun_out/traffic_new.json
tools/traffic.sh hev2 262144 f32 || exit 1
cd $root; tools/traffic.sh hev2 262144 s16 || exit 1
cd $root; tools/traffic.sh hev1 65536 f32 || exit 1
cd $root; tools/traffic.sh hev2_34 65536 f32 || exit 1
cd $root; tools/traffic.sh lc_stereo 65536 f32 || exit 1
cd $root
tools/pmc.sh ${tag}_pmc_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH" --steps 3 --warmup 1 || exit 1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_a gpurun_out/${tag}_pmc_a.csv > /dev/null
tools/pmc.sh ${tag}_pmc_b "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" --steps 3 --warmup 1 || exit 1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_b gpurun_out/${tag}_pmc_b.csv > /dev/null
tools/pmc.sh ${tag}_pmc_c "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" --steps 3 --warmup 1 || exit 1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_c gpurun_out/${tag}_pmc_c.csv > /dev/null
# the same first pass for HE-AACv1 (k_hfadj instead of the fused kernel)
tools/pmc.sh ${tag}_pmc_hev1_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH" --workload hev1 --steps 3 --warmup 1 || exit 1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_hev1_a gpurun_out/${tag}_pmc_hev1_a.csv > /dev/null
echo done
