"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (per-wave averages)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not k.startswith('k_'): continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
names = ['SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INST_CYCLES_VMEM', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_LDS', 'SQ_INSTS_WAVE32_LDS']
names_old = ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES']
for k, v in agg.items():
    w = v['SQ_WAVES']
    print('%-28s launches %3d waves/launch %6.0f' % (k, cnt[k], w / cnt[k]), ' '.join('%s=%.0f' % (c.replace('SQ_', ''), v[c] / w) for c in names if c in v))
