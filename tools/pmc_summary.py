"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: per-wave averages
(waves from the dispatch's grid size), optionally written as CSV for profiles/."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
waves = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not k.startswith('k_'): continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    d = r['Dispatch_Id']
    if d not in disp[k]:
        disp[k].add(d)
        waves[k] += float(r['Grid_Size']) / 64.0
rows = []
for k, v in agg.items():
    w = waves[k]
    print('%-24s launches %3d waves/launch %6.0f ' % (k, len(disp[k]), w / len(disp[k])) +
          ' '.join('%s=%.0f' % (c.replace('SQ_', ''), v[c] / w) for c in sorted(v)))
    rows.append((k, len(disp[k]), w / len(disp[k]), {c: v[c] / w for c in v}))
if len(sys.argv) > 2:
    cols = sorted({c for r in rows for c in r[3]})
    with open(sys.argv[2], 'w') as o:
        o.write('kernel,launches,waves_per_launch,' + ','.join(c + '_per_wave' for c in cols) + '\n')
        for k, n, w, d in rows:
            o.write('"%s",%d,%.0f,' % (k, n, w) + ','.join('%.1f' % d.get(c, 0.0) for c in cols) + '\n')
