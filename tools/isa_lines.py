#!/usr/bin/env python3
"""Static instruction count of one kernel per source phase (hipcc -S -gline-tables-only output).

    python3 tools/isa_lines.py /tmp/isa/k_ps_g.s k_hfps [bucket]

Each instruction is charged to the innermost .loc line in one of this repo's files; lines are
grouped into buckets of `bucket` lines (default 20) per file.  The unit loops are fully unrolled,
so static counts are close to dynamic counts per unit.
"""
import collections, re, sys
path, kern = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 20
files = {}
cur = None
inside = False
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
    if m: files[int(m.group(1))] = m.group(2); continue
    if re.match(r'^_Z\w*%s\w*:' % kern, ln): inside = True; continue
    if inside and s.startswith('.Lfunc_end'): inside = False; continue
    if not inside: continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        f = files.get(int(m.group(1)), '?')
        if f.startswith('k_') : cur = (f, int(m.group(2)))
        continue
    if not s or s.startswith(('.', ';', '//')) or s.endswith(':'): continue
    op = s.split()[0]
    k = ('pk' if op.startswith('v_pk') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
         'vmem' if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')) else 'wait' if op.startswith('s_waitcnt') else
         'salu' if op.startswith('s_') else 'other')
    key = (cur[0], cur[1] // bucket * bucket) if cur else ('?', 0)
    cnt[key] += 1; kinds[key][k] += 1
tot = sum(cnt.values())
print('total', tot)
for key in sorted(cnt):
    print('%-10s %4d-%4d  %6d  %4.1f%%  ' % (key[0], key[1], key[1] + bucket - 1, cnt[key], 100.0 * cnt[key] / tot) +
          ' '.join('%s=%d' % kv for kv in sorted(kinds[key].items())))
allk = collections.Counter()
for k in kinds.values(): allk.update(k)
print(dict(allk))
