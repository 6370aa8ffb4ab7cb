"""Where a kernel spills: scratch_load / scratch_store of one kernel of a device assembly listing, by source file and
20-line block (compile with  hipcc ... -gline-tables-only --cuda-device-only -S file.hip -o file.s).
usage: python3 tools/spill_sites.py file.s <kernel name prefix, mangled, e.g. _Z8k_hfps12>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l][0]
end = start + [i for i, l in enumerate(lines[start:]) if l.strip() == 's_endpgm'][0]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
cur, cnt, tot = None, collections.Counter(), collections.Counter()
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)) // 20 * 20)
        continue
    t = l.strip()
    if t.startswith('scratch_'):
        cnt[(cur, t.split()[0].replace('scratch_', '').replace('_dword', ''))] += 1
    if t and not t.startswith(('.', ';')) and cur:
        tot[cur] += 1
for (loc, op), c in sorted(cnt.items(), key=lambda kv: (kv[0][0] or ('', 0))):
    print('%-14s %5d+  %-10s %4d   (block: %d instructions)' % (loc[0], loc[1], op, c, tot[loc]))
print('instructions', sum(tot.values()), ' scratch ops', sum(cnt.values()))
