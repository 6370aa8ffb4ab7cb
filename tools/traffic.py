"""HBM traffic per frame from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit in
one pass).  usage: traffic.py <fetch_dir> <write_dir> <frames_per_launch> <key> [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950
tallies 128-byte read requests at 64 bytes).  Per kernel and summed over the kernels of one step."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

def per_kernel(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('k_'):
            continue
        tot[k] += float(r['Counter_Value']); cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}

fetch = per_kernel(sys.argv[1], 'FETCH_SIZE'); write = per_kernel(sys.argv[2], 'WRITE_SIZE')
frames = float(sys.argv[3]); key = sys.argv[4]
rec = {"kernels": {}, "note": "KiB per frame; fetch = 2 x FETCH_SIZE (gfx950 correction), %d frames per launch" % frames}
tf = tw = 0.0
for k in sorted(set(fetch) | set(write)):
    f = 2 * fetch.get(k, 0.0) / frames; w = write.get(k, 0.0) / frames
    rec["kernels"][k] = {"fetch_KiB": round(f, 2), "write_KiB": round(w, 2)}
    tf += f; tw += w
rec["fetch_KiB_per_frame"] = round(tf, 2); rec["write_KiB_per_frame"] = round(tw, 2)
rec["bytes_per_frame"] = int((tf + tw) * 1024)
rec["kernel_sha"] = g.load_package().kernel_source_sha()
print(json.dumps(rec, indent=1))
if len(sys.argv) > 5:
    try:
        allrec = json.load(open(sys.argv[5]))
    except Exception:
        allrec = {}
    allrec[key] = rec
    json.dump(allrec, open(sys.argv[5], 'w'), indent=1)
