"""Merge gpurun_out/traffic_new.json (tools/traffic.sh, measured on the GPU box) into profiles/traffic.json and stamp
each merged entry with the git head it belongs to: an entry is accepted only if its kernel_sha equals the hash of the
device sources in the working tree (so the head named is one whose kernels were measured).
usage: python3 tools/traffic_stamp.py [round] [--same-kernels <commit>]
--same-kernels <commit>: the measurement was taken on the snapshot of <commit>, whose device CODE hashes like the working
tree's (checked) -- used in round 4 when the definition of the hash changed between measuring and stamping (comments no
longer count; later the spectral-tools file, which no bench workload launches, left the set).  The entry then names
<commit> as its head and the stamping head as `stamped_at`."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
sha = g.load_package().kernel_source_sha()
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
kfiles = g.load_package().kernel_source_files()
dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--"] + kfiles, text=True).strip()
new = json.load(open(os.path.join(ROOT, "gpurun_out", "traffic_new.json")))
path = os.path.join(ROOT, "profiles", "traffic.json")
allrec = json.load(open(path)) if os.path.exists(path) else {}
same = None
if "--same-kernels" in sys.argv:
    # the measurement's snapshot was <commit>: accept it if that commit's device CODE hashes like the working tree's
    same = sys.argv[sys.argv.index("--same-kernels") + 1]
    pkg = g.load_package()
    there = pkg.kernel_source_sha(read=lambda rel: subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (same, rel)], text=True))
    assert there == sha, "device code differs from %s: %s there, %s here" % (same, there, sha)
    sys.argv = [a for a in sys.argv if a not in ("--same-kernels", same)]
for key, rec in new.items():
    if same:
        rec["kernel_sha"] = sha
    if rec.get("kernel_sha") != sha:
        print("skip %s: measured on kernel sources %s, the tree has %s" % (key, rec.get("kernel_sha"), sha))
        continue
    if same:
        # measured on the snapshot of <commit>, whose kernels are today's: that is the head the entry names
        rec["head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", same], text=True).strip()
        rec["stamped_at"] = head
    else:
        rec["head"] = head + ("+uncommitted kernel changes" if dirty else "")
    if len(sys.argv) > 1:
        rec["round"] = int(sys.argv[1])
    allrec[key] = rec
    print("merged %s: %d bytes per frame at %s" % (key, rec["bytes_per_frame"], rec["head"]))
json.dump(allrec, open(path, "w"), indent=1)
