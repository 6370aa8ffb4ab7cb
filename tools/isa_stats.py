#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels (reads hipcc -S --cuda-device-only output).

    python3 tools/isa_stats.py /tmp/isa/k_ps.s [kernel-substring]

The unit loops are fully unrolled, so the static count of the hot path is close to the
dynamic count per unit; it gives the issue-slot floor a kernel cannot beat
(wave64 VALU = 2 cycles on a SIMD-32, 4 when the wave is alone on its SIMD).
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if op.startswith(("v_mov", "v_cndmask", "v_accvgpr")):
            return "valu_mov"
        if re.match(r"v_(add|sub|mul|fma|mac|max|min|rcp|sqrt|rsq|div|cvt|trunc|floor|exp|log|frexp|ldexp)_?.*f(32|64)", op):
            return "valu_fp"
        return "valu_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = None
    stats = {}
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith((".", "\t")):
            cur = m.group(1)
            stats[cur] = Counter()
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None if line.startswith(".Lfunc_end") else cur
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        stats[cur][classify(op)] += 1
        stats[cur]["op:" + op] += 1
    for k, c in stats.items():
        if want not in k or not c:
            continue
        tot = sum(v for n, v in c.items() if not n.startswith("op:"))
        valu = sum(v for n, v in c.items() if n.startswith("valu"))
        print(f"== {k[:70]}  total {tot}  VALU {valu}")
        print("   " + "  ".join(f"{n}={v}" for n, v in sorted(c.items()) if not n.startswith("op:")))
        top = sorted(((v, n[3:]) for n, v in c.items() if n.startswith("op:")), reverse=True)[:22]
        print("   top: " + "  ".join(f"{n}:{v}" for v, n in top))


if __name__ == "__main__":
    main()
