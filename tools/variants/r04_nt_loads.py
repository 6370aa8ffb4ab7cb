#!/usr/bin/env python3
"""Round 4, E6: streaming (non-temporal) loads for data a kernel reads once.
    python3 r04_nt_loads.py <copied csrc dir> <variant name>
variant name containing "syn": only the synthesis kernels' X rows and ring state (k_he.hip syn_ld4 / syn_ld1);
"all": additionally every GBuf load of k_common.h (cache policy 2 = nt on gfx950)."""
import re
import sys

d, name = sys.argv[1], sys.argv[2]
p = d + "/k_he.hip"
s = open(p).read()
s, n1 = re.subn(r"(syn_ld4\(const f32x4 \*p\) \{ return )\*p; \}", r"\1__builtin_nontemporal_load(p); }", s)
s, n2 = re.subn(r"(syn_ld1\(const float \*p\) \{ return )\*p; \}", r"\1__builtin_nontemporal_load(p); }", s)
assert n1 == 1 and n2 == 1
open(p, "w").write(s)
if "all" in name:
    p = d + "/k_common.h"
    s = open(p).read()
    s, n = re.subn(r"(raw_buffer_load_b(?:32|64)\((?:[^;]|\n)*?\(s \* 4\) & ~4095), 0\)", r"\1, 2)", s)
    assert n == 3, n
    open(p, "w").write(s)
