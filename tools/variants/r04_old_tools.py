#!/usr/bin/env python3
"""Round 4: the spectral-tools kernel as it was before E8 (k_tools.hip of commit 3ae548d), for A/B and bisecting.
    VARIANT_EDIT=tools/variants/r04_old_tools.py tools/build_variants.sh toolsold ""   (needs git)"""
import subprocess, sys, os
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = subprocess.check_output(["git", "-C", root, "show", "3ae548d:ffmpeg-heaac_amd/csrc/k_tools.hip"], text=True)
open(os.path.join(d, "k_tools.hip"), "w").write(src)
