"""The host parsers on damaged input, under AddressSanitizer + UBSan (tests/c/fuzz_parse.c built from the parser
sources): mutated, truncated, spliced and random access units must neither read out of bounds nor leave a record
that breaks validate.h, whatever status they end with."""
import json
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ffmpeg-heaac_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "c", "_build")
EXE = os.path.join(BUILD, "fuzz_parse")
KIND = {"lc_stereo_48k": 0, "hev1_stereo_24k": 1, "hev2_mono_24k": 2, "hev2_implicit_24k": 2, "lc_5_1_48k": 5,
        "lc_pce_5_1_coupled_48k": 6}


def test_parsers_survive_damaged_access_units():
    os.makedirs(BUILD, exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "c", "fuzz_parse.c")] + [os.path.join(CSRC, f) for f in
                                                                ("aac_parse.c", "sbr_parse.c", "sbr_header.c", "adts_split.c")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + srcs +
                              ["-o", EXE, "-lm", "-lpthread"])
    seeds = os.path.join(BUILD, "seeds.bin")
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "bitstreams.json")))
    import numpy as np
    import test_parse_wide as TW
    rng = np.random.default_rng(99)
    with open(seeds, "wb") as f:
        for name, s in sorted(v.items()):
            if name not in KIND:
                continue
            for au in s["access_units"]:
                b = bytes.fromhex(au)
                f.write(struct.pack("<II", KIND[name], len(b)) + b)
            if KIND[name] == 6:                            # the stream's configuration: its program config element
                b = bytes.fromhex(s["asc"])
                f.write(struct.pack("<II", 7, len(b)) + b)
        # access units with coupling channel elements and program config elements (the wide parser entry)
        for i in range(40):
            cpe = bool(i & 1)
            cces = [(int(t), [(1 if cpe else 0, 0, int(rng.integers(0, 4)) if cpe else 2)], int(rng.choice([0, 1, 3])),
                     bool(rng.integers(0, 2))) for t in rng.choice(16, int(rng.integers(1, 3)), replace=False)]
            b, _ = TW.build_au(rng, 3, 2, cpe, cces, pce=bool(i % 3 == 0))
            f.write(struct.pack("<II", 3 if cpe else 4, len(b)) + b)
        # 5.1 access units (the layout parser), some elements with payloads behind them
        import test_parse_layout as TL
        for i in range(24):
            b, _ = TL.build(rng, 3, 2, [(0, 0), (1, 0), (1, 1), (3 if i & 1 else 0, int(i & 2))], sbr_prob=0.5)
            f.write(struct.pack("<II", 5, len(b)) + b)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([EXE, seeds, "400000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-4000:]
    assert p.stdout.strip().endswith("ok"), p.stdout[-2000:]
    import re
    m = re.search(r"parsed (\d+), refused (\d+), frames with start = 1: (\d+)", p.stdout)
    # the run is not vacuous: many frames parse, many are refused, SBR streams keep (re)starting
    assert int(m.group(1)) > 60000 and int(m.group(2)) > 60000 and int(m.group(3)) > 10000, m.group(0)
    m = re.search(r"refused units that leave work for the spectral tools (\d+)", p.stdout)
    assert int(m.group(1)) > 1000, m.group(0)
    m = re.search(r"coupling elements parsed (\d+), ADTS frames delivered (\d+)", p.stdout)
    assert int(m.group(1)) > 5000 and int(m.group(2)) > 10000, m.group(0)
    m = re.search(r"5.1 units parsed (\d+), program config layouts accepted (\d+), out of range (\d+)", p.stdout)
    assert int(m.group(1)) > 3000 and int(m.group(2)) > 1000 and int(m.group(3)) == 0, m.group(0)
    m = re.search(r"coupled layout units parsed (\d+), gain lists landed (\d+)", p.stdout)
    assert int(m.group(1)) > 1000 and int(m.group(2)) > 1000, m.group(0)
