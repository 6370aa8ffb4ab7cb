#!/usr/bin/env python3
"""Write tests/golden/table_sha256.json from the product's host-built tables."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from test_tables import TABLES
pkg = g.build()
out = {n: hashlib.sha256(pkg.get_table(n).tobytes()).hexdigest() for n in TABLES}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "table_sha256.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out))
