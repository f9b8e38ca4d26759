#!/usr/bin/env python3
"""The noise-floor / first-divergence protocol of tests/accuracy_protocol.py on larger samples than the test-suite runs
(TEST INFRASTRUCTURE: imports the oracle).   usage: audit_large.py [n]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
import oracle
import accuracy_protocol as ap
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for wl in ("cfg1", "cfg2"):
    row = ap.run_case(nm, oracle, wl, "passing", n=n, nthreads=16, tight=False, audit=True, audit_max=10 ** 6)
    au = row["divergence_audit"]
    out = {"workload": wl, "family": "passing", "n": n,
           "hip64_vs_oracle64": row["hip64_vs_oracle64"], "hip64tp_vs_oracle64": row["hip64tp_vs_oracle64"],
           "oracle64_vs_reassociated": row["oracle64_vs_reassociated"],
           "audit_hip_vs_oracle": {k: v for k, v in au.items() if k not in ("pairs", "oracle_vs_reassociated")},
           "audit_oracle_vs_reassociated": {k: v for k, v in au["oracle_vs_reassociated"].items() if k != "pairs"},
           "kinds_hip": {k: sum(p.get("kind") == k for p in au["pairs"]) for k in sorted({p.get("kind") for p in au["pairs"]})},
           "unexplained_pairs": [p for p in au["pairs"] if not p["explained"]] + [p for p in au["oracle_vs_reassociated"]["pairs"] if not p["explained"]]}
    print(json.dumps(out), flush=True)
