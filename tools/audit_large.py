#!/usr/bin/env python3
"""The noise-floor / first-divergence / tight-tolerance KKT protocol of tests/accuracy_protocol.py on larger samples than
the test-suite runs (TEST INFRASTRUCTURE: imports the oracle).
   usage: audit_large.py [n] [n_tight] [families: passing closed_loop ...]
   `passing`: configs[1] and configs[2] generators; `closed_loop` / `refscen`: parameter vectors harvested from the closed loop
   at configs[2]'s dimensions (scenarios.harvest_closed_loop: the corridor family / the reference's scenario_0..2 on its
   warehouse map). Every far pair is audited (no truncation); TIGHT_AUDIT=1: the tolerance-1e-8 pairs too."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
import oracle
import accuracy_protocol as ap
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_tight = int(sys.argv[2]) if len(sys.argv) > 2 else 0
families = sys.argv[3:] or ["passing"]


def digest(row, wl, fam):
    au = row["divergence_audit"]
    out = {"workload": wl, "family": fam, "n": row["n"], "converged_frac": row["converged_frac"],
           "hip64_vs_oracle64": row["hip64_vs_oracle64"], "hip64tp_vs_oracle64": row["hip64tp_vs_oracle64"],
           "oracle64_vs_reassociated": row["oracle64_vs_reassociated"],
           "audit_hip_vs_oracle": {k: v for k, v in au.items() if k not in ("pairs", "oracle_vs_reassociated")},
           "audit_oracle_vs_reassociated": {k: v for k, v in au["oracle_vs_reassociated"].items() if k != "pairs"},
           "kinds_hip": {k: sum(p.get("kind") == k for p in au["pairs"]) for k in sorted({p.get("kind") for p in au["pairs"]})},
           "unexplained_pairs": [p for p in au["pairs"] if not p["explained"]] + [p for p in au["oracle_vs_reassociated"]["pairs"] if not p["explained"]]}
    if "tight_kkt_hip64_vs_oracle64" in row:
        out["n_tight"] = row["n_tight"]
        for k in ("hip64_vs_oracle64_tight", "oracle64_vs_reassociated_tight", "tight_kkt_hip64_vs_oracle64", "tight_kkt_oracle64_vs_reassociated"):
            out[k] = row[k]
    return out


for fam in families:
    if fam in ("closed_loop", "refscen"):
        lay = nm.scenarios.BENCH_CONFIGS["cfg2_b65536_n20_4x10"]["layout"]
        cfg = nm.default_config_struct()
        cfg.Ndynobs, cfg.max_active_dynobs = lay.Ndyn, 40
        steps, hfam = ((1, 8, 20), "corridor") if fam == "closed_loop" else ((2, 14, 26), "reference")
        P, _ = nm.scenarios.harvest_closed_loop(cfg, max(3 * n // 2, 96), steps=steps, seed=13, n_ped=4, n_hyp=10, dtype=np.float32, family=hfam)
        tight_audit = bool(os.environ.get("TIGHT_AUDIT"))
        row = ap.run_case_on(nm, oracle, P[:n].astype(np.float64), lay, 40, "cfg2", fam, nthreads=16, tight=n_tight > 0,
                             audit=True, audit_max=10 ** 6, tight_audit=tight_audit, n_tight=n_tight or None, polish=False)
        d = digest(row, "cfg2", fam)
        if tight_audit and "divergence_audit_tight" in row:
            ta = row["divergence_audit_tight"]
            d["tight_audit_hip_vs_oracle"] = {k: v for k, v in ta.items() if k not in ("pairs", "oracle_vs_reassociated")}
            d["tight_audit_oracle_vs_reassociated"] = {k: v for k, v in ta["oracle_vs_reassociated"].items() if k != "pairs"}
            d["tight_unexplained_pairs"] = [p for p in ta["pairs"] if not p["explained"]]
        print(json.dumps(d), flush=True)
        continue
    for wl in ("cfg1", "cfg2"):
        row = ap.run_case(nm, oracle, wl, fam, n=n, nthreads=16, tight=n_tight > 0, audit=True, audit_max=10 ** 6, tight_audit=False,
                          n_tight=n_tight or None)
        print(json.dumps(digest(row, wl, fam)), flush=True)
