#!/usr/bin/env python3
"""Instructions of ONE psi + gradient evaluation by itself (the evaluation kernel, no solver around it), to split the solve
kernel's dynamic instruction count into evaluation and solver. Run under rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU ...:
   rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d <dir> -- python3 tools/pmc_eval_only.py
Evaluates the configs[2] contract family at the solver's final iterates (penalty and multipliers as returned), B = 16384."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
B = 16384
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg2_b65536_n20_4x10"]); lay = spec.pop("layout"); spec.pop("B")
fam = sys.argv[1] if len(sys.argv) > 1 else "toward_robot"
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec)
cfg = nm.default_config_struct()
cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]; cfg.axis_aligned = 1
with nm.Handle(cfg) as h:
    r = h.solve(P)
    c = r["info"][:, 3].astype(np.float32)
    for grad in (True, False):
        e = h.eval(P, r["U"].astype(np.float32), r["y"].astype(np.float32), c, grad=grad)
    print(fam, "evaluations:", B, "with gradient,", B, "without; mean psi", float(np.nanmean(e["psi"])))
