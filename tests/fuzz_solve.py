#!/usr/bin/env python3
"""Randomised differential check of short SOLVES (by hand on the GPU box for long runs; a reduced run with fixed seeds is
part of the -m gpu suite through tests/test_gpu_fuzz.py): fp64, one outer
x four inner iterations, every solver kernel (throughput, latency with 2..4 wavefronts, cooperative with 2..4, automatic)
against the sequential oracle on the cases of tests/fuzz_eval.py -- iteration counts, exit status and controls.
    python tests/fuzz_solve.py [cases] [seed] [outer] [inner]
With more iterations (e.g. 3 x 15: penalty / multiplier updates, L-BFGS ring wrap-around) rounding differences are amplified
along the path; instances whose iteration counts differ from the oracle's are counted, not compared."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dyobav_mpcnwta_warehouse_amd as nm   # noqa: E402
import oracle                               # noqa: E402
from fuzz_eval import make_case             # noqa: E402


def run(cases=100, seed=0, n_outer=1, n_inner=4, out=print):
    rng = np.random.default_rng(seed)
    short = n_outer * n_inner <= 4   # longer runs: rounding differences grow ~4x per iteration on these instances
    worst, flips, total, dus = {}, {}, 0, {}
    for ci in range(cases):
        lay, rows, P, _, _, _ = make_case(rng)
        pr = oracle.Problem(lay.N, lay.Nother, lay.Nstc, lay.Ndyn)
        op = oracle.Options(max_outer=n_outer, max_inner=n_inner, lip_delta=1e-4, lip_eps=1e-4)
        Uo, ro = oracle.solve_batch(pr, op, P, nthreads=4)
        modes = [("throughput", dict(latency_waves=1, coop_waves=1)), ("throughput/reg64", dict(latency_waves=1, coop_waves=1, reg_table=1)), ("latency%d" % (2 + ci % 3), dict(latency_waves=2 + ci % 3, coop_waves=1)),
                 ("coop%d" % (2 + ci % 3), dict(latency_waves=1, coop_waves=2 + ci % 3, reg_table=-1)), ("automatic", dict())]
        for name, ov in modes:
            cfg = nm.default_config_struct()
            cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
            cfg.max_outer_iterations, cfg.max_inner_iterations = n_outer, n_inner
            cfg.lip_eps_f64 = cfg.lip_delta_f64 = 1e-4
            for k, v in ov.items():
                setattr(cfg, k, v)
            with nm.Handle(cfg) as h:
                r = h.solve(P, dtype=np.float64)
            key = name.rstrip("234") if "reg64" not in name else name
            for i in range(P.shape[0]):
                total += 1
                if r["iters"][i, 1] != ro["inner_iters"][i]:
                    flips[key] = flips.get(key, 0) + 1    # a line-search / exit decision flipped by rounding
                    continue
                du = np.abs(r["U"][i] - Uo[i]).max()
                worst[key] = max(worst.get(key, 0.0), du)
                dus.setdefault(key, []).append(du)
                if not np.isfinite(r["U"][i]).all() or (short and not du < 1e-5):
                    out(f"MISMATCH case {ci} N={lay.N} Nother={lay.Nother} Nstc={lay.Nstc} Ndyn={lay.Ndyn} rows={rows} mode={name} "
                          f"instance {i}: max|du| = {du:.3e}")
                    return 1
    out(f"{cases} cases, {total} solves checked against the oracle (fp64, {n_outer} x {n_inner} iterations)")
    for k in sorted(worst):
        out(f"  {k:12s} max|u - u_oracle|: median {np.median(dus[k]):.2e}, worst {worst[k]:.2e}   iteration-count differences {flips.get(k, 0)}")
    return 0


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    sys.exit(run(*a))
