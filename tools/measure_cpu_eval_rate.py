#!/usr/bin/env python3
"""The constant behind `max_solver_time` -> `max_evaluations` (solver.evaluation_budget, INTEGRATION.md "max_solver_time").

The reference's solver stops after `max_solver_time` micro-seconds of ITS wall clock (mpc_builder.py:189), i.e. after as
many psi / grad-psi evaluations as one host core gets through in that time. This script measures that rate with the CPU
oracle -- the generated-code-equivalent path: fp64, cos / sin of every ellipse on every evaluation, cost and gradient as
separate calls, like the CasADi-generated C that OpEn links -- on ONE thread, for several dimension sets, in
points per second (a point = one argument at which psi is formed: orc_result.n_points = the kernels' info[4]) and in
"forward flops" per second (points x F_fwd(dims)), which is what solver.evaluation_budget scales by.

    python tools/measure_cpu_eval_rate.py [instances per dimension set]

Test infrastructure (imports oracle/); bench.py's cpu_baseline leg repeats the measurement on the GPU box's host
(`cpu_baseline.evals_per_s_per_core`).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dyobav_mpcnwta_warehouse_amd as nm   # noqa: E402
import oracle                               # noqa: E402
from dyobav_mpcnwta_warehouse_amd.solver import CPU_FORWARD_FLOPS_PER_S, forward_flops   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    out = []
    for key in ("cfg1_b1024_n20_2x5", "cfg2_b65536_n20_4x10"):
        spec = dict(nm.scenarios.BENCH_CONFIGS[key])
        lay = spec.pop("layout")
        spec.pop("B")
        for fam in ("toward_robot", "passing"):
            P = nm.scenarios.make_batch_chunked(n, lay, ped_mode=fam, dtype=np.float64, **spec)
            pr = oracle.Problem(lay.N, lay.Nother, lay.Nstc, lay.Ndyn)
            oracle.solve_batch(pr, oracle.Options(max_outer=1, max_inner=5), P[:2], nthreads=1)
            t0 = time.perf_counter()
            _, r = oracle.solve_batch(pr, oracle.Options(), P, nthreads=1)
            dt = time.perf_counter() - t0
            pts = int(r["n_points"].sum())
            ff = forward_flops(lay.N, lay.Nother, lay.Nstc, lay.Ndyn)
            out.append({"dims": key, "family": fam, "instances": n, "seconds": round(dt, 3), "points": pts,
                        "cost_calls": int(r["n_cost_evals"].sum()), "grad_calls": int(r["n_grad_evals"].sum()),
                        "points_per_s_per_core": round(pts / dt, 1), "forward_flops_per_point": ff,
                        "forward_flops_per_s_per_core": round(pts * ff / dt, 0)})
            print(json.dumps(out[-1]))
    rate = float(np.median([o["forward_flops_per_s_per_core"] for o in out]))
    print(json.dumps({"median_forward_flops_per_s_per_core": rate,
                      "solver.CPU_FORWARD_FLOPS_PER_S": CPU_FORWARD_FLOPS_PER_S}))


if __name__ == "__main__":
    main()
