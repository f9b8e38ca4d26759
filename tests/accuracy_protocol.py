"""The accuracy protocol of SURVEY.md 8(d) on the BASELINE workload generators -- TEST / BENCH INFRASTRUCTURE.

For every (configuration, scenario family) it solves the same seeded instances on both sides and reports
  * same-status fraction, number of instances that converged on both sides,
  * max / median |u - u_ref| over those, the fraction of them below 1e-4 (north_star's bar), median over all;
three comparisons each:
  ``hip64_vs_oracle64``        HIP fp64 against the CPU oracle (fp64), SAME Lipschitz-estimator step on both sides
                               (1e-4; OpEn's 1e-12 makes the first step length differ by ~1e-3 between any two
                               summation orders, DESIGN.md "parity protocol"), default tolerances and iteration caps
  ``hip64_vs_oracle64_tight``  the same at tolerance 1e-8 with the caps raised (2000 inner x 15 outer)
  ``hip32_vs_hip64``           HIP fp32 against HIP fp64, default tolerances, the same step
  ``hip32polish_vs_hip64polish``  fp32 solve + fp64 continuation of its converged instances (nmpc_config.polish: tolerance
                               1e-6, OpEn's fp64 Lipschitz step) against the fp64 solve + the same continuation, over the
                               instances polished on both sides -- the headline dtype with fp64-grade answers
  ``hip32polish_vs_hip64_tight``  ... against the fp64 solve from scratch at tolerance 1e-8 (the fixed point)
  ``hip64_vs_hip64_tight``     what the DEFAULT tolerance alone pins: fp64 at 1e-4 against fp64 at 1e-8
Used by ``tests/test_gpu_accuracy.py`` (asserts) and by ``bench.py`` (reports the table in its JSON line). Imports the
oracle, so it lives under ``tests/``; the product package never imports it.
"""
from __future__ import annotations

import numpy as np

LIP_STEP = 1e-4
TIGHT = dict(tolerance=1e-8, initial_tolerance=1e-8, delta_tolerance=1e-8)
TIGHT_CAPS = dict(max_inner=2000, max_outer=15)

# workload name -> (scenario key of scenarios.BENCH_CONFIGS, instances per family)
WORKLOADS = {"cfg1": ("cfg1_b1024_n20_2x5", 48), "cfg2": ("cfg2_b65536_n20_4x10", 32), "cfg4": ("cfg4_b8192_n40_8x20", 8)}
FAMILIES = ("toward_robot", "passing")


def _stats(Ua, sa, Ub, sb):
    du = np.abs(np.asarray(Ua, dtype=np.float64) - np.asarray(Ub, dtype=np.float64)).max(axis=1)
    both = (sa == 0) & (sb == 0)
    out = {"n": int(len(du)), "same_status_frac": float(np.mean(sa == sb)), "both_converged": int(both.sum()),
           "median_abs_du_all": float(np.median(du)),
           "max_abs_du_both_converged": None, "median_abs_du_both_converged": None, "frac_lt_1e-4_both_converged": None}
    if both.any():
        out["max_abs_du_both_converged"] = float(du[both].max())
        out["median_abs_du_both_converged"] = float(np.median(du[both]))
        out["frac_lt_1e-4_both_converged"] = float(np.mean(du[both] < 1e-4))
    return out


def _stats_mask(Ua, Ub, mask):
    du = np.abs(np.asarray(Ua, dtype=np.float64) - np.asarray(Ub, dtype=np.float64)).max(axis=1)[mask]
    if du.size == 0:
        return {"n": 0}
    return {"n": int(du.size), "median_abs_du": float(np.median(du)), "q90_abs_du": float(np.quantile(du, 0.9)),
            "max_abs_du": float(du.max()), "frac_lt_1e-4": float(np.mean(du < 1e-4))}


def config_for_layout(nm, layout, n_active, **overrides):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = layout.N, layout.Nother, layout.Nstc, layout.Ndyn
    cfg.max_active_dynobs = n_active
    cfg.lip_eps_f64 = cfg.lip_delta_f64 = cfg.lip_eps_f32 = cfg.lip_delta_f32 = LIP_STEP
    for k, v in overrides.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    return cfg


def run_case(nm, oracle, workload: str, family: str, n: int | None = None, seed: int = 1234, nthreads: int = 8,
             tight: bool = True, akkt_form: int = 0) -> dict:
    """One (configuration, family) row of the table."""
    key, n_default = WORKLOADS[workload]
    spec = dict(nm.scenarios.BENCH_CONFIGS[key])
    layout = spec.pop("layout")
    spec.pop("B")
    spec.pop("seed")
    n = n or n_default
    P = nm.scenarios.make_batch(n, layout, seed=seed, ped_mode=family, **spec)
    pr = oracle.Problem(layout.N, layout.Nother, layout.Nstc, layout.Ndyn)
    n_active = spec["n_ped"] * spec["n_hyp"]
    row = {"workload": workload, "family": family, "lipschitz_step": LIP_STEP, "akkt_form": akkt_form}

    def hip(dtype, **ov):
        with nm.Handle(config_for_layout(nm, layout, n_active, akkt_form=akkt_form, **ov)) as h:
            return h.solve(P.astype(dtype), dtype=dtype)

    # default tolerances / caps
    Uo, ro = oracle.solve_batch(pr, oracle.Options(lip_delta=LIP_STEP, lip_eps=LIP_STEP, akkt_form=akkt_form), P,
                                nthreads=nthreads)
    r64 = hip(np.float64)
    r32 = hip(np.float32)
    row["hip64_vs_oracle64"] = _stats(r64["U"], r64["status"], Uo, ro["status"])
    row["hip32_vs_hip64"] = _stats(r32["U"], r32["status"], r64["U"], r64["status"])
    row["converged_frac"] = {"oracle64": float(np.mean(ro["status"] == 0)), "hip64": float(np.mean(r64["status"] == 0)),
                             "hip32": float(np.mean(r32["status"] == 0))}
    # fp64 continuation of the converged instances (the continuation runs with OpEn's own fp64 Lipschitz step)
    open_step = dict(lip_eps_f64=1e-6, lip_delta_f64=1e-12)
    r32p, r64p = hip(np.float32, polish=1, **open_step), hip(np.float64, polish=1, **open_step)
    p32, p64 = r32p["info"][:, 6] == 1, r64p["info"][:, 6] == 1
    row["polish"] = {"selected32": int((r32p["info"][:, 6] >= 1).sum()), "replaced32": int(p32.sum()),
                     "selected64": int((r64p["info"][:, 6] >= 1).sum()), "replaced64": int(p64.sum())}
    row["hip32polish_vs_hip64polish"] = _stats_mask(r32p["U"], r64p["U"], p32 & p64)
    if tight:
        Uot, rot = oracle.solve_batch(pr, oracle.Options(lip_delta=LIP_STEP, lip_eps=LIP_STEP, akkt_form=akkt_form,
                                                         **TIGHT, **TIGHT_CAPS), P, nthreads=nthreads)
        r64t = hip(np.float64, max_inner_iterations=TIGHT_CAPS["max_inner"], max_outer_iterations=TIGHT_CAPS["max_outer"],
                   **TIGHT)
        row["hip64_vs_oracle64_tight"] = _stats(r64t["U"], r64t["status"], Uot, rot["status"])
        row["hip64_vs_oracle64_tight"]["tolerance"] = TIGHT["tolerance"]
        # the fixed point itself: fp64 from scratch at 1e-8 with OpEn's own Lipschitz step (the 1e-4 step of the parity
        # runs above leaves u perturbed by up to that much whenever an inner solve exits at its first test)
        if workload == "cfg4":          # (N = 40 in fp64 from scratch at 1e-8: minutes; the polish rows above stand alone)
            return row
        r64f = hip(np.float64, max_inner_iterations=TIGHT_CAPS["max_inner"], max_outer_iterations=TIGHT_CAPS["max_outer"],
                   **TIGHT, **open_step)
        t_ok = r64f["status"] == 0
        row["hip32polish_vs_hip64_tight"] = _stats_mask(r32p["U"], r64f["U"], p32 & t_ok)
        row["hip64polish_vs_hip64_tight"] = _stats_mask(r64p["U"], r64f["U"], p64 & t_ok)
        row["hip64_vs_hip64_tight"] = _stats_mask(r64["U"], r64f["U"], (r64["status"] == 0) & t_ok)
        row["hip32_vs_hip64_tight"] = _stats_mask(r32["U"], r64f["U"], (r32["status"] == 0) & t_ok)
    return row


def run_protocol(nm, oracle, workloads=("cfg1", "cfg2", "cfg4"), families=FAMILIES, scale: float = 1.0,
                 nthreads: int = 8, tight: bool = True) -> list:
    rows = []
    for w in workloads:
        for f in families:
            n = max(4, int(WORKLOADS[w][1] * scale))
            rows.append(run_case(nm, oracle, w, f, n=n, nthreads=nthreads, tight=tight))
    return rows
