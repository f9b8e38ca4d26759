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
  ``oracle64_vs_reassociated`` THE ORACLE'S OWN NOISE FLOOR: the fp64 oracle against the same oracle with its sums
                               associated differently (oracle/nmpc_oracle_impl.h, ORC_REASSOC: reverse accumulation order,
                               rollout as X0 + running sum -- no rule, constant or tie-break differs), same instances, same
                               options. Two correct fp64 implementations of this solver scatter a share of the converged
                               instances by far more than 1e-4; ``hip64_vs_oracle64`` has to be read against that share
  ``divergence_audit``         the first-divergence audit (below) of every pair that ends > 1e-4 apart

First-divergence audit. Both sides record one line per inner iteration -- outer / inner index,
Lipschitz doublings, line-search halvings, what became of the L-BFGS pair, gamma, ||gamma fpr||, psi, and the iterate --
the HIP side through ``nmpc_solve_trace_f64`` (the one-wavefront fp64 kernel), the oracle through ``orc_solve_trace_*``,
which also notes the smallest relative margin of the iteration's discrete decisions. What the traces show (and what
``audit_pair`` asserts per pair): the two sides start together (1e-12: the Lipschitz estimate is a finite difference),
the distance grows GRADUALLY over tens of iterations -- the L-BFGS / line-search dynamics on this non-convex problem
amplify a rounding-level difference by a modest factor per iteration -- and only once it has reached the size of the
decision margins do discrete decisions (a doubling more, a halving more) differ and the paths part for good. A genuine
algorithmic difference looks different: a discrete decision that differs while the iterates still agree to rounding,
with a margin far above their distance, or a jump of many orders of magnitude within one iteration.

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


# ---- first-divergence audit -------------------------------------------------------------------------------------
DISCRETE_FIELDS = (0, 1, 2, 3, 4, 14)   # outer, inner index, Lipschitz doublings, line-search halvings, L-BFGS pair, penalty
                                        # (the penalty changes by the factor 5 or not at all: an outer-loop decision)
START_TOL = 1e-8        # distance of the iterates over the first records (same algorithm, same start)
JUMP_LIMIT = 1e8        # growth of the distance within ONE iteration, from a level above rounding (the oracle against its own
                        # re-associated twin reaches 5e6 on one of 512 instances; a difference in a rule jumps >= 1e9)
MARGIN_FACTOR = 1e4     # a differing decision is a tie-break if its relative margin <= MARGIN_FACTOR * (distance before it,
                        # at least 1e-13) * max(1, ||grad psi|| / |psi|) -- what that distance is worth in relative psi


def audit_pair(head_a, U_a, head_b, U_b) -> dict:
    """Lay two iteration traces of the same instance side by side. `head_b` may carry decision margins (column 12, the
    oracle's); `head_a` need not. Returns where and how the two part, and `explained`: True when the record shows
    rounding-level agreement at the start, gradual growth, and -- if a discrete decision differs while the iterates are
    still closer than 1e-6 -- a margin of that decision no larger than MARGIN_FACTOR times the distance already there."""
    n = min(len(head_a), len(head_b))
    if n == 0:
        return {"records": 0, "explained": True, "kind": "no inner iteration on either side"}
    scale = max(1.0, float(np.abs(U_b[:n]).max()))
    d = np.abs(U_a[:n] - U_b[:n]).max(axis=1) / scale
    disc = np.any(head_a[:n][:, DISCRETE_FIELDS] != head_b[:n][:, DISCRETE_FIELDS], axis=1)
    k_disc = int(np.argmax(disc)) if disc.any() else -1
    k_far = int(np.argmax(d > 1e-6)) if (d > 1e-6).any() else -1
    start = float(d[:min(3, n)].max())
    upto = n if k_disc < 0 else max(k_disc, 1)      # growth is judged up to (not across) the first differing decision
    floor = 1e-13
    jumps = d[1:upto] / np.maximum(d[:upto - 1], floor)
    max_jump = float(jumps.max()) if jumps.size else 1.0
    out = {"records": int(n), "start_distance": start, "first_discrete_difference": k_disc, "first_distance_gt_1e-6": k_far,
           "max_growth_per_iteration": max_jump}
    ok = start <= START_TOL and max_jump <= JUMP_LIMIT
    if k_disc >= 0:
        before = float(d[k_disc - 1]) if k_disc > 0 else 0.0
        margin = float(min(abs(head_b[k_disc, 12]), abs(head_b[max(k_disc - 1, 0), 12]))) if head_b.shape[1] > 12 else None
        fields = ("outer", "inner", "lipschitz_doublings", "linesearch_halvings", "lbfgs_pair", "penalty")
        which = [fields[j] for j, f in enumerate(DISCRETE_FIELDS) if head_a[k_disc, f] != head_b[k_disc, f]]
        out.update({"distance_before_it": before, "decision_margin": margin, "differing": which,
                    "record_a": [float(x) for x in head_a[k_disc, :16]], "record_b": [float(x) for x in head_b[k_disc, :16]]})
        if before < 1e-6:      # the iterates still agreed: the decision itself must have been a near-tie
            steep = float(max(1.0, head_b[k_disc, 15], head_b[max(k_disc - 1, 0), 15])) if head_b.shape[1] > 15 else 1.0
            out["steepness"] = steep
            ok = ok and margin is not None and margin <= MARGIN_FACTOR * max(before, 1e-13) * steep
        out["kind"] = "discrete decision after gradual growth" if before >= 1e-6 else "near-tie decision"
    else:
        out["kind"] = "gradual growth, no differing decision" if k_far >= 0 else "agree throughout"
    out["explained"] = bool(ok)
    return out


def divergence_audit(nm, oracle, pr, cfg, P, pairs, opts, reassoc_pairs=()) -> dict:
    """Audit of the instances `pairs` (HIP one-wavefront fp64 kernel vs oracle) and `reassoc_pairs` (oracle vs its
    re-associated twin: what the same audit says about two CPU implementations)."""
    rows, rows_r = [], []
    with nm.Handle(cfg) as h:
        for i in pairs:
            t = h.solve_trace(P[i])
            _, _, _, ho, Uo = oracle.solve_trace(pr, opts, P[i])
            r = audit_pair(t["head"], t["Ut"], ho, Uo)
            r["instance"] = int(i)
            rows.append(r)
    for i in reassoc_pairs:
        _, _, _, ha, Ua = oracle.solve_trace(pr, opts, P[i], reassoc=True)
        _, _, _, ho, Uo = oracle.solve_trace(pr, opts, P[i])
        r = audit_pair(ha, Ua, ho, Uo)
        r["instance"] = int(i)
        rows_r.append(r)

    def digest(rs):
        return {"n_pairs": len(rs), "n_explained": sum(r["explained"] for r in rs),
                "n_unexplained": sum(not r["explained"] for r in rs),
                "n_tie": sum(r.get("kind") == "near-tie decision" for r in rs),
                "median_first_discrete_difference": float(np.median([r["first_discrete_difference"] for r in rs])) if rs else None,
                "max_growth_per_iteration": max([r["max_growth_per_iteration"] for r in rs], default=None),
                "max_start_distance": max([r["start_distance"] for r in rs], default=None)}
    out = digest(rows)
    out["pairs"] = rows
    out["oracle_vs_reassociated"] = dict(digest(rows_r), pairs=rows_r)
    return out


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
             tight: bool = True, akkt_form: int = 0, audit: bool = False, audit_max: int = 24) -> dict:
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
    # the oracle's own noise floor under re-association, same instances, same options
    Ur, rr = oracle.solve_batch(pr, oracle.Options(lip_delta=LIP_STEP, lip_eps=LIP_STEP, akkt_form=akkt_form), P,
                                nthreads=nthreads, reassoc=True)
    row["oracle64_vs_reassociated"] = _stats(Ur, rr["status"], Uo, ro["status"])
    if audit:
        # first-divergence audit: the one-wavefront fp64 kernel (what nmpc_solve_trace_f64 traces) against the oracle, every
        # instance that ends > 1e-4 apart or with another status; and the oracle against its twin, likewise
        tp = dict(latency_waves=1, coop_waves=1, reg_table=-1)
        rtp = hip(np.float64, **tp)
        row["hip64tp_vs_oracle64"] = _stats(rtp["U"], rtp["status"], Uo, ro["status"])
        far = lambda Ua, sa: np.nonzero((np.abs(Ua - Uo).max(axis=1) > 1e-4) | (sa != ro["status"]))[0][:audit_max]
        row["divergence_audit"] = divergence_audit(
            nm, oracle, pr, config_for_layout(nm, layout, n_active, akkt_form=akkt_form, **tp), P, far(rtp["U"], rtp["status"]),
            oracle.Options(lip_delta=LIP_STEP, lip_eps=LIP_STEP, akkt_form=akkt_form), far(Ur, rr["status"]))
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
