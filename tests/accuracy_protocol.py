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

import os

import numpy as np

# threads of the oracle legs: the GPU boxes have >= 16 host cores, this container 8
HOST_THREADS = max(1, min(16, os.cpu_count() or 8))

LIP_STEP = 1e-4
TIGHT = dict(tolerance=1e-8, initial_tolerance=1e-8, delta_tolerance=1e-8)
TIGHT_CAPS = dict(max_inner=2000, max_outer=15)
# Lipschitz-estimator step of the TIGHT runs (both sides). OpEn's estimator leaves u perturbed by its step h (u <- u + h,
# every component), and an outer iteration whose inner solve exits at its first test returns that perturbed point: with the
# 1e-4 of the default-tolerance runs a tight pair can end 1e-4..3e-4 apart for no other reason than WHICH side took that
# exit (measured round 5, configs[1] `passing`, oracle vs its twin: instance 59, 1.3e-4 -- the oracle's second outer
# iteration ran 0 inner iterations, the twin's 147). 1e-7 keeps the finite difference far above fp64 rounding
# (|grad(u + h) - grad(u)| ~ L h ~ 1e-3 against 1e-12) and the perturbation three orders below the 1e-4 bar.
LIP_STEP_TIGHT = 1e-7
# An end point counts as a KKT point of its final inner problem when the natural residual ||u - Proj_U(u - grad psi(u; c, y))||_inf
# -- evaluated by the OTHER implementation -- is below this. Genuine stationary points of these problems come out at 1e-5..2e-3
# (the exit test bounds ||gamma fpr||, i.e. this residual only to ~tol / gamma); end points "converged" after the penalty has
# escalated to 1e7..1e10 come out at 0.5..2: there gamma ~ 1 / c makes ||gamma fpr|| < tol true at non-stationary points.
RHO_KKT = 5e-3     # (1e-2 was 5x above the largest genuine stationary residual measured; 2.5x now, 100x below the escalated ones)

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
START_TOL = 1e-8        # distance of the iterates over the first records (same algorithm, same start), Lipschitz step 1e-4
START_TOL_TIGHT = 1e-6  # ... with the tight runs' step of 1e-7: the first step length is gamma = 0.95 / L with L a finite
                        # difference of gradients over that step, so its rounding error -- the distance after the first
                        # iteration -- grows with 1 / step: measured <= 1.6e-10 at 1e-4 (512 instances), hence <= ~2e-7 at 1e-7
                        # (observed: 1.4e-8 HIP vs oracle, 8e-9 oracle vs twin)
JUMP_LIMIT = 1e8        # growth of the distance within ONE iteration, from a level above rounding (the oracle against its own
                        # re-associated twin reaches 5e6 on one of 512 instances; a difference in a rule jumps >= 1e9)
MARGIN_FACTOR = 1e4     # a differing decision is a tie-break if its relative margin <= MARGIN_FACTOR * (distance before it,
                        # at least 1e-13) * max(1, ||grad psi|| / |psi|) -- what that distance is worth in relative psi


def audit_pair(head_a, U_a, head_b, U_b, start_tol: float = START_TOL) -> dict:
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
    ok = start <= start_tol and max_jump <= JUMP_LIMIT
    if k_disc >= 0:
        before = float(d[k_disc - 1]) if k_disc > 0 else 0.0
        margin = float(min(abs(head_b[k_disc, 12]), abs(head_b[max(k_disc - 1, 0), 12]))) if head_b.shape[1] > 12 else None
        fields = ("outer", "inner", "lipschitz_doublings", "linesearch_halvings", "lbfgs_pair", "penalty")
        which = [fields[j] for j, f in enumerate(DISCRETE_FIELDS) if head_a[k_disc, f] != head_b[k_disc, f]]
        out.update({"distance_before_it": before, "decision_margin": margin, "differing": which,
                    "record_a": [float(x) for x in head_a[k_disc, :16]], "record_b": [float(x) for x in head_b[k_disc, :16]]})
        # the decision that differs must be one the distance already there can flip: its margin (the oracle's record) no
        # larger than MARGIN_FACTOR x that distance x the steepness of psi -- while the iterates still agree to 1e-6 this
        # makes it a near-tie; later (round 4 accepted ANY late decision) it is the same bound with the
        # distance reached by then
        steep = float(max(1.0, head_b[k_disc, 15], head_b[max(k_disc - 1, 0), 15])) if head_b.shape[1] > 15 else 1.0
        out["steepness"] = steep
        out["margin_allowed"] = MARGIN_FACTOR * max(before, 1e-13) * steep
        ok = ok and margin is not None and margin <= out["margin_allowed"]
        out["kind"] = "discrete decision after gradual growth" if before >= 1e-6 else "near-tie decision"
    else:
        out["kind"] = "gradual growth, no differing decision" if k_far >= 0 else "agree throughout"
    out["explained"] = bool(ok)
    return out


def divergence_audit(nm, oracle, pr, cfg, P, pairs, opts, reassoc_pairs=(), nthreads: int = HOST_THREADS, start_tol: float = START_TOL) -> dict:
    """Audit of the instances `pairs` (HIP one-wavefront fp64 kernel vs oracle) and `reassoc_pairs` (oracle vs its
    re-associated twin: what the same audit says about two CPU implementations). The oracle's traces run on a thread pool
    (ctypes releases the GIL); a tight-tolerance trace is up to 30 000 iterations."""
    from concurrent.futures import ThreadPoolExecutor
    need = sorted(set(int(i) for i in pairs) | set(int(i) for i in reassoc_pairs))
    with ThreadPoolExecutor(max(1, nthreads)) as ex:
        base = dict(zip(need, ex.map(lambda i: oracle.solve_trace(pr, opts, P[i])[3:], need)))
        twin = dict(zip([int(i) for i in reassoc_pairs],
                        ex.map(lambda i: oracle.solve_trace(pr, opts, P[i], reassoc=True)[3:], [int(i) for i in reassoc_pairs])))
    rows, rows_r = [], []
    with nm.Handle(cfg) as h:
        for i in pairs:
            t = h.solve_trace(P[i])
            ho, Uo = base[int(i)]
            r = audit_pair(t["head"], t["Ut"], ho, Uo, start_tol)
            r["instance"] = int(i)
            rows.append(r)
    for i in reassoc_pairs:
        ha, Ua = twin[int(i)]
        ho, Uo = base[int(i)]
        r = audit_pair(ha, Ua, ho, Uo, start_tol)
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



# ---- tight-tolerance pairs: are both end points KKT points? (VERDICT r4 "What's weak" 1) -------------------------------
def _box(pr):
    N = pr.N
    return np.tile([pr.lin_vel_min, -pr.ang_vel_max], N), np.tile([pr.lin_vel_max, pr.ang_vel_max], N)


def natural_residual(u, grad, lo, hi):
    """||u - Proj_U(u - grad)||_inf: zero exactly at the stationary points of min psi over the box U (unit step)."""
    return float(np.abs(u - np.clip(u - grad, lo, hi)).max())


def oracle_solve_full(oracle, pr, opts, P, idx, nthreads=HOST_THREADS, reassoc=False):
    """(U, Y, result records) of the instances `idx`, one oracle.solve each (solve_batch does not hand back the
    multipliers), threads over instances (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, nthreads)) as ex:
        out = list(ex.map(lambda i: oracle.solve(pr, opts, P[i], reassoc=reassoc), idx))
    U = np.array([o[0] for o in out]).reshape(len(idx), 2 * pr.N)
    Y = np.array([o[1] for o in out]).reshape(len(idx), 2 * pr.N)
    return U, Y, [o[2] for o in out]


def kkt_classification(oracle, pr, P, idx, end_a, end_b, eval_hip=None) -> dict:
    """Every pair `idx` that converged on both sides at the tight tolerance, classified from its two end points
    end_x = (U[n, 2N], Y[n, 2N], C[n]) -- the controls, the multipliers and the penalty of the final inner problem:
      f, ||F2||_inf, dist_C(F1) at both end points (the oracle's problem functions: pinned to the reference), psi / grad psi
      at (u; c, y) by the oracle AND -- eval_hip(Pn, U, Y, C) -> dict(psi, grad), nmpc_eval_batch_f64 -- by the device, the
      natural residual rho = ||u - Proj_U(u - grad psi)||_inf from both evaluators.
    kind:  "agree"             the end points are within 1e-4;
           "not_kkt"           > 1e-4 apart and an end point is not a stationary point of its own final inner problem
                               (rho > RHO_KKT by the independent evaluator): the exit test ||gamma fpr|| < tol was met
                               because gamma ~ 1 / c after the penalty escalated -- no solver-independent solution exists
                               to agree on, and two runs of ANY implementation stop wherever their paths were;
           "second_kkt_point"  > 1e-4 apart, both end points stationary and feasible, different cost: another local
                               minimum of the non-convex problem;
           "unexplained"       > 1e-4 apart, both stationary, same cost: nothing accounts for it (a bug until shown otherwise).
    """
    lo, hi = _box(pr)
    N = pr.N
    clo = np.r_[np.full(N, pr.lin_acc_min), np.full(N, -pr.ang_acc_max)]
    chi = np.r_[np.full(N, pr.lin_acc_max), np.full(N, pr.ang_acc_max)]
    rows = []
    hip = {}
    if eval_hip is not None and len(idx):
        for name, (U, Y, Cc) in (("a", end_a), ("b", end_b)):
            hip[name] = eval_hip(P[idx], U, Y, Cc)
    for n, i in enumerate(idx):
        rec = {"instance": int(i)}
        for name, (U, Y, Cc) in (("a", end_a), ("b", end_b)):
            u, y, c = U[n], Y[n], float(Cc[n])
            f, F1, F2 = oracle.eval_problem(pr, u, P[i])
            v, g = oracle.psi(pr, u, c, y, P[i])
            rho = natural_residual(u, g, lo, hi)
            rec[name] = {"f": f, "f2_inf": float(np.abs(F2).max()) if F2.size else 0.0,
                         "dist_C_F1": float(np.abs(F1 - np.clip(F1, clo, chi)).max()), "penalty": c, "rho_oracle": rho}
            if hip:
                gh = hip[name]["grad"][n]
                rec[name]["rho_hip"] = natural_residual(u, gh, lo, hi)
                # (relative to the size of psi: at a stationary point the gradient itself is ~1e-5 and made of cancelling terms)
                rec[name]["grad_rel_diff_hip_vs_oracle"] = float(np.abs(gh - g).max() / max(1.0, abs(v), np.abs(g).max()))
                rec[name]["psi_rel_diff_hip_vs_oracle"] = float(abs(hip[name]["psi"][n] - v) / max(1.0, abs(v)))
        du = float(np.abs(end_a[0][n] - end_b[0][n]).max())
        rho_max = max(max(rec[s].get("rho_hip", 0.0), rec[s]["rho_oracle"]) for s in ("a", "b"))
        # an end point counts as NOT stationary only if BOTH evaluators say so (`rho_max` alone took the
        # larger of the two, so that one evaluator's error could excuse a pair); "both end points stationary" keeps the
        # stricter reading: the larger residual of either evaluator at either end
        rho_nonstat = max(min(rec[s].get("rho_hip", rec[s]["rho_oracle"]), rec[s]["rho_oracle"]) for s in ("a", "b"))
        df = abs(rec["a"]["f"] - rec["b"]["f"])
        rec.update({"abs_du": du, "rho_max": rho_max, "rho_nonstationary_by_both": rho_nonstat, "abs_df": df})
        if du <= 1e-4:
            rec["kind"] = "agree"
        elif rho_nonstat > RHO_KKT:
            rec["kind"] = "not_kkt"
        elif df > 1e-9 * max(1.0, abs(rec["a"]["f"])):
            rec["kind"] = "second_kkt_point"
        else:
            rec["kind"] = "unexplained"
        rows.append(rec)
    kinds = [r["kind"] for r in rows]
    kkt = [r for r in rows if r["rho_max"] <= RHO_KKT]
    out = {"n_pairs": len(rows), "n_agree": kinds.count("agree"), "n_not_kkt": kinds.count("not_kkt"),
           "n_second_kkt_point": kinds.count("second_kkt_point"), "n_unexplained": kinds.count("unexplained"),
           # the well-posed comparison: both end points certified stationary by the independent evaluator
           "n_both_kkt": len(kkt), "max_abs_du_both_kkt": max([r["abs_du"] for r in kkt], default=None),
           "frac_lt_1e-4_both_kkt": (float(np.mean([r["abs_du"] < 1e-4 for r in kkt])) if kkt else None),
           "max_rho_of_agreeing_pairs": max([r["rho_max"] for r in rows if r["kind"] == "agree"], default=None),
           "min_rho_of_not_kkt_pairs": min([r["rho_max"] for r in rows if r["kind"] == "not_kkt"], default=None),
           "max_grad_rel_diff_hip_vs_oracle": max([r[s].get("grad_rel_diff_hip_vs_oracle", 0.0) for r in rows for s in ("a", "b")],
                                                  default=None),
           "max_psi_rel_diff_hip_vs_oracle": max([r[s].get("psi_rel_diff_hip_vs_oracle", 0.0) for r in rows for s in ("a", "b")],
                                                 default=None),
           "far_pairs": [r for r in rows if r["kind"] != "agree"]}
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


def run_case(nm, oracle, workload: str, family: str, n: int | None = None, seed: int = 1234, nthreads: int = HOST_THREADS,
             tight: bool = True, akkt_form: int = 0, audit: bool = False, audit_max: int = 24, tight_audit: bool | None = None,
             n_tight: int | None = None, n_polish: int | None = None) -> dict:
    """One (configuration, family) row of the table: seeded instances of the BASELINE generator. ``n_polish``: the polish
    legs (device against device, no oracle involved) on their own, larger sample of that many instances."""
    key, n_default = WORKLOADS[workload]
    spec = dict(nm.scenarios.BENCH_CONFIGS[key])
    layout = spec.pop("layout")
    spec.pop("B")
    spec.pop("seed")
    n = n or n_default
    P = nm.scenarios.make_batch(n, layout, seed=seed, ped_mode=family, **spec)
    P_polish = None if not n_polish else nm.scenarios.make_batch(n_polish, layout, seed=seed + 1, ped_mode=family, **spec)
    return run_case_on(nm, oracle, P, layout, spec["n_ped"] * spec["n_hyp"], workload, family, nthreads=nthreads, tight=tight,
                       akkt_form=akkt_form, audit=audit, audit_max=audit_max, tight_audit=tight_audit,
                       fixed_point=workload != "cfg4", n_tight=n_tight, P_polish=P_polish)


def run_case_on(nm, oracle, P, layout, n_active: int, workload: str, family: str, nthreads: int = HOST_THREADS, tight: bool = True,
                akkt_form: int = 0, audit: bool = False, audit_max: int = 24, tight_audit: bool | None = None,
                fixed_point: bool = True, polish: bool = True, n_tight: int | None = None, P_polish=None) -> dict:
    """The protocol on a given parameter batch ``P[n, np]`` (fp64) of the dimensions ``layout`` with at most ``n_active``
    non-zero obstacle rows -- the BASELINE generators (run_case) or batches harvested from the closed loop
    (scenarios.harvest_closed_loop). ``n_tight``: the tolerance-1e-8 legs (whose CPU side runs up to 2000 x 15
    iterations per instance) use the first n_tight instances only."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    n = P.shape[0]
    pr = oracle.Problem(layout.N, layout.Nother, layout.Nstc, layout.Ndyn)
    row = {"workload": workload, "family": family, "n": int(n), "lipschitz_step": LIP_STEP, "akkt_form": akkt_form}
    tight_audit = audit if tight_audit is None else tight_audit

    def hip(dtype, **ov):
        with nm.Handle(config_for_layout(nm, layout, n_active, akkt_form=akkt_form, **ov)) as h:
            return h.solve(P.astype(dtype), dtype=dtype)

    # default tolerances / caps
    # (hoist_trig: cos / sin of the ellipse angles once per solve instead of per evaluation -- the same bits, tested in
    #  tests/test_oracle_solver.py, for 1.3-2.4x less CPU time in the legs that bound the GPU suite's wall time)
    opt = oracle.Options(lip_delta=LIP_STEP, lip_eps=LIP_STEP, akkt_form=akkt_form, hoist_trig=1)
    Uo, ro = oracle.solve_batch(pr, opt, P, nthreads=nthreads)
    r64 = hip(np.float64)
    r32 = hip(np.float32)
    row["hip64_vs_oracle64"] = _stats(r64["U"], r64["status"], Uo, ro["status"])
    # the oracle's own noise floor under re-association, same instances, same options
    Ur, rr = oracle.solve_batch(pr, opt, P, nthreads=nthreads, reassoc=True)
    row["oracle64_vs_reassociated"] = _stats(Ur, rr["status"], Uo, ro["status"])
    tp = dict(latency_waves=1, coop_waves=1, reg_table=-1)    # the one-wavefront fp64 kernel nmpc_solve_trace_f64 traces
    if audit:
        # first-divergence audit: that kernel against the oracle, every instance that ends > 1e-4 apart or with another
        # status (the first audit_max of them; n_far says how many there were); and the oracle against its twin, likewise
        rtp = hip(np.float64, **tp)
        row["hip64tp_vs_oracle64"] = _stats(rtp["U"], rtp["status"], Uo, ro["status"])
        far_all = lambda Ua, sa, Ub, sb: np.nonzero((np.abs(Ua - Ub).max(axis=1) > 1e-4) | (sa != sb))[0]
        fa, fr = far_all(rtp["U"], rtp["status"], Uo, ro["status"]), far_all(Ur, rr["status"], Uo, ro["status"])
        row["divergence_audit"] = divergence_audit(
            nm, oracle, pr, config_for_layout(nm, layout, n_active, akkt_form=akkt_form, **tp), P, fa[:audit_max], opt, fr[:audit_max],
            nthreads=nthreads)
        row["divergence_audit"]["n_far"] = int(len(fa))
        row["divergence_audit"]["oracle_vs_reassociated"]["n_far"] = int(len(fr))
    row["hip32_vs_hip64"] = _stats(r32["U"], r32["status"], r64["U"], r64["status"])
    row["converged_frac"] = {"oracle64": float(np.mean(ro["status"] == 0)), "hip64": float(np.mean(r64["status"] == 0)),
                             "hip32": float(np.mean(r32["status"] == 0))}
    # fp64 continuation of the converged instances (the continuation runs with OpEn's own fp64 Lipschitz step)
    open_step = dict(lip_eps_f64=1e-6, lip_delta_f64=1e-12)
    if polish:
        if P_polish is not None:     # (device against device: a larger sample of its own costs no oracle time)
            P_keep, P = P, np.ascontiguousarray(P_polish, dtype=np.float64)
        r32p, r64p = hip(np.float32, polish=1, **open_step), hip(np.float64, polish=1, **open_step)
        if P_polish is not None:
            P = P_keep
        p32, p64 = r32p["info"][:, 6] == 1, r64p["info"][:, 6] == 1
        row["polish"] = {"n": int(len(p32)), "selected32": int((r32p["info"][:, 6] >= 1).sum()), "replaced32": int(p32.sum()),
                         "selected64": int((r64p["info"][:, 6] >= 1).sum()), "replaced64": int(p64.sum())}
        row["hip32polish_vs_hip64polish"] = _stats_mask(r32p["U"], r64p["U"], p32 & p64)
    if tight:
        # ---- tolerance 1e-8, caps 2000 x 15, Lipschitz step LIP_STEP_TIGHT on both sides: where two correct solvers that
        #      reach a stationary point must agree
        caps = dict(max_inner_iterations=TIGHT_CAPS["max_inner"], max_outer_iterations=TIGHT_CAPS["max_outer"])
        tl = dict(lip_eps_f64=LIP_STEP_TIGHT, lip_delta_f64=LIP_STEP_TIGHT)
        opt_t = oracle.Options(lip_delta=LIP_STEP_TIGHT, lip_eps=LIP_STEP_TIGHT, akkt_form=akkt_form, hoist_trig=1, **TIGHT, **TIGHT_CAPS)
        nt = n if n_tight is None else min(n, int(n_tight))
        P_all, P = P, P[:nt]                 # (from here on `hip` and every index refer to the tight subset)
        every = np.arange(nt)
        row["n_tight"] = int(nt)
        Uot, Yot, rot = oracle_solve_full(oracle, pr, opt_t, P, every, nthreads)
        sot, cot = np.array([r["status"] for r in rot]), np.array([r["penalty"] for r in rot])
        r64t = hip(np.float64, **caps, **TIGHT, **tl)
        row["hip64_vs_oracle64_tight"] = _stats(r64t["U"], r64t["status"], Uot, sot)
        row["hip64_vs_oracle64_tight"].update(tolerance=TIGHT["tolerance"], lipschitz_step=LIP_STEP_TIGHT)
        # the same for the oracle against its re-associated twin: the noise floor AT THE TIGHT TOLERANCE
        Urt, Yrt, rrt = oracle_solve_full(oracle, pr, opt_t, P, every, nthreads, reassoc=True)
        srt, crt = np.array([r["status"] for r in rrt]), np.array([r["penalty"] for r in rrt])
        row["oracle64_vs_reassociated_tight"] = _stats(Urt, srt, Uot, sot)
        # KKT classification of every pair that converged on both sides (kkt_classification): psi / grad psi at both end
        # points by the oracle and by the device (nmpc_eval_batch_f64, the LDS-table kernel)
        with nm.Handle(config_for_layout(nm, layout, n_active, akkt_form=akkt_form, **tp)) as he:
            ev = lambda Pn, U, Y, Cc: he.eval(Pn, U, Y, Cc, dtype=np.float64)
            both = np.nonzero((r64t["status"] == 0) & (sot == 0))[0]
            row["tight_kkt_hip64_vs_oracle64"] = kkt_classification(
                oracle, pr, P, both, (r64t["U"][both], r64t["y"][both], r64t["info"][both, 3]), (Uot[both], Yot[both], cot[both]), ev)
            both_r = np.nonzero((srt == 0) & (sot == 0))[0]
            row["tight_kkt_oracle64_vs_reassociated"] = kkt_classification(
                oracle, pr, P, both_r, (Urt[both_r], Yrt[both_r], crt[both_r]), (Uot[both_r], Yot[both_r], cot[both_r]), ev)
        if tight_audit:
            # first-divergence audit at the tight tolerance: the pairs > 1e-4 apart (or with another status), traces of
            # the one-wavefront fp64 kernel against the oracle's and of the oracle's twin against the oracle's
            rtt = hip(np.float64, **caps, **TIGHT, **tl, **tp)
            row["hip64tp_vs_oracle64_tight"] = _stats(rtt["U"], rtt["status"], Uot, sot)
            far_all = lambda Ua, sa: np.nonzero((np.abs(Ua - Uot).max(axis=1) > 1e-4) | (sa != sot))[0]
            fa, fr = far_all(rtt["U"], rtt["status"]), far_all(Urt, srt)
            cfg_t = config_for_layout(nm, layout, n_active, akkt_form=akkt_form, **caps, **TIGHT, **tl, **tp)
            na = min(audit_max, 8)          # (a tight trace is up to 30 000 records)
            row["divergence_audit_tight"] = divergence_audit(nm, oracle, pr, cfg_t, P, fa[:na], opt_t, fr[:na], nthreads=nthreads,
                                                             start_tol=START_TOL_TIGHT)
            row["divergence_audit_tight"]["n_far"] = int(len(fa))
            row["divergence_audit_tight"]["oracle_vs_reassociated"]["n_far"] = int(len(fr))
        # the fixed point itself: fp64 from scratch at 1e-8 with OpEn's own Lipschitz step
        if not fixed_point or not polish or P_polish is not None:  # (N = 40 in fp64 from scratch at 1e-8: minutes; the polish rows above stand alone)
            return row
        r64f = hip(np.float64, **caps, **TIGHT, **open_step)
        t_ok = r64f["status"] == 0
        row["hip32polish_vs_hip64_tight"] = _stats_mask(r32p["U"][:nt], r64f["U"], p32[:nt] & t_ok)
        row["hip64polish_vs_hip64_tight"] = _stats_mask(r64p["U"][:nt], r64f["U"], p64[:nt] & t_ok)
        row["hip64_vs_hip64_tight"] = _stats_mask(r64["U"][:nt], r64f["U"], (r64["status"][:nt] == 0) & t_ok)
        row["hip32_vs_hip64_tight"] = _stats_mask(r32["U"][:nt], r64f["U"], (r32["status"][:nt] == 0) & t_ok)
        # every returned instance of the headline dtype -- converged or not -- against the fp64 result of the same status
        # class: what the answers the reference would USE ANYWAY (trajectory_tracker.py:334-335 only prints a bad status)
        # carry by way of accuracy (VERDICT r4 "What's weak" 3)
        row["all_returned_hip32_vs_hip64"] = returned_accuracy(r32, r64)
        row["all_returned_hip32polish_vs_hip64polish"] = returned_accuracy(r32p, r64p)
    return row


def returned_accuracy(ra, rb) -> dict:
    """|u_a - u_b| per instance over ALL returned instances with the same exit status on both sides, by status class."""
    du = np.abs(np.asarray(ra["U"], dtype=np.float64) - np.asarray(rb["U"], dtype=np.float64)).max(axis=1)
    same = ra["status"] == rb["status"]
    out = {"n": int(len(du)), "same_status_frac": float(np.mean(same))}
    for name, m in (("converged", same & (ra["status"] == 0)), ("not_converged", same & (ra["status"] != 0)), ("all", same)):
        d = du[m]
        out[name] = {"n": int(d.size)} if d.size == 0 else {
            "n": int(d.size), "median_abs_du": float(np.median(d)), "frac_lt_1e-4": float(np.mean(d < 1e-4)),
            "frac_lt_1e-3": float(np.mean(d < 1e-3)), "max_abs_du": float(d.max())}
    return out


def run_protocol(nm, oracle, workloads=("cfg1", "cfg2", "cfg4"), families=FAMILIES, scale: float = 1.0,
                 nthreads: int = HOST_THREADS, tight: bool = True) -> list:
    rows = []
    for w in workloads:
        for f in families:
            n = max(4, int(WORKLOADS[w][1] * scale))
            rows.append(run_case(nm, oracle, w, f, n=n, nthreads=nthreads, tight=tight))
    return rows
