#!/usr/bin/env python3
"""Measurement of the f3 path (batched closed-loop evaluator): B warehouse scenarios advanced in lock-step; one JSON
line with scenario-steps/s, the share of the wall time spent inside the solve kernel, and the outcome statistics.
The comparison figure is the reference's own way of doing this: one scenario after another, one solve per step.
   usage: bench_evaluate.py [B] [max_steps] [f32|f64] [n_ped] [n_hyp]
   n_ped x n_hyp = 2 x 1 (default): the shipped yaml's dimensions with the constant-velocity predictor (Ndynobs = 15);
   4 x 10: BASELINE configs[2]'s dimensions (Ndynobs = 40), every pedestrian fanned into 10 hypotheses.
   env: FAMILY=reference (default: the reference's scenario_0..2 on its warehouse map, HUMAN_STAGGER 0.5,
        scenarios.make_reference_scenarios) | corridor (round 5's builder-designed family);
        BUDGET=yaml (nmpc_config.max_evaluations from mpc_fast.yaml's max_solver_time = 0.1 s, solver.evaluation_budget) |
        <count> | 0 (default: iteration caps only); DISPATCH=index; TAIL=<nmpc_config.tail_latency> (-1 = no tail hand-off)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd.evaluate import BatchEvaluator
from dyobav_mpcnwta_warehouse_amd.solver import evaluation_budget

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dtype = {"f32": np.float32, "f64": np.float64}[sys.argv[3] if len(sys.argv) > 3 else "f32"]
n_ped = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n_hyp = int(sys.argv[5]) if len(sys.argv) > 5 else 1
family = os.environ.get("FAMILY", "reference")
make = nm.scenarios.make_reference_scenarios if family == "reference" else nm.scenarios.make_closed_loop_scenarios
stagger = nm.scenarios.HUMAN_STAGGER if family == "reference" else 0.2
sc = make(B, seed=13, n_ped=n_ped)
sidx = sc.pop("scenario_index", None)
cfg = nm.default_config_struct()
if n_ped * n_hyp > cfg.Ndynobs:
    cfg.Ndynobs = n_ped * n_hyp
cfg.max_active_dynobs = n_ped * n_hyp
budget = os.environ.get("BUDGET", "0")
cfg.max_evaluations = (evaluation_budget(100_000, cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs) if budget == "yaml"
                       else int(budget))
cfg.tail_latency = int(os.environ.get("TAIL", "0"))
warm = make(64, seed=14, n_ped=n_ped)
warm.pop("scenario_index", None)
BatchEvaluator(cfg, dtype=dtype, n_hyp=n_hyp, **warm).run(max_steps=3)  # warm-up
ev = BatchEvaluator(cfg, dtype=dtype, human_stagger=stagger, seed=5, n_hyp=n_hyp, **sc)
ev.count_status = True
if os.environ.get("DISPATCH") == "index":      # (diagnostic: switch the history-based dispatch order off)
    ev.dispatch_by_history = False
torch.cuda.synchronize()
t0 = time.perf_counter()
res = ev.run(max_steps=max_steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
scen_steps = int(res.steps.sum())
status_counts = torch.stack(ev.status_counts).cpu().numpy() if ev.status_counts else np.zeros((0, 4), dtype=np.int64)
ok = res.complete & ~res.collision


def metric_block(m):
    if not m.any():
        return None
    return {"runs": int(m.sum()), "smoothness_mean": [float(v) for v in np.nanmean(res.smoothness[m], axis=0)],
            "clearance_mean": float(res.clearance[m].mean()), "clearance_dyn_mean": float(res.clearance_dyn[m].mean()),
            "deviation_mean": float(res.deviation[m, 0].mean()), "deviation_max": float(res.deviation[m, 1].max())}


metrics = metric_block(ok)
by_scenario = None
if sidx is not None:
    by_scenario = {f"scenario_{k}": {"runs": int((sidx == k).sum()), "success_rate": float(ok[sidx == k].mean()),
                                     "mean_steps": float(res.steps[sidx == k].mean()), "metrics": metric_block(ok & (sidx == k))}
                   for k in np.unique(sidx)}
print(json.dumps({
    "metric": "scenario time-steps/sec (f3, batched closed-loop evaluation)", "value": scen_steps / el, "unit": "steps/s",
    "n_gpus": 1, "dtype": "f32" if dtype == np.float32 else "f64", "wall_s": el,
    "family": family, "max_evaluations": int(cfg.max_evaluations), "tail_latency": int(cfg.tail_latency), "human_stagger": stagger,
    "config": {"workload": f"B={B} scenarios x <= {max_steps} steps, mpc_fast.yaml, " + ("scenario_0..2 on the 55-polygon warehouse map" if family == "reference" else "14 map boxes") + f", {n_ped} pedestrians x {n_hyp} "
                           f"hypotheses (Ndynobs = {cfg.Ndynobs}; " + ("constant-velocity rows" if n_hyp == 1 else "fan around the constant-velocity prediction") + ")"},
    "lockstep_steps": len(res.solve_ms), "scenario_steps": scen_steps,
    "solve_kernel_ms_total": float(np.sum(res.solve_ms)), "solve_kernel_share": float(np.sum(res.solve_ms)) * 1e-3 / el,
    "solve_kernel_ms_per_step": [round(float(x), 2) for x in res.solve_ms[:6]] + ["..."] + [round(float(x), 2) for x in res.solve_ms[-3:]],
    # every lock-step: scenarios still running, kernel time of their solves, solves/s of that step
    "per_step": [{"step": kt, "running": int((res.steps > kt).sum()), "solve_ms": round(float(ms), 2),
                  "solves_per_s": round(float((res.steps > kt).sum()) / (float(ms) * 1e-3)),
                  "converged": int(sc_[0]), "out_of_iterations": int(sc_[1]), "out_of_time": int(sc_[2])}
                 for kt, (ms, sc_) in enumerate(zip(res.solve_ms, status_counts))],
    "converged_frac": float(status_counts[:, 0].sum() / max(status_counts.sum(), 1)),
    "out_of_time_frac": float(status_counts[:, 2].sum() / max(status_counts.sum(), 1)),
    # the four metrics of main_pre.py:20-53 / main_base.py:427-435 over the runs that succeeded (no collision, no time-out)
    "metrics_of_successful_runs": metrics,
    "by_scenario": by_scenario,
    "dispatch": "longest first by the previous time step's evaluation counts" if ev.dispatch_by_history and B >= ev.dispatch_min_batch else "index order",
    "complete_rate": float(res.complete.mean()), "collision_rate": float((res.collision & ~res.complete).mean()),
    "mean_steps": float(res.steps.mean())}))
