#!/usr/bin/env python3
"""Measurement of the f3 path (batched closed-loop evaluator): B warehouse scenarios advanced in lock-step; one JSON
line with scenario-steps/s, the share of the wall time spent inside the solve kernel, and the outcome statistics.
The comparison figure is the reference's own way of doing this: one scenario after another, one solve per step.
   usage: bench_evaluate.py [B] [max_steps] [f32|f64] [n_ped] [n_hyp]
   n_ped x n_hyp = 2 x 1 (default): the shipped yaml's dimensions with the constant-velocity predictor (Ndynobs = 15);
   4 x 10: BASELINE configs[2]'s dimensions (Ndynobs = 40), every pedestrian fanned into 10 hypotheses."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd.evaluate import BatchEvaluator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dtype = {"f32": np.float32, "f64": np.float64}[sys.argv[3] if len(sys.argv) > 3 else "f32"]
n_ped = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n_hyp = int(sys.argv[5]) if len(sys.argv) > 5 else 1
sc = nm.scenarios.make_closed_loop_scenarios(B, seed=13, n_ped=n_ped)
cfg = nm.default_config_struct()
if n_ped * n_hyp > cfg.Ndynobs:
    cfg.Ndynobs = n_ped * n_hyp
cfg.max_active_dynobs = n_ped * n_hyp
warm = nm.scenarios.make_closed_loop_scenarios(64, seed=14, n_ped=n_ped)
BatchEvaluator(cfg, dtype=dtype, n_hyp=n_hyp, **warm).run(max_steps=3)  # warm-up
ev = BatchEvaluator(cfg, dtype=dtype, human_stagger=0.2, seed=5, n_hyp=n_hyp, **sc)
if os.environ.get("DISPATCH") == "index":      # (diagnostic: switch the history-based dispatch order off)
    ev.dispatch_by_history = False
torch.cuda.synchronize()
t0 = time.perf_counter()
res = ev.run(max_steps=max_steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
scen_steps = int(res.steps.sum())
print(json.dumps({
    "metric": "scenario time-steps/sec (f3, batched closed-loop evaluation)", "value": scen_steps / el, "unit": "steps/s",
    "n_gpus": 1, "dtype": "f32" if dtype == np.float32 else "f64", "wall_s": el,
    "config": {"workload": f"B={B} scenarios x <= {max_steps} steps, mpc_fast.yaml, 14 map boxes, {n_ped} pedestrians x {n_hyp} "
                           f"hypotheses (Ndynobs = {cfg.Ndynobs}; " + ("constant-velocity rows" if n_hyp == 1 else "fan around the constant-velocity prediction") + ")"},
    "lockstep_steps": len(res.solve_ms), "scenario_steps": scen_steps,
    "solve_kernel_ms_total": float(np.sum(res.solve_ms)), "solve_kernel_share": float(np.sum(res.solve_ms)) * 1e-3 / el,
    "solve_kernel_ms_per_step": [round(float(x), 2) for x in res.solve_ms[:6]] + ["..."] + [round(float(x), 2) for x in res.solve_ms[-3:]],
    # every lock-step: scenarios still running, kernel time of their solves, solves/s of that step
    "per_step": [{"step": kt, "running": int((res.steps > kt).sum()), "solve_ms": round(float(ms), 2),
                  "solves_per_s": round(float((res.steps > kt).sum()) / (float(ms) * 1e-3))} for kt, ms in enumerate(res.solve_ms)],
    "dispatch": "longest first by the previous time step's evaluation counts" if ev.dispatch_by_history and B >= ev.dispatch_min_batch else "index order",
    "complete_rate": float(res.complete.mean()), "collision_rate": float((res.collision & ~res.complete).mean()),
    "mean_steps": float(res.steps.mean())}))
