#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary -> time of the solve kernels PER STEP. A step (one nmpc_solve_batch call)
may consist of several solve-kernel dispatches since round 3 (axis-aligned kernel + its general twin, which returns at
once; pilot + second launch of the resumable solve), so the per-kernel averages of the stats file are per dispatch;
bench.py's roofline.kernel_ms is the HIP-event time around the whole sequence of one call and must agree with the sum
printed here.   kernel_stats_per_step.py <kernel_stats.csv> <steps incl. warm-up>"""
import sys
import pandas as pd
t = pd.read_csv(sys.argv[1]); steps = int(sys.argv[2])
name = "Name" if "Name" in t.columns else t.columns[0]
tot = 0.0
for _, r in t.iterrows():
    n = str(r[name])
    if "solve_" in n or "rank_" in n or "axis_scan" in n or "polish_" in n:
        ms = float(r["TotalDurationNs"]) / 1e6
        tot += ms
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("nmpc::", "")
        short = short[:short.rfind("(")] if short.endswith(")") else short
        print(f"{short[-70:]:72s} calls {int(r['Calls']):4d}  total {ms:10.3f} ms  avg {float(r['AverageNs']) / 1e6:10.3f} ms  per step {ms / steps:10.3f} ms")
print(f"all kernels of a solve call, per step ({steps} steps): {tot / steps:.3f} ms")
