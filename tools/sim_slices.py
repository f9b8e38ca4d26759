#!/usr/bin/env python3
"""List-scheduler replay (2048 resident wavefronts, workgroups start in index order) of time slicing by evaluation count:
every unfinished instance runs at most K more evaluations per launch, the last launch to the end.
usage: sim_slices.py <evals.npz with the per-instance totals> [slots]"""
import heapq, sys
import numpy as np
d = np.load(sys.argv[1]); slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
tot = d["evals"].astype(np.int64); B = len(tot)
def replay(w):
    w = w[w > 0]
    if len(w) <= slots: return float(w.max()) if len(w) else 0.0
    h = list(w[:slots].astype(float)); heapq.heapify(h)
    for x in w[slots:]:
        heapq.heappush(h, heapq.heappop(h) + float(x))
    return max(h)
base = replay(tot)
print(f"B={B} mean {tot.mean():.0f} max {tot.max()} evals; one launch (index order) {base:.0f}; work/capacity {tot.sum()/slots:.0f}; "
      f"LPT {replay(tot[np.argsort(-tot)]):.0f}   [unit: evaluations of one wavefront]")
for plan in ([2000]*3, [2000]*5, [2000]*8, [3000]*3, [3000]*5, [4000]*2, [4000]*3, [4000]*4, [1000]*10, [1000, 1000, 2000, 4000], [500, 500, 1000, 2000, 4000, 4000],
             [6000], [8000], [6000, 4000], [8000, 4000], [10000], [12000], [6000, 3000, 3000, 3000]):
    rem = tot.copy(); t = 0.0; n = []
    for K in plan + [None]:
        run = rem if K is None else np.minimum(rem, K)
        n.append(int((rem > 0).sum()))
        t += replay(run); rem = rem - run
    print(f"  slices {plan} + rest: {t:.0f} ({t/base*100:.1f} % of one launch); instances per launch {n}")
