#!/usr/bin/env python3
"""Which quantity known after the pilot (one outer iteration) predicts the remaining work best? List-scheduler replay of the
second launch for several ranking keys, from a tools/dump_outer_profile.py profile.   usage: exp_rank_key.py <profile.npz> [slots]"""
import heapq
import sys

import numpy as np

d = np.load(sys.argv[1])
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ev = d["evals"].astype(np.int64)
f2 = d["f2"].astype(np.float64)
B = ev.shape[0]
done1 = d["outer"][:, -1] <= 1 if "outer" in d else np.zeros(B, bool)
rest = ev[:, -1] - ev[:, 0]
rest[d["status"][:, 0] == 0] = 0 if (d["status"][:, 0] == 0).any() else rest[d["status"][:, 0] == 0]


def replay(work, order=None):
    w = work if order is None else work[order]
    w = w[w > 0]
    if len(w) <= slots:
        return float(w.max()) if len(w) else 0.0
    h = list(w[:slots].astype(float))
    heapq.heapify(h)
    for x in w[slots:]:
        heapq.heappush(h, heapq.heappop(h) + float(x))
    return max(h)


tau = float(d["ms"][-1]) / replay(ev[:, -1])
pilot = replay(ev[:, 0]) * tau
print(f"B={B}: one launch {float(d['ms'][-1]):.1f} ms; pilot {pilot:.1f} ms; second launch: work/capacity {rest.sum() / slots * tau:.1f} ms, longest {rest.max() * tau:.1f} ms")
f2b = (np.frombuffer(f2[:, 0].astype(np.float32).tobytes(), np.uint32) >> 21).astype(np.int64)    # top 10 bits of the float, as the device sorts
keys = {"index order": None, "perfect (true remaining work)": -rest, "||F2|| (10-bit buckets, what the device does)": -f2b, "||F2|| exact": -f2[:, 0],
        "evaluations of the pilot": -ev[:, 0], "||F2|| buckets, then pilot evaluations": -(f2b * 100000 + ev[:, 0]),
        "pilot evaluations, then ||F2||": -(ev[:, 0] * 4096 + f2b)}
for name, k in keys.items():
    order = None if k is None else np.argsort(k, kind="stable")
    t = replay(rest, order) * tau
    print(f"  second launch ranked by {name:48s}: {t:8.1f} ms  (call {pilot + t:8.1f} ms)")
