#!/usr/bin/env python3
"""Diagnostic: latency of ONE instance (the slowest of the cfg1 batch) alone on the chip, per solver kernel / wavefront
count / kernel variant. cycles/iter at 2.4 GHz; rounds = exchange rounds of the latency kernel (csrc/nmpc_spec.h).
From two wavefront counts: cycles/iter = rounds/iter * R + S gives the cost R of a round and S of the solver logic."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
P = nm.scenarios.make_batch(1024, seed=0).astype(np.float32)
with nm.Handle(nm.default_config_struct()) as h0:
    top = int(np.argmax(h0.solve(P)["iters"][:, 1]))
for axis in (0, -1):
  for w in (1, -1, 2, 3, 4, 6, 8):
    cfg = nm.default_config_struct(); cfg.max_active_dynobs = 10; cfg.latency_waves = w; cfg.axis_aligned = axis
    h = nm.Handle(cfg)
    for _ in range(2):
        o1 = h.solve(P[top:top+1], dtype=np.float32)
    ms = h.last_kernel_ms(); it = int(o1["iters"][0, 1]); r = o1['info'][0,6]
    print(f"instance {top} axis_aligned={axis} w={w}: {ms:.2f} ms, inner {it}, evals {o1['info'][0,4]:.0f}, rounds/iter {r/it:.2f}, cycles/iter {ms*1e-3*2.4e9/it:.0f}")
    h.close()
