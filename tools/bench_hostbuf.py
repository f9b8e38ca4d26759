#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: the same configs[2] batch handed over as HOST buffers (numpy in, numpy
out through nmpc_solve_batch_f32: staged H2D, solved, results copied back), next to the device-resident rate."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
L = nm.scenarios.BENCH_CONFIGS["cfg2_b65536_n20_4x10"]["layout"]
B = 65536
P = nm.scenarios.make_batch_chunked(B, L, seed=1, n_ped=4, n_hyp=10)
cfg = nm.default_config_struct()
cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs, cfg.max_active_dynobs = L.N, L.Nother, L.Nstc, L.Ndyn, 40
with nm.Handle(cfg) as h:
    h.solve(P[:1024])
    t0 = time.perf_counter(); r = h.solve(P); wall = time.perf_counter() - t0
    k_ms = h.last_kernel_ms()
print(json.dumps({"metric": "MPC solves/sec, host buffers in and out (PCIe inclusive)", "value": B / wall, "wall_s": wall,
                  "kernel_ms": k_ms, "bytes_in": int(P.nbytes), "kernel_only_value": B / (k_ms * 1e-3)}))
