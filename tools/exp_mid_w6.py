import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); spec.pop("B")
for B in (64, 256):
    P = np.ascontiguousarray(nm.scenarios.make_batch(B, lay, dtype=np.float32, **spec))
    for hint in (10, 0):
        for lw in (0, 4, 6):
            cfg = nm.default_config_struct(); cfg.max_active_dynobs = hint; cfg.latency_waves = lw
            with nm.Handle(cfg) as h:
                U = np.empty((B, 40), np.float32); info = np.empty((B, 8), np.float32); ms = []
                for _ in range(6):
                    h.solve_raw(np.float32, P, B, U, info=info); ms.append(h.last_kernel_ms())
                print(f"B={B} hint={hint} latency_waves={lw}: W={int(info[0,7])} {np.mean(ms[2:]):.2f} ms  checksum {float(np.abs(U).sum()):.4f}", flush=True)
