import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); B = spec.pop("B")
P = np.ascontiguousarray(nm.scenarios.make_batch(B, lay, dtype=np.float32, **spec))
cfg = nm.default_config_struct()          # all 15 rows of the shipped yaml provisioned, no capacity hint
with nm.Handle(cfg) as h:
    U = np.empty((B, 40), np.float32)
    for _ in range(4):
        h.solve_raw(np.float32, P, B, U)
    print(h.last_launch_info(), h.last_kernel_ms())
