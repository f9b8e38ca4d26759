import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); B = spec.pop("B")
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode="toward_robot", dtype=np.float32, **spec)
hs = {}
for ax in (0, 1, -1):
    cfg = nm.default_config_struct(); cfg.max_active_dynobs = 10; cfg.axis_aligned = ax
    hs[ax] = nm.Handle(cfg)
U = np.empty((B, 40), np.float32)
res = {ax: [] for ax in hs}
for rep in range(8):
    for ax, h in hs.items():
        h.solve_raw(np.float32, P, B, U); res[ax].append(h.last_kernel_ms())
for ax in hs: print("axis_aligned", ax, "min %.2f median %.2f max %.2f ms" % (min(res[ax][1:]), np.median(res[ax][1:]), max(res[ax][1:])))
