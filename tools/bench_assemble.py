#!/usr/bin/env python3
"""Measurement of the f1 kernels (device-side parameter assembly): one JSON line with the HBM roofline of
select_static_kernel + fill_kernel (algorithmic bytes = 2 * w * np per instance: every element of P is produced
from one source element and written once)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dyobav_mpcnwta_warehouse_amd as nm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65535
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt, tdt = np.float32, torch.float32
cfg = nm.default_config_struct()
h = nm.Handle(cfg)
h.set_stream(torch.cuda.current_stream().cuda_stream)
N, M, n_dyn = 20, 64, 10
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.rand(*s, generator=g, device="cuda", dtype=tdt)
state, last_u, refs, speed = r(B, 3) * 10 - 5, r(B, 2), r(B, N, 3), r(B)
tuning, stcw, dynw = r(10), r(N), r(N)
ctr = r(M, 1, 2) * 16 - 8
polys = (ctr + torch.tensor([[1, 1], [-1, 1], [-1, -1], [1, -1]], device="cuda", dtype=tdt) * 0.7).contiguous()
dyn = r(B, n_dyn, N + 1, 6)
P = torch.empty(B, h.np_, dtype=tdt, device="cuda")
def step():
    h.assemble_params(dt, B, P, last_u, state, refs, speed, tuning, stcw, dynw, polys, dyn)
for _ in range(3):
    step()
torch.cuda.synchronize()
ms = []
t0 = time.perf_counter()
for _ in range(steps):
    step(); ms.append(h.last_kernel_ms())
torch.cuda.synchronize()
el = time.perf_counter() - t0
k_ms = float(np.mean(ms))
alg = 2 * 4 * h.np_ * B
print(json.dumps({"metric": "parameter vectors assembled/sec (f1, device-side)", "value": B * steps / el, "unit": "vectors/s",
                  "n_gpus": 1, "steps": steps, "dtype": "f32", "config": {"workload": f"B={B}, N=20, np={h.np_}, M={M} map polygons, {n_dyn} obstacles"},
                  "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": alg / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                               "kernel": "assemble_kernel (selection wavefront + byte-mover wavefronts, fused)", "kernel_ms": k_ms,
                               "algorithmic_bytes_per_launch": alg}}))
