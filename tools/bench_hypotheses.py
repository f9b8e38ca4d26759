#!/usr/bin/env python3
"""Measurement of the f2 kernel (hypotheses -> ellipses): one JSON line with its HBM roofline (algorithmic bytes =
hypotheses read once + obstacle rows written once)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dyobav_mpcnwta_warehouse_amd as nm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, K, N = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (2, 10, 20)   # pedestrians, hypotheses each, horizon
Ndyn = int(sys.argv[6]) if len(sys.argv) > 6 else 15
cfg = nm.default_config_struct(); cfg.N_hor, cfg.Ndynobs = N, Ndyn
h = nm.Handle(cfg); h.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(0)
cur = torch.rand(B, H, 2, generator=g, device="cuda") * 8 - 4
hyp = (cur[:, None, :, None, :] + torch.randn(B, N, H, K, 2, generator=g, device="cuda") * 0.6).reshape(B, N, H * K, 2).contiguous()
dyn = torch.empty(B, Ndyn, N + 1, 6, device="cuda"); nobs = torch.empty(B, dtype=torch.int32, device="cuda")
step = lambda: h.hypotheses_to_ellipses(np.float32, hyp, cur, dyn, nobs)
for _ in range(3): step()
torch.cuda.synchronize(); ms = []; t0 = time.perf_counter()
for _ in range(steps):
    step(); ms.append(h.last_kernel_ms())
torch.cuda.synchronize(); el = time.perf_counter() - t0
k_ms = float(np.mean(ms)); alg = 4 * (hyp.numel() + cur.numel() + dyn.numel() + B)
print(json.dumps({"metric": "hypothesis sets clustered/sec (f2, device-side)", "value": B * steps / el, "unit": "instances/s",
                  "n_gpus": 1, "steps": steps, "dtype": "f32", "config": {"workload": f"B={B}, N={N}, {H} pedestrians x {K} hypotheses"},
                  "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": alg / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel": "hypotheses_kernel" if H * K <= 64 else "hypotheses_wide_kernel",
                               "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg},
                  "mean_clusters": float(nobs.float().mean())}))
