#!/usr/bin/env python3
"""Round 6 experiment behind the tail hand-off: how fast do the kernel families solve ONE long instance on an otherwise idle
chip? The longest instance of the configs[2] `passing` batch (and of the configs[1]-dimension batch), 64 copies of it:
throughput kernel, latency kernel (flat evaluation) with W = 2 / 4, and the TAIL member (latency kernel with the throughput
kernels' gated evaluation, W = 4: 16 copies handed over next to 48 copies of the batch's SHORTEST instance on the throughput kernel).
One JSON line per measurement."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm

for key, hint in (("cfg2_b65536_n20_4x10", 40), ("cfg1_b1024_n20_2x5", 10)):
    spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")
    P = nm.scenarios.make_batch_chunked(16384, lay, ped_mode="passing", dtype=np.float32, **spec)

    def cfg_(**ov):
        cfg = nm.default_config_struct()
        cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
        cfg.max_active_dynobs = hint
        cfg.axis_aligned = 1
        for k, v in ov.items():
            setattr(cfg, k, v)
        return cfg
    with nm.Handle(cfg_(latency_waves=1)) as h:
        r = h.solve(P)
    i = int(np.argmax(r["info"][:, 4]))
    ev = int(r["info"][i, 4])
    P1 = np.ascontiguousarray(np.repeat(P[i:i + 1], 64, axis=0))
    j = int(np.argmin(r["info"][:, 4]))       # the shortest instance: filler for the launch that times the tail member alone
    P2 = np.ascontiguousarray(np.concatenate([np.repeat(P[i:i + 1], 16, axis=0), np.repeat(P[j:j + 1], 48, axis=0)]))
    for name, ov, order in (("throughput kernel, one wavefront", dict(latency_waves=1, tail_latency=-1, staged=-1), False),
                            ("latency kernel W = 2 (flat evaluation)", dict(latency_waves=2, staged=-1), False),
                            ("latency kernel W = 4 (flat evaluation)", dict(latency_waves=4, staged=-1), False),
                            ("tail member W = 4 (gated evaluation): 16 copies handed over, the other 48 rows a short instance", dict(latency_waves=1, tail_latency=16, staged=-1), True)):
        with nm.Handle(cfg_(**ov)) as h:
            if order:
                h.set_dispatch_order(np.arange(64, dtype=np.int32))
            ms = []
            for _ in range(3):
                rr = h.solve(P2 if order else P1)
                ms.append(h.last_kernel_ms())
            li = h.last_launch_info()
        print(json.dumps({"dims": key, "instance": i, "psi_evals": ev, "kernel": name, "kernel_ms": round(float(np.median(ms)), 2),
                          "us_per_evaluation": round(float(np.median(ms)) * 1e3 / ev, 3), "tail_handed_off": li["tail_handed_off"],
                          "same_controls_as_throughput": bool(np.array_equal(rr["U"][0], r["U"][i]))}), flush=True)
