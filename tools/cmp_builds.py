#!/usr/bin/env python3
"""Diagnostic: evaluations and full solves of configs[1] / configs[2] batches, bit for bit, between the shipped library and
another build of the same ABI (build/libnmpc_prev.so -- e.g. the previous commit compiled next to it). usage: cmp_builds.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    if sys.argv[1] != "cur": os.environ["NMPC_HIP_LIBRARY"] = os.path.join(ROOT, "build", "libnmpc_%s.so" % sys.argv[1])
    import numpy as np
    import dyobav_mpcnwta_warehouse_amd as nm
    out = {}
    for key, B in (("cfg1_b1024_n20_2x5", 1024), ("cfg2_b65536_n20_4x10", 16384)):
        spec = dict(nm.scenarios.BENCH_CONFIGS[key]); L = spec.pop("layout"); spec["B"] = B
        P = nm.scenarios.make_batch(layout=L, **spec).astype(np.float32)
        rng = np.random.default_rng(1)
        U = np.stack([rng.uniform(0.3, 1.4, (B, L.N)), rng.uniform(-0.3, 0.3, (B, L.N))], axis=2).reshape(B, -1)
        Y, C = rng.normal(size=(B, 2 * L.N)), rng.uniform(1, 300, B)
        for lw in (0,):
            cfg = nm.default_config_struct(); cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = L.N, L.Nother, L.Nstc, L.Ndyn
            cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]; cfg.latency_waves = lw; cfg.coop_waves = 1
            with nm.Handle(cfg) as h:
                r = h.eval(P, U, Y, C, dtype=np.float32)
                s = h.solve(P, dtype=np.float32)
            out[f"{key}/lw{lw}"] = [r["psi"].astype(np.float32).tobytes().hex()[:4096], r["grad"].astype(np.float32).tobytes().hex()[:4096],
                                   float(np.abs(s["U"]).sum()), int(s["iters"][:, 1].sum()), s["U"].astype(np.float32).tobytes().hex()]
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cmp_%s.json" % sys.argv[1]), "w"))
else:
    for v in ("cur", "prev"):
        subprocess.run([sys.executable, __file__, v], check=True)
    a, b = (json.load(open(os.path.join(ROOT, "gpurun_out", "cmp_%s.json" % v))) for v in ("cur", "prev"))
    for k in a:
        print(k, "psi equal", a[k][0] == b[k][0], "grad equal", a[k][1] == b[k][1], "solve checksums", a[k][2], b[k][2], "iters", a[k][3], b[k][3], "U equal", a[k][4] == b[k][4], "first differing instance", next((i for i in range(len(a[k][4]) // 320) if a[k][4][320*i:320*i+320] != b[k][4][320*i:320*i+320]), None))
