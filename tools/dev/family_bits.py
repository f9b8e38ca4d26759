#!/usr/bin/env python3
"""Do the throughput and the latency kernel FAMILY of one build give the same bits? (handing the tail of a
throughput launch to the latency kernel is only clean if they do.)
usage: family_bits.py <lib.so> [...]   -- per library: configs[1] / configs[2] dimensions, two scenario families, fp32 (and
fp64 at configs[1]): full solves through the throughput kernel (latency_waves = 1), the latency kernel's own code with one
wavefront (-1) and with four (4); compares controls, statuses and iteration counts instance by instance."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
for wl, key, B in (("cfg1", "cfg1_b1024_n20_2x5", 512), ("cfg2", "cfg2_b65536_n20_4x10", 512)):
    for fam in ("toward_robot", "passing"):
        spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")
        P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float64, **spec)
        for dt in ((np.float32, np.float64) if wl == "cfg1" else (np.float32,)):
            res = {}
            for lw in (1, -1, 4):
                cfg = nm.default_config_struct()
                cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
                cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
                cfg.latency_waves, cfg.staged, cfg.axis_aligned = lw, -1, 1
                with nm.Handle(cfg) as h:
                    res[lw] = h.solve(P.astype(dt))
            for a, b in ((1, -1), (-1, 4)):
                same_u = np.all(res[a]["U"] == res[b]["U"], axis=1)
                same_it = np.all(res[a]["iters"] == res[b]["iters"], axis=1)
                du = np.abs(res[a]["U"] - res[b]["U"]).max(axis=1)
                print(f"{wl} {fam:12s} {np.dtype(dt).name}  lw {a:2d} vs {b:2d}: identical controls {same_u.mean():6.1%%}  identical iteration counts "
                      f"{same_it.mean():6.1%%}  median |du| {np.median(du):.1e}  (converged {np.mean(res[a]['status'] == 0):.2f})")
''' % ROOT
for lib in sys.argv[1:]:
    print("==", lib)
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, NMPC_HIP_LIBRARY=os.path.abspath(lib)), capture_output=True, text=True)
    sys.stdout.write(r.stdout if r.returncode == 0 else r.stderr[-1500:])
