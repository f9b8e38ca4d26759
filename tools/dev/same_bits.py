#!/usr/bin/env python3
"""Do two builds of the library give the same bits? (kernel changes that only move work around must)
usage: same_bits.py <libA.so> <libB.so>   -- configs[1] / [2] / [4] dimensions, both families, fp32 + fp64, small batches."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
out = {}
for wl, key, B in (("cfg1", "cfg1_b1024_n20_2x5", 256), ("cfg2", "cfg2_b65536_n20_4x10", 4096), ("cfg4", "cfg4_b8192_n40_8x20", 64)):
    for fam in ("toward_robot", "passing"):
        spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")
        P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float64, **spec)
        for dt in (np.float32, np.float64):
            if dt is np.float64 and wl != "cfg1": P2 = P[:max(B // 8, 16)]
            else: P2 = P
            for lw in ((0, 1) if wl == "cfg1" else (0,)):
                cfg = nm.default_config_struct()
                cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
                cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
                cfg.latency_waves = lw
                with nm.Handle(cfg) as h:
                    r = h.solve(P2.astype(dt))
                out[f"{wl}_{fam}_{np.dtype(dt).name}_lw{lw}"] = np.concatenate([r["U"].ravel().astype(np.float64), r["status"].ravel().astype(np.float64), r["iters"].ravel().astype(np.float64)])
np.savez(sys.argv[1], **out)
''' % ROOT
import numpy as np
files = []
for lib in sys.argv[1:3]:
    f = tempfile.mktemp(suffix=".npz")
    r = subprocess.run([sys.executable, "-c", CODE, f], env=dict(os.environ, NMPC_HIP_LIBRARY=os.path.abspath(lib)), capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-2000:]); sys.exit(1)
    files.append(np.load(f))
bad = 0
for k in files[0].files:
    a, b = files[0][k], files[1][k]
    same = a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    print(f"{k:40s} {'same bits' if same else 'DIFFERENT: max |d| %.3e, %d values' % (np.nanmax(np.abs(a - b)), int((a != b).sum()))}")
    bad += not same
sys.exit(1 if bad else 0)
