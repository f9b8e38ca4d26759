#!/usr/bin/env python3
"""A/B of two library builds on configs[1]'s dimensions WITHOUT the capacity hint (all 15 obstacle rows of the shipped yaml
provisioned): B = 1024 (latency kernel) and B = 65536 (throughput kernel).  usage: ab_nohint.py <libA> <libB>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); spec.pop("B")
for B in (1024, 65536):
    for fam in ("toward_robot", "passing"):
        P = np.ascontiguousarray(nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec))
        cfg = nm.default_config_struct()
        cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
        cfg.max_active_dynobs = 0
        with nm.Handle(cfg) as h:
            U = np.empty((B, 2 * lay.N), np.float32); st = np.empty(B, np.int32)
            ms = []
            for _ in range(4):
                h.solve_raw(np.float32, P, B, U, status=st); ms.append(h.last_kernel_ms())
            k = float(np.mean(ms[1:]))
            print(json.dumps({"lib": os.path.basename(os.environ.get("NMPC_HIP_LIBRARY", "default")), "B": B, "family": fam, "info": h.last_launch_info(),
                              "solves_per_s": round(B / k * 1e3), "kernel_ms": round(k, 2), "converged": float(np.mean(st == 0)), "checksum": float(np.abs(U).sum())}), flush=True)
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:3]:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, NMPC_HIP_LIBRARY=os.path.abspath(lib)), capture_output=True, text=True)
        sys.stdout.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("{")) or out.stderr[-800:]); sys.stdout.flush()
