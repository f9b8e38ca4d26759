#!/usr/bin/env python3
"""Round 6: batches of one to four device fills (2 048 .. 8 192 instances at configs[2]'s dimensions) -- what a closed-loop
evaluation sends once most of its scenarios have finished. The resumable solve (pilot + ranking) and the tail hand-off are
switched on from four fills on; does either pay below? Needs a development build (-DNMPC_DEV_ENV: thresholds from the
environment, in device fills), e.g. build/libnmpc_devenv.so through NMPC_HIP_LIBRARY.
   usage: exp_mid_batches.py [family ...]      one JSON line per (family, B, order, thresholds)   env: SIZES, FILLS (stage/tail,...), LW, DIMS=cfg1|cfg2, BI (batch_invariant), TAIL (tail_latency), STAGED (staged)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
fam, B, order = sys.argv[1], int(sys.argv[2]), sys.argv[3]
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10"}[os.environ.get("DIMS", "cfg2")]
spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")
cfg = nm.default_config_struct()
cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
if fam in ("refscen", "corridor"):
    steps, hf = ((2, 14, 26), "reference") if fam == "refscen" else ((1, 8, 20), "corridor")
    P, _ = nm.scenarios.harvest_closed_loop(cfg, B, steps=steps, seed=13, n_ped=spec["n_ped"], n_hyp=spec["n_hyp"], dtype=np.float32, family=hf)
else:
    P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec)
P = np.ascontiguousarray(P, np.float32)
cfg.axis_aligned = 1
cfg.latency_waves = int(os.environ.get("LW", "0"))
cfg.batch_invariant = int(os.environ.get("BI", "0"))
cfg.tail_latency = int(os.environ.get("TAIL", "0"))
cfg.staged = int(os.environ.get("STAGED", "0"))
U = np.empty((B, 2 * lay.N), np.float32); st = np.empty(B, np.int32); info = np.empty((B, 8), np.float32)
with nm.Handle(cfg) as h:
    h.solve_raw(np.float32, P, B, U, status=st, info=info)
    if order == "lpt":
        h.set_dispatch_order(np.argsort(-info[:, 4], kind="stable").astype(np.int32))
    ms = []
    for _ in range(4):
        h.solve_raw(np.float32, P, B, U, status=st, info=info)
        ms.append(h.last_kernel_ms())
    li = h.last_launch_info()
print(json.dumps({"kernel_ms": round(float(np.mean(ms[1:])), 2), "family": li["family"], "staged": li["staged_outer_iterations"],
                  "tail": li["tail_handed_off"], "checksum": float(np.abs(U).sum()), "longest_evals": int(info[:, 4].max())}))
''' % ROOT
SIZES = [int(x) for x in os.environ.get("SIZES", "2500,4096,6000,8000").split(",")]
FILLS = [tuple(float(y) for y in x.split("/")) for x in os.environ.get("FILLS", "4/4,1/1,1.5/1.5,2/2,4/1,1/4").split(",")]
for fam in sys.argv[1:] or ["passing", "refscen"]:
    for B in SIZES:
        for order in ("index", "lpt"):
            for fills in FILLS:
                env = dict(os.environ, NMPC_STAGE_FILLS=str(fills[0]), NMPC_TAIL_FILLS=str(fills[1]))
                r = subprocess.run([sys.executable, "-c", CODE, fam, str(B), order], env=env, capture_output=True, text=True)
                if r.returncode != 0:
                    print(r.stderr[-800:]); continue
                row = json.loads(r.stdout.strip().splitlines()[-1])
                print(json.dumps({"family_of_instances": fam, "B": B, "order": order, "stage_fills": fills[0], "tail_fills": fills[1], "latency_waves": int(os.environ.get("LW", "0")), "dims": os.environ.get("DIMS", "cfg2"), **row}), flush=True)
