import json, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class A: gpus = 1
env = bench.Env(A())
for kw in ({}, {"polish": True}):
    for wl in ("cfg2", "cfg1"):
        r = bench.run_workload(env, wl, "passing", "f32", 3, 1, **kw)
        print(json.dumps({"row": f"{wl} passing {kw}", "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2), "polish": r["polish"]}), flush=True)
