import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
class A: gpus = 1
env = bench.Env(A())
for rep in range(3):
    for kw in ({}, {"polish": True}):
        r = bench.run_workload(env, "cfg2", "refscen", "f32", 2, 1 if rep == 0 else 0, **kw)
        print(json.dumps({"row": f"cfg2 refscen {kw}", "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2), "polish": r["polish"]}), flush=True)
