#!/usr/bin/env python3
"""configs[1] (B = 1024, latency kernel): wavefronts per instance x staging, index order. Results do not depend on W
(bit for bit, tested), so this is purely a scheduling question.   usage: exp_cfg1_waves.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

class A: gpus = 1
env = bench.Env(A())
for fam in ("toward_robot", "passing"):
    for W in (2, 3, 4):
        for staged in (0, -1):
            r = bench.run_workload(env, "cfg1", fam, "f32", 8, 2, latency_waves=W, staged=staged)
            print(json.dumps({"family": fam, "W": W, "staged": staged, "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2),
                              "kernel": r["roofline"]["kernel"]}), flush=True)
