#!/usr/bin/env python3
"""configs[1] dimensions, latency kernel (W = 4 / 3): kernel time against the batch size -- where the time of B = 1024 goes
(768 workgroups of four wavefronts are resident at 168 registers; the rest waits for a slot).   usage: exp_cfg1_batch.py [W ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

class A: gpus = 1
env = bench.Env(A())
for W in [int(a) for a in sys.argv[1:]] or (4, 3):
    for B in (64, 256, 512, 1024, 2048):
        for staged in (0, -1):
            r = bench.run_workload(env, "cfg1", "toward_robot", "f32", 6, 2, batch=B, latency_waves=W, staged=staged)
            print(json.dumps({"W": W, "B": B, "staged": staged, "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2)}), flush=True)
