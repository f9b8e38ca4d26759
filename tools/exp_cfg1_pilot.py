#!/usr/bin/env python3
"""configs[1]: wavefronts per instance of the PILOT launch of the resumable solve (NMPC_PILOT_WAVES, diagnostic knob) with
W = 4 behind it. usage: exp_cfg1_pilot.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys
sys.path.insert(0, %r)
import bench
class A: gpus = 1
env = bench.Env(A())
for fam in ("toward_robot", "passing"):
    r = bench.run_workload(env, "cfg1", fam, "f32", 10, 2)
    print(json.dumps({"pilot_waves": os.environ.get("NMPC_PILOT_WAVES", "4"), "family": fam, "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2)}), flush=True)
''' % ROOT
for w in ("", "3", "2"):
    env = dict(os.environ)
    if w:
        env["NMPC_PILOT_WAVES"] = w
    else:
        env.pop("NMPC_PILOT_WAVES", None)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    sys.stdout.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("{")))
    sys.stdout.flush()
