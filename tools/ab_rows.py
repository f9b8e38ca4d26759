#!/usr/bin/env python3
"""Same-box A/B of two builds of the library (NMPC_HIP_LIBRARY) on a few bench rows, interleaved.
usage: ab_rows.py <libA.so> <libB.so> [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys
sys.path.insert(0, %r)
import bench
class A: gpus = 1
env = bench.Env(A())
for wl, fam, st in (("cfg2", "toward_robot", 3), ("cfg2", "passing", 3), ("cfg1", "toward_robot", 8)):
    r = bench.run_workload(env, wl, fam, "f32", st, 1)
    print(json.dumps({"lib": os.path.basename(os.environ.get("NMPC_HIP_LIBRARY", "default")), "row": wl + " " + fam, "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2)}), flush=True)
''' % ROOT
libs = [os.path.abspath(a) for a in sys.argv[1:3]]
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, NMPC_HIP_LIBRARY=lib), capture_output=True, text=True)
        sys.stdout.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("{")) or out.stderr[-800:])
        sys.stdout.flush()
