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
    print(json.dumps({"lib": os.path.basename(os.environ.get("NMPC_HIP_LIBRARY", "default")), "row": "cfg1 " + fam, "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2)}), flush=True)
''' % ROOT
for rnd in range(3):
    for lib in sys.argv[1:3]:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, NMPC_HIP_LIBRARY=os.path.abspath(lib)), capture_output=True, text=True)
        sys.stdout.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("{")) or out.stderr[-500:]); sys.stdout.flush()
