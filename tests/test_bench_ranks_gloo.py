"""world_size-2 run of bench.py's OWN rank logic on a GPU-less box (gloo, CPU tensors, a stand-in for the solver handle):
per-rank seed, the all-gather of the controls into [world * B, 2N], the MAX-reduced elapsed time, exactly one JSON line
and only from rank 0 -- so that the first real 8-GPU run of the driver cannot fail on plumbing. No scaling curve has been
measured on hardware; this test is about correctness of the launch path, not about speed."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time, json
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    import bench

    class StubHandle:
        """what run_workload touches of nm.Handle; the "solve" is a deterministic function of the parameter rows"""
        def __init__(self, cfg):
            self.cfg, self.n = cfg, 2 * cfg.N_hor
        def set_stream(self, s): pass
        def set_dispatch_order(self, o): pass
        def solve_raw(self, dtype, P, B, U, cost, status, iters, u0, y, y_in, c0, info, sync=True):
            time.sleep(0.02 * (1 + int(os.environ["RANK"])))            # rank 1 is the slow one
            U.copy_(P[:, :self.n] * 2 + 1)
            cost.zero_(); status.zero_(); iters.fill_(3); info.zero_(); info[:, 4] = 7; info[:, 5] = 5
        def last_kernel_ms(self): return 20.0 * (1 + int(os.environ["RANK"]))
        def kernel_info(self): return {"lds_bytes_f32": 1, "lds_bytes_f64": 2, "lanes_per_step": 3, "waves_per_cu_f32": 4, "waves_per_cu_f64": 5}
        def last_launch_info(self): return {"family": "throughput", "axis_aligned": -1, "staged_outer_iterations": 0, "polish_selected": 0}
        def close(self): pass

    out = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "48", "--workload", "cfg1"],
                     env_factory=lambda a: bench.Env(a, backend="gloo", device="cpu", handle_factory=StubHandle))
    np.savez(sys.argv[2], U=out["U"], gathered=out["gathered"].numpy(), chk=out["P_checksum"])
''')


def test_bench_rank_logic_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path / f"r{rank}.npz")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    # exactly one JSON line, from rank 0; nothing on rank 1's stdout
    lines0 = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines0) == 1 and outs[1][0].strip() == ""
    rec = json.loads(lines0[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["unit"] == "solves/s"
    assert "secondary" not in rec and "cpu_baseline" not in rec and "accuracy" not in rec     # N > 1: the headline only
    # MAX over ranks of the elapsed time: rank 1 sleeps 40 ms per step, rank 0 20 ms
    assert rec["ms_per_step"] >= 40.0 and abs(rec["value"] - 2 * 48 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]
    assert rec["roofline"]["valu_frac"] > 0 and rec["roofline"]["psi_evals_per_solve"] == 7.0
    r0, r1 = (np.load(tmp_path / f"r{r}.npz") for r in range(2))
    assert float(r0["chk"]) != float(r1["chk"])                      # every rank solved its own shard (seed + rank)
    for r in (r0, r1):                                              # gathered = [rank 0's controls; rank 1's], on every rank
        assert r["gathered"].shape == (2 * 48, 40)
        assert np.array_equal(r["gathered"][:48], r0["U"]) and np.array_equal(r["gathered"][48:], r1["U"])
