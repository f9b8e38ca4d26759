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
    assert len(lines0[0]) < 4096, len(lines0[0])          # the driver must be able to parse the line (round 3: 21.7 KB, unparsed)
    rec = json.loads(lines0[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["unit"] == "solves/s"
    assert not {"secondary", "secondary_solves_per_s", "cpu_baseline", "accuracy", "accuracy_summary"} & set(rec)   # N > 1: the headline only
    # MAX over ranks of the elapsed time: rank 1 sleeps 40 ms per step, rank 0 20 ms
    assert rec["ms_per_step"] >= 40.0 and abs(rec["value"] - 2 * 48 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"] + 1e-3
    assert rec["roofline"]["valu_frac"] > 0 and rec["roofline"]["psi_evals_per_solve"] == 7.0
    # SCALE day-one diagnostics: did the process group see N ranks, every rank's own time per step
    # (imbalance between shards), its kernel time, and what the gather costs -- at the top level of the ONE line
    assert rec["ranks"] == 2 and len(rec["per_rank_ms"]) == 2 and len(rec["per_rank_kernel_ms"]) == 2
    # (the gather inside a step makes the fast rank wait for the slow one: the ranks' wall times agree; the imbalance between
    #  the shards shows in their KERNEL times -- rank 1 is the slow one, 40 ms per step against 20)
    assert all(abs(v - rec["ms_per_step"]) < 0.2 * rec["ms_per_step"] for v in rec["per_rank_ms"])
    assert rec["per_rank_kernel_ms"] == [20.0, 40.0]
    assert rec["gather_ms"] >= 0 and rec["gather_bytes_per_rank"] == 48 * 40 * 4
    r0, r1 = (np.load(tmp_path / f"r{r}.npz") for r in range(2))
    assert float(r0["chk"]) != float(r1["chk"])                      # every rank solved its own shard (seed + rank)
    for r in (r0, r1):                                              # gathered = [rank 0's controls; rank 1's], on every rank
        assert r["gathered"].shape == (2 * 48, 40)
        assert np.array_equal(r["gathered"][:48], r0["U"]) and np.array_equal(r["gathered"][48:], r1["U"])


def test_bench_line_is_compact_with_every_optional_part():
    """compact_line() on a full N = 1 record (headline + 13 secondary rows + CPU baseline + the six-row accuracy table:
    round 3's own 21.7 KB record, which the driver could not parse): one strict-JSON line below 4 KB that still carries
    `roofline`, `cpu_baseline` and the flat digests; 8 KB is a hard error."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    detail = json.load(open(os.path.join(ROOT, "profiles", "r03_cfg2_bench.json")))
    for r in detail["secondary"]:
        r["batch_override"] = "at the batch size" in (r.get("note") or "") or "quarter" in (r.get("note") or "")
    line = bench.compact_line(detail)
    assert "\n" not in line and len(line) < bench.LINE_TARGET_BYTES, len(line)
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline",
              "cpu_baseline", "converged_frac", "secondary_solves_per_s", "accuracy_summary"):
        assert k in rec, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "valu_frac"} <= set(rec["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(rec["cpu_baseline"])
    assert len(rec["secondary_solves_per_s"]) == len(detail["secondary"])          # no two rows share a name
    assert all(isinstance(v, (int, float)) for v in rec["secondary_solves_per_s"].values())
    assert abs(rec["value"] - detail["value"]) < 1e-4 * detail["value"]
    # ... and on round 5's record (nested accuracy digest, closed-loop rows, the tight-tolerance KKT keys)
    d5 = json.load(open(os.path.join(ROOT, "profiles", "r05_cfg2_bench_detail.json")))
    line5 = bench.compact_line(d5)
    assert "\n" not in line5 and len(line5) < bench.LINE_TARGET_BYTES, len(line5)
    rec5 = json.loads(line5)
    assert rec5["config"]["max_active_dynobs"] == 40                      # the capacity hint is stated in the line
    assert {"cfg2_closed_loop_f32", "cfg1_f32_nohint", "cfg4_f64"} <= set(rec5["secondary_solves_per_s"])
    assert rec5["closed_loop"]["solves_per_s"] == rec5["secondary_solves_per_s"]["cfg2_closed_loop_f32"]
    assert 0 < rec5["closed_loop"]["converged_frac"] < 1 and rec5["closed_loop"]["psi_evals_per_solve"] > 0
    acc = rec5["accuracy_summary"]
    assert {"cfg2_passing", "cfg2_closed_loop", "cfg1_passing"} <= set(acc)
    for k in ("cfg2_passing", "cfg2_closed_loop", "cfg1_passing"):
        assert acc[k]["tight_hip_orc"]["unexpl"] == 0 and acc[k]["tight_orc_twin"]["unexpl"] == 0 and acc[k]["audit"]["unexpl"] == 0
        assert acc[k]["tight_hip_orc"]["kkt_max_du"] < 1e-4            # both end points stationary -> inside the north star's bar
    # ... and on round 6's record: the reference-scenario row with its budgeted twin, the evaluation-budget leg of the CPU baseline
    d6 = json.load(open(os.path.join(ROOT, "profiles", "r06_cfg2_bench_detail.json")))
    line6 = bench.compact_line(d6)
    assert "\n" not in line6 and len(line6) < bench.LINE_TARGET_BYTES, len(line6)
    rec6 = json.loads(line6)
    assert {"cfg2_refscen_f32", "cfg2_refscen_f32_budget", "cfg2_closed_loop_f32", "cfg2_closed_loop_f32_budget", "cfg2_f32_budget"} <= set(rec6["secondary_solves_per_s"])
    rs = rec6["reference_scenarios"]
    assert rs["solves_per_s"] == rec6["secondary_solves_per_s"]["cfg2_refscen_f32"] and rs["capture_steps"] == [2, 14, 26]
    assert rs["budget"]["max_evaluations"] == 2526 and 0 < rs["budget"]["out_of_time_frac"] < 1
    assert rs["budget"]["solves_per_s"] == rec6["secondary_solves_per_s"]["cfg2_refscen_f32_budget"]
    assert rec6["value"] < 1e5 < rs["solves_per_s"]                       # which side of the north star's 1e5 each falls on
    assert rec6["cpu_baseline"]["evals_per_s_per_core"] > 0 and rec6["cpu_baseline"]["value_with_evaluation_budget"] > rec6["cpu_baseline"]["value"]
    assert rec6["accuracy_summary"]["cfg2_refscen"]["audit"]["unexpl"] == 0 and rec6["accuracy_summary"]["cfg2_refscen"]["tight_hip_orc"]["unexpl"] == 0
    assert rec6["config"]["tail_handed_off"] == 256
    detail["config"]["padding"] = "x" * 9000
    detail2 = dict(detail, config=dict(detail["config"], workload="y" * 9000))
    with pytest.raises(RuntimeError):
        bench.compact_line(detail2)
