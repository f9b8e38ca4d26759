#!/usr/bin/env python3
"""Headline benchmark: batched NMPC solves/sec on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

A "step" is one pass of the hot path over one batch: every rank solves its own shard of B independent MPC
problems (weak scaling: B per GPU is fixed) with the parameter batch already resident in HBM, then (N > 1) the
control sequences are all-gathered over RCCL -- the only collective on this path. Workload at N = 1:
BASELINE.json configs[1] (batch=1024 random init states, N=20, 2 obstacles x 5 WTA hypotheses, fp32).

Rank 0 prints ONE JSON line with the driver's keys plus
  roofline     : HBM roofline of the solve kernel (algorithmic bytes / HIP-event kernel time) -- this path is
                 VALU/latency bound, so the HBM fraction is tiny by construction; the fp32 vector-ALU figure that
                 actually bounds it is reported next to it as roofline.valu
  cpu_baseline : the CPU oracle (plain-C restatement of the reference's OpEn algorithm, kind "port") timed on
                 this box's host cores on the same batch (N = 1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name -> (BASELINE.json description, scenario key)
    "cfg1": ("batch=1024 random init states, N=20, 2 obstacles x 5 WTA hypotheses, fp32, 1 MI355X",
             "cfg1_b1024_n20_2x5"),
    "cfg2": ("batch=65536 main_eva.py scenarios, mpc_fast.yaml N=20, 4 obs x 10 hypotheses, 1 MI355X",
             "cfg2_b65536_n20_4x10"),
    "cfg4": ("long horizon N=40, 8 obs x 20 hypotheses, batch=8192, 1 MI355X (obstacle table streamed from HBM)",
             "cfg4_b8192_n40_8x20"),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector


def measured_traffic(workload, dtype, batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE and
    WRITE_SIZE collected in separate passes, gfx950 correction applied; profiles/rNN_<workload>_traffic.json).
    A counter pass cannot run inside the timed process, so the latest committed measurement that matches the
    workload is reported; None if there is none."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload and rec.get("dtype") == dtype and rec.get("batch") == batch:
            best = rec
    return None if best is None else float(best["hbm_bytes_per_launch"])


def flops_forward(N, Nother, Nstc, Ndyn):
    """Algorithmic flops of one psi evaluation; cost+gradient is counted as 3x."""
    return N * (33 + 8 * (2 * Nother - 1) + 28 * Nstc + 62 * Ndyn) + 10 * N * (N + 1) + 12 * N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg1")
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch (default: BASELINE's)")
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-waves", type=int, default=0,
                    help="nmpc_config.latency_waves: 0 = automatic (library default), 1 = one wavefront per instance, "
                         "4 = latency mode")
    ap.add_argument("--reg-table", type=int, default=0, help="nmpc_config.reg_table: 0 = automatic, -1 = LDS / global table")
    args = ap.parse_args()

    # stdout must carry exactly ONE JSON line: RCCL / HIP libraries print banners and warnings on fd 1, so keep a
    # private copy of the real stdout for the result and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist

    import dyobav_mpcnwta_warehouse_amd as nm

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solver has no CPU path")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ      # under torch.distributed.run the same path runs for N = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world,           # "nccl" is RCCL on ROCm
                                device_id=torch.device("cuda", local_rank))

    desc, key = WORKLOADS[args.workload]
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    t_dtype = torch.float32 if args.dtype == "f32" else torch.float64
    spec = dict(nm.scenarios.BENCH_CONFIGS[key])
    layout = spec.pop("layout")
    B = args.batch or spec.pop("B")
    spec.pop("B", None)
    spec["seed"] = spec["seed"] + rank           # every rank solves a different shard
    P_host = nm.scenarios.make_batch(B, layout, **spec)
    N = layout.N

    cfg = nm.default_config_struct()
    cfg.device_id = local_rank
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = layout.N, layout.Nother, layout.Nstc, layout.Ndyn
    # capacity hint: the workload has n_ped x n_hyp predicted-obstacle hypotheses, the remaining Ndynobs slots are
    # the reference's zero padding (mpc_interface.py:82-88); fewer provisioned rows -> less LDS per instance
    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
    cfg.latency_waves = args.latency_waves
    cfg.reg_table = args.reg_table
    h = nm.Handle(cfg)
    stream = torch.cuda.current_stream()
    h.set_stream(stream.cuda_stream)

    # inputs and outputs resident in HBM before the timed region
    dP = torch.from_numpy(P_host.astype(np_dtype)).cuda()
    dU = torch.empty(B, 2 * N, dtype=t_dtype, device="cuda")
    dcost = torch.empty(B, dtype=t_dtype, device="cuda")
    dstatus = torch.empty(B, dtype=torch.int32, device="cuda")
    diters = torch.empty(B, 2, dtype=torch.int32, device="cuda")
    dinfo = torch.empty(B, 8, dtype=t_dtype, device="cuda")
    gathered = torch.empty(world * B, 2 * N, dtype=t_dtype, device="cuda") if use_dist else None

    kernel_ms = []

    def step(record):
        h.solve_raw(np_dtype, dP, B, dU, dcost, dstatus, diters, None, None, False, None, dinfo, sync=False)
        if use_dist:
            dist.all_gather_into_tensor(gathered, dU)      # RCCL over xGMI: gather the results, nothing else
        if record:
            kernel_ms.append(h.last_kernel_ms())           # HIP events on the launch stream (syncs that stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    status = dstatus.cpu().numpy()
    iters = diters.cpu().numpy()
    info = dinfo.cpu().numpy().astype(np.float64)
    U = dU.cpu().numpy()

    if rank == 0:
        total_solves = world * B * args.steps
        value = total_solves / elapsed
        w = np.dtype(np_dtype).itemsize
        bytes_per_solve = w * (layout.np_ + 2 * N + 4)
        k_ms = float(np.mean(kernel_ms))
        achieved_gbs = bytes_per_solve * B / (k_ms * 1e-3) / 1e9
        ff = flops_forward(layout.N, layout.Nother, layout.Nstc, layout.Ndyn)
        n_psi, n_grad = info[:, 4], info[:, 5]
        waves = int(info[0, 7])          # 0: throughput kernel; > 0: latency kernel with that many wavefronts per instance
        tname = "float" if args.dtype == "f32" else "double"
        lps = h.kernel_info()["lanes_per_step"]
        kernel_name = (f"solve_spec_kernel<{tname}, LPS={lps}> x {waves} wavefronts per instance (latency mode)" if waves
                       else f"solve_kernel<{tname}, LPS={lps}> (one wavefront per instance)")
        flops_launch = float(np.sum((n_psi - n_grad) * ff + n_grad * 3 * ff))
        achieved_tf = flops_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": f"MPC solves/sec (N={layout.N}, batched)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "batch_per_gpu": B, "N_hor": layout.N,
                       "Ndynobs": layout.Ndyn, "Nstcobs": layout.Nstc, "Nother": layout.Nother,
                       "np": layout.np_, "max_active_dynobs": int(cfg.max_active_dynobs), "latency_waves": int(cfg.latency_waves), "sharding": f"{world} x independent shards, all_gather of U"
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": measured_traffic(args.workload, args.dtype, B),
                         "kernel": kernel_name,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_per_solve * B,
                         "valu": {"achieved": achieved_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": achieved_tf / VALU_PEAK_TFLOPS,
                                  "flops_per_psi_eval": ff, "psi_evals_per_solve": float(n_psi.mean()),
                                  "grad_evals_per_solve": float(n_grad.mean())}},
            "solver": {"converged_frac": float(np.mean(status == 0)),
                       "outer_iters_mean": float(iters[:, 0].mean()), "inner_iters_mean": float(iters[:, 1].mean()),
                       "inner_iters_max": int(iters[:, 1].max())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_sample"] = cpu_baseline(layout, P_host, U, status)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()

    h.close()
    if use_dist:
        dist.destroy_process_group()


def usable_cores() -> int:
    """Host cores this process may actually use: CPU affinity capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(layout, P_host, U_gpu, status_gpu):
    """The CPU oracle (kind "port": C restatement of the OpEn algorithm, fp64) on this box's host cores, on the
    timed batch itself (bounded: <= 2048 instances), all cores via OpenMP over instances. Reported, not tuned."""
    import oracle
    pr = oracle.Problem(layout.N, layout.Nother, layout.Nstc, layout.Ndyn)
    cores = usable_cores()
    sample = min(P_host.shape[0], 2048)
    Ps = P_host[:sample]
    oracle.solve_batch(pr, oracle.Options(), Ps[:min(sample, cores)], nthreads=cores)   # warm the threads
    t0 = time.perf_counter()
    Uo, ro = oracle.solve_batch(pr, oracle.Options(), Ps, nthreads=cores)
    t_all = time.perf_counter() - t0
    n1 = min(sample, 32)
    t0 = time.perf_counter()
    oracle.solve_batch(pr, oracle.Options(), Ps[:n1], nthreads=1)
    t_one = time.perf_counter() - t0
    both = (ro["status"] == 0) & (status_gpu[:sample] == 0)
    du = np.abs(U_gpu[:sample].astype(np.float64) - Uo).max(axis=1)
    base = {"value": sample / t_all, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"first {sample} instances of the timed batch, fp64 oracle, OpenMP over instances "
                      f"({t_all:.2f} s wall)",
            "single_core_value": n1 / t_one, "single_core_sample": f"first {n1} instances, 1 thread"}
    parity = {"n": int(sample), "same_status_frac": float(np.mean(ro["status"] == status_gpu[:sample])),
              "both_converged": int(both.sum()),
              "median_abs_du_both_converged": float(np.median(du[both])) if both.any() else None,
              "median_abs_du_all": float(np.median(du)),
              "note": "default tolerances on both sides; see DESIGN.md 'parity protocol' for why max|du| is not "
                      "meaningful at OpEn's default Lipschitz-estimator step"}
    return base, parity


if __name__ == "__main__":
    main()
