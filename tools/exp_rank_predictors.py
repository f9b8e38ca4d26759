#!/usr/bin/env python3
"""Round 6, the drain tail: how good is the resumable solve's ranking, and what would a better key buy?

Per batch: (1) every instance for TWO outer iterations (what the pilot launch leaves: ||F2||, evaluations, inner iterations, the
inner solve's last residual), (2) the full solve (the evaluations each instance needs in the end). The unfinished instances'
REMAINING evaluations are then list-scheduled on the device's wavefront slots (greedy: the next instance in the order goes to
the slot that frees first) under different orders -- index order, the shipped key (top 10 bits of ||F2||), candidates, and
the true remaining work (LPT) -- and the makespan is reported in evaluations and against the bound max(sum / slots, longest).
The features and targets are saved to gpurun_out/rank_features_<family>.npz for fitting off the box.
   usage: exp_rank_predictors.py [passing|refscen|corridor ...]   env: B (65536), SLOTS (2048), DIMS=cfg1|cfg2"""
import heapq, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm

B = int(os.environ.get("B", "65536")); SLOTS = int(os.environ.get("SLOTS", "2048"))
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10"}[os.environ.get("DIMS", "cfg2")]
spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")


def batch(fam):
    if fam in ("refscen", "corridor"):
        cfg = nm.default_config_struct(); cfg.Ndynobs, cfg.max_active_dynobs = lay.Ndyn, spec['n_ped'] * spec['n_hyp']
        steps, hf = ((2, 14, 26), "reference") if fam == "refscen" else ((1, 8, 20), "corridor")
        P, _ = nm.scenarios.harvest_closed_loop(cfg, B, steps=steps, seed=13, n_ped=spec['n_ped'], n_hyp=spec['n_hyp'], dtype=np.float32, family=hf)
        return np.ascontiguousarray(P, np.float32)
    return np.ascontiguousarray(nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec), np.float32)


def solve(P, **ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = spec['n_ped'] * spec['n_hyp']
    for k, v in ov.items():
        setattr(cfg, k, v)
    n = P.shape[0]
    U = np.empty((n, 2 * lay.N), np.float32); st = np.empty(n, np.int32); it = np.empty((n, 2), np.int32); info = np.empty((n, 8), np.float32)
    with nm.Handle(cfg) as h:
        h.solve_raw(np.float32, P, n, U, status=st, iters=it, info=info)
        h.solve_raw(np.float32, P, n, U, status=st, iters=it, info=info)
        ms = h.last_kernel_ms()
    return dict(U=U, status=st, iters=it, info=info, ms=ms)


def makespan(work, order, slots=SLOTS):
    heap = [0.0] * slots
    for w in work[order]:
        heapq.heapreplace(heap, heap[0] + w)
    return max(heap)


def f2_bucket(f):
    return (np.maximum(f, 0).astype(np.float32).view(np.uint32) >> 21).astype(np.int64)


for fam in sys.argv[1:] or ["passing"]:
    P = batch(fam)
    one = solve(P, max_outer_iterations=2, staged=-1, tail_latency=-1)   # (the pilot: parks at the SECOND outer boundary, past the first exit test)
    full = solve(P, staged=-1, tail_latency=-1)
    shipped = solve(P)
    unf = (one["status"] != 0) & (full["iters"][:, 0] > 2)
    rem = np.maximum(full["info"][:, 4] - one["info"][:, 4], 0)[unf].astype(np.float64)
    feats = dict(f2=one["info"][unf, 1], fpr=one["info"][unf, 0], evals1=one["info"][unf, 4], inner1=one["iters"][unf, 1].astype(np.float32),
                 c=one["info"][unf, 3], dyn=one["info"][unf, 2])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"rank_features_{fam}.npz"), rem=rem, total=full["info"][unf, 4],
                        outer_full=full["iters"][unf, 0], inner_full=full["iters"][unf, 1], status_full=full["status"][unf], **feats)
    n = rem.size
    rng = np.random.default_rng(0)
    bound = max(rem.sum() / SLOTS, rem.max())
    orders = {
        "index order": np.arange(n),
        "random": rng.permutation(n),
        "shipped key: ||F2|| (10-bit bucket), descending": np.argsort(-f2_bucket(feats["f2"]), kind="stable"),
        "||F2|| exact, descending": np.argsort(-feats["f2"], kind="stable"),
        "evaluations used in the pilot, descending": np.argsort(-feats["evals1"], kind="stable"),
        "pilot iterations >= 600 first, then ||F2||": np.lexsort((-feats["f2"], -(feats["inner1"] >= 600).astype(np.int8))),
        "pilot iterations >= 600 first, then evaluations": np.lexsort((-feats["evals1"], -(feats["inner1"] >= 600).astype(np.int8))),
        "||F2|| bucket, then evaluations": np.lexsort((-feats["evals1"], -f2_bucket(feats["f2"]))),
        "true remaining work (LPT)": np.argsort(-rem, kind="stable"),
    }
    rows = {k: round(makespan(rem, o) / bound, 3) for k, o in orders.items()}
    top = np.argsort(-rem)[: max(1, n // 100)]
    rk = {}
    for k, o in orders.items():
        pos = np.empty(n, np.int64); pos[o] = np.arange(n)
        rk[k] = float(pos[top].mean() / n)
    print(json.dumps({"family": fam, "B": int(P.shape[0]), "unfinished_after_the_pilot": int(n), "slots": SLOTS,
                      "sum_remaining_evals_per_slot": round(rem.sum() / SLOTS), "longest_remaining": int(rem.max()),
                      "share_of_remaining_work_in_top_1pct": round(float(rem[top].sum() / rem.sum()), 3),
                      "pilot_iterations_ge_600": int((feats["inner1"] >= 600).sum()),
                      "makespan_over_bound": rows, "mean_position_of_top_1pct_in_order": {k: round(v, 3) for k, v in rk.items()},
                      "kernel_ms": {"one launch, index order": round(full["ms"], 2), "shipped": round(shipped["ms"], 2)}}), flush=True)
