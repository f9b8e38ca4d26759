#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of a bench.py command into
profiles/rNN_<workload>_traffic.json -- the record bench.py reads for roofline.traffic.
   make_traffic_json.py <round> <workload> <dtype> <batch> <np> <N> <fetch_dir> <write_dir> <out.json> [steps]
`steps` = timed + warm-up steps of the profiled command: a step (one nmpc_solve_batch call) may consist of several
solve-kernel dispatches (axis-aligned kernel + general twin, pilot + second launch of the resumable solve); the
counters are summed over all of them and divided by the number of steps.
Counter semantics and the gfx950 correction: MI355X_MICROARCH.md, section HBM (FETCH_SIZE tallies 128-B requests at
64 B -> x2 for wide coalesced reads; WRITE_SIZE exact; both in KB)."""
import glob
import json
import sys

import pandas as pd


STEPS = int(sys.argv[10]) if len(sys.argv) > 10 else 1


def mean_counter(d, name):
    vals = []
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        t = pd.read_csv(f)
        t = t[t["Kernel_Name"].str.contains(r"solve_\w*kernel", regex=True) & (t["Counter_Name"] == name)]
        if len(t):
            big = t.groupby("Kernel_Name")["Counter_Value"].sum().idxmax()
            vals.append((t["Counter_Value"].sum() / STEPS, big, t["Dispatch_Id"].nunique()))
    if not vals:
        raise SystemExit(f"no {name} rows under {d}")
    return vals[0]


rnd, workload, dtype, batch, np_, N, fdir, wdir, out = sys.argv[1:10]
batch, np_, N = int(batch), int(np_), int(N)
fetch_kb, kname, nd = mean_counter(fdir, "FETCH_SIZE")
write_kb, _, _ = mean_counter(wdir, "WRITE_SIZE")
w = 4 if dtype == "f32" else 8
alg = w * (np_ + 2 * N + 4) * batch
hbm = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
rec = {"round": int(rnd), "workload": workload, "dtype": dtype, "batch": batch,
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 bench.py (two separate passes; tools/run_profiles.sh)",
       "kernel": kname, "solve_kernel_dispatches": int(nd), "steps": STEPS,
       "FETCH_SIZE_KB_per_step": float(fetch_kb), "WRITE_SIZE_KB_per_step": float(write_kb),
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
