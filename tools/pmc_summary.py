#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: counter totals of the solve kernels PER STEP (= per call of
nmpc_solve_batch: since round 3 a call may enqueue several solve-kernel dispatches -- the axis-aligned kernel and its
general twin, the pilot and the second launch of the resumable solve) and per kernel name.
   pmc_summary.py steps=<timed + warm-up steps of the profiled command> <dir> [<dir> ...]"""
import glob, sys
import pandas as pd
args = sys.argv[1:]
steps = 1
if args and args[0].startswith("steps="):
    steps = int(args.pop(0).split("=")[1])
for d in args:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        t = pd.read_csv(f)
        t = t[t["Kernel_Name"].str.contains(r"solve_\w*kernel", regex=True)]
        if not len(t):
            continue
        print(f"== {f}  steps={steps} solve-kernel dispatches={t['Dispatch_Id'].nunique()}")
        per_disp = t.groupby(["Kernel_Name", "Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().reset_index()
        for name, g in per_disp.groupby("Kernel_Name"):
            short = name.replace("void ", "").replace("(anonymous namespace)::", "")
            short = short[:short.rfind("(")][-60:]
            row = t[t["Kernel_Name"] == name].iloc[0]
            print(f"   kernel {short}: dispatches={g['Dispatch_Id'].nunique()} grid={row['Grid_Size']} vgpr={row['VGPR_Count']} sgpr={row['SGPR_Count']} scratch={row['Scratch_Size']}")
        tot = per_disp.groupby("Counter_Name")["Counter_Value"].sum() / steps
        for k, v in tot.items():
            print(f"   {k:28s} {v:.4e}   per step, all solve kernels")
