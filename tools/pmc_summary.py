#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of the solve kernel."""
import glob, sys
import pandas as pd
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        t = pd.read_csv(f)
        t = t[t["Kernel_Name"].str.contains("solve_\w*kernel", regex=True)]
        g = t.groupby("Counter_Name")["Counter_Value"].mean()
        print(f"== {f}  dispatches={t['Dispatch_Id'].nunique()} grid={t['Grid_Size'].iloc[0]} vgpr={t['VGPR_Count'].iloc[0]} sgpr={t['SGPR_Count'].iloc[0]} scratch={t['Scratch_Size'].iloc[0]}")
        for k, v in g.items():
            print(f"   {k:28s} {v:.4e}")
