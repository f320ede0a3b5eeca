#!/usr/bin/env python3
"""Print the per-launch timeline of the last NR step from a rocprofv3 --kernel-trace CSV (gpurun_out/<dir>)."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
f = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
keys = ("k_factor_w", "k_assemble_w", "k_back_w", "k_lin", "k_mismatch", "k_update", "k_tree")
mine = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
idx = [i for i, r in enumerate(mine) if "k_lin_factor" in r["Kernel_Name"]][-1]
step = mine[idx:]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    nm = [k for k in keys if k in r["Kernel_Name"]][0]
    if nm == "k_lin":
        nm = "k_lin_factor" if "k_lin_factor" in r["Kernel_Name"] else "k_lin_back"
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-14s grid.x %7s  start %8.1f us  dur %7.1f us" % (nm, r["Grid_Size_X"], s, dur))
