#!/usr/bin/env python3
"""Per-launch timeline of the last NR step of ONE scenario group (run with HPF_GROUPS=1) from a rocprofv3 --kernel-trace CSV
(gpurun_out/<dir>), plus per-kernel totals of that step."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
import os
f = max(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
keys = ("k_level", "k_lin_bundle_factor", "k_lin_bundle_back", "k_sleaf_batch", "k_sleaf_back_batch", "k_leaf_back_batch", "k_leaf_batch", "k_leafsum", "k_factor_q", "k_factor_w", "k_back_q", "k_back_w", "k_lin_level_factor", "k_lin_level_back", "k_lin_factor", "k_lin_back", "k_chain_factor", "k_chain_back", "k_mismatch", "k_update", "k_tree", "k_finalize")
mine = sorted([r for r in rows if any(k in r["Kernel_Name"] for k in keys)], key=lambda r: int(r["Start_Timestamp"]))
mm = [i for i, r in enumerate(mine) if "k_mismatch" in r["Kernel_Name"]]       # a step ends with its mismatch kernel
step = mine[mm[-2] + 1:mm[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
tot = {}
prev_end = t0
for r in step:
    nm = [k for k in keys if k in r["Kernel_Name"]][0]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - prev_end) / 1e3
    prev_end = int(r["End_Timestamp"])
    a = tot.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1; a[1] += dur; a[2] += max(gap, 0.0)
    print("%-14s grid %6d x %4d  start %8.1f us  dur %7.1f us  gap before %5.1f us"
          % (nm, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), s, dur, gap))
print()
for nm, (cnt, dur, gap) in tot.items():
    print("%-14s launches %3d  busy %8.1f us  gaps before %7.1f us" % (nm, cnt, dur, gap))
print("step wall %.1f us" % ((prev_end - t0) / 1e3))
