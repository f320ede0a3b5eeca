#!/usr/bin/env python3
"""Counter / known-bytes ratios of tools/pmc_calib.hip from its two rocprofv3 --pmc passes (gpurun_out/calib_FETCH_SIZE,
gpurun_out/calib_WRITE_SIZE) -> profiles/<round>/pmc_calib.json.   python tools/pmc_calib_report.py profiles/r02/pmc_calib.json"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN_TILE = (2 << 30) // 23296 * 23296
known = {"k_load16": 2 << 30, "k_load8": 2 << 30, "k_tile_load": KNOWN_TILE, "k_store16": 2 << 30, "k_store8": 2 << 30,
         "k_tile_store": KNOWN_TILE}
out = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- tools/bin/pmc_calib (two passes)",
       "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(os.path.join(REPO, "gpurun_out", "calib_" + c, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in known:
            if r["Kernel_Name"].startswith(k + "(") or r["Kernel_Name"] == k or (k + "(") in r["Kernel_Name"]:
                agg[k].append(float(r["Counter_Value"]) * 1024.0)
    for k, v in agg.items():
        d = out["kernels"].setdefault(k, {"known_bytes_per_launch": known[k]})
        d[c + "_bytes_per_launch"] = sum(v) / len(v)
        d[c + "_over_known"] = sum(v) / len(v) / known[k]
        d["launches"] = len(v)
out["reading"] = ("loads: FETCH_SIZE_over_known of k_load16 / k_load8 / k_tile_load is the factor by which the counter under-reports "
                  "this pattern's reads (multiply a kernel's FETCH_SIZE by its inverse); stores: WRITE_SIZE_over_known likewise")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
