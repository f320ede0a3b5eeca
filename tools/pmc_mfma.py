#!/usr/bin/env python3
"""Matrix-pipe and LDS counters of the bench kernels from the two rocprofv3 --pmc passes of `tools/measure.sh <tag> mfma`
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES | SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS) -> per kernel sums.
    python tools/pmc_mfma.py profiles/r02/pmc_mfma_lds_<tag>.json [steps_profiled=17]"""
import collections
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else None


out_path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 17
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
dur = collections.defaultdict(float)
N_SIMD, CLK = 1024, 2.4e9          # 256 CUs x 4 SIMDs; shader clock
for d in ("pmc_SQ_VALU_MFMA_BUSY_CYCLES", "pmc_SQ_INSTS_VALU_MFMA_F64"):
    fs = glob.glob(os.path.join(REPO, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    f = max(fs, key=os.path.getmtime)
    seen = set()
    if d.endswith("BUSY_CYCLES"):                       # kernel durations of the same pass (its kernel trace)
        tr = f.replace("counter_collection.csv", "kernel_trace.csv")
        if os.path.exists(tr):
            for r in csv.DictReader(open(tr)):
                k = short(r["Kernel_Name"])
                if k:
                    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), d)
        if d.endswith("BUSY_CYCLES") and key not in seen:
            seen.add(key)
            calls[k] += 1
res = {"command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-iters 0 "
                  "--no-finish --no-single --sweep-1gpu 0 (two passes; tools/measure.sh <tag> mfma)", "steps_profiled": steps, "kernels": {}}
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    d = {n: v for n, v in c.items()}
    if c.get("SQ_BUSY_CYCLES"):
        d["mfma_busy_over_sq_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CYCLES"]
    if dur.get(k):
        d["kernel_time_s"] = dur[k]
        d["mfma_busy_fraction_of_simd_time"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (N_SIMD * dur[k] * CLK)
    d["mfma_f64_insts_per_step"] = c.get("SQ_INSTS_VALU_MFMA_F64", 0.0) / steps
    d["mfma_f64_flop_per_step"] = d["mfma_f64_insts_per_step"] * 2048.0
    d["dispatches"] = calls.get(k, 0)
    res["kernels"][k] = d
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
for k, d in list(res["kernels"].items())[:8]:
    print("%-26s MFMA busy of SIMD-time %.3f   MFMA f64 insts/step %.3g   LDS bank conflict cycles %.3g  LDS active %.3g" %
          (k, d.get("mfma_busy_fraction_of_simd_time", 0), d["mfma_f64_insts_per_step"], d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_ACTIVE_INST_LDS", 0)))
