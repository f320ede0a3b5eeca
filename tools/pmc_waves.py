#!/usr/bin/env python3
"""Per-kernel sums of the counter groups collected by tools/pmc_waves.sh (one --pmc pass per group): what the wavefronts of each kernel
spend their cycles on.   python tools/pmc_waves.py gpurun_out/<tag>_pmcw"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else None


root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id"))
names = sorted({c for v in agg.values() for c in v})
print("counters:", " ".join(names))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    print("\n%s  dispatches %d" % (k, len(disp[k])))
    for n in names:
        if n in c:
            extra = ""
            if n != "SQ_WAVE_CYCLES" and (n.startswith("SQ_WAIT") or n.startswith("SQ_ACTIVE") or n.endswith("BUSY_CYCLES") or n == "SQ_LDS_BANK_CONFLICT" or
                                          n == "SQ_LDS_IDX_ACTIVE" or n.startswith("SQ_INST_CYCLES") or n.startswith("SQ_INST_LEVEL")):
                extra = "  = %.3f of wave cycles" % (c[n] / wc)
            if n.startswith("SQ_INSTS") and c.get("SQ_WAVES"):
                extra = "  = %.1f per wave" % (c[n] / c["SQ_WAVES"])
            print("  %-28s %.4g%s" % (n, c[n], extra))
