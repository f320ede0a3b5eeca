#!/usr/bin/env python3
"""One Newton iteration of the bordered block-tree step as a kernel timeline, from a rocprofv3 --kernel-trace CSV of tools/mesh_time.py:
   rocprofv3 --kernel-trace -d gpurun_out/meshprof -o m --output-format csv -- python tools/mesh_time.py 5 ; python tools/mesh_timeline.py gpurun_out/meshprof/m_kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def nm(r):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    return m.group(1) if m and not m.group(1).startswith("k_Ailk") and m.group(1) != "k_singularity" else "rocSOLVER / rocBLAS"


idx = [i for i, r in enumerate(rows) if nm(r) == "k_border_prepare"]
if not idx:                                             # factor-once form: an iteration from one k_mismatch to the next (sweep, selected inversion, ...)
    idx = [i for i, r in enumerate(rows) if nm(r) == "k_mismatch"]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
if len(sys.argv) > 2 and sys.argv[2] == "two":          # (virtual form: two prepare launches per iteration)
    b = idx[len(idx) // 2 + 2]
t0 = int(rows[a]["Start_Timestamp"])
seg = []
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    k = nm(r)
    if seg and seg[-1][0] == k:
        seg[-1][2] = e
        seg[-1][3] += 1
        seg[-1][4] += e - s
    else:
        seg.append([k, s, e, 1, e - s])
for g in seg:
    print("%-22s start %8.1f end %8.1f us   launches %4d   busy %7.1f us" % tuple(g))
print("iteration wall %.1f us (under the tracer)" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
