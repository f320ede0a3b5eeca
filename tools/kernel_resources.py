#!/usr/bin/env python3
"""VGPR / spill / LDS / occupancy report of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiles for gfx950 without a GPU).   python tools/kernel_resources.py [file.hip] [name filter]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "harmonic-power-flow_amd", "csrc", "hpf_block.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = os.environ.get("HPF_CFLAGS", "").split()
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-fPIC",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "")).split("(")[0]}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("spill", r"VGPR Spill: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-44s %5s %5s %5s %6s %7s %4s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratch", "occ", "lds"))
for r in rows:
    if flt in r["name"]:
        print("%-44s %5d %5d %5d %6d %7d %4d %7d" % (r["name"][:44], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1),
                                                  r.get("spill", -1), r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1)))
