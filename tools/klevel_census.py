#!/usr/bin/env python3
"""Static instruction census of k_level<52> by source phase: builds hpf_block.hip for gfx950 with -g (no GPU needed), disassembles the kernel and
attributes every instruction, through its inline chain (llvm-symbolizer --inlines), to the phase of factor_q_body / the batched body it was inlined
from.  The line ranges below follow the section comments of csrc/hpf_quad.hpp (update them when the file moves).   python tools/klevel_census.py"""
import collections
import os
import re
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
tmp = tempfile.mkdtemp()
src = os.path.join(REPO, "harmonic-power-flow_amd", "csrc", "hpf_block.hip")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-g", "-c", "--cuda-device-only",
                "-o", tmp + "/b.o", src], check=True, capture_output=True)
subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + tmp + "/b.o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                "--output=" + tmp + "/g.o"], check=True, capture_output=True)
sym = subprocess.run([LLVM + "llvm-readelf", "-s", tmp + "/g.o"], capture_output=True, text=True).stdout
m = re.search(r"^\s*\d+:\s+([0-9a-f]+)\s+(\d+)\s+FUNC.*k_levelILi52E", sym, re.M)
start, size = int(m.group(1), 16), int(m.group(2))
dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--start-address=0x%x" % start, "--stop-address=0x%x" % (start + size), tmp + "/g.o"],
                     capture_output=True, text=True).stdout
ins = [(int(a, 16), op) for op, a in re.findall(r"^\s+([a-z_0-9]+)\s.*//\s*([0-9A-Fa-f]{12}):", dis, re.M)]
out = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + tmp + "/g.o", "--inlines", "--functions=short"], input="\n".join("0x%x" % a for a, _ in ins),
                     capture_output=True, text=True).stdout
blocks = out.strip().split("\n\n")


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    for p, k in (("v_", "valu"), ("s_", "salu"), ("ds_", "lds")):
        if op.startswith(p):
            return k
    return "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"


# section comments of factor_q_body -> phases (first line of the section)
marks = [("---- A0.", "A0 Y_N staging"), ("---- A1.", "A1 roles: voltages, G/H, compress"), ("lane = row 2q+tr_", "A1 base diagonal + 2x2 children"),
         ("---- L.", "L lazy leaves + bordered children"), ("---- B0.", "B0 prefetch, super-leaf staging, LEAF image"),
         ("HPF_STAMP(sd3);", "reduction + Y_N, lazy S_k map"), ("    if (cleaf) {", "constant-inverse leaf branch (non-batched)"),
         ("    } else if (sleaf) {", "bordered bus branch (non-batched)"), ("---- A3.", "A3 Norton cross terms"), ("---- A4.", "A4 patch"), ("---- B. remaining", "B dense children"),
         ("---- C. blocked", "C Gauss-Jordan"), ("---- D. inverse", "D store Z, w"), ("---- E. push", "E push (incl. compress pushes)")]
lines = open(os.path.join(REPO, "harmonic-power-flow_amd", "csrc", "hpf_quad.hpp")).read().splitlines()
body0 = next(i for i, l in enumerate(lines) if "void factor_q_body(" in l) + 1
cuts = [(body0, "prologue / node record")]
for key, name in marks:
    ln = next((i + 1 for i, l in enumerate(lines) if key in l and i + 1 > body0), None)
    if ln:
        cuts.append((ln, name))
cuts.sort()
agg = collections.defaultdict(collections.Counter)
for (_, op), blk in zip(ins, blocks):
    fr = blk.split("\n")
    tag = None
    for i in range(0, len(fr) - 1, 2):
        mm = re.match(r"(.*):(\d+):\d+$", fr[i + 1])
        if not mm:
            continue
        ln = int(mm.group(2))
        if fr[i].startswith("factor_q_body"):
            tag = "GJ: compiler-generated (line 0)" if ln == 0 else "GJ: " + [n for c, n in cuts if c <= ln][-1]
        elif fr[i].startswith("sleaf_batch_body"):
            tag = "bordered buses, scenario-batched body"
        elif fr[i].startswith("leaf_batch_body"):
            tag = "lazy leaves, scenario-batched body"
    agg[tag or "k_level dispatch"][kind(op)] += 1
names = ["valu", "salu", "lds", "vmem", "mfma", "other"]
print("%-50s" % ("k_level<52>, %d instructions (-g build)" % len(ins)) + "".join("%7s" % n for n in names))
order = ["GJ: " + n for _, n in cuts] + ["GJ: compiler-generated (line 0)"]
for tag in order + sorted(t for t in agg if t not in order):
    if tag in agg:
        print("%-50s" % tag + "".join("%7d" % agg[tag][n] for n in names))
