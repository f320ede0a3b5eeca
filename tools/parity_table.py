#!/usr/bin/env python3
"""Default-path output against the reference, case by case (GPU): for every golden case captured from the unmodified reference
(tests/golden/*.npz) the result of the DEFAULT `hp.hpf()` call -- iteration count, final mismatch, max|dU| of the converged complex
voltages against the reference's own printed result -- and, where the iteration counts differ (another iterate below the stop
threshold), max|dU| at the FIXED POINT: the reference's algorithm (oracle, bit-identical to it on these cases) continued until the
mismatch stops falling against `hp.hpf(..., extra_iters=3)`.   python tools/parity_table.py > profiles/r03/parity_table.txt"""
import glob
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import harmonic_power_flow_amd as hp              # noqa: E402
import hpf_oracle as o                            # noqa: E402  (checker)
from harmonic_power_flow_amd import synth         # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
INPUTS = os.path.join(GOLD, "inputs")
rows = []
print("%-16s %-11s %4s %4s %10s %10s %12s %12s" % ("case", "solver", "ref", "ours", "ref err", "our err", "|dU| at stop", "|dU| fixed pt"))
for path in sorted(glob.glob(os.path.join(GOLD, "*_H*.npz"))):
    name = os.path.basename(path)[:-4]
    if name.endswith("mesh5") or name.startswith("syn10000"):
        continue
    g = np.load(path, allow_pickle=True)
    if "V_final" not in g.files or len(name.split("_")) != 3 or name.split("_")[2] not in ("c", "uc"):
        continue                                       # (scenario / mesh / config fixtures have their own tests: test_gpu_bench_config.py)
    net_name, hs, cs = name.split("_")[:3]
    hmax, coupled = int(hs[1:]), cs == "c"
    st = hp.Settings(H_MAX=hmax)
    if net_name.startswith("syn"):
        tmp = tempfile.mkdtemp()
        fb, fl = synth.gen(int(net_name[3:]), seed=0, outdir=tmp)
    else:
        fb, fl = os.path.join(INPUTS, net_name + "_buses.csv"), os.path.join(INPUTS, net_name + "_lines.csv")
        if not os.path.exists(fb):
            continue
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    for solver in (("auto",) if net_name.startswith("syn") else ("auto", "block_tree")):
        det = {}
        V, err_h, n_it, _ = hp.hpf(buses, lines, coupled, settings=st, ne_dir=INPUTS, verbose=False, solver=solver, return_jacobian=False, details=det)
        Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
        Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
        d_stop = np.abs(Ud - Ug).max()
        d_fix = float("nan")
        if n_it != int(g["n_iter_h"]) or d_stop >= 1e-8:
            if n <= 300:
                r = o.hpf(o.init_network(fb, fl), st.HARMONICS, coupled, INPUTS, thresh_h=1e-13, max_iter_h=int(g["n_iter_h"]) + 6)
                Uo = r["Vm"] * np.exp(1j * r["Va"])
            else:                                      # (the oracle needs a minute per iteration at 1 000 buses: the reference's last iterate
                Uo = Ug                                #  of syn1000, err 7e-10, IS its fixed point to 1e-9)
            V2, _, _, _ = hp.hpf(buses, lines, coupled, settings=st, ne_dir=INPUTS, verbose=False, solver=solver, return_jacobian=False, extra_iters=3)
            d_fix = np.abs(V2["V_m"].to_numpy() * np.exp(1j * V2["V_a"].to_numpy()) - Uo).max()
        print("%-16s %-11s %4d %4d %10.2e %10.2e %12.2e %12s" % (name, det.get("solver", solver), int(g["n_iter_h"]), n_it, float(g["err_h"]), err_h, d_stop,
                                                                 "%.2e" % d_fix if d_fix == d_fix else "same count"), flush=True)
