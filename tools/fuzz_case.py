#!/usr/bin/env python3
"""ONE case of tools/fuzz_parity.py (python tools/fuzz_case.py n hmax frac n_pv n_ties seed): per-bus deviation of the first Newton step.
Randomised parity sweep on the GPU: the first Newton step (pf seed -> one iteration) of the block-tree path against the dense
rocSOLVER path over random radial feeders (size, share of nonlinear buses, PV buses, harmonic count -> block sizes 12 / 28 / 52 and
the generic kernels).  One step is compared (not converged states) because later iterates of the solver-sensitive cases amplify
rounding differences; the deviation is judged relative to the size of the step (first steps of 10-100 rad occur).  python tools/fuzz_parity.py [cases=24] [seed=0]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth

INPUTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
n, hmax, frac, n_pv, n_ties, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
tmp = tempfile.mkdtemp()
fb, fl = synth.gen(n, seed=seed, frac_nl=frac, outdir=tmp)
if n_ties:                              # loop-closing lines: the block-tree path's bordered step
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgb", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "make_golden_bench.py"))
    mgb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgb)
    mgb.add_ties(fl, n, n_ties, seed=seed)
if n_pv:
    rows = open(fb).read().splitlines()
    for bid in range(2, 2 + n_pv):
        cols = rows[bid].split(";")
        cols[1], cols[2], cols[4], cols[5] = "PV", "gen_%d" % bid, "-120", "0"
        rows[bid] = ";".join(cols)
    open(fb, "w").write("\n".join(rows) + "\n")
st = hp.Settings(H_MAX=hmax)
buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
Hn = len(st.HARMONICS)
if (2 * nn * Hn) ** 2 >= 2 ** 31:
    sys.exit('too large for the dense comparator')
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
S = 2
P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
scale = np.stack([synth.scenario_scale(nn, s) for s in range(S)])
res = {}
for solver in ("dense", "block_tree"):
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver, max_scenarios=S)
    try:
        dm.set_loads(P0 * scale, Q0 * scale)
        dm.set_state(None, None, n_scen=S)
        dm.fund_pf(1e-6, 30)
        seed_state = dm.get_state()
        dm.mismatch(want_f=False)
        dm.iterate(1)
        res[solver] = dm.get_state()
    finally:
        dm.close()

for sc in range(S):
    dVm = np.abs(res["dense"][0][sc] - res["block_tree"][0][sc]).reshape(Hn, nn)
    dVa = np.abs(res["dense"][1][sc] - res["block_tree"][1][sc]).reshape(Hn, nn)
    step = max(np.abs(res["dense"][0][sc] - seed_state[0][sc]).max(), np.abs(res["dense"][1][sc] - seed_state[1][sc]).max(), 1.0)
    worst_bus = np.argsort(-dVa.max(axis=0))[:6]
    print("scenario %d: step %.1e  max|dVm| %.1e max|dVa| %.1e  worst buses %s  their |dVa| %s  worst harmonic positions %s" %
          (sc, step, dVm.max(), dVa.max(), worst_bus.tolist(), ["%.1e" % dVa.max(axis=0)[b] for b in worst_bus], np.argsort(-dVa.max(axis=1))[:4].tolist()))
