#!/usr/bin/env python3
"""One Newton step (and the err history) of the block-tree path against the dense rocSOLVER path on a synthetic feeder:
relative difference of the state after 1, 2, 3 iterations.  python tools/step_check.py [buses] [hmax]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth
INPUTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hmax = int(sys.argv[2]) if len(sys.argv) > 2 else 11
tmp = tempfile.mkdtemp()
fb, fl = synth.gen(n, seed=0, outdir=tmp)
st = hp.Settings(H_MAX=hmax)
buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
res = {}
N_unknowns = 2 * n * len(st.HARMONICS)
if N_unknowns * N_unknowns >= 2 ** 31:
    sys.exit("dense reference impossible at this size (N*N >= 2^31)")
for solver in ("dense", "block_tree"):
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver, max_scenarios=1)
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    dm.mismatch(want_f=False)
    states = []
    for it in range(3):
        dm.iterate(1)
        Vm, Va = dm.get_state()
        states.append((Vm[0].copy(), Va[0].copy()))
    res[solver] = states
    dm.close()
for it in range(3):
    a, b = res["dense"][it], res["block_tree"][it]
    print("after %d iteration(s): max |dVm| %.3e  max |dVa| %.3e   (max |Vm| %.3e)"
          % (it + 1, np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max(), np.abs(a[0]).max()))
