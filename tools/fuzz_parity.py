#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: the first Newton step (pf seed -> one iteration) of the block-tree path, and hpf_sparse_solve on the
CSR Jacobian of the same state, against the dense rocSOLVER path over random feeders (size, share of nonlinear buses, PV buses, loop-closing lines,
harmonic count -> block sizes 12 / 28 / 52 / 100).  One step is compared (not converged states) because later iterates of the solver-sensitive cases amplify
rounding differences; the deviation is judged relative to the size of the step (first steps of 10-100 rad occur).  python tools/fuzz_parity.py [cases=24] [seed=0]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth

INPUTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(cases):
    n = int(rng.integers(33, 420))
    hmax = int(rng.choice([5, 11, 15, 19, 25, 27, 35, 51, 59, 75, 99]))
    if hmax > 51:
        n = min(n, 160)                     # (the dense comparator: N = 2 n Hn)
    frac = float(rng.choice([0.05, 0.15, 0.35, 0.6, 0.85]))
    n_pv = int(rng.choice([0, 0, 1, 2]))
    n_ties = int(rng.choice([0, 0, 0, 1, 3, 6]))
    seed = int(rng.integers(0, 10 ** 6))
    tmp = tempfile.mkdtemp()
    fb, fl = synth.gen(n, seed=seed, frac_nl=frac, outdir=tmp)
    if n_ties:                              # loop-closing lines: the block-tree path's bordered step, hpf_sparse_solve's bordered elimination
        synth.add_ties(fl, n, n_ties, seed=seed)
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
        continue
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
            if solver == "dense":
                # the reference's own call shape on the same state: CSR Jacobian -> update_harmonic_state_vec (hpf_sparse_solve), scenario 0
                f0, _ = dm.mismatch()
                J0 = dm.jacobian_csr(0)
                x0 = np.append(seed_state[1][0][1:], seed_state[0][0][c:])
                x_sp = hp.update_harmonic_state_vec(J0, x0, f0[0])
            dm.mismatch(want_f=False)
            dm.iterate(1)
            res[solver] = dm.get_state()
        finally:
            dm.close()
    dVm = np.abs(res["dense"][0] - res["block_tree"][0]).max()
    dVa = np.abs(res["dense"][1] - res["block_tree"][1]).max()
    step = max(np.abs(res["dense"][0] - seed_state[0]).max(), np.abs(res["dense"][1] - seed_state[1]).max(), 1.0)
    worst = max(worst, dVm / step, dVa / step)
    d_sp = float("nan")
    if True:
        x_dense = np.append(res["dense"][1][0][1:], res["dense"][0][0][c:])
        d_sp = np.abs(x_sp - x_dense).max()
        worst = max(worst, d_sp / step)
    print("case %2d: n=%3d Hn=%2d (b=%3d) nl=%.2f pv=%d ties=%d seed=%6d   first step max|dVm| %.1e max|dVa| %.1e  sparse solve %.1e   (step size %.1e)" %
          (case, nn, Hn, 2 * Hn, frac, n_pv, n_ties, seed, dVm, dVa, d_sp, step), flush=True)
    assert np.isfinite(dVm) and np.isfinite(dVa)
print("worst deviation relative to the step size %.2e" % worst)
sys.exit(0 if worst < 1e-6 else 1)     # wrong algebra shows as O(1); ill-conditioned first steps (100 rad) reach 1e-7 on either path
