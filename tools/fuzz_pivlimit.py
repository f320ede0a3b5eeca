#!/usr/bin/env python3
"""How the static-pivot monitor's limit (option "pivot_growth_limit_log10", default 10) trades accuracy of the fused block-tree step against repeats with
partial pivoting: the radial cases of tools/fuzz_parity.py's generator, ONE Newton iteration through hpf_solve (which repeats flagged scenarios with
pivoting) per limit, against the dense rocSOLVER step.   python tools/fuzz_pivlimit.py [cases=200] [seed=33] [limits=10,8,6,5,4]"""
import os
import sys
import tempfile

import numpy as np

os.environ.setdefault("HPF_ENV_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth

INPUTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 33)
limits = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "10,8,6,5,4").split(",")]
worst = {L: 0.0 for L in limits}
flagged = {L: 0 for L in limits}
over = {L: 0 for L in limits}
done = 0
for case in range(cases):
    n = int(rng.integers(33, 420))
    hmax = int(rng.choice([5, 11, 15, 19, 25, 27, 35, 51, 59, 75, 99]))
    if hmax > 51:
        n = min(n, 160)
    frac = float(rng.choice([0.05, 0.15, 0.35, 0.6, 0.85]))
    n_pv = int(rng.choice([0, 0, 1, 2]))
    n_ties = int(rng.choice([0, 0, 0, 1, 3, 6]))
    seed = int(rng.integers(0, 10 ** 6))
    if n_ties:
        continue                                   # (a meshed handle has no pivoted repeat: radial cases only)
    tmp = tempfile.mkdtemp()
    fb, fl = synth.gen(n, seed=seed, frac_nl=frac, outdir=tmp)
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
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="dense")
    dm.set_loads(P0, Q0)
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    seed_state = dm.get_state()
    dm.solve(1e-30, 1)
    ref = dm.get_state()
    dm.close()
    step = max(np.abs(ref[0] - seed_state[0]).max(), np.abs(ref[1] - seed_state[1]).max(), 1.0)
    line = "case %3d: n=%3d Hn=%2d nl=%.2f pv=%d seed=%6d step %.1e :" % (case, nn, Hn, frac, n_pv, seed, step)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree")
    try:
        for L in limits:
            dm.set_option("pivot_growth_limit_log10", L)
            dm.set_loads(P0, Q0)
            dm.set_state(*seed_state)
            dm.solve(1e-30, 1)
            got = dm.get_state()
            fl_ = int(dm.stats()["flags"][0])
            dev = max(np.abs(got[0] - ref[0]).max(), np.abs(got[1] - ref[1]).max()) / step
            worst[L] = max(worst[L], dev)
            flagged[L] += 1 if (fl_ & 8) else 0
            over[L] += 1 if dev > 1e-7 else 0
            line += "  L=%d %.1e%s" % (L, dev, "*" if fl_ & 16 else ("!" if fl_ & 8 else ""))
    finally:
        dm.close()
    done += 1
    print(line, flush=True)
print("%d radial cases; per limit (log10): worst deviation relative to the step | cases above 1e-7 | cases flagged (repeated with partial pivoting)" % done)
for L in limits:
    print("  limit 1e%-2d  worst %.1e   above 1e-7: %3d   flagged: %3d" % (L, worst[L], over[L], flagged[L]))
