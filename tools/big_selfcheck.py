#!/usr/bin/env python3
"""Self-consistency at scale: a 10 000-bus x 25-harmonic feeder (b = 52 kernels, 520 k unknowns per scenario: no dense comparator), a few
scenarios, default build (k_level, bundle kernels, nested bordered buses) against the plain paths (HPF_FUSELEVEL=0 HPF_LINTREE=0
HPF_SLNEST=0 in a child process): converged voltages one Newton iteration past the stop rule.   python tools/big_selfcheck.py [n=10000] [S=20]"""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")      # A/B tooling: HPF_* switches of the environment reach hpf_create (include/hpf.h)
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = sys.argv[3] if len(sys.argv) > 3 else None


def run():
    import harmonic_power_flow_amd as hp
    from harmonic_power_flow_amd import api, synth
    tmp = tempfile.mkdtemp()
    fb, fl = synth.gen(n, seed=0, outdir=tmp)
    st = hp.Settings(H_MAX=51)
    buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, os.path.join(REPO, "tests", "golden", "inputs"))
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", max_scenarios=S)
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
    scale = np.stack([np.ones(nn)] + [synth.scenario_scale(nn, s) for s in range(1, S)])
    dm.set_loads(P0 * scale, Q0 * scale)
    dm.set_state(None, None, n_scen=S)
    dm.fund_pf(1e-6, 30)
    it, err, _ = dm.solve(1e-4, 50)
    dm.mismatch(want_f=False)
    dm.iterate(1)
    dm.sync()
    Vm, Va = dm.get_state()
    cen = dm.tree_census()
    dm.close()
    return it, err, Vm, Va, cen


if out:
    it, err, Vm, Va, cen = run()
    np.savez(out, it=it, err=err, Vm=Vm, Va=Va)
    print("plain paths:", cen, "iterations", it[:6])
    sys.exit(0)
it, err, Vm, Va, cen = run()
print("default:", cen, "iterations", it[:6], "max err %.2e" % err.max())
tmpf = os.path.join(tempfile.mkdtemp(), "plain.npz")
env = dict(os.environ, HPF_FUSELEVEL="0", HPF_LINTREE="0", HPF_SLNEST="0")
subprocess.check_call([sys.executable, os.path.abspath(__file__), str(n), str(S), tmpf], env=env)
g = np.load(tmpf)
U0, U1 = Vm * np.exp(1j * Va), g["Vm"] * np.exp(1j * g["Va"])
print("converged: default %d / plain %d of %d;  max|dU| default vs plain paths = %.3e" % ((err <= 1e-4).sum(), (g["err"] <= 1e-4).sum(), S, np.abs(U0 - U1).max()))
