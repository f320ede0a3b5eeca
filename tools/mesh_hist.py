#!/usr/bin/env python3
"""Mismatch histories of the two forms of the bordered step on the headline feeder + k ties:  python tools/mesh_hist.py k   (GPU)"""
import os, sys, tempfile
os.environ.setdefault("HPF_ENV_SWITCHES", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth
k = int(sys.argv[1])
tmp = tempfile.mkdtemp(); fb, fl = synth.gen(1000, seed=0, outdir=tmp); synth.add_ties(fl, 1000, k)
st = hp.Settings(H_MAX=51); buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS); NE = hp.import_Norton_Equivalents(buses, True, st, "tests/golden/inputs")
res = {}
for name, opt, solver in (("factor-once", None, "block_tree"), ("virtual", "HPF_MESH_SEL=0", "block_tree")):
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver, options=opt)
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float)); dm.set_state(None, None, n_scen=1); dm.fund_pf(1e-6, 30)
    it, err, hist = dm.solve(1e-4, 50)
    res[name] = hist[0][:it[0] + 1]
    print(name, it[0], " ".join("%.1e" % e for e in hist[0][:it[0] + 1]))
    dm.close()
