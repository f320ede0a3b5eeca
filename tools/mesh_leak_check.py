#!/usr/bin/env python3
"""Device-memory leak check of meshed handles: 90 create / solve / destroy cycles in the three forms of the bordered step.   python tools/mesh_leak_check.py   (GPU)"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import ctypes as C
import numpy as np
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth, _lib
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e6
INP = "tests/golden/inputs"
fb, fl = synth.gen(300, seed=2, outdir=tempfile.mkdtemp()); synth.add_ties(fl, 300, 5)
st = hp.Settings(H_MAX=27); buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS); NE = hp.import_Norton_Equivalents(buses, True, st, INP)
def once(opt=None):
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", max_scenarios=4, options=opt)
    dm.set_loads(np.tile(buses["P"].to_numpy(float), (4, 1)), np.tile(buses["Q"].to_numpy(float), (4, 1))); dm.set_state(None, None, n_scen=4); dm.fund_pf(1e-6, 30); dm.solve(1e-4, 5); dm.close()
once(); once("HPF_BORDER_GJ=0"); once("HPF_MESH_SEL=0")
f0 = free_mb()
for i in range(30):
    once(); once("HPF_BORDER_GJ=0"); once("HPF_MESH_SEL=0")
f1 = free_mb()
print("free device memory before / after 90 create-solve-destroy cycles of meshed handles: %.1f / %.1f MB (difference %.1f MB)" % (f0, f1, f0 - f1))
