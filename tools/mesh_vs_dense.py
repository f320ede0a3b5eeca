#!/usr/bin/env python3
"""Bordered block-tree step against the dense rocSOLVER path on mid-size meshed feeders (ms per Newton iteration, one scenario):
   python tools/mesh_vs_dense.py    (GPU)"""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import harmonic_power_flow_amd as hp              # noqa: E402
from harmonic_power_flow_amd import api, synth    # noqa: E402

INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
COUPLED = os.environ.get("MESH_COUPLED", "1") != "0"     # MESH_COUPLED=0: the reference's uncoupled Norton model
for n, hmax, k in ((40, 11, 1), (40, 51, 2), (100, 27, 3), (100, 51, 3), (150, 51, 5), (300, 27, 8), (300, 51, 4)):
    tmp = tempfile.mkdtemp()
    fb, fl = synth.gen(n, seed=1, outdir=tmp)
    synth.add_ties(fl, n, k)
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, COUPLED, st, INPUTS)
    row = []
    for solver in ("dense", "block_tree"):
        dm = api._device_model(buses, Y, NE, COUPLED, st.HARMONICS, solver=solver)
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(None, None, n_scen=1)
        dm.fund_pf(1e-6, 30)
        seed = dm.get_state()
        dm.solve(1e-4, 3)
        dm.set_state(*seed)
        t0 = time.perf_counter()
        it, err, _ = dm.solve(1e-4, 50)
        t = time.perf_counter() - t0
        row.append("%s %2d it %.2f ms/it" % (solver, it[0], 1e3 * t / max(int(it[0]), 1)))
        N = dm.N
        dm.close()
    print("n = %3d, K = %2d, %d ties (N = %5d, %s): %s" % (n, (hmax + 1) // 2, k, N, "coupled" if COUPLED else "uncoupled", "   ".join(row)))
