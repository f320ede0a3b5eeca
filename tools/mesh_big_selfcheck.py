#!/usr/bin/env python3
"""Large-scale self-check of the factor-once bordered step: BASELINE config 5's feeder (10 000 buses, K = 49: N = 999 998, blocks of 100) with loop-closing
lines, factor-once form against the virtual-sweep form of rounds 2 - 4 (HPF_MESH_SEL=0) -- ms per Newton iteration and the distance of the fixed points.
    python tools/mesh_big_selfcheck.py [buses=10000] [H_MAX=99] [ties=3]     (GPU; the virtual form takes about a second per iteration)"""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import harmonic_power_flow_amd as hp              # noqa: E402
from harmonic_power_flow_amd import api, synth    # noqa: E402

INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
hmax = int(sys.argv[2]) if len(sys.argv) > 2 else 99
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fb, fl = synth.gen(nb, seed=0, outdir=tempfile.mkdtemp())
synth.add_ties(fl, nb, k)
st = hp.Settings(H_MAX=hmax)
buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
out = {}
for name, opt in (("factor-once", None), ("virtual sweeps", "HPF_MESH_SEL=0 HPF_BORDER_SLOTS=32")):      # (1.1 GB of state per virtual slot at this size)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", options=opt)
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    t0 = time.perf_counter()
    it, err, _ = dm.solve(1e-4, 60)
    t = time.perf_counter() - t0
    dm.mismatch(want_f=False)
    dm.iterate(2)
    dm.sync()
    Vm, Va = dm.get_state()
    cs = dm.tree_census()
    out[name] = Vm[0] * np.exp(1j * Va[0])
    print("%-15s N = %d, %d ties (border %d, %d buses on the root paths, %d levels): %d iterations, err %.1e, %.2f ms per iteration"
          % (name, dm.N, cs["ties"], cs["border_unknowns"], cs["root_path_buses"], cs["levels"], it[0], err[0], 1e3 * t / max(int(it[0]), 1)), flush=True)
    dm.close()
print("fixed points of the two forms differ by %.2e" % np.abs(out["factor-once"] - out["virtual sweeps"]).max())
