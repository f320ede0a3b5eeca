#!/usr/bin/env python3
"""Time per Newton iteration of the bordered block-tree step on the headline feeder with k loop-closing lines:  python tools/mesh_time.py [k ...]  (GPU)"""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")      # A/B tooling: HPF_* switches of the environment reach hpf_create (include/hpf.h)
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import harmonic_power_flow_amd as hp              # noqa: E402
from harmonic_power_flow_amd import api, synth    # noqa: E402

INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
for k in [int(a) for a in sys.argv[1:]] or [5, 20]:
    tmp = tempfile.mkdtemp()
    fb, fl = synth.gen(1000, seed=0, outdir=tmp)
    synth.add_ties(fl, 1000, k)
    st = hp.Settings(H_MAX=51)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    S = int(os.environ.get("MESH_S", "1"))                 # scenarios (load scalings of the sweep generator)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", max_scenarios=S)
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
    if S > 1:
        import numpy as np
        scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
        dm.set_loads(P0 * scale, Q0 * scale)
    else:
        dm.set_loads(P0, Q0)
    dm.set_state(None, None, n_scen=S)
    dm.fund_pf(1e-6, 30)
    seed = dm.get_state()
    dm.solve(1e-4, 3)
    dm.set_state(*seed)
    t0 = time.perf_counter()
    it, err, _ = dm.solve(1e-4, 50)
    t = time.perf_counter() - t0
    cs = dm.tree_census()
    print("syn1000 + %d ties: census %s; %d iterations in %.3f s = %.2f ms per iteration (err %.1e); levels %d, border systems through the pivoted LU %d"
          % (k, cs["ties"], it.max(), t, 1e3 * t / it.max(), err.max(), cs["levels"], cs["border_repivots"])
          + ("" if S == 1 else "; %d scenarios, %d scenario-iterations: %.1f us each" % (S, it.sum(), 1e6 * t / it.sum())))
    dm.close()
