import os, sys, tempfile, time
os.environ.setdefault("HPF_ENV_SWITCHES", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth
INP = "tests/golden/inputs"
for nb, hmax, S in ((1000, 51, 128), (10000, 99, 2), (3000, 27, 16), (500, 11, 16)):
    fb, fl = synth.gen(nb, seed=0, outdir=tempfile.mkdtemp())
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS); NE = hp.import_Norton_Equivalents(buses, True, st, INP)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", max_scenarios=S)
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
    scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
    for L in (10, 7, 6, 5, 4, 3):
        dm.set_option("pivot_growth_limit_log10", L)
        dm.set_loads(P0 * scale, Q0 * scale); dm.set_state(None, None, n_scen=S); dm.fund_pf(1e-6, 30)
        t0 = time.perf_counter(); it, err, _ = dm.solve(1e-4, 50); t = time.perf_counter() - t0
        fl_ = dm.stats()["flags"]
        print("n=%5d K=%2d S=%3d limit 1e%-2d: flagged %3d repeated %3d  iterations %d..%d  %.1f ms" % (nb, (hmax + 1) // 2, S, L, int(((fl_ & 8) != 0).sum()), int(((fl_ & 16) != 0).sum()), it.min(), it.max(), 1e3 * t), flush=True)
    dm.close()
