#!/usr/bin/env python3
"""HIP-event span of the harmonic mismatch kernel per launch and the lock-step step time, this library vs HPF_LIB_PATH alternatives are run as
separate processes by the caller.   python tools/mismatch_time.py [S ...]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import harmonic_power_flow_amd as hp  # noqa: E402
from harmonic_power_flow_amd import synth  # noqa: E402

Ss = [int(a) for a in sys.argv[1:]] or [128, 1024]
inp = bench.build_inputs(bench.parse([]), hp)
n = inp["n"]
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
for S in Ss:
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                        inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
    scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
    dm.set_loads(P0 * scale, Q0 * scale)
    dm.set_state(None, None, n_scen=S)
    dm.fund_pf(1e-6, 30)
    f, err = dm.mismatch()
    dm.iterate(3)
    dm.sync()
    for rep in range(2):
        K = 20
        t0 = time.perf_counter()
        dm.iterate(K)
        dm.sync()
        t = time.perf_counter() - t0
        dm.timing(True)
        dm.timing_reset()
        dm.iterate(10)
        dm.sync()
        tim = dm.timing_get()
        dm.timing(False)
        print("S=%5d: step %7.3f ms   mismatch %6.1f us per launch   f checksum %.17g %.17g" % (S, 1e3 * t / K, 1e3 * tim["mismatch"][0] / max(tim["mismatch"][1], 1), float(np.abs(f).sum()), float(err.sum())), flush=True)
    dm.close()
