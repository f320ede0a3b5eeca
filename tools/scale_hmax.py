#!/usr/bin/env python3
"""Lock-step step time of the 1 000-bus feeder for other harmonic counts (block sizes 12 / 28 / 52): python tools/scale_hmax.py [S=128] [hmax ...]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import harmonic_power_flow_amd as hp  # noqa: E402
from harmonic_power_flow_amd import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for hmax in [int(a) for a in sys.argv[2:]] or [11, 27, 51]:
    args = bench.parse.__globals__["argparse"].Namespace(buses=1000, hmax=hmax)
    inp = bench.build_inputs(args, hp)
    n = inp["n"]
    P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                        inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
    scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
    dm.set_loads(P0 * scale, Q0 * scale)
    dm.set_state(None, None, n_scen=S)
    dm.fund_pf(1e-6, 30)
    dm.mismatch(want_f=False)
    dm.iterate(3)
    dm.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        dm.iterate(10)
        dm.sync()
        best = min(best, (time.perf_counter() - t0) / 10)
    print("hmax=%3d (b=%3d) S=%d: %8.3f ms/step  %9.0f NR it/s   %s" % (hmax, 2 * len(inp["st"].HARMONICS), S, 1e3 * best, S / best,
                                                                      {k: v for k, v in dm.tree_census().items() if k in ("gauss_jordan", "bordered", "levels", "fused_levels")}), flush=True)
    dm.close()
