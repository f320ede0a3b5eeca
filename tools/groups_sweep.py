#!/usr/bin/env python3
"""Lock-step NR throughput of the headline feeder vs the number of scenario groups (streams): python tools/groups_sweep.py [S ...]"""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")      # A/B tooling: HPF_* switches of the environment reach hpf_create (include/hpf.h)
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import harmonic_power_flow_amd as hp  # noqa: E402
from harmonic_power_flow_amd import synth  # noqa: E402

Ss = [int(a) for a in sys.argv[1:]] or [128, 1024]
args = bench.parse.__globals__["argparse"].Namespace(buses=1000, hmax=51)
inp = bench.build_inputs(args, hp)
n = inp["n"]
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
for S in Ss:
    for groups in [int(g) for g in os.environ.get("GROUPS_LIST", "3,4,5,6,8").split(",")]:
        dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                            inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
        dm.set_option("scenario_groups", groups)
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
        print("S=%5d groups=%d: %8.3f ms/step  %9.0f NR it/s" % (S, groups, 1e3 * best, S / best), flush=True)
        dm.close()
