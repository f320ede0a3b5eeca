#!/usr/bin/env python3
"""One scenario of the headline feeder, 20 lock-step NR iterations (for rocprofv3 --kernel-trace --stats: the per-kernel durations
on the critical path of a single-scenario iteration, BASELINE config 3)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import harmonic_power_flow_amd as hp  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = bench.argparse.Namespace(buses=1000, hmax=51)
inp = bench.build_inputs(args, hp)
n = inp["n"]
dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                    inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
dm.set_loads(np.tile(P0, (S, 1)), np.tile(Q0, (S, 1)))
dm.set_state(None, None, n_scen=S)
dm.fund_pf(1e-6, 30)
dm.mismatch(want_f=False)
dm.iterate(20)
dm.sync()
dm.close()
