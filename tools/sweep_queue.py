#!/usr/bin/env python3
"""Config-4 sweep on ONE GPU through a handle with fewer slots than scenarios: hpf_solve_queue (slots refilled as scenarios converge) against
fixed waves, end to end (pf included).   python tools/sweep_queue.py [scenarios] [S_max ...]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import harmonic_power_flow_amd as hp  # noqa: E402
from harmonic_power_flow_amd import sweep, synth  # noqa: E402

n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Ss = [int(a) for a in sys.argv[2:]] or [128]
args = bench.parse([])
inp = bench.build_inputs(args, hp)
n = inp["n"]
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
scale = np.stack([synth.scenario_scale(n, s) for s in range(n_scen)])
P, Q = P0 * scale, Q0 * scale
for S in Ss:
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                        inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
    base = None
    CH = [int(c) for c in os.environ.get("HPF_QUEUE_CHUNKS", "1,2,4,2").split(",")]       # queue_chunk values to compare
    for mode, chunk in [("waves", 0)] + [("queue", c) for c in CH] + [("waves", 0)]:
        if chunk:
            dm.set_option("queue_chunk", chunk)
        t0 = time.perf_counter()
        rec = sweep.solve_scenarios(dm, P, Q, refill=(mode == "queue"))
        t = time.perf_counter() - t0
        if base is None:
            base = rec.copy()
        same = np.array_equal(rec.view(np.uint8), base.view(np.uint8))
        print("S_max=%5d %s chunk=%d: %d scenarios, %d iterations in %.1f ms = %.0f NR it/s end to end; records identical to the first run: %s"
              % (S, mode, chunk, n_scen, int(rec["n_iter"].sum()), 1e3 * t, rec["n_iter"].sum() / t, same), flush=True)
    dm.close()
