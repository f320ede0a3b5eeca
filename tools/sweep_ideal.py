#!/usr/bin/env python3
"""How close is hpf_solve (stop rule per scenario, compaction, chunked polling) to the lock-step rate?  Solves the 128 bench scenarios,
takes their iteration counts, measures the lock-step step time at the live counts that occur, and compares the solve's wall time with
sum over iterations of step_time(live scenarios).   python tools/sweep_ideal.py [S=128]"""
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
args = bench.parse.__globals__["argparse"].Namespace(buses=1000, hmax=51)
inp = bench.build_inputs(args, hp)
n = inp["n"]
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)


def model(Smax):
    return hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                          inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=Smax)


dm = model(S)
scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
best = 1e9
for rep in range(3):
    dm.set_loads(P0 * scale, Q0 * scale)
    dm.set_state(None, None, n_scen=S)
    dm.fund_pf(1e-6, 30)
    dm.sync()
    t0 = time.perf_counter()
    it, err, _ = dm.solve(1e-4, 50)
    best = min(best, time.perf_counter() - t0)
it = np.asarray(it)
print("solve: %d iterations in %.2f ms = %.0f it/s; iterations min %d mean %.2f max %d" % (it.sum(), 1e3 * best, it.sum() / best, it.min(), it.mean(), it.max()))
live = [(it > k).sum() for k in range(it.max())]
# lock-step step time at live counts (same handle: the first `cnt` scenarios)
tcache = {}
for cnt in sorted(set(live), reverse=True):
    dm.set_loads((P0 * scale)[:cnt], (Q0 * scale)[:cnt])
    dm.set_state(None, None, n_scen=cnt)
    dm.fund_pf(1e-6, 30)
    dm.mismatch(want_f=False)
    dm.iterate(3)
    dm.sync()
    t0 = time.perf_counter()
    dm.iterate(8)
    dm.sync()
    tcache[cnt] = (time.perf_counter() - t0) / 8
ideal = sum(tcache[c] for c in live)
print("live scenarios per iteration:", live)
print("lock-step ms at those counts:", {c: round(1e3 * t, 3) for c, t in tcache.items()})
print("ideal (sum of lock-step steps) %.2f ms = %.0f it/s; hpf_solve reaches %.1f %% of it" % (1e3 * ideal, it.sum() / ideal, 100 * ideal / best))
dm.close()
