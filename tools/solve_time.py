#!/usr/bin/env python3
"""hpf_solve wall time (stop rule) of the headline feeder at a few batch sizes, and the queued sweep: python tools/solve_time.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import synth, sweep
args = bench.parse.__globals__["argparse"].Namespace(buses=1000, hmax=51)
inp = bench.build_inputs(args, hp)
n = inp["n"]
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
for S in (1, 8, 128):
    dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"], inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
    scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
    best = 1e9
    for rep in range(4):
        dm.set_loads(P0 * scale, Q0 * scale); dm.set_state(None, None, n_scen=S); dm.fund_pf(1e-6, 30); dm.sync()
        t0 = time.perf_counter(); it, err, _ = dm.solve(1e-4, 50); best = min(best, time.perf_counter() - t0)
    print("S=%4d: solve %8.3f ms, %d iterations (max %d) -> %.4f ms per iteration of the batch, %.0f it/s" % (S, 1e3 * best, it.sum(), it.max(), 1e3 * best / it.max(), it.sum() / best))
    if S == 128:
        sc = np.stack([synth.scenario_scale(n, s) for s in range(1024)])
        sweep.solve_scenarios(dm, P0 * sc[:256], Q0 * sc[:256])
        t0 = time.perf_counter(); rec = sweep.solve_scenarios(dm, P0 * sc, Q0 * sc); t = time.perf_counter() - t0
        print("queue: 1024 scenarios through 128 slots: %.1f ms, %.0f it/s" % (1e3 * t, rec["n_iter"].sum() / t))
    dm.close()
