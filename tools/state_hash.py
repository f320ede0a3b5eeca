#!/usr/bin/env python3
"""sha256 of the state after k lock-step iterations of the headline feeder (bit-identity check of two library builds on one box):
HPF_LIB_PATH=<lib.so> python tools/state_hash.py [scenarios] [iterations] [buses] [hmax]"""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import harmonic_power_flow_amd as hp  # noqa: E402
from harmonic_power_flow_amd import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
HM = int(sys.argv[4]) if len(sys.argv) > 4 else 51
args = bench.parse.__globals__["argparse"].Namespace(buses=NB, hmax=HM)
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
dm.iterate(K)
dm.sync()
Vm, Va = dm.get_state()
f, err = dm.mismatch()
print("S=%d K=%d n=%d hmax=%d  state %s  mismatch %s  err[0] %.17g" % (S, K, n, HM, hashlib.sha256(Vm.tobytes() + Va.tobytes()).hexdigest()[:16],
                                                                hashlib.sha256(f.tobytes()).hexdigest()[:16], err[0]))
dm.close()
