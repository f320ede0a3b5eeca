#!/usr/bin/env python3
"""Timing-only ablation of k_mismatch<false>, one scenario group, per-launch HIP-event spans.  Needs a DIAGNOSTIC build of the library whose
kernel takes an `ablate` argument (HPF_DEBUG_ABLATE >> 8: 1 no Norton rows, 2 no row walk, 4 no LDS staging, 8 no reduction / atomic); round 4
built it by patching the round-3 kernel text (four uniform branches) into tools/bin/libhpf_ablate.so -- the product kernel has no such argument:
a run-time flag in front of its load batches changes the code it times (the patched round-4 kernel measured 50 us where the real one takes 38).
Without HPF_LIB_PATH the script times the product kernel (every `ablate` value then reads the same).  Results of round 4: DESIGN.md section 5."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench, harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
inp = bench.build_inputs(bench.parse([]), hp)
n = inp["n"]
P0, Q0 = inp["buses"]["P"].to_numpy(float), inp["buses"]["Q"].to_numpy(float)
for rep in range(2):
    for abl in (0, 1, 2, 8, 1 | 4, 1 | 4 | 8, 1 | 2 | 4, 1 | 2 | 4 | 8):
        os.environ["HPF_DEBUG_ABLATE"] = str(abl << 8)
        dm = hp.DeviceModel(n, inp["m"], inp["c"], inp["st"].HARMONICS, inp["Y"].rowptr, inp["Y"].col, inp["Y"].Yval, inp["dev"],
                            inp["Y_N"], inp["I_N"], inp["n_dev"], True, solver="block_tree", max_scenarios=S)
        dm.set_option("scenario_groups", 1)
        scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
        dm.set_loads(P0 * scale, Q0 * scale)
        dm.set_state(None, None, n_scen=S)
        dm.fund_pf(1e-6, 30)
        dm.mismatch(want_f=False)
        dm.iterate(3)
        dm.sync()
        dm.timing(True)
        dm.timing_reset()
        dm.iterate(10)
        dm.sync()
        tim = dm.timing_get()
        dm.timing(False)
        print("S=%d ablate=%2d: mismatch %6.1f us per launch   update %6.1f" % (S, abl, 1e3 * tim["mismatch"][0] / tim["mismatch"][1], 1e3 * tim["update"][0] / tim["update"][1]), flush=True)
        dm.close()
