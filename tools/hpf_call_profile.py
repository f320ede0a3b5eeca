#!/usr/bin/env python3
"""Where a repeated hp.hpf() call on the headline feeder spends its time on the host (cProfile of the third call).  python tools/hpf_call_profile.py"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import harmonic_power_flow_amd as hp  # noqa: E402
from harmonic_power_flow_amd import synth  # noqa: E402

INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
fb, fl = synth.gen(1000, seed=0, outdir=tempfile.mkdtemp())
st = hp.Settings(H_MAX=51)
buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
for _ in range(2):
    hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
for rj in (True, False):
    t0 = time.perf_counter()
    hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, return_jacobian=rj)
    print("hpf(return_jacobian=%s): %.2f ms" % (rj, 1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
