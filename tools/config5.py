#!/usr/bin/env python3
"""BASELINE config 5 (10 000 buses, K = 49, coupled): run on the GPU (default) or with the CPU oracle (--oracle) and save the
converged voltages for comparison.  The reference itself cannot run this size (dense Y_all = 80 GB, HG:141-143)."""
import os
os.environ.setdefault("HPF_ENV_SWITCHES", "1")      # A/B tooling: HPF_* switches of the environment reach hpf_create (include/hpf.h)
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
n = int(os.environ.get("CFG5_BUSES", "10000"))
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(REPO, "gpurun_out", "config5_gpu.npz")
tmp = tempfile.mkdtemp()
spec = __import__("importlib.util").util.spec_from_file_location("synth", os.path.join(REPO, "harmonic-power-flow_amd", "synth.py"))
synth = __import__("importlib.util").util.module_from_spec(spec)
spec.loader.exec_module(synth)
fb, fl = synth.gen(n, seed=0, outdir=tmp)
if "--oracle" in sys.argv:
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import hpf_oracle as o
    t0 = time.perf_counter()
    net = o.init_network(fb, fl)
    r = o.hpf(net, o.harmonics_upto(99), True, INPUTS)
    print("oracle: n_iter_h %d err %.3e loop %.1f s total %.1f s" % (r["n_iter_h"], r["err_h"], r["loop_s"], time.perf_counter() - t0), flush=True)
    np.savez_compressed(out, Vm=r["Vm"], Va=r["Va"], n_iter=r["n_iter_h"], err=r["err_h"], err_hist=r["err_hist"])
else:
    import harmonic_power_flow_amd as hp
    st = hp.Settings(H_MAX=99)
    t0 = time.perf_counter()
    res = hp.solve(fb, fl, coupled=True, settings=st, ne_dir=INPUTS)
    V = res["V"]
    print("gpu: n_iter_h %d err %.3e total %.1f s solver %s" % (res["n_iter_h"], res["err_h"], time.perf_counter() - t0, res["details"]["solver"]), flush=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, Vm=V["V_m"].to_numpy(), Va=V["V_a"].to_numpy(), n_iter=res["n_iter_h"], err=res["err_h"],
                        err_hist=res["details"]["err_hist"])
