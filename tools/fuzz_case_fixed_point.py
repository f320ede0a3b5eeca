import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import synth
INPUTS = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "inputs")
n, hmax, frac, n_pv, seed = 125, 75, 0.6, 2, 128047
fb, fl = synth.gen(n, seed=seed, frac_nl=frac, outdir=tempfile.mkdtemp())
rows = open(fb).read().splitlines()
for bid in range(2, 2 + n_pv):
    cols = rows[bid].split(";"); cols[1], cols[2], cols[4], cols[5] = "PV", "gen_%d" % bid, "-120", "0"; rows[bid] = ";".join(cols)
open(fb, "w").write("\n".join(rows) + "\n")
st = hp.Settings(H_MAX=hmax)
buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
out = {}
for solver in ("dense", "block_tree"):
    det = {}
    V, err, it, _ = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, solver=solver, details=det, return_jacobian=False, extra_iters=2)
    out[solver] = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
    print(solver, "iterations", it, "err", err, "flags", det["stats"]["flags"], "hist tail", det["err_hist"][-4:])
print("max |dU| fixed points dense vs block_tree: %.2e" % np.abs(out["dense"] - out["block_tree"]).max())
