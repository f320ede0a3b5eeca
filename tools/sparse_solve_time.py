import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import synth
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
INPUTS = os.path.join(REPO, "tests", "golden", "inputs")
for nb, hm in ((1000, 51), (2000, 99), (10000, 99)):
    fb, fl = synth.gen(nb, seed=0, outdir=tempfile.mkdtemp())
    st = hp.Settings(H_MAX=hm)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    det = {}
    hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, details=det, return_jacobian=False, max_iter_h=0)
    V = hp.api._frame(det["seed"][0], det["seed"][1], st.HARMONICS, n)
    J = hp.build_harmonic_jacobian(V, Y, NE, True, buses=buses)
    f, err = hp.harmonic_mismatch(V, Y, buses, NE, settings=st)
    x0 = hp.harmonic_state_vector(V, c=c)
    for rep in range(3):
        t = time.perf_counter(); x1 = hp.update_harmonic_state_vec(J, x0, f); print("n %d hmax %d: N %d nnz %d: %.1f ms" % (nb, hm, J.shape[0], J.nnz, 1e3 * (time.perf_counter() - t)))
    dx = x0 - x1
    print("   residual", np.abs(J @ dx - f).max() / (np.abs(J).dot(np.abs(dx)).max() + np.abs(f).max()))
