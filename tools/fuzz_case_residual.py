#!/usr/bin/env python3
"""Who is right when the dense and the block-tree first Newton steps of a fuzz case differ: relative residuals ||J dx - f|| of the dense rocSOLVER
step, the fused block-tree step, hpf_sparse_solve and SciPy's SuperLU (the reference's solver) on the SAME CSR Jacobian, and their distances.
python tools/fuzz_case_residual.py n hmax frac n_pv seed      (radial feeders)"""
import os, sys, tempfile
import numpy as np
import scipy.sparse.linalg as spl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harmonic_power_flow_amd as hp
from harmonic_power_flow_amd import api, synth
INPUTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inputs")
n, hmax, frac, n_pv, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
fb, fl = synth.gen(n, seed=seed, frac_nl=frac, outdir=tempfile.mkdtemp())
if n_pv:
    rows = open(fb).read().splitlines()
    for bid in range(2, 2 + n_pv):
        cols = rows[bid].split(";")
        cols[1], cols[2], cols[4], cols[5] = "PV", "gen_%d" % bid, "-120", "0"
        rows[bid] = ";".join(cols)
    open(fb, "w").write("\n".join(rows) + "\n")
st = hp.Settings(H_MAX=hmax)
buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
steps = {}
for solver in ("dense", "block_tree"):
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver)
    dm.set_loads(P0, Q0); dm.set_state(None, None, n_scen=1); dm.fund_pf(1e-6, 30)
    Vm0, Va0 = dm.get_state()
    f, _ = dm.mismatch()
    J = dm.jacobian_csr(0)
    dm.iterate(1)
    Vm1, Va1 = dm.get_state()
    steps[solver] = np.append(Va0[0][1:] - Va1[0][1:], Vm0[0][c:] - Vm1[0][c:])
    if solver == "block_tree":
        print("static-pivot flags after the step:", dm.stats()["flags"] if False else "n/a (hpf_iterate)")
    dm.close()
x0 = np.append(Va0[0][1:], Vm0[0][c:])
steps["hpf_sparse_solve"] = x0 - hp.update_harmonic_state_vec(J, x0, f[0])
steps["scipy SuperLU"] = spl.spsolve(J.tocsc(), f[0])
lu = spl.splu(J.tocsc())
ref = steps["scipy SuperLU"].copy()
for _ in range(3):                        # iterative refinement in extended precision of the residual (longdouble accumulate)
    r = f[0].astype(np.longdouble) - (J.astype(np.longdouble) @ ref.astype(np.longdouble))
    ref = ref + lu.solve(np.asarray(r, dtype=float))
steps["refined"] = ref
sc = np.abs(ref).max()
print("step size %.2e, cond estimate (1-norm, via splu) n/a" % sc)
for k, dx in steps.items():
    res = np.abs(J @ dx - f[0]).max() / (np.abs(J).dot(np.abs(dx)).max() + np.abs(f[0]).max())
    print("%-18s rel. residual %.2e   max |dx - refined| / step %.2e" % (k, res, np.abs(dx - ref).max() / sc))
# where the fused step's residual sits: per bus max |J dx - f| of its rows (stacked order: theta rows k - 1, V rows Nc + k - c, k = q n + i), the
# worst buses with their parent and kind (hpf_tree_plan dump of the same model)
if os.environ.get("FUZZ_CASE_BUSES"):
    import ctypes as C
    from harmonic_power_flow_amd import _lib, ingest
    Hn = len(st.HARMONICS)
    Nc = nn * Hn - 1
    r = np.abs(J @ steps["block_tree"] - f[0])
    e = np.abs(steps["block_tree"] - ref)
    per_bus_r, per_bus_e = np.zeros(nn), np.zeros(nn)
    for idx in range(J.shape[0]):
        k = idx - Nc + c if idx >= Nc else idx + 1
        per_bus_r[k % nn] = max(per_bus_r[k % nn], r[idx])
        per_bus_e[k % nn] = max(per_bus_e[k % nn], e[idx])
    dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, True, Hn)
    d = _lib.hpf_desc()
    rowptr = np.ascontiguousarray(Y.rowptr, dtype=np.int32); col = np.ascontiguousarray(Y.col, dtype=np.int32)
    Yv = np.ascontiguousarray(Y.Yval, dtype=np.complex128); dev = np.ascontiguousarray(dev, dtype=np.int32)
    Y_N = np.ascontiguousarray(Y_N, dtype=np.complex128); I_N = np.ascontiguousarray(I_N, dtype=np.complex128)
    d.n, d.m, d.c, d.Hn, d.nnz = nn, m, c, Hn, len(col)
    d.n_dev, d.coupled, d.solver, d.device, d.max_scenarios = int(n_dev), 1, 1, 0, 1
    d.rowptr, d.col = rowptr.ctypes.data_as(_lib.c_int_p), col.ctypes.data_as(_lib.c_int_p)
    d.Yval = Yv.view(np.float64).ctypes.data_as(_lib.c_dbl_p); d.dev_of_bus = dev.ctypes.data_as(_lib.c_int_p)
    d.Y_N = Y_N.view(np.float64).ctypes.data_as(_lib.c_dbl_p); d.I_N = I_N.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
    path = os.path.join(tempfile.mkdtemp(), "plan.txt")
    assert _lib.load().hpf_tree_plan(C.byref(d), path.encode()) == 0
    plan = {int(x.split()[0]): [int(v) for v in x.split()] for x in open(path) if not x.startswith("#")}
    par = np.full(nn, -1)
    for i in range(nn):
        for e2 in range(rowptr[i], rowptr[i + 1]):
            pass
    print("m = %d (buses >= m are nonlinear), c = %d (buses 1 .. c-1 are PV)" % (m, c))
    print("worst residual rows by bus:   bus  residual  error   plan record [k pard height depth kind vector_only hbm_children via_chain compress ...] or 2x2 algebra")
    for i in np.argsort(-per_bus_r)[:12]:
        print("   %4d  %.1e  %.1e   %s" % (i, per_bus_r[i], per_bus_e[i], plan.get(int(i), "2x2 algebra (linear subtree / chain)")))
    print("worst errors by bus:")
    for i in np.argsort(-per_bus_e)[:8]:
        print("   %4d  %.1e  %.1e   %s" % (i, per_bus_r[i], per_bus_e[i], plan.get(int(i), "2x2 algebra (linear subtree / chain)")))
