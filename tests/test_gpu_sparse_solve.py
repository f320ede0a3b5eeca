"""GPU tests of hpf_sparse_solve / api.update_harmonic_state_vec(J sparse) -- HG:476-479 `x - spsolve(J, f)` for the reference's CSR Jacobian at
every size, and of the boundary repairs of round 5 (hpf_num_scenarios, the 64-bit / guarded hpf_dense_solve, a model with one harmonic).
Gates: against the reference's own first iterate (fixtures captured from the unmodified reference) within 1e-9 of the step; against SciPy's SuperLU
(the reference's solver, HG:478) on the same matrix at 1e-9 of the solution; relative residual 1e-12 where the matrix is too large for a host solve."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from conftest import GOLD, INPUTS

pytestmark = pytest.mark.gpu


def _hp():
    import harmonic_power_flow_amd as hp
    return hp


def _syn(hp, n, hmax, tmp_path, n_pv=0, seed=0):
    from harmonic_power_flow_amd import api, synth
    fb, fl = synth.gen(n, seed=seed, outdir=str(tmp_path))
    if n_pv:
        rows = open(fb).read().splitlines()
        for bid in range(2, 2 + n_pv):                      # IDs 2.. -> PV generators (reference dialect of net3_buses.csv)
            cols = rows[bid].split(";")
            cols[1], cols[2], cols[4], cols[5] = "PV", "gen_%d" % bid, "-150", "0"
            rows[bid] = ";".join(cols)
        open(fb, "w").write("\n".join(rows) + "\n")
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree")
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    return st, buses, lines, dm, c


def test_update_harmonic_state_vec_at_the_headline_shape_vs_the_reference_first_iterate(tmp_path):
    """build_harmonic_jacobian -> update_harmonic_state_vec as the reference's loop does (HG:536-542) on the 1 000-bus x
    25-harmonic feeder: J0 (CSR, 1.2 M entries), f0 from the device, the sparse solve, and the result against the REFERENCE's own first iterate
    V_it1 (its SuperLU step) within 1e-9 of the step -- no N x N array on host or device."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, "syn1000_H51_c.npz"), allow_pickle=True)
    st, buses, lines, dm, c = _syn(hp, 1000, 51, tmp_path)
    try:
        Vm0, Va0 = dm.get_state()
        f, _ = dm.mismatch()
        J = dm.jacobian_csr(0)
    finally:
        dm.close()
    assert J.shape == (51998, 51998) and J.nnz == 1221740 and J._hpf_dims == (1000, 1, 26)
    x0 = np.append(Va0[0][1:], Vm0[0][c:])
    t0 = time.perf_counter()
    x1 = hp.update_harmonic_state_vec(J, x0, f[0])
    t1 = time.perf_counter()
    x1b = hp.update_harmonic_state_vec(J, x0, f[0])
    t2 = time.perf_counter()
    assert np.array_equal(x1, x1b)                          # deterministic: fixed child order, no atomics in the arithmetic
    ref = np.append(g["V_it1"][1:, 1], g["V_it1"][c:, 0])
    step = np.abs(ref - np.append(g["V_it0"][1:, 1], g["V_it0"][c:, 0])).max()
    dev = np.abs(x1 - ref).max()
    print(f"\nsparse solve at N = 51 998: {1e3 * (t1 - t0):.1f} ms first call, {1e3 * (t2 - t1):.1f} ms second (reference spsolve: ~1 s); "
          f"max dev from the reference's V_it1 {dev:.2e} on a step of {step:.2e} ({dev / step:.1e} of the step)")
    assert dev <= 1e-9 * step
    # the same call without the dims the matrix carries: from the network of the last init_network call, and explicitly
    J2 = J.copy()
    assert not hasattr(J2, "_hpf_dims")
    assert np.array_equal(hp.update_harmonic_state_vec(J2, x0, f[0]), x1)
    assert np.array_equal(hp.update_harmonic_state_vec(J2.tocoo(), x0, f[0], dims=(1000, 1, 26)), x1)
    # a sparse matrix of this size whose numbering is unknown is refused, not densified (21.6 GB)
    from harmonic_power_flow_amd import api
    api._ctx["buses"] = None
    with pytest.raises(ValueError):
        hp.update_harmonic_state_vec(J2, x0, f[0])
    api._ctx["buses"] = buses


@pytest.mark.parametrize("n,hmax,n_pv", [(60, 11, 0), (120, 11, 2), (200, 27, 0), (150, 51, 3), (90, 1, 0)])
def test_sparse_solve_vs_superlu_on_the_same_matrix(n, hmax, n_pv, tmp_path):
    """The block elimination of the caller's CSR entries against scipy.sparse.linalg.spsolve (SuperLU: the reference's solver, HG:478) on the same
    matrix and right-hand side: every block size class of the kernel (b = 12, 28, 52), PV buses (identity padding of the missing Q / V_m), and a
    model with the fundamental alone (H_MAX = 1: bus 0 has no equation at all -> the dense path)."""
    import scipy.sparse.linalg as spl
    hp = _hp()
    st, buses, lines, dm, c = _syn(hp, n, hmax, tmp_path, n_pv=n_pv)
    try:
        assert c == 1 + n_pv
        Vm0, Va0 = dm.get_state()
        f, _ = dm.mismatch()
        J = dm.jacobian_csr(0)
        dm.solve(1e-4, 1)                                   # the fused block-tree step of the same state
        Vm1, Va1 = dm.get_state()
    finally:
        dm.close()
    x0 = np.append(Va0[0][1:], Vm0[0][c:])
    x1 = hp.update_harmonic_state_vec(J, x0, f[0])
    ref = x0 - spl.spsolve(J.tocsc(), f[0])
    scale = np.abs(ref - x0).max()
    assert np.abs(x1 - ref).max() <= 1e-9 * scale, (np.abs(x1 - ref).max(), scale)
    fused = np.append(Va1[0][1:], Vm1[0][c:])
    assert np.abs(x1 - fused).max() <= 1e-9 * scale


def _add_ties(fl, n, k, seed=42):
    """k loop-closing lines on a synthetic feeder (the generator of the meshed oracle fixtures, oracle/make_golden_bench.py: test infrastructure)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgb", os.path.join(os.path.dirname(GOLD), "..", "oracle", "make_golden_bench.py"))
    mgb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgb)
    return mgb.add_ties(fl, n, k, seed)


def _meshed(hp, n, hmax, k, tmp_path, solver="auto"):
    from harmonic_power_flow_amd import api, synth
    fb, fl = synth.gen(n, seed=0, outdir=str(tmp_path))
    ties = _add_ties(fl, n, k)
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver)
    dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
    dm.set_state(None, None, n_scen=1)
    dm.fund_pf(1e-6, 30)
    return st, dm, c, ties


@pytest.mark.parametrize("n,hmax,k", [(60, 11, 1), (150, 27, 3), (260, 51, 6), (120, 99, 2)])
def test_sparse_solve_of_meshed_networks_vs_superlu(n, hmax, k, tmp_path):
    """Loop-closing lines (round 5): the CSR Jacobian of a feeder with k ties goes through hpf_sparse_solve's bordered elimination (spanning tree
    factorised once, selected inversion over the endpoints' root paths, dense border system) -- against SuperLU on the same matrix at 1e-9 of the
    step, every block-size class, and against the step of the library's own bordered Newton iteration."""
    import scipy.sparse.linalg as spl
    from harmonic_power_flow_amd import _lib
    hp = _hp()
    st, dm, c, ties = _meshed(hp, n, hmax, k, tmp_path, solver="block_tree")
    try:
        assert dm.tree_census()["ties"] == k
        Vm0, Va0 = dm.get_state()
        f, _ = dm.mismatch()
        J = dm.jacobian_csr(0)
        dm.solve(1e-4, 1)
        Vm1, Va1 = dm.get_state()
    finally:
        dm.close()
    x0 = np.append(Va0[0][1:], Vm0[0][c:])
    # the C entry point itself accepts the pattern (no dense fallback of the api involved)
    lib = _lib.load()
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    Jc = J.tocsr()
    indptr, indices, data = Jc.indptr.astype(np.int32), Jc.indices.astype(np.int32), np.ascontiguousarray(Jc.data, dtype=float)
    f0 = np.ascontiguousarray(f[0], dtype=float)
    dx = np.full(f0.size, np.nan)
    rc = lib.hpf_sparse_solve(0, n, c, len(st.HARMONICS), indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip), data.ctypes.data_as(dp),
                              f0.ctypes.data_as(dp), dx.ctypes.data_as(dp))
    assert rc == 0
    ref = spl.spsolve(J.tocsc(), f0)
    scale = np.abs(ref).max()
    assert np.abs(dx - ref).max() <= 1e-9 * scale, (np.abs(dx - ref).max(), scale)
    x1 = hp.update_harmonic_state_vec(J, x0, f0)
    assert np.array_equal(x1, x0 - dx)
    fused = np.append(Va1[0][1:], Vm1[0][c:])
    assert np.abs(x1 - fused).max() <= 1e-8 * scale


def test_sparse_solve_of_the_headline_feeder_with_20_ties(tmp_path):
    """1 000 buses x 25 harmonics + 20 loop-closing lines (40 endpoint buses, border 2 080; the meshed oracle fixture's network): the dense LU of
    this matrix would be 21.6 GB.  Relative residual, SuperLU on the same matrix, wall time."""
    import scipy.sparse.linalg as spl
    hp = _hp()
    st, dm, c, ties = _meshed(hp, 1000, 51, 20, tmp_path, solver="block_tree")
    try:
        Vm0, Va0 = dm.get_state()
        f, _ = dm.mismatch()
        J = dm.jacobian_csr(0)
    finally:
        dm.close()
    x0 = np.append(Va0[0][1:], Vm0[0][c:])
    hp.update_harmonic_state_vec(J, x0, f[0])
    t0 = time.perf_counter()
    x1 = hp.update_harmonic_state_vec(J, x0, f[0])
    t1 = time.perf_counter()
    dx = x0 - x1
    res = np.abs(J @ dx - f[0]).max() / (np.abs(J).dot(np.abs(dx)).max() + np.abs(f[0]).max())
    t2 = time.perf_counter()
    ref = spl.spsolve(J.tocsc(), f[0])
    t3 = time.perf_counter()
    dev = np.abs(dx - ref).max() / np.abs(ref).max()
    print(f"\nsyn1000 + 20 ties, N = {J.shape[0]}: {1e3 * (t1 - t0):.1f} ms (SuperLU on the host: {1e3 * (t3 - t2):.0f} ms); relative residual {res:.1e}, "
          f"{dev:.1e} of the step from SuperLU")
    assert res < 1e-12 and dev < 1e-9


def test_sparse_solve_ring_of_dense_blocks_through_the_c_abi():
    """The smallest meshed pattern, dense random blocks (nothing of the reference's structure): buses 0 - 1 - 2 - 0 (tree 0 -> 1, 0 -> 2, tie 1 - 2) and a
    5-bus graph with two ties sharing an endpoint, against numpy.linalg.solve; a singular border system is reported (HPF_E_SINGULAR)."""
    from harmonic_power_flow_amd import _lib
    import scipy.sparse as sp
    lib = _lib.load()
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)

    def call(n, c, Hn, A, f):
        J = sp.csr_matrix(A)
        indptr, indices = J.indptr.astype(np.int32), J.indices.astype(np.int32)
        data, f = np.ascontiguousarray(J.data, dtype=float), np.ascontiguousarray(f, dtype=float)
        dx = np.full(f.size, np.nan)
        rc = lib.hpf_sparse_solve(0, n, c, Hn, indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip), data.ctypes.data_as(dp), f.ctypes.data_as(dp),
                                  dx.ctypes.data_as(dp))
        return rc, dx
    rng = np.random.default_rng(11)
    for n, c, Hn, edges in ((3, 1, 2, [(0, 1), (0, 2), (1, 2)]), (5, 2, 3, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 1), (4, 2)]), (4, 1, 9, [(0, 1), (1, 2), (2, 3), (3, 0)])):
        Nc = n * Hn - 1
        N = 2 * Nc - (c - 1)

        def bus_of(r):
            k = r - Nc + c if r >= Nc else r + 1
            return k % n
        adj = {(a, b2) for a, b2 in edges} | {(b2, a) for a, b2 in edges} | {(a, a) for a in range(n)}
        A = np.zeros((N, N))
        for r in range(N):
            for cc in range(N):
                if (bus_of(r), bus_of(cc)) in adj:
                    A[r, cc] = rng.normal() + (3.0 * Hn if r == cc else 0.0)
        f = rng.normal(size=N)
        rc, dx = call(n, c, Hn, A, f)
        assert rc == 0, (n, rc)
        ref = np.linalg.solve(A, f)
        assert np.abs(dx - ref).max() <= 1e-11 * np.abs(ref).max(), (n, np.abs(dx - ref).max())
    # an exactly singular BORDER system on a regular tree part: identity diagonal blocks, tree couplings present in the pattern with value 0, and the
    # tie blocks A(1, 2) = A(2, 1) = -I  ->  I + Q^T Z = [I -I; -I I]: rocSOLVER's zero pivot is reported (HPF_E_SINGULAR), nothing is returned as solved
    n, c, Hn = 3, 1, 2
    Nc = n * Hn - 1
    N = 2 * Nc - (c - 1)

    def bl(r):
        t = 1 if r >= Nc else 0
        k = r - Nc + c if t else r + 1
        return k % n, 2 * (k // n) + t
    rows, cols, vals = [], [], []
    for r in range(N):
        for cc in range(N):
            (i, l), (j, lc) = bl(r), bl(cc)
            if i == j:
                v = 1.0 if l == lc else None
            elif {i, j} == {1, 2}:
                v = -1.0 if l == lc else None
            else:
                v = 0.0
            if v is not None:
                rows.append(r), cols.append(cc), vals.append(v)
    Js = sp.csr_matrix((np.array(vals), (np.array(rows), np.array(cols))), shape=(N, N))
    assert (Js.data == 0.0).any()                             # (explicit zeros keep the tree edges 0 - 1, 0 - 2 in the pattern)
    indptr, indices = Js.indptr.astype(np.int32), Js.indices.astype(np.int32)
    data, f = np.ascontiguousarray(Js.data, dtype=float), np.ones(N)
    dx = np.full(N, np.nan)
    rc = lib.hpf_sparse_solve(0, n, c, Hn, indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip), data.ctypes.data_as(dp), f.ctypes.data_as(dp),
                              dx.ctypes.data_as(dp))
    assert rc == 3


def test_sparse_solve_block_size_100_residual(tmp_path):
    """b = 2 Hn = 100 (H_MAX = 99, the block size of BASELINE config 5; R = 7 register tiles, 81 KB of LDS per workgroup): 600 buses, 2.3 M entries;
    relative residual of the solution and agreement with the fused block-tree step."""
    hp = _hp()
    st, buses, lines, dm, c = _syn(hp, 600, 99, tmp_path)
    try:
        Vm0, Va0 = dm.get_state()
        f, _ = dm.mismatch()
        J = dm.jacobian_csr(0)
        dm.solve(1e-4, 1)
        Vm1, Va1 = dm.get_state()
    finally:
        dm.close()
    x0 = np.append(Va0[0][1:], Vm0[0][c:])
    x1 = hp.update_harmonic_state_vec(J, x0, f[0])
    dx = x0 - x1
    res = np.abs(J @ dx - f[0]).max() / (np.abs(J).dot(np.abs(dx)).max() + np.abs(f[0]).max())
    fused = np.append(Va1[0][1:], Vm1[0][c:])
    print(f"\nb = 100: relative residual {res:.1e}, vs the fused step {np.abs(x1 - fused).max():.1e} on {np.abs(dx).max():.1e}")
    assert res < 1e-13
    assert np.abs(x1 - fused).max() <= 1e-9 * np.abs(dx).max()


def test_sparse_solve_error_paths_and_meshed_fallback():
    from harmonic_power_flow_amd import _lib
    import scipy.sparse as sp
    hp = _hp()
    lib = _lib.load()
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)

    def call(n, c, Hn, J, f):
        J = J.tocsr()
        indptr, indices = J.indptr.astype(np.int32), J.indices.astype(np.int32)
        data, f = np.ascontiguousarray(J.data, dtype=float), np.ascontiguousarray(f, dtype=float)
        dx = np.full(f.size, np.nan)
        rc = lib.hpf_sparse_solve(0, n, c, Hn, indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip), data.ctypes.data_as(dp), f.ctypes.data_as(dp),
                                  dx.ctypes.data_as(dp))
        return rc, dx
    # a diagonal matrix on a 3-bus model is not a connected tree (no coupling at all) -> HPF_E_TOPOLOGY; the api then takes the dense LU
    n, c, Hn = 3, 1, 2
    N = 2 * n * Hn - 1 - c
    rc, _ = call(n, c, Hn, sp.identity(N, format="csr") * 2.0, np.ones(N))
    assert rc == -3
    x = hp.update_harmonic_state_vec(sp.identity(N, format="csr") * 2.0, np.zeros(N), np.ones(N), dims=(n, c, Hn))
    assert np.allclose(x, -0.5)
    # a path 0 - 1 - 2 with dense random blocks: solved; the same with one bus block zeroed: HPF_E_SINGULAR
    rng = np.random.default_rng(5)
    Nc = n * Hn - 1

    def bus_of(r):
        k = r - Nc + c if r >= Nc else r + 1
        return k % n
    A = np.zeros((N, N))
    for r in range(N):
        for cc in range(N):
            if abs(bus_of(r) - bus_of(cc)) <= 1:
                A[r, cc] = rng.normal() + (4.0 if r == cc else 0.0)
    f = rng.normal(size=N)
    rc, dx = call(n, c, Hn, sp.csr_matrix(A), f)
    assert rc == 0 and np.abs(A @ dx - f).max() < 1e-11
    A2 = A.copy()
    rows2 = [r for r in range(N) if bus_of(r) == 2]
    A2[np.ix_(rows2, rows2)] = 0.0
    rc, _ = call(n, c, Hn, sp.csr_matrix(A2), f)
    assert rc == 3
    # the reference's meshed nets (net2: a ring of 4 buses): the sparse route's bordered elimination (round 5; dense LU before)
    g = np.load(os.path.join(GOLD, "net2_H11_c.npz"), allow_pickle=True)
    J = sp.csr_matrix((g["J0_data"], (g["J0_row"], g["J0_col"])), shape=tuple(g["J0_shape"]))
    x0 = np.append(g["V_traj"][0][1:, 1], g["V_traj"][0][1:, 0])
    x1 = hp.update_harmonic_state_vec(J, x0, g["f0"], dims=(4, 1, 6))
    ref = np.append(g["V_traj"][1][1:, 1], g["V_traj"][1][1:, 0])
    assert np.abs(x1 - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    # argument errors
    assert call(0, 1, 2, sp.identity(3, format="csr"), np.ones(3))[0] == -1
    assert call(3, 1, 70, sp.identity(2 * 3 * 70 - 2, format="csr"), np.ones(2 * 3 * 70 - 2))[0] == -1      # 2 Hn > 128


def test_dense_solve_refuses_what_does_not_fit_and_takes_the_64_bit_path():
    """hpf_dense_solve: a system whose 8 N^2 bytes exceed the device's free memory returns HPF_E_NOMEM before anything is allocated or read (the
    pointers below are never dereferenced); N = 46 400 (N^2 > 2^31: rocSOLVER's 64-bit entry points, 17.2 GB) solves a diagonally dominant system."""
    from harmonic_power_flow_amd import _lib
    lib = _lib.load()
    dp = C.POINTER(C.c_double)
    one = np.ones(1)
    N_huge = 400000                                         # 1.28 TB
    rc = lib.hpf_dense_solve(0, N_huge, one.ctypes.data_as(dp), one.ctypes.data_as(dp), one.ctypes.data_as(dp))
    assert rc == -4
    N = 46400
    assert N * N >= 2 ** 31
    J = np.zeros((N, N), order="F")
    idx = np.arange(N)
    J[idx, idx] = 4.0
    J[idx[1:], idx[:-1]] = 1.0                              # sub-diagonal
    J[idx[:-1], idx[1:]] = -1.0
    J[0, N - 1] = 0.5                                       # (an entry beyond the 2^31-th element of the column-major array)
    f = np.cos(0.01 * idx)
    dx = np.empty(N)
    rc = lib.hpf_dense_solve(0, N, J.ctypes.data_as(dp), f.ctypes.data_as(dp), dx.ctypes.data_as(dp))
    assert rc == 0
    r = 4.0 * dx
    r[1:] += dx[:-1]
    r[:-1] -= dx[1:]
    r[0] += 0.5 * dx[N - 1]
    assert np.abs(r - f).max() < 1e-12


@pytest.mark.parametrize("solver", ["dense", "block_tree"])
def test_model_with_the_fundamental_alone(solver, tmp_path):
    """Hn = 1 (H_MAX = 1 or 2 -> HARMONICS = [1]) -- the reciprocal division of the per-entry kernels has no 32-bit magic
    for a divisor of 1 and mapped every thread to bus 0.  hpf() of a radial feeder with one harmonic against the oracle, both solvers."""
    import hpf_oracle as o
    hp = _hp()
    from harmonic_power_flow_amd import synth
    fb, fl = synth.gen(300, seed=2, outdir=str(tmp_path))
    st = hp.Settings(H_MAX=1)
    assert st.HARMONICS == [1]
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    det = {}
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, details=det, solver=solver)
    r = o.hpf(o.init_network(fb, fl), st.HARMONICS, True, INPUTS)
    assert det["solver"] == solver and n_iter_h == r["n_iter_h"]
    Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
    assert np.abs(Ud - r["Vm"] * np.exp(1j * r["Va"])).max() < 1e-8
    assert J.shape == (2 * n - 2, 2 * n - 2)
