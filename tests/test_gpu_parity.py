"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  (1) golden vectors captured from the unmodified reference (tests/golden/*.npz), and
  (2) the CPU oracle on the same seeded inputs.
Tolerances: the north-star bar is |dV| < 1e-8 p.u. per harmonic on the converged voltages (complex U and V_m after
the HG:545-549 normalisation) with the SAME iteration count; mismatch / Jacobian entries are compared at
1e-12 relative (FP64 everywhere; device sincos differs from glibc by <= 1 ulp)."""
import glob
import os

import numpy as np
import pytest

import hpf_oracle as o
from conftest import GOLD, INPUTS, check_jacobian_checksums

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*_H*.npz"))
                   if not os.path.basename(p).startswith("syn"))
TOL_V = 1e-8
# The NR trajectory of net1/K=25/coupled is chaotic in its first ~20 iterations and its LENGTH depends on the
# rounding of the linear solver: inside the reference itself, swapping SuperLU for LAPACK gives 32 iterations instead
# of 23, and an iteratively refined (exact to working precision) solve gives 27 (measured with the oracle, DESIGN.md
# "Parity").  All of them end at the same voltages.  For this case the voltages are the gate and the iteration
# count is only reported; every other case is robust and must reproduce the reference's iteration count.
SOLVER_SENSITIVE = {"net1_H51_c"}


def _hp():
    import harmonic_power_flow_amd as hp
    return hp


def _case(name):
    net_name, hs, cs = name.split("_")
    return net_name, int(hs[1:]), cs == "c"


def _paths(net_name):
    return os.path.join(INPUTS, f"{net_name}_buses.csv"), os.path.join(INPUTS, f"{net_name}_lines.csv")


def _model(hp, name, solver="dense", max_scenarios=1):
    from harmonic_power_flow_amd import api
    net_name, hmax, coupled = _case(name)
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, n, c = hp.init_network(*_paths(net_name), settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, coupled, st, INPUTS)
    dm = api._device_model(buses, Y, NE, coupled, st.HARMONICS, solver=solver, max_scenarios=max_scenarios)
    return st, buses, dm


def test_gpu_present_and_native_library_loaded():
    import torch
    assert torch.cuda.is_available()
    from harmonic_power_flow_amd import _lib
    assert _lib.load().hpf_version() >= 100


@pytest.mark.parametrize("name", CASES)
def test_hpf_matches_reference_golden(name):
    """Full drop-in path: CSV ingest -> pf -> harmonic NR -> post-processing, vs the reference's own result."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    net_name, hmax, coupled = _case(name)
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, n, c = hp.init_network(*_paths(net_name), settings=st)
    det = {}
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, coupled, settings=st, ne_dir=INPUTS, verbose=False, details=det)
    assert det["n_iter_f"] == int(g["n_iter_f"])
    np.testing.assert_allclose(np.stack(det["seed"], 1), g["V_pf"], rtol=0, atol=1e-13)
    if name not in SOLVER_SENSITIVE:
        assert n_iter_h == int(g["n_iter_h"]), (n_iter_h, int(g["n_iter_h"]))
    assert n_iter_h < 50 and err_h <= 1e-4
    ge = g["err_hist"]
    np.testing.assert_allclose(det["err_hist"][:3], ge[:3], rtol=1e-9, atol=1e-12 * ge[0])   # atol: round-off floor (lin4)
    Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    dU = np.abs(Ud - Ug).max()
    dVm = np.abs(V["V_m"].to_numpy() - g["V_final"][:, 0]).max()
    print(f"\n{name}: it {n_iter_h} err {err_h:.3e} (ref {float(g['err_h']):.3e}) max|dU| {dU:.2e} max|dVm| {dVm:.2e}")
    assert dU < TOL_V and dVm < TOL_V
    assert (V["V_m"].to_numpy() >= 0).all() and (V["V_a"].to_numpy() >= 0).all() and (V["V_a"].to_numpy() <= 2 * np.pi).all()   # np.mod(-tiny, 2pi) rounds to 2pi, in the reference as well
    thd = hp.get_THD(V).to_numpy()
    np.testing.assert_allclose(thd, g["THD"], rtol=1e-6, atol=1e-12)   # atol: lin4 has harmonic magnitudes at round-off level
    if J is not None:
        assert J.shape == tuple(g["J0_shape"])


@pytest.mark.parametrize("name", CASES)
def test_mismatch_and_jacobian_kernels_vs_golden_and_oracle(name):
    """hpf_mismatch / hpf_jacobian at iteration 0 (vs the reference's f0, J0) and along the golden trajectory
    (vs the oracle)."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    net_name, hmax, coupled = _case(name)
    st, buses, dm = _model(hp, name)
    net = o.init_network(*_paths(net_name))
    rowptr, col, Yval = o.build_admittance_matrices(net, st.HARMONICS)
    mdl = o.Model(net, st.HARMONICS, rowptr, col, Yval, o.import_Norton_Equivalents(net, st.HARMONICS, coupled, INPUTS),
                  coupled)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        traj = g["V_traj"]
        for it in sorted({0, 1, len(traj) // 2, len(traj) - 1}):
            Vm, Va = traj[it][:, 0].copy(), traj[it][:, 1].copy()
            dm.set_state(Vm, Va)
            f, err = dm.mismatch()
            J = dm.jacobian(0)
            f_o, e_o = o.harmonic_mismatch(mdl, Vm.copy(), Va.copy())
            J_o = o.build_harmonic_jacobian(mdl, Vm.copy(), Va.copy()).toarray()
            fs = max(1.0, np.abs(f_o).max())
            assert np.abs(f[0] - f_o).max() <= 1e-12 * fs
            assert abs(err[0] - e_o) <= 1e-12 * fs
            # (lin4 ends with harmonic magnitudes of exactly 0 -> U/V_m = 0/0 in the reference formula too, HG:405)
            assert np.array_equal(np.isnan(J), np.isnan(J_o))
            fin = ~np.isnan(J_o)
            assert np.abs(J[fin] - J_o[fin]).max() <= 1e-12 * np.abs(J_o[fin]).max()
            if it == 0:
                assert np.abs(f[0] - g["f0"]).max() <= 1e-12 * fs
                Jg = np.zeros(tuple(g["J0_shape"]))
                np.add.at(Jg, (g["J0_row"], g["J0_col"]), g["J0_data"])
                assert np.abs(J - Jg).max() <= 1e-12 * np.abs(Jg).max()
    finally:
        dm.close()


@pytest.mark.parametrize("name", CASES)
def test_csr_jacobian_equals_the_dense_target_and_the_reference_pattern(name):
    """hpf_jacobian_csr (the form the reference returns, HG:469-472): the same values as the dense target bit for bit, the
    reference's own stored-entry set (its J0: equal nnz, equal (row, column) pairs), columns ascending inside a row."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    st, buses, dm = _model(hp, name)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        traj = g["V_traj"]
        for it in (0, len(traj) // 2):
            dm.set_state(traj[it][:, 0].copy(), traj[it][:, 1].copy())
            Jc = dm.jacobian_csr(0)
            Jd = dm.jacobian(0)
            assert Jc.nnz == dm.jacobian_nnz() == len(g["J0_data"])
            assert np.array_equal(np.nan_to_num(Jc.toarray(), nan=7.0), np.nan_to_num(Jd, nan=7.0))
            for r in range(Jc.shape[0]):
                assert np.all(np.diff(Jc.indices[Jc.indptr[r]:Jc.indptr[r + 1]]) > 0)
            if it == 0:
                C = Jc.tocoo()
                o1, o2 = np.lexsort((C.col, C.row)), np.lexsort((g["J0_col"], g["J0_row"]))
                assert np.array_equal(C.row[o1], g["J0_row"][o2]) and np.array_equal(C.col[o1], g["J0_col"][o2])
                assert np.abs(C.data[o1] - g["J0_data"][o2]).max() <= 1e-12 * np.abs(g["J0_data"]).max()
    finally:
        dm.close()


@pytest.mark.parametrize("name", ["net1_H11_c", "net3_H51_uc"])
def test_fundamental_pf_kernels(name):
    """hpf_fund_mismatch / hpf_fund_jacobian / hpf_fund_pf vs the oracle's pf (HG:195-275)."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    net_name, hmax, coupled = _case(name)
    st, buses, dm = _model(hp, name)
    net = o.init_network(*_paths(net_name))
    rowptr, col, Yval = o.build_admittance_matrices(net, st.HARMONICS)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(None, None, n_scen=1)
        f, err = dm.mismatch(fund=True)
        n, c = net.n, net.c
        Y1 = o.y_csr(rowptr, col, Yval[0], n).toarray()
        V_vec = np.ones(n, dtype=complex)
        mis = V_vec * np.conj(Y1.dot(V_vec)) + (net.P + 1j * net.Q)
        f_o = np.r_[mis.real[1:], mis.imag[c:]]
        assert np.abs(f[0] - f_o).max() <= 1e-13 * max(1.0, np.abs(f_o).max())
        n_iter, e, hist = dm.fund_pf(1e-6, 30)
        Vm, Va = dm.get_state()
        assert int(n_iter[0]) == int(g["n_iter_f"])
        np.testing.assert_allclose(hist[0, :int(n_iter[0])], g["err_f"], rtol=1e-6, atol=1e-15)
        np.testing.assert_allclose(np.stack([Vm[0], Va[0]], 1), g["V_pf"], rtol=0, atol=1e-13)
    finally:
        dm.close()


def test_batched_scenarios_match_individual_oracle_runs():
    """S scenarios with different loads in one handle: each must equal the oracle run on that load alone."""
    hp = _hp()
    name = "net1_H11_c"
    net_name, hmax, coupled = _case(name)
    S = 5
    st, buses, dm = _model(hp, name, max_scenarios=S)
    rng = np.random.default_rng(7)
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
    scale = rng.uniform(0.5, 1.5, size=(S, len(P0)))
    scale[0] = 1.0
    try:
        dm.set_loads(P0 * scale, Q0 * scale)
        dm.set_state(None, None, n_scen=S)
        nf, ef, hf = dm.fund_pf(1e-6, 30)
        n_iter, err, hist = dm.solve(1e-4, 50)
        Vm, Va = dm.get_state()
        stats = dm.stats()
    finally:
        dm.close()
    for s in range(S):
        net = o.init_network(*_paths(net_name))
        net.P, net.Q = P0 * scale[s], Q0 * scale[s]
        r = o.hpf(net, st.HARMONICS, coupled, INPUTS)
        assert int(n_iter[s]) == r["n_iter_h"], (s, n_iter[s], r["n_iter_h"])
        Ud = Vm[s] * np.exp(1j * Va[s])
        Uo = r["Vm_raw"] * np.exp(1j * r["Va_raw"])
        assert np.abs(Ud - Uo).max() < TOL_V
        assert stats["n_iter"][s] == r["n_iter_h"] and (stats["flags"][s] & 1)
        thd = o.get_THD(np.abs(r["Vm_raw"]), net.n, len(st.HARMONICS))[:, 0].max()
        assert abs(stats["thd_max"][s] - thd) < 1e-6 * thd


def test_state_and_argument_errors():
    hp = _hp()
    from harmonic_power_flow_amd._lib import HpfError
    st, buses, dm = _model(hp, "net2_H11_c")
    try:
        assert dm.S == 0 and dm.S_max == 1   # (both asked from the library: hpf_num_scenarios / hpf_max_scenarios)
        for call in (dm.solve, dm.get_state, dm.stats, dm.mismatch, dm.fund_pf):
            with pytest.raises(HpfError) as e0:
                call()                      # loads / state not set: no batch in the handle -> HPF_E_STATE, nothing is written
            assert e0.value.code == -2
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(None, None, n_scen=1)
        with pytest.raises(HpfError):
            dm.iterate(1)                   # no valid mismatch yet
        with pytest.raises(HpfError):
            dm.jacobian(scen=3)
    finally:
        dm.close()
    with pytest.raises(HpfError) as ei:     # BLOCK_TREE on a network that is not connected from bus 0
        hp.DeviceModel(3, 3, 1, [1, 3], np.array([0, 1, 2, 3]), np.array([0, 1, 2]), np.ones((2, 3), dtype=complex),
                       np.full(3, -1), np.zeros((1, 2, 2), dtype=complex), np.zeros((1, 2), dtype=complex), 1, True, solver="block_tree")
    assert ei.value.code == -3


@pytest.mark.parametrize("name", CASES)
def test_reference_nets_through_the_bordered_block_tree_path(name):
    """net1 (4 loops), net2 / net3 (one ring) are meshed: forced onto the block-tree path they run as spanning tree + loop-closing
    lines (bordered Newton step, 1 + m virtual scenarios) and must land on the reference's converged voltages like the dense path."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    net_name, hmax, coupled = _case(name)
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, n, c = hp.init_network(*_paths(net_name), settings=st)
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, coupled, settings=st, ne_dir=INPUTS, verbose=False, solver="block_tree",
                                   return_jacobian=False)
    Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    print(f"\n{name} (bordered block-tree): it {n_iter_h} (ref {int(g['n_iter_h'])}) err {err_h:.3e} max|dU| {np.abs(Ud - Ug).max():.2e}")
    assert err_h <= 1e-4 and n_iter_h < 50
    if name not in SOLVER_SENSITIVE and name != "net2_H51_c":      # (net2 K=25 coupled: the strict case, its count holds on the dense path)
        assert n_iter_h == int(g["n_iter_h"])
    if n_iter_h == int(g["n_iter_h"]):
        assert np.abs(Ud - Ug).max() < TOL_V
    else:
        # another iteration count = another iterate below the stop threshold: what the stop rule guarantees at the stopped iterates,
        # and the north-star tolerance at the FIXED POINT -- the reference's algorithm (the oracle, bit-identical to it on this case)
        # continued until the mismatch stops falling, against this path continued by three more iterations
        assert np.abs(Ud - Ug).max() < 1e-5                   # (a stop at err_h 2e-5 sits 2.9e-6 from the fixed point on net1 K = 25: the rule bounds the mismatch)
        r = o.hpf(o.init_network(*_paths(net_name)), st.HARMONICS, coupled, INPUTS, thresh_h=1e-13, max_iter_h=int(g["n_iter_h"]) + 6)
        Uo = r["Vm"] * np.exp(1j * r["Va"])
        V2, _, _, _ = hp.hpf(buses, lines, coupled, settings=st, ne_dir=INPUTS, verbose=False, solver="block_tree", return_jacobian=False,
                             extra_iters=3)
        U2 = V2["V_m"].to_numpy() * np.exp(1j * V2["V_a"].to_numpy())
        print(f"   fixed points (oracle continued to err {r['err_h']:.1e}): max|dU| {np.abs(U2 - Uo).max():.2e}")
        assert np.abs(U2 - Uo).max() < TOL_V


def test_max_iter_and_nonconvergence_reporting():
    """A run capped at max_iter_h reports n_iter_h == max_iter_h like the reference (HG:558-559)."""
    hp = _hp()
    st = hp.Settings(H_MAX=11)
    buses, lines, m, n, c = hp.init_network(*_paths("net2"), settings=st)
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, True, max_iter_h=3, settings=st, ne_dir=INPUTS, verbose=False)
    g = np.load(os.path.join(GOLD, "net2_H11_c.npz"), allow_pickle=True)
    assert n_iter_h == 3
    assert abs(err_h - g["err_hist"][3]) <= 1e-6 * g["err_hist"][3]


# ---- synthetic radial feeders: dense (rocSOLVER) and block-tree Newton steps ------------------------------------
def _syn_model(hp, n, hmax, solver, tmp_path, max_scenarios=1, coupled=True):
    from harmonic_power_flow_amd import api, synth
    fb, fl = synth.gen(n, seed=0, outdir=str(tmp_path))
    st = hp.Settings(H_MAX=hmax)
    buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, coupled, st, INPUTS)
    dm = api._device_model(buses, Y, NE, coupled, st.HARMONICS, solver=solver, max_scenarios=max_scenarios)
    return st, buses, lines, dm, (fb, fl)


@pytest.mark.parametrize("n,solver", [(50, "dense"), (50, "block_tree"), (100, "block_tree"), (200, "block_tree"),
                                      (200, "dense")])
def test_synthetic_feeder_vs_reference_golden(n, solver, tmp_path):
    hp = _hp()
    g = np.load(os.path.join(GOLD, f"syn{n}_H11_c.npz"), allow_pickle=True)
    st, buses, lines, dm, _ = _syn_model(hp, n, 11, solver, tmp_path)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(None, None, n_scen=1)
        nf, ef, hf = dm.fund_pf(1e-6, 30)
        seed = dm.get_state()
        np.testing.assert_allclose(np.stack([seed[0][0], seed[1][0]], 1), g["V_pf"], rtol=0, atol=1e-12)
        n_iter, err, hist = dm.solve(1e-4, 50)
        Vm, Va = dm.get_state()
    finally:
        dm.close()
    from harmonic_power_flow_amd.api import _postprocess
    Vm, Va = _postprocess(Vm[0], Va[0])
    Ud = Vm * np.exp(1j * Va)
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    print(f"\nsyn{n} {solver}: it {int(n_iter[0])} (ref {int(g['n_iter_h'])}) err {err[0]:.3e} max|dU| {np.abs(Ud - Ug).max():.2e}")
    assert int(n_iter[0]) == int(g["n_iter_h"])
    assert np.abs(Ud - Ug).max() < TOL_V


@pytest.mark.parametrize("coupled,hmax", [(True, 11), (False, 11), (True, 27)])
def test_block_tree_step_equals_dense_step(tmp_path, coupled, hmax):
    """Three Newton steps from the pf seed: block-tree elimination (dense MFMA blocks + 2x2 linear subtrees; uncoupled =
    everything 2x2) vs rocSOLVER LU on the same Jacobian."""
    hp = _hp()
    out = {}
    for solver in ("dense", "block_tree"):
        st, buses, lines, dm, _ = _syn_model(hp, 100, hmax, solver, tmp_path, coupled=coupled)
        try:
            dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
            dm.set_state(None, None, n_scen=1)
            dm.fund_pf(1e-6, 30)                       # dense: rocSOLVER; block_tree: 2x2 elimination along the tree
            v0 = dm.get_state()
            if "dense" in out:                         # same start for the step comparison (the pf seeds agree to round-off)
                np.testing.assert_allclose(v0[0], out["dense"][0][0], rtol=0, atol=1e-13)
                np.testing.assert_allclose(v0[1], out["dense"][0][1], rtol=0, atol=1e-13)
                v0 = out["dense"][0]
                dm.set_state(v0[0], v0[1])
            dm.mismatch()
            dm.iterate(1)
            dm.sync()
            v1 = dm.get_state()
            e1 = dm.mismatch()[1][0]
            dm.iterate(2)
            dm.sync()
            out[solver] = (v0, v1, e1, dm.get_state())
        finally:
            dm.close()
    (v0d, v1d, ed, v3d), (v0b, v1b, eb, v3b) = out["dense"], out["block_tree"]
    step = np.abs(v1d[0] - v0d[0]).max()
    assert step > 1e-3
    assert np.abs(v1d[0] - v1b[0]).max() <= 1e-10 * max(1.0, step)
    assert np.abs(v1d[1] - v1b[1]).max() <= 1e-10 * max(1.0, np.abs(v1d[1] - v0d[1]).max())
    assert abs(ed - eb) <= 1e-8 * ed
    # three steps of a chaotic iteration amplify the solvers' rounding differences; this is only a sanity bound
    assert np.abs(v3d[0] - v3b[0]).max() <= 1e-5 * max(1.0, np.abs(v3d[0]).max())


def test_auto_solver_and_api_on_radial_feeder(tmp_path):
    hp = _hp()
    from harmonic_power_flow_amd import synth
    fb, fl = synth.gen(50, seed=0, outdir=str(tmp_path))
    st = hp.Settings(H_MAX=11)
    res = hp.solve(fb, fl, coupled=True, settings=st, ne_dir=INPUTS)
    g = np.load(os.path.join(GOLD, "syn50_H11_c.npz"), allow_pickle=True)
    assert res["details"]["solver"] == "block_tree"
    assert res["n_iter_h"] == int(g["n_iter_h"]) and res["converged"]
    V = res["V"]
    Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    assert np.abs(Ud - Ug).max() < TOL_V
    np.testing.assert_allclose(res["THD"].to_numpy(), g["THD"], rtol=1e-6)


def test_scenario_batch_block_tree_matches_oracle(tmp_path):
    """Monte-Carlo load scenarios (BASELINE config 4 shape, small): each scenario equals its own oracle run."""
    hp = _hp()
    from harmonic_power_flow_amd import synth
    S, n = 6, 50
    st, buses, lines, dm, (fb, fl) = _syn_model(hp, n, 11, "block_tree", tmp_path, max_scenarios=S)
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
    scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
    try:
        dm.set_loads(P0 * scale, Q0 * scale)
        dm.set_state(None, None, n_scen=S)
        dm.fund_pf(1e-6, 30)
        n_iter, err, hist = dm.solve(1e-4, 50)
        Vm, Va = dm.get_state()
        stats = dm.stats()
        dm.mismatch(want_f=False)                   # one more Newton iteration on every scenario, for the shallow stops below
        dm.iterate(1)
        Vm2, Va2 = dm.get_state()
    finally:
        dm.close()
    for s in range(S):
        net = o.init_network(fb, fl)
        net.P, net.Q = P0 * scale[s], Q0 * scale[s]
        r = o.hpf(net, st.HARMONICS, True, INPUTS)
        conv_o = r["n_iter_h"] < 50
        assert bool(stats["flags"][s] & 1) == conv_o
        if conv_o:
            assert int(n_iter[s]) == r["n_iter_h"], (s, int(n_iter[s]), r["n_iter_h"])
            Ud = Vm[s] * np.exp(1j * Va[s])
            Uo = r["Vm_raw"] * np.exp(1j * r["Va_raw"])
            if r["err_h"] <= 1e-7:
                assert np.abs(Ud - Uo).max() < TOL_V
            else:
                # The stop rule (1e-4) left the reference's last iterate shallow: what differs between two linear solvers there is
                # the rounding of the last step times the remaining distance (solver-sensitive cases).  Compare at
                # equal depth instead: one more Newton iteration on both sides.
                assert np.abs(Ud - Uo).max() < 1e-6
                r2 = o.hpf_from_model(r["model"], r["Vm_raw"].copy(), r["Va_raw"].copy(), thresh_h=0.0, max_iter_h=1)
                Ud = Vm2[s] * np.exp(1j * Va2[s])
                Uo = r2["Vm_raw"] * np.exp(1j * r2["Va_raw"])
                print(f"\nscenario {s}: reference stopped at {r['err_h']:.1e}; after one more iteration max|dU| {np.abs(Ud - Uo).max():.2e}")
                assert np.abs(Ud - Uo).max() < TOL_V


@pytest.mark.parametrize("n,hmax", [(100, 11), (200, 11), (1000, 51)])
def test_csr_jacobian_of_the_synthetic_feeders_vs_reference_checksums(n, hmax, tmp_path):
    """The reference's first Jacobian of syn100 / syn200 and of the HEADLINE feeder (syn1000, K = 25: N = 51 998, nnz 1 221 740; held
    as nnz, J w, J^T w and row sums of |J| by oracle/make_golden.py) against hpf_jacobian_csr of a BLOCK_TREE handle at the reference's
    own post-pf state: equal nnz, sums at 1e-12 -- no dense N x N anywhere."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, f"syn{n}_H{hmax}_c.npz"), allow_pickle=True)
    st, buses, lines, dm, _ = _syn_model(hp, n, hmax, "block_tree", tmp_path)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(g["V_it0"][:, 0].copy(), g["V_it0"][:, 1].copy())
        J = dm.jacobian_csr(0)
    finally:
        dm.close()
    check_jacobian_checksums(J, g)


def test_jacobian_of_the_last_iteration_at_the_headline_size(tmp_path):
    """hpf() returns the Jacobian of its LAST iteration at every size (HG:537,560): on the 1 000-bus x 26-harmonic feeder the solve keeps
    the state its last Newton step started from and hpf_jacobian_csr_last assembles there -- equal, entry for entry, to
    hpf_jacobian_csr at that iterate of the recorded trajectory; the current state is left untouched."""
    hp = _hp()
    st, buses, lines, dm, _ = _syn_model(hp, 1000, 51, "block_tree", tmp_path)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(None, None, n_scen=1)
        dm.fund_pf(1e-6, 30)
        dm.set_option("keep_previous_state", 1)
        n_iter, err, hist, Vt, At = dm.solve(1e-4, 50, trace=True)
        k = int(n_iter[0])
        Vm_end, Va_end = dm.get_state()
        Jl = dm.jacobian_csr(0, last=True)
        Vm2, Va2 = dm.get_state()
        assert np.array_equal(Vm_end, Vm2) and np.array_equal(Va_end, Va2)
        dm.set_state(Vt[0, k - 1].copy(), At[0, k - 1].copy())
        Jp = dm.jacobian_csr(0)
        dm.set_state(Vt[0, 0].copy(), At[0, 0].copy())
        J0 = dm.jacobian_csr(0)
    finally:
        dm.close()
    assert Jl.nnz == Jp.nnz == 1221740 and Jl.shape == (51998, 51998)
    assert np.array_equal(Jl.indptr, Jp.indptr) and np.array_equal(Jl.indices, Jp.indices) and np.array_equal(Jl.data, Jp.data)
    assert abs(Jl - J0).max() > 1e-3 * abs(Jl).max()
    # ... and through the reference's call shape
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
    assert J is not None and J.format == "csr" and J.nnz == 1221740 and n_iter_h == k
    assert np.array_equal(J.indices, Jl.indices) and np.abs(J.data - Jl.data).max() <= 1e-9 * np.abs(Jl.data).max()


def test_headline_feeder_syn1000_vs_reference_golden(tmp_path):
    """BASELINE config 3: 1 000 buses x 25 harmonics, coupled, one scenario, block-tree Newton step, against the
    reference's own converged voltages (captured by oracle/make_golden.py; 27 iterations, err 7.047e-10)."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, "syn1000_H51_c.npz"), allow_pickle=True)
    st, buses, lines, dm, _ = _syn_model(hp, 1000, 51, "block_tree", tmp_path)
    try:
        dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
        dm.set_state(None, None, n_scen=1)
        nf, ef, hf = dm.fund_pf(1e-6, 30)
        seed = dm.get_state()
        np.testing.assert_allclose(np.stack([seed[0][0], seed[1][0]], 1), g["V_pf"], rtol=0, atol=1e-12)
        f, err0 = dm.mismatch()
        assert np.abs(f[0] - g["f0"]).max() <= 1e-12 * np.abs(g["f0"]).max()
        # The FIRST Newton step of the block-tree path (fused assembly inside the factor kernels + elimination on the tree) against the
        # reference's own first iterate (V_it1: build_harmonic_jacobian + spsolve + update, HG:537-539): within 1e-9 of the step.
        n1, e1, h1 = dm.solve(1e-4, 1)
        Vm1, Va1 = dm.get_state()
        step = max(np.abs(g["V_it1"][:, 0] - g["V_it0"][:, 0]).max(), np.abs(g["V_it1"][:, 1] - g["V_it0"][:, 1]).max())
        d1 = max(np.abs(Vm1[0] - g["V_it1"][:, 0]).max(), np.abs(Va1[0] - g["V_it1"][:, 1]).max())
        print(f"\nsyn1000 first iterate vs the reference's: max dev {d1:.2e} on a step of {step:.2e} ({d1 / step:.1e} of the step)")
        assert d1 <= 1e-9 * step
        assert abs(h1[0, 1] - g["err_hist"][1]) <= 1e-9 * g["err_hist"][1]
        dm.set_state(seed[0], seed[1])
        n_iter, err, hist = dm.solve(1e-4, 50)
        Vm, Va = dm.get_state()
        # The trajectory of this case is chaotic for ~20 iterations (solver-sensitive cases), so WHERE below the
        # 1e-4 stop threshold the last iterate lands differs between linear solvers: the reference's lands at 7e-10.  If ours
        # stops shallower than 1e-7, compare at equal depth: one more Newton iteration from the stopped state.
        polished = err[0] > 1e-7
        if polished:
            dm.mismatch(want_f=False)
            dm.iterate(1)
            Vm2, Va2 = dm.get_state()
    finally:
        dm.close()
    from harmonic_power_flow_amd.api import _postprocess
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    Vm, Va = _postprocess(Vm[0], Va[0])
    Ud = Vm * np.exp(1j * Va)
    ge = g["err_hist"]
    print(f"\nsyn1000 block_tree: it {int(n_iter[0])} (ref {int(g['n_iter_h'])}) err {err[0]:.3e} "
          f"max|dU| {np.abs(Ud - Ug).max():.2e}; err_hist rel dev first 5: "
          + " ".join("%.1e" % (abs(hist[0, i] - ge[i]) / ge[i]) for i in range(5)))
    assert int(n_iter[0]) < 50 and err[0] <= 1e-4
    # the mismatch history against the reference's: the first three entries (initial state, iterations 1 and 2) agree at 1e-9; from
    # iteration 3 on this trajectory amplifies a rounding-level difference of the linear solve by 1e4 per iteration (err 1 067 -> 4 943,
    # steps of radians: DESIGN.md "trajectory sensitivity" -- SuperLU vs LAPACK inside the reference's own arithmetic differ as much),
    # so entries 3 and 4 are held at 1e-5
    for i, tol in enumerate((1e-12, 1e-9, 1e-9, 1e-5, 1e-5)):
        assert abs(hist[0, i] - ge[i]) <= tol * ge[i], (i, hist[0, i], ge[i])
    assert np.abs(Ud - Ug).max() < 1e-6          # what the stop rule itself guarantees on this feeder
    if polished:
        Vm, Va = _postprocess(Vm2[0], Va2[0])
        Ud = Vm * np.exp(1j * Va)
        print(f"   stopped at {err[0]:.1e} > 1e-7: after one more iteration max|dU| {np.abs(Ud - Ug).max():.2e}")
    assert np.abs(Ud - Ug).max() < TOL_V
    assert np.abs(Vm - g["V_final"][:, 0]).max() < TOL_V


def test_block_pivoting_option_gives_same_solution(tmp_path):
    """BLOCK_TREE with partial pivoting (wave Gauss-Jordan) vs the default MFMA static-block inversion: same voltages."""
    hp = _hp()
    res = {}
    for piv in (0, 1):
        st, buses, lines, dm, _ = _syn_model(hp, 100, 11, "block_tree", tmp_path)
        try:
            dm.set_option("block_pivoting", piv)
            dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
            dm.set_state(None, None, n_scen=1)
            dm.fund_pf(1e-6, 30)
            n_iter, err, _ = dm.solve(1e-4, 50)
            res[piv] = (int(n_iter[0]), dm.get_state())
        finally:
            dm.close()
    assert res[0][0] == res[1][0] == 17
    U0 = res[0][1][0][0] * np.exp(1j * res[0][1][1][0])
    U1 = res[1][1][0][0] * np.exp(1j * res[1][1][1][0])
    assert np.abs(U0 - U1).max() < 1e-10


@pytest.mark.parametrize("hmax,pivoting", [(99, 0), (99, 1), (59, 0), (75, 0)])
def test_large_blocks_k49_match_dense(tmp_path, hmax, pivoting):
    """BASELINE config 5 shape in small: K = 49 harmonics (b = 100), and K = 29 / 37 (b = 60 / 76 padded to 100): the multi-wave
    MFMA factor kernel k_factor_q<100> (7 wavefronts per bus, rows spread over two waves) + contracted tree + 2x2 kernels -- and,
    with block_pivoting = 1, the generic 256-thread pivoted kernels on the plain tree -- against the dense rocSOLVER path on a
    60-bus feeder: first Newton step and converged voltages."""
    hp = _hp()
    out = {}
    for solver in ("dense", "block_tree"):
        st, buses, lines, dm, _ = _syn_model(hp, 60, hmax, solver, tmp_path)
        try:
            assert dm.Hn == (hmax + 1) // 2
            if solver == "block_tree":
                dm.set_option("block_pivoting", pivoting)
            dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
            dm.set_state(None, None, n_scen=1)
            dm.fund_pf(1e-6, 30)
            v0 = dm.get_state()
            if "dense" in out:
                dm.set_state(*out["dense"][0])
            dm.mismatch()
            dm.iterate(1)
            dm.sync()
            v1 = dm.get_state()
            dm.set_state(*(out["dense"][0] if "dense" in out else v0))
            n_iter, err, _ = dm.solve(1e-4, 50)
            dm.mismatch(want_f=False)
            dm.iterate(2)
            dm.sync()
            out[solver] = (v0, v1, dm.get_state(), int(n_iter[0]), float(err[0]), dm.stats())
        finally:
            dm.close()
    (v0d, v1d, vfd, itd, ed, _), (v0b, v1b, vfb, itb, eb, stb) = out["dense"], out["block_tree"]
    step = np.abs(v1d[0] - v0d[0]).max()
    assert step > 1e-3
    assert np.abs(v1d[0] - v1b[0]).max() <= 1e-9 * max(1.0, step)
    assert np.abs(v1d[1] - v1b[1]).max() <= 1e-9 * max(1.0, np.abs(v1d[1] - v0d[1]).max())
    assert ed <= 1e-4 and eb <= 1e-4 and (stb["flags"][0] & (8 | 16 | 32)) == 0
    Ud = vfd[0][0] * np.exp(1j * vfd[1][0])
    Ub = vfb[0][0] * np.exp(1j * vfb[1][0])
    print(f"\nH_MAX={hmax} pivoting={pivoting}: dense {itd} it, block_tree {itb} it, fixed points differ by {np.abs(Ud - Ub).max():.2e}")
    assert np.abs(Ud - Ub).max() < TOL_V


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["net2_H11_c", "net1_H11_uc", "net3_H51_c"])
def test_update_harmonic_state_vec_vs_reference_first_step(name):
    """`update_harmonic_state_vec(J, x, f)` (HG:476-479) as a standalone call: with the reference's own iteration-0 Jacobian
    and mismatch (goldens) the GPU solve must land on the reference's iteration-1 state (its SuperLU step) to solver rounding."""
    hp = _hp()
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    import scipy.sparse as sp
    J = sp.csr_matrix((g["J0_data"], (g["J0_row"], g["J0_col"])), shape=tuple(g["J0_shape"]))
    c = int(g["c"]) if "c" in g.files else 1
    Vm0, Va0 = g["V_traj"][0][:, 0], g["V_traj"][0][:, 1]
    x0 = np.append(Va0[1:], Vm0[c:])
    x1 = hp.update_harmonic_state_vec(J, x0, g["f0"])
    Vm1, Va1 = g["V_traj"][1][:, 0], g["V_traj"][1][:, 1]
    ref = np.append(Va1[1:], Vm1[c:])
    assert x1.shape == ref.shape
    assert np.abs(x1 - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_block_tree_with_pv_buses_matches_dense(tmp_path):
    """Radial feeder with PV buses (c = 3: no V_m unknown / Q equation at the fundamental of buses 1, 2): the tree path's identity
    padding (2x2 kernels, contracted chains, Gauss-Jordan blocks, constant-inverse leaves) against the dense rocSOLVER path --
    fundamental power flow seed, first Newton step, converged voltages."""
    hp = _hp()
    from harmonic_power_flow_amd import api, synth
    fb, fl = synth.gen(120, seed=3, outdir=str(tmp_path))
    rows = open(fb).read().splitlines()
    for bid in (2, 3):                                  # IDs 2, 3 -> PV generators (reference dialect of net3_buses.csv)
        cols = rows[bid].split(";")
        cols[1], cols[2], cols[4], cols[5] = "PV", "gen_%d" % bid, "-150", "0"
        rows[bid] = ";".join(cols)
    open(fb, "w").write("\n".join(rows) + "\n")
    st = hp.Settings(H_MAX=11)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    assert c == 3
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    out = {}
    for solver in ("dense", "block_tree"):
        dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver)
        try:
            dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
            dm.set_state(None, None, n_scen=1)
            dm.fund_pf(1e-6, 30)
            v0 = dm.get_state()
            if "dense" in out:
                np.testing.assert_allclose(v0[0], out["dense"][0][0], rtol=0, atol=1e-12)
                np.testing.assert_allclose(v0[1], out["dense"][0][1], rtol=0, atol=1e-12)
                dm.set_state(*out["dense"][0])
            dm.mismatch()
            dm.iterate(1)
            dm.sync()
            v1 = dm.get_state()
            dm.set_state(*(out["dense"][0] if "dense" in out else v0))
            n_iter, err, _ = dm.solve(1e-4, 50)
            out[solver] = (v0, v1, dm.get_state(), int(n_iter[0]), float(err[0]))
        finally:
            dm.close()
    (v0d, v1d, vfd, itd, ed), (v0b, v1b, vfb, itb, eb) = out["dense"], out["block_tree"]
    assert ed <= 1e-4 and eb <= 1e-4 and itd < 50 and itb < 50
    step = np.abs(v1d[0] - v0d[0]).max()
    assert np.abs(v1d[0] - v1b[0]).max() <= 1e-9 * max(1.0, step)
    assert np.abs(v1d[1] - v1b[1]).max() <= 1e-9 * max(1.0, np.abs(v1d[1] - v0d[1]).max())
    from harmonic_power_flow_amd.api import _postprocess
    Ud = (lambda a: a[0] * np.exp(1j * a[1]))(_postprocess(vfd[0][0], vfd[1][0]))
    Ub = (lambda a: a[0] * np.exp(1j * a[1]))(_postprocess(vfb[0][0], vfb[1][0]))
    print(f"\nPV feeder: dense {itd} it (err {ed:.2e}), block_tree {itb} it (err {eb:.2e}), max|dU| {np.abs(Ud - Ub).max():.2e}")
    assert np.abs(Ud - Ub).max() < TOL_V


@pytest.mark.gpu
def test_two_nonlinear_device_types_vs_oracle(tmp_path):
    """Two Norton device types with different tables on one radial feeder (HG:285: one Norton entry per unique component):
    the per-bus table lookup of the mismatch, the 2x2 / Gauss-Jordan kernels and the per-bus constant leaf images, block-tree
    and dense path, against the oracle on the same inputs.  The second type is the smps table scaled by 0.6 (rotated by 0.3 rad)."""
    import shutil
    import pandas as pd
    hp = _hp()
    from harmonic_power_flow_amd import api, synth
    ne_dir = tmp_path / "ne"
    ne_dir.mkdir()
    shutil.copy(os.path.join(INPUTS, "smps_NE.csv"), ne_dir / "smps_NE.csv")
    d = pd.read_csv(os.path.join(INPUTS, "smps_NE.csv"), index_col=["Parameter", "Frequency"])
    f = 0.6 * np.exp(0.3j)
    def scaled(v):
        z = complex(v.strip("()")) * f
        return "(%.17g%+.17gj)" % (z.real, z.imag)
    d2 = d.apply(lambda col: col.apply(scaled))
    d2.to_csv(ne_dir / "led_NE.csv")
    fb, fl = synth.gen(60, seed=5, outdir=str(tmp_path))
    rows = open(fb).read().splitlines()
    for i in range(1, len(rows)):
        cols = rows[i].split(";")
        if cols[1] == "nonlinear" and i % 2 == 0:
            cols[2] = "led"
            rows[i] = ";".join(cols)
    open(fb, "w").write("\n".join(rows) + "\n")
    st = hp.Settings(H_MAX=11)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    assert set(buses.component[buses.type == "nonlinear"]) == {"smps", "led"}
    # at the reference's stop rule (1e-4) the last iterate may be shallow, and what two linear solvers differ by there is rounding
    # times the remaining distance (solver-sensitive cases): bound that loosely, and compare tightly where both
    # sides have converged all the way (threshold 1e-10)
    for thresh, tol in ((1e-4, 1e-6), (1e-10, TOL_V)):
        r = o.hpf(o.init_network(fb, fl), st.HARMONICS, True, str(ne_dir), thresh_h=thresh)
        Uo = r["Vm"] * np.exp(1j * r["Va"])
        for solver in ("dense", "block_tree"):
            V, err_h, n_iter_h, _ = hp.hpf(buses, lines, True, thresh_h=thresh, settings=st, ne_dir=str(ne_dir), solver=solver,
                                           verbose=False, return_jacobian=False)
            Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
            print(f"\ntwo device types, {solver}, thresh {thresh:g}: it {n_iter_h} (oracle {r['n_iter_h']}) err {err_h:.2e} "
                  f"max|dU| {np.abs(Ud - Uo).max():.2e}")
            assert err_h <= thresh and n_iter_h < 50
            assert np.abs(Ud - Uo).max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,frac_nl,n_pv", [(40, 11, 0.2, 0), (75, 12, 0.5, 1), (90, 13, 0.1, 2), (130, 14, 0.35, 0),
                                                  (160, 15, 0.7, 1), (64, 16, 0.05, 0)])
def test_random_feeders_block_tree_vs_dense(tmp_path, n, seed, frac_nl, n_pv):
    """Topology sweep: random radial feeders with different shares of nonlinear buses (long contracted chains at low shares,
    many constant leaves at high shares) and PV buses; converged voltages of the block-tree path vs the dense rocSOLVER path."""
    hp = _hp()
    from harmonic_power_flow_amd import api, synth
    fb, fl = synth.gen(n, seed=seed, frac_nl=frac_nl, outdir=str(tmp_path))
    if n_pv:
        rows = open(fb).read().splitlines()
        for bid in range(2, 2 + n_pv):
            cols = rows[bid].split(";")
            cols[1], cols[2], cols[4], cols[5] = "PV", "gen_%d" % bid, "-120", "0"
            rows[bid] = ";".join(cols)
        open(fb, "w").write("\n".join(rows) + "\n")
    st = hp.Settings(H_MAX=15)
    buses, lines, m, nn, c = hp.init_network(fb, fl, settings=st)
    assert c == 1 + n_pv
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    out = {}
    for solver in ("dense", "block_tree"):
        dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver=solver)
        try:
            dm.set_loads(buses["P"].to_numpy(float), buses["Q"].to_numpy(float))
            dm.set_state(None, None, n_scen=1)
            dm.fund_pf(1e-6, 30)
            if "dense" in out:
                dm.set_state(*out["dense"][0])
            seed_state = dm.get_state()
            n_iter, err, _ = dm.solve(1e-4, 50)
            if err[0] > 1e-7 and n_iter[0] < 50:            # compare at equal depth (see the syn1000 test)
                dm.mismatch(want_f=False)
                dm.iterate(1)
            out[solver] = (seed_state, dm.get_state(), int(n_iter[0]), float(err[0]))
        finally:
            dm.close()
    from harmonic_power_flow_amd.api import _postprocess
    U = {k: (lambda a: a[0] * np.exp(1j * a[1]))(_postprocess(v[1][0][0], v[1][1][0])) for k, v in out.items()}
    print(f"\nn={n} nl={frac_nl} pv={n_pv}: dense {out['dense'][2]} it, block_tree {out['block_tree'][2]} it, "
          f"max|dU| {np.abs(U['dense'] - U['block_tree']).max():.2e}")
    assert out["dense"][3] <= 1e-4 and out["block_tree"][3] <= 1e-4
    assert np.abs(U["dense"] - U["block_tree"]).max() < TOL_V


@pytest.mark.gpu
def test_batch_results_bit_identical_to_single_scenario_solves(tmp_path):
    """A scenario's result must not depend on what else is in the batch, on its position, or on the scenario-group split (three
    stream groups by default): every batched solve is bit-identical to the single-scenario solve, and repeatable."""
    hp = _hp()
    from harmonic_power_flow_amd import api, synth
    fb, fl = synth.gen(300, seed=2, outdir=str(tmp_path))
    st = hp.Settings(H_MAX=51)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)

    def run(ids):
        dm = api._device_model(buses, Y, NE, True, st.HARMONICS, solver="block_tree", max_scenarios=len(ids))
        try:
            scale = np.stack([synth.scenario_scale(n, s) for s in ids])
            dm.set_loads(P0 * scale, Q0 * scale)
            dm.set_state(None, None, n_scen=len(ids))
            dm.fund_pf(1e-6, 30)
            it, err, _ = dm.solve(1e-4, 50)
            Vm, Va = dm.get_state()
        finally:
            dm.close()
        return it, Vm, Va

    ref = {s: run([s]) for s in range(6)}
    for ids in ([0, 1, 2, 3, 4], [5, 4, 3, 2, 1, 0], [3, 3, 1]):
        it, Vm, Va = run(ids)
        for j, s in enumerate(ids):
            assert it[j] == ref[s][0][0]
            assert np.array_equal(Vm[j], ref[s][1][0]) and np.array_equal(Va[j], ref[s][2][0])
    a, b = run([0, 1, 2]), run([0, 1, 2])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
@pytest.mark.parametrize("n,hmax", [(300, 51), (200, 25), (100, 11)])
def test_lazy_leaves_give_the_same_newton_steps(n, hmax, tmp_path, monkeypatch):
    """Parents that rebuild their constant-inverse leaves' Schur complements from per-model images (lazy leaves)
    and super-leaves (bordered low-rank inverse instead of Gauss-Jordan, §5a) must take the same Newton step as the tree built
    without either (HPF_LAZY=0 at hpf_create): the first iteration from the pf seed agrees at rounding level, scenario by
    scenario (b = 52, 26 -> padded 28, 12)."""
    hp = _hp()
    from harmonic_power_flow_amd import synth
    S = 3
    out = {}
    for lazy in ("1", "0"):
        monkeypatch.setenv("HPF_LAZY", lazy)
        st, buses, lines, dm, _ = _syn_model(hp, n, hmax, "block_tree", tmp_path, max_scenarios=S)
        P0, Q0 = buses["P"].to_numpy(float), buses["Q"].to_numpy(float)
        scale = np.stack([synth.scenario_scale(n, s) for s in range(S)])
        try:
            dm.set_loads(P0 * scale, Q0 * scale)
            dm.set_state(None, None, n_scen=S)
            dm.fund_pf(1e-6, 30)
            dm.mismatch(want_f=False)
            dm.iterate(1)
            out[lazy] = dm.get_state()
        finally:
            dm.close()
    dVm = np.abs(out["1"][0] - out["0"][0]).max()
    dVa = np.abs(out["1"][1] - out["0"][1]).max()
    print(f"\nn={n} H_MAX={hmax}: first step lazy vs plain leaves max|dVm| {dVm:.1e} max|dVa| {dVa:.1e}")
    assert dVm < 1e-10 and dVa < 1e-9
