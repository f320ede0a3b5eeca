"""The Python face of the drop-in boundary, called the way INTEGRATION.md §1 promises (run with -m gpu):
reference-style objects in -- the dense `Y_all` DataFrame of HG:139-143,170, the `V` DataFrame of HG:174-184, the Norton dict
of HG:278-310 -- and the reference's return objects out, against the goldens captured from the unmodified reference
(`pf` HG:244, `harmonic_mismatch` HG:360, `build_harmonic_jacobian` HG:401), plus the known-answer test of
`hcne_based_on_fuchs.py` (HF:79-131; the reference's only committed golden, V_log.json iteration 0)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLD, INPUTS

pytestmark = pytest.mark.gpu
CASES = ["net2_H11_c", "net1_H11_uc", "net3_H51_c", "net1_H51_c"]


def _hp():
    import harmonic_power_flow_amd as hp
    return hp


def _setup(hp, name):
    net_name, hs, cs = name.split("_")
    st = hp.Settings(H_MAX=int(hs[1:]))
    buses, lines, m, n, c = hp.init_network(os.path.join(INPUTS, net_name + "_buses.csv"),
                                            os.path.join(INPUTS, net_name + "_lines.csv"), settings=st)
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    idx = pd.MultiIndex.from_product([st.HARMONICS, list(range(n))], names=["harmonic", "bus"])
    Y_all = pd.DataFrame(g["Y_all"], index=idx, columns=[np.arange(n)])          # the reference's object (HG:139-143)
    return st, buses, g, Y_all, idx, cs == "c"


@pytest.mark.parametrize("name", CASES)
def test_pf_with_reference_style_admittance_frame(name):
    """pf(Y, buses) (HG:244-275) with the reference's own `Y_all` frame -> (V, err_t, n_iter_f)."""
    hp = _hp()
    st, buses, g, Y_all, idx, coupled = _setup(hp, name)
    V, err_t, n_iter_f = hp.pf(Y_all, buses, settings=st, verbose=False)
    assert n_iter_f == int(g["n_iter_f"])
    assert list(V.columns) == ["V_m", "V_a"] and V.index.equals(idx)
    np.testing.assert_allclose(V.to_numpy(), g["V_pf"], rtol=0, atol=1e-13)
    assert sorted(err_t) == list(range(n_iter_f))
    np.testing.assert_allclose([err_t[i] for i in range(n_iter_f)], g["err_f"], rtol=1e-6, atol=1e-15)


@pytest.mark.parametrize("name", CASES)
def test_harmonic_mismatch_with_reference_style_objects(name):
    """harmonic_mismatch(V, Y, buses, NE) (HG:360-390) -> (f, err_h): the reference's f0 at the pf seed, and the err_h of every
    iterate of the reference's trajectory."""
    hp = _hp()
    st, buses, g, Y_all, idx, coupled = _setup(hp, name)
    NE = hp.import_Norton_Equivalents(buses, coupled, st, INPUTS)
    traj = g["V_traj"]
    V0 = pd.DataFrame(traj[0], index=idx, columns=["V_m", "V_a"])
    f, err_h = hp.harmonic_mismatch(V0, Y_all, buses, NE, settings=st)
    fs = max(1.0, np.abs(g["f0"]).max())
    assert f.shape == g["f0"].shape
    assert np.abs(f - g["f0"]).max() <= 1e-12 * fs
    assert abs(err_h - g["err_hist"][0]) <= 1e-12 * fs
    for it in (1, len(traj) - 1):
        Vi = pd.DataFrame(traj[it], index=idx, columns=["V_m", "V_a"])
        _, e = hp.harmonic_mismatch(Vi, Y_all, buses, NE, settings=st)
        assert abs(e - g["err_hist"][it]) <= 1e-9 * max(g["err_hist"][it], 1e-3)


@pytest.mark.parametrize("name", CASES)
def test_build_harmonic_jacobian_with_reference_style_objects(name):
    """build_harmonic_jacobian(V, Y, NE, coupled) (HG:401-473) -> scipy CSR in the reference's row / column order = J0."""
    import scipy.sparse as sp
    hp = _hp()
    st, buses, g, Y_all, idx, coupled = _setup(hp, name)
    NE = hp.import_Norton_Equivalents(buses, coupled, st, INPUTS)
    V0 = pd.DataFrame(g["V_traj"][0], index=idx, columns=["V_m", "V_a"])
    J = hp.build_harmonic_jacobian(V0, Y_all, NE, coupled, buses=buses)
    assert sp.issparse(J) and J.format == "csr" and J.shape == tuple(g["J0_shape"])
    Jg = sp.csr_matrix((g["J0_data"], (g["J0_row"], g["J0_col"])), shape=tuple(g["J0_shape"]))
    assert abs(J - Jg).max() <= 1e-12 * abs(Jg).max()
    # one reference-style Newton step from these pieces (HG:537-539): x - J^-1 f lands on the reference's next iterate
    f, _ = hp.harmonic_mismatch(V0, Y_all, buses, NE, settings=st)
    c = int(g["c"])
    x0 = hp.harmonic_state_vector(V0, c=c)
    x1 = hp.update_harmonic_state_vec(J, x0, f)
    V1 = g["V_traj"][1]
    ref = np.append(V1[1:, 1], V1[c:, 0])
    assert np.abs(x1 - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


def test_fuchs_4bus_known_answer():
    """hcne_based_on_fuchs.py:36-131 = example_hpf_fuchs.py:77-130: the 4-bus ring (p.u. data of HF:44-54 written as SI values
    in the reference's CSV dialect: tests/golden/inputs/fuchs4_*.csv), fundamental Newton-Raphson.  The product's `pf` must land
    on the reference's committed golden, V_log.json iteration 0 (10 decimals), and on the captured HF run."""
    hp = _hp()
    st = hp.Settings(H_MAX=5)
    buses, lines, m, n, c = hp.init_network(os.path.join(INPUTS, "fuchs4_buses.csv"), os.path.join(INPUTS, "fuchs4_lines.csv"),
                                            settings=st)
    assert (m, n, c) == (3, 4, 1)
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    V, err_t, n_iter_f = hp.pf(Y, buses, settings=st, verbose=False)
    with open(os.path.join(GOLD, "v_log_iter0.json")) as fh:
        rows = [r for r in json.load(fh)["rows"] if r["harmonic"] == 1]
    Vf = V.loc[1].to_numpy()
    for k, r in enumerate(rows):
        assert abs(Vf[k, 0] - r["V_m"]) < 1e-9 and abs(Vf[k, 1] - r["V_a"]) < 1e-9, (k, Vf[k], r)
    with open(os.path.join(GOLD, "hf_fuchs.json")) as fh:
        hf = np.array(json.load(fh)["V_final"])
    np.testing.assert_allclose(Vf, hf[:4], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", ["net2_H11_c", "net3_H51_uc", "net1_H11_c"])
def test_hpf_returns_the_jacobian_of_its_last_iteration(name):
    """hpf() -> (V, err_h, n_iter_h, J) (HG:511-560): J is the Jacobian built in the LAST iteration, i.e. at the iterate before the
    last update (HG:537) -- kept on the device by the solve (hpf_jacobian_last), compared with build_harmonic_jacobian at the
    reference's own second-to-last iterate."""
    hp = _hp()
    st, buses, g, Y_all, idx, coupled = _setup(hp, name)
    net_name = name.split("_")[0]
    lines = hp.init_network(os.path.join(INPUTS, net_name + "_buses.csv"), os.path.join(INPUTS, net_name + "_lines.csv"), settings=st)[1]
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, coupled, settings=st, ne_dir=INPUTS, verbose=False)
    assert n_iter_h == int(g["n_iter_h"]) and J is not None and J.shape == tuple(g["J0_shape"])
    NE = hp.import_Norton_Equivalents(buses, coupled, st, INPUTS)
    Vprev = pd.DataFrame(g["V_traj"][n_iter_h - 1], index=idx, columns=["V_m", "V_a"])
    Jref = hp.build_harmonic_jacobian(Vprev, Y_all, NE, coupled, buses=buses)
    assert abs(J - Jref).max() <= 1e-6 * abs(Jref).max()
    J0 = hp.build_harmonic_jacobian(pd.DataFrame(g["V_traj"][0], index=idx, columns=["V_m", "V_a"]), Y_all, NE, coupled, buses=buses)
    assert abs(J - J0).max() > 1e-3 * abs(Jref).max()            # (and it is not the iteration-0 Jacobian)


def test_extra_iterations_reach_the_fixed_point(tmp_path):
    """hpf(extra_iters=2): the iterate the reference's stop rule leaves is up to 1e-6 away from the fixed point and depends on the
    linear solver's rounding; two more Newton iterations land on the fixed point, where the oracle (continued the same way) agrees
    to 1e-8 whatever the solver path."""
    import hpf_oracle as o
    hp = _hp()
    from harmonic_power_flow_amd import synth
    fb, fl = synth.gen(200, seed=0, outdir=str(tmp_path))
    st = hp.Settings(H_MAX=11)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    r = o.hpf(o.init_network(fb, fl), st.HARMONICS, True, INPUTS)
    r2 = o.hpf_from_model(r["model"], r["Vm_raw"].copy(), r["Va_raw"].copy(), thresh_h=0.0, max_iter_h=2)
    Vm_o, Va_o = o.postprocess(r2["Vm_raw"], r2["Va_raw"])
    Uo = Vm_o * np.exp(1j * Va_o)
    for solver in ("block_tree", "dense"):
        V, err_h, n_iter_h, _ = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, solver=solver,
                                       return_jacobian=False, extra_iters=2)
        assert n_iter_h == r["n_iter_h"] and abs(err_h - r["err_h"]) <= 1e-3 * r["err_h"]      # the reference's stop is what is reported
        Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
        assert np.abs(Ud - Uo).max() < 1e-10


def test_fuchs_script_last_newton_step_through_hpf_dense_solve():
    """hcne_based_on_fuchs.py:345-346 (`U_new = U - inv(J_5).dot(dM)`): the script's OWN last linear system (14 x 14 J_5, dM, U captured
    from the unmodified script by oracle/make_golden.py; cond(J_5) = 4.4e4, |U| up to 2.5e2 rad) through update_harmonic_state_vec's
    device path (hpf_dense_solve: rocSOLVER LU) lands on the script's U_new."""
    import json
    hp = _hp()
    with open(os.path.join(GOLD, "hf_fuchs.json")) as fh:
        g = json.load(fh)
    J, dM, U, U_new = (np.array(g[k + "_last"]) for k in ("J_5", "dM", "U", "U_new"))
    assert J.shape == (14, 14)
    x = hp.update_harmonic_state_vec(J, U, dM)
    assert np.abs(np.asarray(x).ravel() - U_new).max() < 1e-10 * max(1.0, np.abs(U_new).max())


def test_reference_call_shapes_reuse_a_cached_handle_bit_identically(tmp_path):
    """the reference's sweep is repeated hpf() calls (HG:511).  The call shapes borrow a device handle from an LRU keyed on
    the model (pattern, admittance values, Norton arrays, harmonics, solver, HPF_* environment): the second call on the same network creates
    nothing, other loads on the same network reuse it too, a changed admittance or an environment switch does not; results are bit-identical
    to a call without the cache; the reference-style loop build_harmonic_jacobian -> update_harmonic_state_vec -> harmonic_mismatch (HG:536-542)
    runs on ONE assembly handle."""
    import time
    hp = _hp()
    from harmonic_power_flow_amd import synth
    fb, fl = synth.gen(1000, seed=0, outdir=str(tmp_path))
    st = hp.Settings(H_MAX=51)
    buses, lines, m, n, c = hp.init_network(fb, fl, settings=st)
    hp.close_all()
    hp.handle_cache(0)
    V0, e0, it0, J0 = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
    hp.handle_cache(4)
    base = hp.handle_cache()
    t0 = time.perf_counter()
    V1, e1, it1, J1 = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
    t1 = time.perf_counter()
    V2, e2, it2, J2 = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False)
    t2 = time.perf_counter()
    V3, e3, it3, _ = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, return_jacobian=False)
    t3 = time.perf_counter()
    s = hp.handle_cache()
    assert s["misses"] - base["misses"] == 1 and s["hits"] - base["hits"] == 2 and s["held"] == 1
    for V, e, it in ((V1, e1, it1), (V2, e2, it2), (V3, e3, it3)):
        assert it == it0 and e == e0 and np.array_equal(V.to_numpy(), V0.to_numpy())
    assert np.array_equal(J1.data, J0.data) and np.array_equal(J2.data, J0.data) and np.array_equal(J2.indices, J0.indices)
    print(f"\nhpf() on syn1000 x 26 harmonics, end to end: first (creates the handle) {1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.1f} ms, "
          f"without the Jacobian export {1e3 * (t3 - t2):.1f} ms ({it0} iterations)")
    # other loads: same handle; the result equals a fresh handle's
    b2 = buses.copy()
    b2["P"] = b2["P"] * 0.9
    Va, ea, ita, _ = hp.hpf(b2, lines, True, settings=st, ne_dir=INPUTS, verbose=False, return_jacobian=False)
    assert hp.handle_cache()["hits"] - s["hits"] == 1
    hp.handle_cache(0)
    Vb, eb, itb, _ = hp.hpf(b2, lines, True, settings=st, ne_dir=INPUTS, verbose=False, return_jacobian=False)
    hp.handle_cache(4)
    assert ita == itb and ea == eb and np.array_equal(Va.to_numpy(), Vb.to_numpy()) and not np.array_equal(Va.to_numpy(), V0.to_numpy())
    # a changed line impedance is another model
    l2 = lines.copy()
    l2.loc[0, "R"] = l2.loc[0, "R"] * 1.5
    s = hp.handle_cache()
    hp.hpf(buses, l2, True, settings=st, ne_dir=INPUTS, verbose=False, return_jacobian=False)
    assert hp.handle_cache()["misses"] - s["misses"] == 1
    # the reference-style loop on one assembly handle
    Y = hp.build_admittance_matrices(buses, lines, st.HARMONICS)
    NE = hp.import_Norton_Equivalents(buses, True, st, INPUTS)
    Vs, _, _ = hp.pf(Y, buses, settings=st, verbose=False)
    s = hp.handle_cache()
    V = Vs.copy()
    f, err = hp.harmonic_mismatch(V, Y, buses, NE, settings=st)
    for _ in range(2):
        J = hp.build_harmonic_jacobian(V, Y, NE, True, buses=buses)
        x = hp.update_harmonic_state_vec(J, hp.harmonic_state_vector(V, c=c), f)
        V.iloc[1:, 1] = x[:n * 26 - 1]                          # update_harmonic_voltages, HG:484-485
        V.iloc[c:, 0] = x[n * 26 - 1:]
        f, err = hp.harmonic_mismatch(V, Y, buses, NE, settings=st)
    s2 = hp.handle_cache()
    assert s2["misses"] - s["misses"] == 1 and s2["hits"] - s["hits"] == 4
    g = np.load(os.path.join(GOLD, "syn1000_H51_c.npz"), allow_pickle=True)
    assert abs(err - g["err_hist"][2]) <= 1e-9 * g["err_hist"][2]      # two reference-style iterations land on the reference's err_h
    hp.close_all()
    assert hp.handle_cache()["held"] == 0


def test_init_network_without_csv_is_the_manual_network_of_the_reference(tmp_path):
    """init_network(from_csv=False) (HG:64-74, 97-110, 117-119): the built-in 4-bus ring with pi-model line shunts the reference's manual
    initialisers describe (its own versions raise).  Pinned by the oracle on the same numbers written as CSV files: identical frames, identical
    trajectories (the line shunts B != 0 exercise the off-by-one bus test of HG:163-168)."""
    import hpf_oracle as o
    hp = _hp()
    st = hp.Settings(H_MAX=11)
    buses, lines, m, n, c = hp.init_network(None, None, from_csv=False, settings=st)
    assert (m, n, c) == (3, 4, 1)
    fb, fl = str(tmp_path / "man_buses.csv"), str(tmp_path / "man_lines.csv")
    open(fb, "w").write("ID;type;component;S;P;Q;X_sh\n1;slack;generator;0;0;0;0.005\n2;PQ;lin_load_1;0;100;100;0\n3;PQ;lin_load_2;0;100;100;0\n"
                        "4;nonlinear;smps;0;150;100;0\n")
    open(fl, "w").write("ID;fromID;toID;R;X;G;B\n1;1;2;0.5;0.5;0;0.05\n2;2;3;1;4;0;0.1\n3;3;4;0.5;1;0;0.05\n4;4;1;0.5;1;0;0.05\n")
    b2, l2, m2, n2, c2 = hp.init_network(fb, fl, settings=st)
    for col in ("P", "Q", "X_sh"):
        assert np.array_equal(buses[col].to_numpy(), b2[col].to_numpy())
    for col in ("fromID", "toID", "R", "X", "G", "B"):
        assert np.array_equal(lines[col].to_numpy(float), l2[col].to_numpy(float))
    det = {}
    V, err_h, n_iter_h, J = hp.hpf(buses, lines, True, settings=st, ne_dir=INPUTS, verbose=False, details=det)
    r = o.hpf(o.init_network(fb, fl), st.HARMONICS, True, INPUTS)
    assert n_iter_h == r["n_iter_h"]
    k = min(n_iter_h + 1, 4)
    np.testing.assert_allclose(det["err_hist"][:k], r["err_hist"][:k], rtol=1e-9)
    if err_h <= 1e-4:
        Ud = V["V_m"].to_numpy() * np.exp(1j * V["V_a"].to_numpy())
        assert np.abs(Ud - r["Vm"] * np.exp(1j * r["Va"])).max() < 1e-8
