"""Pin the CPU oracle (oracle/hpf_oracle.py) against golden vectors captured from the unmodified reference
(oracle/make_golden.py).  Tolerances, not bit-equality: the Norton matvec goes through BLAS zgemv whose
summation order depends on the host CPU's OpenBLAS kernel (bit-identical on the capture machine)."""
import glob
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

import hpf_oracle as o

from conftest import GOLD, INPUTS

NET_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*_H*.npz"))
                   if not os.path.basename(p).startswith("syn"))

# reference facts, SURVEY.md §8(c): (n_iter_h, err_h) per case
FACTS = {"net2_H11_uc": (13, 1.825e-07), "net2_H51_uc": (13, 1.825e-07), "net3_H11_uc": (13, 1.825e-07),
         "net2_H11_c": (16, 1.301e-05), "net2_H51_c": (21, 2.419e-06), "net3_H11_c": (13, 3.221e-06),
         "net3_H51_c": (19, 6.735e-07), "net1_H11_uc": (12, 4.380e-06), "net1_H51_uc": (14, 3.683e-06),
         "net1_H11_c": (19, 2.440e-09), "net1_H51_c": (23, 4.667e-11)}


def _case(name):
    net_name, hs, cs = name.split("_")
    return net_name, int(hs[1:]), cs == "c"


def test_all_reference_cases_present():
    # 12 cases on the reference's own nets + 4 on the quirk fixtures authored for this build (quirk5, lin4)
    assert len(NET_CASES) == 16


@pytest.mark.parametrize("name", NET_CASES)
def test_oracle_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    net_name, hmax, coupled = _case(name)
    net = o.init_network(os.path.join(INPUTS, f"{net_name}_buses.csv"), os.path.join(INPUTS, f"{net_name}_lines.csv"))
    assert (net.n, net.m, net.c) == (int(g["n"]), int(g["m"]), int(g["c"]))
    H = o.harmonics_upto(hmax)
    assert list(g["harmonics"]) == H
    r = o.hpf(net, H, coupled, INPUTS, record=True)
    mdl = r["model"]
    n, Hn = net.n, len(H)
    # admittance matrices: bit-identical to the reference's dense Y_all
    Yd = np.vstack([o.y_csr(mdl.rowptr, mdl.col, mdl.Yval[q], n).toarray() for q in range(Hn)])
    assert np.array_equal(Yd, g["Y_all"])
    # Norton parameters in p.u.
    if mdl.NE:
        I_N, Y_N = list(mdl.NE.values())[0]
        assert np.array_equal(I_N, g["I_N"].ravel()) and np.array_equal(Y_N.ravel(), g["Y_N"].ravel())
    else:
        assert "I_N" not in g.files and net.m == net.n
    # fundamental power flow seed
    assert r["n_iter_f"] == int(g["n_iter_f"])
    np.testing.assert_allclose(np.stack(r["seed"], 1), g["V_pf"], rtol=0, atol=1e-14)
    # iteration-0 mismatch and Jacobian
    Vm0, Va0 = (a.copy() for a in r["traj"][0])
    f0, e0 = o.harmonic_mismatch(mdl, Vm0, Va0)
    np.testing.assert_allclose(f0, g["f0"], rtol=0, atol=1e-12 * max(1.0, abs(g["f0"]).max()))
    J0 = o.build_harmonic_jacobian(mdl, Vm0, Va0)
    Jg = sp.coo_matrix((g["J0_data"], (g["J0_row"], g["J0_col"])), shape=tuple(g["J0_shape"])).tocsr()
    assert J0.shape == Jg.shape == (mdl.N, mdl.N)
    assert abs(J0 - Jg).max() <= 1e-12 * abs(Jg).max()
    # trajectory: same iteration count, same stopping error, same final voltages (after HG:545-549)
    assert r["n_iter_h"] == int(g["n_iter_h"])
    if name in FACTS:
        assert r["n_iter_h"] == FACTS[name][0]
        assert abs(r["err_h"] - FACTS[name][1]) <= 1e-3 * FACTS[name][1]
    Uo = r["Vm"] * np.exp(1j * r["Va"])
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    assert np.abs(Uo - Ug).max() < 1e-8          # north-star parity bar
    assert np.abs(r["Vm"] - g["V_final"][:, 0]).max() < 1e-8
    thd = o.get_THD(r["Vm"], n, Hn)
    np.testing.assert_allclose(thd, g["THD"], rtol=1e-7)


def test_default_script_thd():
    """HG:596-623 default run: net2, K=25, uncoupled -> THD_F(bus 4) = 40.74696398381437 %."""
    net = o.init_network(os.path.join(INPUTS, "net2_buses.csv"), os.path.join(INPUTS, "net2_lines.csv"))
    H = o.harmonics_upto(51)
    r = o.hpf(net, H, False, INPUTS)
    thd = o.get_THD(r["Vm"], net.n, len(H))
    assert abs(thd[3, 0] * 100 - 40.74696398381437) < 1e-6


@pytest.mark.parametrize("n,it", [(50, 15), (100, 17), (200, 21)])
def test_oracle_synthetic_feeders(n, it, tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "synth", os.path.join(os.path.dirname(GOLD), "..", "harmonic-power-flow_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    g = np.load(os.path.join(GOLD, f"syn{n}_H11_c.npz"), allow_pickle=True)
    fb, fl = synth.gen(n, seed=0, outdir=str(tmp_path))
    net = o.init_network(fb, fl)
    r = o.hpf(net, o.harmonics_upto(11), True, INPUTS)
    assert r["n_iter_h"] == it == int(g["n_iter_h"])
    Uo = r["Vm"] * np.exp(1j * r["Va"])
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    assert np.abs(Uo - Ug).max() < 1e-8


def test_reference_committed_golden_fundamental_4bus():
    """The only numeric golden committed in the reference: V_log.json iteration-0 rows = result of the 4-bus
    fundamental NR shared by hcne_based_on_fuchs.py:79-131; it is reproduced by the HF run captured in
    hf_fuchs.json (10 decimals)."""
    with open(os.path.join(GOLD, "v_log_iter0.json")) as f:
        rows = [r for r in json.load(f)["rows"] if r["harmonic"] == 1]
    with open(os.path.join(GOLD, "hf_fuchs.json")) as f:
        hf = json.load(f)
    V = np.array(hf["V_final"])
    for k, r in enumerate(rows):
        assert abs(V[k, 0] - r["V_m"]) < 1e-9 and abs(V[k, 1] - r["V_a"]) < 1e-9


def test_oracle_pf_lands_on_the_reference_committed_fuchs_golden():
    """Known-answer test on the reference's ONLY committed numeric golden: the 4-bus ring of hcne_based_on_fuchs.py:36-54 (written
    as SI values in the reference's CSV dialect, tests/golden/inputs/fuchs4_*.csv) through the oracle's ingest, admittance build
    and fundamental NR (HG:244-275 = HF:79-131 in the PyPSA formulation) must give V_log.json iteration 0 (10 decimals)."""
    net = o.init_network(os.path.join(INPUTS, "fuchs4_buses.csv"), os.path.join(INPUTS, "fuchs4_lines.csv"))
    assert (net.m, net.n, net.c) == (3, 4, 1)
    H = o.harmonics_upto(5)
    rowptr, col, Yval = o.build_admittance_matrices(net, H)
    # the admittances are the p.u. values of HF:44-54
    Y1 = o.y_csr(rowptr, col, Yval[0], 4).toarray()
    assert Y1[0, 1] == -1 / (0.01 + 0.01j) and Y1[1, 2] == -1 / (0.02 + 0.08j) and Y1[3, 0] == -1 / (0.01 + 0.02j)
    Vm, Va, err_t, n_iter_f = o.pf(net, rowptr, col, Yval)
    with open(os.path.join(GOLD, "v_log_iter0.json")) as f:
        rows = [r for r in json.load(f)["rows"] if r["harmonic"] == 1]
    for k, r in enumerate(rows):
        assert abs(Vm[k] - r["V_m"]) < 1e-9 and abs(Va[k] - r["V_a"]) < 1e-9


def test_hf_oracle_reproduces_the_fuchs_script_harmonic_newton_raphson():
    """hcne_based_on_fuchs.py:185-356 (the harmonic NR of the fixed 4-bus example with the analytic load g(), HF:170-173), restated in
    oracle/hf_oracle.py, against the run of the unmodified script captured by oracle/make_golden.py -- BIT FOR BIT: both admittance
    matrices, the 3 printed fundamental errors, all 11 harmonic iterates (V_h_log), injections and printed errors, the last iteration's
    linear system (J_5, dM, U, U_new) and the final voltages, V(5, bus 4) = 0.0253363651 at -1.6765972645."""
    import hf_oracle as hf
    with open(os.path.join(GOLD, "hf_fuchs.json")) as f:
        g = json.load(f)
    for h, k in ((1, "Y_f"), (5, "Y_5")):
        assert np.array_equal(hf.admittances(h), np.array(g[k + "_re"]) + 1j * np.array(g[k + "_im"]))
    r = hf.run()
    assert (r["n_iter"], r["n_iter_h"]) == (4, 11) == (g["n_iter"], g["n_iter_h"])
    assert r["err_f_list"] == g["err_f_list"] and r["err_h_list"] == g["err_h_list"] and g["err_h_list"][0] == 200.0
    assert r["err_h"] == g["err_h"]
    L, M = np.array(g["V_h_log"]), np.array(r["V_h_log"])
    assert L.shape == M.shape == (11, 8, 2) and np.array_equal(L, M)
    assert np.array_equal(np.array(g["I_inj_log"]), np.array(r["I_inj_log"]))
    Vf = np.array(g["V_final"])
    assert np.array_equal(Vf[:, 0], r["Vm"].ravel()) and np.array_equal(Vf[:, 1], r["Va"].ravel())
    assert abs(r["Vm"][1, 3] - 0.0253363651) < 1e-10 and abs(r["Va"][1, 3] + 1.6765972645) < 1e-9
    for k in ("J_5", "dM", "U", "U_new"):
        assert np.array_equal(np.array(g[k + "_last"]), r["last"][k]), k
    assert np.array_equal(np.array(g["J_last"]), r["J_fund"])                        # the stale fundamental block (HF:261)


def test_oracle_syn1000_headline_shape(tmp_path):
    """The north-star shape (1 000 buses x 25 harmonics, coupled): 27 iterations, err 7.047e-10."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "synth", os.path.join(os.path.dirname(GOLD), "..", "harmonic-power-flow_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    g = np.load(os.path.join(GOLD, "syn1000_H51_c.npz"), allow_pickle=True)
    fb, fl = synth.gen(1000, seed=0, outdir=str(tmp_path))
    import hashlib
    assert hashlib.sha256(open(fb, "rb").read()).hexdigest().startswith("f0c6440df8e5ece3")
    assert hashlib.sha256(open(fl, "rb").read()).hexdigest().startswith("e1eeedbe982385fa")
    net = o.init_network(fb, fl)
    r = o.hpf(net, o.harmonics_upto(51), True, INPUTS)
    assert r["n_iter_h"] == 27 == int(g["n_iter_h"])
    assert abs(r["err_h"] - 7.047e-10) < 1e-12
    np.testing.assert_allclose(r["err_hist"], g["err_hist"], rtol=1e-6)
    Uo = r["Vm"] * np.exp(1j * r["Va"])
    Ug = g["V_final"][:, 0] * np.exp(1j * g["V_final"][:, 1])
    assert np.abs(Uo - Ug).max() < 1e-8
    # survey checksums of the reference run
    assert abs(r["Vm"].sum() - 11436.6393011143) < 1e-6 and abs(r["Va"].sum() - 71429.0720178031) < 1e-5


def test_config4_scenarios_held_by_the_reference_itself(tmp_path):
    """BASELINE config 4 (Monte-Carlo load scenarios of the 1 000-bus x 25-harmonic feeder): the unmodified reference was run on scenarios 0
    and 127 of the 128-scenario share (oracle/make_golden.py scenref<s>: 32 / 30 iterations, about 20 min each).  The oracle -- run live
    on scenario 127 -- reproduces that run bit for bit (pf seed, every entry of the mismatch history, the final voltages), and so do the
    oracle fixtures the GPU tests compare against (tests/golden/syn1000_H51_scen.npz) for both scenarios."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "synth", os.path.join(os.path.dirname(GOLD), "..", "harmonic-power-flow_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    f = np.load(os.path.join(GOLD, "syn1000_H51_scen.npz"))
    for s, n_it in ((0, 32), (127, 30)):
        g = np.load(os.path.join(GOLD, "syn1000_H51_scenref%d.npz" % s), allow_pickle=True)
        assert int(g["n_iter_h"]) == n_it == int(f["n_iter_%d" % s]) and int(g["scenario"]) == s
        assert np.array_equal(g["err_hist"], f["err_hist_%d" % s])
        assert np.array_equal(g["V_pf"][:1000], f["seed_fund_%d" % s])
        Vm, Va = o.postprocess(f["V_stop_%d" % s][:, 0].copy(), f["V_stop_%d" % s][:, 1].copy())
        assert np.array_equal(Vm, g["V_final"][:, 0]) and np.array_equal(Va, g["V_final"][:, 1])
    fb, fl = synth.gen(1000, seed=0, outdir=str(tmp_path))
    net = o.init_network(fb, fl)
    u = synth.scenario_scale(1000, 127)
    net.P, net.Q = net.P * u, net.Q * u
    r = o.hpf(net, o.harmonics_upto(51), True, INPUTS)
    g = np.load(os.path.join(GOLD, "syn1000_H51_scenref127.npz"), allow_pickle=True)
    assert r["n_iter_h"] == 30 and np.array_equal(np.asarray(r["err_hist"]), g["err_hist"])
    assert np.array_equal(r["Vm"], g["V_final"][:, 0]) and np.array_equal(r["Va"], g["V_final"][:, 1])
