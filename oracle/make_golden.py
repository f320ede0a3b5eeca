#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE — runs only in the build container).

Imports the *unmodified* reference `/root/reference/Harmonic Power Flow/hcne_generalized.py`
(HG) with the environment-only shims of SURVEY.md §8(c) / Appendix C, runs it on the parity
networks and on synthetic feeders, and writes small `.npz` fixtures into `tests/golden/`.
Nothing under `tests -m gpu`, `smoke()` or `bench.py` reads `/root/reference`; the GPU box
only ever sees the fixtures this script wrote.

Shims (no edits to reference source):
  1. HOME -> scratch dir holding `Git/harmonic-power-flow/Circuit Simulation/{smps,SMPS}_NE.csv`
     (HG:289-291 hard-codes that path; net2/net3 say `SMPS`, net1 says `smps`).
  2. `np.Inf = np.inf` (HG:389; removed in NumPy 2).
  3. cwd = scratch dir containing net2_*.csv (HG:596 runs the default case at import).
  4. net1 is fed as a header-normalised copy (`X_shunt`->`X_sh`, `;G;B` = `;0;0` appended) because
     HG cannot read net1's dialect (AttributeError: X_sh).

Usage:  python oracle/make_golden.py [nets] [quirks] [syn] [syn1000] [hf]      (default: nets syn hf)
"""
import contextlib
import io
import json
import os
import runpy
import shutil
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_HPF = os.path.join(REF, "Harmonic Power Flow")
GOLD = os.path.join(REPO, "tests", "golden")
SCRATCH = os.environ.get("HPF_ORACLE_SCRATCH", "/tmp/hpf_oracle_scratch")


def _setup_scratch():
    ne_dir = os.path.join(SCRATCH, "home", "Git", "harmonic-power-flow", "Circuit Simulation")
    work = os.path.join(SCRATCH, "work")
    os.makedirs(ne_dir, exist_ok=True)
    os.makedirs(work, exist_ok=True)
    src_ne = os.path.join(REF, "Circuit Simulation", "smps_NE.csv")
    shutil.copy(src_ne, os.path.join(ne_dir, "smps_NE.csv"))
    shutil.copy(src_ne, os.path.join(ne_dir, "SMPS_NE.csv"))
    for f in os.listdir(REF_HPF):
        if f.startswith("net") and f.endswith(".csv"):
            shutil.copy(os.path.join(REF_HPF, f), os.path.join(work, f))
    # shim 4: header-normalised net1
    with open(os.path.join(work, "net1_buses.csv")) as f:
        lines = f.read().splitlines()
    lines[0] = lines[0].replace("X_shunt", "X_sh")
    with open(os.path.join(work, "net1n_buses.csv"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(work, "net1_lines.csv")) as f:
        lines = [l for l in f.read().splitlines() if l.strip()]
    out = [lines[0] + ";G;B"] + [l + ";0;0" for l in lines[1:]]
    with open(os.path.join(work, "net1n_lines.csv"), "w") as f:
        f.write("\n".join(out) + "\n")
    # data fixtures (inputs): raw reference data files, committed as test inputs
    inp = os.path.join(GOLD, "inputs")
    os.makedirs(inp, exist_ok=True)
    for f in os.listdir(REF_HPF):
        if f.startswith("net") and f.endswith(".csv"):
            shutil.copy(os.path.join(REF_HPF, f), os.path.join(inp, f))
    shutil.copy(src_ne, os.path.join(inp, "smps_NE.csv"))
    os.environ["HOME"] = os.path.join(SCRATCH, "home")
    os.environ["MPLBACKEND"] = "Agg"
    os.chdir(work)
    return work


def _import_reference():
    np.Inf = np.inf                       # shim 2
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_HPF)
    with contextlib.redirect_stdout(io.StringIO()):
        import hcne_generalized as g      # runs the default case (net2, K=25, uncoupled)
    return g


def run_case(g, buses_csv, lines_csv, h_max, coupled, full=True, load_scale=None):
    """Run the reference on one case, recording the NR trajectory through wrappers.
    `load_scale` (per-bus multipliers of P and Q, applied to the p.u. columns after the reference's own ingest): a Monte-Carlo
    load scenario of BASELINE config 4 -- a sweep is repeated `hpf()` calls with other `buses.P/Q` (HG:197, HG:372)."""
    g.HARMONICS = [h for h in range(1, h_max + 1, 2)]
    g.HARMONICS_FREQ = [g.NET_FREQ * i for i in g.HARMONICS]
    g.buses, g.lines, g.m, g.n, g.c = g.init_network(buses_csv, lines_csv)
    if load_scale is not None:
        g.buses.loc[:, "P"] = g.buses.P.to_numpy() * load_scale
        g.buses.loc[:, "Q"] = g.buses.Q.to_numpy() * load_scale
    rec = {"err": [], "V": [], "f0": None, "J0": None, "Y": None, "NE": None, "Vpf": None,
           "n_iter_f": None, "err_f": None, "t_jac": 0.0, "t_mis": 0.0, "t_sol": 0.0}
    o_mis, o_jac, o_pf, o_ne, o_y, o_upd = (g.harmonic_mismatch, g.build_harmonic_jacobian, g.pf,
                                            g.import_Norton_Equivalents, g.build_admittance_matrices,
                                            g.update_harmonic_state_vec)

    def w_mis(V, Y, buses, NE):
        t = time.perf_counter()
        f, e = o_mis(V, Y, buses, NE)
        rec["t_mis"] += time.perf_counter() - t
        if rec["f0"] is None:
            rec["f0"] = np.array(f, dtype=float).copy()
        rec["err"].append(float(e))
        if full or len(rec["V"]) < 2:
            rec["V"].append(V.to_numpy().copy())
        return f, e

    def w_jac(V, Y, NE, coupled):
        t = time.perf_counter()
        J = o_jac(V, Y, NE, coupled)
        rec["t_jac"] += time.perf_counter() - t
        if rec["J0"] is None:
            rec["J0"] = J.tocoo()
        return J

    def w_upd(J, x, f):
        t = time.perf_counter()
        r = o_upd(J, x, f)
        rec["t_sol"] += time.perf_counter() - t
        return r

    def w_pf(Y, buses, *a, **k):
        V, err_t, nf = o_pf(Y, buses, *a, **k)
        rec["Vpf"] = V.to_numpy().copy()
        rec["n_iter_f"] = nf
        rec["err_f"] = np.array([err_t[i] for i in sorted(err_t)], dtype=float)
        return V, err_t, nf

    def w_ne(buses, coupled):
        NE = o_ne(buses, coupled)
        rec["NE"] = {d: (np.asarray(v[0]).astype(complex), np.asarray(v[1]).astype(complex))
                     for d, v in NE.items()}
        return NE

    def w_y(buses, lines, harmonics):
        Y = o_y(buses, lines, harmonics)
        if full:
            rec["Y"] = Y.to_numpy().copy()
        return Y

    g.harmonic_mismatch, g.build_harmonic_jacobian, g.pf = w_mis, w_jac, w_pf
    g.import_Norton_Equivalents, g.build_admittance_matrices = w_ne, w_y
    g.update_harmonic_state_vec = w_upd
    try:
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            V, err_h, n_iter_h, J = g.hpf(g.buses, g.lines, coupled=coupled)
            thd = g.get_THD(V)
        wall = time.perf_counter() - t0
        loop_s = g.t_end_hpf_solve - g.t_start_hpf_solve
    finally:
        (g.harmonic_mismatch, g.build_harmonic_jacobian, g.pf, g.import_Norton_Equivalents,
         g.build_admittance_matrices, g.update_harmonic_state_vec) = o_mis, o_jac, o_pf, o_ne, o_y, o_upd
    out = {
        "harmonics": np.array(g.HARMONICS, dtype=np.int32),
        "n": g.n, "m": g.m, "c": g.c, "coupled": int(coupled),
        "V_pf": rec["Vpf"], "n_iter_f": rec["n_iter_f"], "err_f": rec["err_f"],
        "f0": rec["f0"], "err_hist": np.array(rec["err"]),
        "V_final": V.to_numpy().copy(), "n_iter_h": n_iter_h, "err_h": float(err_h),
        "THD": thd.to_numpy().copy(),
        "wall_s": wall, "loop_s": loop_s,
        "t_jac": rec["t_jac"], "t_mis": rec["t_mis"], "t_sol": rec["t_sol"],
    }
    J0 = rec["J0"]
    if J0 is not None:
        out["J0_shape"] = np.array(J0.shape)
        if full:
            out["J0_row"], out["J0_col"], out["J0_data"] = J0.row.astype(np.int32), J0.col.astype(np.int32), J0.data
        else:   # checksums only (J0 of syn1000 is 1.2 M nnz)
            csr = J0.tocsr()
            w = np.cos(np.arange(J0.shape[1]) * 0.37) + 1.5
            out["J0_nnz"] = csr.nnz
            out["J0_matvec"] = csr @ w
            out["J0_rmatvec"] = csr.T @ w[: J0.shape[0]]
            out["J0_absrowsum"] = np.asarray(abs(csr).sum(axis=1)).ravel()
    if full:
        out["V_traj"] = np.array(rec["V"])           # [n_iter_h+1][Hn*n][2], raw (signed, unwrapped)
        out["Y_all"] = rec["Y"]
    else:
        out["V_it0"], out["V_it1"] = rec["V"][0], rec["V"][1]
    for d, (i_n, y_n) in (rec["NE"] or {}).items():
        out["NE_dev"] = d
        out["I_N"], out["Y_N"] = i_n, y_n             # single device type in all shipped nets
    return out


def main(argv):
    what = set(argv) or {"nets", "syn", "hf"}
    work = _setup_scratch()
    g = _import_reference()
    sys.path.insert(0, os.path.join(REPO, "harmonic-power-flow_amd"))
    import synth
    summary = {}
    if "nets" in what:
        for net, (b, l) in {"net1": ("net1n_buses.csv", "net1n_lines.csv"),
                            "net2": ("net2_buses.csv", "net2_lines.csv"),
                            "net3": ("net3_buses.csv", "net3_lines.csv")}.items():
            for h_max in (11, 51):
                for coupled in (False, True):
                    name = f"{net}_H{h_max}_{'c' if coupled else 'uc'}"
                    out = run_case(g, b, l, h_max, coupled)
                    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
                    summary[name] = (out["n_iter_h"], out["err_h"], out["n_iter_f"])
                    print(name, summary[name], flush=True)
    if "quirks" in what:
        # fixtures authored for this build (tests/golden/inputs): parallel lines (overwrite, HG:151-155), pi-model shunts with the
        # reference's off-by-one (HG:163-168), a bus shunt away from the slack (HG:158-161), a PV bus; and a network without any
        # nonlinear bus (m == n, empty Norton dict)
        for net, hs in (("quirk5", (11,)), ("lin4", (11,))):
            for f in (net + "_buses.csv", net + "_lines.csv"):
                shutil.copy(os.path.join(GOLD, "inputs", f), os.path.join(work, f))
            for h_max in hs:
                for coupled in (False, True):
                    name = f"{net}_H{h_max}_{'c' if coupled else 'uc'}"
                    out = run_case(g, net + "_buses.csv", net + "_lines.csv", h_max, coupled)
                    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
                    summary[name] = (out["n_iter_h"], out["err_h"], out["n_iter_f"])
                    print(name, summary[name], flush=True)
    if "syn" in what:
        for n in (50, 100, 200):
            fb, fl = synth.gen(n, seed=0, outdir=work)
            name = f"syn{n}_H11_c"
            out = run_case(g, os.path.basename(fb), os.path.basename(fl), 11, True, full=(n <= 50))
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
            summary[name] = (out["n_iter_h"], out["err_h"], out["n_iter_f"], out["loop_s"])
            print(name, summary[name], flush=True)
    if "syn1000" in what:
        fb, fl = synth.gen(1000, seed=0, outdir=work)
        name = "syn1000_H51_c"
        out = run_case(g, os.path.basename(fb), os.path.basename(fl), 51, True, full=False)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        summary[name] = (out["n_iter_h"], out["err_h"], out["n_iter_f"], out["loop_s"],
                         out["t_jac"], out["t_mis"], out["t_sol"])
        print(name, summary[name], flush=True)
    for w in sorted(what):
        # config 4 evidence held by the reference itself: Monte-Carlo scenario s of the 128-scenario share (synth.scenario_scale),
        # `scenref<s>` -> tests/golden/syn1000_H51_scenref<s>.npz (about 25 min each: 5 min admittances + 27..35 NR iterations)
        if w.startswith("scenref"):
            s_id = int(w[len("scenref"):])
            fb, fl = synth.gen(1000, seed=0, outdir=work)
            out = run_case(g, os.path.basename(fb), os.path.basename(fl), 51, True, full=False,
                           load_scale=synth.scenario_scale(1000, s_id))
            out["scenario"] = s_id
            for k in ("J0_matvec", "J0_rmatvec", "J0_absrowsum", "f0", "I_N", "Y_N"):      # keep the fixture small
                out.pop(k, None)
            np.savez_compressed(os.path.join(GOLD, "syn1000_H51_scenref%d.npz" % s_id), **out)
            summary[w] = (out["n_iter_h"], out["err_h"], out["n_iter_f"], out["loop_s"])
            print(w, summary[w], flush=True)
    if "hf" in what:
        # hcne_based_on_fuchs.py (HF) is a script: run it in the scratch cwd (it writes V_log/I_log.json there)
        hf_dir = os.path.join(SCRATCH, "hf")
        os.makedirs(hf_dir, exist_ok=True)
        os.chdir(hf_dir)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ns = runpy.run_path(os.path.join(REF_HPF, "hcne_based_on_fuchs.py"))
        os.chdir(work)
        V = ns["V"]
        hf = {"V_final": V.to_numpy().astype(float),
              "index": [list(map(str, t)) for t in V.index.tolist()],
              "n_iter": int(ns.get("n_iter", -1)), "n_iter_h": int(ns.get("n_iter_h", -1))}
        for k in ("err", "err_h"):
            if k in ns:
                hf[k] = float(ns[k])
        # the harmonic NR of HF:185-356: every iterate (V_h_log, HF:186), the injections (I_inj_log, HF:251), the printed error
        # list (HF:354) and the LAST iteration's linear system as the script leaves it in its namespace (J_5 HF:339, dM HF:257,
        # U HF:190, U_new HF:346) -- the known-answer data of the oracle's restatement (oracle/hf_oracle.py) and of hpf_dense_solve
        hf["V_h_log"] = [ns["V_h_log"][i].to_numpy().astype(float).tolist() for i in range(hf["n_iter_h"])]
        hf["I_inj_log"] = [ns["I_inj_log"][i].to_numpy().astype(float).tolist() for i in range(hf["n_iter_h"])]
        hf["err_f_list"] = [float(l.split(":")[1]) for l in buf.getvalue().splitlines() if l.startswith("error_f:")]
        hf["err_h_list"] = [float(l.split(":")[1]) for l in buf.getvalue().splitlines() if l.startswith("error_h:")]
        for k in ("J_5", "dM", "U", "U_new", "J"):
            hf[k + "_last"] = np.asarray(ns[k], dtype=float)
        for k in ("Y_f", "Y_5"):
            hf[k + "_re"] = np.asarray(ns[k]).real
            hf[k + "_im"] = np.asarray(ns[k]).imag
        with open(os.path.join(GOLD, "hf_fuchs.json"), "w") as f:
            json.dump({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in hf.items()}, f, indent=1)
        # the reference's own committed golden: iteration-0 rows of V_log.json (fundamental 4-bus NR result)
        with open(os.path.join(REF_HPF, "V_log.json")) as f:
            vlog = json.load(f)
        rows = [r for r in vlog["data"] if r.get("iteration", r.get("level_0", None)) in (0, "0")]
        with open(os.path.join(GOLD, "v_log_iter0.json"), "w") as f:
            json.dump({"schema_fields": [x["name"] for x in vlog["schema"]["fields"]], "rows": rows}, f, indent=1)
        print("hf", hf["n_iter"], hf["n_iter_h"], hf.get("err_h"), flush=True)
    with open(os.path.join(GOLD, "summary_%s.json" % "_".join(sorted(what))), "w") as f:
        json.dump({k: [float(x) for x in v] for k, v in summary.items()}, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
