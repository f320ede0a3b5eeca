"""CSV bus/line ingest and Norton-parameter import — the reference's file formats, kept as the drop-in boundary.

Follows `Harmonic Power Flow/hcne_generalized.py` (HG): `init_lines_from_csv` HG:45-61, `init_buses_from_csv`
HG:77-94, `init_network` HG:113-128, `import_Norton_Equivalents` HG:278-310.  Unlike the reference, both CSV
dialects are accepted natively (net1: `X_shunt`, no `G;B` columns — SURVEY.md Appendix B) and the Norton CSV is
looked up in a configurable directory, case-insensitively (the reference hard-codes `~/Git/...` and relies on a
case-insensitive file system for `SMPS` vs `smps`, HG:289-290).

`pandas.read_csv` is used on purpose: its float parser decides the input doubles, and parity with the reference
starts at the inputs.
"""
import os

import numpy as np
import pandas as pd

from .settings import Settings


def init_lines_from_csv(filename, settings=None):
    """HG:45-61: `;`-delimited, columns ID;fromID;toID;R;X[;G;B]; R,X -> p.u. impedance, G,B -> p.u. admittance."""
    st = settings or Settings()
    df = pd.read_csv(filename, delimiter=";").dropna(how="all").reset_index(drop=True)
    for col in ("G", "B"):
        if col not in df.columns:
            df[col] = 0.0
    df["R"] = df.R.astype(float) / st.base_impedance
    df["X"] = df.X.astype(float) / st.base_impedance
    df["G"] = df.G.astype(float) / st.base_admittance
    df["B"] = df.B.astype(float) / st.base_admittance
    return df


def init_buses_from_csv(filename, settings=None):
    """HG:77-94: columns ID;type;component;S;P;Q;X_sh[;V_nom] (or net1's ID;type;component;S;X_shunt;P;Q)."""
    st = settings or Settings()
    df = pd.read_csv(filename, delimiter=";").dropna(how="all").reset_index(drop=True)
    if "X_shunt" in df.columns and "X_sh" not in df.columns:
        df = df.rename(columns={"X_shunt": "X_sh"})
    df["S"] = df.S.astype(float) / st.BASE_POWER
    df["P"] = df.P.astype(float) / st.BASE_POWER
    df["Q"] = df.Q.astype(float) / st.BASE_POWER
    df["X_sh"] = df.X_sh.astype(float) / st.base_impedance
    return df


def network_constants(buses):
    """m, n, c of HG:121-127: m = 0-based index of the first nonlinear bus (n if none), c = #PV + 1."""
    nl = buses.index[buses["type"] == "nonlinear"]
    m = int(min(nl)) if len(nl) > 0 else len(buses)
    n = len(buses)
    c = len(buses[buses.type == "PV"]) + 1
    return m, n, c


def validate_bus_order(buses):
    """The reference assumes slack, PV..., PQ..., nonlinear... (HG:83, TODO HG:114) without checking; a violated
    order silently gives wrong index sets, so it is an error here."""
    rank = {"slack": 0, "PV": 1, "PQ": 2, "nonlinear": 3}
    r = [rank.get(t, -1) for t in buses["type"]]
    if -1 in r:
        raise ValueError("unknown bus type %r" % sorted(set(buses["type"]) - set(rank)))
    if r[0] != 0 or r.count(0) != 1 or any(b < a for a, b in zip(r, r[1:])):
        raise ValueError("buses must be ordered slack, PV..., PQ..., nonlinear... (reference contract, HG:83)")


# The network the reference's manual initialisers describe (HG:64-74, HG:97-110: a 4-bus ring like net2 with pi-model line shunts).  The
# reference's own versions cannot run (the bus table is built as an ndarray of strings, so `buses.S/BASE_POWER` raises; HG:110 reads a V_nom column
# that does not exist); what they INTEND is unambiguous -- the same tables through the same p.u. conversion as the CSV path -- and that is what
# from_csv=False gives here.  SI units, columns as in the CSV dialect of net2.
_MANUAL_BUSES = (("ID", "type", "component", "S", "P", "Q", "X_sh"),
                 ((1, "slack", "generator", 0.0, 0.0, 0.0, 0.005), (2, "PQ", "lin_load_1", 0.0, 100.0, 100.0, 0.0),
                  (3, "PQ", "lin_load_2", 0.0, 100.0, 100.0, 0.0), (4, "nonlinear", "smps", 0.0, 150.0, 100.0, 0.0)))
_MANUAL_LINES = (("ID", "fromID", "toID", "R", "X", "G", "B"),
                 ((1, 1, 2, 0.5, 0.5, 0.0, 0.05), (2, 2, 3, 1.0, 4.0, 0.0, 0.1), (3, 3, 4, 0.5, 1.0, 0.0, 0.05), (4, 4, 1, 0.5, 1.0, 0.0, 0.05)))


def init_buses_manually(settings=None):
    """HG:97-110 as intended (see _MANUAL_BUSES)."""
    st = settings or Settings()
    df = pd.DataFrame([dict(zip(_MANUAL_BUSES[0], r)) for r in _MANUAL_BUSES[1]])
    df["S"] = df.S.astype(float) / st.BASE_POWER
    df["P"] = df.P.astype(float) / st.BASE_POWER
    df["Q"] = df.Q.astype(float) / st.BASE_POWER
    df["X_sh"] = df.X_sh.astype(float) / st.base_impedance
    return df


def init_lines_manually(settings=None):
    """HG:64-74."""
    st = settings or Settings()
    df = pd.DataFrame([dict(zip(_MANUAL_LINES[0], r)) for r in _MANUAL_LINES[1]])
    df["R"] = df.R.astype(float) / st.base_impedance
    df["X"] = df.X.astype(float) / st.base_impedance
    df["G"] = df.G.astype(float) / st.base_admittance
    df["B"] = df.B.astype(float) / st.base_admittance
    return df


def init_network(filename_buses, filename_lines, from_csv=True, settings=None):
    """HG:113-128 -> (buses, lines, m, n, c).  from_csv=False: the built-in 4-bus network of the reference's manual initialisers (the file
    names are ignored, as in the reference)."""
    if from_csv:
        buses = init_buses_from_csv(filename_buses, settings)
        lines = init_lines_from_csv(filename_lines, settings)
    else:
        buses = init_buses_manually(settings)
        lines = init_lines_manually(settings)
    validate_bus_order(buses)
    m, n, c = network_constants(buses)
    return buses, lines, m, n, c


def default_ne_dir():
    """Directory holding `<component>_NE.csv`: $HPF_NE_DIR, else the reference's hard-coded location HG:289."""
    return os.environ.get("HPF_NE_DIR") or os.path.expanduser("~/Git/harmonic-power-flow/Circuit Simulation")


def _find_ne_file(ne_dir, device):
    want = (str(device) + "_NE.csv").lower()
    for f in sorted(os.listdir(ne_dir)):
        if f.lower() == want:
            return os.path.join(ne_dir, f)
    raise FileNotFoundError("no Norton-equivalent file for component %r in %s" % (device, ne_dir))


_NORTON_FRAMES = {}


def import_Norton_Equivalents(buses, coupled, settings=None, ne_dir=None):
    """HG:278-310 -> {device: [I_N, Y_N]} as DataFrames in p.u., exactly the reference's objects."""
    st = settings or Settings()
    ne_dir = ne_dir or default_ne_dir()
    want = [int(f) for f in st.HARMONICS_FREQ]
    types = buses["type"].to_numpy()
    comps = buses["component"].to_numpy()
    NE = {}
    for device in dict.fromkeys(comps[types == "nonlinear"]):            # unique device names, first-seen order (HG:285)
        path = _find_ne_file(ne_dir, device)
        # (repeated hpf() calls, HG:511: the two frames of a device are built once per file version / harmonic set / base and handed out as copies)
        fkey = (path, os.stat(path).st_mtime_ns, bool(coupled), tuple(want), float(st.base_current), float(st.base_admittance))
        hit = _NORTON_FRAMES.get(fkey)
        if hit is not None:
            NE[device] = [hit[0].copy(), hit[1].copy()]
            continue
        have, Ycc, Ic, Yuc, Iuc = read_Norton_file(path)
        pos = {f: j for j, f in enumerate(have)}
        missing = [f for f in want if f not in pos]
        if missing:
            raise KeyError("Norton file of %r lacks harmonics at %s Hz" % (device, missing))
        sel = np.array([pos[f] for f in want])
        # SI -> p.u. (HG:298-308); the reference hands out pandas objects labelled by frequency in Hz, and its callers
        # index them positionally (HG:320, 432) or by that label (HG:322): keep both ways working
        if coupled:
            i_n = pd.DataFrame((Ic[sel] / st.base_current)[None, :], columns=want,
                               index=pd.Index([0], name="Frequency"))
            y_n = pd.DataFrame(Ycc[np.ix_(sel, sel)] / st.base_admittance, columns=want,
                               index=pd.MultiIndex.from_product([["Y_N_c"], want], names=["Parameter", "Frequency"]))
        else:
            i_n = pd.DataFrame((Iuc[sel] / st.base_current)[None, :], columns=want,
                               index=pd.Index([0], name="Frequency"))
            y_n = pd.DataFrame((Yuc[sel] / st.base_admittance)[None, :], columns=want,
                               index=pd.Index([0], name="Frequency"))
        if len(_NORTON_FRAMES) > 16:
            _NORTON_FRAMES.clear()
        _NORTON_FRAMES[fkey] = (i_n.copy(), y_n.copy())
        NE[device] = [i_n, y_n]
    return NE


def _cstr(z):
    z = complex(z)
    im = repr(z.imag)
    return "(%s%s%sj)" % (repr(z.real), "" if im.startswith("-") else "+", im)


def export_Norton_Equivalents(filename, freqs, Y_N_c, I_N_c, Y_N_uc, I_N_uc):
    """Write one device's Norton parameters (SI units) in the `<component>_NE.csv` layout the reference's fitting
    script produces (`Circuit Simulation/NE_from_sim.py:195-209`) and `import_Norton_Equivalents` reads: index
    (Parameter, Frequency), one column per frequency in Hz; rows `Y_N_c` x len(freqs) (row label = frequency),
    then `I_N_c`, `Y_N_uc`, `I_N_uc` with Frequency 0.  Values are written as `(a+bj)` with shortest-round-trip
    reprs, so write -> read reproduces every double bit for bit."""
    freqs = [int(f) for f in freqs]
    K = len(freqs)
    Y_N_c = np.asarray(Y_N_c, dtype=np.complex128).reshape(K, K)
    rows = [("Y_N_c", freqs[r], Y_N_c[r]) for r in range(K)]
    for name, v in (("I_N_c", I_N_c), ("Y_N_uc", Y_N_uc), ("I_N_uc", I_N_uc)):
        rows.append((name, 0, np.asarray(v, dtype=np.complex128).reshape(K)))
    with open(filename, "w") as fh:
        fh.write("Parameter,Frequency," + ",".join(str(f) for f in freqs) + "\n")
        for name, f, vals in rows:
            fh.write("%s,%d,%s\n" % (name, f, ",".join(_cstr(z) for z in vals)))
    return filename


_NORTON_FILES = {}       # (real path, mtime_ns, size) -> parsed content: a sweep of hpf() calls parses a Norton file once (35 ms for smps_NE.csv)


def read_Norton_file(filename):
    """Raw (SI) content of a `<component>_NE.csv`: (freqs, Y_N_c [K][K], I_N_c [K], Y_N_uc [K], I_N_uc [K]).
    Parsed value by value with `complex()` (the reference's parser, HG:296) — not through a pandas object->complex
    conversion, which drops the sign of a zero real part.  The parsed content is kept per (path, modification time, size); every call returns
    fresh copies of the arrays."""
    st_ = os.stat(filename)
    key = (os.path.realpath(filename), st_.st_mtime_ns, st_.st_size)
    hit = _NORTON_FILES.get(key)
    if hit is None:
        if len(_NORTON_FILES) >= 32:
            _NORTON_FILES.clear()
        hit = _NORTON_FILES[key] = _parse_Norton_file(filename)
    return (list(hit[0]),) + tuple(a.copy() for a in hit[1:])


def _parse_Norton_file(filename):
    df = pd.read_csv(filename, index_col=["Parameter", "Frequency"], dtype=str)
    df.columns = df.columns.astype(int)
    freqs = list(df.columns)

    def rows(sel):
        return np.array([[complex(v.strip("()")) for v in r] for r in sel.to_numpy()], dtype=np.complex128)
    return (freqs, rows(df.loc[[("Y_N_c", f) for f in freqs], freqs]), rows(df.loc[["I_N_c"]]).reshape(-1),
            rows(df.loc[["Y_N_uc"]]).reshape(-1), rows(df.loc[["I_N_uc"]]).reshape(-1))


def norton_arrays(buses, NE, coupled, Hn):
    """Pack the Norton dict for the device: dev_of_bus[n] (-1 = linear), Y_N [n_dev][Hn][Hn] (coupled) or
    [n_dev][Hn] (uncoupled), I_N [n_dev][Hn], complex128 C-ordered."""
    types = buses["type"].to_numpy()
    comps = buses["component"].to_numpy()
    devices = list(NE.keys())
    dev_of_bus = np.full(len(buses), -1, dtype=np.int32)
    for i in range(len(buses)):
        if types[i] == "nonlinear":
            dev_of_bus[i] = devices.index(comps[i])
    n_dev = max(len(devices), 1)
    I_N = np.zeros((n_dev, Hn), dtype=np.complex128)
    Y_N = np.zeros((n_dev, Hn, Hn) if coupled else (n_dev, Hn), dtype=np.complex128)
    for d, name in enumerate(devices):
        i_n, y_n = NE[name]
        I_N[d] = np.asarray(i_n, dtype=np.complex128).reshape(-1)
        Y_N[d] = np.asarray(y_n, dtype=np.complex128).reshape(Y_N[d].shape)
    return dev_of_bus, np.ascontiguousarray(Y_N), np.ascontiguousarray(I_N), len(devices)
