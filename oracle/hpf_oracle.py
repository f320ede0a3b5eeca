"""CPU ORACLE — test infrastructure, NOT product code.

A NumPy/SciPy restatement of the Newton-Raphson harmonic power flow of the reference
`Harmonic Power Flow/hcne_generalized.py` (HG).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module — as the checker / reported CPU baseline, never as
the thing shipped.  The product (`harmonic-power-flow_amd/`) never imports it and has no CPU fallback.

Pinning: `tests/test_oracle_golden.py` checks this restatement against golden vectors captured from the
unmodified reference by `oracle/make_golden.py` (12 net cases + synthetic feeders): iteration-0 `f`, `J`,
the whole `err_h` trajectory, iteration counts and final voltages.

Every function cites the reference lines it follows.  Where the reference's *operation order or library call*
decides the last bit (FMA-fused NumPy array complex multiply vs. unfused scalar/SciPy-sparse multiply, Smith
complex division, sequential Python `sum`, SuperLU through `spsolve`), the same call or an explicit emulation
is used so that the oracle tracks the reference's NR trajectory, not just its fixed point (SURVEY.md §0 traps).

No pandas in the NR loop; V is two flat float64 arrays `Vm`, `Va` of length Hn*n, stacked harmonic-major
(`k = q*n + i`, HG:139-143,175-179).
"""
import os
import time

import numpy as np
import pandas as pd
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve

# ---- module constants, HG:578-593 ----------------------------------------------------------------------------
BASE_POWER = 1000
BASE_VOLTAGE = 400
NET_FREQ = 50
base_current = BASE_POWER / BASE_VOLTAGE
base_admittance = base_current / BASE_VOLTAGE
base_impedance = 1 / base_admittance


def harmonics_upto(h_max):
    """HG:584  HARMONICS = [1, 3, ..., H_MAX]."""
    return [h for h in range(1, h_max + 1, 2)]


def cmul_unfused(a, b):
    """Complex multiply with every product and sum rounded separately — what NumPy *scalar* complex
    multiplication and SciPy-sparse kernels (csr_matvec / csr_matmat, complex_wrapper) compute."""
    ar, ai, br, bi = np.real(a), np.imag(a), np.real(b), np.imag(b)
    return (ar * br - ai * bi) + 1j * (ar * bi + ai * br)


# ---- ingest, HG:45-61, 77-94, 113-128 ---------------------------------------------------------------------
class Net:
    """Buses/lines in p.u. plus the index constants m, n, c of HG:122-127."""

    def __init__(self, buses, lines):
        self.buses, self.lines = buses, lines
        nl = buses.index[buses["type"] == "nonlinear"]
        self.m = int(min(nl)) if len(nl) > 0 else len(buses)           # HG:122-125 (0-based index)
        self.n = len(buses)                                            # HG:126
        self.c = int((buses["type"] == "PV").sum()) + 1                # HG:127
        self.P = buses["P"].to_numpy(dtype=float)
        self.Q = buses["Q"].to_numpy(dtype=float)
        self.component = buses["component"].to_numpy()


def init_network(filename_buses, filename_lines):
    """HG:113-128 with both CSV dialects (net1: `X_shunt`, no G/B; Appendix B of SURVEY.md)."""
    b = pd.read_csv(filename_buses, delimiter=";")
    if "X_shunt" in b.columns:
        b = b.rename(columns={"X_shunt": "X_sh"})
    for col, base in (("S", BASE_POWER), ("P", BASE_POWER), ("Q", BASE_POWER), ("X_sh", base_impedance)):
        b[col] = b[col].astype(float) / base                            # HG:89-92
    l = pd.read_csv(filename_lines, delimiter=";")
    l = l.dropna(how="all")
    for col in ("G", "B"):
        if col not in l.columns:
            l[col] = 0.0
    l["R"] = l.R.astype(float) / base_impedance                         # HG:57-60
    l["X"] = l.X.astype(float) / base_impedance
    l["G"] = l.G.astype(float) / base_admittance
    l["B"] = l.B.astype(float) / base_admittance
    return Net(b, l)


# ---- admittance matrices, HG:132-171 ------------------------------------------------------------------------
def _pattern(net):
    """CSR pattern shared by all harmonics: off-diagonals from the lines (later lines overwrite earlier ones
    on the same bus pair, HG:151-155) plus the full diagonal.  Returns (rowptr, col, line_of_entry, diag_pos)."""
    n = net.n
    fr = net.lines.fromID.to_numpy(dtype=np.int64) - 1
    to = net.lines.toID.to_numpy(dtype=np.int64) - 1
    owner = {}
    for k in range(len(fr)):                 # assignment order: [f,t] then [t,f]; the last writer wins
        owner[(fr[k], to[k])] = k
        owner[(to[k], fr[k])] = k
    for i in range(n):
        owner.setdefault((i, i), -1)
        # a self-loop line (fromID == toID) would be overwritten by the diagonal rule HG:159-161
        owner[(i, i)] = -1
    keys = sorted(owner)
    rows = np.array([k[0] for k in keys], dtype=np.int64)
    col = np.array([k[1] for k in keys], dtype=np.int32)
    line = np.array([owner[k] for k in keys], dtype=np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    diag_pos = np.nonzero(rows == col)[0]
    return rowptr, col, line, diag_pos, rows


def build_admittance_matrices(net, harmonics):
    """HG:132-171.  Returns (rowptr, col, Yval[Hn][nnz]) — the dense `Y_all` DataFrame of the reference holds the
    same numbers; entries not stored here are exact zeros there."""
    n = net.n
    R = net.lines.R.to_numpy(dtype=float)
    X = net.lines.X.to_numpy(dtype=float)
    G = net.lines.G.to_numpy(dtype=float)
    B = net.lines.B.to_numpy(dtype=float)
    fr = net.lines.fromID.to_numpy(dtype=np.int64)
    to = net.lines.toID.to_numpy(dtype=np.int64)
    X_sh = net.buses["X_sh"].to_numpy(dtype=float)
    rowptr, col, line, diag_pos, rows = _pattern(net)
    nnz = len(col)
    off = line >= 0
    deg = np.diff(rowptr)
    Yval = np.zeros((len(harmonics), nnz), dtype=complex)
    for q, h in enumerate(harmonics):
        yl = -1 / (R + 1j * X * h)                                      # HG:151-152 (Smith division, array loop)
        v = np.zeros(nnz, dtype=complex)
        v[off] = yl[line[off]]
        # HG:159/161: Python `sum(Y[n, :])` = sequential left-to-right sum over the row (diagonal still 0)
        acc = np.zeros(n, dtype=complex)
        for k in range(int(deg.max())):
            sel = np.nonzero(deg > k)[0]
            pos = rowptr[sel] + k
            acc[sel] = acc[sel] + v[pos]
        d = -acc
        if h != 1:                                                      # HG:158-159
            has = X_sh != 0
            d[has] = d[has] + 1 / (1j * X_sh[has] * h)
        # HG:163-168: pi-model shunt with the reference's off-by-one (0-based bus index n0 compared with 1-based
        # fromID/toID); lines visited in file order
        if np.any(G != 0) or np.any(B != 0):
            for n0 in range(n):
                for k in range(len(R)):
                    if fr[k] == n0 or to[k] == n0:
                        d[n0] = d[n0] + (G[k] + 1j * h * B[k]) / 2
        v[diag_pos] = d
        Yval[q] = v
    return rowptr, col, Yval


def y_csr(rowptr, col, yv, n):
    return sp.csr_matrix((yv, col, rowptr), shape=(n, n))


# ---- Norton equivalents, HG:278-310 -------------------------------------------------------------------------
def import_norton(ne_csv, harmonics, coupled):
    """HG:291-309 for one device file.  Returns (I_N[Hn], Y_N) in p.u.; Y_N is Hn×Hn (coupled, row = harmonic of
    the injected current, column = harmonic of the voltage) or length Hn (uncoupled)."""
    freqs = [NET_FREQ * h for h in harmonics]
    df = pd.read_csv(ne_csv, index_col=["Parameter", "Frequency"])
    df.columns = df.columns.astype(int)
    df = df[freqs]
    df = df.apply(lambda colv: colv.apply(lambda val: complex(val.strip("()"))))
    if coupled:
        I_N = (df.loc["I_N_c"] / base_current).to_numpy()[0]
        Y_N = (df.loc[("Y_N_c", freqs), freqs] / base_admittance).to_numpy()
    else:
        I_N = (df.loc["I_N_uc"] / base_current).to_numpy()[0]
        Y_N = (df.loc["Y_N_uc"] / base_admittance).to_numpy()[0]
    return np.asarray(I_N, dtype=complex), np.asarray(Y_N, dtype=complex)


def import_Norton_Equivalents(net, harmonics, coupled, ne_dir):
    """HG:284-310: one entry per unique `component` among nonlinear buses; file `<component>_NE.csv`, looked up
    case-insensitively (the reference relies on a case-insensitive file system for `SMPS` vs `smps`)."""
    NE = {}
    files = {f.lower(): f for f in os.listdir(ne_dir)}
    for dev in pd.unique(net.component[net.buses["type"].to_numpy() == "nonlinear"]):
        fn = files[(str(dev) + "_NE.csv").lower()]
        NE[dev] = import_norton(os.path.join(ne_dir, fn), harmonics, coupled)
    return NE


# ---- fundamental power flow, HG:174-275 ---------------------------------------------------------------------
def init_voltages(n, Hn):
    """HG:174-184."""
    Vm = np.zeros(Hn * n)
    Va = np.zeros(Hn * n)
    Vm[:n] = 1
    Vm[n:] = 0.1
    return Vm, Va


def pf(net, rowptr, col, Yval, thresh_f=1e-6, max_iter_f=30):
    """HG:244-275 (+ HG:187-241).  The reference works on the *dense* fundamental admittance `Y1` (HG:255): its
    matvecs are BLAS zgemv and its Jacobian is dense n×n arithmetic.  Mirrored literally while n is small enough
    for that to be cheap; above it the same formulas run on CSR."""
    n, c = net.n, net.c
    Hn = Yval.shape[0]
    Vm, Va = init_voltages(n, Hn)
    S = net.P + 1j * net.Q                                               # HG:197
    dense = n <= 2500
    Y1s = y_csr(rowptr, col, Yval[0], n)
    Y1 = Y1s.toarray() if dense else Y1s

    def mismatch():
        V_vec = Vm[:n] * np.exp(1j * Va[:n])                             # HG:196
        mis = V_vec * np.conj(Y1.dot(V_vec)) + S                         # HG:198
        f = np.r_[mis.real[1:], mis.imag[c:]]                            # HG:200
        return f, abs(f).max() if len(f) else 0.0

    def jacobian():
        V_vec = Vm[:n] * np.exp(1j * Va[:n])                             # HG:207
        I_diag = sp.diags(Y1 @ V_vec)                                    # HG:208
        V_diag = sp.diags(V_vec)
        V_diag_norm = sp.diags(V_vec / abs(V_vec))                       # HG:210
        dSdA = 1j * V_diag @ (np.conj(I_diag - Y1 @ V_diag))             # HG:212
        dSdV = V_diag_norm @ np.conj(I_diag) + V_diag @ np.conj(Y1 @ V_diag_norm)   # HG:213-214
        dPdA = sp.csr_matrix(dSdA[1:, 1:].real)                          # HG:217-220
        dPdV = sp.csr_matrix(dSdV[1:, c:].real)
        dQdA = sp.csr_matrix(dSdA[c:, 1:].imag)
        dQdV = sp.csr_matrix(dSdV[c:, c:].imag)
        return sp.vstack([sp.hstack([dPdA, dPdV]), sp.hstack([dQdA, dQdV])], format="csr")

    x = np.append(Va[1:n], Vm[c:n])                                      # HG:191
    f, err = mismatch()
    err_t = []
    n_iter_f = 0
    while err > thresh_f and n_iter_f < max_iter_f:                      # HG:259
        J = jacobian()
        x = x - spsolve(J, sp.csr_matrix(f).T)                           # HG:229
        Va[1:n] = x[0:(n - 1)]                                           # HG:234-235
        Vm[c:n] = x[(n - 1):]
        f, err = mismatch()
        err_t.append(err)
        n_iter_f += 1
    return Vm, Va, err_t, n_iter_f


# ---- harmonic NR kernels, HG:313-508 ------------------------------------------------------------------------
class Model:
    """Everything `hpf` needs that does not change during the NR loop."""

    def __init__(self, net, harmonics, rowptr, col, Yval, NE, coupled):
        self.net, self.harmonics, self.coupled = net, list(harmonics), bool(coupled)
        self.n, self.m, self.c = net.n, net.m, net.c
        self.Hn = len(harmonics)
        n, Hn = self.n, self.Hn
        self.rowptr, self.col, self.Yval = rowptr, col, Yval
        self.Y = [y_csr(rowptr, col, Yval[q], n) for q in range(Hn)]
        self.Y_diag = sp.block_diag(self.Y, format="csr")               # HG:407
        self.Y_h = sp.block_diag(self.Y[1:], format="csr") if Hn > 1 else None    # HG:342
        self.Y_f_nl = self.Y[0][self.m:, :]                              # HG:335
        self.Y_f_lin = self.Y[0][1:self.m, :]                            # HG:377
        self.S_lin = (net.P + 1j * net.Q)[1:self.m]                      # HG:372
        self.NE = NE
        self.dev_of_bus = [net.component[i] for i in range(n)]
        # indices of all nonlinear buses at every harmonic, HG:418-420 (harmonic-major)
        self.nl_bus = np.arange(self.m, n)
        self.N_c = n * Hn - 1
        self.N = 2 * self.N_c - (self.c - 1)

    def U(self, Vm, Va):
        return Vm * np.exp(1j * Va)                                      # HG:403 etc. (FMA-free: real*complex)


def current_injections(mdl, i, U):
    """HG:313-323: I_inj = I_N - Y_N·U_bus (coupled) / I_N - diag(Y_N)·U_bus (uncoupled), all harmonics of bus i."""
    I_N, Y_N = mdl.NE[mdl.dev_of_bus[i]]
    V_h = U[i::mdl.n]
    if Y_N.ndim == 2:
        # DataFrame.dot -> np.dot(values, v); `.values` of a single-block frame is F-ordered -> zgemv 'N'
        return I_N - np.dot(np.asfortranarray(Y_N), np.ascontiguousarray(V_h))
    return I_N - np.diag(Y_N).dot(V_h)


def current_balance(mdl, U):
    """HG:326-357."""
    n, m, Hn = mdl.n, mdl.m, mdl.Hn
    dI_f = mdl.Y_f_nl @ U[:n]                                            # HG:339 (csr_matvec, sequential, unfused)
    dI_h = mdl.Y_h @ U[n:] if Hn > 1 else np.zeros(0, dtype=complex)     # HG:345
    for i in range(m, n):                                                # HG:347-354
        I_inj = current_injections(mdl, i, U)
        dI_f[i - m] += I_inj[0]
        dI_h[np.arange(Hn - 1) * n + i] += I_inj[1:]
    return np.concatenate([dI_f, dI_h])


def harmonic_mismatch(mdl, Vm, Va):
    """HG:360-390."""
    n, m, c = mdl.n, mdl.m, mdl.c
    U = mdl.U(Vm, Va)
    V_i = U[1:m]
    Sl = V_i * np.conjugate(mdl.Y_f_lin @ U[:n])                         # HG:379 (array multiply: FMA-fused)
    dS = mdl.S_lin + Sl                                                  # HG:380
    dI = current_balance(mdl, U)
    f_c = np.concatenate([dS, dI])
    f = np.concatenate([f_c.real, f_c[c - 1:].imag])                     # HG:388
    err_h = np.linalg.norm(f, np.inf) if len(f) else 0.0
    return f, err_h


def harmonic_state_vector(mdl, Vm, Va):
    """HG:393-398."""
    return np.append(Va[1:], Vm[mdl.c:])


def build_harmonic_jacobian(mdl, Vm, Va):
    """HG:401-473."""
    n, m, c, Hn = mdl.n, mdl.m, mdl.c, mdl.Hn
    K = Hn - 1
    V_vec = mdl.U(Vm, Va)
    V_norm = V_vec / Vm                                                  # HG:405 (complex/real: times 1/Vm)
    dIdV = mdl.Y_diag @ sp.diags(V_norm)                                 # HG:410
    dIdA = (1j * mdl.Y_diag) @ sp.diags(V_vec)                           # HG:411
    nb = mdl.nl_bus
    if len(nb):
        q = np.arange(Hn)
        if mdl.coupled:                                                  # HG:425-435
            # all (h, p, i): value subtracted at [h*n+i, p*n+i]; scalar (unfused) complex arithmetic
            hh, pp, ii = np.meshgrid(q, q, nb, indexing="ij")
            YN = np.empty((Hn, Hn, len(nb)), dtype=complex)
            for d in set(mdl.dev_of_bus[i] for i in nb):
                sel = np.array([mdl.dev_of_bus[i] == d for i in nb])
                YN[:, :, sel] = mdl.NE[d][1][:, :, None]
        else:                                                            # HG:437-443
            hh, ii = np.meshgrid(q, nb, indexing="ij")
            pp = hh
            YN = np.empty((Hn, len(nb)), dtype=complex)
            for d in set(mdl.dev_of_bus[i] for i in nb):
                sel = np.array([mdl.dev_of_bus[i] == d for i in nb])
                YN[:, sel] = mdl.NE[d][1][:, None]
        rows = (hh * n + ii).ravel()
        cols = (pp * n + ii).ravel()
        YN = YN.ravel()
        sV = cmul_unfused(YN, V_norm[cols])                              # HG:432-433
        jYN = -YN.imag + 1j * YN.real                                    # 1j*Y_N (exact)
        sA = cmul_unfused(jYN, V_vec[cols])                              # HG:434-435
        shape = (n * Hn, n * Hn)
        dIdV = dIdV - sp.csr_matrix((sV, (rows, cols)), shape=shape)
        dIdA = dIdA - sp.csr_matrix((sA, (rows, cols)), shape=shape)
    dIdA = sp.csr_matrix(dIdA)[m:, 1:]                                   # HG:445-446
    dIdV = sp.csr_matrix(dIdV)[m:, c:]

    Y1 = mdl.Y[0]                                                        # HG:451-459
    V1 = V_vec[:n]
    I_diag = sp.diags(Y1 @ V1)
    V_diag = sp.diags(V1)
    V_diag_norm = sp.diags(V1 / Vm[:n])
    dS1dA1 = 1j * V_diag @ (np.conj(I_diag - Y1 @ V_diag))
    dS1dV1 = V_diag_norm @ np.conj(I_diag) + V_diag @ np.conj(Y1 @ V_diag_norm)
    zpad = sp.csr_matrix((n, n * K))
    dSdA = sp.csr_matrix(sp.hstack([dS1dA1, zpad]))                      # HG:461-462
    dSdV = sp.csr_matrix(sp.hstack([dS1dV1, zpad]))
    dPdA = dSdA[1:m, 1:].real                                            # HG:464-467
    dPdV = dSdV[1:m, c:].real
    dQdA = dSdA[c:m, 1:].imag
    dQdV = dSdV[c:m, c:].imag
    J = sp.vstack([sp.hstack([dPdA, dPdV]),                              # HG:469-472
                   sp.hstack([dIdA.real, dIdV.real]),
                   sp.hstack([dQdA, dQdV]),
                   sp.hstack([dIdA.imag, dIdV.imag])], format="csr")
    return J


def update_harmonic_state_vec(J, x, f):
    """HG:476-479 (SuperLU via spsolve)."""
    return x - spsolve(J, f)


def update_harmonic_voltages(mdl, Vm, Va, x):
    """HG:482-485."""
    Nc = mdl.N_c
    Va[1:] = x[:Nc]
    Vm[mdl.c:] = x[Nc:]


def postprocess(Vm, Va):
    """HG:545-549: add pi where the magnitude is negative, wrap all angles to [0, 2pi), take |V_m|."""
    Va = Va.copy()
    Vm = Vm.copy()
    neg = Vm < 0
    Va[neg] += np.pi
    Va = Va % (2 * np.pi)
    Vm[neg] = -Vm[neg]
    return Vm, Va


def get_THD(Vm, n, Hn):
    """HG:563-572 (harmonic labels >= 3 are stack positions >= 1)."""
    V = Vm.reshape(Hn, n)
    thd_f = np.empty(n)
    thd_r = np.empty(n)
    for b in range(n):
        hs = sum(V[1:, b] ** 2)            # Python sum, sequential (HG:567)
        thd_f[b] = np.sqrt(hs) / V[0, b]
        thd_r[b] = np.sqrt(hs) / np.sqrt(sum(V[:, b] ** 2))
    return np.stack([thd_f, thd_r], axis=1)


def hpf_from_model(mdl, Vm, Va, thresh_h=1e-4, max_iter_h=50, record=False, timers=None):
    """The NR loop of HG:530-549 from a given seed (Vm, Va modified in place)."""
    n_iter_h = 0
    f, err_h = harmonic_mismatch(mdl, Vm, Va)                            # HG:531
    x = harmonic_state_vector(mdl, Vm, Va)                               # HG:532
    hist = [err_h]
    traj = [(Vm.copy(), Va.copy())] if record else None
    J = None
    t0 = time.perf_counter()
    while err_h > thresh_h and n_iter_h < max_iter_h:                    # HG:536
        J = build_harmonic_jacobian(mdl, Vm, Va)
        x = update_harmonic_state_vec(J, x, f)
        update_harmonic_voltages(mdl, Vm, Va, x)
        f, err_h = harmonic_mismatch(mdl, Vm, Va)
        hist.append(err_h)
        if record:
            traj.append((Vm.copy(), Va.copy()))
        n_iter_h += 1
    loop_s = time.perf_counter() - t0
    if timers is not None:
        timers["loop_s"] = loop_s
    return {"Vm_raw": Vm, "Va_raw": Va, "err_h": err_h, "n_iter_h": n_iter_h, "J": J,
            "err_hist": np.array(hist), "traj": traj, "loop_s": loop_s}


def hpf(net, harmonics, coupled, ne_dir, thresh_h=1e-4, max_iter_h=50, record=False):
    """HG:511-560.  Returns a dict with raw and post-processed voltages, iteration data and the pf seed."""
    rowptr, col, Yval = build_admittance_matrices(net, harmonics)        # HG:523
    Vm, Va, err_t, n_iter_f = pf(net, rowptr, col, Yval)                 # HG:525
    NE = import_Norton_Equivalents(net, harmonics, coupled, ne_dir)      # HG:528
    mdl = Model(net, harmonics, rowptr, col, Yval, NE, coupled)
    seed = (Vm.copy(), Va.copy())
    out = hpf_from_model(mdl, Vm, Va, thresh_h, max_iter_h, record)
    out["Vm"], out["Va"] = postprocess(out["Vm_raw"], out["Va_raw"])
    out.update(model=mdl, seed=seed, n_iter_f=n_iter_f, err_f=np.array(err_t))
    return out
