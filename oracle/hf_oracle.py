"""CPU restatement of `Harmonic Power Flow/hcne_based_on_fuchs.py` (HF) -- TEST INFRASTRUCTURE, like hpf_oracle.py: only tests/ may import
it.  HF is a fixed script: the 4-bus ring of Fuchs' example, a fundamental Newton-Raphson (HF:79-131) and then a harmonic
Newton-Raphson for the 5th harmonic with an ANALYTIC nonlinear load at bus 4 (g(), HF:170-173) instead of Norton data (HF:185-356).
Restated here as functions on plain arrays (same formulas, same row / column orders, same LAPACK calls: `np.linalg.inv(J).dot(f)`),
pinned by tests/golden/hf_fuchs.json, which oracle/make_golden.py captured from the unmodified script: every iterate (V_h_log), the
printed error lists, the last linear system (J_5, dM, U, U_new) and the final voltages.
Quirks kept: `err` is the norm of the PRE-update mismatch, so the fundamental loop runs one iteration "late";
the harmonic Jacobian reuses the fundamental block `J` of the last fundamental iteration (stale, HF:261); the diagonal rule of Y_f is
only right for a ring (HF:66-74)."""
import numpy as np

PU = 1000.0                                       # HF:13
# HF:44-54 (bus 1 slack with a subtransient reactance for the harmonics, bus 4 nonlinear)
P = np.array([0.0, 100.0, 0.0, 250.0])
Q = np.array([0.0, 100.0, 0.0, 100.0])
X_SHUNT = np.array([0.0001, 0.0, 0.0, 0.0])
LINES = [(1, 2, 0.01, 0.01), (2, 3, 0.02, 0.08), (3, 4, 0.01, 0.02), (4, 1, 0.01, 0.02)]       # fromID, toID, R, X


def admittances(h):
    """Y_f (h = 1, HF:57-74: diagonal = first line leaving + first line arriving, a ring) / Y_5 (h = 5, HF:147-162: diagonal =
    minus the row sum taken before the diagonal is set, plus the slack's 1 / (j X_shunt h))."""
    n = 4
    Y = np.zeros((n, n), dtype=complex)
    lines = [(f, t, np.float64(r), np.float64(x)) for f, t, r, x in LINES]       # (NumPy scalars like the script's DataFrame cells: NumPy's
    for f, t, r, x in lines:                                                     #  complex division rounds differently from Python's)
        Y[f - 1, t - 1] = -1 / (r + 1j * x * h) if h != 1 else -1 / (r + 1j * x)
        Y[t - 1, f - 1] = Y[f - 1, t - 1]
    for k in range(n):
        if h == 1:
            zf = [r + 1j * x for f, t, r, x in lines if f == k + 1][0]
            zt = [r + 1j * x for f, t, r, x in lines if t == k + 1][0]
            Y[k, k] = 1 / zf + 1 / zt
        else:
            Y[k, k] = -sum(Y[k, :]) + (1 / (1j * float(X_SHUNT[k]) * h) if X_SHUNT[k] != 0 else 0)   # (X_shunt: a Python float there)
    return Y


def fundamental(Vm1, Va1, err_max=1e-4, n_iter_max=20):
    """HF:79-131.  Returns (Vm, Va) of the fundamental, the error list and the Jacobian of the LAST iteration (reused by the
    harmonic loop, HF:261)."""
    Yf = admittances(1)
    S = (P + 1j * Q) / PU
    Vm, Va = Vm1.astype(float).copy(), Va1.astype(float).copy()
    errs, n_iter, err, J = [], 1, 1.0, None
    while err > err_max and n_iter <= n_iter_max:
        V = Vm * np.exp(1j * Va)
        dm = V * np.conj(Yf.dot(V)) + S                                            # HF:85
        x = np.stack([Va[1:], Vm[1:]], axis=1).ravel()                             # HF:88-89: (angle, magnitude) per bus, slack cut
        f = np.stack([dm[1:].real, dm[1:].imag], axis=1).ravel()                   # HF:90-93
        Idg, Vdg, Vn = np.diag(Yf.dot(V)), np.diag(V), np.diag(V / abs(V))         # HF:96-98
        dSdt = 1j * Vdg.dot(np.conj(Idg - Yf.dot(Vdg)))                            # HF:101
        dSdV = Vn.dot(np.conj(Idg)) + Vdg.dot(np.conj(Yf.dot(Vn)))                 # HF:105-106
        Jb = np.zeros((8, 8))                                                      # HF:111-117: Fuchs' sorting, 2x2 per bus pair
        Jb[0::2, 0::2], Jb[1::2, 0::2], Jb[0::2, 1::2], Jb[1::2, 1::2] = dSdt.real, dSdt.imag, dSdV.real, dSdV.imag
        J = Jb[2:, 2:]
        x_new = x - np.linalg.inv(J).dot(f)                                        # HF:121-124
        Va[1:], Vm[1:] = x_new[0::2], x_new[1::2]
        err = np.linalg.norm(f, np.inf)                                            # HF:130 (pre-update mismatch)
        errs.append(float(err))
        n_iter += 1
    return Vm, Va, errs, n_iter, J


def g_load(Vm1, Va1, Vm5, Va5):
    """HF:170-173: harmonic current of the nonlinear load at bus 4 from its fundamental and 5th-harmonic voltage."""
    return 0.3 * Vm1 ** 3 * np.exp(3j * Va1) + 0.3 * Vm5 ** 2 * np.exp(3j * Va5)


def harmonic_system(Vm, Va, J_fund):
    """One pass of HF:188-341 at the state (Vm, Va) [2][4] (row 0 fundamental, row 1 h = 5): U, dM, J_5 and the injections."""
    Yf, Y5 = admittances(1), admittances(5)
    S = (P + 1j * Q) / PU
    U = np.stack([Va.ravel(), Vm.ravel()], axis=1).ravel()[2:]                     # HF:189-191
    Vf, V5 = Vm[0] * np.exp(1j * Va[0]), Vm[1] * np.exp(1j * Va[1])
    eps1 = np.arctan(Q[3] / P[3])                                                  # HF:197
    gam1 = Va[0, 3] - eps1
    den = Vm[0, 3] * np.cos(Va[0, 3] - gam1)
    G1 = P[3] / PU * np.cos(gam1) / den + 1j * (P[3] / PU * np.sin(gam1) / den)    # HF:202-208
    G5 = g_load(Vm[0, 3], Va[0, 3], Vm[1, 3], Va[1, 3])                            # HF:211,216
    dW_lin = Vf * np.conj(Yf.dot(Vf)) + S                                          # HF:229
    dW = np.array([dW_lin[1].real, dW_lin[1].imag, dW_lin[2].real, dW_lin[2].imag])
    dI_1 = Yf.dot(Vf)[3] + G1                                                      # HF:237
    I5 = Y5.dot(V5)
    dI_5_nlin = I5[3] + G5                                                         # HF:240-241
    dI = np.array([I5[0].real, I5[0].imag, I5[1].real, I5[1].imag, I5[2].real, I5[2].imag,
                   dI_5_nlin.real, dI_5_nlin.imag, dI_1.real, dI_1.imag])          # HF:249-253
    dM = np.append(dW, dI)
    # Jacobian blocks, HF:260-341
    J1 = J_fund[:4, :]
    dgdt_1 = 0.9j * Vm[0, 3] ** 3 * np.exp(3j * Va[0, 3])                          # HF:269-272
    dgdV_1 = 0.9 * Vm[0, 3] ** 2 * np.exp(3j * Va[0, 3])
    G51 = np.zeros((8, 6))
    G51[6, 4], G51[7, 4], G51[6, 5], G51[7, 5] = dgdt_1.real, dgdt_1.imag, dgdV_1.real, dgdV_1.imag
    dgdt_5 = 0.9j * Vm[1, 3] ** 2 * np.exp(3j * Va[1, 3])                          # HF:281-284
    dgdV_5 = 0.6 * Vm[1, 3] * np.exp(3j * Va[1, 3])
    YG55 = np.zeros((8, 8))
    for i in range(4):                                                             # HF:291-298 (scalar complex products, as there)
        for k in range(4):
            a = 1j * Y5[i, k] * V5[k]                                              # d / d angle of bus k
            e = Y5[i, k] * np.exp(1j * Va[1, k])                                   # d / d magnitude
            YG55[2 * i, 2 * k], YG55[2 * i + 1, 2 * k], YG55[2 * i, 2 * k + 1], YG55[2 * i + 1, 2 * k + 1] = a.real, a.imag, e.real, e.imag
    YG55[6, 6] += dgdt_5.real                                                      # HF:300-304
    YG55[7, 6] += dgdt_5.imag
    YG55[6, 7] += dgdV_5.real
    YG55[7, 7] += dgdV_5.imag
    YG11 = np.zeros((2, 6))
    for k in range(3):                                                             # HF:311-318
        a1 = 1j * Yf[3, k + 1] * Vf[k + 1]
        e1 = Yf[3, k + 1] * np.exp(1j * Va[0, k + 1])
        YG11[0, 2 * k], YG11[1, 2 * k], YG11[0, 2 * k + 1], YG11[1, 2 * k + 1] = a1.real, a1.imag, e1.real, e1.imag
    dIdt_1, dIdV_1 = 1j * G1, -G1 / Vm[0, 3]                                       # HF:322-323
    YG11[0, 4] += dIdt_1.real
    YG11[1, 4] += dIdt_1.imag
    YG11[0, 5] += dIdV_1.real
    YG11[1, 5] += dIdV_1.imag
    J_5 = np.block([[J1, np.zeros((4, 8))], [G51, YG55], [YG11, np.zeros((2, 8))]])  # HF:339-341
    return U, dM, J_5, (G1, G5)


def run(err_h_max=0.01, n_iter_max=20):
    """The whole script: initial values HF:36-41, fundamental NR, harmonic NR (HF:180-356), phase normalisation HF:358-359."""
    Vm = np.array([[1.0] * 4, [0.1] * 4])
    Va = np.zeros((2, 4))
    Vm[0], Va[0], err_f, n_iter, J = fundamental(Vm[0], Va[0])
    log, inj, errs = [], [], []
    err_h, n_iter_h = 1.0, 0
    last = None
    while err_h > err_h_max and n_iter_h < n_iter_max:
        log.append(np.stack([Vm.ravel(), Va.ravel()], axis=1))                     # HF:186
        U, dM, J_5, (G1, G5) = harmonic_system(Vm, Va, J)
        inj.append([[G1.real, G1.imag], [G5.real, G5.imag]])
        err_h = np.linalg.norm(dM, np.inf)                                         # HF:258
        U_new = U - np.linalg.inv(J_5).dot(dM)                                     # HF:345-346
        va, vm = np.append(Va.ravel()[:1], U_new[0::2]), np.append(Vm.ravel()[:1], U_new[1::2])   # HF:349-350
        Va, Vm = va.reshape(2, 4), vm.reshape(2, 4)
        Va[1] = Va[1] + np.pi                                                      # HF:351-352
        Vm[1] = -Vm[1]
        errs.append(float(err_h))
        last = dict(U=U, dM=dM, J_5=J_5, U_new=U_new)
        n_iter_h += 1
    c5 = Vm[1] * np.exp(1j * Va[1])                                                # HF:358-359
    Vm[1], Va[1] = np.abs(c5), np.angle(c5)
    return dict(Vm=Vm, Va=Va, V_h_log=log, I_inj_log=inj, err_f_list=err_f, err_h_list=errs, n_iter=n_iter, n_iter_h=n_iter_h,
                err_h=float(err_h), J_fund=J, last=last)
