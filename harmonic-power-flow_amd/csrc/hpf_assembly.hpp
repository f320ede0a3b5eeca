// Per-entry mathematics of the harmonic power-flow NR assembly (mismatch rows and Jacobian entries).
//
// Everything here is a pure function of (model, voltages) for ONE row / ONE entry, callable from a HIP thread.
// The HIP kernels in hpf_kernels.hip decide the thread mapping and memory layout; this header fixes the
// arithmetic, restated from the reference `Harmonic Power Flow/hcne_generalized.py` (HG) with the *operation
// order and rounding points of the library calls the reference makes* (SURVEY.md §0 parity traps):
//   - SciPy-sparse products (csr_matvec / csr_matmat) and NumPy scalar complex multiply round every product and
//     sum separately          -> cmul_unf, sequential row sums in ascending column order;
//   - NumPy array complex multiply on FMA hardware computes (fma(ar,br,-(ai*bi)), fma(ar,bi,ai*br)) -> cmul_npy;
//   - complex / real divides as multiply by the rounded reciprocal (Smith's algorithm with a zero imaginary part).
// Build with -ffp-contract=off so that nothing else is fused.
//
// Index conventions: stacked index k = q*n + i; complex mismatch row of k is k-1;
//   real row  Re(k) = k-1 (k >= 1),  Im(k) = Nc + k - c (k >= c);   Nc = n*Hn - 1
//   real col  theta(k) = k-1 (k >= 1), V(k) = Nc + k - c (k >= c).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HPF_HD __host__ __device__ __forceinline__
#else
#define HPF_HD inline
#endif

namespace hpf {

struct cplx {
    double re, im;
};

HPF_HD cplx cmul_unf(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
HPF_HD cplx cmul_npy(cplx a, cplx b) { return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)}; }
HPF_HD cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
HPF_HD cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
HPF_HD cplx cconj(cplx a) { return {a.re, -a.im}; }
HPF_HD cplx cneg(cplx a) { return {-a.re, -a.im}; }
HPF_HD cplx cmulj(cplx a) { return {-a.im, a.re}; }   // 1j*a, exact

// Read-only model shared by all scenarios (device pointers inside kernels).
struct Model {
    int n, m, c, Hn, nnz, n_dev, coupled;
    int bus_major = 0;    // layout of the per-scenario voltage arrays U, E (and Vm, Va): 0 stacked index q*n + i (the reference's
                          // order, HG:139-143; host emulation), 1 bus-major i*Hn + q (device: every kernel that walks a bus or a
                          // tree reads the Hn harmonics of a bus as one contiguous run)
    HPF_HD size_t vi(int q, int i) const { return bus_major ? (size_t)i * Hn + q : (size_t)q * n + i; }
    // admittance of stored entry e at harmonic position q: Y[Hn][nnz] (host emulation) or entry-major [nnz][Hn] (device: the
    // Hn harmonics of an entry are contiguous, like the voltages of a bus)
    HPF_HD size_t yi(int q, int e) const { return bus_major ? (size_t)e * Hn + q : (size_t)q * nnz + e; }
    const int* rowptr;    // [n+1]
    const int* col;       // [nnz]
    const int* diag;      // [n] position of the diagonal entry of each row
    const cplx* Y;        // [Hn][nnz]
    const int* dev;       // [n] device type or -1
    const cplx* YN;       // coupled [n_dev][Hn][Hn]; uncoupled [n_dev][Hn]
    const cplx* YNt = nullptr;   // device only, coupled: transposed copy [n_dev][p][q] -- the mismatch kernel's threads of one bus
                                 // (harmonic position q fastest) then read contiguous runs while they walk the columns p
    const cplx* IN;       // [n_dev][Hn]
    const int* rowrec = nullptr; // device only: [n][8] per bus (rowptr[i], rowptr[i+1], col of its first three entries, 0, 0, 0): the mismatch
                                 // kernel finds a row's bounds AND its first neighbours behind ONE 32-byte fetch instead of rowptr -> col
};

// U = Vm*exp(j*Va) (HG:403: real*complex, exact componentwise products) and the "normalised" voltage
//   harmonic NR:     E = U / Vm      (HG:405,422,455; sign-preserving, = U * (1/Vm))
//   fundamental pf:  E = U / |U|     (HG:210)
template <bool FUND>
HPF_HD void polar(double vm, double va, cplx& U, cplx& E) {
    double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(va, &s, &c);
#else
    s = sin(va);
    c = cos(va);
#endif
    U.re = vm * c;
    U.im = vm * s;
    const double scl = 1.0 / (FUND ? hypot(U.re, U.im) : vm);
    E.re = U.re * scl;
    E.im = U.im * scl;
}

// I = sum_e Y[q][e] * U[q*n + col[e]] over row i, ascending column order (csr_matvec; HG:339,345,379).
HPF_HD cplx row_current(const Model& M, const cplx* U, int q, int i) {
    cplx acc = {0.0, 0.0};
    for (int e = M.rowptr[i]; e < M.rowptr[i + 1]; ++e) acc = cadd(acc, cmul_unf(M.Y[M.yi(q, e)], U[M.vi(q, M.col[e])]));
    return acc;
}

// The fundamental power flow multiplies the *dense* Y1 with the voltage vector (HG:198,208: ndarray.dot -> BLAS
// zgemv, transposed kernel for a C-ordered matrix).  OpenBLAS' x86-64 FMA kernel (zgemv_t, 4-column microkernel)
// was characterised against exact-arithmetic emulation: over the first n & ~3 columns it keeps, per output, two
// interleaved partial sums (even / odd column) of the four real products, each an FMA chain, combines each
// partial as (rr - ii, ri + ir) and adds them in order; the last n % 4 columns are accumulated separately as
// t += fma(ar, xr, -(ai*xi)) (and the imaginary analogue) and added last.  Zero entries leave an FMA chain
// unchanged, so only stored entries are visited.
HPF_HD cplx row_current_fund(const Model& M, const cplx* U, int i) {
    const int n1 = M.n & ~3;
    double a[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    cplx t = {0.0, 0.0};
    for (int e = M.rowptr[i]; e < M.rowptr[i + 1]; ++e) {
        const int j = M.col[e];
        const cplx y = M.Y[M.yi(0, e)], u = U[M.vi(0, j)];
        if (j < n1) {
            const int l = j & 1;
            a[l][0] = fma(y.re, u.re, a[l][0]);
            a[l][1] = fma(y.im, u.re, a[l][1]);
            a[l][2] = fma(y.re, u.im, a[l][2]);
            a[l][3] = fma(y.im, u.im, a[l][3]);
        } else {
            t.re += fma(y.re, u.re, -(y.im * u.im));
            t.im += fma(y.re, u.im, y.im * u.re);
        }
    }
    cplx r = {0.0, 0.0};
    r.re += a[0][0] - a[0][3];
    r.im += a[0][2] + a[0][1];
    r.re += a[1][0] - a[1][3];
    r.im += a[1][2] + a[1][1];
    r.re += t.re;
    r.im += t.im;
    return r;
}

// Norton injection of harmonic position q at nonlinear bus i (HG:313-323): I_N[q] - sum_p Y_N[q,p] U[p*n+i].
// The reference evaluates the sum as DataFrame.dot -> BLAS zgemv on an F-ordered matrix (non-transposed kernel).
// OpenBLAS' x86-64 FMA kernel (zgemv_n, 4-column microkernel), characterised the same way: for rows below
// Hn & ~3, columns are taken four at a time; inside a group the four real products (rr, ii, ri, ir) are FMA
// chains, the group contributes (rr - ii, ri + ir) and groups are added in order; the remaining Hn % 4 rows
// are accumulated over all columns as t += fma(ar, xr, -(ai*xi)).  Uncoupled (HG:322: np.diag(Y_N).dot):
// one product, every partial rounded.
HPF_HD cplx norton_injection(const Model& M, const cplx* U, int q, int i) {
    const int d = M.dev[i];
    const cplx in = M.IN[(size_t)d * M.Hn + q];
    cplx acc = {0.0, 0.0};
    if (M.coupled) {
        // Y_N[q, p]: row q of the device's matrix, or column q of its transposed copy (same values, coalesced on the device)
        const cplx* yn = M.YNt ? M.YNt + (size_t)d * M.Hn * M.Hn + q : M.YN + ((size_t)d * M.Hn + q) * M.Hn;
        const size_t ys = M.YNt ? (size_t)M.Hn : 1;
        if (q < (M.Hn & ~3)) {
            for (int p0 = 0; p0 < M.Hn; p0 += 4) {
                double rr = 0, ii = 0, ri = 0, ir = 0;
                const int p1 = p0 + 4 < M.Hn ? p0 + 4 : M.Hn;
                for (int p = p0; p < p1; ++p) {
                    const cplx u = U[M.vi(p, i)], y = yn[p * ys];
                    rr = fma(y.re, u.re, rr);
                    ri = fma(y.re, u.im, ri);
                    ii = fma(y.im, u.im, ii);
                    ir = fma(y.im, u.re, ir);
                }
                acc.re += rr - ii;
                acc.im += ri + ir;
            }
        } else {
            for (int p = 0; p < M.Hn; ++p) {
                const cplx u = U[M.vi(p, i)], y = yn[p * ys];
                acc.re += fma(y.re, u.re, -(y.im * u.im));
                acc.im += fma(y.re, u.im, y.im * u.re);
            }
        }
    } else {
        // np.diag(Y_N).dot(U_bus): the same zgemv_n kernel on a diagonal matrix -> one product per row, rounded
        // like the kernel's row class (unfused partials in the 4-row body, fused in the Hn % 4 tail rows)
        const cplx y = M.YN[(size_t)d * M.Hn + q], u = U[M.vi(q, i)];
        acc = q < (M.Hn & ~3) ? cmul_unf(y, u) : cmul_npy(y, u);
    }
    return csub(in, acc);
}

// Complex mismatch of stacked index k >= 1 (HG:360-390): power balance for linear buses at the fundamental,
// current balance otherwise.  FUND: fundamental power flow (HG:195-202), every bus is a power row.
// Iout (optional): receives the network current I of a power row -- the Jacobian's diagonal power entries (HG:451-459) need
// the same sum again, and the block-tree kernels read it back instead of re-walking the admittance row.
template <bool FUND>
HPF_HD cplx mismatch_row_qi(const Model& M, const cplx* U, const double* P, const double* Q, int q, int i, cplx* Iout = nullptr) {
    const cplx I = FUND ? row_current_fund(M, U, i) : row_current(M, U, q, i);
    if (FUND || (q == 0 && i < M.m)) {
        if (Iout) Iout[i] = I;
        // V_i * conj(Y_ij @ V_j): NumPy array multiply (HG:198,379), then + S
        const cplx sl = cmul_npy(U[M.vi(0, i)], cconj(I));
        return {P[i] + sl.re, Q[i] + sl.im};
    }
    if (i >= M.m) return cadd(I, norton_injection(M, U, q, i));   // HG:351,354
    return I;
}

template <bool FUND>
HPF_HD cplx mismatch_row(const Model& M, const cplx* U, const double* P, const double* Q, int k, cplx* Iout = nullptr) {
    const int q = FUND ? 0 : k / M.n;
    const int i = FUND ? k : k - q * M.n;
    return mismatch_row_qi<FUND>(M, U, P, Q, q, i, Iout);
}

// ---- Jacobian ---------------------------------------------------------------------------------------------
// One stored admittance entry e = (i, j) at harmonic position q contributes a 2x2 real block
//   [ dRe/dtheta  dRe/dV ; dIm/dtheta  dIm/dV ]  of complex row k = q*n+i w.r.t. column k' = q*n+j.
// Emit is a functor  emit(k_row, t_row, k_col, t_col, value)  with t_row 0=Re,1=Im and t_col 0=theta,1=V; it
// drops combinations that are not unknowns/equations (k_row < 1 for Re, < c for Im; k_col < 1 / < c).
struct Blk2 {
    cplx dA, dV;   // complex derivative w.r.t. angle and magnitude; the real block is (Re dA, Re dV; Im dA, Im dV)
};

// Current-balance rows (k >= m), HG:403-411 and the p == h Norton terms of HG:432-435 / HG:442-443.
HPF_HD Blk2 jac_current_entry(const Model& M, const cplx* U, const cplx* E, int q, int i, int j, int e) {
    const cplx y = M.Y[M.yi(q, e)];
    const size_t kc = M.vi(q, j);
    Blk2 b;
    b.dV = cmul_unf(y, E[kc]);               // Y_diag @ V_norm_diag
    b.dA = cmul_unf(cmulj(y), U[kc]);        // (1j*Y_diag) @ V_diag
    if (i == j && i >= M.m) {
        const int d = M.dev[i];
        const cplx yn = M.coupled ? M.YN[((size_t)d * M.Hn + q) * M.Hn + q] : M.YN[(size_t)d * M.Hn + q];
        b.dV = csub(b.dV, cmul_unf(yn, E[kc]));
        b.dA = csub(b.dA, cmul_unf(cmulj(yn), U[kc]));
    }
    return b;
}

// Diagonal current-balance entry (i == j) with the device type of the bus supplied by the caller (-1: linear bus).
HPF_HD Blk2 jac_current_diag(const Model& M, const cplx* U, const cplx* E, int q, int i, int e, int d) {
    const cplx y = M.Y[M.yi(q, e)];
    const size_t kc = M.vi(q, i);
    Blk2 b;
    b.dV = cmul_unf(y, E[kc]);
    b.dA = cmul_unf(cmulj(y), U[kc]);
    if (i >= M.m) {
        const cplx yn = M.coupled ? M.YN[((size_t)d * M.Hn + q) * M.Hn + q] : M.YN[(size_t)d * M.Hn + q];
        b.dV = csub(b.dV, cmul_unf(yn, E[kc]));
        b.dA = csub(b.dA, cmul_unf(cmulj(yn), U[kc]));
    }
    return b;
}

// Coupled Norton cross terms q != p at nonlinear bus i (HG:425-435): entry [q*n+i, p*n+i] = 0 - Y_N[q,p]*...
HPF_HD Blk2 jac_norton_cross(const Model& M, const cplx* U, const cplx* E, int q, int p, int i) {
    const int d = M.dev[i];
    const cplx yn = M.YN[((size_t)d * M.Hn + q) * M.Hn + p];
    const size_t kc = M.vi(p, i);
    Blk2 b;
    b.dV = cneg(cmul_unf(yn, E[kc]));
    b.dA = cneg(cmul_unf(cmulj(yn), U[kc]));
    return b;
}

// Power rows at the fundamental (HG:451-459 for the harmonic NR, HG:207-214 for pf): entry (i, j) of
//   dSdA = 1j*diag(U) @ conj(diag(I) - Y1 @ diag(U)),   dSdV = diag(E) @ conj(diag(I)) + diag(U) @ conj(Y1 @ diag(E)).
// (jac_power_diag: the diagonal entry with the row current I supplied by the caller)
HPF_HD Blk2 jac_power_diag(const Model& M, const cplx* U, const cplx* E, int i, int e, cplx I) {
    const cplx y = M.Y[M.yi(0, e)];
    const cplx ui = U[M.vi(0, i)], ei = E[M.vi(0, i)];
    const cplx yu = cmul_unf(y, ui);
    const cplx ye = cmul_unf(y, ei);
    Blk2 b;
    const cplx t = csub(I, yu);
    b.dV = cadd(cmul_unf(ei, cconj(I)), cmul_unf(ui, cconj(ye)));
    b.dA = cmul_unf(cmulj(ui), cconj(t));
    return b;
}

template <bool FUND>
HPF_HD Blk2 jac_power_entry(const Model& M, const cplx* U, const cplx* E, int i, int j, int e) {
    const cplx y = M.Y[M.yi(0, e)];
    const cplx ui = U[M.vi(0, i)];
    const cplx yu = cmul_unf(y, U[M.vi(0, j)]);
    const cplx ye = cmul_unf(y, E[M.vi(0, j)]);
    Blk2 b;
    cplx t;
    if (i == j) {
        const cplx I = FUND ? row_current_fund(M, U, i) : row_current(M, U, 0, i);
        t = csub(I, yu);
        b.dV = cadd(cmul_unf(E[M.vi(0, i)], cconj(I)), cmul_unf(ui, cconj(ye)));
    } else {
        t = cneg(yu);
        b.dV = cmul_unf(ui, cconj(ye));
    }
    b.dA = cmul_unf(cmulj(ui), cconj(t));
    return b;
}

// All Jacobian contributions of stored entry e of row i at harmonic position q (harmonic NR).
template <class Emit>
HPF_HD void jac_entry(const Model& M, const cplx* U, const cplx* E, int q, int i, int e, Emit& emit) {
    const int j = M.col[e];
    const int kr = q * M.n + i, kc = q * M.n + j;
    Blk2 b;
    if (q == 0 && i < M.m) {
        if (i == 0) return;                              // slack bus has no equation at the fundamental
        b = jac_power_entry<false>(M, U, E, i, j, e);
    } else {
        b = jac_current_entry(M, U, E, q, i, j, e);
    }
    emit(kr, 0, kc, 0, b.dA.re);
    emit(kr, 0, kc, 1, b.dV.re);
    emit(kr, 1, kc, 0, b.dA.im);
    emit(kr, 1, kc, 1, b.dV.im);
}

template <class Emit>
HPF_HD void jac_cross(const Model& M, const cplx* U, const cplx* E, int q, int p, int i, Emit& emit) {
    const Blk2 b = jac_norton_cross(M, U, E, q, p, i);
    const int kr = q * M.n + i, kc = p * M.n + i;
    emit(kr, 0, kc, 0, b.dA.re);
    emit(kr, 0, kc, 1, b.dV.re);
    emit(kr, 1, kc, 0, b.dA.im);
    emit(kr, 1, kc, 1, b.dV.im);
}

// Fundamental power-flow Jacobian entry (HG:205-223): every bus 1..n-1 is a power row, harmonic position 0 only.
template <class Emit>
HPF_HD void jac_entry_fund(const Model& M, const cplx* U, const cplx* E, int i, int e, Emit& emit) {
    if (i == 0) return;
    const int j = M.col[e];
    const Blk2 b = jac_power_entry<true>(M, U, E, i, j, e);
    emit(i, 0, j, 0, b.dA.re);
    emit(i, 0, j, 1, b.dV.re);
    emit(i, 1, j, 0, b.dA.im);
    emit(i, 1, j, 1, b.dV.im);
}

// Dense column-major target in the reference's row/column order (HG:469-472).  Nc = number of complex rows.
struct DenseEmit {
    double* J;
    int N, Nc, c;
    HPF_HD void operator()(int kr, int tr, int kc, int tc, double v) const {
        if (tr == 0 ? kr < 1 : kr < c) return;
        if (tc == 0 ? kc < 1 : kc < c) return;
        const int r = tr == 0 ? kr - 1 : Nc + kr - c;
        const int col = tc == 0 ? kc - 1 : Nc + kc - c;
        J[(size_t)col * N + r] = v;
    }
};

// ---- The Jacobian as the reference returns it: the stacked real matrix of HG:469-472 in CSR form (scipy.sparse.csr_matrix) ---------
// Real row r: r < Nc -> Re part of complex row k = r + 1; else Im part of k = r - Nc + c.  The columns of a row, ascending: the theta
// columns (kc - 1) of its complex entries, then the V columns (Nc + kc - c); the complex entries of row k = q*n + i in ascending
// stacked column kc = p*n + j: for p < q the Norton cross term (p, i) of a nonlinear bus (HG:425-435), at p = q the stored
// admittance entries of row i (HG:403-411 / power rows HG:451-462), for p > q the cross terms again.
// Which entries EXIST follows the reference's construction: `block_diag` of the dense per-harmonic admittance arrays keeps the
// non-zero values only (HG:407), the diagonal of a power row comes from `diags(...)` (HG:454-459: always stored), the diagonal of a
// nonlinear bus and its cross terms from the `-=` on the lil_matrix (HG:432-435: stored unless the product is exactly zero, i.e.
// unless Y_N[q,p] is).  State-dependent exact cancellations are not tracked: the pattern is a property of the model.
struct JRow {
    int k, t, q, i;
    bool power, cross;
};

HPF_HD JRow jcsr_row(const Model& M, int Nc, int r) {
    JRow R;
    R.t = r < Nc ? 0 : 1;
    R.k = R.t ? r - Nc + M.c : r + 1;
    R.q = R.k / M.n;
    R.i = R.k - R.q * M.n;
    R.power = R.q == 0 && R.i < M.m;
    R.cross = M.coupled && R.i >= M.m;
    return R;
}

HPF_HD bool jcsr_has_entry(const Model& M, const JRow& R, int e, int j) {
    if (j == R.i && (R.power || R.i >= M.m)) return true;
    const cplx y = M.Y[M.yi(R.q, e)];
    return y.re != 0.0 || y.im != 0.0;
}

HPF_HD bool jcsr_has_cross(const Model& M, const JRow& R, int p) {
    const cplx y = M.YN[((size_t)M.dev[R.i] * M.Hn + R.q) * M.Hn + p];
    return y.re != 0.0 || y.im != 0.0;
}

// visit the complex entries of row R in ascending stacked column order: f(kc, e, j, p) with e >= 0 for an admittance entry (i, j),
// e = -1 for the Norton cross term of harmonic position p
template <class F>
HPF_HD void jcsr_walk(const Model& M, const JRow& R, F& f) {
    for (int p = R.cross ? 0 : R.q; p < (R.cross ? M.Hn : R.q + 1); ++p) {
        if (p == R.q) {
            for (int e = M.rowptr[R.i]; e < M.rowptr[R.i + 1]; ++e) {
                const int j = M.col[e];
                if (jcsr_has_entry(M, R, e, j)) f(p * M.n + j, e, j, p);
            }
        } else if (jcsr_has_cross(M, R, p)) {
            f(p * M.n + R.i, -1, R.i, p);
        }
    }
}

struct JCount {
    int c, n_theta, n_v;
    HPF_HD void operator()(int kc, int, int, int) {
        n_theta += kc >= 1;
        n_v += kc >= c;
    }
};

// stored entries of real row r (theta columns, V columns)
HPF_HD JCount jcsr_count_row(const Model& M, int Nc, int r) {
    const JRow R = jcsr_row(M, Nc, r);
    JCount cnt{M.c, 0, 0};
    jcsr_walk(M, R, cnt);
    return cnt;
}

struct JFill {
    const Model& M;
    const cplx *U, *E;
    JRow R;
    int Nc;
    int* col;
    double* val;
    long long pos_t, pos_v;
    HPF_HD void operator()(int kc, int e, int j, int p) {
        const Blk2 b = e >= 0 ? (R.power ? jac_power_entry<false>(M, U, E, R.i, j, e) : jac_current_entry(M, U, E, R.q, R.i, j, e))
                              : jac_norton_cross(M, U, E, R.q, p, R.i);
        if (kc >= 1) {
            if (col) col[pos_t] = kc - 1;
            val[pos_t++] = R.t ? b.dA.im : b.dA.re;
        }
        if (kc >= M.c) {
            if (col) col[pos_v] = Nc + kc - M.c;
            val[pos_v++] = R.t ? b.dV.im : b.dV.re;
        }
    }
};

// write real row r: its entries start at `start` (= indptr[r]); col may be nullptr (values only)
HPF_HD void jcsr_fill_row(const Model& M, const cplx* U, const cplx* E, int Nc, int r, long long start, int* col, double* val) {
    const JCount cnt = jcsr_count_row(M, Nc, r);
    JFill fl{M, U, E, jcsr_row(M, Nc, r), Nc, col, val, start, start + cnt.n_theta};
    jcsr_walk(M, fl.R, fl);
}

// Real mismatch vector layout f = [Re f_c ; Im f_c[c-1:]] (HG:388).
HPF_HD void store_mismatch(double* f, int Nc, int c, int k, cplx v) {
    f[k - 1] = v.re;
    if (k >= c) f[Nc + k - c] = v.im;
}

}  // namespace hpf
