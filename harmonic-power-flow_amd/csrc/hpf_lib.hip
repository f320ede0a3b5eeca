// libhpf.so — C ABI (include/hpf.h), assembly kernels and the dense (rocSOLVER) Newton step.
//
// Data layout in HBM (all FP64 / complex128, scenario index slowest):
//   model, shared by all scenarios:  rowptr[n+1], col[nnz], erow[nnz] (row of each stored entry), diag[n],
//       Y[Hn][nnz] complex (one CSR pattern for all harmonics: lanes sweep the entries of one harmonic, so the
//       admittance read of the Jacobian kernel is a fully coalesced 16 B/lane stream), dev[n],
//       Y_N[n_dev][Hn][Hn], I_N[n_dev][Hn].
//   state: Vm,Va[S][Hn*n]; U,E[S][Hn*n] complex (polar -> rectangular once per iteration, reused by mismatch and
//       Jacobian; stacked harmonic-major so that for a fixed harmonic consecutive lanes touch consecutive buses);
//       P,Q[S][n]; f[S][N]; dense J[S][N*N] column-major in the reference's row/column order (HG:469-472).
// Kernels and what bounds them (algorithmic bytes per NR iteration per scenario):
//   k_mismatch   HBM: 16*Hn*nnz (Y) + 16*Hn*n (U) + 8*N (f) + pattern;   k_jac_*  HBM: Y + U,E + 32*E_cplx written;
//   dense solve  FP64 matrix pipe: 2/3 N^3 + 2 N^2 flop (rocSOLVER getrf/getrs).
#include <math.h>
#include <chrono>
#include <functional>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rocsolver/rocsolver.h>

#include "hpf_internal.hpp"

using namespace hpf;

#define HIPCHK(expr)                              \
    do {                                          \
        hipError_t _e = (expr);                   \
        if (_e != hipSuccess) {                   \
            h->last_detail = (int)_e;             \
            return HPF_E_HIP;                     \
        }                                         \
    } while (0)

#define BLASCHK(expr)                             \
    do {                                          \
        rocblas_status _s = (expr);               \
        if (_s != rocblas_status_success) {       \
            h->last_detail = (int)_s;             \
            return HPF_E_ROCSOLVER;               \
        }                                         \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int TPB = 256;

// polar -> rectangular for `count` stacked entries of every scenario (count = Hn*n, or n for the fundamental pf)
// Device layout of Vm, Va, U, E: bus-major, entry (bus i, harmonic position q) at i*Hn + q (Model::vi); the C ABI keeps the
// reference's stacked order and hpf_set_state / hpf_get_state transpose.  FUND: the n entries of harmonic position 0.
template <bool FUND>
__global__ void k_polar(int count, int stride, int Hn, const double* __restrict__ Vm, const double* __restrict__ Va,
                        cplx* __restrict__ U, cplx* __restrict__ E) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= count) return;
    const size_t o = (size_t)blockIdx.y * stride + (FUND ? (size_t)k * Hn : (size_t)k);
    cplx u, e;
    polar<FUND>(Vm[o], Va[o], u, e);
    U[o] = u;
    E[o] = e;
}

// XCD-aware launch geometry of the per-element kernels (thread per (bus, harmonic) of one scenario): workgroups are dealt round-robin
// over the 8 XCDs by their linear id (MI355X_MICROARCH.md, workgroup dispatch), and every XCD has its own 4 MiB L2.  A 1-D grid
// whose id is (scenario block, x block, scenario mod 8) keeps ALL workgroups of a scenario on one XCD, so the scenario's voltages --
// gathered again by the neighbours' rows and by the Norton rows of the same bus -- are fetched into one L2 instead of eight.
// With fewer than 8 scenarios that placement would leave XCDs empty (ONE scenario: the whole kernel on 32 of the 256 CUs -- 196 us
// instead of 30 for the mismatch of the 10 000-bus x 49-harmonic feeder): the workgroups are then dealt over the whole chip.
// ... realised WITHOUT index arithmetic through a 3-D grid (the linear workgroup id is x + gx (y + gy z)): S >= 8: grid (8, nbx, ceil(S/8))
// -> x = scenario mod 8 = the XCD, y = x block, z = scenario block; S < 8: grid (nbx, S, 1).  (The decode of a 1-D id took two integer
// divisions by run-time values per workgroup: ~50 scalar instructions and two quarter-rate reciprocals in front of the first load.)
__device__ __forceinline__ bool xcd_map(int S, int& bx, int& slot) {
    if (S < 8) {
        bx = blockIdx.x;
        slot = blockIdx.y;
        return true;
    }
    bx = blockIdx.y;
    slot = blockIdx.z * 8 + blockIdx.x;
    return slot < S;
}
__host__ inline dim3 xcd_grid(int nbx, int S) {
    return S < 8 ? dim3((unsigned)nbx, (unsigned)S, 1) : dim3(8, (unsigned)nbx, (unsigned)((S + 7) / 8));
}
// unsigned division of t < 2^32 / d by a run-time d through its reciprocal m = floor(2^32 / d) + 1 (host: div_magic): exact there
// (hpf_create refuses models with (n Hn + 256) Hn >= 2^32).  d = 1 has no 32-bit reciprocal: magic 0 stands for "t itself" (a model
// with the fundamental alone, H_MAX = 1 or 2).
__device__ __forceinline__ int div_by(int t, unsigned magic) { return magic ? (int)__umulhi((unsigned)t, magic) : t; }
__host__ inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1ull); }

// ||.||_inf with NaN propagation: |x| as its IEEE bit pattern is monotone for non-negative doubles, and every NaN
// pattern compares above +inf, so an unsigned max reproduces np.linalg.norm(f, inf) including its NaN result
// (HG:389) and is independent of the reduction order.
__device__ __forceinline__ unsigned long long abs_bits(double v) {
    return (unsigned long long)__double_as_longlong(fabs(v));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Norton injection of harmonic position q (HG:313-323: I_N[q] - sum_p Y_N[q,p] U_p) with both operands in LDS: the device type's
// Y_N^T (ynl[p*Hn + q]) and the bus's Hn voltages (ul[p]).  Same operations in the same order as norton_injection (hpf_assembly.hpp):
// the reference's zgemv_n rounding -- 4-column groups of FMA chains, groups added in order; the Hn % 4 tail rows fused per column.
__device__ __forceinline__ cplx norton_injection_lds(const Model& M, int d, const cplx* __restrict__ ynl, const cplx* __restrict__ ul, int q) {
    const int Hn = M.Hn;
    const cplx in = M.IN[(size_t)d * Hn + q];
    cplx acc = {0.0, 0.0};
    const cplx* yq = ynl + q;
    if (q < (Hn & ~3)) {
        int p0 = 0;
        for (; p0 + 4 <= Hn; p0 += 4) {                  // a full group: its eight LDS reads first, then the four FMA chains
            cplx y4[4], u4[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                y4[pp] = yq[(p0 + pp) * Hn];
                u4[pp] = ul[p0 + pp];
            }
            double rr = 0, ii = 0, ri = 0, ir = 0;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                rr = fma(y4[pp].re, u4[pp].re, rr);
                ri = fma(y4[pp].re, u4[pp].im, ri);
                ii = fma(y4[pp].im, u4[pp].im, ii);
                ir = fma(y4[pp].im, u4[pp].re, ir);
            }
            acc.re += rr - ii;
            acc.im += ri + ir;
        }
        if (p0 < Hn) {                                   // the last, shorter group
            double rr = 0, ii = 0, ri = 0, ir = 0;
            for (int p = p0; p < Hn; ++p) {
                const cplx u = ul[p], y = yq[p * Hn];
                rr = fma(y.re, u.re, rr);
                ri = fma(y.re, u.im, ri);
                ii = fma(y.im, u.im, ii);
                ir = fma(y.im, u.re, ir);
            }
            acc.re += rr - ii;
            acc.im += ri + ir;
        }
    } else {
        for (int p = 0; p < Hn; ++p) {
            const cplx u = ul[p], y = yq[p * Hn];
            acc.re += fma(y.re, u.re, -(y.im * u.im));
            acc.im += fma(y.re, u.im, y.im * u.re);
        }
    }
    return csub(in, acc);
}

// harmonic_mismatch (HG:360-390).  One workgroup = one scenario x a tile of consecutive buses (thread t = i*Hn + q: bus-major, so a
// workgroup's 256 rows are ~10 whole buses and their voltages one contiguous run).  Staged in LDS per workgroup (coupled Norton data):
// the device type's Y_N^T (Hn x Hn complex: 10.8 KB at K = 25, shared by every bus of that type -- L2-resident) and the tile's bus
// voltages; the Norton coupling rows of a nonlinear bus (HG:313-323: the only O(Hn^2) part of the mismatch) then run out of LDS.
// The network part walks the CSR row (ascending columns, csr_matvec order) with 16-byte gathers of the neighbours' voltages, which
// the XCD-aware placement (xcd_map) keeps in ONE L2 per scenario.  ||f||_inf: wave shuffle + one partial maximum per wavefront.
// f: the reference's stacked real layout (HG:388; dense solver, C ABI) or nullptr; fb: bus-major image [bus][2q + (Re|Im)] with
// stride Bst and zeros where there is no equation (tree kernels) or nullptr.
template <bool FUND>
__global__ __launch_bounds__(TPB, 8) void k_mismatch(Model M, int count, int N, int Nc, const int* __restrict__ active, const cplx* __restrict__ U,
                           const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ f,
                           unsigned long long* __restrict__ errpart, int pstride, cplx* __restrict__ I0, double* __restrict__ fb, int Bst,
                           int s0, int S_cnt, unsigned hn_magic) {
    extern __shared__ cplx mm_lds[];                    // [Hn*Hn] Y_N^T of the tile's first device type | [tile buses][Hn] voltages
    int bx, slot;
    if (!xcd_map(S_cnt, bx, slot)) return;
    const int s = active ? active[slot + s0] : slot + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tid = threadIdx.x;
    const int t = bx * TPB + tid;
    const cplx* Us = U + (size_t)s * M.n * M.Hn;
    const int Hn = M.Hn;
    const bool live = t < count;
    const int i = live ? (FUND ? t : div_by(t, hn_magic)) : M.n - 1, q = (FUND || !live) ? 0 : t - i * Hn;
    // The kernel is bound by the chain of dependent fetches of a wavefront, so the harmonic variant issues them in BATCHES of independent,
    // branch-free loads (indices clamped into the row, values masked afterwards): A = the row bounds, Y_N^T and the tile's bus voltages
    // (for LDS); B = Y of the row's first PF entries AND the neighbours' voltages (their columns come with the row bounds in the bus's row
    // record: two dependent round trips, not three).  Longer rows finish in a loop.  Same
    // operations in the same order as mismatch_row / row_current: bit-identical results.
    constexpr int PF = 3;                               // (a feeder row holds 3 entries on average)
    int d0 = -1, i_first = 0;
    int e0 = 0, e1 = 0;
    cplx yv[PF], ug[PF];
    if (!FUND) {
        const int4 rr = reinterpret_cast<const int4*>(M.rowrec)[2 * i];          // (rowptr[i], rowptr[i+1], col[e0], col[e0+1])
        const int rr2 = M.rowrec[8 * i + 4];                                      //  col[e0+2]
        e0 = rr.x;
        e1 = rr.y;
        const bool has_nl = M.coupled && M.YNt;
        i_first = div_by(bx * TPB, hn_magic);
        int i_last = div_by(bx * TPB + TPB - 1, hn_magic);
        if (i_last > M.n - 1) i_last = M.n - 1;
        const bool stage = has_nl && i_last >= M.m;     // the tile holds nonlinear buses (they come last in the bus order, HG:83)
        const int nyn = Hn * Hn, nv = (i_last - i_first + 1) * Hn;          // nv <= 2 TPB
        double2 tv0 = {0.0, 0.0}, tv1 = {0.0, 0.0};
        if (stage) {
            d0 = M.dev[i_first > M.m ? i_first : M.m];
            const double2* Ut = reinterpret_cast<const double2*>(Us + (size_t)i_first * Hn);
            tv0 = Ut[tid < nv ? tid : 0];
            tv1 = Ut[tid + TPB < nv ? tid + TPB : 0];
            const double2* src = reinterpret_cast<const double2*>(M.YNt + (size_t)d0 * nyn);
            double2* dst = reinterpret_cast<double2*>(mm_lds);
            for (int base = tid; base < nyn; base += 4 * TPB) {                 // (26 harmonics: one pass of three loads in flight)
                const double2 y0 = src[base], y1 = src[base + TPB < nyn ? base + TPB : 0], y2 = src[base + 2 * TPB < nyn ? base + 2 * TPB : 0],
                              y3 = src[base + 3 * TPB < nyn ? base + 3 * TPB : 0];
                dst[base] = y0;
                if (base + TPB < nyn) dst[base + TPB] = y1;
                if (base + 2 * TPB < nyn) dst[base + 2 * TPB] = y2;
                if (base + 3 * TPB < nyn) dst[base + 3 * TPB] = y3;
            }
        }
        static_assert(PF == 3, "row records hold the first three neighbours");
        const int jc[PF] = {rr.z, rr.w, rr2};               // (clamped into the row by the host: entries past the row's end repeat its last)
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = e0 + u < e1 ? e0 + u : e1 - 1;
            yv[u] = M.Y[(size_t)e * Hn + q];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) ug[u] = Us[(size_t)jc[u] * Hn + q];
        if (stage) {
            double2* ul = reinterpret_cast<double2*>(mm_lds + nyn);
            if (tid < nv) ul[tid] = tv0;
            if (tid + TPB < nv) ul[tid + TPB] = tv1;
        }
        if (has_nl) __syncthreads();
    }
    unsigned long long b = 0;
    if (live) {
        const int k = q * M.n + i;
        cplx v = {0.0, 0.0};
        if (k >= 1) {
            if (FUND) {
                v = mismatch_row_qi<true>(M, Us, P + (size_t)s * M.n, Q + (size_t)s * M.n, 0, i, I0 ? I0 + (size_t)s * M.n : nullptr);   // (pf's tree kernels read the row currents back)
            } else {
                cplx I = {0.0, 0.0};                    // row_current: ascending columns, every product and sum rounded (csr_matvec)
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const cplx nx = cadd(I, cmul_unf(yv[u], ug[u]));
                    const bool has = e0 + u < e1;       // (component selects: a select of the struct goes through scratch)
                    I.re = has ? nx.re : I.re;
                    I.im = has ? nx.im : I.im;
                }
                for (int e = e0 + PF; e < e1; ++e) I = cadd(I, cmul_unf(M.Y[(size_t)e * Hn + q], Us[(size_t)M.col[e] * Hn + q]));
                if (q == 0 && i < M.m) {                // power row (HG:372-380)
                    if (I0) I0[(size_t)s * M.n + i] = I;
                    const cplx sl = cmul_npy(Us[(size_t)i * Hn], cconj(I));
                    v = {P[(size_t)s * M.n + i] + sl.re, Q[(size_t)s * M.n + i] + sl.im};
                } else if (d0 >= 0 && i >= M.m && M.dev[i] == d0) {
                    // current-balance row of a nonlinear bus (HG:351,354): network current + Norton injection out of LDS
                    v = cadd(I, norton_injection_lds(M, d0, mm_lds, mm_lds + Hn * Hn + (i - i_first) * Hn, q));
                } else if (i >= M.m) {
                    v = cadd(I, norton_injection(M, Us, q, i));
                } else {
                    v = I;
                }
            }
            if (f) store_mismatch(f + (size_t)s * N, Nc, M.c, k, v);
            b = abs_bits(v.re);
            if (k >= M.c) {
                const unsigned long long bi = abs_bits(v.im);
                b = bi > b ? bi : b;
            }
        }
        if (!FUND && fb) {
            double* o = fb + ((size_t)s * M.n + i) * Bst + 2 * q;
            *reinterpret_cast<double2*>(o) = double2{k >= 1 ? v.re : 0.0, k >= M.c ? v.im : 0.0};
        }
    }
    // ||f||_inf of the scenario: every WAVEFRONT leaves its partial maximum (no LDS, no workgroup barrier, no atomic -- 102 workgroups of a
    // scenario used to meet on one L2 line); the consumer of the norm (k_finalize, k_queue_first, k_err_reduce) takes the maximum of the
    // scenario's partials.  A maximum does not depend on the order: the result is the same word as before.
    b = wave_max_u64(b);
    if ((tid & 63) == 0) errpart[(size_t)s * pstride + (size_t)bx * (TPB / 64) + (tid >> 6)] = b;
}
// Dense Jacobian, network entries: one thread per (harmonic position, stored admittance entry) of one scenario.
template <bool FUND>
__global__ void k_jac_dense(Model M, int total, int N, int Nc, size_t J_stride, const int* __restrict__ active,
                            const int* __restrict__ erow, const cplx* __restrict__ U, const cplx* __restrict__ E,
                            double* __restrict__ J) {
    const int s = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (s < 0) return;
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= total) return;
    const int q = t / M.nnz, e = t - q * M.nnz;
    const int i = erow[e];
    DenseEmit em{J + (size_t)s * J_stride, N, Nc, M.c};
    const size_t so = (size_t)s * M.n * M.Hn;
    if (FUND)
        jac_entry_fund(M, U + so, E + so, i, e, em);
    else
        jac_entry(M, U + so, E + so, q, i, e, em);
}

// Dense Jacobian, coupled Norton cross terms q != p at nonlinear buses (HG:425-435).
__global__ void k_jac_cross_dense(Model M, int total, int N, int Nc, size_t J_stride, const int* __restrict__ active,
                                  const cplx* __restrict__ U, const cplx* __restrict__ E, double* __restrict__ J) {
    const int s = active ? active[blockIdx.y] : (int)blockIdx.y;
    if (s < 0) return;
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= total) return;
    // consecutive threads -> consecutive buses (coalesced U/E reads for a fixed column harmonic p)
    const int nnl = M.n - M.m;
    const int i = M.m + t % nnl;
    const int qp = t / nnl;
    const int q = qp / M.Hn, p = qp - q * M.Hn;
    if (p == q) return;
    DenseEmit em{J + (size_t)s * J_stride, N, Nc, M.c};
    const size_t so = (size_t)s * M.n * M.Hn;
    jac_cross(M, U + so, E + so, q, p, i, em);
}

// The Jacobian in CSR form (HG:469-472 as the reference returns it, hpf_jacobian_csr): one thread per real row.  k_jcsr_count leaves
// the row lengths, k_jcsr_scan turns them into indptr (one workgroup: chunk sums, LDS scan of the 1 024 partials, chunk prefixes),
// k_jcsr_fill writes column indices and values of scenario `s` behind indptr[r] (per-entry arithmetic: hpf_assembly.hpp, the same
// functions as the dense target).
__global__ void k_jcsr_count(Model M, int N, int Nc, int* __restrict__ cnt) {
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= N) return;
    const JCount c = jcsr_count_row(M, Nc, r);
    cnt[r] = c.n_theta + c.n_v;
}

__global__ __launch_bounds__(1024) void k_jcsr_scan(int N, int* __restrict__ indptr, long long* __restrict__ total) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int chunk = (N + 1023) / 1024;
    const int b = tid * chunk < N ? tid * chunk : N, e = b + chunk < N ? b + chunk : N;
    long long s = 0;
    for (int i = b; i < e; ++i) s += indptr[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long base = tid ? part[tid - 1] : 0;
    for (int i = b; i < e; ++i) {
        const int c = indptr[i];
        indptr[i] = (int)base;
        base += c;
    }
    if (tid == 1023) {
        *total = part[1023];
        indptr[N] = part[1023] < 0x7fffffffll ? (int)part[1023] : 0x7fffffff;
    }
}

__global__ void k_jcsr_fill(Model M, int N, int Nc, const int* __restrict__ indptr, const cplx* __restrict__ U,
                            const cplx* __restrict__ E, int* __restrict__ col, double* __restrict__ val) {
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= N) return;
    jcsr_fill_row(M, U, E, Nc, r, indptr[r], col, val);
}

// x <- x - step, scattered back into (Va, Vm) (HG:478,484-485 / HG:229,234-235), then refresh U, E of the entry.
template <bool FUND>
__global__ void k_update(int n, int Hn, int c, int count, int stride, int N, int Nc, const int* __restrict__ active,
                         const double* __restrict__ step, double* __restrict__ Vm, double* __restrict__ Va,
                         cplx* __restrict__ U, cplx* __restrict__ E,
                         const double* __restrict__ xbus, int Bst, int s0, unsigned hn_magic) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= count) return;
    const int i = FUND ? t : div_by(t, hn_magic), q = FUND ? 0 : t - i * Hn;          // thread t = i*Hn + q: bus-major state arrays
    const int k = q * n + i;                                             // stacked index (HG:139-143)
    const size_t o = (size_t)s * stride + (size_t)i * Hn + q;
    double va = Va[o], vm = Vm[o];
    if (xbus) {
        // multi-wave block-tree sweep: the Newton step stays in its bus-major image [bus][2q+t] (no scattered copy into the
        // reference's stacked order by the back-substitution kernels)
        const double2 dx = *reinterpret_cast<const double2*>(xbus + ((size_t)s * n + i) * Bst + 2 * q);
        if (k >= 1) va = va - dx.x;
        if (k >= c) vm = vm - dx.y;
    } else {
        const double* d = step + (size_t)s * N;
        if (k >= 1) va = va - d[k - 1];
        if (k >= c) vm = vm - d[Nc + k - c];
    }
    Va[o] = va;
    Vm[o] = vm;
    cplx u, e;
    polar<FUND>(vm, va, u, e);
    U[o] = u;
    E[o] = e;
}

// Per-scenario bookkeeping of the NR loop (HG:536-542 / HG:259-265).  The set of running scenarios is a SLOT LIST: active[i] =
// scenario id that slot i runs, or -1 (frozen scenario / empty slot); every kernel of the loop maps its blockIdx.y through it.
// first: slot i <- scenario i, record the initial mismatch, apply the stop rule (mask: a repeat pass starts masked scenarios only).
// else: one thread per slot; a scenario that meets the stop rule freezes (its slot becomes -1).
// max over the np partial maxima a scenario's mismatch launch left (k_mismatch), taken by one wavefront; every lane receives it
__device__ __forceinline__ unsigned long long slot_err_bits(const unsigned long long* __restrict__ part, int np) {
    unsigned long long b = 0ull;
    for (int i = threadIdx.x & 63; i < np; i += 64) {
        const unsigned long long v = part[i];
        b = v > b ? v : b;
    }
    return wave_max_u64(b);
}

// (one wavefront per slot: it first reduces the scenario's partial maxima)
__global__ __launch_bounds__(64) void k_finalize(int S, int first, double thresh, int max_iter, int hist_cap, int hist_off,
                           const unsigned long long* __restrict__ errpart, int pstride, int np, double* __restrict__ err,
                           int* __restrict__ niter, int* __restrict__ active, int* __restrict__ nactive,
                           double* __restrict__ hist, int s0, const int* __restrict__ mask) {
    const int sl = blockIdx.x;
    if (sl >= S) return;
    const int slot = sl + s0;
    if (first) {
        const int s = slot;
        if (mask && !mask[s]) {           // repeat pass: the other scenarios keep their result and stay frozen
            if (threadIdx.x == 0) active[slot] = -1;
            return;
        }
        const double e = __longlong_as_double((long long)slot_err_bits(errpart + (size_t)s * pstride, np));
        if (threadIdx.x != 0) return;
        err[s] = e;
        niter[s] = 0;
        if (hist && hist_off == 0) hist[(size_t)s * hist_cap] = e;
        const int a = (e > thresh) && (0 < max_iter);
        active[slot] = a ? s : -1;
        if (a) atomicAdd(nactive, 1);
        return;
    }
    const int s = active[slot];
    if (s < 0) return;
    const double e = __longlong_as_double((long long)slot_err_bits(errpart + (size_t)s * pstride, np));
    if (threadIdx.x != 0) return;
    const int it = niter[s] + 1;
    err[s] = e;
    niter[s] = it;
    if (hist) hist[(size_t)s * hist_cap + it - 1 + (hist_off == 0 ? 1 : 0)] = e;
    const int a = (e > thresh) && (it < max_iter);
    if (!a) active[slot] = -1;
}

// ||f||_inf of every scenario -> out[s] as the bit pattern of the double (hpf_mismatch: the host copies it)
__global__ __launch_bounds__(64) void k_err_reduce(const unsigned long long* __restrict__ errpart, int pstride, int np,
                                                   unsigned long long* __restrict__ out) {
    const unsigned long long b = slot_err_bits(errpart + (size_t)blockIdx.x * pstride, np);
    if (threadIdx.x == 0) out[blockIdx.x] = b;
}

// Stable compaction of the slot list (running scenarios to the front, -1 behind them) and their count -> *count.  One workgroup;
// runs between two chunks of iterations, so that the next chunk launches grids over the running scenarios only and the
// 16-scenario tiles of the leaf kernels stay full.
__global__ __launch_bounds__(1024) void k_compact(int S, int* __restrict__ active, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < S; i0 += 1024) {
        const int i = i0 + tid;
        const int v = i < S ? active[i] : -1;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(v >= 0);
        const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __builtin_popcountll(bal);
        __syncthreads();                  // (also: every read of active[i0 .. i0+1023] precedes the writes below)
        int off = base;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        if (v >= 0) active[off + before] = v;        // off + before <= i: never overtakes an unread entry of a later tile
        __syncthreads();
        if (tid == 0) base += tot;
        __syncthreads();
    }
    for (int i = base + tid; i < S; i += 1024) active[i] = -1;
    if (tid == 0) *count = base;
}

__global__ void k_fill(double* p, size_t count, double v) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i < count) p[i] = v;
}

__global__ void k_set_int(int* p, int count, int v) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < count) p[i] = v;
}

// scenarios that need the repeat pass with partial pivoting: a static pivot block went over the limit (pivflag bit 0) or the
// mismatch became non-finite.  mask[s] = 1, pivflag[s] |= 2 ("repeated"); k_restore_masked then resets their state.
__global__ void k_mark_repeat(int S, const double* __restrict__ err, int* __restrict__ pivflag, int* __restrict__ mask,
                              int* __restrict__ count) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double e = err[s];
    const int m = ((pivflag[s] & 1) || e != e || isinf(e)) ? 1 : 0;
    mask[s] = m;
    if (m) {
        pivflag[s] = (pivflag[s] & 1) | 2;      // bit 1: repeated (bit 0 stays: why)
        atomicAdd(count, 1);
    }
}

// option "keep_previous_state": before a Newton step, the running scenarios' voltages -> the "previous state" copy (the reference
// returns the Jacobian of the LAST iteration, HG:537/560, i.e. the one built at the state before the last update)
__global__ void k_keep_prev(int count, const int* __restrict__ active, const double* __restrict__ Vm, const double* __restrict__ Va,
                            double* __restrict__ Vmp, double* __restrict__ Vap, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;
    if (s < 0) return;
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k < count) {
        Vmp[(size_t)s * count + k] = Vm[(size_t)s * count + k];
        Vap[(size_t)s * count + k] = Va[(size_t)s * count + k];
    }
}

// mask (per scenario, 0 / 1) -> slot list of a repeat pass: slot s runs scenario s or nothing
__global__ void k_mask_to_list(int S, const int* __restrict__ mask, int* __restrict__ list) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) list[s] = mask[s] ? s : -1;
}

__global__ void k_restore_masked(int count, const int* __restrict__ mask, const double* __restrict__ Vm0,
                                 const double* __restrict__ Va0, double* __restrict__ Vm, double* __restrict__ Va,
                                 double* __restrict__ hist, int hist_cap) {
    const int s = blockIdx.y;
    if (!mask[s]) return;
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k < count) {
        Vm[(size_t)s * count + k] = Vm0[(size_t)s * count + k];
        Va[(size_t)s * count + k] = Va0[(size_t)s * count + k];
    }
    if (hist && k < hist_cap) hist[(size_t)s * hist_cap + k] = NAN;
}

__global__ void k_init_voltages(int Hn, int count, double* Vm, double* Va) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= count) return;
    const size_t o = (size_t)blockIdx.y * count + k;
    Vm[o] = (k % Hn) == 0 ? 1.0 : 0.1;     // HG:181-183 (bus-major: harmonic position k % Hn)
    Va[o] = 0.0;
}

// get_THD (HG:563-572) THD_F per bus, max over buses, plus the result flags; one block per scenario.
__global__ void k_stats(int n, int Hn, double thresh, int max_iter, const double* __restrict__ Vm,
                        const double* __restrict__ err, const int* __restrict__ niter, const int* __restrict__ pivflag,
                        hpf_stat* __restrict__ out) {
    const int s = blockIdx.x;
    const double* V = Vm + (size_t)s * n * Hn;
    double best = 0.0;
    bool nan = false;
    for (int b = threadIdx.x; b < n; b += TPB) {
        double hs = 0.0;
        for (int q = 1; q < Hn; ++q) hs = hs + V[(size_t)b * Hn + q] * V[(size_t)b * Hn + q];
        const double t = sqrt(hs) / fabs(V[(size_t)b * Hn]);
        if (t != t) nan = true;
        best = t > best ? t : best;
    }
    unsigned long long bits = nan ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(best);
    bits = wave_max_u64(bits);
    __shared__ unsigned long long red[TPB / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r = red[0];
        for (int w = 1; w < TPB / 64; ++w) r = red[w] > r ? red[w] : r;
        hpf_stat st;
        st.n_iter = niter[s];
        const double e = err[s];
        st.err = e;
        const int pf = pivflag ? pivflag[s] : 0;
        st.flags = (e <= thresh ? 1 : 0) | ((niter[s] >= max_iter && !(e <= thresh)) ? 2 : 0) | ((e != e || isinf(e)) ? 4 : 0) |
                   ((pf & 1) ? 8 : 0) | ((pf & 2) ? 16 : 0) | ((pf & 4) ? 32 : 0);
        st.thd_max = __longlong_as_double((long long)r);
        out[s] = st;
    }
}


// ---- hpf_solve_queue: a sweep of more scenarios than the handle has slots -------------------------------------------------------------
// Slot storage s in [0, S_max) holds scenario slot_scen[s] (global id, -1: free).  Between two chunks of iterations (after k_compact:
// running storages in front of the slot list, their count in *count): every storage that is not running and still holds a scenario has
// met the stop rule -> harvest list; every free storage takes the next pending scenario -> new list, appended to the slot list.
// One workgroup; the storage table sits in LDS and one thread walks it in order (deterministic assignment).
__global__ __launch_bounds__(1024) void k_queue_refill(int S_max, int n_total, int* __restrict__ active, int* __restrict__ count,
                                                        int* __restrict__ slot_scen, int* __restrict__ next, int* __restrict__ hlist,
                                                        int* __restrict__ hg, int* __restrict__ newlist, int* __restrict__ base_out) {
    extern __shared__ int q_lds[];                      // [S_max] busy | [S_max] scenario of the storage
    int* busy = q_lds;
    int* scen = q_lds + S_max;
    const int tid = threadIdx.x, cnt = *count;
    for (int s = tid; s < S_max; s += 1024) {
        busy[s] = 0;
        scen[s] = slot_scen[s];
        hlist[s] = -1;
        newlist[s] = -1;
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += 1024) busy[active[i]] = 1;
    __syncthreads();
    if (tid == 0) {
        int nh = 0, nn = 0, nx = *next;
        for (int s = 0; s < S_max; ++s) {
            if (busy[s]) continue;
            if (scen[s] >= 0) {
                hlist[nh] = s;
                hg[nh] = scen[s];
                ++nh;
                scen[s] = -1;
            }
            if (nx < n_total) {
                scen[s] = nx++;
                newlist[nn] = s;
                active[cnt + nn] = s;
                ++nn;
            }
        }
        *next = nx;
        *base_out = cnt;
        *count = cnt + nn;
    }
    __syncthreads();
    for (int s = tid; s < S_max; s += 1024) slot_scen[s] = scen[s];
}

// result record (k_stats) and, if asked for, the raw voltages in the ABI's stacked order q*n + i of every harvested storage -> the
// per-scenario outputs of the sweep
__global__ void k_queue_harvest(int n, int Hn, double thresh, int max_iter, const int* __restrict__ hlist, const int* __restrict__ hg,
                                const double* __restrict__ Vm, const double* __restrict__ Va, const double* __restrict__ err,
                                const int* __restrict__ niter, const int* __restrict__ pivflag, hpf_stat* __restrict__ qstats,
                                double* __restrict__ qVm, double* __restrict__ qVa) {
    const int s = hlist[blockIdx.x];
    if (s < 0) return;
    const int g = hg[blockIdx.x];
    const double* V = Vm + (size_t)s * n * Hn;
    const double* A = Va + (size_t)s * n * Hn;
    double best = 0.0;
    bool nan = false;
    for (int b = threadIdx.x; b < n; b += TPB) {
        double hs = 0.0;
        for (int q = 1; q < Hn; ++q) hs = hs + V[(size_t)b * Hn + q] * V[(size_t)b * Hn + q];
        const double t = sqrt(hs) / fabs(V[(size_t)b * Hn]);
        if (t != t) nan = true;
        best = t > best ? t : best;
    }
    unsigned long long bits = nan ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(best);
    bits = wave_max_u64(bits);
    __shared__ unsigned long long red[TPB / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r = red[0];
        for (int w = 1; w < TPB / 64; ++w) r = red[w] > r ? red[w] : r;
        hpf_stat st;
        st.n_iter = niter[s];
        const double e = err[s];
        st.err = e;
        const int pf = pivflag[s];
        st.flags = (e <= thresh ? 1 : 0) | ((niter[s] >= max_iter && !(e <= thresh)) ? 2 : 0) | ((e != e || isinf(e)) ? 4 : 0) |
                   ((pf & 1) ? 8 : 0) | ((pf & 4) ? 32 : 0);
        st.thd_max = __longlong_as_double((long long)r);
        qstats[g] = st;
    }
    if (qVm) {
        double* om = qVm + (size_t)g * n * Hn;
        double* oa = qVa + (size_t)g * n * Hn;
        for (int k = threadIdx.x; k < n * Hn; k += TPB) {       // k = q*n + i (coalesced stores), source bus-major
            const int q = k / n, i = k - q * n;
            om[k] = V[(size_t)i * Hn + q];
            oa[k] = A[(size_t)i * Hn + q];
        }
    }
}

// a new scenario moves into every storage of the new list: loads, the reference's start (HG:174-184) with the fundamental entries from
// its power-flow seed, U / E, counters
__global__ void k_queue_init(int n, int Hn, const int* __restrict__ newlist, const int* __restrict__ slot_scen, const double* __restrict__ qP,
                             const double* __restrict__ qQ, const double* __restrict__ seedVm, const double* __restrict__ seedVa,
                             double* __restrict__ P, double* __restrict__ Q, double* __restrict__ Vm, double* __restrict__ Va,
                             cplx* __restrict__ U, cplx* __restrict__ E, int* __restrict__ niter, int* __restrict__ pivflag) {
    const int s = newlist[blockIdx.y];
    if (s < 0) return;
    const int g = slot_scen[s];
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= n * Hn) return;
    const int i = k / Hn, q = k - i * Hn;
    const size_t o = (size_t)s * n * Hn + k;
    const double vm = q == 0 ? seedVm[(size_t)g * n + i] : 0.1, va = q == 0 ? seedVa[(size_t)g * n + i] : 0.0;
    Vm[o] = vm;
    Va[o] = va;
    cplx u, e;
    polar<false>(vm, va, u, e);
    U[o] = u;
    E[o] = e;
    if (q == 0) {
        P[(size_t)s * n + i] = qP[(size_t)g * n + i];
        Q[(size_t)s * n + i] = qQ[(size_t)g * n + i];
    }
    if (k == 0) {
        niter[s] = 0;
        pivflag[s] = 0;
    }
}

// the initial mismatch of the new scenarios against the stop rule (HG:531,536): a scenario that meets it at once leaves the slot list
__global__ __launch_bounds__(64) void k_queue_first(int S_max, double thresh, int max_iter, const int* __restrict__ newlist, const int* __restrict__ base,
                              const unsigned long long* __restrict__ errpart, int pstride, int np, double* __restrict__ err,
                              int* __restrict__ active) {
    const int idx = blockIdx.x;                          // one wavefront per entry of the new list
    if (idx >= S_max) return;
    const int s = newlist[idx];
    if (s < 0) return;
    const double e = __longlong_as_double((long long)slot_err_bits(errpart + (size_t)s * pstride, np));
    if (threadIdx.x != 0) return;
    err[s] = e;
    if (!((e > thresh) && (0 < max_iter))) active[*base + idx] = -1;
}

// fundamental entries (harmonic position 0) of the first S scenarios' voltages -> seed arrays of scenarios g0 .. g0 + S - 1
__global__ void k_queue_keep_seed(int n, int Hn, int g0, const double* __restrict__ Vm, const double* __restrict__ Va,
                                  double* __restrict__ seedVm, double* __restrict__ seedVa) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const size_t o = (size_t)blockIdx.y * n * Hn + (size_t)i * Hn;
    seedVm[(size_t)(g0 + blockIdx.y) * n + i] = Vm[o];
    seedVa[(size_t)(g0 + blockIdx.y) * n + i] = Va[o];
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
namespace {

template <class T>
int dev_alloc(hpf_handle* h, T** p, size_t count) {
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        *p = nullptr;
        return e == hipErrorOutOfMemory ? HPF_E_NOMEM : HPF_E_HIP;
    }
    return HPF_OK;
}

template <class T>
int dev_upload(hpf_handle* h, T** p, const T* src, size_t count) {
    int r = dev_alloc(h, p, count);
    if (r) return r;
    if (count) HIPCHK(hipMemcpy(*p, src, count * sizeof(T), hipMemcpyHostToDevice));
    return HPF_OK;
}

inline dim3 grid2(int count, int S) { return dim3((unsigned)((count + TPB - 1) / TPB), (unsigned)S, 1); }

int ensure_dense(hpf_handle* h, int Nsys) {
    size_t want = (size_t)Nsys * Nsys;                         // (N * N >= 2^31: rocSOLVER's 64-bit entry points, dense_solve)
    if (h->solver == HPF_SOLVER_DENSE && (size_t)h->N * h->N > want) want = (size_t)h->N * h->N;
    if (h->d_J && h->J_elems_per_scen >= want) return HPF_OK;
    if (h->d_J) {
        hipFree(h->d_J);
        hipFree(h->d_ipiv);
        hipFree(h->d_info);
        h->d_J = nullptr;
        h->d_ipiv = nullptr;
        h->d_info = nullptr;
    }
    int r = dev_alloc(h, &h->d_J, want * (size_t)h->S_max);
    if (r) return r;
    const int Nmax = h->N > h->Nf ? h->N : h->Nf;
    if ((r = dev_alloc(h, &h->d_ipiv, 2 * (size_t)Nmax * h->S_max))) return r;       // (room for the int64 pivots of the 64-bit path)
    if ((r = dev_alloc(h, &h->d_info, 2 * (size_t)h->S_max))) return r;
    HIPCHK(hipMemset(h->d_info, 0, sizeof(int) * 2 * h->S_max));
    h->J_elems_per_scen = want;
    return HPF_OK;
}

int resolve_spans(hpf_handle* h) {
    if (h->spans.empty() && h->ts_next == 0) return HPF_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    for (auto& sp : h->spans) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, sp.e0, sp.e1);
        h->t_ms[sp.which] += ms;
        h->t_n[sp.which] += 1;
        hipEventDestroy(sp.e0);
        hipEventDestroy(sp.e1);
    }
    h->spans.clear();
    if (h->d_tstamp && h->ts_next > 0) {              // device-clock durations of the stamped general-factor-kernel launches
        std::vector<unsigned long long> ts(2 * (size_t)h->ts_next);
        HIPCHK(hipMemcpy(ts.data(), h->d_tstamp, sizeof(unsigned long long) * ts.size(), hipMemcpyDeviceToHost));
        int khz = 100000;
        hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device);
        for (int i = 0; i < h->ts_next; ++i)
            if (ts[2 * i + 1] > ts[2 * i] && ts[2 * i] != ~0ull) {
                h->t_ms[T_GJ_DEV] += (double)(ts[2 * i + 1] - ts[2 * i]) / (double)khz;
                h->t_n[T_GJ_DEV] += 1;
            }
        h->ts_next = 0;
        std::vector<unsigned long long> init(2 * (size_t)hpf_handle::TS_CAP);
        for (size_t i = 0; i < init.size(); i += 2) {
            init[i] = ~0ull;
            init[i + 1] = 0ull;
        }
        HIPCHK(hipMemcpy(h->d_tstamp, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice));
    }
    return HPF_OK;
}


// launch context = (stream, first scenario, scenario count) used by the launch helpers
inline void set_ctx(hpf_handle* h, hipStream_t st, int s0, int cnt) {
    h->cur_stream = st;
    h->cur_s0 = s0;
    h->cur_S = cnt;
}
inline void full_ctx(hpf_handle* h) { set_ctx(h, h->stream, 0, h->S); }

inline int groups_for(const hpf_handle* h, int count) {
    if (h->solver != HPF_SOLVER_BLOCK_TREE || h->n_ties > 0) return 1;
    int g = h->n_groups;
    while (g > 1 && count < 32 * g) --g;       // at least 32 scenarios per group (below that the launches of a group no longer fill their levels: tools/groups_sweep.py)
    return g < 1 ? 1 : g;
}
inline int groups_for(const hpf_handle* h) { return groups_for(h, h->S); }

// Run body() once per scenario group (slots [0, count) of the active list / scenarios [0, S) split evenly), each group on its
// own stream between a fork and a join with the main stream.
template <class F>
int for_groups(hpf_handle* h, int count, F body) {
    const int G = groups_for(h, count);
    if (G == 1) {
        set_ctx(h, h->stream, 0, count);
        const int rc1 = body();
        full_ctx(h);
        return rc1;
    }
    HIPCHK(hipEventRecord(h->fork_ev, h->stream));
    int rc = HPF_OK;
    // group boundaries on multiples of 16 slots: the scenario-batched workgroups of the tree kernels take 16 scenarios each, an even
    // split of e.g. 128 into 43 + 43 + 42 would run 3 x 3 tiles with ragged ends instead of 3 + 2 + 3 full ones
    auto bound = [&](int g) { return g >= G ? count : (int)(16 * (((long long)count * g / G + 8) / 16)); };
    for (int g = 0; g < G && rc == HPF_OK; ++g) {
        const int s0 = bound(g), s1 = bound(g + 1);
        HIPCHK(hipStreamWaitEvent(group_stream(h, g), h->fork_ev, 0));
        set_ctx(h, group_stream(h, g), s0, s1 - s0);
        rc = body();
        HIPCHK(hipEventRecord(h->join_ev[g], group_stream(h, g)));
    }
    for (int g = 0; g < G; ++g) HIPCHK(hipStreamWaitEvent(h->stream, h->join_ev[g], 0));
    full_ctx(h);
    return rc;
}

// polar + mismatch (+ optional err conversion) for the current state
template <bool FUND>
int launch_polar(hpf_handle* h) {
    full_ctx(h);
    const int count = FUND ? h->n : h->n * h->Hn;
    hipLaunchKernelGGL((k_polar<FUND>), grid2(count, h->S), dim3(TPB), 0, h->stream, count, h->n * h->Hn, h->Hn, h->d_Vm,
                       h->d_Va, h->d_U, h->d_E);
    HIPCHK(hipGetLastError());
    return HPF_OK;
}

// the Newton step of the multi-wave block-tree sweep works on bus-major images of the mismatch and of the step
static bool bus_images(const hpf_handle* h) { return h->solver == HPF_SOLVER_BLOCK_TREE && h->has_ctree && h->gj_mode == 1; }
static int tree_bst(const hpf_handle* h) { const int b = 2 * h->Hn; return b <= 12 ? 12 : (b <= 28 ? 28 : (b <= 52 ? 52 : (b <= 100 ? 100 : b))); }   // = wave_block_size, or b

// partial maxima a mismatch launch leaves per scenario (one per wavefront of its workgroups): fundamental pf / harmonic mismatch
template <bool FUND>
static int err_parts(const hpf_handle* h) { return (TPB / 64) * (((FUND ? h->n : h->n * h->Hn) + TPB - 1) / TPB); }

// stacked: also write the mismatch in the reference's stacked order (C ABI, dense solver, single-wave / generic tree kernels)
template <bool FUND>
int launch_mismatch(hpf_handle* h, const int* active, bool stacked = true) {
    ScopedTimer t(h, T_MISMATCH);
    const int count = FUND ? h->n : h->n * h->Hn;
    const int N = FUND ? h->Nf : h->N;
    const int Nc = FUND ? h->n - 1 : h->Nc;
    const bool img = !FUND && h->d_fb && bus_images(h);
    if (count > 1) {
        const int nbx = (count + TPB - 1) / TPB;
        // LDS: Y_N^T of one device type + the voltages of the workgroup's tile of buses (harmonic mismatch with coupled Norton data)
        const size_t lds = (!FUND && h->coupled && h->n > h->m)
                               ? sizeof(cplx) * ((size_t)h->Hn * h->Hn + (size_t)(TPB / h->Hn + 2) * h->Hn) : 0;
        if (lds > 64 * 1024) {                      // Hn >= 62 (H_MAX >= 123): beyond the default dynamic-LDS limit of a kernel
            if (lds > 160 * 1024) return HPF_E_ARG;
            // (per launch: the attribute belongs to the device the handle runs on, and the call costs nothing next to the launch)
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mismatch<FUND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        hipLaunchKernelGGL((k_mismatch<FUND>), xcd_grid(nbx, h->cur_S), dim3(TPB), lds, h->cur_stream, h->M, count, N, Nc,
                           active, h->d_U, h->d_P, h->d_Q, (stacked || !img) ? h->d_f : nullptr, h->d_errpart, h->errpart_stride, h->d_I0,
                           img ? h->d_fb : nullptr, tree_bst(h), h->cur_s0, h->cur_S, div_magic(h->Hn));
        HIPCHK(hipGetLastError());
    }
    return HPF_OK;
}

template <bool FUND>
int launch_jacobian_dense(hpf_handle* h, const int* active) {
    ScopedTimer t(h, T_JACOBIAN);
    const int N = FUND ? h->Nf : h->N;
    const int Nc = FUND ? h->n - 1 : h->Nc;
    HIPCHK(hipMemsetAsync(h->d_J, 0, sizeof(double) * h->J_elems_per_scen * (size_t)h->S, h->stream));
    const int total = (FUND ? 1 : h->Hn) * h->nnz;
    hipLaunchKernelGGL((k_jac_dense<FUND>), grid2(total, h->S), dim3(TPB), 0, h->stream, h->M, total, N, Nc,
                       h->J_elems_per_scen, active, h->d_erow, h->d_U, h->d_E, h->d_J);
    HIPCHK(hipGetLastError());
    if (!FUND && h->coupled && h->n > h->m) {
        const int tot = (h->n - h->m) * h->Hn * h->Hn;
        hipLaunchKernelGGL(k_jac_cross_dense, grid2(tot, h->S), dim3(TPB), 0, h->stream, h->M, tot, N, Nc,
                           h->J_elems_per_scen, active, h->d_U, h->d_E, h->d_J);
        HIPCHK(hipGetLastError());
    }
    return HPF_OK;
}

// f <- J^{-1} f for every scenario (rocSOLVER LU with partial pivoting)
int dense_solve(hpf_handle* h, int Nsys) {
    ScopedTimer t(h, T_SOLVE);
    if (Nsys <= 0) return HPF_OK;
    if (ensure_blas(h)) return HPF_E_ROCSOLVER;
    BLASCHK(rocblas_set_stream(h->blas, h->stream));
    if ((long long)Nsys * Nsys >= (1ll << 31)) {
        // beyond 32-bit element offsets (N > 46 340, e.g. the 1 000-bus x 26-harmonic feeder as a dense system: 21.6 GB per scenario):
        // rocSOLVER's 64-bit entry points, one scenario after the other; int64 pivots / info (check_info reads them as such)
        const int Nmax = h->N > h->Nf ? h->N : h->Nf;
        int64_t* ip = reinterpret_cast<int64_t*>(h->d_ipiv);
        int64_t* inf = reinterpret_cast<int64_t*>(h->d_info);
        for (int sc = 0; sc < h->S; ++sc) {
            double* Js = h->d_J + (size_t)sc * h->J_elems_per_scen;
            BLASCHK(rocsolver_dgetrf_64(h->blas, Nsys, Nsys, Js, Nsys, ip + (size_t)sc * Nmax, inf + sc));
            BLASCHK(rocsolver_dgetrs_64(h->blas, rocblas_operation_none, Nsys, 1, Js, Nsys, ip + (size_t)sc * Nmax, h->d_f + (size_t)sc * Nsys, Nsys));
        }
        h->info64 = true;
        return HPF_OK;
    }
    h->info64 = false;
    if (h->S == 1) {
        BLASCHK(rocsolver_dgetrf(h->blas, Nsys, Nsys, h->d_J, Nsys, h->d_ipiv, h->d_info));
        BLASCHK(rocsolver_dgetrs(h->blas, rocblas_operation_none, Nsys, 1, h->d_J, Nsys, h->d_ipiv, h->d_f, Nsys));
    } else {
        const rocblas_stride sA = (rocblas_stride)h->J_elems_per_scen;
        const int Nmax = h->N > h->Nf ? h->N : h->Nf;
        const rocblas_stride sF = (rocblas_stride)Nsys;
        BLASCHK(rocsolver_dgetrf_strided_batched(h->blas, Nsys, Nsys, h->d_J, Nsys, sA, h->d_ipiv, Nmax, h->d_info, h->S));
        BLASCHK(rocsolver_dgetrs_strided_batched(h->blas, rocblas_operation_none, Nsys, 1, h->d_J, Nsys, sA, h->d_ipiv,
                                                 Nmax, h->d_f, Nsys, sF, h->S));
    }
    return HPF_OK;
}

template <bool FUND>
int launch_update(hpf_handle* h, const int* active) {
    ScopedTimer t(h, T_UPDATE);
    const int count = FUND ? h->n : h->n * h->Hn;
    const int N = FUND ? h->Nf : h->N;
    const int Nc = FUND ? h->n - 1 : h->Nc;
    const bool busx = !FUND && bus_images(h);
    const int bw = tree_bst(h);
    hipLaunchKernelGGL((k_update<FUND>), grid2(count, h->cur_S), dim3(TPB), 0, h->cur_stream, h->n, h->Hn, h->c, count,
                       h->n * h->Hn, N, Nc, active, h->d_f, h->d_Vm, h->d_Va, h->d_U, h->d_E,
                       busx ? h->d_x : nullptr, bw, h->cur_s0, div_magic(h->Hn));
    HIPCHK(hipGetLastError());
    return HPF_OK;
}

// One Newton step: Jacobian at the current state, step = J^{-1} f into d_f.
template <bool FUND>
int newton_step(hpf_handle* h, const int* active) {
    int r;
    if (h->solver == HPF_SOLVER_BLOCK_TREE && h->n_ties == 0)
        return FUND ? tree_fund_step(h, active != nullptr) : tree_newton_step(h, active != nullptr);
    if (h->solver == HPF_SOLVER_BLOCK_TREE && !FUND) return tree_newton_step_bordered(h, active != nullptr);
    // (a meshed network on the block-tree path takes its fundamental power flow through the dense LU: Nf = 2n - 1 - c is small)
    const int Nsys = FUND ? h->Nf : h->N;
    if ((r = ensure_dense(h, Nsys))) return r;
    if ((r = launch_jacobian_dense<FUND>(h, active))) return r;
    return dense_solve(h, Nsys);
}

int check_info(hpf_handle* h, const std::vector<int>& was_active) {
    if (!h->d_info) return HPF_OK;
    std::vector<int> info(2 * (size_t)h->S);
    HIPCHK(hipMemcpyAsync(info.data(), h->d_info, sizeof(int) * 2 * h->S, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int s = 0; s < h->S; ++s) {
        const long long v = h->info64 ? reinterpret_cast<const int64_t*>(info.data())[s] : (long long)info[s];
        if (was_active[s] >= 0 && v != 0) {
            h->last_detail = (int)v;
            return HPF_E_SINGULAR;
        }
    }
    return HPF_OK;
}

// state of every scenario after iteration `it` -> the caller's trace arrays (hpf_set_trace), ABI order q*n + i
int trace_record(hpf_handle* h, int it) {
    if (!h->trace_Vm || it >= h->trace_cap) return HPF_OK;
    const size_t count = (size_t)h->n * h->Hn, cnt = (size_t)h->S * count;
    std::vector<double> tm(cnt), ta(cnt);
    HIPCHK(hipMemcpyAsync(tm.data(), h->d_Vm, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ta.data(), h->d_Va, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int sc = 0; sc < h->S; ++sc) {
        double* om = h->trace_Vm + ((size_t)sc * h->trace_cap + it) * count;
        double* oa = h->trace_Va + ((size_t)sc * h->trace_cap + it) * count;
        for (int i = 0; i < h->n; ++i)
            for (int q = 0; q < h->Hn; ++q) {
                om[(size_t)q * h->n + i] = tm[(size_t)sc * count + (size_t)i * h->Hn + q];
                oa[(size_t)q * h->n + i] = ta[(size_t)sc * count + (size_t)i * h->Hn + q];
            }
    }
    return HPF_OK;
}

// pinned double buffer + events through which the host reads the slot counters one chunk late (every object under its own check: a failed
// creation is retried by the next call instead of leaving a null event behind)
int ensure_poll_buffers(hpf_handle* h) {
    for (int i = 0; i < 2; ++i) {
        if (!h->h_act[i]) HIPCHK(hipHostMalloc((void**)&h->h_act[i], sizeof(int) * 4, hipHostMallocDefault));
        if (!h->poll_ev[i]) HIPCHK(hipEventCreateWithFlags(&h->poll_ev[i], hipEventDisableTiming));
    }
    return HPF_OK;
}

// One pass of the NR loop (HG:530-542 / HG:257-265) from the current state over the scenarios selected by `mask` (nullptr: all).
template <bool FUND>
int nr_pass(hpf_handle* h, double thresh, int max_iter, const int* mask) {
    int r;
    const int S = h->S;
    const int hist_off = FUND ? 1 : 0;                // pf records only post-update errors (HG:264)
    if ((r = launch_polar<FUND>(h))) return r;
    if (mask) hipLaunchKernelGGL(k_mask_to_list, dim3((S + 63) / 64), dim3(64), 0, h->stream, S, mask, h->d_active);
    if ((r = launch_mismatch<FUND>(h, mask ? h->d_active : nullptr, false))) return r;
    HIPCHK(hipMemsetAsync(h->d_nactive, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)S), dim3(64), 0, h->stream, S, 1, thresh, max_iter, h->hist_cap,
                       hist_off, h->d_errpart, h->errpart_stride, err_parts<FUND>(h), h->d_err, h->d_niter, h->d_active, h->d_nactive, h->d_hist, 0, mask);
    const bool trace = !FUND && h->trace_Vm != nullptr;
    if (trace && !mask && (r = trace_record(h, 0))) return r;
    // The per-scenario stop rule lives on the device (k_finalize after every iteration, in the scenario group's own pipeline):
    // frozen scenarios are skipped by every kernel, so the host only has to notice when NO scenario is active any more, and
    // noticing late changes nothing in the results.  BLOCK_TREE: the host looks every `chunk` iterations, and it looks at chunk
    // c - 1 while chunk c is already queued (pinned double buffer + events), so the device never drains between chunks.
    const bool pipelined = !FUND && h->solver == HPF_SOLVER_BLOCK_TREE && !trace && h->n_ties == 0;
    const int chunk = pipelined ? (S >= 8 ? 4 : 2) : 1;
    auto enqueue = [&](int todo, int slots) -> int {
        auto body = [&]() -> int {
            int rr;
            for (int j = 0; j < todo; ++j) {
                if (!FUND && h->keep_prev && h->d_Vmp)
                    hipLaunchKernelGGL(k_keep_prev, grid2(h->n * h->Hn, h->cur_S), dim3(TPB), 0, h->cur_stream, h->n * h->Hn,
                                       h->d_active, h->d_Vm, h->d_Va, h->d_Vmp, h->d_Vap, h->cur_s0);
                if ((rr = newton_step<FUND>(h, h->d_active))) return rr;
                if ((rr = launch_update<FUND>(h, h->d_active))) return rr;
                if ((rr = launch_mismatch<FUND>(h, h->d_active, false))) return rr;
                hipLaunchKernelGGL(k_finalize, dim3((unsigned)h->cur_S), dim3(64), 0, h->cur_stream, h->cur_S, 0, thresh,
                                   max_iter, h->hist_cap, hist_off, h->d_errpart, h->errpart_stride, err_parts<FUND>(h), h->d_err, h->d_niter, h->d_active,
                                   h->d_nactive, h->d_hist, h->cur_s0, (const int*)nullptr);
            }
            return HPF_OK;
        };
        if (FUND) {
            full_ctx(h);
            return body();
        }
        return for_groups(h, slots, body);
    };
    if (pipelined) {
        if ((r = ensure_poll_buffers(h))) return r;
        // Between two chunks the slot list is compacted on the device (running scenarios first) and their count goes to the
        // host; the count the host knows is one chunk old, i.e. an upper bound (the count only falls): it sizes the grids and
        // the scenario groups of the next chunk.  Slots behind the true count hold -1 and their workgroups exit at once.
        auto compact_and_post = [&](int buf) -> int {
            hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, S, h->d_active, h->d_nactive);
            HIPCHK(hipMemcpyAsync(h->h_act[buf], h->d_nactive, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipEventRecord(h->poll_ev[buf], h->stream));
            return HPF_OK;
        };
        int it = 0, c = 0, n_ub = S;
        if ((r = compact_and_post(0))) return r;
        for (;;) {
            // With few scenarios left the device is not kept busy anyway: look at the count of the chunk just queued before queueing
            // the next one (no trailing chunk of empty launches).  Otherwise queue chunk c + 1 (iterations it .. it + todo) BEFORE
            // looking at the count chunk c left, so that the device never drains between chunks.
            const bool lagged = n_ub > 8 || S <= 8;      // (a handle of a few scenarios keeps the queue fed: its chunks are short anyway)
            const int ch = lagged ? chunk : 2;
            if (!lagged) {
                HIPCHK(hipEventSynchronize(h->poll_ev[c & 1]));
                const int cnt0 = h->h_act[c & 1][0];
                if (cnt0 == 0 || it >= max_iter) break;
                n_ub = cnt0;
            }
            const int todo = (max_iter - it) < ch ? (max_iter - it) : ch;
            if (todo > 0) {
                if ((r = enqueue(todo, n_ub))) return r;
                if ((r = compact_and_post((c + 1) & 1))) return r;
                it += todo;
            }
            if (!lagged) {
                ++c;
                continue;
            }
            HIPCHK(hipEventSynchronize(h->poll_ev[c & 1]));
            const int cnt = h->h_act[c & 1][0];
            if (cnt == 0 || todo == 0) break;
            n_ub = cnt;
            ++c;
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        return HPF_OK;
    }
    auto count_active = [&](const int* a) {
        int c = 0;
        for (int s = 0; s < S; ++s) c += a[s] >= 0;
        return c;
    };
    std::vector<int> act(S), was(S);
    HIPCHK(hipMemcpyAsync(act.data(), h->d_active, sizeof(int) * S, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    int nactive = count_active(act.data());
    int it = 0;
    while (nactive > 0 && it < max_iter) {
        was = act;                        // (no compaction on this path: slot i runs scenario i)
        h->host_act = act;                // (the bordered step of a meshed network walks the running scenarios on the host)
        if ((r = enqueue(1, S))) return r;
        HIPCHK(hipMemcpyAsync(act.data(), h->d_active, sizeof(int) * S, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        nactive = count_active(act.data());
        if (h->solver == HPF_SOLVER_DENSE)
            if ((r = check_info(h, was))) return r;
        ++it;
        if (trace && (r = trace_record(h, it))) return r;
    }
    return HPF_OK;
}

template <bool FUND>
int nr_loop(hpf_handle* h, double thresh, int max_iter, int* n_iter, double* err, double* err_hist) {
    if (!h->loads_set || !h->state_set || h->S < 1) return HPF_E_STATE;
    if (max_iter < 0) return HPF_E_ARG;
    int r;
    const int S = h->S;
    const int cap = max_iter + 1;
    if (h->hist_cap < cap) {
        if (h->d_hist) hipFree(h->d_hist);
        if ((r = dev_alloc(h, &h->d_hist, (size_t)cap * h->S_max))) return r;
        h->hist_cap = cap;
    }
    {
        const size_t cnt = (size_t)h->hist_cap * S;
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((cnt + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->d_hist, cnt,
                           (double)NAN);
    }
    // BLOCK_TREE inverts the bus blocks with a STATIC pivot order (4x4 blocks on the matrix cores).  Every pivot block is watched
    // (inv4_cofactor_lane): a scenario in which one amplifies by more than piv_limit, or whose mismatch becomes non-finite, is
    // repeated from the state this call was entered with, with partial pivoting over the whole block (pivoted wave Gauss-Jordan
    // on the uncontracted tree).  hpf_stat.flags bit 3 / bit 4 report it.
    const bool can_repeat = !FUND && h->solver == HPF_SOLVER_BLOCK_TREE && h->has_ctree && h->gj_mode == 1 && h->auto_repivot &&
                            h->n_ties == 0;      // (the bordered step of a meshed network runs in the bus-image layout of the static-pivot kernels only)
    const size_t count = (size_t)h->n * h->Hn;
    if (!FUND) HIPCHK(hipMemsetAsync(h->d_pivflag, 0, sizeof(int) * S, h->stream));
    if (can_repeat) {
        if (!h->d_Vm0) {
            if ((r = dev_alloc(h, &h->d_Vm0, (size_t)h->S_max * count))) return r;
            if ((r = dev_alloc(h, &h->d_Va0, (size_t)h->S_max * count))) return r;
            if ((r = dev_alloc(h, &h->d_mask, (size_t)h->S_max))) return r;
        }
        HIPCHK(hipMemcpyAsync(h->d_Vm0, h->d_Vm, sizeof(double) * S * count, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_Va0, h->d_Va, sizeof(double) * S * count, hipMemcpyDeviceToDevice, h->stream));
    }
    if ((r = nr_pass<FUND>(h, thresh, max_iter, nullptr))) return r;
    if (can_repeat) {
        int nrep = 0;
        HIPCHK(hipMemsetAsync(h->d_nactive, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_mark_repeat, dim3((S + 63) / 64), dim3(64), 0, h->stream, S, h->d_err, h->d_pivflag, h->d_mask,
                           h->d_nactive);
        HIPCHK(hipMemcpyAsync(&nrep, h->d_nactive, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (nrep > 0) {
            const unsigned gx = (unsigned)(((count > (size_t)h->hist_cap ? count : (size_t)h->hist_cap) + TPB - 1) / TPB);
            hipLaunchKernelGGL(k_restore_masked, dim3(gx, (unsigned)S), dim3(TPB), 0, h->stream, (int)count, h->d_mask, h->d_Vm0,
                               h->d_Va0, h->d_Vm, h->d_Va, h->d_hist, h->hist_cap);
            h->gj_mode = 0;
            r = nr_pass<FUND>(h, thresh, max_iter, h->d_mask);
            h->gj_mode = 1;
            if (r) return r;
        }
    }
    h->mismatch_valid = false;   // frozen scenarios leave stale rows in d_f: hpf_mismatch before hpf_iterate
    if (!FUND) h->prev_valid = h->keep_prev && h->d_Vmp;
    if (n_iter) HIPCHK(hipMemcpy(n_iter, h->d_niter, sizeof(int) * S, hipMemcpyDeviceToHost));
    if (err) HIPCHK(hipMemcpy(err, h->d_err, sizeof(double) * S, hipMemcpyDeviceToHost));
    if (err_hist) {
        const int cols = FUND ? max_iter : max_iter + 1;
        HIPCHK(hipMemcpy2D(err_hist, sizeof(double) * cols, h->d_hist, sizeof(double) * h->hist_cap,
                           sizeof(double) * cols, S, hipMemcpyDeviceToHost));
    }
    if (!FUND) {
        hipLaunchKernelGGL(k_stats, dim3(S), dim3(TPB), 0, h->stream, h->n, h->Hn, thresh, max_iter, h->d_Vm, h->d_err,
                           h->d_niter, h->d_pivflag, h->d_stats);
        std::vector<int> pf(S);
        HIPCHK(hipMemcpyAsync(pf.data(), h->d_pivflag, sizeof(int) * S, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int s = 0; s < S; ++s)
            if (pf[s] & 4) {               // the pivoted Gauss-Jordan met an exactly zero pivot: singular Jacobian block
                h->last_detail = s;
                return HPF_E_SINGULAR;
            }
    }
    return HPF_OK;
}

// hpf_solve_queue on the fast path (radial BLOCK_TREE, static-pivot kernels): all scenarios' loads and power-flow seeds resident in
// HBM, the harmonic NR runs in chunks of iterations over the slot list; between chunks finished scenarios are harvested and their
// storages refilled (k_queue_*).  The host looks at the counters one chunk late (pinned double buffer + events, as in nr_pass).
int solve_queue_fast(hpf_handle* h, int n_total, const double* P, const double* Q, double thresh_f, int max_iter_f, double thresh,
                     int max_iter, hpf_stat* stats, double* Vm, double* Va) {
    int r = HPF_OK;
    const int n = h->n, Hn = h->Hn, S_max = h->S_max;
    const size_t count = (size_t)n * Hn;
    const bool info = h->sw("HPF_QUEUE_INFO") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double *qP = nullptr, *qQ = nullptr, *sVm = nullptr, *sVa = nullptr, *qVm = nullptr, *qVa = nullptr;
    hpf_stat* qst = nullptr;
    int* qi = nullptr;                                   // slot_scen [S_max] | hlist | hg | newlist | next, base
    auto cleanup = [&](int code) {
        hipStreamSynchronize(h->stream);
        void* ptrs[] = {qP, qQ, sVm, sVa, qVm, qVa, qst, qi};
        for (void* q : ptrs)
            if (q) hipFree(q);
        return code;
    };
    if ((r = dev_alloc(h, &qP, (size_t)n_total * n)) || (r = dev_alloc(h, &qQ, (size_t)n_total * n)) ||
        (r = dev_alloc(h, &sVm, (size_t)n_total * n)) || (r = dev_alloc(h, &sVa, (size_t)n_total * n)) ||
        (r = dev_alloc(h, &qst, (size_t)n_total)) || (r = dev_alloc(h, &qi, (size_t)4 * S_max + 2)))
        return cleanup(r);
    if (Vm && ((r = dev_alloc(h, &qVm, (size_t)n_total * count)) || (r = dev_alloc(h, &qVa, (size_t)n_total * count)))) return cleanup(r);
    int *slot_scen = qi, *hlist = qi + S_max, *hg = qi + 2 * S_max, *newlist = qi + 3 * S_max, *next = qi + 4 * S_max, *base = next + 1;
    if (hipMemcpyAsync(qP, P, sizeof(double) * (size_t)n_total * n, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(qQ, Q, sizeof(double) * (size_t)n_total * n, hipMemcpyHostToDevice, h->stream) != hipSuccess)
        return cleanup(HPF_E_HIP);
    // ---- fundamental power flow (HG:244-275) of every scenario from the reference's start, in waves of S_max; only the fundamental
    //      entries are kept (the harmonic rows of the seed are the constants of HG:181-183)
    for (int g0 = 0; g0 < n_total; g0 += S_max) {
        const int S = n_total - g0 < S_max ? n_total - g0 : S_max;
        h->S = S;
        hipMemcpyAsync(h->d_P, qP + (size_t)g0 * n, sizeof(double) * (size_t)S * n, hipMemcpyDeviceToDevice, h->stream);
        hipMemcpyAsync(h->d_Q, qQ + (size_t)g0 * n, sizeof(double) * (size_t)S * n, hipMemcpyDeviceToDevice, h->stream);
        hipLaunchKernelGGL(k_init_voltages, grid2((int)count, S), dim3(TPB), 0, h->stream, Hn, (int)count, h->d_Vm, h->d_Va);
        h->loads_set = h->state_set = true;
        if ((r = nr_loop<true>(h, thresh_f, max_iter_f, nullptr, nullptr, nullptr))) return cleanup(r);
        hipLaunchKernelGGL(k_queue_keep_seed, grid2(n, S), dim3(TPB), 0, h->stream, n, Hn, g0, h->d_Vm, h->d_Va, sVm, sVa);
    }
    if (info) {
        hipStreamSynchronize(h->stream);
        fprintf(stderr, "hpf queue: %d scenarios, %d slots: uploads + pf of all scenarios %.2f ms\n", n_total, S_max, ms_since(t_0));
    }
    const auto t_1 = std::chrono::steady_clock::now();
    // ---- harmonic NR with refill -------------------------------------------------------------------------------------------------
    const int S_used = n_total < S_max ? n_total : S_max;
    h->S = S_used;
    h->mismatch_valid = false;
    h->prev_valid = false;
    {
        std::vector<int> init((size_t)4 * S_max + 2, -1);
        init[(size_t)4 * S_max] = 0;                    // next
        init[(size_t)4 * S_max + 1] = 0;                // base
        if (hipMemcpyAsync(qi, init.data(), sizeof(int) * init.size(), hipMemcpyHostToDevice, h->stream) != hipSuccess) return cleanup(HPF_E_HIP);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return cleanup(HPF_E_HIP);      // (init is a local)
    }
    if (hipMemsetAsync(h->d_nactive, 0, sizeof(int), h->stream) != hipSuccess) return cleanup(HPF_E_HIP);
    if (hipMemsetAsync(h->d_pivflag, 0, sizeof(int) * S_max, h->stream) != hipSuccess) return cleanup(HPF_E_HIP);
    hipLaunchKernelGGL(k_set_int, dim3((unsigned)((S_max + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->d_active, S_max, -1);   // (empty slot list)
    if ((r = ensure_poll_buffers(h))) return cleanup(r);
    const size_t q_lds = sizeof(int) * 2 * (size_t)S_max;
    // one round between two chunks: compact -> harvest / refill -> initial mismatch of the new scenarios -> counters to the host
    auto round = [&](int buf) -> int {
        full_ctx(h);
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, S_max, h->d_active, h->d_nactive);
        hipLaunchKernelGGL(k_queue_refill, dim3(1), dim3(1024), q_lds, h->stream, S_max, n_total, h->d_active, h->d_nactive, slot_scen, next,
                           hlist, hg, newlist, base);
        hipLaunchKernelGGL(k_queue_harvest, dim3((unsigned)S_max), dim3(TPB), 0, h->stream, n, Hn, thresh, max_iter, hlist, hg, h->d_Vm,
                           h->d_Va, h->d_err, h->d_niter, h->d_pivflag, qst, qVm, qVa);
        hipLaunchKernelGGL(k_queue_init, grid2((int)count, S_max), dim3(TPB), 0, h->stream, n, Hn, newlist, slot_scen, qP, qQ, sVm, sVa,
                           h->d_P, h->d_Q, h->d_Vm, h->d_Va, h->d_U, h->d_E, h->d_niter, h->d_pivflag);
        set_ctx(h, h->stream, 0, S_max);
        int rr = launch_mismatch<false>(h, newlist, false);
        full_ctx(h);
        if (rr) return rr;
        hipLaunchKernelGGL(k_queue_first, dim3((unsigned)S_max), dim3(64), 0, h->stream, S_max, thresh, max_iter, newlist, base,
                           h->d_errpart, h->errpart_stride, err_parts<false>(h), h->d_err, h->d_active);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->h_act[buf], h->d_nactive, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_act[buf] + 1, next, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->poll_ev[buf], h->stream));
        return HPF_OK;
    };
    auto enqueue = [&](int todo, int slots) -> int {
        auto body = [&]() -> int {
            int rr;
            for (int j = 0; j < todo; ++j) {
                if ((rr = newton_step<false>(h, h->d_active))) return rr;
                if ((rr = launch_update<false>(h, h->d_active))) return rr;
                if ((rr = launch_mismatch<false>(h, h->d_active, false))) return rr;
                hipLaunchKernelGGL(k_finalize, dim3((unsigned)h->cur_S), dim3(64), 0, h->cur_stream, h->cur_S, 0, thresh, max_iter, 1, 0,
                                   h->d_errpart, h->errpart_stride, err_parts<false>(h), h->d_err, h->d_niter, h->d_active, h->d_nactive, (double*)nullptr, h->cur_s0, (const int*)nullptr);
            }
            return HPF_OK;
        };
        return for_groups(h, slots, body);
    };
    const int chunk = h->queue_chunk > 0 ? h->queue_chunk : 4;
    int c = 0, n_ub = S_used;
    if ((r = round(0))) return cleanup(r);
    const long long max_rounds = ((long long)(n_total + S_used - 1) / S_used + 2) * ((max_iter + chunk - 1) / chunk + 2) + 8;
    for (long long rd = 0; rd < max_rounds; ++rd) {
        // queue the next chunk BEFORE looking at what the previous round left (the device never drains between chunks)
        if (n_ub > 0 && (r = enqueue(chunk, n_ub))) return cleanup(r);
        if ((r = round((c + 1) & 1))) return cleanup(r);
        if (hipEventSynchronize(h->poll_ev[c & 1]) != hipSuccess) return cleanup(HPF_E_HIP);
        const int cnt = h->h_act[c & 1][0], nxt = h->h_act[c & 1][1];
        ++c;
        if (cnt == 0 && nxt >= n_total) break;           // nothing was running and nothing was pending: every scenario is harvested
        n_ub = nxt < n_total ? S_used : cnt;             // pending scenarios: every storage may be running after the next refill
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) return cleanup(HPF_E_HIP);
    if (info) fprintf(stderr, "hpf queue: harmonic NR with refill %.2f ms (%d rounds of %d iterations)\n", ms_since(t_1), c, chunk);
    if (h->h_act[c & 1][0] != 0 || h->h_act[c & 1][1] < n_total) return cleanup(HPF_E_STATE);     // (round cap hit: cannot happen)
    if (stats && hipMemcpy(stats, qst, sizeof(hpf_stat) * (size_t)n_total, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(HPF_E_HIP);
    if (Vm && (hipMemcpy(Vm, qVm, sizeof(double) * (size_t)n_total * count, hipMemcpyDeviceToHost) != hipSuccess ||
               hipMemcpy(Va, qVa, sizeof(double) * (size_t)n_total * count, hipMemcpyDeviceToHost) != hipSuccess))
        return cleanup(HPF_E_HIP);
    // the handle is left without a defined batch: loads and state have to be set again before the per-batch entry points
    h->loads_set = h->state_set = false;
    h->S = 0;
    return cleanup(HPF_OK);
}

void free_all(hpf_handle* h) {
    void* ptrs[] = {h->d_rowrec, h->d_rowptr, h->d_col, h->d_diag, h->d_erow, h->d_dev, h->d_Y, h->d_YN, h->d_YNt, h->d_IN, h->d_P, h->d_Q,
                    h->d_Vm, h->d_Va, h->d_U, h->d_E, h->d_I0, h->d_f, h->d_errbits, h->d_errpart, h->d_err, h->d_niter, h->d_active,
                    h->d_nactive, h->d_pivflag, h->d_mask, h->d_Vm0, h->d_Va0, h->d_Vmp, h->d_Vap, h->d_swapVm, h->d_swapVa, h->d_tstamp, h->d_hist, h->d_stats, h->d_J, h->d_ipiv, h->d_info, h->d_Z, h->d_w, h->d_x, h->d_linA, h->d_C, h->d_dbg, h->d_H, h->d_chG, h->d_chH, h->d_chD, h->d_chy, h->d_chZ, h->d_lfK, h->d_lfS, h->d_fb, h->d_F, h->d_H2, h->d_jptr, h->d_jcol, h->d_jval};
    for (void* p : ptrs)
        if (p) hipFree(p);
    tree_free(h);
    for (auto& sp : h->spans) {
        hipEventDestroy(sp.e0);
        hipEventDestroy(sp.e1);
    }
    for (int g = 0; g < 8; ++g) {
        if (h->gstream[g]) hipStreamDestroy(h->gstream[g]);
        if (h->join_ev[g]) hipEventDestroy(h->join_ev[g]);
    }
    if (h->fork_ev) hipEventDestroy(h->fork_ev);
    for (int i = 0; i < 2; ++i) {
        if (h->h_act[i]) hipHostFree(h->h_act[i]);
        if (h->poll_ev[i]) hipEventDestroy(h->poll_ev[i]);
    }
    if (h->blas) rocblas_destroy_handle(h->blas);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

int hpf_version(void) { return 100; }

const char* hpf_strerror(int code) {
    switch (code) {
        case HPF_OK: return "success";
        case HPF_E_ARG: return "invalid argument";
        case HPF_E_STATE: return "call order violated (loads/state/mismatch not set)";
        case HPF_E_TOPOLOGY: return "BLOCK_TREE solver: network not connected from bus 0, pattern not symmetric, or too many loop-closing lines (border > 16 384 unknowns)";
        case HPF_E_NOMEM: return "out of device memory";
        case HPF_E_HIP: return "HIP runtime error";
        case HPF_E_ROCSOLVER: return "rocBLAS/rocSOLVER error";
        case HPF_E_SINGULAR: return "singular Jacobian (zero pivot)";
        default: return "unknown error";
    }
}

int hpf_last_error_detail(const hpf_handle* h) { return h ? h->last_detail : 0; }

int hpf_create(hpf_handle** out, const hpf_desc* d) { return hpf_create_opts(out, d, nullptr); }

int hpf_create_opts(hpf_handle** out, const hpf_desc* d, const char* options) {
    if (!out || !d) return HPF_E_ARG;
    *out = nullptr;
    if (d->n < 1 || d->Hn < 1 || d->m < 1 || d->m > d->n || d->c < 1 || d->c > d->m || d->nnz < d->n ||
        d->max_scenarios < 1 || !d->rowptr || !d->col || !d->Yval || !d->dev_of_bus)
        return HPF_E_ARG;
    if (d->m < d->n && (d->n_dev < 1 || !d->Y_N || !d->I_N)) return HPF_E_ARG;
    if (d->solver != HPF_SOLVER_DENSE && d->solver != HPF_SOLVER_BLOCK_TREE) return HPF_E_ARG;
    // (the per-entry kernels split a thread id t <= n Hn + 255 into (bus, harmonic) through a 32-bit reciprocal of Hn: exact for t Hn < 2^32;
    //  and the stacked index n Hn itself has to fit an int with room for the 2 N + 1 sizes derived from it)
    if (((long long)d->n * d->Hn + 256) * d->Hn >= (1ll << 32) || (long long)d->n * d->Hn >= (1ll << 29)) return HPF_E_ARG;
    // host-side validation of the pattern: sorted columns, diagonal present, indices in range
    std::vector<int> diag(d->n, -1), erow(d->nnz);
    if (d->rowptr[0] != 0 || d->rowptr[d->n] != d->nnz) return HPF_E_ARG;
    for (int i = 0; i < d->n; ++i) {
        if (d->rowptr[i + 1] < d->rowptr[i]) return HPF_E_ARG;
        for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e) {
            const int j = d->col[e];
            if (j < 0 || j >= d->n) return HPF_E_ARG;
            if (e > d->rowptr[i] && d->col[e - 1] >= j) return HPF_E_ARG;
            if (j == i) diag[i] = e;
            erow[e] = i;
        }
        if (diag[i] < 0) return HPF_E_ARG;
        const int dv = d->dev_of_bus[i];
        if (i >= d->m ? (dv < 0 || dv >= d->n_dev) : dv != -1) return HPF_E_ARG;
    }
    hpf_handle* h = new (std::nothrow) hpf_handle();
    if (!h) return HPF_E_NOMEM;
    const auto t_create = std::chrono::steady_clock::now();
    int r = HPF_OK;
    auto fail = [&](int code) {
        free_all(h);
        delete h;
        return code;
    };
    h->n = d->n; h->m = d->m; h->c = d->c; h->Hn = d->Hn; h->nnz = d->nnz; h->n_dev = d->n_dev;
    h->coupled = d->coupled ? 1 : 0; h->solver = d->solver; h->device = d->device; h->S_max = d->max_scenarios;
    h->Nc = d->n * d->Hn - 1;
    h->N = 2 * h->Nc - (d->c - 1);
    h->Nf = 2 * d->n - 1 - d->c;
    if (options) h->opts = options;
    if (const char* es = getenv("HPF_ENV_SWITCHES")) h->env_switches = atoi(es) != 0;
    // (dense systems beyond N * N = 2^31 -- 1 000 buses x 26 harmonics is already N = 51 998 -- go through rocSOLVER's 64-bit entry
    //  points, dense_solve; memory, 8 N^2 bytes per scenario, is what bounds them: HPF_E_NOMEM from the allocation)
    if (const char* ab = h->sw("HPF_DEBUG_ABLATE")) h->debug_ablate = atoi(ab);
    if (const char* gm = h->sw("HPF_GJ_MODE")) h->gj_mode = atoi(gm) ? 1 : 0;
    if (const char* lb = h->sw("HPF_LEAFBATCH")) h->leafbatch = atoi(lb) ? 1 : 0;
    if (const char* fl = h->sw("HPF_FUSELEVEL")) h->fuse_levels = atoi(fl) ? 1 : 0;
    if (const char* fb = h->sw("HPF_FUSEBACK")) h->fuse_back = atoi(fb) ? 1 : 0;
    if (const char* fm = h->sw("HPF_FUSEBACK_MAX")) h->fuse_back_max = atoi(fm);
    if (const char* bs = h->sw("HPF_BORDER_SLOTS")) h->border_slot_cap = atoi(bs);
    if (hipSetDevice(d->device) != hipSuccess) return fail(HPF_E_HIP);
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(HPF_E_HIP);
    h->stream = h->own_stream;
    if (const char* gs = h->sw("HPF_GROUPS")) h->n_groups = atoi(gs) < 1 ? 1 : (atoi(gs) > 8 ? 8 : atoi(gs));
    for (int g = 0; g < 8; ++g) {
        if (g > 0 && hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking) != hipSuccess) return fail(HPF_E_HIP);   // (group 0: group_stream)
        if (hipEventCreateWithFlags(&h->join_ev[g], hipEventDisableTiming) != hipSuccess) return fail(HPF_E_HIP);
    }
    if (hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) != hipSuccess) return fail(HPF_E_HIP);
    set_ctx(h, h->stream, 0, 0);
    // (the rocBLAS handle is created on first use, ensure_blas: the radial block-tree path never needs it and its creation costs
    //  more than the whole set-up of a 10 000-bus model)
    h->S_alloc = h->S_max;
    if (d->solver == HPF_SOLVER_BLOCK_TREE) {
        // loop-closing lines of a meshed network: the bordered Newton step needs 1 + m virtual scenario slots behind the real ones
        if ((r = tree_find_ties(h, d))) return fail(r);
        if (h->n_ties > 0) h->gj_mode = 1;           // (HPF_GJ_MODE=0 does not apply: the bordered step needs the static-pivot kernels' layout)
        if (h->n_ties > 0) h->S_alloc = h->S_max + border_slots(h);
    }
    const size_t HnN = (size_t)d->Hn * d->n, S = (size_t)h->S_alloc;
    const size_t ynsz = (size_t)d->n_dev * d->Hn * (d->coupled ? d->Hn : 1);
    if ((r = dev_upload(h, &h->d_rowptr, d->rowptr, (size_t)d->n + 1))) return fail(r);
    if ((r = dev_upload(h, &h->d_col, d->col, (size_t)d->nnz))) return fail(r);
    if ((r = dev_upload(h, &h->d_diag, diag.data(), (size_t)d->n))) return fail(r);
    if ((r = dev_upload(h, &h->d_erow, erow.data(), (size_t)d->nnz))) return fail(r);
    if ((r = dev_upload(h, &h->d_dev, d->dev_of_bus, (size_t)d->n))) return fail(r);
    {   // row records of the mismatch kernel (Model::rowrec)
        std::vector<int> rec((size_t)d->n * 8, 0);
        for (int i = 0; i < d->n; ++i) {
            const int e0 = d->rowptr[i], e1 = d->rowptr[i + 1];
            rec[(size_t)i * 8] = e0;
            rec[(size_t)i * 8 + 1] = e1;
            for (int u = 0; u < 3; ++u) rec[(size_t)i * 8 + 2 + u] = d->col[e0 + u < e1 ? e0 + u : e1 - 1];
        }
        if ((r = dev_upload(h, &h->d_rowrec, rec.data(), rec.size()))) return fail(r);
    }
    {   // device copy of the admittances: entry-major [nnz][Hn] (Model::yi)
        std::vector<cplx> yt((size_t)d->Hn * d->nnz);
        const cplx* src = (const cplx*)d->Yval;
        for (int q = 0; q < d->Hn; ++q)
            for (int e = 0; e < d->nnz; ++e) yt[(size_t)e * d->Hn + q] = src[(size_t)q * d->nnz + e];
        if ((r = dev_upload(h, &h->d_Y, yt.data(), yt.size()))) return fail(r);
    }
    if ((r = dev_upload(h, &h->d_YN, (const cplx*)d->Y_N, ynsz))) return fail(r);
    if (h->coupled) {                                     // transposed copy for the mismatch kernel (Model::YNt)
        std::vector<cplx> yt(ynsz);
        const cplx* src = (const cplx*)d->Y_N;
        for (int dv = 0; dv < d->n_dev; ++dv)
            for (int q = 0; q < d->Hn; ++q)
                for (int p2 = 0; p2 < d->Hn; ++p2)
                    yt[((size_t)dv * d->Hn + p2) * d->Hn + q] = src[((size_t)dv * d->Hn + q) * d->Hn + p2];
        if ((r = dev_upload(h, &h->d_YNt, yt.data(), ynsz))) return fail(r);
    }
    if ((r = dev_upload(h, &h->d_IN, (const cplx*)d->I_N, (size_t)d->n_dev * d->Hn))) return fail(r);
    if ((r = dev_alloc(h, &h->d_P, S * d->n))) return fail(r);
    if ((r = dev_alloc(h, &h->d_Q, S * d->n))) return fail(r);
    if ((r = dev_alloc(h, &h->d_Vm, S * HnN))) return fail(r);
    if ((r = dev_alloc(h, &h->d_Va, S * HnN))) return fail(r);
    if ((r = dev_alloc(h, &h->d_U, S * HnN))) return fail(r);
    if ((r = dev_alloc(h, &h->d_E, S * HnN))) return fail(r);
    if ((r = dev_alloc(h, &h->d_I0, S * (size_t)h->n))) return fail(r);
    if ((r = dev_alloc(h, &h->d_f, S * (size_t)(h->N > h->Nf ? h->N : h->Nf)))) return fail(r);
    if ((r = dev_alloc(h, &h->d_errbits, S))) return fail(r);
    h->errpart_stride = err_parts<false>(h) > err_parts<true>(h) ? err_parts<false>(h) : err_parts<true>(h);
    if ((r = dev_alloc(h, &h->d_errpart, S * (size_t)h->errpart_stride))) return fail(r);
    if (hipMemset(h->d_errpart, 0, sizeof(unsigned long long) * S * (size_t)h->errpart_stride) != hipSuccess) return fail(HPF_E_HIP);
    if ((r = dev_alloc(h, &h->d_err, S))) return fail(r);
    if ((r = dev_alloc(h, &h->d_niter, S))) return fail(r);
    if ((r = dev_alloc(h, &h->d_active, S))) return fail(r);
    if ((r = dev_alloc(h, &h->d_nactive, (size_t)1))) return fail(r);
    if ((r = dev_alloc(h, &h->d_pivflag, S))) return fail(r);
    if (hipMemset(h->d_pivflag, 0, sizeof(int) * S) != hipSuccess) return fail(HPF_E_HIP);
    if ((r = dev_alloc(h, &h->d_stats, S))) return fail(r);
    Model& M = h->M;
    M.n = d->n; M.m = d->m; M.c = d->c; M.Hn = d->Hn; M.nnz = d->nnz; M.n_dev = d->n_dev; M.coupled = h->coupled; M.bus_major = 1;
    M.rowptr = h->d_rowptr; M.col = h->d_col; M.diag = h->d_diag; M.Y = h->d_Y; M.dev = h->d_dev;
    M.YN = h->d_YN; M.IN = h->d_IN; M.YNt = h->d_YNt; M.rowrec = h->d_rowrec;
    if (d->solver == HPF_SOLVER_BLOCK_TREE) {
        const auto t_tree = std::chrono::steady_clock::now();
        if ((r = tree_build(h, d))) return fail(r);
        if ((r = tree_sel_build(h, d))) return fail(r);
        h->setup_ms[1] = h->tree.plan_ms + h->ctree.plan_ms;
        h->setup_ms[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tree).count() - h->setup_ms[1];
        const auto t_al = std::chrono::steady_clock::now();
        if ((r = tree_alloc_scenarios(h))) return fail(r);
        h->setup_ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_al).count();
    }
    h->setup_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
    *out = h;
    return HPF_OK;
}

int hpf_destroy(hpf_handle* h) {
    if (!h) return HPF_E_ARG;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
    return HPF_OK;
}

int hpf_num_unknowns(const hpf_handle* h) { return h ? h->N : HPF_E_ARG; }
int hpf_num_unknowns_fund(const hpf_handle* h) { return h ? h->Nf : HPF_E_ARG; }
int hpf_num_scenarios(const hpf_handle* h) { return h ? ((h->loads_set || h->state_set) ? h->S : 0) : HPF_E_ARG; }
int hpf_max_scenarios(const hpf_handle* h) { return h ? h->S_max : HPF_E_ARG; }
int hpf_tree_levels(const hpf_handle* h) { return h ? active_tree(const_cast<hpf_handle*>(h)).n_levels : HPF_E_ARG; }
int hpf_tree_depths(const hpf_handle* h) { return h ? active_tree(const_cast<hpf_handle*>(h)).n_depths : HPF_E_ARG; }

int hpf_set_loads(hpf_handle* h, int n_scen, const double* P, const double* Q) {
    if (!h || !P || !Q || n_scen < 1 || n_scen > h->S_max) return HPF_E_ARG;
    if (h->state_set && n_scen != h->S) h->state_set = false;
    h->S = n_scen;
    HIPCHK(hipMemcpyAsync(h->d_P, P, sizeof(double) * (size_t)n_scen * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_Q, Q, sizeof(double) * (size_t)n_scen * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->loads_set = true;
    h->mismatch_valid = false;
    h->prev_valid = false;
    return HPF_OK;
}

int hpf_set_state(hpf_handle* h, int n_scen, const double* Vm, const double* Va) {
    if (!h || n_scen < 1 || n_scen > h->S_max || (Vm == nullptr) != (Va == nullptr)) return HPF_E_ARG;
    if (h->loads_set && n_scen != h->S) return HPF_E_ARG;
    h->S = n_scen;
    const int count = h->n * h->Hn;
    if (Vm) {
        // ABI: stacked order q*n + i per scenario (HG:174-184); device: bus-major i*Hn + q
        std::vector<double> tm((size_t)n_scen * count), ta((size_t)n_scen * count);
        for (int sc = 0; sc < n_scen; ++sc)
            for (int q = 0; q < h->Hn; ++q)
                for (int i = 0; i < h->n; ++i) {
                    tm[(size_t)sc * count + (size_t)i * h->Hn + q] = Vm[(size_t)sc * count + (size_t)q * h->n + i];
                    ta[(size_t)sc * count + (size_t)i * h->Hn + q] = Va[(size_t)sc * count + (size_t)q * h->n + i];
                }
        // on the handle's stream: ordered behind whatever hpf_iterate left in flight (the group streams join h->stream)
        HIPCHK(hipMemcpyAsync(h->d_Vm, tm.data(), sizeof(double) * tm.size(), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_Va, ta.data(), sizeof(double) * ta.size(), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    } else {
        hipLaunchKernelGGL(k_init_voltages, grid2(count, n_scen), dim3(TPB), 0, h->stream, h->Hn, count, h->d_Vm, h->d_Va);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->state_set = true;
    h->mismatch_valid = false;
    h->prev_valid = false;
    return HPF_OK;
}

int hpf_get_state(hpf_handle* h, double* Vm, double* Va) {
    if (!h || !Vm || !Va) return HPF_E_ARG;
    if (!h->state_set || h->S < 1) return HPF_E_STATE;
    const size_t count = (size_t)h->n * h->Hn, cnt = (size_t)h->S * count;
    std::vector<double> tm(cnt), ta(cnt);
    HIPCHK(hipMemcpyAsync(tm.data(), h->d_Vm, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ta.data(), h->d_Va, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int sc = 0; sc < h->S; ++sc)                                   // device bus-major -> ABI stacked order
        for (int q = 0; q < h->Hn; ++q)
            for (int i = 0; i < h->n; ++i) {
                Vm[(size_t)sc * count + (size_t)q * h->n + i] = tm[(size_t)sc * count + (size_t)i * h->Hn + q];
                Va[(size_t)sc * count + (size_t)q * h->n + i] = ta[(size_t)sc * count + (size_t)i * h->Hn + q];
            }
    return HPF_OK;
}

static int mismatch_impl(hpf_handle* h, bool fund, double* f, double* err) {
    if (!h) return HPF_E_ARG;
    if (!h->loads_set || !h->state_set || h->S < 1) return HPF_E_STATE;
    int r;
    if ((r = fund ? launch_polar<true>(h) : launch_polar<false>(h))) return r;
    if ((r = fund ? launch_mismatch<true>(h, nullptr) : launch_mismatch<false>(h, nullptr))) return r;
    hipLaunchKernelGGL(k_err_reduce, dim3((unsigned)h->S), dim3(64), 0, h->stream, h->d_errpart, h->errpart_stride,
                       fund ? err_parts<true>(h) : err_parts<false>(h), h->d_errbits);
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N = fund ? h->Nf : h->N;
    if (f) HIPCHK(hipMemcpy(f, h->d_f, sizeof(double) * (size_t)h->S * N, hipMemcpyDeviceToHost));
    if (err) HIPCHK(hipMemcpy(err, h->d_errbits, sizeof(double) * h->S, hipMemcpyDeviceToHost));
    h->mismatch_valid = !fund;
    return HPF_OK;
}

int hpf_mismatch(hpf_handle* h, double* f, double* err) { return mismatch_impl(h, false, f, err); }
int hpf_fund_mismatch(hpf_handle* h, double* f, double* err) { return mismatch_impl(h, true, f, err); }

static int jacobian_impl(hpf_handle* h, bool fund, int scen, double* J) {
    if (!h || !J || scen < 0 || scen >= h->S) return HPF_E_ARG;
    if (!h->loads_set || !h->state_set) return HPF_E_STATE;
    int r;
    const int Nsys = fund ? h->Nf : h->N;
    if ((r = ensure_dense(h, Nsys))) return r;
    if ((r = fund ? launch_polar<true>(h) : launch_polar<false>(h))) return r;
    if ((r = fund ? launch_jacobian_dense<true>(h, nullptr) : launch_jacobian_dense<false>(h, nullptr))) return r;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(J, h->d_J + (size_t)scen * h->J_elems_per_scen, sizeof(double) * (size_t)Nsys * Nsys,
                     hipMemcpyDeviceToHost));
    return HPF_OK;
}

int hpf_jacobian(hpf_handle* h, int scen, double* J) { return jacobian_impl(h, false, scen, J); }

// indptr of the CSR Jacobian (a property of the model): built on the device at the first request, kept for the handle's life
static int jcsr_pattern(hpf_handle* h) {
    if (h->d_jptr) return HPF_OK;
    int r;
    int* ptr = nullptr;
    long long* tot = nullptr;
    if ((r = dev_alloc(h, &ptr, (size_t)h->N + 1))) return r;
    if ((r = dev_alloc(h, &tot, (size_t)1))) {
        hipFree(ptr);
        return r;
    }
    hipLaunchKernelGGL(k_jcsr_count, dim3((unsigned)((h->N + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->M, h->N, h->Nc, ptr);
    hipLaunchKernelGGL(k_jcsr_scan, dim3(1), dim3(1024), 0, h->stream, h->N, ptr, tot);
    long long nnz = 0;
    hipError_t e = hipMemcpyAsync(&nnz, tot, sizeof(long long), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(tot);
    if (e != hipSuccess) {
        hipFree(ptr);
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    if (nnz >= 0x7fffffffll) {           // 32-bit column indices / offsets, like scipy's default index type
        hipFree(ptr);
        return HPF_E_ARG;
    }
    h->d_jptr = ptr;
    h->jnnz = nnz;
    return HPF_OK;
}

static int jacobian_csr_impl(hpf_handle* h, int scen, int32_t* indptr, int32_t* indices, double* data) {
    if (!h || !data || scen < 0 || scen >= h->S) return HPF_E_ARG;
    if (!h->loads_set || !h->state_set) return HPF_E_STATE;
    int r;
    if ((r = jcsr_pattern(h))) return r;
    if (!h->d_jval && (r = dev_alloc(h, &h->d_jval, (size_t)h->jnnz))) return r;      // (each buffer under its own check: a failed second
    if (!h->d_jcol && (r = dev_alloc(h, &h->d_jcol, (size_t)h->jnnz))) return r;      //  allocation is retried by the next call)
    if ((r = launch_polar<false>(h))) return r;
    const size_t so = (size_t)scen * h->n * h->Hn;
    hipLaunchKernelGGL(k_jcsr_fill, dim3((unsigned)((h->N + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->M, h->N, h->Nc, h->d_jptr,
                       h->d_U + so, h->d_E + so, indices ? h->d_jcol : (int*)nullptr, h->d_jval);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    if (indptr) HIPCHK(hipMemcpy(indptr, h->d_jptr, sizeof(int32_t) * ((size_t)h->N + 1), hipMemcpyDeviceToHost));
    if (indices) HIPCHK(hipMemcpy(indices, h->d_jcol, sizeof(int32_t) * (size_t)h->jnnz, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(data, h->d_jval, sizeof(double) * (size_t)h->jnnz, hipMemcpyDeviceToHost));
    return HPF_OK;
}

int hpf_jacobian_nnz(hpf_handle* h, int64_t* nnz) {
    if (!h || !nnz) return HPF_E_ARG;
    const int r = jcsr_pattern(h);
    if (r) return r;
    *nnz = (int64_t)h->jnnz;
    return HPF_OK;
}

int hpf_jacobian_csr(hpf_handle* h, int scen, int32_t* indptr, int32_t* indices, double* data) {
    return jacobian_csr_impl(h, scen, indptr, indices, data);
}

// run fn() with the voltages of option "keep_previous_state" in place of the current ones, then put the current state back
static int with_previous_state(hpf_handle* h, int scen, const std::function<int()>& fn) {
    if (!h || scen < 0 || scen >= h->S) return HPF_E_ARG;
    if (!h->keep_prev || !h->d_Vmp || !h->state_set) return HPF_E_STATE;
    {   // the kept state exists for scenarios that took at least one Newton step in the last hpf_solve
        int ni = 0;
        HIPCHK(hipMemcpy(&ni, h->d_niter + scen, sizeof(int), hipMemcpyDeviceToHost));
        if (ni <= 0 || !h->prev_valid) return HPF_E_STATE;
    }
    const size_t cnt = (size_t)h->S * h->n * h->Hn;
    // (the swap buffers belong to the handle: no allocation / release per call -- a hipFree right after another handle returned tens of GB to
    //  the driver was seen to take 60 ms)
    int r;
    if (!h->d_swapVm && (r = dev_alloc(h, &h->d_swapVm, (size_t)h->S_alloc * h->n * h->Hn))) return r;
    if (!h->d_swapVa && (r = dev_alloc(h, &h->d_swapVa, (size_t)h->S_alloc * h->n * h->Hn))) return r;
    double *tm = h->d_swapVm, *ta = h->d_swapVa;
    hipMemcpyAsync(tm, h->d_Vm, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->stream);
    hipMemcpyAsync(ta, h->d_Va, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->stream);
    hipMemcpyAsync(h->d_Vm, h->d_Vmp, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->stream);
    hipMemcpyAsync(h->d_Va, h->d_Vap, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->stream);
    r = fn();
    hipMemcpyAsync(h->d_Vm, tm, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->stream);
    hipMemcpyAsync(h->d_Va, ta, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->stream);
    if (!r) r = launch_polar<false>(h);
    hipStreamSynchronize(h->stream);
    return r;
}

int hpf_jacobian_last(hpf_handle* h, int scen, double* J) {
    if (!J) return HPF_E_ARG;
    return with_previous_state(h, scen, [&]() { return jacobian_impl(h, false, scen, J); });
}

int hpf_jacobian_csr_last(hpf_handle* h, int scen, int32_t* indptr, int32_t* indices, double* data) {
    if (!data) return HPF_E_ARG;
    return with_previous_state(h, scen, [&]() { return jacobian_csr_impl(h, scen, indptr, indices, data); });
}
int hpf_fund_jacobian(hpf_handle* h, int scen, double* J) { return jacobian_impl(h, true, scen, J); }

int hpf_fund_pf(hpf_handle* h, double thresh, int max_iter, int* n_iter, double* err, double* err_hist) {
    if (!h) return HPF_E_ARG;
    return nr_loop<true>(h, thresh, max_iter, n_iter, err, err_hist);
}

int hpf_solve(hpf_handle* h, double thresh, int max_iter, int* n_iter, double* err, double* err_hist) {
    if (!h) return HPF_E_ARG;
    return nr_loop<false>(h, thresh, max_iter, n_iter, err, err_hist);
}

int hpf_solve_queue(hpf_handle* h, int n_total, const double* P, const double* Q, double thresh_f, int max_iter_f, double thresh,
                    int max_iter, hpf_stat* stats, double* Vm, double* Va) {
    if (!h || !P || !Q || n_total < 1 || max_iter < 0 || max_iter_f < 0 || (Vm == nullptr) != (Va == nullptr)) return HPF_E_ARG;
    const bool fast = h->solver == HPF_SOLVER_BLOCK_TREE && h->n_ties == 0 && h->has_ctree && h->gj_mode == 1 &&
                      bus_images(h) && h->S_max <= 4096 && !h->trace_Vm;      // (k_queue_refill keeps its storage table in LDS: 8 B per slot)
    if (fast) return solve_queue_fast(h, n_total, P, Q, thresh_f, max_iter_f, thresh, max_iter, stats, Vm, Va);
    // every other handle (dense solver, meshed network, pivoted mode): waves of up to S_max scenarios through the per-batch entry points
    const size_t cnt = (size_t)h->n * h->Hn;
    for (int g0 = 0; g0 < n_total; g0 += h->S_max) {
        const int S = n_total - g0 < h->S_max ? n_total - g0 : h->S_max;
        int r;
        if ((r = hpf_set_loads(h, S, P + (size_t)g0 * h->n, Q + (size_t)g0 * h->n))) return r;
        if ((r = hpf_set_state(h, S, nullptr, nullptr))) return r;
        if ((r = hpf_fund_pf(h, thresh_f, max_iter_f, nullptr, nullptr, nullptr))) return r;
        if ((r = hpf_solve(h, thresh, max_iter, nullptr, nullptr, nullptr))) return r;
        if (stats && (r = hpf_get_stats(h, stats + g0))) return r;
        if (Vm && (r = hpf_get_state(h, Vm + (size_t)g0 * cnt, Va + (size_t)g0 * cnt))) return r;
    }
    h->loads_set = h->state_set = false;                 // (as on the queued path: the handle is left without a defined batch)
    h->S = 0;
    return HPF_OK;
}

// `iters` unconditional iterations of every scenario, enqueued group by group on the group streams (fork / join with h->stream)
static int iterate_enqueue(hpf_handle* h, int iters) {
    // iteration-major enqueue order (all groups' step i before any group's step i+1) keeps the group pipelines in phase
    const int G = groups_for(h);
    int r = HPF_OK;
    if (G > 1) {
        HIPCHK(hipEventRecord(h->fork_ev, h->stream));
        for (int g = 0; g < G; ++g) HIPCHK(hipStreamWaitEvent(group_stream(h, g), h->fork_ev, 0));
    }
    auto bound = [&](int g) { return g >= G ? h->S : (int)(16 * (((long long)h->S * g / G + 8) / 16)); };   // (tile-aligned groups, for_groups)
    for (int it = 0; it < iters && r == HPF_OK; ++it) {
        for (int g = 0; g < G && r == HPF_OK; ++g) {
            if (G > 1)
                set_ctx(h, group_stream(h, g), bound(g), bound(g + 1) - bound(g));
            else
                full_ctx(h);
            if ((r = newton_step<false>(h, nullptr))) break;
            if ((r = launch_update<false>(h, nullptr))) break;
            r = launch_mismatch<false>(h, nullptr, false);
        }
    }
    if (G > 1) {
        for (int g = 0; g < G; ++g) {
            HIPCHK(hipEventRecord(h->join_ev[g], group_stream(h, g)));
            HIPCHK(hipStreamWaitEvent(h->stream, h->join_ev[g], 0));
        }
    }
    full_ctx(h);
    return r;
}

int hpf_iterate(hpf_handle* h, int iters) {
    if (!h || iters < 0) return HPF_E_ARG;
    if (!h->loads_set || !h->state_set || !h->mismatch_valid) return HPF_E_STATE;
    // (replaying a captured hipGraph of several iterations was measured again in round 2, with 3 / 4 / 6 / 8 scenario groups at
    //  128 and 1 024 scenarios: 0..-6 % -- the step is not bound by the host's launch rate)
    return iterate_enqueue(h, iters);
}

int hpf_get_stats(hpf_handle* h, hpf_stat* stats) {
    if (!h || !stats) return HPF_E_ARG;
    if (!h->state_set || h->S < 1) return HPF_E_STATE;     // (no batch in the handle, e.g. after hpf_solve_queue)
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(stats, h->d_stats, sizeof(hpf_stat) * h->S, hipMemcpyDeviceToHost));
    return HPF_OK;
}

int hpf_get_stats_dev(hpf_handle* h, void* stats_dev) {
    if (!h || !stats_dev) return HPF_E_ARG;
    if (!h->state_set || h->S < 1) return HPF_E_STATE;
    HIPCHK(hipMemcpyAsync(stats_dev, h->d_stats, sizeof(hpf_stat) * h->S, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HPF_OK;
}

int hpf_debug_stamps(hpf_handle* h, long long* out, int count) {
    if (!h || !out || !h->d_dbg || count > h->S_max * h->n * 8) return HPF_E_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->d_dbg, sizeof(long long) * count, hipMemcpyDeviceToHost));
    return HPF_OK;
}

int hpf_set_option(hpf_handle* h, const char* name, int value) {
    if (!h || !name) return HPF_E_ARG;
    if (!strcmp(name, "scenario_groups")) {         // independent scenario pipelines on separate streams (1..8)
        if (value < 1 || value > 8) return HPF_E_ARG;
        h->n_groups = value;
        return HPF_OK;
    }
    if (!strcmp(name, "block_pivoting")) {          // 1: partial pivoting (wave Gauss-Jordan), 0: static 4x4 blocks on MFMA
        if (value && h->n_ties > 0) return HPF_E_STATE;    // meshed network: the bordered Newton step exists for the static-pivot kernels only
        h->gj_mode = value ? 0 : 1;
        return HPF_OK;
    }
    if (!strcmp(name, "pivot_growth_limit_log10")) { // static pivot order: amplification limit of a 4x4 pivot block, 10^value
        if (value < 0 || value > 300) return HPF_E_ARG;
        h->piv_limit = pow(10.0, (double)value);
        return HPF_OK;
    }
    if (!strcmp(name, "keep_previous_state")) {     // hpf_solve keeps, per scenario, the state its last Newton step started from
        h->keep_prev = value ? 1 : 0;
        if (h->keep_prev && !h->d_Vmp) {
            int rr;
            if ((rr = dev_alloc(h, &h->d_Vmp, (size_t)h->S_alloc * h->n * h->Hn))) return rr;
            if ((rr = dev_alloc(h, &h->d_Vap, (size_t)h->S_alloc * h->n * h->Hn))) return rr;
        }
        return HPF_OK;
    }
    if (!strcmp(name, "border_pivoting")) {         // meshed BLOCK_TREE handles: 1 = the border system always through the pivoted LU
        h->border_pivoting = value ? 1 : 0;
        return HPF_OK;
    }
    if (!strcmp(name, "queue_chunk")) {             // hpf_solve_queue: Newton iterations between two harvest / refill rounds
        if (value < 1 || value > 16) return HPF_E_ARG;
        h->queue_chunk = value;
        return HPF_OK;
    }
    if (!strcmp(name, "auto_repivot")) {            // 0: flagged scenarios are only reported (flags bit 3), not repeated
        h->auto_repivot = value ? 1 : 0;
        return HPF_OK;
    }
    return HPF_E_ARG;
}

int hpf_set_trace(hpf_handle* h, double* Vm_traj, double* Va_traj, int cap) {
    if (!h || (Vm_traj == nullptr) != (Va_traj == nullptr) || (Vm_traj && cap < 1)) return HPF_E_ARG;
    h->trace_Vm = Vm_traj;
    h->trace_Va = Va_traj;
    h->trace_cap = Vm_traj ? cap : 0;
    return HPF_OK;
}

int hpf_set_stream(hpf_handle* h, void* s) {
    if (!h) return HPF_E_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = s ? (hipStream_t)s : h->own_stream;
    full_ctx(h);
    return HPF_OK;
}

int hpf_sync(hpf_handle* h) {
    if (!h) return HPF_E_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    return HPF_OK;
}

int hpf_timing_enable(hpf_handle* h, int on) {
    if (!h) return HPF_E_ARG;
    if (on && !h->d_tstamp && h->solver == HPF_SOLVER_BLOCK_TREE) {
        std::vector<unsigned long long> init(2 * (size_t)hpf_handle::TS_CAP);
        for (size_t i = 0; i < init.size(); i += 2) {
            init[i] = ~0ull;
            init[i + 1] = 0ull;
        }
        int rr = dev_upload(h, &h->d_tstamp, init.data(), init.size());
        if (rr) return rr;
    }
    int r = resolve_spans(h);
    h->timing = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return r;
}

int hpf_timing_get(hpf_handle* h, int which, double* ms, int64_t* launches) {
    if (!h || which < 0 || which >= T_COUNT) return HPF_E_ARG;
    int r = resolve_spans(h);
    if (r) return r;
    if (ms) *ms = h->t_ms[which];
    if (launches) *launches = h->t_n[which];
    return HPF_OK;
}

int hpf_timing_reset(hpf_handle* h) {
    if (!h) return HPF_E_ARG;
    int r = resolve_spans(h);
    for (int i = 0; i < T_COUNT; ++i) {
        h->t_ms[i] = 0;
        h->t_n[i] = 0;
    }
    return r;
}

// update_harmonic_state_vec (HG:476-479) as a standalone call: dx = J^-1 f for a caller-supplied dense column-major J
// (rocSOLVER getrf / getrs, partial pivoting).  No handle: the reference function is stateless too.  N * N >= 2^31 (N > 46 340): rocSOLVER's
// 64-bit entry points, exactly as dense_solve does for a handle; a system whose 8 N^2 bytes do not fit the device's free memory is refused with
// HPF_E_NOMEM before anything is allocated (the sparse form of the same call, hpf_sparse_solve, has no such bound).
int hpf_dense_solve(int device, int N, const double* J_colmajor, const double* f, double* dx) {
    if (N < 1 || !J_colmajor || !f || !dx) return HPF_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return HPF_E_HIP;
    const size_t elems = (size_t)N * N;
    const bool wide = elems >= ((size_t)1 << 31);
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return HPF_E_HIP;
        // J + rocSOLVER's workspace (a few panels) + pivots: leave 5 % + 64 MiB of headroom
        const double need = 8.0 * (double)elems * 1.05 + 64.0 * 1048576.0;
        if (need > (double)free_b) return HPF_E_NOMEM;
    }
    rocblas_handle blas = nullptr;
    double *dJ = nullptr, *df = nullptr;
    int64_t *dip = nullptr, *dinfo = nullptr;            // (sized for the 64-bit path; the 32-bit one uses the front half)
    int64_t info64 = 0;
    int info32 = 0, rc = HPF_OK;
    if (rocblas_create_handle(&blas) != rocblas_status_success) return HPF_E_ROCSOLVER;
    if (hipMalloc((void**)&dJ, sizeof(double) * elems) != hipSuccess || hipMalloc((void**)&df, sizeof(double) * N) != hipSuccess ||
        hipMalloc((void**)&dip, sizeof(int64_t) * N) != hipSuccess || hipMalloc((void**)&dinfo, sizeof(int64_t)) != hipSuccess)
        rc = HPF_E_NOMEM;
    if (!rc && (hipMemcpy(dJ, J_colmajor, sizeof(double) * elems, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(df, f, sizeof(double) * N, hipMemcpyHostToDevice) != hipSuccess))
        rc = HPF_E_HIP;
    if (!rc) {
        rocblas_status st;
        if (wide) {
            st = rocsolver_dgetrf_64(blas, N, N, dJ, N, dip, dinfo);
            if (st == rocblas_status_success) st = rocsolver_dgetrs_64(blas, rocblas_operation_none, N, 1, dJ, N, dip, df, N);
        } else {
            st = rocsolver_dgetrf(blas, N, N, dJ, N, reinterpret_cast<int*>(dip), reinterpret_cast<int*>(dinfo));
            if (st == rocblas_status_success)
                st = rocsolver_dgetrs(blas, rocblas_operation_none, N, 1, dJ, N, reinterpret_cast<int*>(dip), df, N);
        }
        if (st == rocblas_status_memory_error)
            rc = HPF_E_NOMEM;
        else if (st != rocblas_status_success)
            rc = HPF_E_ROCSOLVER;
    }
    if (!rc && (hipMemcpy(wide ? (void*)&info64 : (void*)&info32, dinfo, wide ? sizeof(int64_t) : sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(dx, df, sizeof(double) * N, hipMemcpyDeviceToHost) != hipSuccess))
        rc = HPF_E_HIP;
    if (!rc && (wide ? info64 != 0 : info32 != 0)) rc = HPF_E_SINGULAR;
    hipFree(dJ);
    hipFree(df);
    hipFree(dip);
    hipFree(dinfo);
    rocblas_destroy_handle(blas);
    return rc;
}

double hpf_solve_flops(const hpf_handle* h) {
    if (!h) return 0.0;
    if (h->solver == HPF_SOLVER_BLOCK_TREE) return active_tree(const_cast<hpf_handle*>(h)).flops_factor;
    const double N = h->N;
    return (2.0 / 3.0) * N * N * N + 2.0 * N * N;
}

double hpf_solve_bytes(const hpf_handle* h) {
    if (!h) return 0.0;
    if (h->solver == HPF_SOLVER_BLOCK_TREE) return active_tree(const_cast<hpf_handle*>(h)).bytes_factor;
    const double N = h->N;
    return 8.0 * (2.0 * N * N + 2.0 * N);           // Jacobian written by the assembly, read and written back by getrf
}

int hpf_kernel_model(const hpf_handle* h, int which, double* bytes, double* flops, int* launches) {
    if (!h) return HPF_E_ARG;
    double by = 0.0, fl = 0.0;
    int ln = 0;
    if (h->solver == HPF_SOLVER_BLOCK_TREE) {
        const Tree& T = active_tree(const_cast<hpf_handle*>(h));
        if (which == T_GJ && tree_levels_fused(const_cast<hpf_handle*>(h))) {      // k_level: every dense bus, one launch per level
            by = T.bytes_factor; fl = T.flops_factor; ln = T.n_levels;
        } else if (which == T_GJ) {
            by = T.bytes_gj; fl = T.flops_gj; ln = T.n_gj_launches;
        } else if (which == T_SOLVE) {
            by = T.bytes_factor; fl = T.flops_factor; ln = T.n_levels;
        } else if (which == T_BACK) {
            by = T.bytes_back; fl = T.flops_per_solve - T.flops_factor; ln = T.n_depths;
        } else {
            return HPF_E_ARG;
        }
    } else if (which == T_SOLVE) {
        by = hpf_solve_bytes(h); fl = hpf_solve_flops(h); ln = 1;
    } else {
        return HPF_E_ARG;
    }
    if (bytes) *bytes = by;
    if (flops) *flops = fl;
    if (launches) *launches = ln;
    return HPF_OK;
}

int hpf_tree_census(const hpf_handle* h, int* counts, int n_counts) {
    if (!h || !counts || n_counts < 0) return HPF_E_ARG;
    if (h->solver != HPF_SOLVER_BLOCK_TREE) return HPF_E_STATE;
    const Tree& T = active_tree(const_cast<hpf_handle*>(h));
    const int fused = tree_levels_fused(const_cast<hpf_handle*>(h)) ? 1 : 0;
    for (int i = 0; i < n_counts; ++i) counts[i] = i < 8 ? T.census[i] : (i == 8 ? h->n_ties : (i == 9 ? fused : (i == 10 ? T.n_comp : (i == 11 ? h->border_repivots : (i == 12 ? h->m_border : (i == 13 ? (h->mesh_sel ? h->sel_nP : 0) : (i == 14 ? (h->mesh_sel ? (h->border_gj ? 2 : 1) : 0) : 0)))))));
    return HPF_OK;
}

int hpf_setup_times(const hpf_handle* h, double* ms, int n_ms) {
    if (!h || !ms || n_ms < 0) return HPF_E_ARG;
    for (int i = 0; i < n_ms; ++i) ms[i] = i < 4 ? h->setup_ms[i] : 0.0;
    return HPF_OK;
}

int hpf_scenario_groups(const hpf_handle* h, int live) {
    if (!h || live < 0) return HPF_E_ARG;
    return groups_for(h, live);
}

int hpf_tree_plan(const hpf_desc* d, const char* path) {
    if (!d || !path || d->n < 1 || !d->rowptr || !d->col || !d->Yval || !d->dev_of_bus) return HPF_E_ARG;
    if (d->nnz < d->n + 2 * (d->n - 1) || ((d->nnz - d->n) & 1)) return HPF_E_TOPOLOGY;           // (a connected symmetric pattern has at least the tree's entries)
    if (d->max_scenarios < 1) return HPF_E_ARG;
    remove(path);                                                           // (a stale file of an earlier run must not pass for this one)
    return tree_plan_dump(d, path);
}

double hpf_back_bytes(const hpf_handle* h) {
    if (!h || h->solver != HPF_SOLVER_BLOCK_TREE) return 0.0;
    return active_tree(const_cast<hpf_handle*>(h)).bytes_back;
}

}  // extern "C"
