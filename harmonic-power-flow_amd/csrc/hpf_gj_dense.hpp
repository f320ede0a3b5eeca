// In-place Gauss-Jordan inversion of one b x b block held by a 256-thread workgroup (16 x 16 threads), partial (row) pivoting over the
// whole block.  Thread (tr, tc) = (tid >> 4, tid & 15) owns the R x R sub-grid {tr + 16 a} x {tc + 16 c} in registers; the rows / columns of
// each step are broadcast through LDS (2 barriers per step).  Shared by the generic block-tree factor kernel (k_tree_factor, hpf_block.hip:
// blocks assembled in registers) and by the CSR-in block solve (k_csr_factor, hpf_csr_solve.hip: blocks given by the caller).
//
// On return  Rm[i * ldr + cc] = ((P D)^-1)[i][cc]  (LDS, b x b, leading dimension ldr = b | 1),  pfwd / pinv = the row permutation and its
// inverse:  D^-1 y = Rm * (y[pfwd[.]]),  D^-1 A = Rm[:, pinv[.]] * A;  piv[j] = pivot row of step j.  A zero pivot leaves inf / NaN behind and is
// reported through *zero_pivot (LDS flag, may be nullptr).
#pragma once
#include <hip/hip_runtime.h>

namespace hpf {

// LDS layout of the workgroup: doubles Rm[b * ldr] | colbuf[b] | rowr[b] | rowj[b] | ybuf[b], then ints piv[b] | pinv[b] | pfwd[b]
struct GjDenseLds {
    double *Rm, *colbuf, *rowr, *rowj, *ybuf;
    int *piv, *pinv, *pfwd;
    int ldr;
    __device__ __forceinline__ GjDenseLds(double* lds, int b) {
        ldr = b | 1;
        Rm = lds;
        colbuf = Rm + (size_t)b * ldr;
        rowr = colbuf + b;
        rowj = rowr + b;
        ybuf = rowj + b;
        piv = (int*)(ybuf + b);
        pinv = piv + b;
        pfwd = pinv + b;
    }
};
inline size_t gj_dense_lds_bytes(int b) {
    const int ldr = b | 1;
    return sizeof(double) * ((size_t)b * ldr + 4 * (size_t)b) + sizeof(int) * 3 * (size_t)b;
}

template <int R>
__device__ __forceinline__ void gj_dense_invert(double (&a)[R][R], int b, const GjDenseLds& L, int* zero_pivot) {
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    double *colbuf = L.colbuf, *rowr = L.rowr, *rowj = L.rowj;
    for (int j = 0; j < b; ++j) {
        // column j -> LDS
        if (tc == (j & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai) {
                const int i = tr + 16 * ai;
                if (i < b) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci)
                        if (ci == (j >> 4)) colbuf[i] = a[ai][ci];
                }
            }
        }
        __syncthreads();
        // pivot row: largest |.| among rows >= j (lowest index wins ties), computed redundantly by every thread
        int r = j;
        double best = fabs(colbuf[j]);
        for (int i = j + 1; i < b; ++i) {
            const double v = fabs(colbuf[i]);
            if (v > best) {
                best = v;
                r = i;
            }
        }
        // rows r and j -> LDS
        if (tr == (r & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai)
                if (ai == (r >> 4)) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) {
                        const int cc = tc + 16 * ci;
                        if (cc < b) rowr[cc] = a[ai][ci];
                    }
                }
        }
        if (r != j && tr == (j & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai)
                if (ai == (j >> 4)) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) {
                        const int cc = tc + 16 * ci;
                        if (cc < b) rowj[cc] = a[ai][ci];
                    }
                }
        }
        if (tid == 0) {
            L.piv[j] = r;
            if (zero_pivot && !(best > 0.0)) *zero_pivot = j + 1;      // (zero or NaN column: singular block)
        }
        __syncthreads();
        const double inv = 1.0 / colbuf[r];
        // scaled pivot row values of my columns (column j itself becomes 1/pivot)
        double prow[R];
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            prow[ci] = cc < b ? (cc == j ? inv : rowr[cc] * inv) : 0.0;
        }
#pragma unroll
        for (int ai = 0; ai < R; ++ai) {
            const int i = tr + 16 * ai;
            if (i >= b) continue;
            if (i == j) {
#pragma unroll
                for (int ci = 0; ci < R; ++ci) a[ai][ci] = prow[ci];
            } else {
                // after the swap row r holds the old row j
                const double fct = (i == r) ? colbuf[j] : colbuf[i];
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    if (cc >= b) continue;
                    const double base = (i == r) ? rowj[cc] : a[ai][ci];
                    a[ai][ci] = (cc == j) ? -fct * inv : fma(-fct, prow[ci], base);
                }
            }
        }
        // the LDS buffers are rewritten only after the next barrier pair, but colbuf is rewritten first:
        __syncthreads();
    }
    // R = (P D)^{-1} -> LDS; permutation pi with (P x)[k] = x[pi[k]]
#pragma unroll
    for (int ai = 0; ai < R; ++ai) {
        const int i = tr + 16 * ai;
        if (i >= b) continue;
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            if (cc < b) L.Rm[(size_t)i * L.ldr + cc] = a[ai][ci];
        }
    }
    if (tid == 0) {
        for (int i = 0; i < b; ++i) L.pfwd[i] = i;
        for (int j = 0; j < b; ++j) {
            const int r = L.piv[j];
            const int tmp = L.pfwd[j];
            L.pfwd[j] = L.pfwd[r];
            L.pfwd[r] = tmp;
        }
        for (int i = 0; i < b; ++i) L.pinv[L.pfwd[i]] = i;
    }
    __syncthreads();
}

// The same inversion WITHOUT pivoting (static pivot order 0, 1, ..., b - 1): on return a = D^-1 in the threads' registers, no permutation.
// Two barriers per step, no pivot search.  cb / rr: two LDS arrays of b doubles.  A zero (or NaN) pivot sets *zero_pivot = step + 1.
template <int R>
__device__ __forceinline__ void gj_dense_invert_npvt(double (&a)[R][R], int b, double* cb, double* rr, int* zero_pivot) {
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    for (int j = 0; j < b; ++j) {
        if (tc == (j & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai) {
                const int i = tr + 16 * ai;
                if (i < b) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci)
                        if (ci == (j >> 4)) cb[i] = a[ai][ci];
                }
            }
        }
        if (tr == (j & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai)
                if (ai == (j >> 4)) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) {
                        const int cc = tc + 16 * ci;
                        if (cc < b) rr[cc] = a[ai][ci];
                    }
                }
        }
        __syncthreads();
        const double piv = cb[j];
        if (tid == 0 && zero_pivot && !(fabs(piv) > 0.0)) *zero_pivot = j + 1;
        const double inv = 1.0 / piv;
        double prow[R];
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            prow[ci] = cc < b ? (cc == j ? inv : rr[cc] * inv) : 0.0;
        }
#pragma unroll
        for (int ai = 0; ai < R; ++ai) {
            const int i = tr + 16 * ai;
            if (i >= b) continue;
            if (i == j) {
#pragma unroll
                for (int ci = 0; ci < R; ++ci) a[ai][ci] = prow[ci];
            } else {
                const double fct = cb[i];
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    if (cc >= b) continue;
                    a[ai][ci] = (cc == j) ? -fct * inv : fma(-fct, prow[ci], a[ai][ci]);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace hpf
