// Batches of b x b block products, one product per workgroup -- the arithmetic of the selected inversions of the meshed paths: hpf_sparse_solve
// (hpf_csr_solve.hip, blocks given by the caller) and the factor-once bordered Newton step (hpf_block.hip, tree_sel_run: blocks read from the
// inverse slots of the block-tree sweep).
#pragma once
#include <hip/hip_runtime.h>

#include "hpf_gj_dense.hpp"

namespace hpf {
namespace {                  // (kernels with internal linkage: every translation unit that includes this launches its own copy)

// One b x b block product per workgroup:  C = add + alpha A B  (A == nullptr: C = add + alpha B; add == nullptr: zero).  Thread (tr, tc) owns the
// R x R sub-grid {tr + 16 a} x {tc + 16 c} like the factor kernel; the operands come from L2 (every block is read by many jobs of a launch).
// A launch may carry a second grid dimension y (scenarios of a batch): every operand then moves by its own stride (in doubles) per y.
struct BlkJob {
    const double *A, *B, *add;
    double* C;
    double alpha;
    long long sA, sB, sAdd, sC;
};
// K-chunk of the LDS staging: 2 b kc doubles within 64 KB (no launch attribute), a multiple of 4
inline int blk_jobs_kc(int b) {
    int kc = ((8192 - b) / (2 * b)) & ~3;                 // b (kc + 1) + kc b <= 8 192 doubles (kc a multiple of 4: a chunk's zero padding stays inside)
    const int bpad = (b + 3) & ~3;
    return kc > bpad ? bpad : (kc < 4 ? 4 : kc);
}
inline size_t blk_jobs_lds(int b) { return sizeof(double) * ((size_t)b * (blk_jobs_kc(b) + 1) + (size_t)blk_jobs_kc(b) * b); }

template <int R>
__global__ __launch_bounds__(256) void k_blk_jobs(int b, int kc, const BlkJob* __restrict__ jobs) {
    extern __shared__ double blk_lds[];
    BlkJob jb = jobs[blockIdx.x];
    {
        const long long y = blockIdx.y;
        if (jb.A) jb.A += y * jb.sA;
        jb.B += y * jb.sB;
        if (jb.add) jb.add += y * jb.sAdd;
        jb.C += y * jb.sC;
    }
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    double o[R][R];
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) o[ai][ci] = 0.0;
    if (jb.A) {
        // A[:, k0 : k0 + kc] and B[k0 : k0 + kc, :] go through LDS (16-byte coalesced loads, all issued before the first store: one memory round
        // trip per chunk; b is even, so every row starts 16-byte aligned); the products then read As by row -- one address per 16 lanes -- and Bs
        // by column, four k at a time (a chunk is padded with zeros to a multiple of four)
        const int lda = kc + 1;
        double* As = blk_lds;
        double* Bs = blk_lds + (size_t)b * lda;
        const double2* A2 = reinterpret_cast<const double2*>(jb.A);
        const double2* B2 = reinterpret_cast<const double2*>(jb.B);
        const int bh = b >> 1;
        for (int k0 = 0; k0 < b; k0 += kc) {
            const int kn = b - k0 < kc ? b - k0 : kc, kn4 = (kn + 3) & ~3, knh = kn >> 1;      // (k0, kn even)
            if (k0) __syncthreads();
            constexpr int LQ = 6;                            // loads in flight per thread and operand
            for (int base = 0; base < b * knh; base += 256 * LQ) {
                double2 va[LQ], vb[LQ];
#pragma unroll
                for (int u = 0; u < LQ; ++u) {
                    const int idx = base + u * 256 + tid;
                    va[u] = double2{0.0, 0.0};
                    vb[u] = double2{0.0, 0.0};
                    if (idx < b * knh) {
                        const int i = idx / knh, kh = idx - i * knh;
                        va[u] = A2[(size_t)i * bh + (k0 >> 1) + kh];
                        vb[u] = B2[(size_t)(k0 >> 1) * b + idx];           // rows k0 .. k0 + kn - 1 of B are contiguous: kn * b / 2 pairs
                    }
                }
#pragma unroll
                for (int u = 0; u < LQ; ++u) {
                    const int idx = base + u * 256 + tid;
                    if (idx < b * knh) {
                        const int i = idx / knh, kh = idx - i * knh;
                        As[i * lda + 2 * kh] = va[u].x;
                        As[i * lda + 2 * kh + 1] = va[u].y;
                        Bs[2 * idx] = vb[u].x;
                        Bs[2 * idx + 1] = vb[u].y;
                    }
                }
            }
            if (kn4 > kn) {                                  // zero padding of the chunk's last k group
                for (int idx = tid; idx < b * (kn4 - kn); idx += 256) {
                    const int i = idx / (kn4 - kn), kk = kn + idx - i * (kn4 - kn);
                    As[i * lda + kk] = 0.0;
                    Bs[kk * b + i] = 0.0;
                }
            }
            __syncthreads();
            for (int kk = 0; kk < kn4; kk += 4) {
                double g[R][4], z[4][R];
#pragma unroll
                for (int ai = 0; ai < R; ++ai) {
                    const int i = tr + 16 * ai;
#pragma unroll
                    for (int u = 0; u < 4; ++u) g[ai][u] = i < b ? As[i * lda + kk + u] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) {
                        const int cc = tc + 16 * ci;
                        z[u][ci] = cc < b ? Bs[(kk + u) * b + cc] : 0.0;
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ai = 0; ai < R; ++ai)
#pragma unroll
                        for (int ci = 0; ci < R; ++ci) o[ai][ci] = fma(g[ai][u], z[u][ci], o[ai][ci]);
            }
        }
    } else {
#pragma unroll
        for (int ai = 0; ai < R; ++ai)
#pragma unroll
            for (int ci = 0; ci < R; ++ci) {
                const int i = tr + 16 * ai, cc = tc + 16 * ci;
                if (i < b && cc < b) o[ai][ci] = jb.B[(size_t)i * b + cc];
            }
    }
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int i = tr + 16 * ai, cc = tc + 16 * ci;
            if (i < b && cc < b) {
                const double base = jb.add ? jb.add[(size_t)i * b + cc] : 0.0;
                jb.C[(size_t)i * b + cc] = fma(jb.alpha, o[ai][ci], base);
            }
        }
}

// one block inverted in place, no pivoting (gj_dense_invert_npvt); *flag <- nonzero when a pivot is zero
template <int R>
__global__ __launch_bounds__(256) void k_blk_invert(int b, double* __restrict__ blk, long long stride, int* __restrict__ flag) {
    __shared__ double cb[128], rr[128];
    __shared__ int zp;
    blk += (long long)blockIdx.x * stride;
    flag += blockIdx.x;
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    if (tid == 0) zp = 0;
    double a[R][R];
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int i = tr + 16 * ai, cc = tc + 16 * ci;
            a[ai][ci] = (i < b && cc < b) ? blk[(size_t)i * b + cc] : 0.0;
        }
    __syncthreads();
    gj_dense_invert_npvt<R>(a, b, cb, rr, &zp);
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int i = tr + 16 * ai, cc = tc + 16 * ci;
            if (i < b && cc < b) blk[(size_t)i * b + cc] = a[ai][ci];
        }
    if (tid == 0 && zp) atomicMax(flag, zp);
}
template <int R>
inline void launch_invert_R(int b, double* blk, long long stride, int ny, int* flag, hipStream_t st) {
    hipLaunchKernelGGL((k_blk_invert<R>), dim3((unsigned)ny), dim3(256), 0, st, b, blk, stride, flag);
}
inline void launch_invert(int R, int b, double* blk, long long stride, int ny, int* flag, hipStream_t st) {
    switch (R) {
        case 1: launch_invert_R<1>(b, blk, stride, ny, flag, st); break;
        case 2: launch_invert_R<2>(b, blk, stride, ny, flag, st); break;
        case 3: launch_invert_R<3>(b, blk, stride, ny, flag, st); break;
        case 4: launch_invert_R<4>(b, blk, stride, ny, flag, st); break;
        case 5: launch_invert_R<5>(b, blk, stride, ny, flag, st); break;
        case 6: launch_invert_R<6>(b, blk, stride, ny, flag, st); break;
        case 7: launch_invert_R<7>(b, blk, stride, ny, flag, st); break;
        default: launch_invert_R<8>(b, blk, stride, ny, flag, st); break;
    }
}

template <int R>
inline void launch_jobs_R(int b, int count, const BlkJob* jobs, hipStream_t st, int ny) {
    if (count > 0 && ny > 0) hipLaunchKernelGGL((k_blk_jobs<R>), dim3((unsigned)count, (unsigned)ny), dim3(256), blk_jobs_lds(b), st, b, blk_jobs_kc(b), jobs);
}
inline void launch_jobs(int R, int b, int count, const BlkJob* jobs, hipStream_t st, int ny = 1) {
    switch (R) {
        case 1: launch_jobs_R<1>(b, count, jobs, st, ny); break;
        case 2: launch_jobs_R<2>(b, count, jobs, st, ny); break;
        case 3: launch_jobs_R<3>(b, count, jobs, st, ny); break;
        case 4: launch_jobs_R<4>(b, count, jobs, st, ny); break;
        case 5: launch_jobs_R<5>(b, count, jobs, st, ny); break;
        case 6: launch_jobs_R<6>(b, count, jobs, st, ny); break;
        case 7: launch_jobs_R<7>(b, count, jobs, st, ny); break;
        default: launch_jobs_R<8>(b, count, jobs, st, ny); break;
    }
}

}  // namespace
}  // namespace hpf
