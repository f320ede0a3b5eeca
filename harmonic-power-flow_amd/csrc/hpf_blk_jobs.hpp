// Batches of b x b block products, one product per workgroup -- the arithmetic of the selected inversions of the meshed paths: hpf_sparse_solve
// (hpf_csr_solve.hip, blocks given by the caller) and the factor-once bordered Newton step (hpf_block.hip, tree_sel_run: blocks read from the
// inverse slots of the block-tree sweep).
#pragma once
#include <hip/hip_runtime.h>

namespace hpf {
namespace {                  // (kernels with internal linkage: every translation unit that includes this launches its own copy)

// One b x b block product per workgroup:  C = add + alpha A B  (A == nullptr: C = add + alpha B; add == nullptr: zero).  Thread (tr, tc) owns the
// R x R sub-grid {tr + 16 a} x {tc + 16 c} like the factor kernel; the operands come from L2 (every block is read by many jobs of a launch).
struct BlkJob {
    const double *A, *B, *add;
    double* C;
    double alpha;
};
// K-chunk of the LDS staging: 2 b kc doubles within 64 KB (no launch attribute), a multiple of 4
inline int blk_jobs_kc(int b) {
    int kc = ((8192 - b) / (2 * b)) & ~3;                 // b (kc + 1) + kc b <= 8 192 doubles
    const int bpad = (b + 3) & ~3;
    return kc > bpad ? bpad : (kc < 4 ? 4 : kc);
}
inline size_t blk_jobs_lds(int b) { return sizeof(double) * ((size_t)b * (blk_jobs_kc(b) + 1) + (size_t)blk_jobs_kc(b) * b); }

template <int R>
__global__ __launch_bounds__(256) void k_blk_jobs(int b, int kc, const BlkJob* __restrict__ jobs) {
    extern __shared__ double blk_lds[];
    const BlkJob jb = jobs[blockIdx.x];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    double o[R][R];
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) o[ai][ci] = 0.0;
    if (jb.A) {
        // A[:, k0 : k0 + kc] and B[k0 : k0 + kc, :] go through LDS (coalesced loads issued at once: one memory round trip per chunk; the products
        // then read As by row -- one address per 16 lanes -- and Bs by column)
        const int lda = kc + 1;
        double* As = blk_lds;
        double* Bs = blk_lds + (size_t)b * lda;
        for (int k0 = 0; k0 < b; k0 += kc) {
            const int kn = b - k0 < kc ? b - k0 : kc;
            if (k0) __syncthreads();
            for (int idx = tid; idx < b * kn; idx += 256) {
                const int i = idx / kn, kk = idx - i * kn;
                As[i * lda + kk] = jb.A[(size_t)i * b + k0 + kk];
            }
            for (int idx = tid; idx < kn * b; idx += 256) Bs[idx] = jb.B[(size_t)k0 * b + idx];
            __syncthreads();
            for (int kk = 0; kk < kn; ++kk) {
                double g[R], z[R];
#pragma unroll
                for (int ai = 0; ai < R; ++ai) {
                    const int i = tr + 16 * ai;
                    g[ai] = i < b ? As[i * lda + kk] : 0.0;
                }
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    z[ci] = cc < b ? Bs[kk * b + cc] : 0.0;
                }
#pragma unroll
                for (int ai = 0; ai < R; ++ai)
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) o[ai][ci] = fma(g[ai], z[ci], o[ai][ci]);
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

template <int R>
inline void launch_jobs_R(int b, int count, const BlkJob* jobs, hipStream_t st) {
    if (count > 0) hipLaunchKernelGGL((k_blk_jobs<R>), dim3((unsigned)count), dim3(256), blk_jobs_lds(b), st, b, blk_jobs_kc(b), jobs);
}
inline void launch_jobs(int R, int b, int count, const BlkJob* jobs, hipStream_t st) {
    switch (R) {
        case 1: launch_jobs_R<1>(b, count, jobs, st); break;
        case 2: launch_jobs_R<2>(b, count, jobs, st); break;
        case 3: launch_jobs_R<3>(b, count, jobs, st); break;
        case 4: launch_jobs_R<4>(b, count, jobs, st); break;
        case 5: launch_jobs_R<5>(b, count, jobs, st); break;
        case 6: launch_jobs_R<6>(b, count, jobs, st); break;
        case 7: launch_jobs_R<7>(b, count, jobs, st); break;
        default: launch_jobs_R<8>(b, count, jobs, st); break;
    }
}

}  // namespace
}  // namespace hpf
