// Micro-benchmark of the wave-level Gauss-Jordan inversion (csrc/hpf_gj.hpp): single-wave latency in shader cycles
// (s_memtime) and throughput at full occupancy.  Diagnostic tool, not part of libhpf.so.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I harmonic-power-flow_amd/csrc tools/gj_micro.hip -o gpurun_out/gj_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <cmath>
#include "hpf_gj.hpp"
#include "hpf_gj_mfma.hpp"

#ifndef GJB
#define GJB 52
#endif
#ifndef GJVAR
#define GJVAR 0
#endif
using namespace hpf;

template <int B>
__global__ __launch_bounds__(64) void k_gj(const double* __restrict__ Ain, double* __restrict__ Aout, long long* cyc, int nsteps) {
    const int lane = threadIdx.x;
    const size_t blk = blockIdx.x;
    __shared__ __attribute__((aligned(16))) double rowbuf[B];
    __shared__ double ybc;
    __shared__ int rj[B];
    double a[B];
    const double* A = Ain + blk * B * B;
#pragma unroll
    for (int c = 0; c < B; ++c) a[c] = lane < B ? A[(size_t)c * B + lane] : 0.0;   // column-major input: coalesced
    double y = lane < B ? 1.0 + lane : 0.0;
    int myj = 0;
    double mypiv = 1.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
#if GJVAR == 1
    gauss_jordan_wave_rl<B>(a, y, lane, nsteps, rj, myj, mypiv);
#else
    gauss_jordan_wave<B>(a, y, lane, nsteps, rowbuf, &ybc, rj, myj, mypiv);
#endif
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (lane < B && nsteps == B) {   // rj[] is only defined after a complete elimination
        const double invp = 1.0 / mypiv;
        double* O = Aout + blk * B * B;
#pragma unroll
        for (int j = 0; j < B; ++j) O[(size_t)rj[j] * B + myj] = a[j] * invp;     // AinvT[c][i]
    }
    if (nsteps != B && lane < B) Aout[blk * B * B + lane] = a[0] + y;   // keep the loads alive
    if (lane == 0 && cyc) cyc[blk] = t1 - t0;
}


// MFMA variant: matrix in accumulator layout, static 4x4 pivot blocks
template <int B>
__global__ __launch_bounds__(64) void k_gj_mfma(const double* __restrict__ Ain, double* __restrict__ Aout, long long* cyc, int nsteps) {
    constexpr int NT = (B + 15) / 16;
    const int lane = threadIdx.x, lg = lane >> 4, jj = lane & 15;
    const size_t blk = blockIdx.x;
    __shared__ double panel[NT * 64 + 16];
    d4_t c[NT][NT];
    const double* A = Ain + blk * B * B;
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int tc = 0; tc < NT; ++tc)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 16 * tr + lg + 4 * reg, cc = 16 * tc + jj;
                c[tr][tc][reg] = (i < B && cc < B) ? A[(size_t)cc * B + i] : (i == cc ? 1.0 : 0.0);
            }
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (nsteps == B) gauss_jordan_mfma<NT, (B + 3) / 4>(c, panel);      // (step count is a compile-time parameter now)
    const long long t1 = __builtin_amdgcn_s_memtime();
    double* O = Aout + blk * B * B;
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int tc = 0; tc < NT; ++tc)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 16 * tr + lg + 4 * reg, cc = 16 * tc + jj;
                if (i < B && cc < B) O[(size_t)cc * B + i] = c[tr][tc][reg];       // AinvT[c][i]
            }
    if (lane == 0 && cyc) cyc[blk] = t1 - t0;
}

#if GJVAR == 2
#define KGJ k_gj_mfma
#else
#define KGJ k_gj
#endif

int main(int argc, char** argv) {
    constexpr int B = GJB;
    const int nblk = argc > 1 ? std::max(1, atoi(argv[1])) : 131072;
    std::vector<double> A((size_t)nblk * B * B);
    srand(1);
    for (size_t i = 0; i < A.size(); ++i) A[i] = (rand() / (double)RAND_MAX) - 0.5;
    for (int blk = 0; blk < nblk; ++blk)
        for (int d = 0; d < B; ++d) A[(size_t)blk * B * B + (size_t)d * B + d] += 0.5 * B;   // diagonally dominant (static-pivot variants need it; the feeder blocks are block-dominant)
    double *dA, *dO;
    long long* dC;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dO, A.size() * 8);
    hipMalloc(&dC, (size_t)nblk * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // correctness on block 0
    hipLaunchKernelGGL((KGJ<B>), dim3(1), dim3(64), 0, 0, dA, dO, dC, B);
    std::vector<double> O((size_t)B * B);
    hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < B; ++j) {
            double s = 0;   // (A * Ainv)[i][j], A col-major A[c*B+r], O = AinvT[c][i] = Ainv[i][c]
            for (int k = 0; k < B; ++k) s += A[(size_t)k * B + i] * O[(size_t)j * B + k];
            maxerr = std::max(maxerr, std::fabs(s - (i == j)));
        }
    printf("B=%d variant %d  |A*Ainv - I|max = %.3e\n", B, GJVAR, maxerr);
    if (const char* dbg = getenv("GJ_NBS")) {   // debug: run only GJ_NBS block steps on matrix 0 and dump input + state
        hipLaunchKernelGGL((KGJ<B>), dim3(1), dim3(64), 0, 0, dA, dO, dC, -atoi(dbg));
        hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen("gpurun_out/gj_dump.bin", "wb");
        if (f) {
            fwrite(A.data(), 8, (size_t)B * B, f);
            fwrite(O.data(), 8, (size_t)B * B, f);
            fclose(f);
        }
        return 0;
    }
    for (int grid : {1, 256, 1024, 2048, 4096, nblk}) {
        if (grid > nblk) continue;   // never launch more blocks than matrices were allocated
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((KGJ<B>), dim3(grid), dim3(64), 0, 0, dA, dO, dC, B);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> C(grid);
        hipMemcpy(C.data(), dC, (size_t)grid * 8, hipMemcpyDeviceToHost);
        std::sort(C.begin(), C.end());
        printf("grid %7d: %.3f ms  -> %.1f ns/block; in-kernel GJ cycles median %lld (%.0f per step), min %lld max %lld\n", grid, ms,
               1e6 * ms / grid, C[grid / 2], (double)C[grid / 2] / B, C[0], C[grid - 1]);
    }
    // zero steps: load/store overhead only
    hipEventRecord(e0);
    hipLaunchKernelGGL((KGJ<B>), dim3(nblk), dim3(64), 0, 0, dA, dO, (long long*)nullptr, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %7d, 0 steps (load+store only): %.3f ms\n", nblk, ms);
    return 0;
}
