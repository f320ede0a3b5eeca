// Experiment (round 4): BLOCKED Gauss-Jordan of the 52 x 52 bus block with 16 x 16 pivot blocks against the kernel's 13 steps of 4 x 4 pivot
// blocks (hpf_quad.hpp phase C), in isolation.  The owner wave of a diagonal tile inverts it alone (four 4 x 4 sub-steps inside the tile, no
// workgroup barrier), then ONE barrier per 16 columns: the other waves apply W16 = D^-1 to their pivot rows and a rank-16 update (K = 16 MFMA
// chains) to their other tiles.  3 blocked steps + the last 4-column step = 4 barriers instead of 13.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/experiments/gj16_micro.hip -o tools/bin/gj16_micro
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../harmonic-power-flow_amd/csrc/hpf_gj_mfma.hpp"
using namespace hpf;

constexpr int B = 52, NT = 4;

__device__ __forceinline__ void init_tiles(d4_t (&ct)[NT], int wv, int lg, int jj) {
    const int col = 16 * wv + jj;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * tr + 4 * reg + lg;
            const unsigned hsh = (row * 73u + col * 151u + blockIdx.x * 7u) % 1000u;
            double v = (row == col) ? 40.0 + 0.01 * hsh : 0.001 * hsh - 0.5;
            if (row >= B || col > B) v = (row == col) ? 1.0 : 0.0;          // identity padding (column B = right-hand side)
            if (row >= B && col == B) v = 0.0;
            ct[tr][reg] = v;
        }
}

// one step of the kernel's loop (4 x 4 pivot block st), all tiles
__device__ __forceinline__ void step4(d4_t (&ct)[NT], int st, int wv, int lane, int lg, int jj, double (*panel)[NT * 64], double (*wl)[16],
                                      double (*pv)[16]) {
    const int tP = st >> 2, rg = st & 3, j0 = 4 * (st & 3), buf = st & 1;
    const bool incol = jj >= j0 && jj < j0 + 4;
    bool weak_ = false;
    if (wv == tP) {
        if (incol) pv[buf][lg * 4 + (jj - j0)] = ct[tP][rg];
        const double wji = inv4_cofactor_lane(pv[buf], lane, 1e10, weak_);
        if (incol) {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
        }
        if (lane < 16) wl[buf][(lane & 3) * 4 + (lane >> 2)] = wji;
    }
    __syncthreads();
    double aop[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) {
        const double v = panel[buf][(16 * tr + jj) * 4 + lg];
        aop[tr] = (tr == tP && incol) ? 0.0 : -v;
    }
    const double aw = jj < 4 ? wl[buf][jj * 4 + lg] : 0.0;
    const d4_t z = {0.0, 0.0, 0.0, 0.0};
    const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ct[tP][rg], z, 0, 0, 0);
    double rfin = d[0];
    if (wv == tP && incol) {
        rfin = wl[buf][lg * 4 + (jj - j0)];
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) ct[tr] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) ct[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[tr], rfin, ct[tr], 0, 0, 0);
    ct[tP][rg] = rfin;
}

template <int NST>
__global__ __launch_bounds__(256, 4) void k_ref(long long* out, double* res) {
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double panel[2][NT * 64];
    __shared__ double wl[2][16], pv[2][16];
    __shared__ double occ_pad[4096];                       // 32 KB: four workgroups per CU, like the product kernel
    d4_t ct[NT];
    init_tiles(ct, wv, lg, jj);
    if (tid == 999) occ_pad[blockIdx.x & 4095] = 1.0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int st = 0; st < NST; ++st) step4(ct, st, wv, lane, lg, jj, panel, wl, pv);
    double acc = 0.0;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) acc += ct[tr][reg];
    asm volatile("" : "+v"(acc));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (res && blockIdx.x < 4)
        for (int tr = 0; tr < NT; ++tr)
            for (int reg = 0; reg < 4; ++reg) res[((size_t)blockIdx.x * 64 + 16 * tr + 4 * reg + lg) * 64 + 16 * wv + jj] = ct[tr][reg];
}

#define WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
constexpr int WS = 17;      // row stride (doubles) of the 16 x 16 LDS images read as MFMA A operands: odd -> the 16 rows hit different banks

template <int NBLK, int LAST = B / 4>
__global__ __launch_bounds__(256, 4) void k_blk16(long long* out, double* res) {
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double panel[2][NT * 64];
    __shared__ double wl[2][16], pv[2][16];
    __shared__ double pcol[16 * 4];                       // owner-private: pivot columns of the diagonal tile during its own inversion
    __shared__ double img[2][NT][16 * WS];                // [buffer][tile tr]: tr == T: W16, else the owner's column panel A_iP (row-major 16 x 16)
    d4_t ct[NT];
    init_tiles(ct, wv, lg, jj);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int T = 0; T < NBLK; ++T) {
        const int buf = T & 1;
        if (wv == T) {
            // ---- phase 1: the owner inverts its diagonal tile alone (Gauss-Jordan inside the tile, four 4 x 4 pivot sub-steps) ----------
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int j0 = 4 * s;
                const bool incol = jj >= j0 && jj < j0 + 4;
                bool weak_ = false;
                if (incol) pv[0][lg * 4 + (jj - j0)] = ct[T][s];
                WAVE_LDS_FENCE();                         // (one wave, no workgroup barrier: other LANES' LDS writes must be ordered before this lane's reads)
                const double wji = inv4_cofactor_lane(pv[0], lane, 1e10, weak_);
                if (incol) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) pcol[(lg + 4 * reg) * 4 + (jj - j0)] = ct[T][reg];
                }
                if (lane < 16) wl[0][(lane & 3) * 4 + (lane >> 2)] = wji;
                WAVE_LDS_FENCE();
                const double v = pcol[jj * 4 + lg];
                const double aopl = incol ? 0.0 : -v;
                const double aw = jj < 4 ? wl[0][jj * 4 + lg] : 0.0;
                const d4_t z = {0.0, 0.0, 0.0, 0.0};
                const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ct[T][s], z, 0, 0, 0);
                double rfin = d[0];
                if (incol) {
                    rfin = wl[0][lg * 4 + (jj - j0)];
                    ct[T] = d4_t{0.0, 0.0, 0.0, 0.0};
                }
                ct[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(aopl, rfin, ct[T], 0, 0, 0);
                ct[T][s] = rfin;
                WAVE_LDS_FENCE();                         // (the next sub-step overwrites pv / pcol / wl)
            }
            // ---- phase 2: W16 and the column panel -> LDS (row-major 16 x 16 images) ----------------------------------------------------
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) img[buf][tr][(4 * reg + lg) * WS + jj] = ct[tr][reg];
        }
        __syncthreads();
        // ---- phase 3: every wave: pivot rows R <- W16 R, other tiles A_i <- A_i - A_iP R (K = 16 chains); owner: A_iP <- -A_iP W16 --------
        d4_t Rn;                                            // -R_new (B operand of the rank-16 update), row groups k4
        if (wv == T) {
            Rn = -ct[T];
        } else {
            d4_t r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) r = __builtin_amdgcn_mfma_f64_16x16x4f64(img[buf][T][jj * WS + 4 * k4 + lg], ct[T][k4], r, 0, 0, 0);
            ct[T] = r;
            Rn = -r;
        }
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) {
            if (tr == T) continue;
            d4_t a = (wv == T) ? d4_t{0.0, 0.0, 0.0, 0.0} : ct[tr];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) a = __builtin_amdgcn_mfma_f64_16x16x4f64(img[buf][tr][jj * WS + 4 * k4 + lg], Rn[k4], a, 0, 0, 0);
            ct[tr] = a;
        }
    }
    // the last 4 pivot columns (48..51) as one ordinary step
#pragma unroll
    for (int st = 4 * NBLK; st < LAST; ++st) step4(ct, st, wv, lane, lg, jj, panel, wl, pv);
    double acc = 0.0;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) acc += ct[tr][reg];
    asm volatile("" : "+v"(acc));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (res && blockIdx.x < 4)
        for (int tr = 0; tr < NT; ++tr)
            for (int reg = 0; reg < 4; ++reg) res[((size_t)blockIdx.x * 64 + 16 * tr + 4 * reg + lg) * 64 + 16 * wv + jj] = ct[tr][reg];
}

template <class K>
void run(const char* name, K kern, int nblk, long long* d_out) {
    std::vector<long long> h(nblk);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), 0, 0, d_out, (double*)nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h.data(), d_out, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-28s blocks %5d  median %7lld cycles per block (memtime ticks)   kernel %8.1f us\n", name, nblk, h[nblk / 2], 1e3 * ms);
}

int main() {
    const int maxblk = 256 * 32;
    long long* d_out;
    double *d_a, *d_b;
    hipMalloc(&d_out, sizeof(long long) * maxblk);
    hipMalloc(&d_a, sizeof(double) * 4 * 64 * 64);
    hipMalloc(&d_b, sizeof(double) * 4 * 64 * 64);
    hipLaunchKernelGGL(k_ref<4>, dim3(4), dim3(256), 0, 0, d_out, d_a);
    hipLaunchKernelGGL((k_blk16<1, 4>), dim3(4), dim3(256), 0, 0, d_out, d_b);
    {
        std::vector<double> a(64 * 64), b(64 * 64);
        hipMemcpy(a.data(), d_a, sizeof(double) * a.size(), hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), d_b, sizeof(double) * b.size(), hipMemcpyDeviceToHost);
        for (int tr = 0; tr < 4; ++tr)
            for (int tc = 0; tc < 4; ++tc) {
                double md = 0;
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) md = std::max(md, std::fabs(a[(16 * tr + i) * 64 + 16 * tc + j] - b[(16 * tr + i) * 64 + 16 * tc + j]));
                printf("after the first 16 columns: tile (%d, %d) max |diff| %.3e\n", tr, tc, md);
            }
    }
    hipLaunchKernelGGL(k_ref<B / 4>, dim3(4), dim3(256), 0, 0, d_out, d_a);
    std::vector<double> a(4 * 64 * 64), b(4 * 64 * 64);
    hipMemcpy(a.data(), d_a, sizeof(double) * a.size(), hipMemcpyDeviceToHost);
    for (int nb = 0; nb <= 3; ++nb) {
        if (nb == 0) hipLaunchKernelGGL(k_blk16<0>, dim3(4), dim3(256), 0, 0, d_out, d_b);
        if (nb == 1) hipLaunchKernelGGL(k_blk16<1>, dim3(4), dim3(256), 0, 0, d_out, d_b);
        if (nb == 2) hipLaunchKernelGGL(k_blk16<2>, dim3(4), dim3(256), 0, 0, d_out, d_b);
        if (nb == 3) hipLaunchKernelGGL(k_blk16<3>, dim3(4), dim3(256), 0, 0, d_out, d_b);
        hipMemcpy(b.data(), d_b, sizeof(double) * b.size(), hipMemcpyDeviceToHost);
        double md = 0, mv = 0;
        int wr = -1, wc = -1;
        for (size_t i = 0; i < 64 * 64; ++i) {
            if (std::fabs(a[i] - b[i]) > md) { md = std::fabs(a[i] - b[i]); wr = (int)(i / 64); wc = (int)(i % 64); }
            mv = std::max(mv, std::fabs(a[i]));
        }
        printf("%d blocked steps (16 x 16 pivot blocks) vs the kernel's 4 x 4 steps: max |diff| %.3e at (%d, %d)  (max |entry| %.3e)\n", nb, md, wr, wc, mv);
    }
    for (int nblk : {1, 256, 1024, 4096, 8192}) {
        run("4x4 steps (13 barriers)", k_ref<B / 4>, nblk, d_out);
        run("16x16 blocks (4 barriers)", k_blk16<3>, nblk, d_out);
    }
    return 0;
}
