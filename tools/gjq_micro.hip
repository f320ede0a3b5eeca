// Micro-benchmark of the multi-wave blocked Gauss-Jordan loop (hpf_quad.hpp, phase C) in isolation: cycles per block step with
// pieces of the step ablated at run time.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/gjq_micro.hip -o tools/bin/gjq_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "../harmonic-power-flow_amd/csrc/hpf_gj_mfma.hpp"
using namespace hpf;

constexpr int B = 52, NT = 4;

// abl bits: 1 skip inverse (identity W), 2 skip update MFMAs, 4 skip panel MFMA, 8 skip barrier, 16 skip panel write
template <int ABL>
__global__ __launch_bounds__(256, 4) void k_gj(long long* out, double* sink) {
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double panel[2][NT * 64];
    __shared__ double wl[2][16], pv[2][16];
    d4_t ct[NT];
    const int col = 16 * wv + jj;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * tr + 4 * reg + lg;
            const unsigned hsh = (row * 73u + col * 151u + blockIdx.x * 7u) % 1000u;
            ct[tr][reg] = (row == col) ? 40.0 + 0.01 * hsh : 0.001 * hsh - 0.5;
        }
    if (tid < 32) { wl[tid >> 4][tid & 15] = ((tid & 15) % 5 == 0) ? 1.0 : 0.0; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    bool weak_ = false;
#pragma unroll
    for (int st = 0; st < B / 4; ++st) {
        const int tP = st >> 2, rg = st & 3, j0 = 4 * (st & 3), buf = st & 1;
        const bool incol = jj >= j0 && jj < j0 + 4;
        if (wv == tP) {
            if (ABL & 32) {                     // variant: pivot block, then the panel, then the inverse (its LDS reads behind the writes)
                if (incol) pv[buf][lg * 4 + (jj - j0)] = ct[tP][rg];
                if (incol) {
#pragma unroll
                    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
                }
                const double wji = inv4_cofactor_lane(pv[buf], lane, 1e10, weak_);
                if (lane < 16) wl[buf][(lane & 3) * 4 + (lane >> 2)] = wji;
            } else if (!(ABL & 1)) {
                if (incol) pv[buf][lg * 4 + (jj - j0)] = ct[tP][rg];
                const double wji = inv4_cofactor_lane(pv[buf], lane, 1e10, weak_);
                if (lane < 16) wl[buf][(lane & 3) * 4 + (lane >> 2)] = wji;
            }
            if (!(ABL & 16) && !(ABL & 32) && incol) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
            }
        }
        if (!(ABL & 8)) __syncthreads();
        double aop[NT];
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) {
            const double v = panel[buf][(16 * tr + jj) * 4 + lg];
            aop[tr] = (tr == tP && incol) ? 0.0 : -v;
        }
        const double aw = jj < 4 ? wl[buf][jj * 4 + lg] : 0.0;
        double rfin;
        if (!(ABL & 4)) {
            const d4_t z = {0.0, 0.0, 0.0, 0.0};
            const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ct[tP][rg], z, 0, 0, 0);
            rfin = d[0];
        } else {
            rfin = aw + ct[tP][rg];
        }
        if (wv == tP && incol) {
            rfin = wl[buf][lg * 4 + (jj - j0)];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) ct[tr] = d4_t{0.0, 0.0, 0.0, 0.0};
        }
        if (!(ABL & 2)) {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) ct[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[tr], rfin, ct[tr], 0, 0, 0);
        } else {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) ct[tr][0] += aop[tr] * rfin;
        }
        ct[tP][rg] = rfin;
    }
    double acc = 0.0;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) acc += ct[tr][reg];
    asm volatile("" : "+v"(acc));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345.678) sink[0] = acc;
}

// variant: the owner only writes the pivot columns; after the barrier EVERY wave gathers the 4x4 pivot block from the panel and
// computes its own operand element of W (no scratch hop, no W hop)
template <int ABL>
__global__ __launch_bounds__(256, 4) void k_gj2(long long* out, double* sink) {
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double panel[2][NT * 64];
    d4_t ct[NT];
    const int col = 16 * wv + jj;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * tr + 4 * reg + lg;
            const unsigned hsh = (row * 73u + col * 151u + blockIdx.x * 7u) % 1000u;
            ct[tr][reg] = (row == col) ? 40.0 + 0.01 * hsh : 0.001 * hsh - 0.5;
        }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int st = 0; st < B / 4; ++st) {
        const int tP = st >> 2, rg = st & 3, j0 = 4 * (st & 3), buf = st & 1;
        const bool incol = jj >= j0 && jj < j0 + 4;
        if (wv == tP && incol) {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
        }
        __syncthreads();
        double aop[NT];
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) {
            const double v = panel[buf][(16 * tr + jj) * 4 + lg];
            aop[tr] = (tr == tP && incol) ? 0.0 : -v;
        }
        const double* pvb = panel[buf] + 16 * st;
        const double wa = inv4_cofactor_ij(pvb, lg, jj & 3);            // W[jj&3][lg]
        const double aw = jj < 4 ? wa : 0.0;
        const d4_t z = {0.0, 0.0, 0.0, 0.0};
        const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ct[tP][rg], z, 0, 0, 0);
        double rfin = d[0];
        if (wv == tP) {
            const double wsel = inv4_cofactor_ij(pvb, jj & 3, lg);       // W[lg][jj&3]
            if (incol) {
                rfin = wsel;
#pragma unroll
                for (int tr = 0; tr < NT; ++tr) ct[tr] = d4_t{0.0, 0.0, 0.0, 0.0};
            }
        }
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) ct[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[tr], rfin, ct[tr], 0, 0, 0);
        ct[tP][rg] = rfin;
    }
    double acc = 0.0;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) acc += ct[tr][reg];
    asm volatile("" : "+v"(acc));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (tid < 64) sink[1 + blockIdx.x * 64 + tid] = acc;
}

template <int ABL>
__global__ __launch_bounds__(256, 4) void k_gj_ref(double* sink) {    // reference result of the scratch-hop form, same matrices
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double panel[2][NT * 64];
    __shared__ double wl[2][16], pv[2][16];
    d4_t ct[NT];
    const int col = 16 * wv + jj;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * tr + 4 * reg + lg;
            const unsigned hsh = (row * 73u + col * 151u + blockIdx.x * 7u) % 1000u;
            ct[tr][reg] = (row == col) ? 40.0 + 0.01 * hsh : 0.001 * hsh - 0.5;
        }
    __syncthreads();
    bool weak_ = false;
#pragma unroll
    for (int st = 0; st < B / 4; ++st) {
        const int tP = st >> 2, rg = st & 3, j0 = 4 * (st & 3), buf = st & 1;
        const bool incol = jj >= j0 && jj < j0 + 4;
        if (wv == tP) {
            if (incol) pv[buf][lg * 4 + (jj - j0)] = ct[tP][rg];
            const double wji = inv4_cofactor_lane(pv[buf], lane, 1e10, weak_);
            if (lane < 16) wl[buf][(lane & 3) * 4 + (lane >> 2)] = wji;
            if (incol) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
            }
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
    double acc = 0.0;
    for (int tr = 0; tr < NT; ++tr)
        for (int reg = 0; reg < 4; ++reg) acc += ct[tr][reg];
    if (tid < 64) sink[1 + blockIdx.x * 64 + tid] = acc;
}

template <int ABL>
void run2(const char* name, int nblk, long long* d_out, double* d_sink) {
    std::vector<long long> h(nblk);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_gj2<ABL>, dim3(nblk), dim3(256), 0, 0, d_out, d_sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_out, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-44s blocks %5d  median %7lld cycles = %6.0f per step\n", name, nblk, h[nblk / 2], h[nblk / 2] / 13.0);
}

template <int ABL>
void run(const char* name, int nblk, long long* d_out, double* d_sink) {
    std::vector<long long> h(nblk);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_gj<ABL>, dim3(nblk), dim3(256), 0, 0, d_out, d_sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_out, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-44s blocks %5d  median %7lld cycles = %6.0f per step\n", name, nblk, h[nblk / 2], h[nblk / 2] / 13.0);
}

int main() {
    const int maxblk = 256 * 16;
    long long* d_out;
    double* d_sink;
    hipMalloc(&d_out, sizeof(long long) * maxblk);
    hipMalloc(&d_sink, 8 * (1 + 64 * maxblk));
    {   // results of the two forms on the same matrices (wave-0 column sums of the inverse)
        std::vector<double> a(1 + 64 * 4), b2(1 + 64 * 4);
        hipLaunchKernelGGL(k_gj_ref<0>, dim3(4), dim3(256), 0, 0, d_sink);
        hipMemcpy(a.data(), d_sink, sizeof(double) * a.size(), hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_gj2<0>, dim3(4), dim3(256), 0, 0, d_out, d_sink);
        hipMemcpy(b2.data(), d_sink, sizeof(double) * b2.size(), hipMemcpyDeviceToHost);
        double md = 0.0, mv = 0.0;
        for (size_t i = 1; i < a.size(); ++i) {
            md = std::max(md, std::abs(a[i] - b2[i]));
            mv = std::max(mv, std::abs(a[i]));
        }
        printf("  all-waves-invert vs scratch-hop form: max |diff| %.3e (max |value| %.3e)\n", md, mv);
    }
    for (int nblk : {1, 256, 1024, 4096}) {
        if (nblk > maxblk) continue;
        run<0>("full", nblk, d_out, d_sink);
        run<32>("panel write before the inverse", nblk, d_out, d_sink);
        run2<0>("every wave inverts from the panel", nblk, d_out, d_sink);
        run<1>("no inverse", nblk, d_out, d_sink);
        run<2>("no update MFMAs", nblk, d_out, d_sink);
        run<4>("no panel MFMA", nblk, d_out, d_sink);
        run<6>("no MFMAs at all", nblk, d_out, d_sink);
        run<8>("no barrier", nblk, d_out, d_sink);
        run<16>("no panel write", nblk, d_out, d_sink);
        run<1 | 16>("no inverse, no panel write", nblk, d_out, d_sink);
        run<1 | 2 | 4 | 16>("only barrier + LDS reads", nblk, d_out, d_sink);
    }
    return 0;
}
