// In-place inversion of one dense b x b block (row-major in memory) on the FP64 matrix cores: the blocked Gauss-Jordan of the block-tree factor
// kernel (hpf_quad.hpp, section C: static 4 x 4 pivot blocks, tile columns spread over NT wavefronts, columns in blocks of 16 inverted by the
// owning wave alone) as a kernel of its own, for the border systems of the meshed paths (border_block_gj, hpf_block.hip).  b <= B, B in
// {12, 28, 52, 100}; rows / columns b .. 16 NT - 1 are identity padding (their steps are no-ops).  A pivot block whose inverse amplifies by more
// than `limit` (inv4_cofactor_lane) raises *flag: the caller then repeats the solve with a pivoted LU.
#pragma once
#include <hip/hip_runtime.h>

#include "hpf_gj_mfma.hpp"

namespace hpf {
namespace {

#define HPF_BI_WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int B>
__global__ __launch_bounds__(64 * ((B + 16) / 16)) void k_blk_invert_mfma(int b, double* __restrict__ blk, long long stride, double limit, int* __restrict__ flag) {
    constexpr int NT = (B + 16) / 16;
    constexpr int NB16 = B / 16, WS = 17, NBUF = NB16 > 1 ? 2 : 1;
    __shared__ double img_s[(NB16 > 0 ? NBUF : 1) * NT * 16 * WS];
    __shared__ double panel[2][NT * 64];
    __shared__ double wl[2][16];
    __shared__ double pv[2][16];
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = 16 * wv + jj;
    blk += (long long)blockIdx.x * stride;               // one block per workgroup: a batch of systems (scenarios) per launch
    flag += blockIdx.x;
    d4_t ct[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * tr + 4 * reg + lg;
            ct[tr][reg] = (row < b && col < b) ? blk[(size_t)row * b + col] : (row == col ? 1.0 : 0.0);
        }
    bool any_weak = false;
    if constexpr (NB16 > 0) {
        double* const img = img_s;
        double* const pcol = &panel[1][0];
        __syncthreads();
#pragma unroll
        for (int T = 0; T < NB16; ++T) {
            double* const im = img + (size_t)(NBUF > 1 ? (T & 1) : 0) * NT * 16 * WS;
            if (wv == T) {
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    const int j0 = 4 * ss;
                    const bool incol = jj >= j0 && jj < j0 + 4;
                    if (incol) pv[0][lg * 4 + (jj - j0)] = ct[T][ss];
                    HPF_BI_WAVE_LDS_FENCE();
                    bool weak;
                    const double wji = inv4_cofactor_lane(pv[0], lane, limit, weak);
                    if (incol) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) pcol[(lg + 4 * reg) * 4 + (jj - j0)] = ct[T][reg];
                    }
                    any_weak = any_weak || (weak && lane < 16);
                    if (lane < 16) wl[0][(lane & 3) * 4 + (lane >> 2)] = wji;
                    HPF_BI_WAVE_LDS_FENCE();
                    const double vcol = pcol[jj * 4 + lg];
                    const double aopl = incol ? 0.0 : vcol;
                    const double aw = jj < 4 ? wl[0][jj * 4 + lg] : 0.0;
                    const d4_t z = {0.0, 0.0, 0.0, 0.0};
                    const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ct[T][ss], z, 0, 0, 0);
                    double rfin = d[0];
                    if (incol) {
                        rfin = wl[0][lg * 4 + (jj - j0)];
                        ct[T] = d4_t{0.0, 0.0, 0.0, 0.0};
                    }
                    ct[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(aopl, -rfin, ct[T], 0, 0, 0);
                    ct[T][ss] = rfin;
                    HPF_BI_WAVE_LDS_FENCE();
                }
#pragma unroll
                for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) im[tr * 16 * WS + (4 * reg + lg) * WS + jj] = ct[tr][reg];
            }
            __syncthreads();
            d4_t Rn;
            if (wv == T) {
                Rn = -ct[T];
            } else {
                d4_t r4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(im[T * 16 * WS + jj * WS + 4 * k4 + lg], ct[T][k4], r4, 0, 0, 0);
                ct[T] = r4;
                Rn = -r4;
            }
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                if (tr == T) continue;
                d4_t a4 = ct[tr];
                if (wv == T) a4 = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(im[tr * 16 * WS + jj * WS + 4 * k4 + lg], Rn[k4], a4, 0, 0, 0);
                ct[tr] = a4;
            }
        }
    }
#pragma unroll
    for (int st = 4 * NB16; st < B / 4; ++st) {
        const int tP = st >> 2, rg = st & 3, j0 = 4 * (st & 3), buf = st & 1;
        const bool incol = jj >= j0 && jj < j0 + 4;
        if (wv == tP) {
            if (incol) pv[buf][lg * 4 + (jj - j0)] = ct[tP][rg];
            HPF_BI_WAVE_LDS_FENCE();
            bool weak;
            const double wji = inv4_cofactor_lane(pv[buf], lane, limit, weak);
            if (incol) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
            }
            any_weak = any_weak || (weak && lane < 16);
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
    if (any_weak) atomicOr(flag, 1);
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * tr + 4 * reg + lg;
            if (row < b && col < b) blk[(size_t)row * b + col] = ct[tr][reg];
        }
}

inline bool launch_invert_mfma(int BW, int b, double* blk, long long stride, int ny, double limit, int* flag, hipStream_t st) {
    switch (BW) {
        case 12: hipLaunchKernelGGL((k_blk_invert_mfma<12>), dim3((unsigned)ny), dim3(64), 0, st, b, blk, stride, limit, flag); return true;
        case 28: hipLaunchKernelGGL((k_blk_invert_mfma<28>), dim3((unsigned)ny), dim3(128), 0, st, b, blk, stride, limit, flag); return true;
        case 52: hipLaunchKernelGGL((k_blk_invert_mfma<52>), dim3((unsigned)ny), dim3(256), 0, st, b, blk, stride, limit, flag); return true;
        case 100: hipLaunchKernelGGL((k_blk_invert_mfma<100>), dim3((unsigned)ny), dim3(448), 0, st, b, blk, stride, limit, flag); return true;
        default: return false;
    }
}

}  // namespace
}  // namespace hpf
