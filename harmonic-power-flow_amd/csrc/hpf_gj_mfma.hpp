// Helpers of the blocked in-place Gauss-Jordan inversion on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) used by the
// multi-wave block-tree kernels (hpf_quad.hpp, hpf_leafbatch.hpp).
//
// A block (b x b, padded to 16*NT) lives in registers in the MFMA accumulator layout: tile (tr, tc), register reg,
// lane l = 16*lg + jj  <->  element (row 16*tr + lg + 4*reg, column 16*tc + jj)        (cdna_hip_programming.md §3).
// Block step s eliminates the 4 rows/columns P = 4s..4s+3 with the 4x4 pivot block on the diagonal (STATIC pivot order:
// the pivot block is two harmonics = two physical 2x2 blocks):
//     W = A_PP^-1,  A_Pj <- W A_Pj (j not in P),  A_iP <- -A_iP W (i not in P),  A_ij <- A_ij - A_iP A_Pj,  A_PP <- W.
// Row 4s+k of the pivot block sits in lane group lg = k, register s&3 of tile-row s>>2 -- exactly the B-operand layout (lane
// 16k+jj supplies B[k][jj]) -- so the pivot rows feed the MFMA with no data movement; the pivot columns (A operand) go through
// a 64x4 LDS panel.  The static order is watched: every pivot block reports how far its inverse amplifies (inv4_cofactor_lane).
#pragma once
#include <hip/hip_runtime.h>

namespace hpf {

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double rl_f64(double v, int lane_uniform) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane_uniform);
    hi = __builtin_amdgcn_readlane(hi, lane_uniform);
    return __hiloint2double(hi, lo);
}

// reciprocal on the pivot critical path: v_rcp_f64 (~2^-23 relative) + two Newton steps -> within 1 ulp of 1/x, a third of
// the latency of the IEEE division sequence
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

// inverse of a 2x2 [a b; c d]
__device__ __forceinline__ void inv2(double a, double b, double c, double d, double& ia, double& ib, double& ic, double& id) {
    const double r = rcp_nr(fma(a, d, -(b * c)));
    ia = d * r;
    ib = -b * r;
    ic = -c * r;
    id = a * r;
}

// Lane-parallel inverse of a 4x4 block held in LDS (pv[r*4+c], row-major): lane l computes ONE cofactor, (i, j) = ((l>>2)&3,
// l&3), from nine lane-addressed LDS reads; the determinant is the Laplace expansion along row i, summed over the lane quad
// with two DPP steps; returns W[j][i] = C_ij / det (the lane's element of the inverse, transposed position).  ~20 dependent
// FP64 operations instead of the ~80 of the uniform Schur-complement form (inv4), no v_readlane.  Every lane of the wave
// takes part (lanes >= 16 repeat the pattern).
// `weak`: pivot-growth monitor of the static pivot order.  det = sum_j a_ij C_ij; |a_ij C_ij| / |det| = |a_ij W_ji| is a lower
// bound of the block's condition number, so a term that exceeds `limit` times |det| (or a zero / non-finite det) marks a pivot
// block whose inverse amplifies rounding errors by more than `limit`: the scenario is then repeated with partial pivoting.
__device__ __forceinline__ double inv4_cofactor_lane(const double* pv, int lane, double limit, bool& weak) {
    const int i = (lane >> 2) & 3, j = lane & 3;
    const int r0 = i == 0 ? 1 : 0, r1 = i <= 1 ? 2 : 1, r2 = i == 3 ? 2 : 3;
    const int c0 = j == 0 ? 1 : 0, c1 = j <= 1 ? 2 : 1, c2 = j == 3 ? 2 : 3;
    const double a00 = pv[r0 * 4 + c0], a01 = pv[r0 * 4 + c1], a02 = pv[r0 * 4 + c2];
    const double a10 = pv[r1 * 4 + c0], a11 = pv[r1 * 4 + c1], a12 = pv[r1 * 4 + c2];
    const double a20 = pv[r2 * 4 + c0], a21 = pv[r2 * 4 + c1], a22 = pv[r2 * 4 + c2];
    const double aij = pv[i * 4 + j];
    const double m0 = fma(a11, a22, -(a12 * a21));
    const double m1 = fma(a10, a22, -(a12 * a20));
    const double m2 = fma(a10, a21, -(a11 * a20));
    double cof = fma(a02, m2, fma(a00, m0, -(a01 * m1)));
    cof = ((i + j) & 1) ? -cof : cof;
    double det = aij * cof;
    {
        int lo = __double2loint(det), hi = __double2hiint(det);
        det += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false));       // quad_perm [1,0,3,2]
        lo = __double2loint(det), hi = __double2hiint(det);
        det += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false));       // quad_perm [2,3,0,1]
    }
    weak = !(fabs(det) * limit >= fabs(aij * cof));          // (also true for det = 0 and for NaN)
    return cof * rcp_nr(det);
}

// The same with the cofactor position (i, j) chosen by the caller: returns C_ij / det = W[j][i].  The four lanes of a DPP quad
// must hold either the four j of one i (row expansion of the determinant) or the four i of one j (column expansion).
__device__ __forceinline__ double inv4_cofactor_ij(const double* pv, int i, int j) {
    const int r0 = i == 0 ? 1 : 0, r1 = i <= 1 ? 2 : 1, r2 = i == 3 ? 2 : 3;
    const int c0 = j == 0 ? 1 : 0, c1 = j <= 1 ? 2 : 1, c2 = j == 3 ? 2 : 3;
    const double a00 = pv[r0 * 4 + c0], a01 = pv[r0 * 4 + c1], a02 = pv[r0 * 4 + c2];
    const double a10 = pv[r1 * 4 + c0], a11 = pv[r1 * 4 + c1], a12 = pv[r1 * 4 + c2];
    const double a20 = pv[r2 * 4 + c0], a21 = pv[r2 * 4 + c1], a22 = pv[r2 * 4 + c2];
    const double aij = pv[i * 4 + j];
    const double m0 = fma(a11, a22, -(a12 * a21));
    const double m1 = fma(a10, a22, -(a12 * a20));
    const double m2 = fma(a10, a21, -(a11 * a20));
    double cof = fma(a02, m2, fma(a00, m0, -(a01 * m1)));
    cof = ((i + j) & 1) ? -cof : cof;
    double det = aij * cof;
    {
        int lo = __double2loint(det), hi = __double2hiint(det);
        det += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false));       // quad_perm [1,0,3,2]
        lo = __double2loint(det), hi = __double2hiint(det);
        det += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false));       // quad_perm [2,3,0,1]
    }
    return cof * rcp_nr(det);
}

// Cross-lane partner values in the accumulator layout: lane ^ 1 holds the neighbouring column, lane ^ 16 the neighbouring row.
__device__ __forceinline__ double xor1_f64(double v) {          // DPP quad_perm [1,0,3,2]
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor16_f64(double v) {         // ds_swizzle, bit mode: and 0x1f, or 0, xor 0x10
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
    return __hiloint2double(hi, lo);
}

}  // namespace hpf
