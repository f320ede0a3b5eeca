// Blocked in-place Gauss-Jordan inversion on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), one wavefront per matrix.
//
// The matrix (b x b, padded to 16*NT) lives in registers in the MFMA accumulator layout: tile (tr, tc), register reg,
// lane l = 16*lg + jj  <->  element (row 16*tr + lg + 4*reg, column 16*tc + jj)        (cdna_hip_programming.md §3).
// Block step s eliminates the 4 rows/columns P = 4s..4s+3 with the 4x4 pivot block on the diagonal (STATIC pivot order:
// the pivot block is two harmonics = two physical 2x2 blocks, inverted by a 2x2 Schur complement):
//     W = A_PP^-1,  A_Pj <- W A_Pj (j not in P),  A_iP <- -A_iP W (i not in P),  A_ij <- A_ij - A_iP A_Pj,  A_PP <- W.
// Why the layout fits: row 4s+k of the pivot block sits in lane group lg = k, register s&3 of tile-row s>>2 — exactly
// the B-operand layout (lane 16k+jj supplies B[k][jj]) — so the pivot rows feed the MFMA with no data movement; the
// pivot columns (A operand, lane 16k+ii supplies A[ii][k]) go through a 64x4 LDS panel.  Per block step: 4 MFMAs scale
// the pivot rows (W padded to 16x4 times the old rows) and 16 MFMAs apply the rank-4 update with the pivot rows masked
// out of the A operand and the pivot columns' accumulators zeroed, so the same instruction also produces -A_iP W.
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

// inverse of a 4x4 given as 2x2 blocks [P Q; R S] through the Schur complement T = S - R P^-1 Q (wave-uniform data)
__device__ __forceinline__ void inv4(const double (&m)[4][4], double (&w)[4][4]) {
    double p00, p01, p10, p11;
    inv2(m[0][0], m[0][1], m[1][0], m[1][1], p00, p01, p10, p11);                 // P^-1
    // X = P^-1 Q
    const double x00 = fma(p00, m[0][2], p01 * m[1][2]), x01 = fma(p00, m[0][3], p01 * m[1][3]);
    const double x10 = fma(p10, m[0][2], p11 * m[1][2]), x11 = fma(p10, m[0][3], p11 * m[1][3]);
    // Yr = R P^-1
    const double y00 = fma(m[2][0], p00, m[2][1] * p10), y01 = fma(m[2][0], p01, m[2][1] * p11);
    const double y10 = fma(m[3][0], p00, m[3][1] * p10), y11 = fma(m[3][0], p01, m[3][1] * p11);
    // T = S - R X
    const double t00 = m[2][2] - fma(m[2][0], x00, m[2][1] * x10), t01 = m[2][3] - fma(m[2][0], x01, m[2][1] * x11);
    const double t10 = m[3][2] - fma(m[3][0], x00, m[3][1] * x10), t11 = m[3][3] - fma(m[3][0], x01, m[3][1] * x11);
    double i00, i01, i10, i11;
    inv2(t00, t01, t10, t11, i00, i01, i10, i11);                                  // T^-1 -> lower right
    w[2][2] = i00; w[2][3] = i01; w[3][2] = i10; w[3][3] = i11;
    // lower left = -T^-1 (R P^-1)
    w[2][0] = -fma(i00, y00, i01 * y10); w[2][1] = -fma(i00, y01, i01 * y11);
    w[3][0] = -fma(i10, y00, i11 * y10); w[3][1] = -fma(i10, y01, i11 * y11);
    // upper right = -X T^-1
    const double u00 = -fma(x00, i00, x01 * i10), u01 = -fma(x00, i01, x01 * i11);
    const double u10 = -fma(x10, i00, x11 * i10), u11 = -fma(x10, i01, x11 * i11);
    w[0][2] = u00; w[0][3] = u01; w[1][2] = u10; w[1][3] = u11;
    // upper left = P^-1 - (upper right)(R P^-1)
    w[0][0] = p00 - fma(u00, y00, u01 * y10); w[0][1] = p01 - fma(u00, y01, u01 * y11);
    w[1][0] = p10 - fma(u10, y00, u11 * y10); w[1][1] = p11 - fma(u10, y01, u11 * y11);
}

// Lane-parallel inverse of a 4x4 block held in LDS (pv[r*4+c], row-major): lane l computes ONE cofactor, (i, j) = ((l>>2)&3,
// l&3), from nine lane-addressed LDS reads; the determinant is the Laplace expansion along row i, summed over the lane quad
// with two DPP steps; returns W[j][i] = C_ij / det (the lane's element of the inverse, transposed position).  ~20 dependent
// FP64 operations instead of the ~80 of the uniform Schur-complement form (inv4), no v_readlane.  Every lane of the wave
// takes part (lanes >= 16 repeat the pattern).
__device__ __forceinline__ double inv4_cofactor_lane(const double* pv, int lane) {
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

// c: NT x NT accumulator tiles; NBS: number of 4x4 block steps (compile time: identity-padded rows / columns are no-ops,
// and a fixed step count keeps the accumulators in place between steps); panel: LDS, NT*64 + 16 doubles.
template <int NT, int NBS>
__device__ __forceinline__ void gauss_jordan_mfma(d4_t (&c)[NT][NT], double* panel) {
    const int lane = threadIdx.x, lg = lane >> 4, jj = lane & 15;
    double* wl = panel + NT * 64;
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
        {
            constexpr int dummy = 0;
            (void)dummy;
            const int tP = s >> 2, rg = s & 3, j0 = 4 * (s & 3);
            const bool incol = jj >= j0 && jj < j0 + 4;
            // 1. pivot block -> uniform registers, 2. its inverse
            double m[4][4], w[4][4];
            {
                const double app = c[tP][tP][rg];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) m[k][k2] = rl_f64(app, 16 * k + j0 + k2);
            }
            inv4(m, w);
            __syncthreads();
            // 3. pivot columns (64 x 4 panel) and W -> LDS
            if (incol) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) panel[(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = c[tr][tP][reg];
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) wl[k * 4 + k2] = w[k][k2];
            }
            __syncthreads();
            // 4. operands
            double aop[NT];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                const double v = panel[(16 * tr + jj) * 4 + lg];
                aop[tr] = (tr == tP && incol) ? 0.0 : -v;            // pivot rows are not updated by the rank-4 MFMA
            }
            const double aw = jj < 4 ? wl[jj * 4 + lg] : 0.0;         // A operand of the row scaling: W padded to 16 x 4
            const double wsel = incol ? wl[lg * 4 + (jj - j0)] : 0.0; // W[lg][jj - j0]
            // 5. new pivot rows W * A_P,:
            double rfin[NT];
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) {
                const d4_t z = {0.0, 0.0, 0.0, 0.0};
                const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, c[tP][tc][rg], z, 0, 0, 0);
                rfin[tc] = (tc == tP && incol) ? wsel : d[0];
            }
            // 6. rank-4 update (also produces -A_iP W in the zeroed pivot columns)
            if (incol) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr) c[tr][tP] = d4_t{0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int tc = 0; tc < NT; ++tc)
                    c[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[tr], rfin[tc], c[tr][tc], 0, 0, 0);
            // 7. pivot rows
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) c[tP][tc][rg] = rfin[tc];
        }
    }
}


// Layout conversion through LDS.  Row-per-lane registers a[0..B-1] (+ right-hand side y, placed in column B, which is
// never a pivot column, so the elimination turns it into A^-1 y) -> accumulator tiles.  tbuf: LDS, 64*17 doubles.
template <int B, int NT>
__device__ __forceinline__ void rows_to_tiles(const double (&a)[B], double y, int lane, d4_t (&c)[NT][NT], double* tbuf) {
    static_assert(B < 16 * NT, "a spare column is needed for the right-hand side");
    const int lg = lane >> 4, jj = lane & 15;
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int col = 16 * tc + j;
            double v;
            if (lane < B && col < B)
                v = a[col < B ? col : 0];
            else if (lane < B && col == B)
                v = y;
            else
                v = (lane == col) ? 1.0 : 0.0;      // identity rows (lane >= B) and columns (col > B)
            tbuf[lane * 17 + j] = v;
        }
        __syncthreads();
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) c[tr][tc][reg] = tbuf[(16 * tr + lg + 4 * reg) * 17 + jj];
    }
}

// Accumulator tiles -> global memory: transposed inverse AinvT[col][row] (B x B, coalesced per column) and w = column B.
template <int B, int NT>
__device__ __forceinline__ void tiles_to_global(const d4_t (&c)[NT][NT], int lane, double* tbuf, double* __restrict__ AinvT,
                                                double* __restrict__ w) {
    const int lg = lane >> 4, jj = lane & 15;
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
        __syncthreads();
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) tbuf[jj * 65 + 16 * tr + lg + 4 * reg] = c[tr][tc][reg];
        __syncthreads();
        if (lane < B) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int col = 16 * tc + j;
                if (col < B)
                    AinvT[(size_t)col * B + lane] = tbuf[j * 65 + lane];
                else if (col == B)
                    w[lane] = tbuf[j * 65 + lane];
            }
        }
    }
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

// Child -> parent Schur complement in the accumulator layout.  With the inverse Ainv (and w = Ainv y in column B) in the
// tiles, overwrite them by   C = G Ainv H   (columns < B)   and   G w   (column B),   where G = A(parent, child) and
// H = A(child, parent) are harmonic-diagonal: gl[q*4 + 2*t + t1] = G_q[t][t1], hl[q*4 + 2*t2 + tc] = H_q[t2][tc] (LDS,
// zero for q >= Hn).  Element (i, c) needs the 2x2 block of Ainv around it: own, row partner (lane^16), column partner
// (lane^1) and the diagonal partner.
template <int B, int NT>
__device__ __forceinline__ void schur_tiles(d4_t (&c)[NT][NT], int lane, const double* gl, const double* hl) {
    const int lg = lane >> 4, jj = lane & 15;
    const int ti = lg & 1, tcn = jj & 1;
    constexpr int tcB = B >> 4, jjB = B & 15;
    double ha[NT], hb[NT];
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
        const int p = 8 * tc + (jj >> 1);
        ha[tc] = hl[p * 4 + 2 * tcn + tcn];             // H[tcn][tcn]
        hb[tc] = hl[p * 4 + 2 * (tcn ^ 1) + tcn];       // H[tcn^1][tcn]
        if (tc == tcB && jj == jjB) {                   // right-hand-side column: C = G w
            ha[tc] = 1.0;
            hb[tc] = 0.0;
        }
        if (16 * tc + jj > B) {
            ha[tc] = 0.0;
            hb[tc] = 0.0;
        }
    }
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int q = 8 * tr + 2 * reg + (lg >> 1);
            const double ga = gl[q * 4 + 2 * ti + ti];          // G[ti][ti]
            const double gb = gl[q * 4 + 2 * ti + (ti ^ 1)];    // G[ti][ti^1]
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) {
                const double own = c[tr][tc][reg];
                const double rowp = xor16_f64(own);
                const double colp = xor1_f64(own);
                const double both = xor1_f64(rowp);
                c[tr][tc][reg] = fma(gb, fma(both, hb[tc], rowp * ha[tc]), ga * fma(colp, hb[tc], own * ha[tc]));
            }
        }
}

// Accumulator tiles -> global, B+1 columns (columns 0..B-1 and the right-hand-side column B), column-major [col][row].
template <int B, int NT>
__device__ __forceinline__ void tiles_to_global_aug(const d4_t (&c)[NT][NT], int lane, double* tbuf, double* __restrict__ C) {
    const int lg = lane >> 4, jj = lane & 15;
#pragma unroll
    for (int tc = 0; tc < NT; ++tc) {
        __syncthreads();
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) tbuf[jj * 65 + 16 * tr + lg + 4 * reg] = c[tr][tc][reg];
        __syncthreads();
        if (lane < B) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int col = 16 * tc + j;
                if (col <= B) C[(size_t)col * B + lane] = tbuf[j * 65 + lane];
            }
        }
    }
}

}  // namespace hpf
