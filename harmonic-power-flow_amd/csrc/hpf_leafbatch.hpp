// Lazy leaves, 16 scenarios per workgroup (included by hpf_block.hip after hpf_quad.hpp).
//
// A lazy constant-inverse leaf owes the factor sweep only vectors: w = D^-1 y, G w for its parent, its 2x2 core K
// with the position-0 borders, S^-1 and A(k,parent) for the back sweep.  The one O(b^2) piece, Drect^-1 y, multiplies a PER-MODEL
// matrix with a per-scenario vector -- across scenarios that is a GEMM:
//     Drect^-1 y = [u; Vh + Lc u],   V = [0 Lr; 0 Ahh^-1] y,   u = K (y0 + V0),   K = (c0 + Delta_polar S_0^-1)^-1.
// One workgroup takes one leaf and SB = 16 scenarios: V for all 16 is b/4 rank-4 MFMAs per 16-row tile (A operand = the image in
// operand layout, read ONCE per 16 scenarios instead of once per scenario; B operand = the 16 right-hand sides from LDS), the
// roles of k_factor_q (right-hand side with the 2x2-algebra children folded in, S^-1, coupling blocks) run per (scenario, row) /
// (scenario, harmonic) thread.  Results are the same quantities k_factor_q<B, true> leaves for a lazy leaf.
#pragma once

constexpr int LB_SB = 16;      // scenarios per workgroup
constexpr int LBP = LB_SB + 1; // row stride of the [row][scenario] arrays in LDS (odd: a wave's 16 rows x 4 scenarios spread over the banks
                               // instead of meeting in one bank pair -- LDS bank conflicts per LDS instruction 6.9 -> see profiles/r03)

// per-leaf constants (Tree::d_lbimg): A-operand image [NTR][KS][64] | Lc as real b x 2 | c0 2x2
template <int B>
struct LeafBatchImg {
    static constexpr int NTR = (B + 15) / 16, KS = (B + 3) / 4;
    static constexpr int SZ = NTR * KS * 64 + 2 * B + 4;
};

// per-bus constants of a bordered bus (Tree::d_sbimg): A-operand image [NTR][KS][64] of [0 0; 0 Ahh^-1] -- with the rows of Qb in
// the padding rows B.. of the last row tile when m <= 10 of them fit (QB_ROWS: r = Qb v falls out of the same MFMAs) -- and Pb
// (b x m) as a second A operand [NTR][3][64] (x += Pb y: three rank-4 steps)
template <int B>
struct SleafImg {
    static constexpr int NTR = (B + 15) / 16, KS = (B + 3) / 4, KP = 3;
    static constexpr int MAIN = NTR * KS * 64, SZ = MAIN + NTR * KP * 64;
    static constexpr bool QB_ROWS = 16 * NTR - B >= 10;
};

constexpr int LEAF_BATCH_LDS = 2 * 64 * LBP + LB_SB * 4 + LB_SB * 2;      // doubles

template <int B>
__device__ __forceinline__ void leaf_batch_body(
    double* __restrict__ smem_, const int bx_, const int by_, const Model& M, const TreeDev& T, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt,
    const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, const double* __restrict__ fall, double* __restrict__ wall,
    const double* __restrict__ linAall, double* __restrict__ Call, double* __restrict__ Hall, const double* __restrict__ chG,
    const double* __restrict__ chH, const double* __restrict__ chD, const double* __restrict__ chy,
    const double* __restrict__ lbimg, double* __restrict__ lfK, double* __restrict__ lfS, int s0) {
    constexpr int NT = (B + 16) / 16;                       // tile columns of the block kernels (slot stride of C)
    constexpr size_t CT = (size_t)NT * NT * 256;
    constexpr int NTR = LeafBatchImg<B>::NTR, KS = LeafBatchImg<B>::KS, H2 = B / 2;
    const int4* nd = reinterpret_cast<const int4*>(nodes) + (FDESC / 4) * (size_t)bx_;
    const int4 nd0 = nd[0], nd1 = nd[1], nd3 = nd[3];
    const int k = nd0.x, par = nd0.y;
    const int e_dn_k = nd1.x, e_up_k = nd1.y, lin_beg = nd1.z, lin_end = nd1.z + nd1.w;
    const bool via_chain = (nd3.z & 1) != 0;
    const int slot = nd3.w - 1;                              // leaf slot (Tree::d_Minv numbering)
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = M.n, c = M.c, Hn = M.Hn;
    const int sc = tid >> 4, l16 = tid & 15;                 // role mapping: 16 threads per scenario
    const int sl = by_ * LB_SB + sc;                  // scenario inside the launch
    const int ss = sl < S_cnt ? (active ? active[sl + s0] : sl + s0) : -1;      // slot -> scenario (active list; -1: frozen / empty slot)
    const bool live = ss >= 0;
    const int s = live ? ss : 0;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* img = lbimg + (size_t)slot * LeafBatchImg<B>::SZ;
    const double* lcimg = img + NTR * KS * 64;               // R(Lc): [row][2] (rows 0, 1: identity)
    const double* c0img = lcimg + 2 * B;

    double* Y = smem_;                                       // right-hand sides [row][scenario]
    double* V = Y + 64 * LBP;                              // [0 Lr; 0 Ahh^-1] y
    double* DL = V + 64 * LBP;                             // Delta_polar per scenario
    double* UK = DL + LB_SB * 4;                             // u = K (y0 + V0)
    constexpr int QI = (H2 + 15) / 16;                       // harmonics per thread of a scenario's 16
    double sir[QI][4], glr[QI][4];                           // S_q^-1 and A(parent, k) of the thread's harmonics (same mapping in R2, K, F)

    // ---- R1. rows: right-hand side and the 2x2 term of the fundamental, the 2x2-algebra children folded in (k_factor_q, wave 0
    //      role + children, here one thread per (scenario, row) and every child of the bus in list order) -------------------------
    {
        // (the thread's four rows side by side: the children loop runs once, with the loads of all four rows in flight)
        double e0[4] = {0.0, 0.0, 0.0, 0.0}, e1[4] = {0.0, 0.0, 0.0, 0.0}, ey[4] = {0.0, 0.0, 0.0, 0.0};
        double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0}, fy[4] = {0.0, 0.0, 0.0, 0.0};
        bool ok[4];
        cplx ukr[4], ekr[4];
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) {
            const int row = l16 + 16 * pz, q = row >> 1, tr_ = row & 1;
            ok[pz] = live && row < b && loc_valid(n, c, k, row);
            ukr[pz] = cplx{0.0, 0.0};
            ekr[pz] = cplx{0.0, 0.0};
            if (ok[pz]) {
                const size_t kq = (size_t)k * Hn + q;
                fy[pz] = fall[((size_t)s * n + k) * B + row];
                if (via_chain) {
                    const size_t o = so + kq;
                    a0[pz] = chD[o * 4 + 2 * tr_];
                    a1[pz] = chD[o * 4 + 2 * tr_ + 1];
                    ay[pz] = chy[o * 2 + tr_];
                }
                if (lin_beg < lin_end) {
                    ukr[pz] = U[kq];
                    ekr[pz] = E[kq];
                }
            }
        }
        const double* ws = wall + (size_t)s * n * B;
        const double* linA = linAall + so * 4;
        const int4* c3 = reinterpret_cast<const int4*>(T.child3);
        for (int cp = lin_beg; cp < lin_end; ++cp) {
            const int4 cr = c3[cp];
            const int ch = cr.x;
#pragma unroll
            for (int pz = 0; pz < 4; ++pz) {
                if (!ok[pz]) continue;
                const int row = l16 + 16 * pz, q = row >> 1, tr_ = row & 1;
                const cplx uk = ukr[pz], ek = ekr[pz];
                const cplx ydn = M.Y[(size_t)cr.y * Hn + q], yup = M.Y[(size_t)cr.z * Hn + q];
                const cplx uc = U[(size_t)ch * Hn + q], ec = E[(size_t)ch * Hn + q];
                const double2* pic = reinterpret_cast<const double2*>(linA + ((size_t)ch * Hn + q) * 4);
                const double2 ic01 = pic[0], ic23 = pic[1];
                const double2 wc = *reinterpret_cast<const double2*>(ws + (size_t)ch * B + 2 * q);
                const Blk2 g = (q == 0 && k < M.m) ? blk_power_off(ydn, uk, uc, ec) : blk_current(ydn, uc, ec);   // A(k, child)
                const Blk2 hb = (q == 0 && ch < M.m) ? blk_power_off(yup, uc, uk, ek) : blk_current(yup, uk, ek);  // A(child, k)
                const double g0 = pick(g, tr_, 0);
                const double g1 = loc_valid(n, c, ch, 2 * q + 1) ? pick(g, tr_, 1) : 0.0;
                double h4[4];
                mask_block(n, c, q, ch, k, hb, h4);
                const double v0 = fma(g1, ic23.x, g0 * ic01.x), v1 = fma(g1, ic23.y, g0 * ic01.y);
                e0[pz] += fma(v1, h4[2], v0 * h4[0]);
                e1[pz] += fma(v1, h4[3], v0 * h4[1]);
                ey[pz] = fma(g0, wc.x, ey[pz]);
                ey[pz] = fma(g1, wc.y, ey[pz]);
            }
        }
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) {
            const int row = l16 + 16 * pz;
            Y[row * LBP + sc] = ok[pz] ? fy[pz] + ay[pz] - ey[pz] : 0.0;
            if (row < 2) {
                DL[sc * 4 + row * 2] = ok[pz] ? a0[pz] - e0[pz] : 0.0;
                DL[sc * 4 + row * 2 + 1] = ok[pz] ? a1[pz] - e1[pz] : 0.0;
            }
        }
    }
    // ---- R2. harmonics: S_q^-1, the coupling blocks with the parent (H is kept for the back sweep), the position-0 borders ----
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int q = l16 + 16 * it;
        double si[4] = {1.0, 0.0, 0.0, 1.0}, g4[4] = {0.0, 0.0, 0.0, 0.0}, h4[4] = {0.0, 0.0, 0.0, 0.0};
        if (live && q < Hn) {
            const cplx uk = U[(size_t)k * Hn + q], ek = E[(size_t)k * Hn + q];
            const double idet = 1.0 / (-(uk.im * ek.im) - ek.re * uk.re);      // S = [-ui er; ur ei]
            si[0] = ek.im * idet;   si[1] = -ek.re * idet;
            si[2] = -uk.re * idet;  si[3] = -uk.im * idet;
            cplx up = {0.0, -1.0}, ep = {0.0, 1.0};
            if (!via_chain || q == 0) {
                up = U[(size_t)par * Hn + q];
                ep = E[(size_t)par * Hn + q];
            }
            if (via_chain) {
                const double2* pg = reinterpret_cast<const double2*>(chG + (so + (size_t)k * Hn + q) * 4);
                const double2* ph = reinterpret_cast<const double2*>(chH + (so + (size_t)k * Hn + q) * 4);
                const double2 ga = pg[0], gb = pg[1], ha = ph[0], hb = ph[1];
                g4[0] = ga.x; g4[1] = ga.y; g4[2] = gb.x; g4[3] = gb.y;
                h4[0] = ha.x; h4[1] = ha.y; h4[2] = hb.x; h4[3] = hb.y;
            } else {
                const cplx ydn = M.Y[(size_t)e_dn_k * Hn + q], yup = M.Y[(size_t)e_up_k * Hn + q];
                const Blk2 g = (q == 0 && par < M.m) ? blk_power_off(ydn, up, uk, ek) : blk_current(ydn, uk, ek);   // row par, col k
                const Blk2 hh = (q == 0 && k < M.m) ? blk_power_off(yup, uk, up, ep) : blk_current(yup, up, ep);     // row k, col par
                mask_block(n, c, q, par, k, g, g4);
                mask_block(n, c, q, k, par, hh, h4);
            }
            double* Hk = Hall + ((size_t)s * n + k) * Hn * 4 + q * 4;
            double* Sk = lfS + (so + (size_t)k * Hn + q) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Hk[e] = h4[e];
                Sk[e] = si[e];
            }
            if (q == 0) {                                    // G0 S_c^-1, H0 S_p^-1 for the parent's rebuild
                const double ip = 1.0 / (-(up.im * ep.im) - ep.re * up.re);
                const double sp0 = ep.im * ip, sp1 = -ep.re * ip, sp2 = -up.re * ip, sp3 = -up.im * ip;
                double* kk = lfK + ((size_t)s * n + k) * 12 + 4;
                kk[0] = fma(g4[1], si[2], g4[0] * si[0]);  kk[1] = fma(g4[1], si[3], g4[0] * si[1]);
                kk[2] = fma(g4[3], si[2], g4[2] * si[0]);  kk[3] = fma(g4[3], si[3], g4[2] * si[1]);
                kk[4] = fma(h4[1], sp2, h4[0] * sp0);  kk[5] = fma(h4[1], sp3, h4[0] * sp1);
                kk[6] = fma(h4[3], sp2, h4[2] * sp0);  kk[7] = fma(h4[3], sp3, h4[2] * sp1);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sir[it][e] = si[e];
            glr[it][e] = g4[e];
        }
    }
    __syncthreads();
    // ---- M. V = [0 Lr; 0 Ahh^-1] Y for the 16 scenarios: wave w owns rows 16 w .. 16 w + 15 ---------------------------------------
    if (wv < NTR) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        const double* ia = img + (size_t)wv * KS * 64 + lane;
#pragma unroll 4
        for (int ks = 0; ks < KS; ++ks) {
            const double a = ia[(size_t)ks * 64];
            const double bop = Y[(4 * ks + lg) * LBP + jj];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) V[(16 * wv + 4 * reg + lg) * LBP + jj] = acc[reg];
    }
    __syncthreads();
    // ---- K. per scenario: K = (c0 + Delta_polar S_0^-1)^-1, u = K (y0 + V0) --------------------------------------------------------
    if (l16 == 0) {
        const double* si = sir[0];                           // (q = 0 belongs to this thread)
        const double* dl = DL + sc * 4;
        const double q00 = c0img[0] + fma(dl[1], si[2], dl[0] * si[0]), q01 = c0img[1] + fma(dl[1], si[3], dl[0] * si[1]);
        const double q10 = c0img[2] + fma(dl[3], si[2], dl[2] * si[0]), q11 = c0img[3] + fma(dl[3], si[3], dl[2] * si[1]);
        double k00, k01, k10, k11;
        inv2(q00, q01, q10, q11, k00, k01, k10, k11);
        const double r0 = Y[sc] + V[sc], r1 = Y[LBP + sc] + V[LBP + sc];
        UK[sc * 2] = fma(k01, r1, k00 * r0);
        UK[sc * 2 + 1] = fma(k11, r1, k10 * r0);
        if (live) {
            double* kk = lfK + ((size_t)s * n + k) * 12;
            kk[0] = k00; kk[1] = k01; kk[2] = k10; kk[3] = k11;
        }
    }
    __syncthreads();
    // ---- F. per (scenario, harmonic): x_rect = [u; Vh + Lc u], w = S_q^-1 x_rect, G w for the parent -----------------------------
    if (live) {
        const double u0 = UK[sc * 2], u1 = UK[sc * 2 + 1];
        double* wk = wall + ((size_t)s * n + k) * B;
        double* Ck = Call + ((size_t)s * n + k) * CT;
#pragma unroll
        for (int it = 0; it < QI; ++it) {
            const int q = l16 + 16 * it;
            if (q >= H2) continue;
            double x0 = u0, x1 = u1;
            if (q > 0) {
                const double* lc = lcimg + (2 * q) * 2;
                x0 = V[(2 * q) * LBP + sc] + fma(lc[1], u1, lc[0] * u0);
                x1 = V[(2 * q + 1) * LBP + sc] + fma(lc[3], u1, lc[2] * u0);
            }
            const double* si = sir[it];
            const bool in = 2 * q < b;
            const double w0 = in ? fma(si[1], x1, si[0] * x0) : 0.0, w1 = in ? fma(si[3], x1, si[2] * x0) : 0.0;
            const double* g = glr[it];
            wk[2 * q] = w0;
            wk[2 * q + 1] = w1;
            Ck[2 * q] = fma(g[1], w1, g[0] * w0);
            Ck[2 * q + 1] = fma(g[3], w1, g[2] * w0);
        }
    }
}

template <int B>
__global__ __launch_bounds__(256, 4) void k_leaf_batch(
    Model M, TreeDev T, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt,
    const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, const double* __restrict__ fall, double* __restrict__ wall,
    const double* __restrict__ linAall, double* __restrict__ Call, double* __restrict__ Hall, const double* __restrict__ chG,
    const double* __restrict__ chH, const double* __restrict__ chD, const double* __restrict__ chy,
    const double* __restrict__ lbimg, double* __restrict__ lfK, double* __restrict__ lfS, int s0) {
    __shared__ __attribute__((aligned(16))) double smem[LEAF_BATCH_LDS];
    leaf_batch_body<B>(smem, blockIdx.x, blockIdx.y, M, T, nodes, b, active, S_cnt, Uall, Eall, fall, wall, linAall, Call, Hall, chG, chH, chD, chy,
                       lbimg, lfK, lfS, s0);
}

template <int B>
int launch_leaf_batch(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    ScopedTimer t(h, T_SOLVE);
    const dim3 grid((unsigned)count, (unsigned)((h->cur_S + LB_SB - 1) / LB_SB));
    hipLaunchKernelGGL((k_leaf_batch<B>), grid, dim3(256), 0, h->cur_stream, h->M, T, nodes, 2 * h->Hn, active, h->cur_S, h->d_U,
                       h->d_E, h->d_fb, h->d_w, h->d_linA, h->d_C, h->d_H, h->d_chG, h->d_chH, h->d_chD, h->d_chy,
                       active_tree(h).d_lbimg, h->d_lfK, h->d_lfS, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

// Back sweep of the constant-inverse leaves, 16 scenarios per workgroup:  x_k = w_k - S^-1 Drect^-1 t,  t = A(k,parent) x_parent,
// Drect^-1 t = [u; Vh + Lc u],  V = [0 Lr; 0 Ahh^-1] t  (the same per-model image on the matrix cores),  u = K (t0 + V0).
// nodes: Tree::d_bdesc records (bus, parent, leaf slot + 1, 0) of leaves only.
template <int B>
__device__ __forceinline__ void leaf_back_batch_body(
    const int bx_, const int by_, const Model& M, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt, const double* __restrict__ wall,
    double* __restrict__ xall, const double* __restrict__ Hall, const double* __restrict__ lbimg, const double* __restrict__ lfK,
    const double* __restrict__ lfS, int s0) {
    constexpr int NTR = LeafBatchImg<B>::NTR, KS = LeafBatchImg<B>::KS, H2 = B / 2, QI = (H2 + 15) / 16;
    const int4 kp = reinterpret_cast<const int4*>(nodes)[bx_];
    const int k = kp.x, par = kp.y, slot = kp.z - 1;
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = M.n, Hn = M.Hn;
    const int sc = tid >> 4, l16 = tid & 15;
    const int sl = by_ * LB_SB + sc;
    const int ss = sl < S_cnt ? (active ? active[sl + s0] : sl + s0) : -1;      // slot -> scenario (active list; -1: frozen / empty slot)
    const bool live = ss >= 0;
    const int s = live ? ss : 0;
    double* xs = xall + (size_t)s * n * B;
    const double* img = lbimg + (size_t)slot * LeafBatchImg<B>::SZ;
    const double* lcimg = img + NTR * KS * 64;

    __shared__ double TT[64 * LBP];                        // t = A(k,parent) x_parent, [row][scenario]
    __shared__ double V[64 * LBP];
    __shared__ double UK[LB_SB * 2];

    // one round trip for everything addressed by the record: the lane's image column, K, and Lc / S^-1 / w of the thread's harmonics
    double ia[KS], k4[4] = {0.0, 0.0, 0.0, 0.0}, lc4[QI][4], si4[QI][4], w4[QI][2];
    if (wv < NTR) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ia[ks] = img[((size_t)wv * KS + ks) * 64 + lane];
    }
    if (live && l16 == 0) {
        const double* kk = lfK + ((size_t)s * n + k) * 12;
        k4[0] = kk[0]; k4[1] = kk[1]; k4[2] = kk[2]; k4[3] = kk[3];
    }
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int q = l16 + 16 * it;
        w4[it][0] = w4[it][1] = 0.0;
        si4[it][0] = si4[it][3] = 1.0;
        si4[it][1] = si4[it][2] = 0.0;
        lc4[it][0] = lc4[it][1] = lc4[it][2] = lc4[it][3] = 0.0;
        if (live && q < H2) {
            const double2 w2 = *reinterpret_cast<const double2*>(wall + ((size_t)s * n + k) * B + 2 * q);
            w4[it][0] = w2.x;
            w4[it][1] = w2.y;
            const double2* lp = reinterpret_cast<const double2*>(lcimg + (2 * q) * 2);
            const double2 l0 = lp[0], l1 = lp[1];
            lc4[it][0] = l0.x; lc4[it][1] = l0.y; lc4[it][2] = l1.x; lc4[it][3] = l1.y;
            if (q < Hn) {
                const double2* sp = reinterpret_cast<const double2*>(lfS + (((size_t)s * n + k) * Hn + q) * 4);
                const double2 a = sp[0], c2 = sp[1];
                si4[it][0] = a.x; si4[it][1] = a.y; si4[it][2] = c2.x; si4[it][3] = c2.y;
            }
        }
    }
    for (int q = l16; q < 32; q += 16) {
        double t0 = 0.0, t1 = 0.0;
        if (live && q < Hn) {
            const double* hk = Hall + (((size_t)s * n + k) * Hn + q) * 4;
            const double2 xp = *reinterpret_cast<const double2*>(xs + (size_t)par * B + 2 * q);
            t0 = fma(hk[1], xp.y, hk[0] * xp.x);
            t1 = fma(hk[3], xp.y, hk[2] * xp.x);
        }
        TT[(2 * q) * LBP + sc] = t0;
        TT[(2 * q + 1) * LBP + sc] = t1;
    }
    __syncthreads();
    if (wv < NTR) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double bop = TT[(4 * ks + lg) * LBP + jj];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[ks], bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) V[(16 * wv + 4 * reg + lg) * LBP + jj] = acc[reg];
    }
    __syncthreads();
    if (l16 == 0) {
        const double r0 = TT[sc] + V[sc], r1 = TT[LBP + sc] + V[LBP + sc];                      // [I Lr] t
        UK[sc * 2] = fma(k4[1], r1, k4[0] * r0);
        UK[sc * 2 + 1] = fma(k4[3], r1, k4[2] * r0);
    }
    __syncthreads();
    if (live) {
        const double u0 = UK[sc * 2], u1 = UK[sc * 2 + 1];
#pragma unroll
        for (int it = 0; it < QI; ++it) {
            const int q = l16 + 16 * it;
            if (q >= H2) continue;
            double x0 = u0, x1 = u1;
            if (q > 0) {
                x0 = V[(2 * q) * LBP + sc] + fma(lc4[it][1], u1, lc4[it][0] * u0);
                x1 = V[(2 * q + 1) * LBP + sc] + fma(lc4[it][3], u1, lc4[it][2] * u0);
            }
            const double d0 = fma(si4[it][1], x1, si4[it][0] * x0), d1 = fma(si4[it][3], x1, si4[it][2] * x0);
            *reinterpret_cast<double2*>(xs + (size_t)k * B + 2 * q) = double2{w4[it][0] - d0, w4[it][1] - d1};
        }
    }
}

template <int B>
__global__ __launch_bounds__(256) void k_leaf_back_batch(
    Model M, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt, const double* __restrict__ wall,
    double* __restrict__ xall, const double* __restrict__ Hall, const double* __restrict__ lbimg, const double* __restrict__ lfK,
    const double* __restrict__ lfS, int s0) {
    leaf_back_batch_body<B>(blockIdx.x, blockIdx.y, M, nodes, b, active, S_cnt, wall, xall, Hall, lbimg, lfK, lfS, s0);
}

template <int B>
int launch_leaf_back_batch(hpf_handle* h, const int* nodes, int count, const int* active) {
    const dim3 grid((unsigned)count, (unsigned)((h->cur_S + LB_SB - 1) / LB_SB));
    hipLaunchKernelGGL((k_leaf_back_batch<B>), grid, dim3(256), 0, h->cur_stream, h->M, nodes, 2 * h->Hn, active, h->cur_S, h->d_w,
                       h->d_x, h->d_H, active_tree(h).d_lbimg, h->d_lfK, h->d_lfS, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

// Back sweep of the super-leaves, 16 scenarios per workgroup:  x_k = w_k - S_k^-1 M_k^-1 (Wd^-1 t),  t = A(k,parent) x_parent,
//     M_k^-1 v = [0 0; 0 Ahh_k^-1] v + Pb (T^-1 (Qb v)),
// the Ahh_k^-1 part on the matrix cores (per-model image in A-operand layout), the m <= 10 border unknowns per thread; T^-1 and
// W_k^-1 were left by the factor kernel at the head of the bus's (otherwise unused) inverse slot.
// nodes: records of 8 ints (bus, parent, slot in Tree::d_sbimg, offset of [Tc | Pb | Qb] in Tree::d_lzimg, m, 0, 0, 0).
template <int B>
__device__ __forceinline__ void sleaf_back_batch_body(
    const int bx_, const int by_, const Model& M, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt, const double* __restrict__ wall,
    double* __restrict__ xall, const double* __restrict__ Hall, const double* __restrict__ sbimg, const double* __restrict__ lzimg,
    const double* __restrict__ Zall, const double* __restrict__ lfS, int s0) {
    constexpr int NT = (B + 16) / 16;
    constexpr size_t CT = (size_t)NT * NT * 256;
    constexpr int NTR = SleafImg<B>::NTR, KS = SleafImg<B>::KS, KP = SleafImg<B>::KP, H2 = B / 2, QI = (H2 + 15) / 16;
    constexpr bool QBR = SleafImg<B>::QB_ROWS;
    const int4* rec = reinterpret_cast<const int4*>(nodes) + 2 * (size_t)bx_;
    const int4 r0 = rec[0], r1 = rec[1];
    const int k = r0.x, par = r0.y, m = r1.x;
    const double* img = sbimg + (size_t)r0.z * SleafImg<B>::SZ;
    const double* pbm = lzimg + (size_t)r0.w + m * m;           // Pb [b][m]
    const double* qbm = pbm + (size_t)b * m;                    // Qb [m][b]
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = M.n, Hn = M.Hn;
    const int sc = tid >> 4, l16 = tid & 15;
    const int sl = by_ * LB_SB + sc;
    const int ss = sl < S_cnt ? (active ? active[sl + s0] : sl + s0) : -1;      // slot -> scenario (active list; -1: frozen / empty slot)
    const bool live = ss >= 0;
    const int s = live ? ss : 0;
    double* xs = xall + (size_t)s * n * B;
    const double* tk = Zall + ((size_t)s * n + k) * CT;         // T^-1 [10][10] | W_k^-1 [4]

    __shared__ double TT[64 * LBP];                           // v = Wd^-1 t, [row][scenario]
    __shared__ double V[64 * LBP];
    __shared__ double RR[16 * LBP];                           // r = Qb v, then y = T^-1 r, [border unknown][scenario]
    __shared__ double YY[16 * LBP];

    // every operand whose address comes from the record alone is requested here, in one round trip with A(k,parent) and x_parent:
    // the image column of the lane's MFMAs, the thread's row of T^-1, S^-1 and w of its harmonics
    double ia[KS], pa[KP], trow[10], w4[QI][2], si4[QI][4];
    if (wv < NTR) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ia[ks] = img[((size_t)wv * KS + ks) * 64 + lane];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) pa[kp] = img[SleafImg<B>::MAIN + ((size_t)wv * KP + kp) * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) trow[j] = (live && l16 < m && j < m) ? tk[l16 * 10 + j] : 0.0;
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int q = l16 + 16 * it;
        w4[it][0] = w4[it][1] = 0.0;
        si4[it][0] = si4[it][3] = 1.0;
        si4[it][1] = si4[it][2] = 0.0;
        if (live && q < H2) {
            const double2 w2 = *reinterpret_cast<const double2*>(wall + ((size_t)s * n + k) * B + 2 * q);
            w4[it][0] = w2.x;
            w4[it][1] = w2.y;
            if (q < Hn) {
                const double2* sp = reinterpret_cast<const double2*>(lfS + (((size_t)s * n + k) * Hn + q) * 4);
                const double2 a = sp[0], c2 = sp[1];
                si4[it][0] = a.x; si4[it][1] = a.y; si4[it][2] = c2.x; si4[it][3] = c2.y;
            }
        }
    }
    for (int q = l16; q < 32; q += 16) {
        double t0 = 0.0, t1 = 0.0;
        if (live && q < Hn) {
            const double* hk = Hall + (((size_t)s * n + k) * Hn + q) * 4;
            const double2 xp = *reinterpret_cast<const double2*>(xs + (size_t)par * B + 2 * q);
            t0 = fma(hk[1], xp.y, hk[0] * xp.x);
            t1 = fma(hk[3], xp.y, hk[2] * xp.x);
            if (q == 0) {                                       // power rows of a linear bus arrive in polar form: W_k^-1 (identity otherwise)
                const double a = t0, c2 = t1;
                t0 = fma(tk[101], c2, tk[100] * a);
                t1 = fma(tk[103], c2, tk[102] * a);
            }
        }
        TT[(2 * q) * LBP + sc] = t0;
        TT[(2 * q + 1) * LBP + sc] = t1;
    }
    __syncthreads();
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (wv < NTR) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double bop = TT[(4 * ks + lg) * LBP + jj];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ia[ks], bop, acc, 0, 0, 0);
        }
        if (QBR && wv == NTR - 1) {                             // rows B.. of the last row tile: r = Qb v
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = 16 * (NTR - 1) + 4 * reg + lg - B;
                if (j >= 0 && j < 16) RR[j * LBP + jj] = acc[reg];
            }
        }
    }
    if (!QBR) {
        double r = 0.0;                                         // r_j = Qb[j][:] v   (thread (scenario, j))
        if (l16 < m) {
            const double* qr = qbm + (size_t)l16 * b;
            for (int col = 0; col < b; ++col) r = fma(qr[col], TT[col * LBP + sc], r);
        }
        RR[l16 * LBP + sc] = r;
    }
    __syncthreads();
    {
        double y = 0.0;                                         // y_i = T^-1[i][:] r
#pragma unroll
        for (int j = 0; j < 10; ++j) y = fma(trow[j], (j < m ? RR[j * LBP + sc] : 0.0), y);
        YY[l16 * LBP + sc] = (live && l16 < m) ? y : 0.0;
    }
    __syncthreads();
    if (wv < NTR) {                                             // V = [0 0; 0 Ahh^-1] v + Pb y
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const double bop = YY[(4 * kp + lg) * LBP + jj];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kp], bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) V[(16 * wv + 4 * reg + lg) * LBP + jj] = acc[reg];
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int it = 0; it < QI; ++it) {
            const int q = l16 + 16 * it;
            if (q >= H2) continue;
            const double x0 = V[(2 * q) * LBP + sc], x1 = V[(2 * q + 1) * LBP + sc];
            const double d0 = fma(si4[it][1], x1, si4[it][0] * x0), d1 = fma(si4[it][3], x1, si4[it][2] * x0);
            *reinterpret_cast<double2*>(xs + (size_t)k * B + 2 * q) = double2{w4[it][0] - d0, w4[it][1] - d1};
        }
    }
}

template <int B>
__global__ __launch_bounds__(256) void k_sleaf_back_batch(
    Model M, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt, const double* __restrict__ wall,
    double* __restrict__ xall, const double* __restrict__ Hall, const double* __restrict__ sbimg, const double* __restrict__ lzimg,
    const double* __restrict__ Zall, const double* __restrict__ lfS, int s0) {
    sleaf_back_batch_body<B>(blockIdx.x, blockIdx.y, M, nodes, b, active, S_cnt, wall, xall, Hall, sbimg, lzimg, Zall, lfS, s0);
}

template <int B>
int launch_sleaf_back_batch(hpf_handle* h, const int* nodes, int count, const int* active) {
    const dim3 grid((unsigned)count, (unsigned)((h->cur_S + LB_SB - 1) / LB_SB));
    hipLaunchKernelGGL((k_sleaf_back_batch<B>), grid, dim3(256), 0, h->cur_stream, h->M, nodes, 2 * h->Hn, active, h->cur_S, h->d_w,
                       h->d_x, h->d_H, active_tree(h).d_sbimg, active_tree(h).d_lzimg, h->d_Z, h->d_lfS, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

// Factor sweep of the super-leaves whose parent rebuilds their Schur complement (HPF_SLLAZY), 16 scenarios per workgroup: such a
// bus owes the sweep vectors only -- w = S^-1 M^-1 Wd^-1 y, G w, T^-1 (+ W^-1), S^-1, A(k,parent), its position-0 borders.
// Roles as in k_leaf_batch (plus the power-row diagonal of a linear bus and the G w of its lazy leaves), T assembled and inverted
// per scenario by the scenario's 16 threads (thread r owns row r; the pivot row goes through LDS), the Ahh^-1 part on the matrix
// cores.  nodes: Tree::d_fdesc records (int 39: slot of the bus's image in Tree::d_sbimg).
constexpr int SLEAF_BATCH_LDS = 2 * 64 * LBP + LB_SB * 100 + LB_SB * 10 + 3 * LB_SB * 4 + 2 * 16 * LBP;      // doubles

template <int B>
__device__ __forceinline__ void sleaf_batch_body(
    double* __restrict__ smem_, const int bx_, const int by_, const Model& M, const TreeDev& T, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt,
    const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, const double* __restrict__ fall, double* __restrict__ wall,
    const double* __restrict__ linAall, double* __restrict__ Call, double* __restrict__ Hall, const cplx* __restrict__ I0all,
    const double* __restrict__ chG, const double* __restrict__ chH, const double* __restrict__ chD, const double* __restrict__ chy,
    const double* __restrict__ sbimg, double* __restrict__ Zall, double* __restrict__ lfK, double* __restrict__ lfS, int s0,
    long long* __restrict__ dbg, int ablate) {
#ifdef HPF_FACTOR_STAMPS
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define HPF_SLSTAMP(i) if (ablate & 16) tq[i] = __builtin_amdgcn_s_memtime()
#else
#define HPF_SLSTAMP(i)
#endif
    HPF_SLSTAMP(0);
    constexpr int NT = (B + 16) / 16;
    constexpr size_t CT = (size_t)NT * NT * 256;
    constexpr int NTR = SleafImg<B>::NTR, KS = SleafImg<B>::KS, KP = SleafImg<B>::KP, H2 = B / 2, QI = (H2 + 15) / 16;
    constexpr bool QBR = SleafImg<B>::QB_ROWS;
    const int4* nd = reinterpret_cast<const int4*>(nodes) + (FDESC / 4) * (size_t)bx_;
    const int4 nd0 = nd[0], nd1 = nd[1], nd3 = nd[3], lzA = nd[7], lzB = nd[8], lzC = nd[9];
    const int k = nd0.x, par = nd0.y, diag_e = nd0.z;
    const int e_dn_k = nd1.x, e_up_k = nd1.y, lin_beg = nd1.z, lin_end = nd1.z + nd1.w;
    const bool via_chain = (nd3.z & 1) != 0;
    // nested bordered children (at most two): bus, border unknowns; their T (not inverted) sits on this bus's diagonal
    const int L = lzA.y, cs0 = lzC.x, cs1 = lzC.y, mc0 = lzC.z & 0xff, mc1 = (lzC.z >> 8) & 0xff;
    const int m = 2 + 2 * L + mc0 + mc1;
    const double* simg = T.lzimg + (size_t)lzB.w;                // Tc [m][m] | Pb [b][m] | Qb [m][b]
    const double* pbm = simg + m * m;
    const double* qbm = pbm + (size_t)b * m;
    const double* img = sbimg + (size_t)lzC.w * SleafImg<B>::SZ;   // [0 0; 0 Ahh^-1] (+ the rows of Qb), A-operand layout | Pb as an A operand
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = M.n, c = M.c, Hn = M.Hn;
    const int sc = tid >> 4, l16 = tid & 15;
    const int sl = by_ * LB_SB + sc;
    const int ss = sl < S_cnt ? (active ? active[sl + s0] : sl + s0) : -1;      // slot -> scenario (active list; -1: frozen / empty slot)
    const bool live = ss >= 0;
    const int s = live ? ss : 0;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* Cs = Call + (size_t)s * n * CT;
    const bool linear_k = k < M.m;

    double* Y = smem_;                                           // v = Wd^-1 y, [row][scenario]
    double* V = Y + 64 * LBP;
    double* AUG = V + 64 * LBP;                                // T, then T^-1, per scenario [10][10]
    double* PR = AUG + LB_SB * 100;                              // pivot row of the step
    double* DL = PR + LB_SB * 10;
    double* S0 = DL + LB_SB * 4;
    double* WI = S0 + LB_SB * 4;
    double* RR = WI + LB_SB * 4;
    double* YY = RR + 16 * LBP;

    // ---- R1. rows --------------------------------------------------------------------------------------------------------------
    {
        double e0[4] = {0.0, 0.0, 0.0, 0.0}, e1[4] = {0.0, 0.0, 0.0, 0.0}, ey[4] = {0.0, 0.0, 0.0, 0.0};
        double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0}, fy[4] = {0.0, 0.0, 0.0, 0.0};
        bool ok[4];
        cplx ukr[4], ekr[4];
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) {
            const int row = l16 + 16 * pz, q = row >> 1, tr_ = row & 1;
            ok[pz] = live && row < b && loc_valid(n, c, k, row);
            ukr[pz] = cplx{0.0, 0.0};
            ekr[pz] = cplx{0.0, 0.0};
            if (ok[pz]) {
                const size_t kq = (size_t)k * Hn + q;
                fy[pz] = fall[((size_t)s * n + k) * B + row];
                if (via_chain) {
                    const size_t o = so + kq;
                    a0[pz] = chD[o * 4 + 2 * tr_];
                    a1[pz] = chD[o * 4 + 2 * tr_ + 1];
                    ay[pz] = chy[o * 2 + tr_];
                }
                ukr[pz] = U[kq];
                ekr[pz] = E[kq];
                // G w of the lazy leaves: y -= sum
                const double g0 = lzA.z >= 0 ? Cs[(size_t)lzA.z * CT + row] : 0.0, g1 = lzA.w >= 0 ? Cs[(size_t)lzA.w * CT + row] : 0.0;
                const double g2 = lzB.x >= 0 ? Cs[(size_t)lzB.x * CT + row] : 0.0, g3 = lzB.y >= 0 ? Cs[(size_t)lzB.y * CT + row] : 0.0;
                ay[pz] -= (g0 + g1) + (g2 + g3);
                const double g4 = cs0 >= 0 ? Cs[(size_t)cs0 * CT + row] : 0.0, g5 = cs1 >= 0 ? Cs[(size_t)cs1 * CT + row] : 0.0;
                ay[pz] -= g4 + g5;                                   // ... and of the bordered children
                if (row < 2 && linear_k) {                           // power rows: the state-dependent diagonal of the fundamental
                    const Blk2 blk = blk_power_diag(M.Y[(size_t)diag_e * Hn], ukr[pz], ekr[pz], I0all[(size_t)s * n + k]);
                    a0[pz] += pick(blk, tr_, 0);
                    a1[pz] += pick(blk, tr_, 1);
                }
            }
        }
        const double* ws = wall + (size_t)s * n * B;
        const double* linA = linAall + so * 4;
        const int4* c3 = reinterpret_cast<const int4*>(T.child3);
        for (int cp = lin_beg; cp < lin_end; ++cp) {
            const int4 cr = c3[cp];
            const int ch = cr.x;
#pragma unroll
            for (int pz = 0; pz < 4; ++pz) {
                if (!ok[pz]) continue;
                const int row = l16 + 16 * pz, q = row >> 1, tr_ = row & 1;
                const cplx uk = ukr[pz], ek = ekr[pz];
                const cplx ydn = M.Y[(size_t)cr.y * Hn + q], yup = M.Y[(size_t)cr.z * Hn + q];
                const cplx uc = U[(size_t)ch * Hn + q], ec = E[(size_t)ch * Hn + q];
                const double2* pic = reinterpret_cast<const double2*>(linA + ((size_t)ch * Hn + q) * 4);
                const double2 ic01 = pic[0], ic23 = pic[1];
                const double2 wc = *reinterpret_cast<const double2*>(ws + (size_t)ch * B + 2 * q);
                const Blk2 g = (q == 0 && k < M.m) ? blk_power_off(ydn, uk, uc, ec) : blk_current(ydn, uc, ec);
                const Blk2 hb = (q == 0 && ch < M.m) ? blk_power_off(yup, uc, uk, ek) : blk_current(yup, uk, ek);
                const double g0 = pick(g, tr_, 0);
                const double g1 = loc_valid(n, c, ch, 2 * q + 1) ? pick(g, tr_, 1) : 0.0;
                double h4[4];
                mask_block(n, c, q, ch, k, hb, h4);
                const double v0 = fma(g1, ic23.x, g0 * ic01.x), v1 = fma(g1, ic23.y, g0 * ic01.y);
                e0[pz] += fma(v1, h4[2], v0 * h4[0]);
                e1[pz] += fma(v1, h4[3], v0 * h4[1]);
                ey[pz] = fma(g0, wc.x, ey[pz]);
                ey[pz] = fma(g1, wc.y, ey[pz]);
            }
        }
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) {
            const int row = l16 + 16 * pz;
            Y[row * LBP + sc] = ok[pz] ? fy[pz] + ay[pz] - ey[pz] : 0.0;
            if (row < 2) {
                DL[sc * 4 + row * 2] = ok[pz] ? a0[pz] - e0[pz] : 0.0;
                DL[sc * 4 + row * 2 + 1] = ok[pz] ? a1[pz] - e1[pz] : 0.0;
            }
        }
    }
    HPF_SLSTAMP(1);
    // ---- R2. harmonics ---------------------------------------------------------------------------------------------------------
    double sir[QI][4], glr[QI][4];
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int q = l16 + 16 * it;
        double si[4] = {1.0, 0.0, 0.0, 1.0}, g4[4] = {0.0, 0.0, 0.0, 0.0}, h4[4] = {0.0, 0.0, 0.0, 0.0};
        if (live && q < Hn) {
            const cplx uk = U[(size_t)k * Hn + q], ek = E[(size_t)k * Hn + q];
            const double idet = 1.0 / (-(uk.im * ek.im) - ek.re * uk.re);
            si[0] = ek.im * idet;   si[1] = -ek.re * idet;
            si[2] = -uk.re * idet;  si[3] = -uk.im * idet;
            cplx up = {0.0, -1.0}, ep = {0.0, 1.0};
            if (!via_chain || q == 0) {
                up = U[(size_t)par * Hn + q];
                ep = E[(size_t)par * Hn + q];
            }
            if (via_chain) {
                const double2* pg = reinterpret_cast<const double2*>(chG + (so + (size_t)k * Hn + q) * 4);
                const double2* ph = reinterpret_cast<const double2*>(chH + (so + (size_t)k * Hn + q) * 4);
                const double2 ga = pg[0], gb = pg[1], ha = ph[0], hb = ph[1];
                g4[0] = ga.x; g4[1] = ga.y; g4[2] = gb.x; g4[3] = gb.y;
                h4[0] = ha.x; h4[1] = ha.y; h4[2] = hb.x; h4[3] = hb.y;
            } else {
                const cplx ydn = M.Y[(size_t)e_dn_k * Hn + q], yup = M.Y[(size_t)e_up_k * Hn + q];
                const Blk2 g = (q == 0 && par < M.m) ? blk_power_off(ydn, up, uk, ek) : blk_current(ydn, uk, ek);
                const Blk2 hh = (q == 0 && k < M.m) ? blk_power_off(yup, uk, up, ep) : blk_current(yup, up, ep);
                mask_block(n, c, q, par, k, g, g4);
                mask_block(n, c, q, k, par, hh, h4);
            }
            double* Hk = Hall + ((size_t)s * n + k) * Hn * 4 + q * 4;
            double* Sk = lfS + (so + (size_t)k * Hn + q) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Hk[e] = h4[e];
                Sk[e] = si[e];
            }
            if (q == 0) {
                const double ip = 1.0 / (-(up.im * ep.im) - ep.re * up.re);
                const double sp0 = ep.im * ip, sp1 = -ep.re * ip, sp2 = -up.re * ip, sp3 = -up.im * ip;
                double* kk = lfK + ((size_t)s * n + k) * 12 + 4;
                kk[0] = fma(g4[1], si[2], g4[0] * si[0]);  kk[1] = fma(g4[1], si[3], g4[0] * si[1]);
                kk[2] = fma(g4[3], si[2], g4[2] * si[0]);  kk[3] = fma(g4[3], si[3], g4[2] * si[1]);
                kk[4] = fma(h4[1], sp2, h4[0] * sp0);  kk[5] = fma(h4[1], sp3, h4[0] * sp1);
                kk[6] = fma(h4[3], sp2, h4[2] * sp0);  kk[7] = fma(h4[3], sp3, h4[2] * sp1);
                double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;      // W_k^-1 = W_k / |U|^2 for a linear bus
                if (linear_k) {
                    const double iu = 1.0 / fma(uk.re, uk.re, uk.im * uk.im);
                    w00 = uk.re * iu; w01 = uk.im * iu; w10 = w01; w11 = -w00;
                }
                WI[sc * 4] = w00; WI[sc * 4 + 1] = w01; WI[sc * 4 + 2] = w10; WI[sc * 4 + 3] = w11;
                double* tk = Zall + ((size_t)s * n + k) * CT + 100;
                tk[0] = w00; tk[1] = w01; tk[2] = w10; tk[3] = w11;
            }
        }
        if (q == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) S0[sc * 4 + e] = si[e];
            if (!(live && q < Hn)) { WI[sc * 4] = 1.0; WI[sc * 4 + 1] = 0.0; WI[sc * 4 + 2] = 0.0; WI[sc * 4 + 3] = 1.0; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sir[it][e] = si[e];
            glr[it][e] = g4[e];
        }
    }
    // T <- Tc (per scenario copy), rows by the scenario's threads
    if (l16 < m)
        for (int c2 = 0; c2 < m; ++c2) AUG[sc * 100 + l16 * 10 + c2] = simg[l16 * m + c2];
    HPF_SLSTAMP(2);
    __syncthreads();
    HPF_SLSTAMP(3);
    // ---- T. per-scenario parts of T (as in k_factor_q's super-leaf branch), v = Wd^-1 y -----------------------------------------
    {
        double* aug = AUG + sc * 100;
        const double* wi = WI + sc * 4;
        if (l16 == 0) {
            const double* si = S0 + sc * 4;
            const double* dl = DL + sc * 4;
            const double e00 = fma(dl[1], si[2], dl[0] * si[0]), e01 = fma(dl[1], si[3], dl[0] * si[1]);
            const double e10 = fma(dl[3], si[2], dl[2] * si[0]), e11 = fma(dl[3], si[3], dl[2] * si[1]);
            aug[0] += fma(wi[1], e10, wi[0] * e00);
            aug[1] += fma(wi[1], e11, wi[0] * e01);
            aug[10] += fma(wi[3], e10, wi[2] * e00);
            aug[11] += fma(wi[3], e11, wi[2] * e01);
            const double y0 = Y[sc], y1 = Y[LBP + sc];                  // right-hand side rows 0 / 1: W_k^-1
            Y[sc] = fma(wi[1], y1, wi[0] * y0);
            Y[LBP + sc] = fma(wi[3], y1, wi[2] * y0);
        } else if (l16 <= L && live) {
            const int i = l16 - 1, bc = 2 + 2 * i;
            const int leaf = i == 0 ? lzA.z : (i == 1 ? lzA.w : (i == 2 ? lzB.x : lzB.y));
            const double* kk = lfK + ((size_t)s * n + leaf) * 12;
            double q00, q01, q10, q11;
            inv2(kk[0], kk[1], kk[2], kk[3], q00, q01, q10, q11);
            aug[bc * 10 + bc] += q00;
            aug[bc * 10 + bc + 1] += q01;
            aug[(bc + 1) * 10 + bc] += q10;
            aug[(bc + 1) * 10 + bc + 1] += q11;
            aug[bc] -= fma(wi[1], kk[6], wi[0] * kk[4]);
            aug[bc + 1] -= fma(wi[1], kk[7], wi[0] * kk[5]);
            aug[10 + bc] -= fma(wi[3], kk[6], wi[2] * kk[4]);
            aug[10 + bc + 1] -= fma(wi[3], kk[7], wi[2] * kk[5]);
            aug[bc * 10] -= kk[8];
            aug[bc * 10 + 1] -= kk[9];
            aug[(bc + 1) * 10] -= kk[10];
            aug[(bc + 1) * 10 + 1] -= kk[11];
        } else if (live && l16 - L - 1 < (cs0 >= 0) + (cs1 >= 0)) {
            // bordered child: its T_c in the place of K_c^-1; of its border columns only the first (the child's own position 0)
            // reaches position 0 here: -W_k^-1 (G0 S_c^-1) above, -W_c^-1 (H0 S_k^-1) on the left
            const int zc = l16 - L - 1;
            const int cb = zc == 0 ? cs0 : cs1, mc = zc == 0 ? mc0 : mc1, bc = 2 + 2 * L + (zc ? mc0 : 0);
            const double* tcd = Zall + ((size_t)s * n + cb) * CT;     // T_c^-1 [10][10] | W_c^-1 [4] | T_c [10][10]
            const double* kk = lfK + ((size_t)s * n + cb) * 12;
            for (int i = 0; i < mc; ++i)
                for (int j = 0; j < mc; ++j) aug[(bc + i) * 10 + bc + j] += tcd[104 + i * 10 + j];
            aug[bc] -= fma(wi[1], kk[6], wi[0] * kk[4]);
            aug[bc + 1] -= fma(wi[1], kk[7], wi[0] * kk[5]);
            aug[10 + bc] -= fma(wi[3], kk[6], wi[2] * kk[4]);
            aug[10 + bc + 1] -= fma(wi[3], kk[7], wi[2] * kk[5]);
            const double* wc = tcd + 100;
            aug[bc * 10] -= fma(wc[1], kk[10], wc[0] * kk[8]);
            aug[bc * 10 + 1] -= fma(wc[1], kk[11], wc[0] * kk[9]);
            aug[(bc + 1) * 10] -= fma(wc[3], kk[10], wc[2] * kk[8]);
            aug[(bc + 1) * 10 + 1] -= fma(wc[3], kk[11], wc[2] * kk[9]);
        }
    }
    __syncthreads();
    HPF_SLSTAMP(4);
    // ---- I. in-place inversion with partial pivoting, one scenario per 16 threads (thread r = row r; pivot row through LDS) -----
    {
        double* aug = AUG + sc * 100;
        double* pr = PR + sc * 10;
        double row[10];
#pragma unroll
        for (int c2 = 0; c2 < 10; ++c2) row[c2] = (l16 < m && c2 < m) ? aug[l16 * 10 + c2] : 0.0;
        if (live && l16 < m) {                                   // T itself for a bordered parent (behind T^-1 and W^-1)
            double* tk = Zall + ((size_t)s * n + k) * CT + 104 + l16 * 10;
#pragma unroll
            for (int c2 = 0; c2 < 10; ++c2)
                if (c2 < m) tk[c2] = row[c2];
        }
        bool used = l16 >= m;
        int mycol = 0, pcol[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            pcol[j] = 0;
            if (j < m) {
                const double cand = used ? -1.0 : fabs(row[j]);
                double mx = cand;
                mx = fmax(mx, dpp_f64<0xB1>(mx));
                mx = fmax(mx, dpp_f64<0x4E>(mx));
                mx = fmax(mx, dpp_f64<0x141>(mx));
                mx = fmax(mx, dpp_f64<0x140>(mx));
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(!used && cand == mx);
                const unsigned grp = (unsigned)((bal >> (16 * lg)) & 0xffffull);
                const int pi = __builtin_ctz(grp | 0x8000u);
                const bool me = l16 == pi;
                if (me) {
#pragma unroll
                    for (int c2 = 0; c2 < 10; ++c2) pr[c2] = row[c2];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double prow[10];
#pragma unroll
                for (int c2 = 0; c2 < 10; ++c2) prow[c2] = pr[c2];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const double ipv = 1.0 / prow[j];
                const double f = me ? 0.0 : row[j] * ipv;
#pragma unroll
                for (int c2 = 0; c2 < 10; ++c2)
                    row[c2] = c2 == j ? (me ? ipv : -f) : (me ? prow[c2] * ipv : fma(-f, prow[c2], row[c2]));
                used = used || me;
                mycol = me ? j : mycol;
                pcol[j] = pi;
            }
        }
        if (l16 < m) {
            double* tk = Zall + ((size_t)s * n + k) * CT;
#pragma unroll
            for (int c2 = 0; c2 < 10; ++c2)
                if (c2 < m) {
                    aug[mycol * 10 + pcol[c2]] = row[c2];
                    if (live) tk[mycol * 10 + pcol[c2]] = row[c2];
                }
        }
    }
    __syncthreads();
    HPF_SLSTAMP(5);
    // ---- M. V = [0 0; 0 Ahh^-1] v on the matrix cores; r = Qb v (rows B.. of the same MFMAs where they fit) ------------------------
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (wv < NTR) {
        const double* ia = img + (size_t)wv * KS * 64 + lane;
#pragma unroll 4
        for (int ks = 0; ks < KS; ++ks) {
            const double a = ia[(size_t)ks * 64];
            const double bop = Y[(4 * ks + lg) * LBP + jj];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop, acc, 0, 0, 0);
        }
        if (QBR && wv == NTR - 1) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = 16 * (NTR - 1) + 4 * reg + lg - B;
                if (j >= 0 && j < 16) RR[j * LBP + jj] = acc[reg];
            }
        }
    }
    if (!QBR) {
        double r = 0.0;
        if (l16 < m) {
            const double* qr = qbm + (size_t)l16 * b;
            for (int col = 0; col < b; ++col) r = fma(qr[col], Y[col * LBP + sc], r);
        }
        RR[l16 * LBP + sc] = r;
    }
    __syncthreads();
    {
        double y = 0.0;
        if (l16 < m) {
            const double* tr = AUG + sc * 100 + l16 * 10;
            for (int j = 0; j < m; ++j) y = fma(tr[j], RR[j * LBP + sc], y);
        }
        YY[l16 * LBP + sc] = y;
    }
    __syncthreads();
    if (wv < NTR) {                                              // ... + Pb y: three more rank-4 steps on the same accumulators
        const double* pa = img + SleafImg<B>::MAIN + (size_t)wv * KP * 64 + lane;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const double a = pa[(size_t)kp * 64];
            const double bop = YY[(4 * kp + lg) * LBP + jj];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) V[(16 * wv + 4 * reg + lg) * LBP + jj] = acc[reg];
    }
    __syncthreads();
    HPF_SLSTAMP(6);
    // ---- F. x_rect = V + Pb y, w = S^-1 x_rect, G w --------------------------------------------------------------------------
    if (live) {
        double* wk = wall + ((size_t)s * n + k) * B;
        double* Ck = Call + ((size_t)s * n + k) * CT;
#pragma unroll
        for (int it = 0; it < QI; ++it) {
            const int q = l16 + 16 * it;
            if (q >= H2) continue;
            const double x0 = V[(2 * q) * LBP + sc], x1 = V[(2 * q + 1) * LBP + sc];
            const double* si = sir[it];
            const bool in = 2 * q < b;
            const double w0 = in ? fma(si[1], x1, si[0] * x0) : 0.0, w1 = in ? fma(si[3], x1, si[2] * x0) : 0.0;
            const double* g = glr[it];
            wk[2 * q] = w0;
            wk[2 * q + 1] = w1;
            Ck[2 * q] = fma(g[1], w1, g[0] * w0);
            Ck[2 * q + 1] = fma(g[3], w1, g[2] * w0);
        }
    }
#ifdef HPF_FACTOR_STAMPS
    if ((ablate & 16) && dbg && live && l16 == 0) {     // phases of this scenario's 16 threads (cycles): rows | harmonics | wait | T | inversion | MFMA + Qb v | T^-1 r | tail
        HPF_SLSTAMP(7);
        long long* o = dbg + ((size_t)s * n + k) * 8;
        o[0] = tq[1] - tq[0];
        o[1] = tq[2] - tq[1];
        o[2] = tq[3] - tq[2];
        o[3] = ((tq[4] - tq[3]) & 0xfffff) | (((tq[5] - tq[4]) & 0xfffff) << 20) | (((tq[6] - tq[5]) & 0xfffff) << 40);
        o[4] = tq[7] - tq[6];
        o[5] = tq[7] - tq[0];
        o[6] = 1000 + m;
        o[7] = (k >= M.m) | ((lin_end - lin_beg) << 1);
    }
#endif
#undef HPF_SLSTAMP
}

template <int B>
__global__ __launch_bounds__(256, 4) void k_sleaf_batch(
    Model M, TreeDev T, const int* __restrict__ nodes, int b, const int* __restrict__ active, int S_cnt,
    const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, const double* __restrict__ fall, double* __restrict__ wall,
    const double* __restrict__ linAall, double* __restrict__ Call, double* __restrict__ Hall, const cplx* __restrict__ I0all,
    const double* __restrict__ chG, const double* __restrict__ chH, const double* __restrict__ chD, const double* __restrict__ chy,
    const double* __restrict__ sbimg, double* __restrict__ Zall, double* __restrict__ lfK, double* __restrict__ lfS, int s0,
    long long* __restrict__ dbg, int ablate) {
    __shared__ __attribute__((aligned(16))) double smem[SLEAF_BATCH_LDS];
    sleaf_batch_body<B>(smem, blockIdx.x, blockIdx.y, M, T, nodes, b, active, S_cnt, Uall, Eall, fall, wall, linAall, Call, Hall, I0all, chG, chH, chD,
                        chy, sbimg, Zall, lfK, lfS, s0, dbg, ablate);
}

template <int B>
int launch_sleaf_batch(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    ScopedTimer t(h, T_SOLVE);
    const dim3 grid((unsigned)count, (unsigned)((h->cur_S + LB_SB - 1) / LB_SB));
    hipLaunchKernelGGL((k_sleaf_batch<B>), grid, dim3(256), 0, h->cur_stream, h->M, T, nodes, 2 * h->Hn, active, h->cur_S, h->d_U,
                       h->d_E, h->d_fb, h->d_w, h->d_linA, h->d_C, h->d_H, h->d_I0, h->d_chG, h->d_chH, h->d_chD, h->d_chy,
                       active_tree(h).d_sbimg, h->d_Z, h->d_lfK, h->d_lfS, h->cur_s0, h->d_dbg, h->debug_ablate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}
