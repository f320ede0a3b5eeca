// The 2x2-per-harmonic algebra of the contracted tree: linear subtrees, contracted chains and their bundles (k_lin_level_*, k_chain_*, k_lin_bundle_*,
// k_lin_tree_*).  Included by hpf_block.hip right after hpf_quad.hpp (same translation unit; split off in round 4).
#pragma once

// =============================================================================================================
// Level-parallel kernels of the 2x2-per-harmonic algebra (contracted tree).  The all-linear subtrees are at most a few buses
// deep: instead of one thread walking a whole subtree (k_lin_factor: every bus costs a chain of dependent loads), the buses
// are grouped by their height inside the subtree and one launch handles one height, one thread per (bus, harmonic,
// scenario), with every operand address coming from an 8-int record and all loads of a thread issued before the arithmetic.
// =============================================================================================================
struct Rec8 {
    int k, diag_e, parent, e_up, e_dn, cbeg, nch, pad;
};

// un-eliminated 2x2 diagonal block of a bus k of the 2x2 algebra at harmonic position q with identity padding (diag2x2), value form
__device__ __forceinline__ void diag2x2_val(int n, int c, int m, int q, int k, cplx yd, cplx uk, cplx ek, cplx I0v, cplx yn,
                                            double m2[4]) {
    // (k >= m only with uncoupled Norton data, where nonlinear buses stay in the 2x2 algebra: diagonal Norton term HG:442-443)
    const Blk2 blk = (q == 0 && k < m) ? blk_power_diag(yd, uk, ek, I0v) : blk_current_diag(yd, uk, ek, yn, k >= m);
    const bool v0 = loc_valid(n, c, k, 2 * q), v1 = loc_valid(n, c, k, 2 * q + 1);
    m2[0] = v0 ? blk.dA.re : 1.0;
    m2[1] = (v0 && v1) ? blk.dV.re : 0.0;
    m2[2] = (v0 && v1) ? blk.dA.im : 0.0;
    m2[3] = v1 ? blk.dV.im : 1.0;
}

// masked coupling block A(row bus i, column bus j) at q from values: y = Y_q[i,j], Ui (row bus, used by power rows), Uj / Ej
__device__ __forceinline__ void coupling_val(int n, int c, int m, int q, int i, int j, cplx y, cplx Ui, cplx Uj, cplx Ej, double out[4]) {
    const Blk2 blk = (q == 0 && i < m) ? blk_power_off(y, Ui, Uj, Ej) : blk_current(y, Uj, Ej);
    mask_block(n, c, q, i, j, blk, out);
}

// fold the 2x2-algebra children [cbeg, cbeg+nch) of bus k into (m2, y0, y1): m2 -= A(k,ch) D_ch^-1 A(ch,k), y -= A(k,ch) w_ch
__device__ __forceinline__ void fold_children(const Model& M, const TreeDev& T, const cplx* U, const cplx* E, const double* linA,
                                              const double* ws, int Bst, int q, int k, cplx uk, cplx ek, int cbeg, int nch,
                                              double m2[4], double& y0, double& y1, int m_eff) {
    const int n = M.n, c = M.c, Hn = M.Hn;
    const int4* c3 = reinterpret_cast<const int4*>(T.child3);
    for (int j = 0; j < nch; ++j) {
        const int4 cr = c3[cbeg + j];
        const int ch = cr.x;
        const cplx ydn = M.Y[(size_t)cr.y * Hn + q], yup = M.Y[(size_t)cr.z * Hn + q];
        const cplx uc = U[(size_t)ch * Hn + q], ec = E[(size_t)ch * Hn + q];
        const double2* pic = reinterpret_cast<const double2*>(linA + ((size_t)ch * Hn + q) * 4);
        const double2 ic01 = pic[0], ic23 = pic[1];
        const double2 wc = *reinterpret_cast<const double2*>(ws + (size_t)ch * Bst + 2 * q);
        __builtin_amdgcn_sched_barrier(0);
        double g4[4], h4[4], gi[4], gh[4];
        coupling_val(n, c, m_eff, q, k, ch, ydn, uk, uc, ec, g4);            // A(k, child)
        coupling_val(n, c, m_eff, q, ch, k, yup, uc, uk, ek, h4);            // A(child, k)
        const double ic[4] = {ic01.x, ic01.y, ic23.x, ic23.y};
        mul22(g4, ic, gi);
        mul22(gi, h4, gh);
#pragma unroll
        for (int e = 0; e < 4; ++e) m2[e] -= gh[e];
        y0 -= fma(g4[1], wc.y, g4[0] * wc.x);
        y1 -= fma(g4[3], wc.y, g4[2] * wc.x);
    }
}

// one (bus record, harmonic position, scenario) of the factor sweep of the 2x2 algebra: D_k^-1 and w_k = D_k^-1 y_k with the
// children folded in.  fund: fundamental power flow (HG:205-223) -- harmonic position 0 only, every bus a power row
// (m_eff = n), mismatch in the stacked order of `pf`; the records then cover the whole tree
__device__ __forceinline__ void lin_factor_item(const Model& M, const TreeDev& T, const int* __restrict__ rec, int pos, int q, int s,
                                                int N, int Nc, int Bst, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                const double* __restrict__ fall, double* linAall, double* wall,
                                                const cplx* __restrict__ I0all, int fund) {
    const int n = M.n, c = M.c, Hn = M.Hn;
    const int m_eff = fund ? n : M.m;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    double* linA = linAall + so * 4;
    double* ws = wall + (size_t)s * n * Bst;
    const int4 r0 = reinterpret_cast<const int4*>(rec)[2 * pos], r1 = reinterpret_cast<const int4*>(rec)[2 * pos + 1];
    const int k = r0.x;
    const cplx yd = M.Y[(size_t)r0.y * Hn + q];
    const cplx uk = U[(size_t)k * Hn + q], ek = E[(size_t)k * Hn + q];
    cplx I0v = {0.0, 0.0}, yn = {0.0, 0.0};
    if (q == 0 && k < m_eff) I0v = I0all[(size_t)s * n + k];
    if (k >= m_eff) yn = M.coupled ? M.YN[((size_t)r1.w * Hn + q) * Hn + q] : M.YN[(size_t)r1.w * Hn + q];
    double y0, y1;
    if (fund) {
        const double* f = fall + (size_t)s * N;
        y0 = k >= 1 ? f[k - 1] : 0.0;
        y1 = k >= c ? f[Nc + k - c] : 0.0;
    } else {
        const double2 fy = *reinterpret_cast<const double2*>(fall + ((size_t)s * n + k) * Bst + 2 * q);   // bus-major mismatch image
        y0 = fy.x;
        y1 = fy.y;
    }
    __builtin_amdgcn_sched_barrier(0);
    double m2[4];
    diag2x2_val(n, c, m_eff, q, k, yd, uk, ek, I0v, yn, m2);
    fold_children(M, T, U, E, linA, ws, Bst, q, k, uk, ek, r1.y, r1.z, m2, y0, y1, m_eff);
    double di[4];
    inv2(m2[0], m2[1], m2[2], m2[3], di[0], di[1], di[2], di[3]);
    double* ik = linA + ((size_t)k * Hn + q) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) ik[e] = di[e];
    double* wk = ws + (size_t)k * Bst + 2 * q;
    wk[0] = fma(di[1], y1, di[0] * y0);
    wk[1] = fma(di[3], y1, di[2] * y0);
}

__global__ __launch_bounds__(128) void k_lin_level_factor(Model M, TreeDev T, const int* __restrict__ rec, int count, int N, int Nc,
                                                          int Bst, const int* __restrict__ active, const cplx* __restrict__ Uall,
                                                          const cplx* __restrict__ Eall, const double* __restrict__ fall,
                                                          double* __restrict__ linAall, double* __restrict__ wall,
                                                          const cplx* __restrict__ I0all, int fund, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tix = blockIdx.x * 128 + threadIdx.x;
    const int HnE = fund ? 1 : M.Hn;
    if (tix >= count * HnE) return;
    lin_factor_item(M, T, rec, tix / HnE, tix % HnE, s, N, Nc, Bst, Uall, Eall, fall, linAall, wall, I0all, fund);
}

// The same sweep in ONE launch (harmonic Newton step): the all-linear subtrees are independent of each other, so a workgroup takes
// a bundle of WHOLE subtrees (records sorted by height inside the bundle, bptr: nh + 1 offsets per bundle) and walks the heights
// with a workgroup barrier in between -- a parent's operands (D_child^-1, w_child) were written by threads of the same
// workgroup.  Same arithmetic per (bus, harmonic) as the level kernel: bit-identical results, nh - 1 launches fewer.
__global__ __launch_bounds__(256) void k_lin_tree_factor(Model M, TreeDev T, const int* __restrict__ rec, const int* __restrict__ bptr,
                                                         int nh, int N, int Nc, int Bst, const int* __restrict__ active,
                                                         const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                         const double* __restrict__ fall, double* linAall, double* wall,
                                                         const cplx* __restrict__ I0all, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;
    if (s < 0) return;
    const int* bp = bptr + (size_t)blockIdx.x * (nh + 1);
    const int Hn = M.Hn, end = bp[nh];
    for (int hh = 0; hh < nh; ++hh) {
        const int beg = bp[hh], nxt = bp[hh + 1];
        for (int it = threadIdx.x; it < (nxt - beg) * Hn; it += 256)
            lin_factor_item(M, T, rec, beg + it / Hn, it % Hn, s, N, Nc, Bst, Uall, Eall, fall, linAall, wall, I0all, 0);
        if (nxt == end) break;                                   // (uniform: nothing of this bundle above this height)
        __syncthreads();
    }
}

// one (bus record, harmonic position, scenario) of the back sweep of the 2x2 algebra: x_k = w_k - D_k^-1 A(k, parent) x_parent
__device__ __forceinline__ void lin_back_item(const Model& M, const TreeDev& T, const int* __restrict__ rec, int pos, int q, int s, int N,
                                              int Nc, int Bst, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                              const double* __restrict__ linAall, const double* __restrict__ wall, double* xall,
                                              double* step, int fund) {
    const int n = M.n, c = M.c, Hn = M.Hn;
    const int m_eff = fund ? n : M.m;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* linA = linAall + so * 4;
    const double* ws = wall + (size_t)s * n * Bst;
    double* xs = xall + (size_t)s * n * Bst;
    double* st = step + (size_t)s * N;
    const int4 r0 = reinterpret_cast<const int4*>(rec)[2 * pos], r1 = reinterpret_cast<const int4*>(rec)[2 * pos + 1];
    const int k = r0.x, par = r0.z;
    const double2 wk = *reinterpret_cast<const double2*>(ws + (size_t)k * Bst + 2 * q);
    double x0 = wk.x, x1 = wk.y;
    if (par >= 0) {
        const cplx yup = M.Y[(size_t)r0.w * Hn + q];
        const cplx uk = U[(size_t)k * Hn + q];
        const cplx up = U[(size_t)par * Hn + q], ep = E[(size_t)par * Hn + q];
        const double2 xp = *reinterpret_cast<const double2*>(xs + (size_t)par * Bst + 2 * q);
        const double2* pik = reinterpret_cast<const double2*>(linA + ((size_t)k * Hn + q) * 4);
        const double2 i01 = pik[0], i23 = pik[1];
        __builtin_amdgcn_sched_barrier(0);
        double h4[4];
        coupling_val(n, c, m_eff, q, k, par, yup, uk, up, ep, h4);           // A(k, parent)
        const double t0 = fma(h4[1], xp.y, h4[0] * xp.x), t1 = fma(h4[3], xp.y, h4[2] * xp.x);
        x0 -= fma(i01.y, t1, i01.x * t0);
        x1 -= fma(i23.y, t1, i23.x * t0);
    }
    (void)r1;
    double* xk = xs + (size_t)k * Bst + 2 * q;
    xk[0] = x0;
    xk[1] = x1;
    if (step) {
        const int kst = q * n + k;
        if (kst >= 1) st[kst - 1] = x0;
        if (kst >= c) st[Nc + kst - c] = x1;
    }
}

__global__ __launch_bounds__(128) void k_lin_level_back(Model M, TreeDev T, const int* __restrict__ rec, int count, int N, int Nc,
                                                        int Bst, const int* __restrict__ active, const cplx* __restrict__ Uall,
                                                        const cplx* __restrict__ Eall, const double* __restrict__ linAall,
                                                        const double* __restrict__ wall, double* __restrict__ xall,
                                                        double* __restrict__ step, int fund, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tix = blockIdx.x * 128 + threadIdx.x;
    const int HnE = fund ? 1 : M.Hn;
    if (tix >= count * HnE) return;
    lin_back_item(M, T, rec, tix / HnE, tix % HnE, s, N, Nc, Bst, Uall, Eall, linAall, wall, xall, step, fund);
}

// ... and the back sweep in one launch: the same bundles, heights top-down (x of a subtree's root needs x of its dense / chain
// parent, complete before this launch; below it every x_parent comes from the workgroup itself)
__global__ __launch_bounds__(256) void k_lin_tree_back(Model M, TreeDev T, const int* __restrict__ rec, const int* __restrict__ bptr, int nh,
                                                       int N, int Nc, int Bst, const int* __restrict__ active,
                                                       const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                       const double* __restrict__ linAall, const double* __restrict__ wall,
                                                       double* xall, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;
    if (s < 0) return;
    const int* bp = bptr + (size_t)blockIdx.x * (nh + 1);
    const int Hn = M.Hn, end = bp[nh];
    bool first = true;
    for (int hh = nh - 1; hh >= 0; --hh) {
        const int beg = bp[hh], nxt = bp[hh + 1];
        if (beg == end) continue;                                // (uniform: the bundle is lower than this height)
        if (!first) __syncthreads();
        first = false;
        for (int it = threadIdx.x; it < (nxt - beg) * Hn; it += 256)
            lin_back_item(M, T, rec, beg + it / Hn, it % Hn, s, N, Nc, Bst, Uall, Eall, linAall, wall, xall, nullptr, 0);
    }
}

// Contracted chains (see k_chain_factor for the algebra): same elimination, operands of a chain bus loaded in one batch.
__device__ __forceinline__ void chain_factor_item(const Model& M, const TreeDev& T, const int* __restrict__ crec, const int* __restrict__ cnode,
                                                  int r, int q, int s, int Bst, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                  const double* __restrict__ fall, double* linAall, double* wall,
                                                  const cplx* __restrict__ I0all, double* __restrict__ chG, double* __restrict__ chH,
                                                  double* __restrict__ chD, double* __restrict__ chy, double* __restrict__ chZ) {
    const int n = M.n, c = M.c, Hn = M.Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    double* linA = linAall + so * 4;
    double* ws = wall + (size_t)s * n * Bst;
    const int4 h0 = reinterpret_cast<const int4*>(crec)[2 * r], h1 = reinterpret_cast<const int4*>(crec)[2 * r + 1];
    const int ch = h0.x, beg = h0.w, len = h1.x;
    const cplx uch = U[(size_t)ch * Hn + q], ech = E[(size_t)ch * Hn + q];
    const cplx y_kc = M.Y[(size_t)h0.y * Hn + q], y_ck = M.Y[(size_t)h0.z * Hn + q];
    double a_kc[4], a_ck[4];
    double dD[4] = {0.0, 0.0, 0.0, 0.0}, dy[2] = {0.0, 0.0}, cD[4] = {0.0, 0.0, 0.0, 0.0}, cy[2] = {0.0, 0.0};
    for (int idx = 0; idx < len; ++idx) {
        const int4 r0 = reinterpret_cast<const int4*>(cnode)[2 * (beg + idx)], r1 = reinterpret_cast<const int4*>(cnode)[2 * (beg + idx) + 1];
        const int k = r0.x, up = r0.z;
        const cplx yd = M.Y[(size_t)r0.y * Hn + q];
        const cplx uk = U[(size_t)k * Hn + q], ek = E[(size_t)k * Hn + q];
        const cplx uu = U[(size_t)up * Hn + q], eu = E[(size_t)up * Hn + q];
        const cplx y_ku = M.Y[(size_t)r0.w * Hn + q], y_uk = M.Y[(size_t)r1.x * Hn + q];
        cplx I0v = {0.0, 0.0};
        if (q == 0 && k < M.m) I0v = I0all[(size_t)s * n + k];
        const double2 fy = *reinterpret_cast<const double2*>(fall + ((size_t)s * n + k) * Bst + 2 * q);
        double y0 = fy.x + cy[0];
        double y1 = fy.y + cy[1];
        __builtin_amdgcn_sched_barrier(0);
        if (idx == 0) {
            coupling_val(n, c, M.m, q, k, ch, y_kc, uk, uch, ech, a_kc);     // A(k1, ch)
            coupling_val(n, c, M.m, q, ch, k, y_ck, uch, uk, ek, a_ck);      // A(ch, k1)
        }
        double m2[4];
        diag2x2_val(n, c, M.m, q, k, yd, uk, ek, I0v, cplx{0.0, 0.0}, m2);      // chain buses are linear buses
#pragma unroll
        for (int e = 0; e < 4; ++e) m2[e] += cD[e];
        fold_children(M, T, U, E, linA, ws, Bst, q, k, uk, ek, r1.y, r1.z, m2, y0, y1, M.m);
        double di[4];
        inv2(m2[0], m2[1], m2[2], m2[3], di[0], di[1], di[2], di[3]);
        double* ik = linA + ((size_t)k * Hn + q) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) ik[e] = di[e];
        const double w0 = fma(di[1], y1, di[0] * y0), w1 = fma(di[3], y1, di[2] * y0);
        double* wk = ws + (size_t)k * Bst + 2 * q;
        wk[0] = w0;
        wk[1] = w1;
        double a_ku[4], a_uk[4], zc[4], zu[4], t4[4];
        coupling_val(n, c, M.m, q, k, up, y_ku, uk, uu, eu, a_ku);           // A(k, up)
        coupling_val(n, c, M.m, q, up, k, y_uk, uu, uk, ek, a_uk);           // A(up, k)
        mul22(di, a_kc, zc);
        mul22(di, a_ku, zu);
        double* zk = chZ + (so + (size_t)k * Hn + q) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) zk[e] = zc[e];
        mul22(a_ck, zc, t4);
#pragma unroll
        for (int e = 0; e < 4; ++e) dD[e] -= t4[e];
        dy[0] -= fma(a_ck[1], w1, a_ck[0] * w0);
        dy[1] -= fma(a_ck[3], w1, a_ck[2] * w0);
        double n_ck[4], n_kc[4];
        mul22(a_ck, zu, n_ck);
        mul22(a_uk, zc, n_kc);
        mul22(a_uk, zu, t4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cD[e] = -t4[e];
            a_ck[e] = -n_ck[e];
            a_kc[e] = -n_kc[e];
        }
        cy[0] = -fma(a_uk[1], w1, a_uk[0] * w0);
        cy[1] = -fma(a_uk[3], w1, a_uk[2] * w0);
    }
    const size_t o = (so + (size_t)ch * Hn + q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        chG[o * 4 + e] = a_kc[e];
        chH[o * 4 + e] = a_ck[e];
        chD[o * 4 + e] = dD[e];
    }
    chy[o * 2 + 0] = dy[0];
    chy[o * 2 + 1] = dy[1];
}

__global__ __launch_bounds__(128) void k_chain_factor2(Model M, TreeDev T, const int* __restrict__ crec, const int* __restrict__ cnode,
                                                       int nchains, int N, int Nc, int Bst, const int* __restrict__ active,
                                                       const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                       const double* __restrict__ fall, double* __restrict__ linAall,
                                                       double* __restrict__ wall, const cplx* __restrict__ I0all,
                                                       double* __restrict__ chG, double* __restrict__ chH, double* __restrict__ chD,
                                                       double* __restrict__ chy, double* __restrict__ chZ, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tix = blockIdx.x * 128 + threadIdx.x;
    if (tix >= nchains * M.Hn) return;
    chain_factor_item(M, T, crec, cnode, tix / M.Hn, tix % M.Hn, s, Bst, Uall, Eall, fall, linAall, wall, I0all, chG, chH, chD, chy, chZ);
}

__device__ __forceinline__ void chain_back_item(const Model& M, const int* __restrict__ crec, const int* __restrict__ cnode, int r, int q, int s,
                                                int N, int Nc, int Bst, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                const double* __restrict__ linAall, const double* __restrict__ wall, double* xall,
                                                double* step, const double* __restrict__ chZ) {
    const int n = M.n, c = M.c, Hn = M.Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* linA = linAall + so * 4;
    const double* ws = wall + (size_t)s * n * Bst;
    double* xs = xall + (size_t)s * n * Bst;
    double* st = step + (size_t)s * N;
    const int4 h0 = reinterpret_cast<const int4*>(crec)[2 * r], h1 = reinterpret_cast<const int4*>(crec)[2 * r + 1];
    const int ch = h0.x, beg = h0.w, len = h1.x;
    const double2 xc = *reinterpret_cast<const double2*>(xs + (size_t)ch * Bst + 2 * q);
    for (int idx = len - 1; idx >= 0; --idx) {
        const int4 r0 = reinterpret_cast<const int4*>(cnode)[2 * (beg + idx)];
        const int k = r0.x, up = r0.z;
        const cplx y_ku = M.Y[(size_t)r0.w * Hn + q];
        const cplx uk = U[(size_t)k * Hn + q];
        const cplx uu = U[(size_t)up * Hn + q], eu = E[(size_t)up * Hn + q];
        const double2 xp = *reinterpret_cast<const double2*>(xs + (size_t)up * Bst + 2 * q);
        const double2* pik = reinterpret_cast<const double2*>(linA + ((size_t)k * Hn + q) * 4);
        const double2 i01 = pik[0], i23 = pik[1];
        const double2* pzk = reinterpret_cast<const double2*>(chZ + (so + (size_t)k * Hn + q) * 4);
        const double2 z01 = pzk[0], z23 = pzk[1];
        const double2 wk = *reinterpret_cast<const double2*>(ws + (size_t)k * Bst + 2 * q);
        __builtin_amdgcn_sched_barrier(0);
        double h4[4];
        coupling_val(n, c, M.m, q, k, up, y_ku, uk, uu, eu, h4);             // A(k, up)
        const double t0 = fma(h4[1], xp.y, h4[0] * xp.x), t1 = fma(h4[3], xp.y, h4[2] * xp.x);
        const double x0 = wk.x - fma(i01.y, t1, i01.x * t0) - fma(z01.y, xc.y, z01.x * xc.x);
        const double x1 = wk.y - fma(i23.y, t1, i23.x * t0) - fma(z23.y, xc.y, z23.x * xc.x);
        double* xk = xs + (size_t)k * Bst + 2 * q;
        xk[0] = x0;
        xk[1] = x1;
        if (step) {
            const int kst = q * n + k;
            if (kst >= 1) st[kst - 1] = x0;
            if (kst >= c) st[Nc + kst - c] = x1;
        }
    }
}

__global__ __launch_bounds__(128) void k_chain_back2(Model M, TreeDev T, const int* __restrict__ crec, const int* __restrict__ cnode,
                                                     int nchains, int N, int Nc, int Bst, const int* __restrict__ active,
                                                     const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                     const double* __restrict__ linAall, const double* __restrict__ wall,
                                                     double* __restrict__ xall, double* __restrict__ step,
                                                     const double* __restrict__ chZ, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tix = blockIdx.x * 128 + threadIdx.x;
    if (tix >= nchains * M.Hn) return;
    chain_back_item(M, crec, cnode, tix / M.Hn, tix % M.Hn, s, N, Nc, Bst, Uall, Eall, linAall, wall, xall, step, chZ);
}

// The 2x2 algebra of the all-linear subtrees with ONE memory round trip per sweep (harmonic Newton step; k_lin_tree_* above are the
// fallback).  What an elimination step of bus k needs from memory does not depend on the steps before it: its own diagonal block and
// right-hand side and the coupling blocks with its parent, G = A(par,k), H = A(k,par), come from the state (U, E), the network (Y)
// and the mismatch image -- only D_child^-1 and w_child come from the children.  So a workgroup (a bundle of whole subtrees, NP
// items (bus, harmonic) per thread) first requests the operands of ALL its items at once and forms (M0, y0, G, H) in registers; the
// heights are then walked with a workgroup barrier in between and NOTHING but LDS traffic: a child leaves its Schur contribution
// G D^-1 H, G w in the slot (parent's first child slot + its ordinal), the parent subtracts its children's slots in list order --
// the arithmetic and its order are those of fold_children / lin_factor_item, bit for bit.  D^-1 and w go to HBM as before (dense
// parents, chains and the back sweep read them).
// The contracted chains ride in the same launches: a chain and the linear subtrees hanging off its buses sit in one bundle, the
// chain walk (chain_factor_item: it folds those subtrees from D^-1, w in HBM, written by this workgroup) follows the last height;
// the back sweep walks the chains first.  cbptr / cblist: chains per bundle (null: chains have their own launches).
// Records: Rec8 with cbeg = first child slot of the bus; xrec[record] = (own slot or -1 for a subtree root, local index of the parent
// inside the bundle or -1); bptr: nh + 1 record offsets per bundle (heights ascending).
#ifndef HPF_LBF_OCC
#define HPF_LBF_OCC 3       // waves per SIMD the one-round-trip 2x2 kernels (NP = 1) are compiled for (factor / back)
#endif
#ifndef HPF_LBB_OCC
#define HPF_LBB_OCC 4
#endif
template <int NP>
__global__ __launch_bounds__(256, NP == 1 ? HPF_LBF_OCC : 1) void k_lin_bundle_factor(Model M, const int* __restrict__ rec, const int2* __restrict__ xrec,
                                                           const int* __restrict__ bptr, int nh, int Bst, const int* __restrict__ active,
                                                           const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                           const double* __restrict__ fall, double* linAall, double* wall,
                                                           const cplx* __restrict__ I0all, int s0, TreeDev T,
                                                           const int* __restrict__ cbptr, const int* __restrict__ cblist,
                                                           const int* __restrict__ crec, const int* __restrict__ cnode,
                                                           double* __restrict__ chG, double* __restrict__ chH, double* __restrict__ chD,
                                                           double* __restrict__ chy, double* __restrict__ chZ) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;
    if (s < 0) return;
    __shared__ double ctr[256 * NP * 6];
    const int* bp = bptr + (size_t)blockIdx.x * (nh + 1);
    const int n = M.n, c = M.c, Hn = M.Hn, base = bp[0], nb = bp[nh] - base;
    const int nitems = nb * Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    double* linA = linAall + so * 4;
    double* ws = wall + (size_t)s * n * Bst;
    int lbv[NP], qv[NP], kv[NP], slot[NP], cs[NP], nch[NP];
    bool ok[NP];
    int4 r0v[NP], r1v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {                               // round trip 1: the records
        const int i = threadIdx.x + 256 * p;
        ok[p] = i < nitems;
        const int ic = ok[p] ? i : 0;
        lbv[p] = ic / Hn;
        qv[p] = ic - lbv[p] * Hn;
        r0v[p] = reinterpret_cast<const int4*>(rec)[2 * (size_t)(base + lbv[p])];
        r1v[p] = reinterpret_cast<const int4*>(rec)[2 * (size_t)(base + lbv[p]) + 1];
        slot[p] = xrec[base + lbv[p]].x;
    }
    cplx ydv[NP], ukv[NP], ekv[NP], I0v[NP], ynv[NP], ydnv[NP], yupv[NP], upv[NP], epv[NP];
    double2 fyv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {                               // round trip 2: every operand of every item
        const int k = r0v[p].x, q = qv[p], par = r0v[p].z;
        kv[p] = k;
        cs[p] = r1v[p].y;
        nch[p] = r1v[p].z;
        ydv[p] = M.Y[(size_t)r0v[p].y * Hn + q];
        ukv[p] = U[(size_t)k * Hn + q];
        ekv[p] = E[(size_t)k * Hn + q];
        I0v[p] = cplx{0.0, 0.0};
        ynv[p] = cplx{0.0, 0.0};
        if (q == 0 && k < M.m) I0v[p] = I0all[(size_t)s * n + k];
        if (k >= M.m) ynv[p] = M.coupled ? M.YN[((size_t)r1v[p].w * Hn + q) * Hn + q] : M.YN[(size_t)r1v[p].w * Hn + q];
        fyv[p] = *reinterpret_cast<const double2*>(fall + ((size_t)s * n + k) * Bst + 2 * q);
        ydnv[p] = yupv[p] = upv[p] = epv[p] = cplx{0.0, 0.0};
        if (slot[p] >= 0) {                                      // (a subtree root's coupling with its dense / chain parent is the parent's business)
            ydnv[p] = M.Y[(size_t)r1v[p].x * Hn + q];
            yupv[p] = M.Y[(size_t)r0v[p].w * Hn + q];
            upv[p] = U[(size_t)par * Hn + q];
            epv[p] = E[(size_t)par * Hn + q];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    double m2[NP][4], y2[NP][2], g4[NP][4], h4[NP][4];
    int hgt[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int k = kv[p], q = qv[p], par = r0v[p].z;
        diag2x2_val(n, c, M.m, q, k, ydv[p], ukv[p], ekv[p], I0v[p], ynv[p], m2[p]);
        y2[p][0] = fyv[p].x;
        y2[p][1] = fyv[p].y;
        if (slot[p] >= 0) {
            coupling_val(n, c, M.m, q, par, k, ydnv[p], upv[p], ukv[p], ekv[p], g4[p]);      // A(par, k)
            coupling_val(n, c, M.m, q, k, par, yupv[p], ukv[p], upv[p], epv[p], h4[p]);      // A(k, par)
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) g4[p][e] = h4[p][e] = 0.0;
        }
        int hh = 0;
        for (int a = 1; a < nh; ++a) hh += (lbv[p] >= bp[a] - base) ? 1 : 0;
        hgt[p] = ok[p] ? hh : -1;
    }
    for (int hh = 0; hh < nh; ++hh) {
        if (hh > 0) __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (hgt[p] != hh) continue;
            const int q = qv[p];
            double mm[4] = {m2[p][0], m2[p][1], m2[p][2], m2[p][3]}, y0 = y2[p][0], y1 = y2[p][1];
            for (int j = 0; j < nch[p]; ++j) {
                const double* c6 = ctr + ((size_t)(cs[p] + j) * Hn + q) * 6;
#pragma unroll
                for (int e = 0; e < 4; ++e) mm[e] -= c6[e];
                y0 -= c6[4];
                y1 -= c6[5];
            }
            double di[4];
            inv2(mm[0], mm[1], mm[2], mm[3], di[0], di[1], di[2], di[3]);
            const double w0 = fma(di[1], y1, di[0] * y0), w1 = fma(di[3], y1, di[2] * y0);
            double2* ik = reinterpret_cast<double2*>(linA + ((size_t)kv[p] * Hn + q) * 4);
            ik[0] = double2{di[0], di[1]};
            ik[1] = double2{di[2], di[3]};
            *reinterpret_cast<double2*>(ws + (size_t)kv[p] * Bst + 2 * q) = double2{w0, w1};
            if (slot[p] >= 0) {
                double gi[4], gh[4];
                mul22(g4[p], di, gi);
                mul22(gi, h4[p], gh);
                double* c6 = ctr + ((size_t)slot[p] * Hn + q) * 6;
#pragma unroll
                for (int e = 0; e < 4; ++e) c6[e] = gh[e];
                c6[4] = fma(g4[p][1], w1, g4[p][0] * w0);
                c6[5] = fma(g4[p][3], w1, g4[p][2] * w0);
            }
        }
        if (bp[hh + 1] == bp[nh]) break;                         // (uniform: nothing of this bundle above this height)
    }
    if (cbptr) {                                                 // the bundle's contracted chains (their linear subtrees are done)
        const int c0 = cbptr[blockIdx.x], nc2 = cbptr[blockIdx.x + 1] - c0;
        if (nc2 > 0) {
            __syncthreads();
            for (int t = threadIdx.x; t < nc2 * Hn; t += 256)
                chain_factor_item(M, T, crec, cnode, cblist[c0 + t / Hn], t % Hn, s, Bst, Uall, Eall, fall, linAall, wall, I0all, chG, chH, chD,
                                  chy, chZ);
        }
    }
}

// ... and its back sweep: x_k = w_k - D_k^-1 (A(k,par) x_par).  D^-1, w, A(k,par) of every item in one round trip (a subtree root
// also fetches x of its dense / chain parent), then the heights top-down with x_par through LDS.
template <int NP>
__global__ __launch_bounds__(256, NP == 1 ? HPF_LBB_OCC : 1) void k_lin_bundle_back(Model M, const int* __restrict__ rec, const int2* __restrict__ xrec,
                                                         const int* __restrict__ bptr, int nh, int Bst, const int* __restrict__ active,
                                                         const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                         const double* __restrict__ linAall, const double* __restrict__ wall,
                                                         double* xall, int s0, const int* __restrict__ cbptr,
                                                         const int* __restrict__ cblist, const int* __restrict__ crec,
                                                         const int* __restrict__ cnode, const double* __restrict__ chZ) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;
    if (s < 0) return;
    __shared__ double xl[256 * NP * 2];
    const int* bp = bptr + (size_t)blockIdx.x * (nh + 1);
    const int n = M.n, c = M.c, Hn = M.Hn, base = bp[0], nb = bp[nh] - base;
    const int nitems = nb * Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* linA = linAall + so * 4;
    const double* ws = wall + (size_t)s * n * Bst;
    double* xs = xall + (size_t)s * n * Bst;
    int lbv[NP], qv[NP], lpar[NP];
    bool ok[NP];
    int4 r0v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = threadIdx.x + 256 * p;
        ok[p] = i < nitems;
        const int ic = ok[p] ? i : 0;
        lbv[p] = ic / Hn;
        qv[p] = ic - lbv[p] * Hn;
        r0v[p] = reinterpret_cast<const int4*>(rec)[2 * (size_t)(base + lbv[p])];
        lpar[p] = xrec[base + lbv[p]].y;
    }
    cplx yupv[NP], ukv[NP], upv[NP], epv[NP];
    double2 i01[NP], i23[NP], wkv[NP], xpv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int k = r0v[p].x, par = r0v[p].z > 0 ? r0v[p].z : 0, q = qv[p];      // (par < 0: the network's root, when it is a 2x2 bus itself)
        yupv[p] = M.Y[(size_t)r0v[p].w * Hn + q];
        ukv[p] = U[(size_t)k * Hn + q];
        upv[p] = U[(size_t)par * Hn + q];
        epv[p] = E[(size_t)par * Hn + q];
        const double2* pik = reinterpret_cast<const double2*>(linA + ((size_t)k * Hn + q) * 4);
        i01[p] = pik[0];
        i23[p] = pik[1];
        wkv[p] = *reinterpret_cast<const double2*>(ws + (size_t)k * Bst + 2 * q);
        xpv[p] = double2{0.0, 0.0};
    }
    if (cbptr) {                                                 // the bundle's chains first: subtree roots below take x of their chain bus
        const int c0 = cbptr[blockIdx.x], nc2 = cbptr[blockIdx.x + 1] - c0;
        if (nc2 > 0) {
            for (int t = threadIdx.x; t < nc2 * Hn; t += 256)
                chain_back_item(M, crec, cnode, cblist[c0 + t / Hn], t % Hn, s, 0, 0, Bst, Uall, Eall, linAall, wall, xall, nullptr, chZ);
            __syncthreads();
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
        if (lpar[p] < 0 && r0v[p].z >= 0) xpv[p] = *reinterpret_cast<const double2*>(xs + (size_t)r0v[p].z * Bst + 2 * qv[p]);
    __builtin_amdgcn_sched_barrier(0);
    double h4[NP][4];
    int hgt[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        coupling_val(n, c, M.m, qv[p], r0v[p].x, r0v[p].z > 0 ? r0v[p].z : 0, yupv[p], ukv[p], upv[p], epv[p], h4[p]);      // A(k, parent)
        if (r0v[p].z < 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) h4[p][e] = 0.0;          // no parent: x = w
        }
        int hh = 0;
        for (int a = 1; a < nh; ++a) hh += (lbv[p] >= bp[a] - base) ? 1 : 0;
        hgt[p] = ok[p] ? hh : -1;
    }
    bool first = true;
    for (int hh = nh - 1; hh >= 0; --hh) {
        if (bp[hh] == bp[nh]) continue;                          // (uniform: the bundle is lower than this height)
        if (!first) __syncthreads();
        first = false;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (hgt[p] != hh) continue;
            const int q = qv[p];
            double2 xp = xpv[p];
            if (lpar[p] >= 0) xp = *reinterpret_cast<const double2*>(xl + ((size_t)lpar[p] * Hn + q) * 2);
            const double t0 = fma(h4[p][1], xp.y, h4[p][0] * xp.x), t1 = fma(h4[p][3], xp.y, h4[p][2] * xp.x);
            const double x0 = wkv[p].x - fma(i01[p].y, t1, i01[p].x * t0);
            const double x1 = wkv[p].y - fma(i23[p].y, t1, i23[p].x * t0);
            *reinterpret_cast<double2*>(xl + ((size_t)lbv[p] * Hn + q) * 2) = double2{x0, x1};
            *reinterpret_cast<double2*>(xs + (size_t)r0v[p].x * Bst + 2 * q) = double2{x0, x1};
        }
    }
}
