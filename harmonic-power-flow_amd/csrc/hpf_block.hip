// BLOCK_TREE Newton step: bus-major block elimination of the harmonic Jacobian along a radial feeder.
//
// Why: in the reference's stacked ordering (HG:469-472) the Jacobian of an n-bus, Hn-harmonic feeder is one sparse
// N x N matrix (N = 2 n Hn - 1 - c) that the reference hands to SuperLU (HG:478).  Ordered bus-major it is a block
// matrix on the network graph with blocks of size b = 2 Hn: the diagonal block of a nonlinear bus is dense (Norton
// cross-coupling, HG:425-435), every off-diagonal block is "harmonic-diagonal" (2x2 per harmonic, HG:403-411).
// On a tree, eliminating buses leaves -> root creates no fill:
//     Z_k = D_k^{-1} A(k,parent),  w_k = D_k^{-1} y_k,
//     D_p -= A(p,k) Z_k,  y_p -= A(p,k) w_k          (pulled by the parent, fixed child order -> deterministic)
//     x_root = w_root,  x_k = w_k - Z_k x_parent      (root -> leaves)
// Missing unknowns/equations (slack at h=1, V_m/Q of PV buses at h=1) are padded with identity rows so that every
// block is b x b.
//
// One workgroup (256 threads = 16 x 16) owns one (bus, scenario) pair.  D_k is ASSEMBLED IN REGISTERS from U/E/Y/Y_N
// (it never exists in HBM), each thread owning the R x R sub-grid {tr+16a} x {tc+16c}; the children's Z blocks are
// pulled from HBM; D_k is inverted in place by Gauss-Jordan with partial (row) pivoting, rows/columns of each step
// broadcast through LDS (2 barriers per step); Z_k (b x b) and w_k (b) are the only HBM writes.
// Bound: FP64 FMA rate (2 b^3 flop per bus) against 24 b^2 bytes of Z traffic per bus -> ~4.3 flop/B at b = 52.
#include <algorithm>
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <rocsolver/rocsolver.h>
#include <thread>

#include "hpf_internal.hpp"
#include "hpf_gj.hpp"
#include "hpf_blk_jobs.hpp"
#include "hpf_blk_invert_mfma.hpp"
#include "hpf_gj_dense.hpp"
#include "hpf_gj_mfma.hpp"

using namespace hpf;

#ifdef HPF_FACTOR_STAMPS
#define HPF_STAMP_DECL long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0
#define HPF_STAMP(v) \
    if (ablate & 16) v = __builtin_amdgcn_s_memtime()
#else
#define HPF_STAMP_DECL
#define HPF_STAMP(v)
#endif

namespace {

constexpr int FDESC = 48;   // ints per node record of the multi-wave factor kernel (Tree::d_fdesc)

struct TreeDev {
    const int* parent;
    const int* child_ptr;
    const int* child;
    const int* e_up;
    const int* e_dn;
    const int* child_mid;   // children [child_ptr, child_mid) have all-linear subtrees, [child_mid, child_ptr+1) are dense
    const int* lin_ptr;
    const int* lin_post;
    const int* child3;      // [child-list position][4]: child, e_dn[child], e_up[child], 0
    const int* dchild;      // dense children (through contracted chains) of the dense buses, see the node records
    const int* chain_ptr;   // contracted chains of pass-through buses, bottom-up
    const int* chain_nodes;
    const int* chain_ch;    // the dense bus below the chain
    const int* lzrec;       // lazy-leaf records of the parents (Tree::d_lzrec)
    const double* lzimg;    // and their per-model images (Tree::d_lzimg)
    double* cF;             // compress steps: per-scenario slots (hpf_handle::d_F), A(v,c) blocks (d_H2), the pending children
    double* cH2;
    const int* comp_child;
    int n_comp;
};

// validity of local index l = 2q+t of bus i as an unknown / equation (same rule for both, see hpf_assembly.hpp)
__device__ __forceinline__ bool loc_valid(int n, int c, int i, int l) {
    const int kst = (l >> 1) * n + i;
    return (l & 1) ? kst >= c : kst >= 1;
}

__device__ __forceinline__ double pick(const Blk2& b, int t, int tc) {
    return t == 0 ? (tc == 0 ? b.dA.re : b.dV.re) : (tc == 0 ? b.dA.im : b.dV.im);
}

// off-diagonal 2x2 block of row bus i w.r.t. column bus j (stored entry e) at harmonic position q
__device__ __forceinline__ Blk2 offdiag_block(const Model& M, const cplx* U, const cplx* E, int q, int i, int j, int e) {
    if (q == 0 && i < M.m) return jac_power_entry<false>(M, U, E, i, j, e);
    return jac_current_entry(M, U, E, q, i, j, e);
}

template <int R>
__global__ __launch_bounds__(256) void k_tree_factor(Model M, TreeDev T, const int* __restrict__ nodes, int b, int N,
                                                     int Nc, const int* __restrict__ active,
                                                     const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                     const double* __restrict__ fall, double* __restrict__ Zall,
                                                     double* __restrict__ wall, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int k = nodes[blockIdx.x];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    const int n = M.n, c = M.c;
    const size_t so = (size_t)s * n * M.Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* f = fall + (size_t)s * N;
    const size_t bb = (size_t)b * b;
    double* Zs = Zall + (size_t)s * n * bb;
    double* ws = wall + (size_t)s * n * b;

    extern __shared__ double lds[];
    const GjDenseLds L(lds, b);              // Rm [b][ldr] | colbuf, rowr, rowj, ybuf [b] | piv, pinv, pfwd [b]  (hpf_gj_dense.hpp)
    const int ldr = L.ldr;
    double *Rm = L.Rm, *ybuf = L.ybuf;
    int *pinv = L.pinv, *pfwd = L.pfwd;

    // ---- A. assemble D_k in registers -------------------------------------------------------------------------
    double a[R][R];
    const int diag_e = M.diag[k];
#pragma unroll
    for (int ai = 0; ai < R; ++ai) {
        const int i = tr + 16 * ai;
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            double v = 0.0;
            if (i < b && cc < b) {
                const bool vi = loc_valid(n, c, k, i), vc = loc_valid(n, c, k, cc);
                if (!vi || !vc) {
                    v = (i == cc) ? 1.0 : 0.0;
                } else {
                    const int q = i >> 1, p = cc >> 1;
                    if (q == p) {
                        const Blk2 blk = (q == 0 && k < M.m) ? jac_power_entry<false>(M, U, E, k, k, diag_e)
                                                              : jac_current_entry(M, U, E, q, k, k, diag_e);
                        v = pick(blk, i & 1, cc & 1);
                    } else if (k >= M.m && M.coupled) {
                        v = pick(jac_norton_cross(M, U, E, q, p, k), i & 1, cc & 1);
                    }
                }
            }
            a[ai][ci] = v;
        }
    }
    // right-hand side y_k = mismatch rows of this bus
    if (tid < b) {
        const int kst = (tid >> 1) * n + k;
        double v = 0.0;
        if (loc_valid(n, c, k, tid)) v = (tid & 1) ? f[Nc + kst - c] : f[kst - 1];
        ybuf[tid] = v;
    }
    __syncthreads();

    // ---- B. pull the children's Schur complements (fixed order) ------------------------------------------------
    for (int cp = T.child_ptr[k]; cp < T.child_ptr[k + 1]; ++cp) {
        const int ch = T.child[cp];
        const int e = T.e_dn[ch];
        const double* Zc = Zs + (size_t)ch * bb;
        const double* wc = ws + (size_t)ch * b;
#pragma unroll
        for (int ai = 0; ai < R; ++ai) {
            const int i = tr + 16 * ai;
            if (i >= b) continue;
            if (!loc_valid(n, c, k, i)) continue;
            const int q = i >> 1, t = i & 1;
            const Blk2 blk = offdiag_block(M, U, E, q, k, ch, e);
            const double g0 = pick(blk, t, 0);                                     // child angle is always an unknown
            const double g1 = loc_valid(n, c, ch, 2 * q + 1) ? pick(blk, t, 1) : 0.0;
            const double* z0 = Zc + (size_t)(2 * q) * b;
            const double* z1 = z0 + b;
#pragma unroll
            for (int ci = 0; ci < R; ++ci) {
                const int cc = tc + 16 * ci;
                if (cc < b) {
                    a[ai][ci] = fma(-g0, z0[cc], a[ai][ci]);
                    a[ai][ci] = fma(-g1, z1[cc], a[ai][ci]);
                }
            }
            if (tc == 0) {
                double y = ybuf[i];
                y = fma(-g0, wc[2 * q], y);
                y = fma(-g1, wc[2 * q + 1], y);
                ybuf[i] = y;
            }
        }
    }

    // ---- C / D. in-place Gauss-Jordan inversion with partial pivoting: R = (P D)^{-1} -> LDS (Rm), permutation pfwd / pinv --------
    gj_dense_invert<R>(a, b, L, nullptr);

    // ---- E. w_k = D^{-1} y,  Z_k = D^{-1} A(k, parent) ----------------------------------------------------------
    if (tid < b) {
        double acc = 0.0;
        const double* Rrow = Rm + (size_t)tid * ldr;
        for (int kk = 0; kk < b; ++kk) acc = fma(Rrow[kk], ybuf[pfwd[kk]], acc);
        ws[(size_t)k * b + tid] = acc;
    }
    const int par = T.parent[k];
    if (par >= 0) {
        const int e = T.e_up[k];
        double* Zk = Zs + (size_t)k * bb;
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            if (cc >= b) continue;
            const int q = cc >> 1, tcol = cc & 1;
            double b0 = 0.0, b1 = 0.0;
            if (loc_valid(n, c, par, cc)) {
                const Blk2 blk = offdiag_block(M, U, E, q, k, par, e);
                if (loc_valid(n, c, k, 2 * q)) b0 = pick(blk, 0, tcol);
                if (loc_valid(n, c, k, 2 * q + 1)) b1 = pick(blk, 1, tcol);
            }
            const int k0 = pinv[2 * q], k1 = pinv[2 * q + 1];
#pragma unroll
            for (int ai = 0; ai < R; ++ai) {
                const int i = tr + 16 * ai;
                if (i >= b) continue;
                const double* Rrow = Rm + (size_t)i * ldr;
                Zk[(size_t)i * b + cc] = fma(Rrow[k1], b1, Rrow[k0] * b0);
            }
        }
    }
}

// root -> leaves: x_k = w_k - Z_k x_parent, written bus-major (for the children) and in the stacked order of the
// reference's state vector (for the update kernel).
__global__ __launch_bounds__(256) void k_tree_back(int n, int c, int Hn, TreeDev T, const int* __restrict__ nodes, int b,
                                                   int N, int Nc, const int* __restrict__ active,
                                                   const double* __restrict__ Zall, const double* __restrict__ wall,
                                                   double* __restrict__ xall, double* __restrict__ step, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int k = nodes[blockIdx.x];
    const size_t bb = (size_t)b * b;
    const double* Zk = Zall + ((size_t)s * n + k) * bb;
    const double* wk = wall + ((size_t)s * n + k) * b;
    double* xs = xall + (size_t)s * n * b;
    double* st = step + (size_t)s * N;
    const int par = T.parent[k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < b; i += 4) {
        double acc = 0.0;
        if (par >= 0) {
            const double* xp = xs + (size_t)par * b;
            for (int cc = lane; cc < b; cc += 64) acc = fma(Zk[(size_t)i * b + cc], xp[cc], acc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        }
        if (lane == 0) {
            const double x = wk[i] - acc;
            xs[(size_t)k * b + i] = x;
            const int kst = (i >> 1) * n + k;
            if (i & 1) {
                if (kst >= c) st[Nc + kst - c] = x;
            } else {
                if (kst >= 1) st[kst - 1] = x;
            }
        }
    }
}


// =============================================================================================================
// Wave-per-bus kernels (b <= 52): one 64-lane wavefront owns one (bus, scenario) pair, lane = matrix row.
//
//  * D_k lives in registers, one row per lane (a[c], c static).  No workgroup barrier, no inter-wave traffic.
//  * Gauss-Jordan without row scaling and without row swaps (implicit partial pivoting): at step j the pivot lane r is
//    the not-yet-used lane with the largest |a[.][j]| (wave-wide max of the IEEE bit pattern, lane id in the low bits);
//    it broadcasts its row through LDS, every other lane subtracts g = a[j]/pivot times that row.  Column j is replaced
//    in place by the j-th column of the inverse.  The register array ROTATES by one position per step (the FMA writes
//    logical column c+1 into physical register c), so that the current column is always register 0 and the loop body
//    is the same for every j: a rolled loop with static register indices, ~1.5 KB of code instead of B unrolled steps.
//    Each lane is the pivot exactly once (step myj); at the end lane x holds row myj(x) of the inverse times its
//    pivot, register j being the column r_j (the lane that was pivot at step j).
//  * Output is the TRANSPOSED inverse AinvT[c][i] = D_k^{-1}[i][c] (b x b, stored B x B) and w = D_k^{-1} y: for a
//    fixed register j all lanes store into one row r_j of AinvT -> one coalesced 8*B-byte store per register.
//    The parent applies the harmonic-diagonal coupling blocks on the fly:
//        D_p -= A(p,k) D_k^{-1} A(k,p):   v = g0*Ainv[2q][:] + g1*Ainv[2q+1][:]  (16-B pair loads, coalesced)
//                                          D_p[i][2p+t'] -= v[2p]*Bup[p][0][t'] + v[2p+1]*Bup[p][1][t'].
// =============================================================================================================
// fundamental power flow (HG:205-223): every bus is a power row, E = U/|U|
__device__ __forceinline__ void coupling_block_fund(const Model& M, const cplx* U, const cplx* E, int i, int j, int e, double out[4]) {
    const Blk2 blk = jac_power_entry<true>(M, U, E, i, j, e);
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
            out[tr * 2 + tc] = (loc_valid(M.n, M.c, i, tr) && loc_valid(M.n, M.c, j, tc)) ? pick(blk, tr, tc) : 0.0;
}

// 2x2 coupling block A(row bus i, col bus j) at harmonic p, masked to existing equations / unknowns -> out[tr*2+tc]
__device__ __forceinline__ void coupling_block(const Model& M, const cplx* U, const cplx* E, int p, int i, int j, int e,
                                               double out[4]) {
    const Blk2 blk = offdiag_block(M, U, E, p, i, j, e);
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
            out[tr * 2 + tc] = (loc_valid(M.n, M.c, i, 2 * p + tr) && loc_valid(M.n, M.c, j, 2 * p + tc))
                                   ? pick(blk, tr, tc) : 0.0;
}


// Row `lane` (= 2q+t) of the un-eliminated diagonal block D_k and its right-hand side y: network entry (k,k) on the harmonic
// diagonal (+ the p == q Norton term) minus the Schur complements of the children whose whole subtree is linear (harmonic-
// diagonal, inverted by k_lin_factor), and, at a nonlinear bus, the Norton cross-coupling -Y_N[q,p]*(E | jU)_{p,k}
// (HG:425-435) with the bus voltages staged once in LDS.  Identity padding for missing unknowns / equations.
template <int B>
__device__ __forceinline__ void assemble_row(const Model& M, const TreeDev& T, const cplx* U, const cplx* E,
                                             const double* f, const double* ws, const double* linA, int k, int lane,
                                             int b, int Nc, int ablate, double* ue /*LDS (B/2)*8*/, double (&a)[B],
                                             double& y) {
    const int n = M.n, c = M.c, Hn = M.Hn;
    const int q = lane >> 1, t = lane & 1;
    const bool rowvalid = lane < b && loc_valid(n, c, k, lane);
    y = 0.0;
    if (rowvalid) {
        const int kst = q * n + k;
        y = t ? f[Nc + kst - c] : f[kst - 1];
    }
    double d0 = 0.0, d1 = 0.0;
    if (rowvalid && !(ablate & 4)) {
        const int diag_e = M.diag[k];
        const Blk2 blk = (q == 0 && k < M.m) ? jac_power_entry<false>(M, U, E, k, k, diag_e)
                                              : jac_current_entry(M, U, E, q, k, k, diag_e);
        d0 = pick(blk, t, 0);
        d1 = pick(blk, t, 1);
        for (int cp = T.child_ptr[k]; cp < T.child_mid[k]; ++cp) {
            const int ch = T.child[cp];
            const Blk2 g = offdiag_block(M, U, E, q, k, ch, T.e_dn[ch]);     // A(parent, child), my harmonic
            const double g0 = pick(g, t, 0);
            const double g1 = loc_valid(n, c, ch, 2 * q + 1) ? pick(g, t, 1) : 0.0;
            double h4[4];
            coupling_block(M, U, E, q, ch, k, T.e_up[ch], h4);               // A(child, parent)
            const double* ic = linA + ((size_t)ch * Hn + q) * 4;
            const double v0 = fma(g1, ic[2], g0 * ic[0]), v1 = fma(g1, ic[3], g0 * ic[1]);
            d0 -= fma(v1, h4[2], v0 * h4[0]);
            d1 -= fma(v1, h4[3], v0 * h4[1]);
            const double* wc = ws + (size_t)ch * B;
            y = fma(-g0, wc[2 * q], y);
            y = fma(-g1, wc[2 * q + 1], y);
        }
    }
    // Norton cross-coupling of a nonlinear bus (HG:425-435): entry [2q+t][2p+t'] = -(Y_N[q,p] * (jU | E)_{p,k}) picked Re/Im.
    // Written out per component, with the exact rounding of the reference expression (every product and the final sum
    // rounded once, a negation is exact):   v = yi*P + yr*Q   with (P,Q) a signed selection of the bus voltage parts that
    // depends only on (t, t') -> staged per harmonic in LDS as tab[t][p] = {P0, Q0, P1, Q1}:
    //   t=0: dA.re = yi*ur + yr*ui,      dV.re = yi*ei + yr*(-er);     t=1: dA.im = yi*ui + yr*(-ur),  dV.im = yi*(-er) + yr*(-ei).
    const bool nl = k >= M.m && M.coupled && !(ablate & 4);
    if (nl) {
        __syncthreads();
        if (lane < Hn) {
            const cplx u = U[M.vi(lane, k)], e = E[M.vi(lane, k)];
            double* t0 = ue + lane * 4;
            double* t1 = ue + (B / 2) * 4 + lane * 4;
            t0[0] = u.re;  t0[1] = u.im;   t0[2] = e.im;   t0[3] = -e.re;
            t1[0] = u.im;  t1[1] = -u.re;  t1[2] = -e.re;  t1[3] = -e.im;
        }
        __syncthreads();
        const cplx* ynrow = M.YN + ((size_t)M.dev[k] * Hn + (rowvalid ? q : 0)) * Hn;
        const double* tab = ue + t * (B / 2) * 4;
        // the lane's Y_N row is fetched in two batches with ALL loads of a batch in flight (sched_barrier pins them in
        // front of their uses; left alone, the scheduler serialises one load per harmonic = one memory round trip each)
        constexpr int HB = (B / 2 + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            cplx ynb[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int p = half * HB + j;
                ynb[j] = ynrow[p < Hn ? p : 0];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int p = half * HB + j;
                if (p < B / 2) {
                    const double v0 = ynb[j].im * tab[p * 4 + 0] + ynb[j].re * tab[p * 4 + 1];
                    const double v1 = ynb[j].im * tab[p * 4 + 2] + ynb[j].re * tab[p * 4 + 3];
                    a[2 * p] = (p == q) ? d0 : v0;
                    a[2 * p + 1] = (p == q) ? d1 : v1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int p = 0; p < B / 2; ++p) {
            a[2 * p] = (p == q) ? d0 : 0.0;
            a[2 * p + 1] = (p == q) ? d1 : 0.0;
        }
    }
    // identity padding: missing unknowns / equations exist only at the slack and PV buses (harmonic position 0) and, when the
    // block is padded (b < B), beyond b -> wave-uniform rare branch
    if (k < c || b < B) {
#pragma unroll
        for (int cc = 0; cc < B; ++cc) {
            const bool cv = cc < b && loc_valid(n, c, k, cc);
            if (!cv || !rowvalid) a[cc] = (cc == lane) ? 1.0 : 0.0;
        }
    }
}

// One wave per (bus, scenario), lane = row: Gauss-Jordan with partial pivoting over the whole block on the uncontracted tree
// ("block_pivoting" = 1; also the path a scenario is repeated on when the static-pivot MFMA kernels flag it, hpf.h).
template <int B>
__global__ __launch_bounds__(64, 1) void k_factor_w(Model M, TreeDev T, const int* __restrict__ nodes, int b, int N, int Nc,
                                                 const int* __restrict__ active, const cplx* __restrict__ Uall,
                                                 const cplx* __restrict__ Eall, const double* __restrict__ fall,
                                                 double* __restrict__ Aall, double* __restrict__ wall,
                                                 const double* __restrict__ linAall, int* __restrict__ pivflag, int ablate,
                                                 int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int k = nodes[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = M.n, c = M.c, Hn = M.Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* f = fall + (size_t)s * N;
    constexpr size_t BB = (size_t)B * B;
    double* As = Aall + (size_t)s * n * BB;
    double* ws = wall + (size_t)s * n * B;

    __shared__ double bup[(B / 2) * 8];
    __shared__ int rj[B];

    const int q = lane >> 1, t = lane & 1;
    const bool rowvalid = lane < b && loc_valid(n, c, k, lane);

    // ---- A. row `lane` of D_k and y (assemble_row) --------------------------------------------------------------------
    double a[B], y;
    assemble_row<B>(M, T, U, E, f, ws, linAall + (size_t)s * n * Hn * 4, k, lane, b, Nc, ablate, bup, a, y);

    // ---- B. dense children (fixed order): pull from the children's transposed inverses --------------------------------
    for (int cp = T.child_mid[k]; cp < ((ablate & 2) ? 0 : T.child_ptr[k + 1]); ++cp) {
        const int ch = T.child[cp];
        __syncthreads();
        if (lane < Hn) {
            double blk4[4];
            coupling_block(M, U, E, lane, ch, k, T.e_up[ch], blk4);      // A(child, parent) at harmonic `lane`
            bup[lane * 4 + 0] = blk4[0];
            bup[lane * 4 + 1] = blk4[1];
            bup[lane * 4 + 2] = blk4[2];
            bup[lane * 4 + 3] = blk4[3];
        }
        double g0 = 0.0, g1 = 0.0;
        if (rowvalid) {
            const Blk2 blk = offdiag_block(M, U, E, q, k, ch, T.e_dn[ch]);   // A(parent, child) at my harmonic
            g0 = pick(blk, t, 0);
            g1 = loc_valid(n, c, ch, 2 * q + 1) ? pick(blk, t, 1) : 0.0;
        }
        const double* Ac = As + (size_t)ch * BB;
        const double* wc = ws + (size_t)ch * B;
        const int qq = q < B / 2 ? q : 0;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < B / 2; ++p) {
            const double2 z0 = *reinterpret_cast<const double2*>(Ac + (size_t)(2 * p) * B + 2 * qq);
            const double2 z1 = *reinterpret_cast<const double2*>(Ac + (size_t)(2 * p + 1) * B + 2 * qq);
            const double v0 = fma(g1, z0.y, g0 * z0.x);       // (A(p,k) Ainv)[i][2p]
            const double v1 = fma(g1, z1.y, g0 * z1.x);       // (A(p,k) Ainv)[i][2p+1]
            a[2 * p] = fma(-v0, bup[p * 4 + 0], a[2 * p]);
            a[2 * p] = fma(-v1, bup[p * 4 + 2], a[2 * p]);
            a[2 * p + 1] = fma(-v0, bup[p * 4 + 1], a[2 * p + 1]);
            a[2 * p + 1] = fma(-v1, bup[p * 4 + 3], a[2 * p + 1]);
        }
        y = fma(-g0, wc[2 * qq], y);
        y = fma(-g1, wc[2 * qq + 1], y);
    }

    // ---- C. Gauss-Jordan with implicit partial pivoting, rotating registers (hpf_gj.hpp) ---------------------
    int myj = 0;
    double mypiv = 1.0;
    if (ablate & 1) ablate |= 8;
    gauss_jordan_wave_rl<B>(a, y, lane, (ablate & 1) ? 0 : B, rj, myj, mypiv);
    // an exactly zero pivot = a singular block (what rocSOLVER reports as info > 0 on the dense path): HPF_E_SINGULAR
    if (lane < B && mypiv == 0.0) atomicOr(pivflag + s, 4);

    // ---- D. store the transposed inverse and w -------------------------------------------------------------
    __syncthreads();
    if (lane < B && !(ablate & 8)) {
        const double invp = 1.0 / mypiv;
        double* Ak = As + (size_t)k * BB;
#pragma unroll
        for (int j = 0; j < B; ++j) Ak[(size_t)rj[j] * B + myj] = a[j] * invp;
        ws[(size_t)k * B + myj] = y * invp;
    }
}

// root -> leaves: x_k = w_k - D_k^{-1} (A(k,parent) x_parent); one wave per (bus, scenario), lane = row
template <int B>
__global__ __launch_bounds__(64) void k_back_w(Model M, TreeDev T, const int* __restrict__ nodes, int b, int N, int Nc,
                                               const int* __restrict__ active, const cplx* __restrict__ Uall,
                                               const cplx* __restrict__ Eall, const double* __restrict__ Aall,
                                               const double* __restrict__ wall, double* __restrict__ xall,
                                               double* __restrict__ step, const double* __restrict__ Hall, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int k = nodes[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = M.n, c = M.c, Hn = M.Hn;
    constexpr size_t BB = (size_t)B * B;
    const double* Ak = Aall + ((size_t)s * n + k) * BB;
    const double* wk = wall + ((size_t)s * n + k) * B;
    double* xs = xall + (size_t)s * n * B;
    __shared__ double tb[B];
    const int par = T.parent[k];
    double x = 0.0;
    if (lane < B) x = wk[lane];
    if (par >= 0) {
        const size_t so = (size_t)s * n * Hn;
        const double* xp = xs + (size_t)par * B;
        double tv = 0.0;
        const int p = lane >> 1, tr = lane & 1;
        if (lane < b && p < Hn) {
            if (Hall) {                                      // stored by the factor kernel's push phase
                const double* hk = Hall + ((size_t)s * n + k) * Hn * 4 + p * 4 + tr * 2;
                tv = fma(hk[1], xp[2 * p + 1], hk[0] * xp[2 * p]);
            } else {
                double blk4[4];
                coupling_block(M, Uall + so, Eall + so, p, k, par, T.e_up[k], blk4);
                tv = fma(blk4[tr * 2 + 1], xp[2 * p + 1], blk4[tr * 2] * xp[2 * p]);
            }
        }
        if (lane < B) tb[lane] = tv;
        __syncthreads();
        if (lane < B) {
            constexpr int QB = (B + 3) / 4;                  // four batches, every load of a batch in flight
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                double av[QB];
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    const int cc = qb * QB + j;
                    av[j] = cc < B ? Ak[(size_t)cc * B + lane] : 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    const int cc = qb * QB + j;
                    if (cc < B) x = fma(-av[j], tb[cc], x);
                }
            }
        }
    }
    if (lane < B) {
        xs[(size_t)k * B + lane] = x;
        if (lane < b) {
            const int kst = (lane >> 1) * n + k;
            double* st = step + (size_t)s * N;
            if (lane & 1) {
                if (kst >= c) st[Nc + kst - c] = x;
            } else {
                if (kst >= 1) st[kst - 1] = x;
            }
        }
    }
}


// =============================================================================================================
// Buses whose whole subtree is linear (no nonlinear bus below them; with uncoupled Norton data: every bus): their
// Schur-complemented diagonal block stays harmonic-diagonal, i.e. Hn independent 2x2 systems.  One thread per
// (maximal linear subtree, harmonic, scenario) walks the subtree in post-order (factor) / reverse post-order (back).
// =============================================================================================================
template <bool FUND>
__device__ __forceinline__ void diag2x2(const Model& M, const cplx* U, const cplx* E, const cplx* I0, int q, int k, double m2[4]) {
    // harmonic NR power rows: row current kept by the mismatch kernel of the same state (I0)
    const Blk2 blk = FUND ? jac_power_entry<true>(M, U, E, k, k, M.diag[k])
                          : ((q == 0 && k < M.m) ? jac_power_diag(M, U, E, k, M.diag[k], I0[k])
                                                  : jac_current_entry(M, U, E, q, k, k, M.diag[k]));
    const bool v0 = loc_valid(M.n, M.c, k, 2 * q), v1 = loc_valid(M.n, M.c, k, 2 * q + 1);
    m2[0] = v0 ? blk.dA.re : 1.0;
    m2[1] = (v0 && v1) ? blk.dV.re : 0.0;
    m2[2] = (v0 && v1) ? blk.dA.im : 0.0;
    m2[3] = v1 ? blk.dV.im : 1.0;
}

template <bool FUND>
__global__ __launch_bounds__(128) void k_lin_factor(Model M, TreeDev T, int nroots, int N, int Nc, int Bst,
                                                    const int* __restrict__ active, const cplx* __restrict__ Uall,
                                                    const cplx* __restrict__ Eall, const double* __restrict__ fall,
                                                    double* __restrict__ linAall, double* __restrict__ wall,
                                                    const cplx* __restrict__ I0all, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tix = blockIdx.x * 128 + threadIdx.x;
    const int HnE = FUND ? 1 : M.Hn;                               // FUND: harmonic position 0 only
    if (tix >= nroots * HnE) return;
    const int q = tix % HnE, r = tix / HnE;                        // consecutive threads: consecutive harmonics
    const int n = M.n, c = M.c, Hn = M.Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* f = fall + (size_t)s * N;
    double* linA = linAall + so * 4;
    double* ws = wall + (size_t)s * n * Bst;
    for (int idx = T.lin_ptr[r]; idx < T.lin_ptr[r + 1]; ++idx) {
        const int k = T.lin_post[idx];
        double m2[4];
        diag2x2<FUND>(M, U, E, FUND ? nullptr : I0all + (size_t)s * n, q, k, m2);
        const int kst = q * n + k;
        double y0 = kst >= 1 ? f[kst - 1] : 0.0;
        double y1 = kst >= c ? f[Nc + kst - c] : 0.0;
        for (int cp = T.child_ptr[k]; cp < T.child_ptr[k + 1]; ++cp) {      // all children of a linear-subtree bus are linear
            const int ch = T.child[cp];
            double g4[4], h4[4];
            if (FUND) {
                coupling_block_fund(M, U, E, k, ch, T.e_dn[ch], g4);
                coupling_block_fund(M, U, E, ch, k, T.e_up[ch], h4);
            } else {
                coupling_block(M, U, E, q, k, ch, T.e_dn[ch], g4);           // A(k, child)
                coupling_block(M, U, E, q, ch, k, T.e_up[ch], h4);           // A(child, k)
            }
            const double* ic = linA + ((size_t)ch * Hn + q) * 4;
            const double gi0 = fma(g4[1], ic[2], g4[0] * ic[0]), gi1 = fma(g4[1], ic[3], g4[0] * ic[1]);
            const double gi2 = fma(g4[3], ic[2], g4[2] * ic[0]), gi3 = fma(g4[3], ic[3], g4[2] * ic[1]);
            m2[0] -= fma(gi1, h4[2], gi0 * h4[0]);
            m2[1] -= fma(gi1, h4[3], gi0 * h4[1]);
            m2[2] -= fma(gi3, h4[2], gi2 * h4[0]);
            m2[3] -= fma(gi3, h4[3], gi2 * h4[1]);
            const double* wc = ws + (size_t)ch * Bst + 2 * q;
            y0 -= fma(g4[1], wc[1], g4[0] * wc[0]);
            y1 -= fma(g4[3], wc[1], g4[2] * wc[0]);
        }
        double i0, i1, i2, i3;
        inv2(m2[0], m2[1], m2[2], m2[3], i0, i1, i2, i3);
        double* ik = linA + ((size_t)k * Hn + q) * 4;
        ik[0] = i0;
        ik[1] = i1;
        ik[2] = i2;
        ik[3] = i3;
        double* wk = ws + (size_t)k * Bst + 2 * q;
        wk[0] = fma(i1, y1, i0 * y0);
        wk[1] = fma(i3, y1, i2 * y0);
    }
}

template <bool FUND>
__global__ __launch_bounds__(128) void k_lin_back(Model M, TreeDev T, int nroots, int N, int Nc, int Bst,
                                                  const int* __restrict__ active, const cplx* __restrict__ Uall,
                                                  const cplx* __restrict__ Eall, const double* __restrict__ linAall,
                                                  const double* __restrict__ wall, double* __restrict__ xall,
                                                  double* __restrict__ step, int s0) {
    const int s = active ? active[blockIdx.y + s0] : (int)blockIdx.y + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int tix = blockIdx.x * 128 + threadIdx.x;
    const int HnE = FUND ? 1 : M.Hn;
    if (tix >= nroots * HnE) return;
    const int q = tix % HnE, r = tix / HnE;
    const int n = M.n, c = M.c, Hn = M.Hn;
    const size_t so = (size_t)s * n * Hn;
    const double* linA = linAall + so * 4;
    const double* ws = wall + (size_t)s * n * Bst;
    double* xs = xall + (size_t)s * n * Bst;
    double* st = step + (size_t)s * N;
    for (int idx = T.lin_ptr[r + 1] - 1; idx >= T.lin_ptr[r]; --idx) {      // reverse post-order: parents first
        const int k = T.lin_post[idx];
        const int par = T.parent[k];
        const double* wk = ws + (size_t)k * Bst + 2 * q;
        double x0 = wk[0], x1 = wk[1];
        if (par >= 0) {
            double h4[4];
            if (FUND)
                coupling_block_fund(M, Uall + so, Eall + so, k, par, T.e_up[k], h4);
            else
                coupling_block(M, Uall + so, Eall + so, q, k, par, T.e_up[k], h4);   // A(k, parent)
            const double* xp = xs + (size_t)par * Bst + 2 * q;
            const double t0 = fma(h4[1], xp[1], h4[0] * xp[0]), t1 = fma(h4[3], xp[1], h4[2] * xp[0]);
            const double* ik = linA + ((size_t)k * Hn + q) * 4;
            x0 -= fma(ik[1], t1, ik[0] * t0);
            x1 -= fma(ik[3], t1, ik[2] * t0);
        }
        double* xk = xs + (size_t)k * Bst + 2 * q;
        xk[0] = x0;
        xk[1] = x1;
        const int kst = q * n + k;
        if (kst >= 1) st[kst - 1] = x0;
        if (kst >= c) st[Nc + kst - c] = x1;
    }
}

// =============================================================================================================
// Contracted chains of pass-through buses (Tree::chain_*; linear bus, exactly one dense child): every block involved is
// harmonic-diagonal, so a chain is eliminated bottom-up in 2x2-per-harmonic algebra BEFORE the dense levels.  Eliminating
// bus k between the dense bus ch below it and its parent `up` leaves
//     D_ch += -A(ch,k) D_k^-1 A(k,ch),   y_ch += -A(ch,k) D_k^-1 y_k,
//     A'(ch,up) = -A(ch,k) D_k^-1 A(k,up),   A'(up,ch) = -A(up,k) D_k^-1 A(k,ch),
//     D_up += -A(up,k) D_k^-1 A(k,up),   y_up += -A(up,k) D_k^-1 y_k     (carried if `up` is the next chain bus; the dense bus
//                                                                          on top of the chain folds them like a linear child)
// One thread per (chain, harmonic, scenario).
// =============================================================================================================
__device__ __forceinline__ void mul22(const double a[4], const double b[4], double o[4]) {
    o[0] = fma(a[1], b[2], a[0] * b[0]);
    o[1] = fma(a[1], b[3], a[0] * b[1]);
    o[2] = fma(a[3], b[2], a[2] * b[0]);
    o[3] = fma(a[3], b[3], a[2] * b[1]);
}

template <int B>
int launch_factor_w(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    ScopedTimer t(h, T_SOLVE);
    hipLaunchKernelGGL((k_factor_w<B>), dim3((unsigned)count, (unsigned)h->cur_S), dim3(64), 0, h->cur_stream, h->M, T, nodes,
                       2 * h->Hn, h->N, h->Nc, active, h->d_U, h->d_E, h->d_f, h->d_Z, h->d_w, h->d_linA, h->d_pivflag, h->debug_ablate, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

template <int B>
int launch_back_w(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    hipLaunchKernelGGL((k_back_w<B>), dim3((unsigned)count, (unsigned)h->cur_S), dim3(64), 0, h->cur_stream, h->M, T, nodes,
                       2 * h->Hn, h->N, h->Nc, active, h->d_U, h->d_E, h->d_Z, h->d_w, h->d_x, h->d_f,
                       h->gj_mode ? h->d_H : nullptr, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

#include "hpf_quad.hpp"
#include "hpf_lin2x2.hpp"
#include "hpf_leafbatch.hpp"

// One launch per elimination level (b = 52 blocks, 256-thread workgroups): the scenario-batched workgroups of the level (lazy leaves
// at level 0, vector-only bordered buses above: 16 scenarios each) and its Gauss-Jordan / leaf workgroups (one per bus and scenario)
// are independent of each other, so they share ONE grid -- batched workgroups first (they live longest) -- instead of two
// dependent launches in the stream: the level costs the longer of the two lives, not their sum.  The three bodies carve their LDS
// from one buffer (the largest of the three).  kind: 0 no batched workgroups, 1 k_leaf_batch's, 2 k_sleaf_batch's.
#ifndef HPF_BATCH_XCD
#define HPF_BATCH_XCD 1    // k_level: the scenario-batched workgroups of a bus on one XCD (0: bus-fastest ids; A/B)
#endif
template <int B>
constexpr int level_lds() {
    constexpr int a = factor_q_lds<B>(), b2 = SLEAF_BATCH_LDS, c2 = LEAF_BATCH_LDS;
    return a > b2 ? (a > c2 ? a : c2) : (b2 > c2 ? b2 : c2);
}

template <int B>
__global__ __launch_bounds__(256, HPF_Q_OCC) void k_level(
    Model M, TreeDev T, const int* __restrict__ nodes, int kind, int nbatch, int ngen, int b, int N, int Nc,
    const int* __restrict__ active, int S_cnt, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
    const double* __restrict__ fall, double* __restrict__ Zall, double* __restrict__ wall, const double* __restrict__ linAall,
    double* __restrict__ Call, double* __restrict__ Hall, const cplx* __restrict__ I0all, const double* __restrict__ chG,
    const double* __restrict__ chH, const double* __restrict__ chD, const double* __restrict__ chy,
    const double* __restrict__ Minv, const double* __restrict__ lbimg, const double* __restrict__ sbimg, double* __restrict__ lfK,
    double* __restrict__ lfS, long long* __restrict__ dbg, int ablate, int s0, int* __restrict__ pivflag, double piv_limit,
    unsigned long long* __restrict__ tstamp) {
    // 256 threads: what the scenario-batched bodies are written for.  The factor body of a smaller block (b <= 28: 2 wavefronts, b <= 12:
    // one) runs on the first 64 * NT threads; the other wavefronts of such a workgroup end at once (a workgroup barrier waits for the
    // surviving wavefronts only).
    static_assert(64 * ((B + 16) / 16) <= 256, "k_level: blocks of up to 52 rows");
    __shared__ __attribute__((aligned(16))) double smem[level_lds<B>()];
    const int ytiles = (S_cnt + LB_SB - 1) / LB_SB;
#if HPF_BATCH_XCD
    // scenario-batched workgroups (homogeneous within a level): id -> (bus mod 8, scenario tile, bus / 8) -- every scenario tile of bus k on XCD
    // k mod 8, one after the other, in every scenario group alike: the bus's per-model operand images (27 - 33 KB) are fetched into ONE L2
    const int nbb = kind ? ((nbatch + 7) / 8) * 8 * ytiles : 0;
    if ((int)blockIdx.x < nbb) {
        const int t_ = (int)blockIdx.x >> 3;
        const int by = t_ % ytiles, bx = (t_ / ytiles) * 8 + ((int)blockIdx.x & 7);
        if (bx >= nbatch) return;
#else
    const int nbb = kind ? nbatch * ytiles : 0;
    if ((int)blockIdx.x < nbb) {
        const int bx = (int)blockIdx.x % nbatch, by = (int)blockIdx.x / nbatch;
#endif
        if (tstamp && threadIdx.x == 0) atomicMin(tstamp, (unsigned long long)wall_clock64());
        if (kind == 1)
            leaf_batch_body<B>(smem, bx, by, M, T, nodes, b, active, S_cnt, Uall, Eall, fall, wall, linAall, Call, Hall, chG, chH, chD, chy,
                               lbimg, lfK, lfS, s0);
        else
            sleaf_batch_body<B>(smem, bx, by, M, T, nodes, b, active, S_cnt, Uall, Eall, fall, wall, linAall, Call, Hall, I0all, chG, chH,
                                chD, chy, sbimg, Zall, lfK, lfS, s0, dbg, ablate);
        if (tstamp) {
            __syncthreads();
            if (threadIdx.x == 0) atomicMax(tstamp + 1, (unsigned long long)wall_clock64());
        }
    } else {
        // Blocks of 12 / 28: the factor body runs on NT < 4 wavefronts of this 256-thread workgroup and the others END here, before the
        // body's workgroup barriers.  That relies on the gfx9 barrier rule -- s_barrier releases when every wavefront of the workgroup that
        // has not terminated has arrived (s_endpgm takes a wavefront out of the count) --, not on anything HIP promises for a divergent
        // __syncthreads().  Only levels that have batched workgroups come here at these sizes (level_is_fused); covered on the GPU by
        // test_tree_build_variants_take_the_same_newton_steps (K = 5 / 13 against HPF_FUSELEVEL=0, bit for bit).
        if (64 * ((B + 16) / 16) < 256 && (int)threadIdx.x >= 64 * ((B + 16) / 16)) return;
        const int i = (int)blockIdx.x - nbb;
        factor_q_body<B, false>(FqLds<B>::carve(smem), i % ngen, i / ngen, M, T, nodes + FDESC * (size_t)nbatch, b, N, Nc, active, Uall, Eall, fall, Zall, wall,
                                linAall, Call, Hall, I0all, chG, chH, chD, chy, Minv, lfK, lfS, dbg, ablate, s0, pivflag, piv_limit, tstamp);
    }
}

template <int B>
int launch_level(hpf_handle* h, const TreeDev& T, const int* nodes, int kind, int nbatch, int ngen, const int* active) {
    ScopedTimer t(h, T_GJ);
    unsigned long long* ts = nullptr;                   // device-clock stamps of this launch (timing leg)
    if (h->timing && h->d_tstamp && h->ts_next < hpf_handle::TS_CAP) ts = h->d_tstamp + 2 * (size_t)(h->ts_next++);
    const Tree& tr = active_tree(h);
    const unsigned ytiles = (unsigned)((h->cur_S + LB_SB - 1) / LB_SB);
#if HPF_BATCH_XCD
    const unsigned grid = (kind ? (unsigned)(((nbatch + 7) / 8) * 8) * ytiles : 0u) + (unsigned)ngen * (unsigned)h->cur_S;
#else
    const unsigned grid = (kind ? (unsigned)nbatch * ytiles : 0u) + (unsigned)ngen * (unsigned)h->cur_S;
#endif
    if (grid == 0) return HPF_OK;
    hipLaunchKernelGGL((k_level<B>), dim3(grid), dim3(256), 0, h->cur_stream, h->M, T, nodes, kind, nbatch, ngen, 2 * h->Hn,
                       h->N, h->Nc, active, h->cur_S, h->d_U, h->d_E, h->d_fb, h->d_Z, h->d_w, h->d_linA, h->d_C, h->d_H, h->d_I0, h->d_chG,
                       h->d_chH, h->d_chD, h->d_chy, tr.d_Minv, tr.d_lbimg, tr.d_sbimg, h->d_lfK, h->d_lfS, h->d_dbg, h->debug_ablate,
                       h->cur_s0, h->d_pivflag, h->piv_limit, ts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

// padded block size of the multi-wave MFMA path: 12 / 28 / 52, and 100 for 52 < b <= 100 (K <= 49: BASELINE config 5; the general
// Gauss-Jordan kernel only -- no constant-inverse leaves / lazy leaves / super-leaves there yet); 0: the 256-thread generic kernels
int wave_block_size(int b) { return b <= 12 ? 12 : (b <= 28 ? 28 : (b <= 52 ? 52 : (b <= 100 ? 100 : 0))); }

template <class T>
int upload(hpf_handle* h, T** dst, const std::vector<T>& v) {
    const size_t cnt = v.empty() ? 1 : v.size();
    hipError_t e = hipMalloc((void**)dst, cnt * sizeof(T));
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return e == hipErrorOutOfMemory ? HPF_E_NOMEM : HPF_E_HIP;
    }
    if (!v.empty()) {
        e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            h->last_detail = (int)e;
            return HPF_E_HIP;
        }
    }
    return HPF_OK;
}

size_t factor_lds_bytes(int b) { return gj_dense_lds_bytes(b); }

template <int R>
int launch_factor(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    ScopedTimer t(h, T_SOLVE);
    const int b = 2 * h->Hn;
    const size_t lds = factor_lds_bytes(b);
    if (lds > 64 * 1024) {
        // (per launch, like launch_mismatch: the attribute belongs to the device the handle runs on -- a process-wide "already set" flag
        //  would skip it for a second device -- and the call costs nothing next to the launch)
        if (lds > 160 * 1024) return HPF_E_ARG;
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_factor<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) {
            h->last_detail = (int)ea;
            return HPF_E_HIP;
        }
    }
    hipLaunchKernelGGL((k_tree_factor<R>), dim3((unsigned)count, (unsigned)h->cur_S), dim3(256), lds, h->cur_stream, h->M, T,
                       nodes, b, h->N, h->Nc, active, h->d_U, h->d_E, h->d_f, h->d_Z, h->d_w, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}


// Back sweep, one launch per depth (round 4): the scenario-batched workgroups of the depth's bordered buses and constant-inverse leaves (16 scenarios
// each: k_sleaf_back_batch / k_leaf_back_batch bodies) share the grid with the per-scenario workgroups of its Gauss-Jordan buses (k_back_q body) --
// a bordered bus or a leaf needs its dense parent's x only, so it belongs to its own depth instead of waiting for the last one (four more dependent
// launches behind the depths' before).  Blocks of 52 only (the bodies of smaller blocks run on fewer wavefronts than the batched ones).
template <int B>
__global__ __launch_bounds__(256) void k_level_back(
    Model M, TreeDev T, const int* __restrict__ sl_nodes, int n_sl, const int* __restrict__ lf_nodes, int n_lf, const int* __restrict__ gj_nodes, int n_gj,
    int b, int N, int Nc, const int* __restrict__ active, int S_cnt, const double* __restrict__ Zall, const double* __restrict__ wall,
    double* __restrict__ xall, const double* __restrict__ Hall, const double* __restrict__ Minv, const double* __restrict__ lbimg,
    const double* __restrict__ sbimg, const double* __restrict__ lzimg, const double* __restrict__ lfK, const double* __restrict__ lfS, int s0) {
    static_assert(64 * ((B + 16) / 16) == 256, "k_level_back: blocks of 52 rows (four wavefronts in every body)");
    const int ytiles = (S_cnt + LB_SB - 1) / LB_SB;
    const int bid = (int)blockIdx.x;
#if HPF_BATCH_XCD
    // (the scenario-batched workgroups as in k_level: every scenario tile of a bus on XCD (bus mod 8), one after the other)
    const int nb_sl = ((n_sl + 7) / 8) * 8 * ytiles, nb_lf = ((n_lf + 7) / 8) * 8 * ytiles;
    if (bid < nb_sl) {
        const int t_ = bid >> 3, bx = (t_ / ytiles) * 8 + (bid & 7);
        if (bx >= n_sl) return;
        sleaf_back_batch_body<B>(bx, t_ % ytiles, M, sl_nodes, b, active, S_cnt, wall, xall, Hall, sbimg, lzimg, Zall, lfS, s0);
    } else if (bid < nb_sl + nb_lf) {
        const int i = bid - nb_sl, t_ = i >> 3, bx = (t_ / ytiles) * 8 + (i & 7);
        if (bx >= n_lf) return;
        leaf_back_batch_body<B>(bx, t_ % ytiles, M, lf_nodes, b, active, S_cnt, wall, xall, Hall, lbimg, lfK, lfS, s0);
    } else {
#else
    const int nb_sl = n_sl * ytiles, nb_lf = n_lf * ytiles;
    if (bid < nb_sl) {
        sleaf_back_batch_body<B>(bid % n_sl, bid / n_sl, M, sl_nodes, b, active, S_cnt, wall, xall, Hall, sbimg, lzimg, Zall, lfS, s0);
    } else if (bid < nb_sl + nb_lf) {
        const int i = bid - nb_sl;
        leaf_back_batch_body<B>(i % n_lf, i / n_lf, M, lf_nodes, b, active, S_cnt, wall, xall, Hall, lbimg, lfK, lfS, s0);
    } else {
#endif
        const int i = bid - nb_sl - nb_lf;
        back_q_body<B>(i % n_gj, i / n_gj, M, T, gj_nodes, b, N, Nc, active, Zall, wall, xall, (double*)nullptr, Hall, Minv, lfK, lfS, s0);
    }
}

template <int B>
int launch_level_back(hpf_handle* h, const TreeDev& T, const int* sl_nodes, int n_sl, const int* lf_nodes, int n_lf, const int* gj_nodes, int n_gj,
                      const int* active) {
    const Tree& tr = active_tree(h);
    const unsigned ytiles = (unsigned)((h->cur_S + LB_SB - 1) / LB_SB);
#if HPF_BATCH_XCD
    const unsigned grid = (unsigned)(((n_sl + 7) / 8) * 8 + ((n_lf + 7) / 8) * 8) * ytiles + (unsigned)n_gj * (unsigned)h->cur_S;
#else
    const unsigned grid = (unsigned)(n_sl + n_lf) * ytiles + (unsigned)n_gj * (unsigned)h->cur_S;
#endif
    if ((unsigned)(n_sl + n_lf) * ytiles + (unsigned)n_gj * (unsigned)h->cur_S == 0) return HPF_OK;
    hipLaunchKernelGGL((k_level_back<B>), dim3(grid), dim3(256), 0, h->cur_stream, h->M, T, sl_nodes, n_sl, lf_nodes, n_lf, gj_nodes, n_gj, 2 * h->Hn,
                       h->N, h->Nc, active, h->cur_S, h->d_Z, h->d_w, h->d_x, h->d_H, tr.d_Minv, tr.d_lbimg, tr.d_sbimg, tr.d_lzimg, h->d_lfK, h->d_lfS,
                       h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

// =============================================================================================================
// Meshed networks on the block-tree path: "bordered Newton step".
// The admittance pattern = BFS spanning tree + k loop-closing lines (ties).  With J_t = the Jacobian without the ties' off-
// diagonal blocks (their admittances stay on the diagonals: J_t's bus blocks are J's), J = J_t + E_T Q^T, where T = the ties'
// endpoint buses, E_T = identity columns of their equations (m = |T| 2Hn) and row (i, l) of Q^T = sum over ties (i,j) of
// A(i,j)[l, :] E_j^T -- harmonic-diagonal 2x2 blocks like every off-diagonal block.  Then (Woodbury in its bordered form)
//     y = J_t^-1 f,  Z = J_t^-1 E_T,  (I + Q^T Z) g = Q^T y,  x = y - Z g.
// y and the m columns of Z are 1 + m right-hand sides of the SAME tree system: they run as 1 + m virtual scenarios (same
// voltages, different mismatch images) through the unchanged block-tree kernels -- every kernel stays as tested; the m x m
// border system goes to rocSOLVER.  Cost: (1 + m) x one scenario's step per real scenario (m <= 1024).
// =============================================================================================================
// virtual slot v0 + j <- state of real scenario r, j = blockIdx.y; mismatch image of right-hand side v = vfirst + j: v = 0 the scenario's
// own, v >= 1 the unit vector of border row v - 1; with g (second pass): the scenario's own minus E_T g
__global__ __launch_bounds__(256) void k_border_prepare(int n, int Hn, int Bst, int b, int r, int v0, int vfirst, const int* __restrict__ tb_bus,
                                                        int n_tb, const double* __restrict__ g, cplx* __restrict__ U, cplx* __restrict__ E,
                                                        cplx* __restrict__ I0, double* __restrict__ fb) {
    const int j = blockIdx.y, v = vfirst + j;
    const size_t dst = (size_t)(v0 + j);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n * Hn) {
        U[dst * n * Hn + t] = U[(size_t)r * n * Hn + t];
        E[dst * n * Hn + t] = E[(size_t)r * n * Hn + t];
    }
    if (t < n) I0[dst * n + t] = I0[(size_t)r * n + t];
    if (t < n * Bst) {
        double val = 0.0;
        if (v == 0) {
            val = fb[(size_t)r * n * Bst + t];
            if (g) {                                           // second pass: f - E_T g
                const int bus = t / Bst, l = t - bus * Bst;
                if (l < b)
                    for (int a = 0; a < n_tb; ++a)
                        if (tb_bus[a] == bus) val -= g[a * b + l];
            }
        } else {
            const int a = (v - 1) / b, l = (v - 1) - a * b;
            if (t == tb_bus[a] * Bst + l) val = 1.0;
        }
        fb[dst * n * Bst + t] = val;
    }
}

// border system: M[row, col] = delta + (Q^T z_col)[row], rhs[row] = (Q^T y)[row]; row = (endpoint a, local row l), right-hand side
// v = vfirst + blockIdx.y sits in virtual slot v0 + blockIdx.y: v = 0 -> y (the border system's right-hand side), v >= 1 -> column v - 1
__global__ __launch_bounds__(256) void k_border_build(Model M, int Bst, int b, int r, int v0, int vfirst, int m, const int* __restrict__ tb_bus,
                                                      const int* __restrict__ tb_ptr, const int* __restrict__ tb_adj,
                                                      const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                      const double* __restrict__ xall, double* __restrict__ bM, double* __restrict__ brhs) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int col = vfirst + blockIdx.y;               // 0: right-hand side, 1 + c: column c of the border matrix
    if (row >= m) return;
    const int a = row / b, l = row - a * b, q = l >> 1, t = l & 1;
    const int i = tb_bus[a];
    const size_t so = (size_t)r * M.n * M.Hn;
    const double* x = xall + (size_t)(v0 + blockIdx.y) * M.n * Bst;
    double acc = 0.0;
    for (int e = tb_ptr[a]; e < tb_ptr[a + 1]; ++e) {
        const int j = tb_adj[3 * e], ent = tb_adj[3 * e + 1];
        double g4[4];
        coupling_block(M, Uall + so, Eall + so, q, i, j, ent, g4);          // A(i, j) at harmonic position q, masked
        acc = fma(g4[t * 2], x[(size_t)j * Bst + 2 * q], acc);
        acc = fma(g4[t * 2 + 1], x[(size_t)j * Bst + 2 * q + 1], acc);
    }
    if (col == 0)
        brhs[row] = acc;
    else
        bM[(size_t)(col - 1) * m + row] = acc + (row == col - 1 ? 1.0 : 0.0);
}

// x(real scenario r) <- x(virtual slot v0) (second pass: J_t^-1 (f - E_T g)); the virtual sweeps' static-pivot flags fold into r's
__global__ __launch_bounds__(256) void k_border_finish(int count, int r, int v0, int nv, double* __restrict__ xall, int* __restrict__ pivflag) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < count) xall[(size_t)r * count + t] = xall[(size_t)v0 * count + t];
    if (t < nv && pivflag[v0 + t]) {
        atomicOr(pivflag + r, pivflag[v0 + t]);
        pivflag[v0 + t] = 0;
    }
}

}  // namespace

#include "hpf_tree_plan.hpp"      // the host-side planner: tree_build_into, tree_plan_dump, tree_build

namespace hpf {

// the batched workgroups of elimination level l (lazy leaves at level 0, vector-only bordered buses above: 16 scenarios each);
// kind 1 / 2 / 0 (none) -- the one place the factor sweep and the census take it from
static int level_batched(const hpf_handle* h, const Tree& T, int l, int* kind) {
    int nbatch = (l == 0 && h->leafbatch && h->has_ctree && T.lvl_all_leaf[0]) ? T.n_lazy_level0 : 0;
    const bool slbatch = l > 0 && h->leafbatch && h->has_ctree && l < (int)T.lvl_nbatch.size() && T.lvl_nbatch[l] > 0;
    if (slbatch) nbatch = T.lvl_nbatch[l];
    if (kind) *kind = nbatch > 0 ? (slbatch ? 2 : 1) : 0;
    return nbatch;
}

// elimination level l of the current mode is ONE launch of k_level<BW> (else: k_leaf_batch / k_sleaf_batch + k_factor_q, or k_factor_q
// alone): blocks of 52 always, smaller blocks only where the level has batched workgroups to put next to the per-scenario ones
static bool level_is_fused(const hpf_handle* h, const Tree& T, int l, int BW) {
    if (h->gj_mode != 1 || !h->fuse_levels || (BW != 12 && BW != 28 && BW != 52)) return false;
    return BW == 52 || level_batched(h, T, l, nullptr) > 0;
}

// hpf_tree_census[9] / hpf_kernel_model: every elimination level of the factor sweep is one k_level launch
bool tree_levels_fused(hpf_handle* h) {
    if (h->solver != HPF_SOLVER_BLOCK_TREE) return false;
    const Tree& T = active_tree(h);
    const int BW = wave_block_size(2 * h->Hn);
    if (T.n_levels == 0) return false;
    for (int l = 0; l < T.n_levels; ++l)
        if (T.lvl_ptr[l + 1] > T.lvl_ptr[l] && !level_is_fused(h, T, l, BW)) return false;
    return true;
}

// the tree the Newton step of the current mode runs on
Tree& active_tree(hpf_handle* h) { return (h->has_ctree && h->gj_mode == 1) ? h->ctree : h->tree; }

static void tree_free_one(Tree& T);
static void tree_free_one_fwd(Tree& T) { tree_free_one(T); }
static void tree_free_one(Tree& T) {
    void* ptrs[] = {T.d_parent, T.d_lvl_nodes, T.d_dep_nodes, T.d_child_ptr, T.d_child, T.d_e_up, T.d_e_dn,
                    T.d_child_mid, T.d_lin, T.d_lin_ptr, T.d_lin_post, T.d_all_ptr, T.d_all_post, T.d_fdesc, T.d_child3,
                    T.d_bdesc, T.d_dchild, T.d_chain_ptr, T.d_chain_nodes, T.d_chain_ch, T.d_Minv, T.d_lrec, T.d_crec, T.d_cnode, T.d_arec,
                    T.d_lzrec, T.d_lzimg, T.d_lbimg, T.d_bleaf, T.d_bsleaf, T.d_bsleaf_dep, T.d_sbimg, T.d_lbrec, T.d_lbptr, T.d_lb2rec, T.d_lb2x, T.d_lb2ptr, T.d_lb2cptr, T.d_lb2clist,
                    T.d_comp_child};
    for (void* p : ptrs)
        if (p) hipFree(p);
}

void tree_free(hpf_handle* h) {
    tree_free_one(h->tree);
    tree_free_one(h->ctree);
    void* bp[] = {h->d_tb_bus, h->d_tb_ptr, h->d_tb_adj, h->d_bM, h->d_brhs, h->d_bipiv, h->d_binfo, h->d_bM0, h->d_brhs0,
                  h->d_sel_P, h->d_sel_pidx, h->d_sel_toff, h->d_sel_S, h->d_sel_Z, h->d_sel_Up, h->d_sel_W, h->d_sel_X, h->d_sel_tie, h->d_sel_jobs,
                  h->d_sel_hl, h->d_sel_slot, h->d_sel_cptr, h->d_sel_clist, h->d_sel_dw, h->d_bB, h->d_bgj_jobs,
                  h->d_sel_bM, h->d_sel_rhs, h->d_sel_g, h->d_sel_res, h->d_sel_info};
    for (void* q2 : bp)
        if (q2) hipFree(q2);
}

int tree_alloc_scenarios(hpf_handle* h) {
    const int bw = wave_block_size(2 * h->Hn);
    const size_t b = bw ? (size_t)bw : 2 * (size_t)h->Hn, S = h->S_alloc, n = h->n;
    hipError_t e;
    const size_t ct = bw ? (size_t)(((bw + 16) / 16) * ((bw + 16) / 16) * 256) : 0;     // accumulator-tile image of a block
    if ((e = hipMalloc((void**)&h->d_Z, sizeof(double) * S * n * (b * b > ct ? b * b : ct))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_w, sizeof(double) * S * n * b)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_x, sizeof(double) * S * n * b)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_linA, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_H, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_fb, sizeof(double) * S * n * b)) != hipSuccess ||
        (h->has_ctree && ((e = hipMalloc((void**)&h->d_chG, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess ||
                          (e = hipMalloc((void**)&h->d_chH, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess ||
                          (e = hipMalloc((void**)&h->d_chD, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess ||
                          (e = hipMalloc((void**)&h->d_chy, sizeof(double) * S * n * (size_t)h->Hn * 2)) != hipSuccess ||
                          (e = hipMalloc((void**)&h->d_chZ, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess ||
                          (e = hipMalloc((void**)&h->d_lfK, sizeof(double) * S * n * 12)) != hipSuccess ||
                          (e = hipMalloc((void**)&h->d_lfS, sizeof(double) * S * n * (size_t)h->Hn * 4)) != hipSuccess)) ||
        ((h->debug_ablate & 16) && (e = hipMalloc((void**)&h->d_dbg, sizeof(long long) * S * n * 8)) != hipSuccess) ||
        (bw && (e = hipMalloc((void**)&h->d_C, sizeof(double) * S * n * (size_t)(((bw + 16) / 16) * ((bw + 16) / 16) * 256))) != hipSuccess) ||
        (h->has_ctree && h->ctree.n_comp > 0 &&
         ((e = hipMalloc((void**)&h->d_F, sizeof(double) * S * (size_t)h->ctree.n_comp * 3 * ct)) != hipSuccess ||
          (e = hipMalloc((void**)&h->d_H2, sizeof(double) * S * (size_t)h->ctree.n_comp * (size_t)h->Hn * 4)) != hipSuccess))) {
        h->last_detail = (int)e;
        return e == hipErrorOutOfMemory ? HPF_E_NOMEM : HPF_E_HIP;
    }
    if (h->d_dbg) hipMemset(h->d_dbg, 0, sizeof(long long) * S * n * 8);
    return HPF_OK;
}


// Fundamental power-flow Newton step on a radial network (pf, HG:244-275): every bus is a plain power bus there, so the whole
// tree is one "linear subtree" at harmonic position 0 -> the same 2x2 elimination, one thread per scenario walking the post-order.
int tree_fund_step(hpf_handle* h, bool only_active) {
    Tree& T = h->tree;
    const int* active = only_active ? h->d_active : nullptr;
    const TreeDev td{T.d_parent, T.d_child_ptr, T.d_child, T.d_e_up, T.d_e_dn, T.d_child_mid, T.d_all_ptr, T.d_all_post, T.d_child3, T.d_dchild, T.d_chain_ptr, T.d_chain_nodes, T.d_chain_ch};
    const int b = 2 * h->Hn;
    const int BW = wave_block_size(b);
    const int Bst = BW ? BW : b;
    ScopedTimer t(h, T_SOLVE);
    if (BW) {
        // level-parallel: one launch per height of the tree, one thread per (bus, scenario) (k_lin_level_*, fund = 1)
        for (int hh = 0; hh < T.n_all_heights; ++hh) {
            const int cnt = T.ah_ptr[hh + 1] - T.ah_ptr[hh];
            hipLaunchKernelGGL(k_lin_level_factor, dim3((unsigned)((cnt + 127) / 128), (unsigned)h->cur_S), dim3(128), 0,
                               h->cur_stream, h->M, td, T.d_arec + 8 * (size_t)T.ah_ptr[hh], cnt, h->Nf, h->n - 1, Bst, active, h->d_U,
                               h->d_E, h->d_f, h->d_linA, h->d_w, h->d_I0, 1, h->cur_s0);
        }
        for (int hh = T.n_all_heights - 1; hh >= 0; --hh) {
            const int cnt = T.ah_ptr[hh + 1] - T.ah_ptr[hh];
            hipLaunchKernelGGL(k_lin_level_back, dim3((unsigned)((cnt + 127) / 128), (unsigned)h->cur_S), dim3(128), 0, h->cur_stream,
                               h->M, td, T.d_arec + 8 * (size_t)T.ah_ptr[hh], cnt, h->Nf, h->n - 1, Bst, active, h->d_U, h->d_E,
                               h->d_linA, h->d_w, h->d_x, h->d_f, 1, h->cur_s0);
        }
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) {
            h->last_detail = (int)e2;
            return HPF_E_HIP;
        }
        return HPF_OK;
    }
    hipLaunchKernelGGL((k_lin_factor<true>), dim3(1, (unsigned)h->cur_S), dim3(128), 0, h->cur_stream, h->M, td, 1, h->Nf,
                       h->n - 1, Bst, active, h->d_U, h->d_E, h->d_f, h->d_linA, h->d_w, nullptr, h->cur_s0);
    hipLaunchKernelGGL((k_lin_back<true>), dim3(1, (unsigned)h->cur_S), dim3(128), 0, h->cur_stream, h->M, td, 1, h->Nf,
                       h->n - 1, Bst, active, h->d_U, h->d_E, h->d_linA, h->d_w, h->d_x, h->d_f, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

int tree_newton_step(hpf_handle* h, bool only_active) {
    Tree& T = active_tree(h);
    const int* active = only_active ? h->d_active : nullptr;
    const TreeDev td{T.d_parent, T.d_child_ptr, T.d_child, T.d_e_up, T.d_e_dn, T.d_child_mid, T.d_lin_ptr, T.d_lin_post, T.d_child3, T.d_dchild, T.d_chain_ptr, T.d_chain_nodes, T.d_chain_ch, T.d_lzrec, T.d_lzimg,
                     h->d_F, h->d_H2, T.d_comp_child, T.n_comp};
    const int b = 2 * h->Hn;
    const int R = (b + 15) / 16;
    const int BW = wave_block_size(b);
    const int lin_threads = T.n_lin_roots * h->Hn;
    const int Bst = BW ? BW : b;
    if (!h->tree_back_only) {                                        // (factor-once bordered step: its second pass is a back sweep alone)
        const bool lvl2x2 = h->has_ctree && h->gj_mode == 1;         // level-parallel 2x2 kernels (records of the contracted tree)
        if (lvl2x2) {
            if (T.n_lin_bundles2 > 0) {                            // ... with one memory round trip (k_lin_bundle_factor)
                const dim3 g2((unsigned)T.n_lin_bundles2, (unsigned)h->cur_S);
#define HPF_LB_F(NP_)                                                                                                                  \
    hipLaunchKernelGGL((k_lin_bundle_factor<NP_>), g2, dim3(256), 0, h->cur_stream, h->M, T.d_lb2rec, (const int2*)T.d_lb2x, T.d_lb2ptr, \
                       T.n_lin_heights, Bst, active, h->d_U, h->d_E, h->d_fb, h->d_linA, h->d_w, h->d_I0, h->cur_s0, td,                   \
                       T.chains_bundled ? T.d_lb2cptr : (const int*)nullptr, T.d_lb2clist, T.d_crec, T.d_cnode, h->d_chG, h->d_chH, h->d_chD, \
                       h->d_chy, h->d_chZ)
                if (T.lin_np == 1) HPF_LB_F(1);
                else if (T.lin_np == 2) HPF_LB_F(2);
                else HPF_LB_F(4);
#undef HPF_LB_F
            } else if (T.n_lin_bundles > 0)                        // every height of the all-linear subtrees in one launch
                hipLaunchKernelGGL(k_lin_tree_factor, dim3((unsigned)T.n_lin_bundles, (unsigned)h->cur_S), dim3(256), 0, h->cur_stream,
                                   h->M, td, T.d_lbrec, T.d_lbptr, T.n_lin_heights, h->N, h->Nc, Bst, active, h->d_U, h->d_E,
                                   h->d_fb, h->d_linA, h->d_w, h->d_I0, h->cur_s0);
            for (int hh = 0; hh < T.n_lin_heights && T.n_lin_bundles == 0 && T.n_lin_bundles2 == 0; ++hh) {
                const int cnt = T.lh_ptr[hh + 1] - T.lh_ptr[hh];
                if (cnt == 0) continue;
                hipLaunchKernelGGL(k_lin_level_factor, dim3((unsigned)((cnt * h->Hn + 127) / 128), (unsigned)h->cur_S), dim3(128), 0,
                                   h->cur_stream, h->M, td, T.d_lrec + 8 * (size_t)T.lh_ptr[hh], cnt, h->N, h->Nc, Bst, active,
                                   h->d_U, h->d_E, h->d_fb, h->d_linA, h->d_w, h->d_I0, 0, h->cur_s0);
            }
            if (T.n_chains > 0 && !(T.chains_bundled && T.n_lin_bundles2 > 0))
                hipLaunchKernelGGL(k_chain_factor2, dim3((unsigned)((T.n_chains * h->Hn + 127) / 128), (unsigned)h->cur_S), dim3(128),
                                   0, h->cur_stream, h->M, td, T.d_crec, T.d_cnode, T.n_chains, h->N, h->Nc, Bst, active, h->d_U,
                                   h->d_E, h->d_fb, h->d_linA, h->d_w, h->d_I0, h->d_chG, h->d_chH, h->d_chD, h->d_chy, h->d_chZ,
                                   h->cur_s0);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                h->last_detail = (int)e;
                return HPF_E_HIP;
            }
        }
        if (!lvl2x2 && lin_threads > 0) {
            hipLaunchKernelGGL((k_lin_factor<false>), dim3((unsigned)((lin_threads + 127) / 128), (unsigned)h->cur_S), dim3(128), 0,
                               h->cur_stream, h->M, td, T.n_lin_roots, h->N, h->Nc, Bst, active, h->d_U, h->d_E, h->d_f,
                               h->d_linA, h->d_w, h->d_I0, h->cur_s0);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                h->last_detail = (int)e;
                return HPF_E_HIP;
            }
        }
        for (int l = 0; l < T.n_levels; ++l) {
            const int cnt = T.lvl_ptr[l + 1] - T.lvl_ptr[l];
            if (cnt == 0) continue;
            const int* nodes = T.d_lvl_nodes + T.lvl_ptr[l];
            int r;
            // level 0 of the contracted tree: its lazy leaves come first and go 16 scenarios per workgroup (k_leaf_batch)
            int bkind = 0;                                      // (h->leafbatch == 0: one workgroup per (leaf, scenario), no batched ones)
            const int nbatch = level_batched(h, T, l, &bkind);
            const bool slbatch = bkind == 2;
            switch (BW) {                       // (timing spans: one per kernel launch, inside the launch helpers)
#define HPF_FACTOR_CASE(BB_)                                                                                  \
    case BB_:                                                                                                 \
        if (level_is_fused(h, T, l, BB_)) {                 /* one launch per level: batched and per-scenario workgroups                          \
                                                      side by side (small blocks without batched workgroups: k_factor_q's own grid and LDS) */ \
            r = launch_level<BB_>(h, td, T.d_fdesc + FDESC * (size_t)T.lvl_ptr[l], nbatch > 0 ? (slbatch ? 2 : 1) : 0, nbatch,       \
                                  cnt - nbatch, active);                                                      \
            break;                                                                                            \
        }                                                                                                     \
        if (h->gj_mode == 1 && nbatch > 0) {                                                                  \
            r = slbatch ? launch_sleaf_batch<BB_>(h, td, T.d_fdesc + FDESC * (size_t)T.lvl_ptr[l], nbatch, active) \
                        : launch_leaf_batch<BB_>(h, td, T.d_fdesc + FDESC * (size_t)T.lvl_ptr[l], nbatch, active); \
            if (!r && cnt > nbatch)                                                                           \
                r = launch_factor_q<BB_>(h, td, T.d_fdesc + FDESC * (size_t)(T.lvl_ptr[l] + nbatch), cnt - nbatch, active, !slbatch); \
            break;                                                                                            \
        }                                                                                                     \
        r = h->gj_mode == 1 ? launch_factor_q<BB_>(h, td, T.d_fdesc + FDESC * (size_t)T.lvl_ptr[l], cnt, active, \
                                                  T.lvl_all_leaf[l] != 0)                                       \
                            : launch_factor_w<BB_>(h, td, nodes, cnt, active);                               \
        break
                HPF_FACTOR_CASE(12);
                HPF_FACTOR_CASE(28);
                HPF_FACTOR_CASE(52);
#undef HPF_FACTOR_CASE
                case 100:       // 52 < b <= 100: general multi-wave kernel for every dense bus; pivoted mode = generic kernels below
                    if (h->gj_mode == 1) {
                        r = launch_factor_q<100>(h, td, T.d_fdesc + FDESC * (size_t)T.lvl_ptr[l], cnt, active, T.lvl_all_leaf[l] != 0);
                        break;
                    }
                    [[fallthrough]];
                default:
                    switch (R) {
                        case 4: r = launch_factor<4>(h, td, nodes, cnt, active); break;
                        case 5: r = launch_factor<5>(h, td, nodes, cnt, active); break;
                        case 6: r = launch_factor<6>(h, td, nodes, cnt, active); break;
                        case 7: r = launch_factor<7>(h, td, nodes, cnt, active); break;
                        default: return HPF_E_ARG;
                    }
            }
            if (r) return r;
        }
    }
    ScopedTimer tb(h, T_BACK);
    // one launch per depth for all bus kinds (k_level_back) where the batched bodies exist and every body has four wavefronts
    // (groups of up to 32 scenarios only: larger launches are throughput-bound, and k_back_q alone runs at twice the
    //  occupancy of the fused kernel -- 5.2 vs 5.4 ms per step at 1 024 scenarios, 0.297 vs 0.288 ms at one)
    bool fused_back = h->fuse_back && h->cur_S <= h->fuse_back_max && h->leafbatch && h->has_ctree && h->gj_mode == 1 && BW == 52 &&
                      (int)T.bsl_dep_ptr.size() == T.n_depths + 1 && (int)T.bleaf_dep_ptr.size() == T.n_depths + 1 && (int)T.dep_nleaf.size() >= T.n_depths;
    for (int dl = 1; fused_back && dl < T.n_depths; ++dl)         // (a depth's batched records are exactly its leaves + bordered buses)
        fused_back = T.bsl_dep_ptr[dl + 1] - T.bsl_dep_ptr[dl] + T.bleaf_dep_ptr[dl + 1] - T.bleaf_dep_ptr[dl] == T.dep_nleaf[dl];
    if (fused_back && (T.bsl_dep_ptr[1] != 0 || T.bleaf_dep_ptr[1] != 0)) fused_back = false;
    for (int dl = 0; dl < T.n_depths; ++dl) {
        const int cnt = T.dep_ptr[dl + 1] - T.dep_ptr[dl];
        if (cnt == 0) continue;
        const int* nodes = T.d_dep_nodes + T.dep_ptr[dl];
        int r = HPF_OK;
        const int leafbatch_b = h->leafbatch;
        const int nlb = (leafbatch_b && h->has_ctree && dl > 0 && dl < (int)T.dep_nleaf.size()) ? T.dep_nleaf[dl] : 0;
        if (fused_back) {
            const int n_sl = dl > 0 ? T.bsl_dep_ptr[dl + 1] - T.bsl_dep_ptr[dl] : 0, n_lf = dl > 0 ? T.bleaf_dep_ptr[dl + 1] - T.bleaf_dep_ptr[dl] : 0;
            if ((r = launch_level_back<52>(h, td, T.d_bsleaf_dep + 8 * (size_t)T.bsl_dep_ptr[dl], n_sl, T.d_bleaf + 4 * (size_t)T.bleaf_dep_ptr[dl], n_lf,
                                           T.d_bdesc + 4 * (size_t)(T.dep_ptr[dl] + nlb), cnt - nlb, active)))
                return r;
            continue;
        }
        switch (BW) {
            case 12:
                if (h->gj_mode == 1 && nlb > 0) {                    // (the leaves of this depth wait for the one batched launch below)
                    if (cnt > nlb) r = launch_back_q<12>(h, td, T.d_bdesc + 4 * (size_t)(T.dep_ptr[dl] + nlb), cnt - nlb, active);
                    break;
                }
                r = h->gj_mode == 1 ? launch_back_q<12>(h, td, T.d_bdesc + 4 * (size_t)T.dep_ptr[dl], cnt, active) : launch_back_w<12>(h, td, nodes, cnt, active);
                break;
            case 28:
                if (h->gj_mode == 1 && nlb > 0) {                    // (the leaves of this depth wait for the one batched launch below)
                    if (cnt > nlb) r = launch_back_q<28>(h, td, T.d_bdesc + 4 * (size_t)(T.dep_ptr[dl] + nlb), cnt - nlb, active);
                    break;
                }
                r = h->gj_mode == 1 ? launch_back_q<28>(h, td, T.d_bdesc + 4 * (size_t)T.dep_ptr[dl], cnt, active) : launch_back_w<28>(h, td, nodes, cnt, active);
                break;
            case 52:
                if (h->gj_mode == 1 && nlb > 0) {                    // (the leaves of this depth wait for the one batched launch below)
                    if (cnt > nlb) r = launch_back_q<52>(h, td, T.d_bdesc + 4 * (size_t)(T.dep_ptr[dl] + nlb), cnt - nlb, active);
                    break;
                }
                r = h->gj_mode == 1 ? launch_back_q<52>(h, td, T.d_bdesc + 4 * (size_t)T.dep_ptr[dl], cnt, active) : launch_back_w<52>(h, td, nodes, cnt, active);
                break;
            case 100:
                if (h->gj_mode == 1) {
                    r = launch_back_q<100>(h, td, T.d_bdesc + 4 * (size_t)T.dep_ptr[dl], cnt, active);
                    break;
                }
                [[fallthrough]];
            default: {
                hipLaunchKernelGGL(k_tree_back, dim3((unsigned)cnt, (unsigned)h->cur_S), dim3(256), 0, h->cur_stream, h->n, h->c,
                                   h->Hn, td, nodes, b, h->N, h->Nc, active, h->d_Z, h->d_w, h->d_x, h->d_f, h->cur_s0);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) {
                    h->last_detail = (int)e;
                    return HPF_E_HIP;
                }
            }
        }
        if (r) return r;
    }
    {
        // every constant-inverse leaf at once, 16 scenarios per workgroup: a leaf's x needs its parent's only, and nothing of the
        // dense tree hangs below a leaf (the 2x2 kernels that do come next)
        const int leafbatch_e = h->leafbatch && !fused_back;
        if (leafbatch_e && h->has_ctree && h->gj_mode == 1 && T.n_bsleaf > 0) {    // super-leaves first: leaves hang below them
            int r = HPF_OK;
            for (size_t gi = 0; gi + 1 < T.bsleaf_ptr.size() && !r; ++gi) {      // nested bordered buses first (by nesting order)
                const int b0 = T.bsleaf_ptr[gi], bc2 = T.bsleaf_ptr[gi + 1] - b0;
                if (bc2 <= 0) continue;
                switch (BW) {
                    case 12: r = launch_sleaf_back_batch<12>(h, T.d_bsleaf + 8 * (size_t)b0, bc2, active); break;
                    case 28: r = launch_sleaf_back_batch<28>(h, T.d_bsleaf + 8 * (size_t)b0, bc2, active); break;
                    case 52: r = launch_sleaf_back_batch<52>(h, T.d_bsleaf + 8 * (size_t)b0, bc2, active); break;
                    default: break;
                }
            }
            if (r) return r;
        }
        if (leafbatch_e && h->has_ctree && h->gj_mode == 1 && T.n_bleaf > 0) {
            int r = HPF_OK;
            switch (BW) {
                case 12: r = launch_leaf_back_batch<12>(h, T.d_bleaf, T.n_bleaf, active); break;
                case 28: r = launch_leaf_back_batch<28>(h, T.d_bleaf, T.n_bleaf, active); break;
                case 52: r = launch_leaf_back_batch<52>(h, T.d_bleaf, T.n_bleaf, active); break;
                default: break;
            }
            if (r) return r;
        }
    }
    const bool lvl2x2b = h->has_ctree && h->gj_mode == 1;
    if (lvl2x2b) {
        if (T.n_chains > 0 && !(T.chains_bundled && T.n_lin_bundles2 > 0))
            hipLaunchKernelGGL(k_chain_back2, dim3((unsigned)((T.n_chains * h->Hn + 127) / 128), (unsigned)h->cur_S), dim3(128), 0,
                               h->cur_stream, h->M, td, T.d_crec, T.d_cnode, T.n_chains, h->N, h->Nc, Bst, active, h->d_U, h->d_E,
                               h->d_linA, h->d_w, h->d_x, (double*)nullptr, h->d_chZ, h->cur_s0);
        if (T.n_lin_bundles2 > 0) {
            const dim3 g2((unsigned)T.n_lin_bundles2, (unsigned)h->cur_S);
#define HPF_LB_B(NP_)                                                                                                                \
    hipLaunchKernelGGL((k_lin_bundle_back<NP_>), g2, dim3(256), 0, h->cur_stream, h->M, T.d_lb2rec, (const int2*)T.d_lb2x, T.d_lb2ptr, \
                       T.n_lin_heights, Bst, active, h->d_U, h->d_E, h->d_linA, h->d_w, h->d_x, h->cur_s0,                                 \
                       T.chains_bundled ? T.d_lb2cptr : (const int*)nullptr, T.d_lb2clist, T.d_crec, T.d_cnode, h->d_chZ)
            if (T.lin_np == 1) HPF_LB_B(1);
            else if (T.lin_np == 2) HPF_LB_B(2);
            else HPF_LB_B(4);
#undef HPF_LB_B
        } else if (T.n_lin_bundles > 0)
            hipLaunchKernelGGL(k_lin_tree_back, dim3((unsigned)T.n_lin_bundles, (unsigned)h->cur_S), dim3(256), 0, h->cur_stream, h->M,
                               td, T.d_lbrec, T.d_lbptr, T.n_lin_heights, h->N, h->Nc, Bst, active, h->d_U, h->d_E, h->d_linA, h->d_w,
                               h->d_x, h->cur_s0);
        for (int hh = T.n_lin_heights - 1; hh >= 0 && T.n_lin_bundles == 0 && T.n_lin_bundles2 == 0; --hh) {
            const int cnt = T.lh_ptr[hh + 1] - T.lh_ptr[hh];
            if (cnt == 0) continue;
            hipLaunchKernelGGL(k_lin_level_back, dim3((unsigned)((cnt * h->Hn + 127) / 128), (unsigned)h->cur_S), dim3(128), 0,
                               h->cur_stream, h->M, td, T.d_lrec + 8 * (size_t)T.lh_ptr[hh], cnt, h->N, h->Nc, Bst, active, h->d_U,
                               h->d_E, h->d_linA, h->d_w, h->d_x, (double*)nullptr, 0, h->cur_s0);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            h->last_detail = (int)e;
            return HPF_E_HIP;
        }
        return HPF_OK;
    }
    if (lin_threads > 0) {
        hipLaunchKernelGGL((k_lin_back<false>), dim3((unsigned)((lin_threads + 127) / 128), (unsigned)h->cur_S), dim3(128), 0, h->cur_stream,
                           h->M, td, T.n_lin_roots, h->N, h->Nc, Bst, active, h->d_U, h->d_E, h->d_linA, h->d_w, h->d_x,
                           h->d_f, h->cur_s0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            h->last_detail = (int)e;
            return HPF_E_HIP;
        }
    }
    return HPF_OK;
}

// BFS spanning tree of the admittance pattern from bus 0 (the SAME visiting order as tree_build_into) and the entries that are
// not tree edges: loop-closing lines.  Fills h->n_ties / n_tb / m_border and the device lists; radial networks: all zero.
int tree_find_ties(hpf_handle* h, const hpf_desc* d) {
    const int n = d->n;
    std::vector<int> parent(n, -2), order;
    order.reserve(n);
    order.push_back(0);
    parent[0] = -1;
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const int i = order[oi];
        for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e) {
            const int j = d->col[e];
            if (j != i && parent[j] == -2) {
                parent[j] = i;
                order.push_back(j);
            }
        }
    }
    if ((int)order.size() != n) return HPF_E_TOPOLOGY;               // not connected
    std::vector<std::vector<std::pair<int, int>>> adj(n);           // per bus: (other endpoint, CSR entry (i, j)) of its ties
    int n_entries = 0;
    for (int i = 0; i < n; ++i)
        for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e) {
            const int j = d->col[e];
            if (j == i || parent[i] == j || parent[j] == i) continue;
            adj[i].push_back({j, e});
            ++n_entries;
        }
    if (n_entries & 1) return HPF_E_TOPOLOGY;                       // pattern not symmetric
    for (int i = 0; i < n; ++i)
        for (auto& pr : adj[i]) {
            bool back = false;
            for (auto& q2 : adj[pr.first]) back = back || q2.first == i;
            if (!back) return HPF_E_TOPOLOGY;
        }
    h->n_ties = n_entries / 2;
    h->n_tb = 0;
    h->m_border = 0;
    if (h->n_ties == 0) return HPF_OK;
    std::vector<int> tb_bus, tb_ptr(1, 0), tb_adj;
    for (int i = 0; i < n; ++i) {
        if (adj[i].empty()) continue;
        tb_bus.push_back(i);
        for (auto& pr : adj[i]) {
            tb_adj.push_back(pr.first);
            tb_adj.push_back(pr.second);
            tb_adj.push_back(0);
        }
        tb_ptr.push_back((int)tb_adj.size() / 3);
    }
    h->n_tb = (int)tb_bus.size();
    h->m_border = h->n_tb * 2 * d->Hn;
    if (h->m_border > 16384 || wave_block_size(2 * d->Hn) == 0) return HPF_E_TOPOLOGY;  // stated bound of the bordered step (hpf.h): the
                                                                                        // m x m border system is dense (2 GiB at 16 384)
    // Factor-once form (default; HPF_MESH_SEL=0: the m virtual sweeps of rounds 2-4): the buses on the endpoints' root paths are marked for the
    // planner.  Coupled models only (an uncoupled model has no dense block on the 2x2 path), and while the blocks of J_t^-1 E_T on those
    // paths fit comfortably (|P| n_tb b^2 doubles).
    h->tb_bus_host = tb_bus;
    h->mesh_sel = false;
    h->sel_forced.clear();
    {
        const char* ms = h->sw("HPF_MESH_SEL");
        if (!(ms && atoi(ms) == 0) && d->coupled) {
            std::vector<char> fp(n, 0);
            size_t nP = 0;
            for (int i : tb_bus)
                for (int k = i; k >= 0 && !fp[k]; k = parent[k]) {
                    fp[k] = 1;
                    ++nP;
                }
            const double bytes = 8.0 * (double)nP * (double)h->n_tb * 4.0 * d->Hn * d->Hn;
            if (bytes <= 16.0 * 1073741824.0) {
                h->mesh_sel = true;
                h->sel_forced = std::move(fp);
            }
        }
    }
    if (h->plan_only) return HPF_OK;                     // (hpf_tree_plan: the host side alone -- ties, endpoint buses, the planner's mask)
    int r;
    if ((r = upload(h, &h->d_tb_bus, tb_bus))) return r;
    if ((r = upload(h, &h->d_tb_ptr, tb_ptr))) return r;
    if ((r = upload(h, &h->d_tb_adj, tb_adj))) return r;
    const size_t m = (size_t)h->m_border;
    if (hipMalloc((void**)&h->d_bM, sizeof(double) * m * m) != hipSuccess || hipMalloc((void**)&h->d_brhs, sizeof(double) * m) != hipSuccess ||
        hipMalloc((void**)&h->d_bipiv, sizeof(int) * m) != hipSuccess || hipMalloc((void**)&h->d_binfo, sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&h->d_bM0, sizeof(double) * m * m) != hipSuccess || hipMalloc((void**)&h->d_brhs0, sizeof(double) * (2 * m + 2)) != hipSuccess)
        return HPF_E_NOMEM;
    return HPF_OK;
}

// Newton step of a network with loop-closing lines (see the kernels above), per running real scenario r:
//   pass 1: the right-hand sides f and the m unit vectors of the border rows through tree_newton_step, border_slots(h) virtual slots at a
//           time (same voltages as r) -> the m x m border system, one chunk of columns per sweep; rocSOLVER LU -> g;
//   pass 2: x = J_t^-1 (f - E_T g), one more sweep in the first virtual slot.
// Nothing of Z = J_t^-1 E_T is kept, so m is bounded by the dense border system only.  Called with the launch context on h->stream
// over real slots.
int ensure_blas(hpf_handle* h) {
    if (h->blas) return HPF_OK;
    return rocblas_create_handle(&h->blas) == rocblas_status_success ? HPF_OK : HPF_E_ROCSOLVER;
}

// virtual scenario slots of the bordered step: all 1 + m right-hand sides at once up to 256, above that chunks of up to 1 024 (a chunk
// runs in the throughput regime of the tree kernels: 6.5 us per scenario-step at 256 live slots, 5.5 at 1 024; 72 MB of state per slot)
int border_slots(const hpf_handle* h) {
    if (h->mesh_sel) return 0;                           // factor-once form: the scenarios are swept in their own slots
    const int c = h->border_slot_cap < 16 ? 16 : h->border_slot_cap;   // (HPF_BORDER_SLOTS, read by hpf_create into the handle)
    return h->m_border + 1 < 256 ? h->m_border + 1 : (h->m_border + 1 < c ? ((h->m_border + 1 + 15) / 16) * 16 : c);
}

// max |v_i| of a vector (NaN-propagating) -> *out; one workgroup
__global__ __launch_bounds__(1024) void k_border_absmax(int m, const double* __restrict__ v, double* __restrict__ out) {
    __shared__ double red[1024];
    double a = 0.0;
    bool nan = false;
    for (int i = threadIdx.x; i < m; i += 1024) {
        const double x = fabs(v[i]);
        nan = nan || x != x;
        a = x > a ? x : a;
    }
    red[threadIdx.x] = nan ? NAN : a;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double p = red[threadIdx.x], q = red[threadIdx.x + off];
            red[threadIdx.x] = (p != p || q != q) ? NAN : (q > p ? q : p);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// ---- factor-once bordered step: selected inversion over the endpoints' root paths ----------------------------------------------------------
// (the algebra of hpf_csr_solve.hip's meshed branch, with the blocks read from the sweep of the scenarios' own slots: S_k^-1 from the inverse slot of
//  the plain Gauss-Jordan bus k (tile image -> dense, row-major), A(k, parent) / A(parent, k) harmonic-diagonal 2x2 blocks of the scenario's state)
// Every kernel below works on a BATCH of scenarios: grid dimension y (z) = position in the batch = index of the scenario's buffers; the scenario
// itself is the one in slot s0 + y of the (active) slot list, -1 = frozen / empty: nothing to do.
__device__ __forceinline__ int sel_scenario(const int* __restrict__ active, int s0, int y) { return active ? active[s0 + y] : s0 + y; }

// one workgroup per bus q of P and scenario:  S[q] = S_k^-1,  Z[q] = S_k^-1 A(k, p),  Up[q] = S_p^-1 A(p, k)
__global__ __launch_bounds__(256) void k_sel_prepare(Model M, int b, size_t CT, const int* __restrict__ active, int s0, int nP, const int* __restrict__ P,
                                                     const int* __restrict__ toff, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall,
                                                     const double* __restrict__ Zall, double* __restrict__ Sd, double* __restrict__ Zd,
                                                     double* __restrict__ Up) {
    __shared__ double hg[2][64 * 4];                     // [0]: A(k, p), [1]: A(p, k), per harmonic 2x2 row-major (Hn <= 64)
    const int v = sel_scenario(active, s0, blockIdx.y);
    if (v < 0) return;
    const int q = blockIdx.x, k = P[4 * q], par = P[4 * q + 1];
    const int Hn = M.Hn, n = M.n;
    const size_t so = (size_t)v * n * Hn, bb = (size_t)b * b, yo = (size_t)blockIdx.y * nP * bb;
    const int pbus = par >= 0 ? P[4 * par] : -1;
    if ((int)threadIdx.x < 2 * Hn && pbus >= 0) {
        const int w = threadIdx.x / Hn, hq = threadIdx.x - w * Hn;
        double g4[4];
        if (w == 0)
            coupling_block(M, Uall + so, Eall + so, hq, k, pbus, P[4 * q + 2], g4);
        else
            coupling_block(M, Uall + so, Eall + so, hq, pbus, k, P[4 * q + 3], g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) hg[w][hq * 4 + e] = g4[e];
    }
    __syncthreads();
    const double* Zk = Zall + ((size_t)v * n + k) * CT;
    const double* Zp = pbus >= 0 ? Zall + ((size_t)v * n + pbus) * CT : nullptr;
    for (int idx = threadIdx.x; idx < b * b; idx += 256) {
        const int i = idx / b, cc = idx - i * b, hq = cc >> 1, t2 = cc & 1;
        Sd[yo + q * bb + idx] = Zk[toff[idx]];
        double z = 0.0, u = 0.0;
        if (pbus >= 0) {
            const int o0 = toff[i * b + 2 * hq], o1 = toff[i * b + 2 * hq + 1];
            z = fma(Zk[o1], hg[0][hq * 4 + 2 + t2], Zk[o0] * hg[0][hq * 4 + t2]);
            u = fma(Zp[o1], hg[1][hq * 4 + 2 + t2], Zp[o0] * hg[1][hq * 4 + t2]);
        }
        Zd[yo + q * bb + idx] = z;
        Up[yo + q * bb + idx] = u;
    }
}

// coupling blocks of the ties at the scenario's state: tie[(e * Hn + q) * 4 + .] = A(i, j) at harmonic position q, e the directed tie (tb_adj order)
__global__ void k_sel_ties(Model M, const int* __restrict__ active, int s0, int n_dir, const int* __restrict__ tb_bus, const int* __restrict__ tb_ptr,
                           const int* __restrict__ tb_adj, int n_tb, const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, double* __restrict__ tie) {
    const int v = sel_scenario(active, s0, blockIdx.y);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (v < 0 || t >= n_dir * M.Hn) return;
    const int e = t / M.Hn, q = t - e * M.Hn;
    int a = 0;
    while (a + 1 < n_tb && tb_ptr[a + 1] <= e) ++a;
    const size_t so = (size_t)v * M.n * M.Hn;
    double g4[4];
    coupling_block(M, Uall + so, Eall + so, q, tb_bus[a], tb_adj[3 * e], tb_adj[3 * e + 1], g4);
#pragma unroll
    for (int u = 0; u < 4; ++u) tie[((size_t)blockIdx.y * n_dir * M.Hn + t) * 4 + u] = g4[u];
}

// right-hand side Q^T y of the border system from the scenario's first sweep (x = y in the bus-image layout, stride Bst)
__global__ void k_border_rhs_sel(int b, int Hn, int m, int n, int Bst, int n_dir, const int* __restrict__ active, int s0, const int* __restrict__ tb_ptr,
                                 const int* __restrict__ tb_adj, const double* __restrict__ tie, const double* __restrict__ xall, double* __restrict__ brhs) {
    const int v = sel_scenario(active, s0, blockIdx.y);
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (v < 0 || row >= m) return;
    const int a = row / b, l = row - a * b, q = l >> 1, t = l & 1;
    const double* x = xall + (size_t)v * n * Bst;
    double acc = 0.0;
    for (int e = tb_ptr[a]; e < tb_ptr[a + 1]; ++e) {
        const double* g4 = tie + (((size_t)blockIdx.y * n_dir + e) * Hn + q) * 4;
        const int j = tb_adj[3 * e];
        acc = fma(g4[t * 2], x[(size_t)j * Bst + 2 * q], acc);
        acc = fma(g4[t * 2 + 1], x[(size_t)j * Bst + 2 * q + 1], acc);
    }
    brhs[(size_t)blockIdx.y * m + row] = acc;
}

// border matrix I + Q^T Z (column-major m x m) from the blocks X[j, t] of the selected inversion: thread = (row (a, l), column (t, cc)), z = batch
__global__ __launch_bounds__(256) void k_border_build_sel(int b, int Hn, int m, int n_tb, int nP, int n_dir, const int* __restrict__ active, int s0,
                                                          const int* __restrict__ tb_ptr, const int* __restrict__ tb_adj, const int* __restrict__ pidx,
                                                          const double* __restrict__ tie, const double* __restrict__ X, double* __restrict__ bM,
                                                          double* __restrict__ bB) {
    const int y = blockIdx.z;
    if (sel_scenario(active, s0, y) < 0) return;
    const int row = blockIdx.x * 256 + threadIdx.x, col = blockIdx.y;
    if (row >= m) return;
    const int a = row / b, l = row - a * b, q = l >> 1, t = l & 1;
    const int tcol = col / b, cc = col - tcol * b;
    const size_t bb = (size_t)b * b;
    const double* Xy = X + (size_t)y * nP * n_tb * bb;
    double acc = row == col ? 1.0 : 0.0;
    for (int e = tb_ptr[a]; e < tb_ptr[a + 1]; ++e) {
        const double* g4 = tie + (((size_t)y * n_dir + e) * Hn + q) * 4;
        const double* Xj = Xy + ((size_t)pidx[tb_adj[3 * e]] * n_tb + tcol) * bb;
        acc = fma(g4[t * 2], Xj[(size_t)(2 * q) * b + cc], acc);
        acc = fma(g4[t * 2 + 1], Xj[(size_t)(2 * q + 1) * b + cc], acc);
    }
    bM[(size_t)y * m * m + (size_t)col * m + row] = acc;
    if (bB) bB[(size_t)y * n_tb * (n_tb + 1) * bb + ((size_t)a * (n_tb + 1) + tcol) * bb + (size_t)l * b + cc] = acc;      // block layout (border_block_gj)
}

// right-hand side -> block column n_tb of the block layout (first column of each block, the rest zero) | solution back into a vector
__global__ void k_border_rhs_blocks(int b, int n_tb, const double* __restrict__ rhs, double* __restrict__ bB) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tb * b * b) return;
    const int s = t / (b * b), e = t - s * b * b, l = e / b, cc = e - l * b;
    bB[(size_t)blockIdx.y * n_tb * (n_tb + 1) * b * b + ((size_t)s * (n_tb + 1) + n_tb) * b * b + e] =
        cc == 0 ? rhs[(size_t)blockIdx.y * n_tb * b + (size_t)s * b + l] : 0.0;
}
__global__ void k_border_g_blocks(int b, int n_tb, const double* __restrict__ bB, double* __restrict__ g) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tb * b) return;
    const int s = t / b, l = t - s * b;
    g[(size_t)blockIdx.y * n_tb * b + t] = bB[(size_t)blockIdx.y * n_tb * (n_tb + 1) * b * b + ((size_t)s * (n_tb + 1) + n_tb) * b * b + (size_t)l * b];
}

// residual check of a border solve against the untouched column-major copy: res[2 y] = max |rhs - M g|, res[2 y + 1] = max |rhs| (as bit patterns of
// non-negative doubles: atomicMax on 64-bit integers orders them; a NaN anywhere raises the first word to +inf)
__global__ __launch_bounds__(256) void k_border_check(int m, const double* __restrict__ bM, const double* __restrict__ rhs, const double* __restrict__ g,
                                                      unsigned long long* __restrict__ res) {
    // 16 rows x 16 column parts per workgroup: thread (r, part) sums every 16th column of its row
    __shared__ double part_s[16][17];
    const int y = blockIdx.y, r = threadIdx.x & 15, part = threadIdx.x >> 4, row = blockIdx.x * 16 + r;
    double acc = 0.0;
    if (row < m) {
        const double* Mr = bM + (size_t)y * m * m + row;
        const double* gy = g + (size_t)y * m;
        for (int c2 = part; c2 < m; c2 += 16) acc = fma(Mr[(size_t)c2 * m], gy[c2], acc);
    }
    part_s[part][r] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        double rr = 0.0, a = 0.0;
        if (row < m) {
            double sum = 0.0;
#pragma unroll
            for (int p2 = 0; p2 < 16; ++p2) sum += part_s[p2][r];
            const double f0 = rhs[(size_t)y * m + row];
            rr = fabs(f0 - sum);
            a = fabs(f0);
            if (!(rr == rr) || !(a == a)) rr = INFINITY;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            rr = fmax(rr, __shfl_xor(rr, off, 64));
            a = fmax(a, __shfl_xor(a, off, 64));
        }
        if (threadIdx.x == 0) {
            atomicMax(res + 2 * y, (unsigned long long)__double_as_longlong(rr));
            atomicMax(res + 2 * y + 1, (unsigned long long)__double_as_longlong(a));
        }
    }
}

// Second pass of the factor-once form.  x = J_t^-1 (f - E_T g) differs from y = J_t^-1 f only through the forward vectors of the buses of P (the
// subtrees off P hold no endpoint bus: their forward vectors stay): dw_k = S_k^-1 dy_k - sum over the children c of k in P of Up_c dw_c,
// dy = -g at the endpoint buses.  One workgroup per bus of P, height level and scenario; the forward vector of the bus in the scenario's slot
// (stride Bst) is corrected in place -- the back sweep alone then gives x.
__global__ __launch_bounds__(256) void k_sel_dw(int b, int Bst, int n, int nP, int m, const int* __restrict__ active, int s0, const int* __restrict__ nodes,
                                                const int* __restrict__ P, const int* __restrict__ slot, const int* __restrict__ cptr,
                                                const int* __restrict__ clist, const double* __restrict__ Sd, const double* __restrict__ Up,
                                                const double* __restrict__ g, double* __restrict__ dw, double* __restrict__ wall) {
    const int v = sel_scenario(active, s0, blockIdx.y);
    if (v < 0) return;
    const int q = nodes[blockIdx.x], k = P[4 * q];
    const int part = threadIdx.x & 3;                    // four threads per row, every fourth column each
    const size_t bb = (size_t)b * b, yo = (size_t)blockIdx.y * nP * bb;
    const double* gy = g + (size_t)blockIdx.y * m;
    double* dwy = dw + (size_t)blockIdx.y * nP * b;
    const int sl = slot[q];
    for (int i = threadIdx.x >> 2; i < ((b + 63) & ~63); i += 64) {
        double acc = 0.0;
        if (i < b) {
            if (sl >= 0) {
                const double* Sr = Sd + yo + (size_t)q * bb + (size_t)i * b;
                for (int j = part; j < b; j += 4) acc = fma(-Sr[j], gy[(size_t)sl * b + j], acc);
            }
            for (int cp = cptr[q]; cp < cptr[q + 1]; ++cp) {
                const int c = clist[cp];
                const double* Ur = Up + yo + (size_t)c * bb + (size_t)i * b;
                const double* dc = dwy + (size_t)c * b;
                for (int j = part; j < b; j += 4) acc = fma(-Ur[j], dc[j], acc);
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (part == 0 && i < b) {
            dwy[(size_t)q * b + i] = acc;
            wall[((size_t)v * n + k) * Bst + i] += acc;
        }
    }
}

// Host side, once per handle (after tree_build): P ordered by depth, the forward pairs ordered by height, the block-product jobs with their
// final device addresses, the buffers.
int tree_sel_build(hpf_handle* h, const hpf_desc* d) {
    if (!h->mesh_sel) return HPF_OK;
    const Tree& T = active_tree(h);
    const int n = h->n, b = 2 * h->Hn, mT = h->n_tb;
    const size_t bb = (size_t)b * b;
    if ((int)T.plain_gj.size() != n || (int)T.toff_tab.size() != b * b || h->Hn > 64) return HPF_E_STATE;
    std::vector<int> order, depth(n, 0), height(n, 0);
    {
        std::vector<std::vector<int>> ch(n);
        for (int i = 1; i < n; ++i) ch[T.parent[i]].push_back(i);
        order.push_back(0);
        for (size_t oi = 0; oi < order.size(); ++oi)
            for (int j : ch[order[oi]]) {
                depth[j] = depth[order[oi]] + 1;
                order.push_back(j);
            }
        for (int oi = n - 1; oi > 0; --oi) {
            const int i = order[oi];
            if (height[i] + 1 > height[T.parent[i]]) height[T.parent[i]] = height[i] + 1;
        }
    }
    std::vector<int> e_up(n, -1), e_dn(n, -1);
    for (int i = 1; i < n; ++i) {
        const int p = T.parent[i];
        for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e)
            if (d->col[e] == p) e_up[i] = e;
        for (int e = d->rowptr[p]; e < d->rowptr[p + 1]; ++e)
            if (d->col[e] == i) e_dn[i] = e;
    }
    std::vector<int> pidx(n, -1), Pbus, Prec;
    for (int oi = 0; oi < n; ++oi) {
        const int k = order[oi];
        if (!h->sel_forced[k]) continue;
        if (!T.plain_gj[k]) return HPF_E_STATE;              // (the planner keeps marked buses on the plain Gauss-Jordan path)
        pidx[k] = (int)Pbus.size();
        Pbus.push_back(k);
    }
    const size_t nP = Pbus.size();
    int max_depth = 0, max_height = 0;
    for (int k : Pbus) {
        Prec.push_back(k);
        Prec.push_back(k > 0 ? pidx[T.parent[k]] : -1);
        Prec.push_back(k > 0 ? e_up[k] : 0);
        Prec.push_back(k > 0 ? e_dn[k] : 0);
        max_depth = std::max(max_depth, depth[k]);
        max_height = std::max(max_height, height[k]);
    }
    struct Pair {
        int bus, t, pred, child;
    };
    std::vector<Pair> raw;
    for (int t = 0; t < mT; ++t) {
        int pred = -1, chb = -1;
        for (int k = h->tb_bus_host[t]; k >= 0; k = T.parent[k]) {
            raw.push_back({k, t, pred, chb});
            pred = (int)raw.size() - 1;
            chb = k;
        }
    }
    const size_t npairs = raw.size();
    std::vector<int> perm(npairs), newidx(npairs), pair_of(nP * (size_t)mT, -1);
    for (size_t q = 0; q < npairs; ++q) perm[q] = (int)q;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int c2) { return height[raw[a].bus] < height[raw[c2].bus]; });
    for (size_t q = 0; q < npairs; ++q) newidx[perm[q]] = (int)q;
    std::vector<Pair> pairs(npairs);
    std::vector<size_t> lvl_cnt(max_height + 2, 0);
    for (size_t q = 0; q < npairs; ++q) {
        pairs[q] = raw[perm[q]];
        if (pairs[q].pred >= 0) pairs[q].pred = newidx[pairs[q].pred];
        lvl_cnt[height[pairs[q].bus] + 1]++;
        pair_of[(size_t)pidx[pairs[q].bus] * mT + pairs[q].t] = (int)q;
    }
    {
        // second pass: buses of P by height, their children in P, their endpoint number
        std::vector<int> byh(nP), slot(nP, -1), cptr(nP + 1, 0), clist;
        for (size_t q = 0; q < nP; ++q) byh[q] = (int)q;
        std::stable_sort(byh.begin(), byh.end(), [&](int a, int c2) { return height[Pbus[a]] < height[Pbus[c2]]; });
        h->sel_hl_ptr.assign(1, 0);
        for (size_t q = 0; q < nP; ++q)
            if (q + 1 == nP || height[Pbus[byh[q + 1]]] != height[Pbus[byh[q]]]) h->sel_hl_ptr.push_back(q + 1);
        for (int t = 0; t < mT; ++t) slot[pidx[h->tb_bus_host[t]]] = t;
        for (size_t q = 1; q < nP; ++q) cptr[pidx[T.parent[Pbus[q]]] + 1]++;
        for (size_t q = 0; q < nP; ++q) cptr[q + 1] += cptr[q];
        clist.assign(nP > 1 ? nP - 1 : 1, 0);
        std::vector<int> pos(cptr.begin(), cptr.end() - 1);
        for (size_t q = 1; q < nP; ++q) clist[pos[pidx[T.parent[Pbus[q]]]]++] = (int)q;
        int r2;
        if ((r2 = upload(h, &h->d_sel_hl, byh))) return r2;
        if ((r2 = upload(h, &h->d_sel_slot, slot))) return r2;
        if ((r2 = upload(h, &h->d_sel_cptr, cptr))) return r2;
        if ((r2 = upload(h, &h->d_sel_clist, clist))) return r2;
    }
    h->sel_nP = (int)nP;
    h->sel_npairs = (int)npairs;
    h->sel_R = (b + 15) / 16;
    // Buffers for `sel_cap` scenarios at a time (a batch of the group's running scenarios goes through the selected inversion and the border solve
    // together): as many as fit HPF_MESH_BATCH_GB (default 48) of the handle's capacity, at least one.
    const size_t mb = (size_t)h->m_border;
    {
        const char* bg = h->sw("HPF_BORDER_GJ");
        const int lim = bg ? atoi(bg) : 96;
        h->border_gj = mT <= lim && mT >= 1;
        const double per = 8.0 * ((double)bb * (3.0 * nP + npairs + (double)nP * mT + (h->border_gj ? (double)mT * (mT + 1) : 0.0)) + (double)mb * mb + (double)nP * b +
                                  3.0 * mb + 8.0 * h->n_ties * h->Hn);
        const char* gb = h->sw("HPF_MESH_BATCH_GB");
        double budget = (gb ? atof(gb) : 48.0) * 1073741824.0;
        size_t free_b = 0, total_b = 0;                  // (never more than half of what the device has free at this point: the scenario state follows)
        if (!gb && hipMemGetInfo(&free_b, &total_b) == hipSuccess && budget > 0.5 * (double)free_b) budget = 0.5 * (double)free_b;
        long long cap = (long long)(budget / per);
        cap = cap < 1 ? 1 : (cap > h->S_max ? h->S_max : cap);
        h->sel_cap = (int)cap;
    }
    const size_t cap = (size_t)h->sel_cap;
    h->sel_res_host.assign(2 * cap, 0ull);
    h->sel_info_host.assign(cap, 0);
    if (hipMalloc((void**)&h->d_sel_dw, sizeof(double) * cap * nP * b) != hipSuccess) return HPF_E_NOMEM;
    auto dalloc = [&](double** p2, size_t cnt) { return hipMalloc((void**)p2, sizeof(double) * (cnt ? cnt : 1)) == hipSuccess; };
    if (!dalloc(&h->d_sel_S, cap * nP * bb) || !dalloc(&h->d_sel_Z, cap * nP * bb) || !dalloc(&h->d_sel_Up, cap * nP * bb) ||
        !dalloc(&h->d_sel_W, cap * npairs * bb) || !dalloc(&h->d_sel_X, cap * nP * (size_t)mT * bb) ||
        !dalloc(&h->d_sel_tie, cap * 2 * h->n_ties * h->Hn * 4) || !dalloc(&h->d_sel_bM, cap * mb * mb) || !dalloc(&h->d_sel_rhs, cap * mb) ||
        !dalloc(&h->d_sel_g, cap * mb) || hipMalloc((void**)&h->d_sel_res, sizeof(unsigned long long) * 2 * cap) != hipSuccess ||
        hipMalloc((void**)&h->d_sel_info, sizeof(int) * cap) != hipSuccess)
        return HPF_E_NOMEM;
    const long long sS = (long long)(nP * bb), sW = (long long)(npairs * bb), sX = (long long)(nP * (size_t)mT * bb);
    int r;
    if ((r = upload(h, &h->d_sel_P, Prec))) return r;
    if ((r = upload(h, &h->d_sel_pidx, pidx))) return r;
    if ((r = upload(h, &h->d_sel_toff, T.toff_tab))) return r;
    std::vector<BlkJob> jobs;
    jobs.reserve(npairs + nP * (size_t)mT);
    h->sel_fwd_beg.assign(1, 0);
    {
        size_t q = 0;
        for (int l = 0; l <= max_height; ++l) {
            for (size_t e = 0; e < lvl_cnt[l + 1]; ++e, ++q) {
                const Pair& pr = pairs[q];
                if (pr.pred < 0)
                    jobs.push_back({nullptr, h->d_sel_S + (size_t)pidx[pr.bus] * bb, nullptr, h->d_sel_W + q * bb, 1.0, 0, sS, 0, sW});
                else
                    jobs.push_back({h->d_sel_Up + (size_t)pidx[pr.child] * bb, h->d_sel_W + (size_t)pr.pred * bb, nullptr, h->d_sel_W + q * bb, -1.0, sS, sW, 0, sW});
            }
            h->sel_fwd_beg.push_back(jobs.size());
        }
    }
    h->sel_back_beg.assign(1, jobs.size());
    {
        size_t q = 0;                                        // (Pbus is ordered by depth)
        for (int dl = 0; dl <= max_depth; ++dl) {
            for (; q < nP && depth[Pbus[q]] == dl; ++q) {
                const int k = Pbus[q];
                for (int t = 0; t < mT; ++t) {
                    const int pq = pair_of[q * mT + t];
                    const double* wq = pq >= 0 ? h->d_sel_W + (size_t)pq * bb : nullptr;
                    double* xo = h->d_sel_X + (q * mT + t) * bb;
                    if (k == 0)
                        jobs.push_back({nullptr, wq, nullptr, xo, 1.0, 0, sW, 0, sX});
                    else
                        jobs.push_back({h->d_sel_Z + q * bb, h->d_sel_X + ((size_t)pidx[T.parent[k]] * mT + t) * bb, wq, xo, -1.0, sS, sX, sW, sX});
                }
            }
            h->sel_back_beg.push_back(jobs.size());
        }
    }
    if (hipMalloc(&h->d_sel_jobs, sizeof(BlkJob) * (jobs.size() ? jobs.size() : 1)) != hipSuccess) return HPF_E_NOMEM;
    if (hipMemcpy(h->d_sel_jobs, jobs.data(), sizeof(BlkJob) * jobs.size(), hipMemcpyHostToDevice) != hipSuccess) return HPF_E_HIP;
    // Border systems of up to HPF_BORDER_GJ blocks (default 96) are solved by a block Gauss-Jordan elimination on the b x b grid with the block-product
    // kernel instead of rocSOLVER's unpivoted LU (hundreds of small launches at these sizes): step k inverts block (k, k) in place on the matrix
    // cores (k_blk_invert_mfma: static 4 x 4 pivot blocks under the same watch as the tree's; the residual check and the pivoted rocSOLVER
    // fallback of the border solve stay), scales block row k, and eliminates block column k from every other row -- three launches per step;
    // the right-hand side rides as block column n_tb.  Measured on syn1000 + 5 / 20 / 40 / 80 ties (n_tb = 10 / 39 / 78 / 150): 1.40 / 3.2 / 9.0 / 47 ms
    // per Newton step against 2.2 / 5.7 / 11.7 / 31 ms with rocSOLVER (1.5 x the flops of an LU, block products at ~2 TFLOP/s).
    {
        const char* bl = h->sw("HPF_BORDER_PIVLIM");
        if (bl && atof(bl) > 0.0) h->border_piv_limit = atof(bl);
        const char* bm = h->sw("HPF_BORDER_GJ_MFMA");
        h->border_gj_mfma = !(bm && atoi(bm) == 0);     // 0: the diagonal blocks through the VALU Gauss-Jordan (gj_dense_invert_npvt; A/B)
        if (h->border_gj) {
            const size_t W = (size_t)mT + 1;
            if (!dalloc(&h->d_bB, cap * (size_t)mT * W * bb)) return HPF_E_NOMEM;
            const long long sB = (long long)((size_t)mT * W * bb);
            std::vector<BlkJob> gj;
            h->bgj_beg.assign(1, 0);
            auto blk = [&](size_t s2, size_t t2) { return h->d_bB + (s2 * W + t2) * bb; };
            for (size_t k = 0; k < (size_t)mT; ++k) {
                for (size_t t2 = k + 1; t2 < W; ++t2) gj.push_back({blk(k, k), blk(k, t2), nullptr, blk(k, t2), 1.0, sB, sB, 0, sB});
                h->bgj_beg.push_back(gj.size());
                for (size_t i = 0; i < (size_t)mT; ++i)
                    if (i != k)
                        for (size_t t2 = k + 1; t2 < W; ++t2) gj.push_back({blk(i, k), blk(k, t2), blk(i, t2), blk(i, t2), -1.0, sB, sB, sB, sB});
                h->bgj_beg.push_back(gj.size());
            }
            if (hipMalloc(&h->d_bgj_jobs, sizeof(BlkJob) * (gj.size() ? gj.size() : 1)) != hipSuccess) return HPF_E_NOMEM;
            if (hipMemcpy(h->d_bgj_jobs, gj.data(), sizeof(BlkJob) * gj.size(), hipMemcpyHostToDevice) != hipSuccess) return HPF_E_HIP;
        }
    }
    if (h->sw("HPF_TREE_INFO"))
        fprintf(stderr, "hpf tree: factor-once bordered step: %d ties, %d endpoint buses (border %d), %zu buses on their root paths, %zu forward pairs, "
                        "%zu block products per Newton step and scenario\n", h->n_ties, mT, h->m_border, nP, npairs, jobs.size());
    return HPF_OK;
}

// the selected inversion of the scenarios in slots s0 .. s0 + cn - 1 (their factors are in their slots) -> border matrices, right-hand sides
static void tree_sel_run(hpf_handle* h, const int* active, int s0, int cn, hipStream_t st) {
    const int b = 2 * h->Hn, BW = wave_block_size(b), NT = (BW + 16) / 16, m = h->m_border, n_dir = 2 * h->n_ties;
    const size_t CT = (size_t)NT * NT * 256;
    const BlkJob* jobs = static_cast<const BlkJob*>(h->d_sel_jobs);
    hipLaunchKernelGGL(k_sel_ties, dim3((unsigned)((n_dir * h->Hn + 255) / 256), (unsigned)cn), dim3(256), 0, st, h->M, active, s0, n_dir, h->d_tb_bus, h->d_tb_ptr,
                       h->d_tb_adj, h->n_tb, h->d_U, h->d_E, h->d_sel_tie);
    hipLaunchKernelGGL(k_border_rhs_sel, dim3((unsigned)((m + 255) / 256), (unsigned)cn), dim3(256), 0, st, b, h->Hn, m, h->n, BW, n_dir, active, s0, h->d_tb_ptr,
                       h->d_tb_adj, (const double*)h->d_sel_tie, (const double*)h->d_x, h->d_sel_rhs);
    hipLaunchKernelGGL(k_sel_prepare, dim3((unsigned)h->sel_nP, (unsigned)cn), dim3(256), 0, st, h->M, b, CT, active, s0, h->sel_nP, h->d_sel_P, h->d_sel_toff,
                       h->d_U, h->d_E, h->d_Z, h->d_sel_S, h->d_sel_Z, h->d_sel_Up);
    for (size_t l = 0; l + 1 < h->sel_fwd_beg.size(); ++l)
        launch_jobs(h->sel_R, b, (int)(h->sel_fwd_beg[l + 1] - h->sel_fwd_beg[l]), jobs + h->sel_fwd_beg[l], st, cn);
    for (size_t l = 0; l + 1 < h->sel_back_beg.size(); ++l)
        launch_jobs(h->sel_R, b, (int)(h->sel_back_beg[l + 1] - h->sel_back_beg[l]), jobs + h->sel_back_beg[l], st, cn);
    hipLaunchKernelGGL(k_border_build_sel, dim3((unsigned)((m + 255) / 256), (unsigned)m, (unsigned)cn), dim3(256), 0, st, b, h->Hn, m, h->n_tb, h->sel_nP, n_dir,
                       active, s0, h->d_tb_ptr, h->d_tb_adj, h->d_sel_pidx, (const double*)h->d_sel_tie, (const double*)h->d_sel_X, h->d_sel_bM,
                       h->border_gj ? h->d_bB : (double*)nullptr);
}

// the block Gauss-Jordan solve of the cn border systems in h->d_bB (block layout, right-hand sides h->d_sel_rhs) -> h->d_sel_g, weak / zero pivots -> h->d_sel_info
static void border_block_gj(hpf_handle* h, int cn, hipStream_t st) {
    const int b = 2 * h->Hn, mT = h->n_tb, R = h->sel_R;
    const size_t bb = (size_t)b * b, W = (size_t)mT + 1;
    const long long sB = (long long)((size_t)mT * W * bb);
    const BlkJob* jobs = static_cast<const BlkJob*>(h->d_bgj_jobs);
    hipLaunchKernelGGL(k_border_rhs_blocks, dim3((unsigned)((mT * bb + 255) / 256), (unsigned)cn), dim3(256), 0, st, b, mT, (const double*)h->d_sel_rhs, h->d_bB);
    for (int k = 0; k < mT; ++k) {
        double* dk = h->d_bB + ((size_t)k * W + k) * bb;
        if (!h->border_gj_mfma || !launch_invert_mfma(wave_block_size(b), b, dk, sB, cn, h->border_piv_limit, h->d_sel_info, st)) launch_invert(R, b, dk, sB, cn, h->d_sel_info, st);
        launch_jobs(R, b, (int)(h->bgj_beg[2 * k + 1] - h->bgj_beg[2 * k]), jobs + h->bgj_beg[2 * k], st, cn);
        launch_jobs(R, b, (int)(h->bgj_beg[2 * k + 2] - h->bgj_beg[2 * k + 1]), jobs + h->bgj_beg[2 * k + 1], st, cn);
    }
    hipLaunchKernelGGL(k_border_g_blocks, dim3((unsigned)((mT * b + 255) / 256), (unsigned)cn), dim3(256), 0, st, b, mT, (const double*)h->d_bB, h->d_sel_g);
}

// Newton step of a meshed handle in the factor-once form, for the running scenarios of slots h->cur_s0 .. + h->cur_S (see tree_newton_step_bordered):
//   1. ONE sweep of the tree for all of them (their own right-hand sides, their own slots: y = J_t^-1 f and the factors);
//   2. in batches of sel_cap scenarios: selected inversion -> border matrices, border solve (block Gauss-Jordan, or rocSOLVER's unpivoted LU one
//      system after the other), residual check of every solution against the untouched column-major copy -- ONE host synchronisation per batch --,
//      pivoted rocSOLVER LU for the systems that fail it, correction of the forward vectors on P;
//   3. the back sweep alone for all of them.
static int tree_newton_step_sel(hpf_handle* h, bool only_active) {
    const int m = h->m_border, b = 2 * h->Hn, BW = wave_block_size(b), n = h->n;
    const int s0 = h->cur_s0, cnt = h->cur_S;
    hipStream_t st = h->cur_stream;
    const int* active = only_active ? h->d_active : nullptr;
    int rc = tree_newton_step(h, only_active);
    if (rc) return rc;
    for (int c0 = 0; c0 < cnt; c0 += h->sel_cap) {
        const int cn = cnt - c0 < h->sel_cap ? cnt - c0 : h->sel_cap;
        const int sb = s0 + c0;
        tree_sel_run(h, active, sb, cn, st);
        hipMemsetAsync(h->d_sel_info, 0, sizeof(int) * cn, st);
        hipMemsetAsync(h->d_sel_res, 0, sizeof(unsigned long long) * 2 * cn, st);
        if (h->border_gj) {
            border_block_gj(h, cn, st);
        } else {
            if (ensure_blas(h) || rocblas_set_stream(h->blas, st) != rocblas_status_success) return HPF_E_ROCSOLVER;
            for (int y = 0; y < cn; ++y) {                     // (large borders: one system after the other through the handle's scratch matrix)
                hipMemcpyAsync(h->d_bM, h->d_sel_bM + (size_t)y * m * m, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToDevice, st);
                hipMemcpyAsync(h->d_sel_g + (size_t)y * m, h->d_sel_rhs + (size_t)y * m, sizeof(double) * m, hipMemcpyDeviceToDevice, st);
                if (rocsolver_dgetrf_npvt(h->blas, m, m, h->d_bM, m, h->d_sel_info + y) != rocblas_status_success ||
                    rocblas_dtrsv(h->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_unit, m, h->d_bM, m, h->d_sel_g + (size_t)y * m, 1) != rocblas_status_success ||
                    rocblas_dtrsv(h->blas, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, m, h->d_bM, m, h->d_sel_g + (size_t)y * m, 1) != rocblas_status_success)
                    return HPF_E_ROCSOLVER;
            }
        }
        hipLaunchKernelGGL(k_border_check, dim3((unsigned)((m + 15) / 16), (unsigned)cn), dim3(256), 0, st, m, (const double*)h->d_sel_bM, (const double*)h->d_sel_rhs,
                           (const double*)h->d_sel_g, h->d_sel_res);
        if (hipMemcpyAsync(h->sel_res_host.data(), h->d_sel_res, sizeof(unsigned long long) * 2 * cn, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(h->sel_info_host.data(), h->d_sel_info, sizeof(int) * cn, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            h->last_detail = (int)hipGetLastError();
            return HPF_E_HIP;
        }
        const bool info_on = h->sw("HPF_BORDER_INFO") != nullptr;
        for (int y = 0; y < cn; ++y) {
            const int sc = only_active ? ((size_t)(sb + y) < h->host_act.size() ? h->host_act[sb + y] : -1) : sb + y;
            if (sc < 0) continue;
            double r0, r1;
            memcpy(&r0, &h->sel_res_host[2 * y], sizeof(double));
            memcpy(&r1, &h->sel_res_host[2 * y + 1], sizeof(double));
            const int info = h->sel_info_host[y];
            if (info_on)
                fprintf(stderr, "hpf border system (m = %d, scenario %d): |rhs - M g| / |rhs| = %.2e, info %d\n", m, sc, r1 > 0.0 ? r0 / r1 : 0.0, info);
            if (!(h->border_pivoting || info != 0 || !(r0 <= 1e-10 * r1) || !(r1 < INFINITY))) continue;
            // this system again, with partial pivoting, from the untouched copy
            ++h->border_repivots;
            if (ensure_blas(h) || rocblas_set_stream(h->blas, st) != rocblas_status_success) return HPF_E_ROCSOLVER;
            int pinfo = 0;
            hipMemcpyAsync(h->d_bM, h->d_sel_bM + (size_t)y * m * m, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(h->d_sel_g + (size_t)y * m, h->d_sel_rhs + (size_t)y * m, sizeof(double) * m, hipMemcpyDeviceToDevice, st);
            if (rocsolver_dgetrf(h->blas, m, m, h->d_bM, m, h->d_bipiv, h->d_binfo) != rocblas_status_success ||
                rocsolver_dgetrs(h->blas, rocblas_operation_none, m, 1, h->d_bM, m, h->d_bipiv, h->d_sel_g + (size_t)y * m, m) != rocblas_status_success)
                return HPF_E_ROCSOLVER;
            if (hipMemcpyAsync(&pinfo, h->d_binfo, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                h->last_detail = (int)hipGetLastError();
                return HPF_E_HIP;
            }
            if (pinfo != 0) {                             // exactly singular border system
                h->last_detail = sc;
                return HPF_E_SINGULAR;
            }
        }
        for (size_t l = 0; l + 1 < h->sel_hl_ptr.size(); ++l)
            hipLaunchKernelGGL(k_sel_dw, dim3((unsigned)(h->sel_hl_ptr[l + 1] - h->sel_hl_ptr[l]), (unsigned)cn), dim3(256), 0, st, b, BW, n, h->sel_nP, m, active, sb,
                               h->d_sel_hl + h->sel_hl_ptr[l], h->d_sel_P, h->d_sel_slot, h->d_sel_cptr, h->d_sel_clist, (const double*)h->d_sel_S,
                               (const double*)h->d_sel_Up, (const double*)h->d_sel_g, h->d_sel_dw, h->d_w);
    }
    h->tree_back_only = true;
    rc = tree_newton_step(h, only_active);
    h->tree_back_only = false;
    if (rc) return rc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

int tree_newton_step_bordered(hpf_handle* h, bool only_active) {
    if (h->mesh_sel) return tree_newton_step_sel(h, only_active);
    const int m = h->m_border, VC = border_slots(h), v0 = h->S_max;
    const int b = 2 * h->Hn, BW = wave_block_size(b), n = h->n;
    const int s0 = h->cur_s0, cnt = h->cur_S;
    hipStream_t st = h->stream;
    std::vector<int> todo;
    for (int slot = s0; slot < s0 + cnt; ++slot) {
        const int sc = only_active ? ((size_t)slot < h->host_act.size() ? h->host_act[slot] : -1) : slot;
        if (sc >= 0) todo.push_back(sc);
    }
    const int save_groups = h->n_groups;
    const int cmax = n * BW;
    // the virtual scenarios of one chunk: slots [v0, v0 + V), scenario groups on their own streams as usual
    auto sweep = [&](int V) -> int {
        const int G = V >= 128 ? (save_groups < 1 ? 1 : (save_groups > 4 ? 4 : save_groups)) : (V >= 96 ? (save_groups > 3 ? 3 : (save_groups < 1 ? 1 : save_groups)) : 1);
        if (G > 1) hipEventRecord(h->fork_ev, st);
        int rc = HPF_OK;
        for (int g = 0; g < G && rc == HPF_OK; ++g) {
            const int a0 = (int)((long long)V * g / G), a1 = (int)((long long)V * (g + 1) / G);
            hipStream_t gs = G > 1 ? group_stream(h, g) : st;
            if (G > 1) hipStreamWaitEvent(gs, h->fork_ev, 0);
            h->cur_stream = gs;
            h->cur_s0 = v0 + a0;
            h->cur_S = a1 - a0;
            rc = tree_newton_step(h, false);
            if (G > 1) hipEventRecord(h->join_ev[g], gs);
        }
        if (G > 1)
            for (int g = 0; g < G; ++g) hipStreamWaitEvent(st, h->join_ev[g], 0);
        h->cur_stream = st;
        h->cur_s0 = s0;
        h->cur_S = cnt;
        return rc;
    };
    for (int r : todo) {
        for (int c0 = 0; c0 < 1 + m; c0 += VC) {
            const int V = 1 + m - c0 < VC ? 1 + m - c0 : VC;
            hipLaunchKernelGGL(k_border_prepare, dim3((unsigned)((cmax + 255) / 256), (unsigned)V), dim3(256), 0, st, n, h->Hn, BW, b, r, v0, c0,
                               h->d_tb_bus, h->n_tb, (const double*)nullptr, h->d_U, h->d_E, h->d_I0, h->d_fb);
            int rc = sweep(V);
            if (rc) return rc;
            hipLaunchKernelGGL(k_border_build, dim3((unsigned)((m + 255) / 256), (unsigned)V), dim3(256), 0, st, h->M, BW, b, r, v0, c0, m,
                               h->d_tb_bus, h->d_tb_ptr, h->d_tb_adj, h->d_U, h->d_E, h->d_x, h->d_bM, h->d_brhs);
        }
        if (ensure_blas(h) || rocblas_set_stream(h->blas, st) != rocblas_status_success) return HPF_E_ROCSOLVER;
        // Border system (I + Q^T Z) g = Q^T y.  rocSOLVER's pivoted getrf spends most of its time in thousands of tiny pivot-search / swap
        // kernels at these sizes (m = 520: a third of the whole bordered iteration, m = 2 080: 10 ms), so the LU runs WITHOUT pivoting first
        // (getrf_npvt + two triangular solves) and its solution is checked against a kept copy of the system: a residual above 1e-10 of the
        // right-hand side (pivot growth), a zero pivot or a non-finite entry sends the system through the pivoted LU.
        double* rwork = h->d_brhs0 + m;                   // [m] residual rhs - M g (d_brhs0[0, m) keeps the right-hand side itself)
        double* res = h->d_brhs0 + 2 * (size_t)m;         // [2]: max |rhs - M g|, max |rhs|
        int info = 0;
        double hres[2] = {0.0, 0.0};
        hipMemcpyAsync(h->d_bM0, h->d_bM, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(h->d_brhs0, h->d_brhs, sizeof(double) * m, hipMemcpyDeviceToDevice, st);
        hipMemcpyAsync(rwork, h->d_brhs, sizeof(double) * m, hipMemcpyDeviceToDevice, st);
        const double one = 1.0, neg = -1.0;
        if (rocsolver_dgetrf_npvt(h->blas, m, m, h->d_bM, m, h->d_binfo) != rocblas_status_success ||
            rocblas_dtrsv(h->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_unit, m, h->d_bM, m, h->d_brhs, 1) != rocblas_status_success ||
            rocblas_dtrsv(h->blas, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, m, h->d_bM, m, h->d_brhs, 1) != rocblas_status_success)
            return HPF_E_ROCSOLVER;
        hipLaunchKernelGGL(k_border_absmax, dim3(1), dim3(1024), 0, st, m, (const double*)h->d_brhs0, res + 1);
        if (rocblas_dgemv(h->blas, rocblas_operation_none, m, m, &neg, h->d_bM0, m, h->d_brhs, 1, &one, rwork, 1) != rocblas_status_success)
            return HPF_E_ROCSOLVER;                       // rwork <- rhs - M g
        hipLaunchKernelGGL(k_border_absmax, dim3(1), dim3(1024), 0, st, m, (const double*)rwork, res);
        if (hipMemcpyAsync(&info, h->d_binfo, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(hres, res, sizeof(double) * 2, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            h->last_detail = (int)hipGetLastError();
            return HPF_E_HIP;
        }
        if (h->sw("HPF_BORDER_INFO"))
            fprintf(stderr, "hpf border system (m = %d, scenario %d): |rhs - M g| / |rhs| = %.2e, info %d\n", m, r, hres[1] > 0.0 ? hres[0] / hres[1] : 0.0, info);
        if (h->border_pivoting || info != 0 || !(hres[0] <= 1e-10 * hres[1]) || !(hres[1] < INFINITY)) {
            hipMemcpyAsync(h->d_bM, h->d_bM0, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToDevice, st);       // the kept system, untouched
            hipMemcpyAsync(h->d_brhs, h->d_brhs0, sizeof(double) * m, hipMemcpyDeviceToDevice, st);
            ++h->border_repivots;
            if (rocsolver_dgetrf(h->blas, m, m, h->d_bM, m, h->d_bipiv, h->d_binfo) != rocblas_status_success ||
                rocsolver_dgetrs(h->blas, rocblas_operation_none, m, 1, h->d_bM, m, h->d_bipiv, h->d_brhs, m) != rocblas_status_success)
                return HPF_E_ROCSOLVER;
            if (hipMemcpyAsync(&info, h->d_binfo, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                h->last_detail = (int)hipGetLastError();
                return HPF_E_HIP;
            }
        }
        if (info != 0) {                                  // exactly singular border system (rocSOLVER info > 0)
            h->last_detail = r;
            return HPF_E_SINGULAR;
        }
        hipLaunchKernelGGL(k_border_prepare, dim3((unsigned)((cmax + 255) / 256), 1u), dim3(256), 0, st, n, h->Hn, BW, b, r, v0, 0,
                           h->d_tb_bus, h->n_tb, (const double*)h->d_brhs, h->d_U, h->d_E, h->d_I0, h->d_fb);
        int rc = sweep(1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_border_finish, dim3((unsigned)((std::max(cmax, VC) + 255) / 256)), dim3(256), 0, st, cmax, r, v0, VC, h->d_x,
                           h->d_pivflag);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

}  // namespace hpf
