// BLOCK_TREE factor / back-substitution kernels with ONE WORKGROUP OF NT WAVEFRONTS per (bus, scenario) pair.
//
// The bus block (b x b, padded to 16*NT with the right-hand side in column B) lives in FP64 MFMA accumulator tiles, as in
// hpf_gj_mfma.hpp, but wave `wv` of the workgroup owns tile COLUMN wv only (NT tiles = 8*NT VGPRs instead of 8*NT*NT), so 5-6
// waves fit on a SIMD instead of 2 and the latency-bound phases of one block (assembly loads, child Schur complements, stores)
// run NT-wide while other blocks keep the matrix cores busy.  Column ownership makes the blocked Gauss-Jordan step local:
//   - new pivot rows  W * A_P,:  of the own column tile: the old rows sit in the own accumulators in B-operand layout;
//   - the pivot columns (A operand of every wave) and W = A_PP^-1 come from the wave that owns tile column s>>2, through a
//     double-buffered LDS panel -> one workgroup barrier per block step, 1 + NT MFMAs per wave and step;
//   - the child -> parent Schur complement G A^-1 H (schur_tiles) needs lane^1 / lane^16 partners only: no cross-wave traffic.
// The un-eliminated block is assembled directly in tile layout: lane (lg, jj) of wave wv holds column 16*wv+jj = (harmonic
// p, component t') and rows 16*tr + lg + 4*reg, so its Norton cross terms (HG:425-435) are 4*NT independent Y_N loads times
// two per-lane constants.  The harmonic-diagonal part (network entry, linear children) is computed per row by wave 0 and
// patched in through LDS.
//
// HBM layout of the inverse for the back sweep: the tile image of TileIO (a lane's row groups pairwise adjacent: 16-byte
// accesses, 1 KB coalesced rows), the same layout as the Schur-complement slots and the per-model leaf images; w = A^-1 y
// additionally goes to the [bus][B] array shared with the linear-subtree kernels.
#pragma once

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// sum over the 16 lanes of a DPP row (all lanes receive it)
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);    // row_half_mirror
    v += dpp_f64<0x140>(v);    // row_mirror
    return v;
}

// Value forms of the per-entry Jacobian formulas of hpf_assembly.hpp (same operations, operands already in registers).
// blk_current: current-balance row, network entry y, column-bus voltages (HG:403-411).
__device__ __forceinline__ Blk2 blk_current(cplx y, cplx Uj, cplx Ej) {
    Blk2 b;
    b.dV = cmul_unf(y, Ej);
    b.dA = cmul_unf(cmulj(y), Uj);
    return b;
}
// diagonal current-balance entry incl. the p == h Norton term of a nonlinear bus (HG:432-435)
__device__ __forceinline__ Blk2 blk_current_diag(cplx y, cplx U, cplx E, cplx yn, bool nonlinear) {
    Blk2 b = blk_current(y, U, E);
    if (nonlinear) {
        b.dV = csub(b.dV, cmul_unf(yn, E));
        b.dA = csub(b.dA, cmul_unf(cmulj(yn), U));
    }
    return b;
}
// power row i, off-diagonal column j (HG:451-459, jac_power_entry with i != j)
__device__ __forceinline__ Blk2 blk_power_off(cplx y, cplx Ui, cplx Uj, cplx Ej) {
    const cplx yu = cmul_unf(y, Uj);
    const cplx ye = cmul_unf(y, Ej);
    Blk2 b;
    b.dV = cmul_unf(Ui, cconj(ye));
    b.dA = cmul_unf(cmulj(Ui), cconj(cneg(yu)));
    return b;
}
// power row, diagonal entry, row current I supplied (jac_power_diag)
__device__ __forceinline__ Blk2 blk_power_diag(cplx y, cplx U, cplx E, cplx I) {
    const cplx yu = cmul_unf(y, U);
    const cplx ye = cmul_unf(y, E);
    Blk2 b;
    b.dV = cadd(cmul_unf(E, cconj(I)), cmul_unf(U, cconj(ye)));
    b.dA = cmul_unf(cmulj(U), cconj(csub(I, yu)));
    return b;
}
// 2x2 real block of (row bus i, column bus j) at harmonic position q, masked to existing equations / unknowns (coupling_block)
__device__ __forceinline__ void mask_block(int n, int c, int q, int i, int j, const Blk2& blk, double out[4]) {
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
            out[tr * 2 + tc] = (loc_valid(n, c, i, 2 * q + tr) && loc_valid(n, c, j, 2 * q + tc)) ? pick(blk, tr, tc) : 0.0;
}

// HBM image of a block in accumulator-tile layout (Schur complements C, inverses Z).  A lane owns the B/4 row groups e = 4 tr + reg
// (rows 16 tr + 4 reg + lg < B) of its column.  Consecutive row groups (2p, 2p+1) are interleaved so that the lane's two values are
// adjacent: one 16-byte access per lane and pair (1 KB per wave instruction, half the memory requests of 8-byte accesses); B/4 is
// odd, the last row group stays 8 bytes wide.  Of the LAST tile column only the first LW columns exist (matrix columns
// 16(NT-1)..B-1 and the right-hand side in column B; B = 52: 5 of 16): its rows are stored LW wide.
template <int B>
struct TileIO {
    static constexpr int NT = (B + 16) / 16;
    static constexpr int LW = (B + 1 - 16 * (NT - 1)) <= 8 ? 8 : 16;
    static constexpr int RG = (NT - 1) * 64 + 4 * LW;                  // doubles per row group
    static constexpr int NE = B / 4, NP = NE / 2;                      // row groups, pairs of row groups
    static_assert(NE == 2 * NP + 1, "odd number of row groups");
    __device__ static __forceinline__ size_t off2(int p, int wv, int lg, int jj) {      // first of the two adjacent values
        return (size_t)p * 2 * RG + (wv < NT - 1 ? wv * 128 + (lg * 16 + jj) * 2 : (NT - 1) * 128 + (lg * LW + jj) * 2);
    }
    __device__ static __forceinline__ size_t off1(int wv, int lg, int jj) {             // the unpaired last row group
        return (size_t)NP * 2 * RG + (wv < NT - 1 ? wv * 64 + lg * 16 + jj : (NT - 1) * 64 + lg * LW + jj);
    }
    __device__ static __forceinline__ size_t off(int tr, int reg, int wv, int lg, int jj) {
        const int e = tr * 4 + reg;
        return e < 2 * NP ? off2(e >> 1, wv, lg, jj) + (e & 1) : off1(wv, lg, jj);
    }
    __device__ static __forceinline__ bool ok(int wv, int jj) { return wv < NT - 1 || jj < LW; }
    // the lane's NE values, v[e] = row group e (entries e >= NE untouched); lanes outside the stored columns get zeros
    __device__ static __forceinline__ void load(const double* base, int wv, int lg, int jj, double (&v)[NT * 4]) {
        const bool in = ok(wv, jj);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double2 t = {0.0, 0.0};
            if (in) t = *reinterpret_cast<const double2*>(base + off2(p, wv, lg, jj));
            v[2 * p] = t.x;
            v[2 * p + 1] = t.y;
        }
        v[NE - 1] = in ? base[off1(wv, lg, jj)] : 0.0;
    }
    __device__ static __forceinline__ void store(double* base, int wv, int lg, int jj, const double (&v)[NT * 4]) {
        if (!ok(wv, jj)) return;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<double2*>(base + off2(p, wv, lg, jj)) = double2{v[2 * p], v[2 * p + 1]};
        base[off1(wv, lg, jj)] = v[NE - 1];
    }
};

// the lane's NT A operands of one lazy-leaf pair, stored [lane][NT]
template <int NT>
__device__ __forceinline__ void lz_aop(const double* p, double (&a)[NT]) {
    if constexpr (NT % 2 == 0) {
#pragma unroll
        for (int t2 = 0; t2 < NT / 2; ++t2) {
            const double2 v = *reinterpret_cast<const double2*>(p + 2 * t2);
            a[2 * t2] = v.x;
            a[2 * t2 + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = p[t];
    }
}

// orders the LDS accesses of ONE wavefront (writes of some lanes before reads of others) where no workgroup barrier does: waits for the
// wave's outstanding LDS operations and keeps the compiler from moving memory accesses across
#define HPF_WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#ifndef HPF_LEAF_XCD
#define HPF_LEAF_XCD 1     // leaf-only launches: all scenarios of a leaf on one XCD (0: the plain (leaf, scenario) grid; A/B)
#endif
#ifndef HPF_Q100L_OCC
#define HPF_Q100L_OCC 2    // ... and of its leaf-only instantiation
#endif
#ifndef HPF_Q100_OCC
#define HPF_Q100_OCC 2     // the same for 52 < B <= 100 (7 waves per workgroup: 2 -> one workgroup per CU, 4 -> two)
#endif
#ifndef HPF_Q_OCC
#define HPF_Q_OCC 4        // waves per SIMD the B = 52 factor kernel is compiled for (register budget 512 / HPF_Q_OCC)
#endif

// LDS arrays of the factor body.  k_factor_q hands in its own static arrays (the ones an instantiation never touches -- most of
// them for LEAF -- then cost nothing); k_level carves them from the buffer it shares with the scenario-batched bodies.
template <int B>
struct FqLds {
    static constexpr int NT = (B + 16) / 16, RP = 16 * NT > 64 ? 16 * NT : 64;
    cplx* ynl;                      // [(B/2)^2]   Y_N of the bus's device type
    double* tab;                    // [(B/2) * 8]
    double* dgb;                    // [RP * 3]    per row: harmonic-diagonal 2x2 part (d0, d1) and right-hand side y
    double (*cc)[RP * 3];           // [NT]        the linear children's contributions to (d0, d1, y), one slot per wave
    double (*panel)[NT * 64];       // [2]
    double (*wl)[16];               // [2]
    double (*pv)[16];               // [2]
    double* gl;                     // [NT * 32]
    double* hl;                     // [NT * 32]
    double* slb;                    // [2 * B * 10 + 204]  super-leaf constants (B <= 52)
    double* gl2;                    // [NT * 32]   compress step: A(c, k) and A(k, c) of the pending child c (B <= 52: inside slb -- a
    double* hl2;                    // [NT * 32]   compressed bus is never a bordered one)
    double* pbuf;                   // [min(B, 64) * (B + 1)]  dense push of a bus with a dense coupling block: the whole buffer, reused
    __device__ __forceinline__ static FqLds carve(double* smem_) {
        FqLds L;
        L.ynl = reinterpret_cast<cplx*>(smem_);                  // (first: 16-byte aligned)
        L.tab = smem_ + 2 * (B / 2) * (B / 2);
        L.dgb = L.tab + (B / 2) * 8;
        L.cc = reinterpret_cast<double (*)[RP * 3]>(L.dgb + RP * 3);
        L.panel = reinterpret_cast<double (*)[NT * 64]>(L.dgb + RP * 3 + NT * RP * 3);
        L.wl = reinterpret_cast<double (*)[16]>(L.dgb + RP * 3 + NT * RP * 3 + 2 * NT * 64);
        L.pv = reinterpret_cast<double (*)[16]>(L.dgb + RP * 3 + NT * RP * 3 + 2 * NT * 64 + 32);
        L.gl = L.dgb + RP * 3 + NT * RP * 3 + 2 * NT * 64 + 64;
        L.hl = L.gl + NT * 32;
        L.slb = L.hl + NT * 32;
        L.gl2 = B <= 52 ? L.slb : L.slb + 2;
        L.hl2 = L.gl2 + NT * 32;
        L.pbuf = smem_;
        return L;
    }
};

// LDS of the factor kernel in doubles when carved from one buffer (FqLds::carve)
template <int B>
constexpr int factor_q_lds() {
    constexpr int NT = (B + 16) / 16, RP = 16 * NT > 64 ? 16 * NT : 64;
    return (B / 2) * 8 + RP * 3 + NT * RP * 3 + 2 * NT * 64 + 32 + 32 + NT * 32 + NT * 32 + 2 * (B / 2) * (B / 2) +
           (B <= 52 ? 2 * B * 10 + 200 + 4 : 2 + 2 * NT * 32);
}
static_assert(factor_q_lds<52>() >= 52 * 53 && factor_q_lds<28>() >= 28 * 29 && factor_q_lds<12>() >= 12 * 13 && factor_q_lds<100>() >= 64 * 101,
              "dense-push buffer of a compress step fits the factor body's LDS");

template <int B, bool LEAF>
__device__ __forceinline__ void factor_q_body(
    const FqLds<B>& lds_, const int bx_, const int by_, const Model& M, const TreeDev& T, const int* __restrict__ nodes, int b, int N, int Nc, const int* __restrict__ active,
    const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, const double* __restrict__ fall, double* __restrict__ Zall,
    double* __restrict__ wall, const double* __restrict__ linAall, double* __restrict__ Call, double* __restrict__ Hall,
    const cplx* __restrict__ I0all, const double* __restrict__ chG, const double* __restrict__ chH,
    const double* __restrict__ chD, const double* __restrict__ chy, const double* __restrict__ Minv,
    double* __restrict__ lfK, double* __restrict__ lfS, long long* __restrict__ dbg, int ablate, int s0,
    int* __restrict__ pivflag, double piv_limit, unsigned long long* __restrict__ tstamp) {
    constexpr int NT = (B + 16) / 16;
    constexpr size_t CT = (size_t)NT * NT * 256;
    constexpr int tcB = B >> 4, jjB = B & 15;
    constexpr int RP = 16 * NT > 64 ? 16 * NT : 64;     // rows the per-row roles cover (one lane per row, 64 per wave: B > 64 takes two waves)
    constexpr bool SPECIAL = B <= 52;                   // lazy leaves and super-leaves exist (tree_build: b <= 52 only; constant-inverse leaves: every b)
    HPF_STAMP_DECL;      // cycle stamps of wave 0: -DHPF_FACTOR_STAMPS build only
#ifdef HPF_FACTOR_STAMPS
    long long sa = 0, sc = 0, gown = 0, gwait = 0, gmf = 0;
#endif
    HPF_STAMP(st0);
    const int s = active ? active[by_ + s0] : (int)by_ + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    // timing leg only (hpf_timing_enable): first start / last end of the launch's workgroups on the device's constant-rate clock
    if (tstamp && threadIdx.x == 0) atomicMin(tstamp, (unsigned long long)wall_clock64());
    // node record: everything the block needs to form its addresses, behind one scalar load (Tree::d_fdesc)
    const int4* nd = reinterpret_cast<const int4*>(nodes) + (FDESC / 4) * (size_t)bx_;
    const int4 nd0 = nd[0], nd1 = nd[1], nd2 = nd[2], nd3 = nd[3];
    const int k = nd0.x, par = nd0.y, diag_e = nd0.z, devk = nd0.w;
    const int e_dn_k = nd1.x, e_up_k = nd1.y, lin_beg = nd1.z, lin_end = nd1.z + nd1.w;
    const int den_beg = nd2.x, n_den = nd2.y;
    const bool via_chain = (nd3.z & 1) != 0; // linked to the dense parent through a contracted chain (k_chain_factor)
    const bool lazy_leaf = SPECIAL && (nd3.z & 2) != 0; // the parent rebuilds this leaf's Schur complement itself: only G w goes to HBM
    const int cleafv = nd3.w;                // constant-inverse leaf: 1 + slot in Minv (0: general path; < 0: lazy-leaf record)
    const bool cleaf = LEAF || cleafv > 0;
    const bool sleaf = SPECIAL && !LEAF && (nd3.z & 4) != 0;   // super-leaf: every dense child is a lazy leaf -> bordered low-rank inverse, no Gauss-Jordan
    const bool slback = SPECIAL && !LEAF && (nd3.z & 8) != 0;  // ... whose back sweep rebuilds D^-1 t from T^-1 (k_sleaf_back_batch): no inverse goes to HBM
    const bool lazy = SPECIAL && !LEAF && cleafv < 0;   // this bus has lazy leaves below it
    const bool cleafr = cleaf || sleaf;      // roles of a constant-part bus: S^-1 staged, network diagonal lives in the images
    // compress steps on the Gauss-Jordan skeleton (tree_build_into, "compress"): role 1 = this bus v is eliminated BEFORE its pending
    // child c (four pushes: onto the parent p, onto c, and the two dense fill blocks between p and c); role 2 = this bus is such a
    // child: one more Schur complement (from v), and its coupling with its new parent p is the dense pair (Gd, Hd) left by v
    int4 cpA = {0, 0, 0, 0}, cpB = {0, 0, 0, 0};
    if (!LEAF) {
        cpA = nd[10];
        cpB = nd[11];
    }
    const int crole = LEAF ? 0 : cpA.x;
    const bool cmp_v = crole == 1, cmp_c = crole == 2;
    const int cmp_ci = cpA.y;
    // lazy-leaf record (Tree::d_lzrec, 8 ints: image offset, L, leaf ids[4])
    int4 lzA = {0, 0, -1, -1}, lzB = {-1, -1, 0, 0}, lzC = {-1, -1, 0, 0};
    if (lazy) {                               // (inline copy of the record: ints 28..39 of the node record, same scalar round trip)
        lzA = nd[7];
        lzB = nd[8];
        lzC = nd[9];                          // lazy super-leaf children: buses, their numbers of border unknowns (8 bits each)
    }
#ifdef HPF_FACTOR_STAMPS
    long long sd1 = 0, sd2 = 0, sd3 = 0, se1 = 0, se2 = 0;
    {
        int kk = k;
        asm volatile("" : "+s"(kk));
        sd1 = __builtin_amdgcn_s_memtime();
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = M.n, c = M.c, Hn = M.Hn;
    const size_t so = (size_t)s * n * Hn;
    const cplx* U = Uall + so;
    const cplx* E = Eall + so;
    const double* Cs = Call + (size_t)s * n * CT;

    cplx* const ynl = lds_.ynl;
    double* const tab = lds_.tab;
    double* const dgb = lds_.dgb;
    double (*const cc)[RP * 3] = lds_.cc;
    double (*const panel)[NT * 64] = lds_.panel;
    double (*const wl)[16] = lds_.wl;
    double (*const pv)[16] = lds_.pv;
    double* const gl = lds_.gl;
    double* const hl = lds_.hl;
    double* const slb = lds_.slb;

    const bool nl = k >= M.m && M.coupled;
    const int col = 16 * wv + jj, p = col >> 1, t1 = col & 1;      // own column = (harmonic position p, component t1)
    const int t = lg & 1;                                          // component of every row this lane holds

    // ---- A0. the device type's Y_N (Hn x Hn complex, shared by every scenario: L2) -> LDS, once per block, by the upper half of
    //      the waves (their roles are the light ones): loaded and stored at once -- held in registers across the roles the values
    //      get spilled to scratch, and the reload costs more than the L2 round trip exposed here --------------------------------
    if (nl && !cleafr && wv >= NT / 2) {
        constexpr int YT = 64 * (NT - NT / 2);                                   // staging threads
        constexpr int YNL = ((B / 2) * (B / 2) + YT - 1) / YT;
        const double2* ynd = reinterpret_cast<const double2*>(M.YN + (size_t)devk * Hn * Hn);
        double2* yl = reinterpret_cast<double2*>(ynl);
        const int t0 = tid - 64 * (NT / 2);
        if constexpr (YNL <= 6) {
            double2 y0 = {0.0, 0.0}, y1 = y0, y2 = y0, y3 = y0, y4 = y0, y5 = y0;      // (named scalars: an array here goes to scratch)
            if (YNL > 0 && t0 < Hn * Hn) y0 = ynd[t0];
            if (YNL > 1 && t0 + YT < Hn * Hn) y1 = ynd[t0 + YT];
            if (YNL > 2 && t0 + 2 * YT < Hn * Hn) y2 = ynd[t0 + 2 * YT];
            if (YNL > 3 && t0 + 3 * YT < Hn * Hn) y3 = ynd[t0 + 3 * YT];
            if (YNL > 4 && t0 + 4 * YT < Hn * Hn) y4 = ynd[t0 + 4 * YT];
            if (YNL > 5 && t0 + 5 * YT < Hn * Hn) y5 = ynd[t0 + 5 * YT];
            __builtin_amdgcn_sched_barrier(0);
            if (YNL > 0 && t0 < Hn * Hn) yl[t0] = y0;
            if (YNL > 1 && t0 + YT < Hn * Hn) yl[t0 + YT] = y1;
            if (YNL > 2 && t0 + 2 * YT < Hn * Hn) yl[t0 + 2 * YT] = y2;
            if (YNL > 3 && t0 + 3 * YT < Hn * Hn) yl[t0 + 3 * YT] = y3;
            if (YNL > 4 && t0 + 4 * YT < Hn * Hn) yl[t0 + 4 * YT] = y4;
            if (YNL > 5 && t0 + 5 * YT < Hn * Hn) yl[t0 + 5 * YT] = y5;
        } else {
            for (int i = t0; i < Hn * Hn; i += YT) yl[i] = ynd[i];
        }
    }
    // ---- A1. roles before the first barrier.  Every role FIRST issues all its loads (addresses come from the node record
    //      alone), then computes from registers with the value forms of the per-entry formulas (blk_current / blk_power_off /
    //      blk_power_diag): one memory round trip per role instead of one per operand group.
    //      Last wave: bus voltages of the Norton cross terms -> LDS --------------------------------------------------------------
    if ((nl || lazy) && wv == NT - 1 && lane < B / 2) {
        cplx u = {0.0, -1.0}, e = {0.0, 1.0};               // padding harmonics: S = [-ui er; ur ei] = identity
        if (lane < Hn) {
            u = U[(size_t)k * Hn + lane];
            e = E[(size_t)k * Hn + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        double* t0 = tab + lane * 4;
        if (cleafr) {
            // polar -> rectangular map of harmonic q: S = [dU/dtheta | dU/dV] = [-ui er; ur ei]; keep S^-1 (row-major)
            const double idet = 1.0 / (-(u.im * e.im) - e.re * u.re);
            t0[0] = e.im * idet;   t0[1] = -e.re * idet;
            t0[2] = -u.re * idet;  t0[3] = -u.im * idet;
        } else if (lane < Hn) {
            double* t1p = tab + (B / 2) * 4 + lane * 4;
            t0[0] = u.re;   t0[1] = u.im;    t0[2] = e.im;    t0[3] = -e.re;
            t1p[0] = u.im;  t1p[1] = -u.re;  t1p[2] = -e.re;  t1p[3] = -e.im;
        }
    }
    // wave 1: coupling blocks with the parent, G = A(parent, k) and H = A(k, parent), for the push (E) and the back sweep
    if (par >= 0 && !cmp_c && wv == (NT > 1 ? 1 : 0) && lane < NT * 8) {
        double g4[4] = {0.0, 0.0, 0.0, 0.0}, h4[4] = {0.0, 0.0, 0.0, 0.0};
        cplx lzu = {0.0, -1.0}, lze = {0.0, 1.0}, lzup = {0.0, -1.0}, lzep = {0.0, 1.0};
        if (lazy_leaf && lane == 0) {        // fundamental voltages of the leaf and of its dense parent (polar maps S_c, S_p at q = 0)
            lzu = U[(size_t)k * Hn];
            lze = E[(size_t)k * Hn];
            lzup = U[(size_t)par * Hn];
            lzep = E[(size_t)par * Hn];
        }
        if (lane < Hn) {
            const int q = lane;
            if (via_chain) {
                const double2* pg = reinterpret_cast<const double2*>(chG + (so + (size_t)k * Hn + q) * 4);
                const double2* ph = reinterpret_cast<const double2*>(chH + (so + (size_t)k * Hn + q) * 4);
                const double2 ga = pg[0], gb = pg[1], ha = ph[0], hb = ph[1];
                g4[0] = ga.x; g4[1] = ga.y; g4[2] = gb.x; g4[3] = gb.y;
                h4[0] = ha.x; h4[1] = ha.y; h4[2] = hb.x; h4[3] = hb.y;
            } else {
                const cplx ydn = M.Y[(size_t)e_dn_k * Hn + q], yup = M.Y[(size_t)e_up_k * Hn + q];
                const cplx uk = U[(size_t)k * Hn + q], ek = E[(size_t)k * Hn + q];
                const cplx up = U[(size_t)par * Hn + q], ep = E[(size_t)par * Hn + q];
                __builtin_amdgcn_sched_barrier(0);
                const Blk2 g = (q == 0 && par < M.m) ? blk_power_off(ydn, up, uk, ek) : blk_current(ydn, uk, ek);   // row par, col k
                const Blk2 hh = (q == 0 && k < M.m) ? blk_power_off(yup, uk, up, ep) : blk_current(yup, up, ep);     // row k, col par
                mask_block(n, c, q, par, k, g, g4);
                mask_block(n, c, q, k, par, hh, h4);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gl[lane * 4 + e] = g4[e];
            hl[lane * 4 + e] = h4[e];
        }
        if (lazy_leaf && lane == 0) {
            // the state-dependent harmonic position 0 of the borders the parent rebuilds the Schur complement from:
            // G0 S_c^-1 and H0 S_p^-1 (G0, H0 = the polar 2x2 blocks just formed: line, power row or contracted chain alike)
            const double ic = 1.0 / (-(lzu.im * lze.im) - lze.re * lzu.re), ip = 1.0 / (-(lzup.im * lzep.im) - lzep.re * lzup.re);
            const double sc0 = lze.im * ic, sc1 = -lze.re * ic, sc2 = -lzu.re * ic, sc3 = -lzu.im * ic;
            const double sp0 = lzep.im * ip, sp1 = -lzep.re * ip, sp2 = -lzup.re * ip, sp3 = -lzup.im * ip;
            double* kk = lfK + ((size_t)s * n + k) * 12 + 4;
            kk[0] = fma(g4[1], sc2, g4[0] * sc0);  kk[1] = fma(g4[1], sc3, g4[0] * sc1);
            kk[2] = fma(g4[3], sc2, g4[2] * sc0);  kk[3] = fma(g4[3], sc3, g4[2] * sc1);
            kk[4] = fma(h4[1], sp2, h4[0] * sp0);  kk[5] = fma(h4[1], sp3, h4[0] * sp1);
            kk[6] = fma(h4[3], sp2, h4[2] * sp0);  kk[7] = fma(h4[3], sp3, h4[2] * sp1);
        }
        if (lane < Hn) {
            double* Hk = Hall + ((size_t)s * n + k) * Hn * 4 + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) Hk[e] = h4[e];
        }
    }
    // compress step: G2 = A(c, k), H2 = A(k, c) of the pending child c (the line itself or what the contracted chain between the two
    // left of it, exactly as the child's own role would form them); H2 is kept for the back sweep (x_k needs x_c as well)
    if (cmp_v && wv == (NT > 2 ? 2 : 0) && lane < NT * 8) {
        double g4[4] = {0.0, 0.0, 0.0, 0.0}, h4[4] = {0.0, 0.0, 0.0, 0.0};
        const int cb = cpA.z;
        if (lane < Hn) {
            const int q = lane;
            if (cpB.y) {                                     // chG[c] = A'(k, c), chH[c] = A'(c, k)
                const double2* pg = reinterpret_cast<const double2*>(chG + (so + (size_t)cb * Hn + q) * 4);
                const double2* ph = reinterpret_cast<const double2*>(chH + (so + (size_t)cb * Hn + q) * 4);
                const double2 ga = pg[0], gb = pg[1], ha = ph[0], hb = ph[1];
                h4[0] = ga.x; h4[1] = ga.y; h4[2] = gb.x; h4[3] = gb.y;
                g4[0] = ha.x; g4[1] = ha.y; g4[2] = hb.x; g4[3] = hb.y;
            } else {
                const cplx ydn = M.Y[(size_t)cpA.w * Hn + q], yup = M.Y[(size_t)cpB.x * Hn + q];   // entries (k, c), (c, k)
                const cplx uk = U[(size_t)k * Hn + q], ek = E[(size_t)k * Hn + q];
                const cplx uc = U[(size_t)cb * Hn + q], ec = E[(size_t)cb * Hn + q];
                __builtin_amdgcn_sched_barrier(0);
                const Blk2 hh = (q == 0 && k < M.m) ? blk_power_off(ydn, uk, uc, ec) : blk_current(ydn, uc, ec);     // row k, col c
                const Blk2 g = (q == 0 && cb < M.m) ? blk_power_off(yup, uc, uk, ek) : blk_current(yup, uk, ek);     // row c, col k
                mask_block(n, c, q, cb, k, g, g4);
                mask_block(n, c, q, k, cb, hh, h4);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lds_.gl2[lane * 4 + e] = g4[e];
            lds_.hl2[lane * 4 + e] = h4[e];
        }
        if (lane < Hn) {
            double* Hk = T.cH2 + ((size_t)s * T.n_comp + cmp_ci) * Hn * 4 + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) Hk[e] = h4[e];
        }
    }
    {
        // lane = row 2q+tr_: wave 0 forms the network part of the harmonic-diagonal 2x2 and the right-hand side; the children
        // folded in 2x2-per-harmonic algebra (linear subtrees, contracted chains) are dealt to the waves, lightest roles first
        // (B > 64: rows 64.. are formed by wave 1 and the children loop takes a second slice of rows)
#pragma unroll
        for (int rb = 0; rb < RP; rb += 64) {
        const int rw = rb + lane;
        const int q = rw >> 1, tr_ = rw & 1;
        const bool rowvalid = rw < b && loc_valid(n, c, k, rw);
        const size_t kq = (size_t)k * Hn + (q < Hn ? q : 0);
        if (wv == rb / 64) {
            double y = 0.0, d0 = 0.0, d1 = 0.0;
            if (rowvalid) {
                const bool prow = q == 0 && k < M.m;                     // power row (HG:451-459)
                const double fy = fall[((size_t)s * n + k) * B + rw];  // bus-major mismatch image (k_mismatch)
                const cplx yd = M.Y[(size_t)diag_e * Hn + q];
                const cplx uk = U[kq], ek = E[kq];
                cplx yn = {0.0, 0.0}, I0v = {0.0, 0.0};
                if (k >= M.m) yn = M.coupled ? M.YN[((size_t)devk * Hn + q) * Hn + q] : M.YN[(size_t)devk * Hn + q];
                if (prow) I0v = I0all[(size_t)s * n + k];                // kept by the mismatch kernel of this very state
                double a0 = 0.0, a1 = 0.0, ay = 0.0;
                if (lazy) {                                              // G w of the lazy leaves (rows of their slots): y -= sum
                    const double g0 = lzA.z >= 0 ? Cs[(size_t)lzA.z * CT + lane] : 0.0;
                    const double g1 = lzA.w >= 0 ? Cs[(size_t)lzA.w * CT + lane] : 0.0;
                    const double g2 = lzB.x >= 0 ? Cs[(size_t)lzB.x * CT + lane] : 0.0;
                    const double g3 = lzB.y >= 0 ? Cs[(size_t)lzB.y * CT + lane] : 0.0;
                    const double g4s = lzC.x >= 0 ? Cs[(size_t)lzC.x * CT + lane] : 0.0;
                    const double g5s = lzC.y >= 0 ? Cs[(size_t)lzC.y * CT + lane] : 0.0;
                    ay = -(((g0 + g1) + (g2 + g3)) + (g4s + g5s));
                }
                if (via_chain) {                                         // what the elimination of the chain above left here
                    const size_t o = so + (size_t)k * Hn + q;
                    a0 = chD[o * 4 + 2 * tr_];
                    a1 = chD[o * 4 + 2 * tr_ + 1];
                    ay += chy[o * 2 + tr_];
                }
                __builtin_amdgcn_sched_barrier(0);
                const Blk2 blk = prow ? blk_power_diag(yd, uk, ek, I0v) : blk_current_diag(yd, uk, ek, yn, k >= M.m);
                // (constant-inverse leaf: the network part lives in the precomputed inverse; rows 0/1 carry the state-dependent
                //  2x2 term of the fundamental, i.e. what the 2x2-algebra neighbours left there)
                d0 = ((cleafr && !prow) ? 0.0 : pick(blk, tr_, 0)) + a0;     // (power rows of a linear super-leaf: state dependent, kept)
                d1 = ((cleafr && !prow) ? 0.0 : pick(blk, tr_, 1)) + a1;
                y = fy + ay;
            }
#ifdef HPF_FACTOR_STAMPS
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(y));
            sd2 = __builtin_amdgcn_s_memtime();
#endif
            if (rw < RP) {                                  // (RP = 112 at B = 100: the second slice is 48 rows)
                dgb[rw * 3 + 0] = d0;
                dgb[rw * 3 + 1] = d1;
                dgb[rw * 3 + 2] = y;
            }
        }
        const int slot = (wv + NT - (2 % NT)) % NT;          // waves 2, 3 have the lightest roles: they take the first children
        if (lin_beg + slot < lin_end) {
            double e0 = 0.0, e1 = 0.0, ey = 0.0;
            // the first child of the slot comes with the node record (ints 16..27: child, e_dn, e_up), later ones through child3
            const int sl4 = slot < 4 ? slot : 0;
            int4 cr = {nodes[FDESC * (size_t)bx_ + 16 + 3 * sl4], nodes[FDESC * (size_t)bx_ + 17 + 3 * sl4],
                       nodes[FDESC * (size_t)bx_ + 18 + 3 * sl4], 0};
            if (rowvalid) {
                const double* ws = wall + (size_t)s * n * B;
                const double* linA = linAall + so * 4;
                const int4* c3 = reinterpret_cast<const int4*>(T.child3);
                const cplx uk = U[kq], ek = E[kq];
                for (int cp = lin_beg + slot; cp < lin_end; cp += NT) {
                    if (cp != lin_beg + slot || slot >= 4) cr = c3[cp];
                    const int ch = cr.x;
                    const cplx ydn = M.Y[(size_t)cr.y * Hn + q], yup = M.Y[(size_t)cr.z * Hn + q];
                    const cplx uc = U[(size_t)ch * Hn + q], ec = E[(size_t)ch * Hn + q];
                    const double2* pic = reinterpret_cast<const double2*>(linA + ((size_t)ch * Hn + q) * 4);
                    const double2 ic01 = pic[0], ic23 = pic[1];
                    const double2 wc = *reinterpret_cast<const double2*>(ws + (size_t)ch * B + 2 * q);
                    __builtin_amdgcn_sched_barrier(0);
                    const Blk2 g = (q == 0 && k < M.m) ? blk_power_off(ydn, uk, uc, ec) : blk_current(ydn, uc, ec);   // A(k, child)
                    const Blk2 hb = (q == 0 && ch < M.m) ? blk_power_off(yup, uc, uk, ek) : blk_current(yup, uk, ek);  // A(child, k)
                    const double g0 = pick(g, tr_, 0);
                    const double g1 = loc_valid(n, c, ch, 2 * q + 1) ? pick(g, tr_, 1) : 0.0;
                    double h4[4];
                    mask_block(n, c, q, ch, k, hb, h4);
                    const double v0 = fma(g1, ic23.x, g0 * ic01.x), v1 = fma(g1, ic23.y, g0 * ic01.y);
                    e0 += fma(v1, h4[2], v0 * h4[0]);
                    e1 += fma(v1, h4[3], v0 * h4[1]);
                    ey = fma(g0, wc.x, ey);
                    ey = fma(g1, wc.y, ey);
                }
            }
            if (rw < RP) {
                cc[slot][rw * 3 + 0] = e0;
                cc[slot][rw * 3 + 1] = e1;
                cc[slot][rw * 3 + 2] = ey;
            }
        }
    }
    }
    // ---- B0. Schur complement of the FIRST dense child (accumulator layout, own tile column): the loads are issued here, after
    //      the role loads (loads return in order: a role must not wait behind them), and first touched after the assembly ----------
#ifdef HPF_FACTOR_STAMPS
    se1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- L. lazy leaves (Tree::d_lzrec): the Schur complements of the constant-inverse leaves c hanging directly under this bus,
    //      sum_c A(k,c) D_c^-1 A(c,k) = [ R(sum_c C0_c) + sum_c Gc_c K_c Hr_c ] S_k,
    //      with per-model images (L2 / Infinity Cache) for the harmonics q >= 1 of the borders Gc (b x 2), Hr (2 x b) -- constant
    //      there even through a contracted chain --, their state-dependent position 0 (G0 S_c^-1, H0 S_k^-1: power rows of a PQ
    //      bus, chain buses) and the 2x2 cores K_c = (c0 + D)^-1 left by the leaf launch (lfK), and the polar map S_k of this bus
    //      on the columns.  Two leaves per rank-4 MFMA, at most 4 lazy leaves per bus.  Only G w comes from the leaves' slots (wave 0
    //      subtracts it from the right-hand side).  Nothing here needs the staged LDS data: the accumulation runs before the
    //      first barrier, the S / W maps are applied after it. ---------------------------------------------------------------------
    d4_t xt[NT];
    if (lazy && !sleaf) {
        const int np = (lzA.y + 1) >> 1;
        const double* aimg = T.lzimg + (size_t)lzA.x + CT;
        const double* himg = aimg + (size_t)np * NT * 64;
        const double* Ks = lfK + (size_t)s * n * 12 + 2 * (lg & 1);
        const int leaf0 = (lg >> 1) ? lzA.w : lzA.z, leaf1 = (lg >> 1) ? lzB.y : lzB.x;
        // every load of the phase is issued at once (addresses from the scalar record), then the MFMAs
        double2 k0 = {0.0, 0.0}, k1 = {0.0, 0.0};
        double h00 = 0.0, h01 = 0.0, h10 = 0.0, h11 = 0.0, a0[NT], a1[NT];
        if (leaf0 >= 0) k0 = *reinterpret_cast<const double2*>(Ks + (size_t)leaf0 * 12);
        if (leaf1 >= 0) k1 = *reinterpret_cast<const double2*>(Ks + (size_t)leaf1 * 12);
        // images: per pair the A operands [lane][NT] and the two rows of R(Hr) [tile column][lane][2] -- 16-byte accesses
        {
            const double2 h = *reinterpret_cast<const double2*>(himg + (size_t)(wv * 64 + lane) * 2);
            h00 = h.x;
            h01 = h.y;
        }
        lz_aop<NT>(aimg + (size_t)lane * NT, a0);
        if (np > 1) {
            const double2 h = *reinterpret_cast<const double2*>(himg + (size_t)((NT + wv) * 64 + lane) * 2);
            h10 = h.x;
            h11 = h.y;
            lz_aop<NT>(aimg + (size_t)(64 + lane) * NT, a1);
        } else {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) a1[tr] = 0.0;
        }
        // harmonic position 0 of the borders is per scenario (left by the leaves next to their cores): rows 0 / 1 of the A
        // operand = G0 S_c^-1, columns 0 / 1 of R(Hr) = H0 S_k^-1 (the S_k map below then restores H0)
        if (jj < 2) {
            const double* L0 = lfK + ((size_t)s * n + (leaf0 >= 0 ? leaf0 : 0)) * 12;
            const double* L1 = lfK + ((size_t)s * n + (leaf1 >= 0 ? leaf1 : 0)) * 12;
            if (leaf0 >= 0) a0[0] = L0[4 + jj * 2 + (lg & 1)];
            if (leaf1 >= 0) a1[0] = L1[4 + jj * 2 + (lg & 1)];
            if (wv == 0) {
                if (leaf0 >= 0) {
                    h00 = L0[8 + jj];
                    h01 = L0[10 + jj];
                }
                if (leaf1 >= 0) {
                    h10 = L1[8 + jj];
                    h11 = L1[10 + jj];
                }
            }
        }
        {
            double xv[NT * 4];
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) xv[e] = 0.0;
            TileIO<B>::load(T.lzimg + (size_t)lzA.x, wv, lg, jj, xv);     // sum of the leaves' constant parts (tile image)
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) xt[e >> 2][e & 3] = xv[e];
        }
        const double b0 = fma(k0.y, h01, k0.x * h00);                    // (K_c R(Hr_c))[a][col], a = lg & 1, c = leaf lg >> 1
        const double b1 = fma(k1.y, h11, k1.x * h10);
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) xt[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[tr], b0, xt[tr], 0, 0, 0);
        if (np > 1) {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) xt[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[tr], b1, xt[tr], 0, 0, 0);
        }
#ifdef HPF_FACTOR_STAMPS
        asm volatile("" : "+v"(xt[0]), "+v"(xt[NT - 1]));
        se2 = __builtin_amdgcn_s_memtime();
#endif
        // lazy super-leaf children (at most two): the same rebuild with the m x m core T^-1 the child left at the head of its
        // inverse slot, m <= 10 border unknowns in chunks of four; position 0 of the borders from the child's record
        // (G0 S_c^-1 on rows 0 / 1 of the first two unknowns, W_c^-1 H0 S_k^-1 on columns 0 / 1)
        // Every load of a child is issued in ONE batch (one memory round trip per child instead of one per chunk): the child's T^-1 | W^-1
        // (104 doubles, per scenario) goes through a per-wave LDS strip (slb is free in a Gauss-Jordan workgroup; its head holds the
        // compress role's blocks), the image columns and A operands stay in registers.
        const double* simgs = himg + (size_t)np * NT * 2 * 64;
        for (int zc = 0; zc < 2; ++zc) {
            const int child = zc == 0 ? lzC.x : lzC.y;
            if (child < 0) break;
            const int mz = (lzC.z >> (8 * zc)) & 0xff;
            const double* za = simgs + (size_t)zc * (3 * 64 * NT + 10 * B);
            const double* zq = za + 3 * 64 * NT;
            const double* tk = Zall + ((size_t)s * n + child) * CT;                 // T^-1 [10][10] | W^-1 [4]
            const double* kc = lfK + ((size_t)s * n + child) * 12;
            double* tl = slb + 2 * NT * 32 + (wv * 2 + zc) * 104;
            double2 tke = {0.0, 0.0};
            if (lane < 52) tke = reinterpret_cast<const double2*>(tk)[lane];
            double qh[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) qh[j] = (j < mz && col < b) ? zq[(size_t)j * B + col] : 0.0;
            double h0 = 0.0, h1 = 0.0, g0 = 0.0;
            if (col < 2) {
                h0 = kc[8 + col];
                h1 = kc[10 + col];
            }
            if (jj < 2 && lg < 2) g0 = kc[4 + jj * 2 + lg];                        // rows 0 / 1: (G0 S_c^-1)[jj][i], i = lg (first chunk)
            double az[3][NT];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr) az[ch][tr] = 0.0;
                if (4 * ch < mz) lz_aop<NT>(za + ((size_t)ch * 64 + lane) * NT, az[ch]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (lane < 52) *reinterpret_cast<double2*>(tl + 2 * lane) = tke;
            HPF_WAVE_LDS_FENCE();                                                // (one wave: the other lanes' writes before this lane's reads)
            if (col < 2) {                                               // (W^-1 H0 S_k^-1)[j][col], j < 2
                qh[0] = fma(tl[101], h1, tl[100] * h0);
                qh[1] = fma(tl[103], h1, tl[102] * h0);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (4 * ch < mz) {
                    const int i = 4 * ch + lg;
                    const double* tr_ = tl + (i < mz ? i : 0) * 10;
                    double bop = 0.0;
#pragma unroll
                    for (int j = 0; j < 10; ++j)
                        if (j < mz) bop = fma(tr_[j], qh[j], bop);
                    bop = i < mz ? bop : 0.0;
                    if (jj < 2) az[ch][0] = ch == 0 ? g0 : 0.0;
#pragma unroll
                    for (int tr = 0; tr < NT; ++tr) xt[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(az[ch][tr], bop, xt[tr], 0, 0, 0);
                }
            }
            HPF_WAVE_LDS_FENCE();
        }
    }
    double sumc[NT * 4];
#pragma unroll
    for (int e = 0; e < NT * 4; ++e) sumc[e] = 0.0;
    if (n_den > 0) TileIO<B>::load(Cs + (size_t)nd2.z * CT, wv, lg, jj, sumc);
    // super-leaf: stage the per-model constants and everything of T that does not depend on this bus's own roles (all of it but
    // the 2x2 term of the fundamental) while the roles' loads are in flight
    if (sleaf) {
        const int L = lzA.y, m = 2 + 2 * L, m2 = 2 * m;
        const double* simg = T.lzimg + (size_t)lzB.w;                    // Tc [m][m] | Pb [b][m] | Qb [m][b]
        double* aug = slb + 2 * B * 10;
        for (int idx = tid; idx < b * m; idx += 64 * NT) {
            slb[idx] = simg[m * m + idx];
            slb[B * 10 + idx] = simg[m * m + b * m + idx];
        }
        if (wv == (NT > 2 ? 2 : 0)) {
            if (lane < m)
                for (int r2 = 0; r2 < m; ++r2) aug[r2 * 20 + lane] = simg[r2 * m + lane];       // T <- Tc (columns m.. receive T^-1 later)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // linear (PQ) bus: its row pair 0 is  W_k (current-row form),  W_k = [ur ui; ui -ur]  (U conj(.)),  W_k^-1 = W_k / |U|^2:
            // the system is solved in the current-row form, W_k^-1 goes on row pair 0 of the borders and on columns 0 / 1 of Qb
            double w00 = 1.0, w01 = 0.0, w10 = 0.0, w11 = 1.0;
            if (k < M.m && lane <= L) {
                const cplx u0 = U[(size_t)k * Hn];
                const double iu = 1.0 / fma(u0.re, u0.re, u0.im * u0.im);
                w00 = u0.re * iu;
                w01 = u0.im * iu;
                w10 = w01;
                w11 = -w00;
            }
            if (lane == 0) {
                double* wi = aug + 10 * 20;                              // behind the 10 x 20 matrix: W_k^-1 for the later phases
                wi[0] = w00; wi[1] = w01; wi[2] = w10; wi[3] = w11;
            }
            if (lane >= 1 && lane <= L) {                                // leaf i: K_i^-1 on the diagonal, -G0 S_c^-1 / -H0 S_k^-1 borders
                const int i = lane - 1, bc = 2 + 2 * i;
                const int leaf = i == 0 ? lzA.z : (i == 1 ? lzA.w : (i == 2 ? lzB.x : lzB.y));
                const double* kk = lfK + ((size_t)s * n + leaf) * 12;
                double q00, q01, q10, q11;
                inv2(kk[0], kk[1], kk[2], kk[3], q00, q01, q10, q11);
                aug[bc * 20 + bc] += q00;
                aug[bc * 20 + bc + 1] += q01;
                aug[(bc + 1) * 20 + bc] += q10;
                aug[(bc + 1) * 20 + bc + 1] += q11;
                aug[bc] -= fma(w01, kk[6], w00 * kk[4]);                   // W_k^-1 (G0 S_c^-1)
                aug[bc + 1] -= fma(w01, kk[7], w00 * kk[5]);
                aug[20 + bc] -= fma(w11, kk[6], w10 * kk[4]);
                aug[20 + bc + 1] -= fma(w11, kk[7], w10 * kk[5]);
                aug[bc * 20] -= kk[8];
                aug[bc * 20 + 1] -= kk[9];
                aug[(bc + 1) * 20] -= kk[10];
                aug[(bc + 1) * 20 + 1] -= kk[11];
            }
        }
    }
    // leaf-only launches: the per-model image of the leaf (L2 / Infinity Cache) is requested here, behind the role loads, and
    // first touched after the barrier
    d4_t ct[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) ct[tr] = d4_t{0.0, 0.0, 0.0, 0.0};
    if (LEAF) {
        double mv[NT * 4];
#pragma unroll
        for (int e = 0; e < NT * 4; ++e) mv[e] = 0.0;
        TileIO<B>::load(Minv + (size_t)(cleafv - 1) * CT, wv, lg, jj, mv);
#pragma unroll
        for (int e = 0; e < NT * 4; ++e) ct[e >> 2][e & 3] = mv[e];
    }
    HPF_STAMP(sd3);
    __syncthreads();
    HPF_STAMP(sa);
    {
        const int nw = (lin_end - lin_beg) < NT ? (lin_end - lin_beg) : NT;
        for (int idx = tid; idx < RP * 3; idx += 64 * NT) {
            double v = dgb[idx];
            for (int w2 = 0; w2 < nw; ++w2) v -= cc[w2][idx];       // fixed order
            dgb[idx] = v;
        }
    }

    if (lazy && !sleaf) {
        const double* t0 = tab + (p < Hn ? p : 0) * 4;                   // [ur, ui, ei, -er] of the column's harmonic
        const bool mcol = col < B;                                       // (column B already holds G w: untouched)
        const double sown = mcol ? (t1 ? t0[2] : -t0[1]) : 1.0;          // S[t1][t1],  S = [-ui er; ur ei]
        const double soth = mcol ? (t1 ? -t0[3] : t0[0]) : 0.0;          // S[t1^1][t1]
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double v = xt[tr][reg];
                xt[tr][reg] = fma(xor1_f64(v), soth, v * sown);
            }
#pragma unroll
        for (int e = 0; e < NT * 4; ++e)
            if (16 * (e >> 2) + 4 * (e & 3) < B) sumc[e] += xt[e >> 2][e & 3];
    }

    if (cleaf) {
        // ================= constant-inverse leaf (Tree::d_Minv) =====================================================
        // In rectangular coordinates the block is  R(Yc) + E0 D E0^T : Yc constant, D = Delta_polar S_0^-1 the 2x2 term of the
        // fundamental.  With the per-model image  [c0 Lr; Lc Ahh^-1]  (tile layout):
        //     Drect^-1 = [0 0; 0 Ahh^-1] + [I; Lc] (c0 + D)^-1 [I Lr];
        // polar inverse = S^-1 Drect^-1 (row pairs scaled by the 2x2 S_q^-1);  w = A^-1 y by row sums.
        if (!LEAF) {
            double mv[NT * 4];
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) mv[e] = 0.0;
            TileIO<B>::load(Minv + (size_t)(cleafv - 1) * CT, wv, lg, jj, mv);
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) ct[e >> 2][e & 3] = mv[e];
        }
        double* mc = &panel[0][0];          // [I; Lc]  as mc[row*2 + a]
        double* mr = &panel[1][0];          // [I  Lr]  as mr[a*MRS + col]  (positions (a, 0..1) hold c0)
        constexpr int MRS = 16 * NT;        // columns of the padded block (panel[1] holds NT*64 doubles)
        if (wv == 0 && jj < 2) {
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) mc[(16 * tr + 4 * reg + lg) * 2 + jj] = ct[tr][reg];
        }
        if (lg < 2) mr[lg * MRS + col] = ct[0][0];
        __syncthreads();
        double kv0, kv1;                    // ((c0 + D)^-1 [I Lr])[a][col]
        {
            const double si0 = tab[0], si1 = tab[1], si2 = tab[2], si3 = tab[3];           // S_0^-1
            const double p00 = dgb[0], p01 = dgb[1], p10 = dgb[3], p11 = dgb[4];           // Delta_polar
            const double q00 = mr[0] + fma(p01, si2, p00 * si0), q01 = mr[1] + fma(p01, si3, p00 * si1);      // c0 + Delta_polar S_0^-1
            const double q10 = mr[MRS] + fma(p11, si2, p10 * si0), q11 = mr[MRS + 1] + fma(p11, si3, p10 * si1);
            double k00, k01, k10, k11;
            inv2(q00, q01, q10, q11, k00, k01, k10, k11);
            const double r0 = col < 2 ? (col == 0 ? 1.0 : 0.0) : mr[col];
            const double r1 = col < 2 ? (col == 1 ? 1.0 : 0.0) : mr[MRS + col];
            kv0 = fma(k01, r1, k00 * r0);
            kv1 = fma(k11, r1, k10 * r0);
            if (tid == 0) {                 // the back sweep rebuilds  A^-1 t  from the image, this 2x2 and S^-1: no inverse goes to HBM
                double* kk = lfK + ((size_t)s * n + k) * 12;
                kk[0] = k00; kk[1] = k01; kk[2] = k10; kk[3] = k11;
            }
            if (tid < Hn * 4) lfS[(so + (size_t)k * Hn) * 4 + tid] = tab[tid];
        }
        const double yc = col < B ? dgb[col * 3 + 2] : 0.0;
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (16 * tr + 4 * reg >= B) continue;
                const int row = 16 * tr + 4 * reg + lg;
                double v = (row < 2 || col < 2) ? 0.0 : ct[tr][reg];
                const double c0r = row < 2 ? (row == 0 ? 1.0 : 0.0) : mc[row * 2];
                const double c1r = row < 2 ? (row == 1 ? 1.0 : 0.0) : mc[row * 2 + 1];
                v = fma(c1r, kv1, fma(c0r, kv0, v));                                        // + [I; Lc] (c0 + D)^-1 [I Lr]
                const double pr = xor16_f64(v);                                             // the other row of the harmonic
                const double* si = tab + (row >> 1) * 4 + 2 * t;
                v = t ? fma(si[1], v, si[0] * pr) : fma(si[1], pr, si[0] * v);              // S_q^-1 from the left
                ct[tr][reg] = v;
                const double sm = row_sum16(v * yc);
                if (jj == 0) cc[wv][row] = sm;
            }
        __syncthreads();
        if (wv == tcB && jj == jjB) {                                                       // w = A^-1 y into column B
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (16 * tr + 4 * reg >= B) continue;
                    const int row = 16 * tr + 4 * reg + lg;
                    double acc = cc[0][row];
#pragma unroll
                    for (int w2 = 1; w2 < NT; ++w2) acc += cc[w2][row];
                    ct[tr][reg] = acc;
                }
        }
        HPF_STAMP(st1);
        HPF_STAMP(st3);
    } else if (sleaf) {
        // ================= super-leaf: all dense children are lazy leaves =================================
        //   Drect^-1 = [0 0; 0 Ahh^-1] + Pb T^-1 Qb,   T = Tc + blockdiag(D, K_1^-1, ..., K_L^-1) - borders(G0 S_c^-1, H0 S_k^-1),
        //   m = 2 + 2L <= 10: T is inverted in LDS by wave 0 (Gauss-Jordan, partial pivoting); Tc, Pb, Qb, the Ahh^-1 image: per model.
        const int L = lzA.y, m = 2 + 2 * L, m2 = 2 * m;
        double* aug = slb + 2 * B * 10;                                  // [m][20]: T | I  ->  I | T^-1   (staged before barrier 1)
        double* pbl = slb;                                               // Pb [b][m], then Qb [m][b]
        double* qbl = pbl + B * 10;
        if (wv == 0) {
            if (lane == 0) {                                             // D = Delta_polar S_0^-1 (the 2x2 term of the fundamental)
                const double si0 = tab[0], si1 = tab[1], si2 = tab[2], si3 = tab[3];
                const double p00 = dgb[0], p01 = dgb[1], p10 = dgb[3], p11 = dgb[4];
                const double e00 = fma(p01, si2, p00 * si0), e01 = fma(p01, si3, p00 * si1);
                const double e10 = fma(p11, si2, p10 * si0), e11 = fma(p11, si3, p10 * si1);
                const double* wi = aug + 10 * 20;                        // W_k^-1 (identity for a nonlinear bus)
                aug[0] += fma(wi[1], e10, wi[0] * e00);
                aug[1] += fma(wi[1], e11, wi[0] * e01);
                aug[20] += fma(wi[3], e10, wi[2] * e00);
                aug[21] += fma(wi[3], e11, wi[2] * e01);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // In-place Gauss-Jordan inversion with partial pivoting in registers: lane r owns row r of T; the pivot row is broadcast
            // with v_readlane, rows are never swapped.  The lane that was the pivot of column j ends up with row j of T^-1, and
            // its column position c belongs to the unit vector of the pivot lane of column c (pcol): T^-1[j][pcol[c]] = row[c].
            double row[10];
#pragma unroll
            for (int c2 = 0; c2 < 10; ++c2) row[c2] = (lane < m && c2 < m) ? aug[lane * 20 + c2] : 0.0;
            bool used = lane >= m;
            int mycol = 0, pcol[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                pcol[j] = 0;
                if (j < m) {                                             // (m is workgroup-uniform)
                    const double cand = used ? -1.0 : fabs(row[j]);
                    double mx = cand;                                    // max over the 16 lanes of the DPP row (all receive it)
                    mx = fmax(mx, dpp_f64<0xB1>(mx));
                    mx = fmax(mx, dpp_f64<0x4E>(mx));
                    mx = fmax(mx, dpp_f64<0x141>(mx));
                    mx = fmax(mx, dpp_f64<0x140>(mx));
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(!used && cand == mx);
                    const int pi = __builtin_ctzll(bal | (1ull << 63));
                    double prow[10];
#pragma unroll
                    for (int c2 = 0; c2 < 10; ++c2)
                        prow[c2] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(row[c2]), pi),
                                                    __builtin_amdgcn_readlane(__double2loint(row[c2]), pi));
                    const double ipv = 1.0 / prow[j];
                    const bool me = lane == pi;
                    const double f = me ? 0.0 : row[j] * ipv;
#pragma unroll
                    for (int c2 = 0; c2 < 10; ++c2)
                        row[c2] = c2 == j ? (me ? ipv : -f) : (me ? prow[c2] * ipv : fma(-f, prow[c2], row[c2]));
                    used = used || me;
                    mycol = me ? j : mycol;
                    pcol[j] = pi;
                }
            }
            if (lane < m) {
#pragma unroll
                for (int c2 = 0; c2 < 10; ++c2)
                    if (c2 < m) aug[mycol * 20 + m + pcol[c2]] = row[c2];
            }
            if (slback) {                                                // T^-1 [m][m] and W_k^-1 at the head of the (unused) inverse slot
                double* tk = Zall + ((size_t)s * n + k) * CT;
                if (lane < m) {
#pragma unroll
                    for (int c2 = 0; c2 < 10; ++c2)
                        if (c2 < m) tk[mycol * 10 + pcol[c2]] = row[c2];
                }
                if (lane < 4) tk[100 + lane] = aug[10 * 20 + lane];
            }
        }
        if (slback && tid < Hn * 4) lfS[(so + (size_t)k * Hn) * 4 + tid] = tab[tid];
        __syncthreads();
        // [0 0; 0 Ahh^-1] + Pb (T^-1 Qb) on the matrix cores: per chunk of four border unknowns one rank-4 MFMA per tile
        // (A operand = rows of Pb, B operand = the own column of T^-1 Qb: 10 FMAs per chunk and lane).  The image is fetched
        // here and not earlier: held across the inversion it would push the kernel over its register budget.
        {
            double mv[NT * 4];
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) mv[e] = 0.0;
            TileIO<B>::load(Minv + (size_t)(lzB.z - 1) * CT, wv, lg, jj, mv);
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) ct[e >> 2][e & 3] = mv[e];
        }
        {
            double qb[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) qb[j] = (j < m && col < b) ? qbl[j * b + col] : 0.0;
            if (col < 2) {                                               // right-hand side rows 0 / 1 arrive in polar power-row form: W_k^-1
                const double* wi = aug + 10 * 20;
                qb[0] = wi[col];
                qb[1] = wi[2 + col];
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (4 * ch < m) {                                        // (workgroup-uniform)
                    const int i = 4 * ch + lg;                           // border unknown of this lane's operand element
                    const double* wr = aug + (i < m ? i : 0) * 20 + m;
                    double bop = 0.0;
#pragma unroll
                    for (int j = 0; j < 10; ++j)
                        if (j < m) bop = fma(wr[j], qb[j], bop);
                    bop = i < m ? bop : 0.0;
#pragma unroll
                    for (int tr = 0; tr < NT; ++tr) {
                        const int prow_ = 16 * tr + jj;
                        const double aop = (prow_ < b && i < m) ? pbl[prow_ * m + i] : 0.0;
                        ct[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, ct[tr], 0, 0, 0);
                    }
                }
            }
        }
        const double yc = col < B ? dgb[col * 3 + 2] : 0.0;
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (16 * tr + 4 * reg >= B) continue;
                const int row = 16 * tr + 4 * reg + lg;
                double v = row < b ? ct[tr][reg] : 0.0;
                const double pr = xor16_f64(v);                                             // the other row of the harmonic
                const double* si = tab + (row >> 1) * 4 + 2 * t;
                v = t ? fma(si[1], v, si[0] * pr) : fma(si[1], pr, si[0] * v);              // S_q^-1 from the left
                ct[tr][reg] = v;
                const double sm = row_sum16(v * yc);
                if (jj == 0) panel[0][wv * 64 + row] = sm;
            }
        __syncthreads();
        if (wv == tcB && jj == jjB) {                                                       // w = A^-1 y into column B
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (16 * tr + 4 * reg >= B) continue;
                    const int row = 16 * tr + 4 * reg + lg;
                    double acc = panel[0][row];
#pragma unroll
                    for (int w2 = 1; w2 < NT; ++w2) acc += panel[0][w2 * 64 + row];
                    ct[tr][reg] = acc;
                }
        }
        HPF_STAMP(st1);
        HPF_STAMP(st3);
    } else {
    // ---- A3. own tile column: Norton cross terms -Y_N[q,p] * (jU | E)_{p,k} picked Re/Im (assemble_row's component form:
    //      v = yi*P + yr*Q, (P,Q) depend on (t, t', p) only -> two per-lane constants) ----------------------------------------
    if (nl) {
        const int pp = p < B / 2 ? p : 0;
        const double P = tab[t * (B / 2) * 4 + pp * 4 + 2 * t1];
        const double Q = tab[t * (B / 2) * 4 + pp * 4 + 2 * t1 + 1];
        const cplx* yn = ynl + (p < Hn ? p : 0);
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = 8 * tr + 2 * reg + (lg >> 1);
                const cplx y = yn[(q < Hn ? q : 0) * Hn];
                ct[tr][reg] = y.im * P + y.re * Q;
            }
    }
    HPF_STAMP(sc);
    __syncthreads();
    // ---- A4. patch (selects only): harmonic-diagonal 2x2, right-hand side in column B, identity padding for missing
    //      unknowns / equations (slack and PV buses at harmonic position 0; rows / columns beyond b) ---------------------------
    {
        const bool cvalid = col < b && (col >= 2 || (col == 0 ? k >= 1 : k >= c));
        const bool colB = col == B, colgt = col > B;
        const int dsel = colB ? 2 : t1;
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * tr + 4 * reg + lg;
                const double ident = (row == col) ? 1.0 : 0.0;
                if (16 * tr + 4 * reg >= B) {                        // compile-time: identity rows beyond the block
                    ct[tr][reg] = ident;
                } else {
                    bool rv = row < b;
                    if (tr == 0 && reg == 0) rv = rv && (lg >= 2 || (lg == 0 ? k >= 1 : k >= c));
                    const double dv = dgb[row * 3 + dsel];
                    double v = ct[tr][reg];
                    v = ((row >> 1) == p) ? dv : v;
                    v = (rv && cvalid) ? v : ident;
                    v = colB ? dv : v;
                    v = colgt ? 0.0 : v;
                    ct[tr][reg] = v;
                }
            }
    }
    HPF_STAMP(st1);

    // ---- B. remaining dense children (fixed order), then subtract the sum -----------------------------------------------------
    for (int i = 1; i < n_den; ++i) {
        const int ch = i == 1 ? nd2.w : (i == 2 ? nd3.x : (i == 3 ? nd3.y : T.dchild[den_beg + i]));
        const double* Cc = Cs + (size_t)ch * CT;
        double tmp[NT * 4];
        TileIO<B>::load(Cc, wv, lg, jj, tmp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < NT * 4; ++e)
            if (16 * (e >> 2) + 4 * (e & 3) < B) sumc[e] += tmp[e];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (cmp_c) {                                            // compress step: A(k, v) D_v^-1 A(v, k) and A(k, v) w_v from the bus v above
        double tmp[NT * 4];
        TileIO<B>::load(T.cF + ((size_t)s * T.n_comp + cmp_ci) * 3 * CT, wv, lg, jj, tmp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < NT * 4; ++e)
            if (16 * (e >> 2) + 4 * (e & 3) < B) sumc[e] += tmp[e];
    }
#pragma unroll
    for (int e = 0; e < NT * 4; ++e)
        if (16 * (e >> 2) + 4 * (e & 3) < B) ct[e >> 2][e & 3] -= sumc[e];

    HPF_STAMP(st3);
    // ---- C. blocked Gauss-Jordan, static 4x4 pivot blocks (hpf_gj_mfma.hpp), tile columns spread over the waves ------------
    // (all B/4 block steps are always performed: rows / columns beyond b are identity-padded, their steps are no-ops in
    //  exact arithmetic, and a step count known at compile time keeps the accumulators in place -- no copies between steps)
    // Columns 0 .. 16 NB16 - 1 go in BLOCKS of 16 (round 4): the wave that owns diagonal tile T inverts it ALONE -- the same four 4 x 4 pivot
    // sub-steps, inside the tile, ordered by wave-level LDS fences instead of workgroup barriers -- and puts W16 = D^-1 and its column panel
    // A_iP into LDS (row-major 16 x 16 images, row stride 17: the A-operand reads of 16 rows hit 16 banks); after ONE barrier every wave forms
    // its new pivot rows R = W16 R and the rank-16 update A_i -= A_iP R as K = 16 MFMA chains (the owner: A_iP <- -A_iP W16).  Block
    // Gauss-Jordan with the same pivot order: the result differs from the 13-step form at rounding level (1e-17 on the test matrices of
    // tools/experiments/gj16_micro.hip), with 4 workgroup barriers instead of 13 at b = 52 (7 instead of 25 at b = 100).  The images live in
    // the assembly's arrays (Y_N, tab, dgb, cc: dead from here on -- the barrier below ends their last readers).
    constexpr int NB16 = B / 16;
    if constexpr (NB16 > 0) {
        constexpr int WS = 17, NBUF = NB16 > 1 ? 2 : 1;
        static_assert(2 * (B / 2) * (B / 2) + (B / 2) * 8 + RP * 3 + NT * RP * 3 >= NBUF * NT * 16 * WS, "block images fit the dead assembly arrays");
        double* const img = lds_.pbuf;
        double* const pcol = &panel[1][0];               // owner-private: pivot columns of the diagonal tile during its inversion
        __syncthreads();
#pragma unroll
        for (int T = 0; T < NB16; ++T) {
            double* const im = img + (size_t)(NBUF > 1 ? (T & 1) : 0) * NT * 16 * WS;
#ifdef HPF_FACTOR_STAMPS
            const long long g0_ = __builtin_amdgcn_s_memtime();
#endif
            if (wv == T) {
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    const int j0 = 4 * ss;
                    const bool incol = jj >= j0 && jj < j0 + 4;
                    if (incol) pv[0][lg * 4 + (jj - j0)] = ct[T][ss];
                    HPF_WAVE_LDS_FENCE();                 // (one wave, no workgroup barrier: the other LANES' writes before this lane's reads)
                    bool weak;
                    const double wji = inv4_cofactor_lane(pv[0], lane, piv_limit, weak);
                    if (incol) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) pcol[(lg + 4 * reg) * 4 + (jj - j0)] = ct[T][reg];
                    }
                    if (weak && lane < 16) atomicOr(pivflag + s, 1);      // static pivot order under watch (see the 4 x 4 steps below)
                    if (lane < 16) wl[0][(lane & 3) * 4 + (lane >> 2)] = wji;
                    HPF_WAVE_LDS_FENCE();
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
                    HPF_WAVE_LDS_FENCE();                 // (the next sub-step overwrites pv / pcol / wl)
                }
#pragma unroll
                for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) im[tr * 16 * WS + (4 * reg + lg) * WS + jj] = ct[tr][reg];
            }
#ifdef HPF_FACTOR_STAMPS
            const long long g1_ = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef HPF_FACTOR_STAMPS
            const long long g2_ = __builtin_amdgcn_s_memtime();
#endif
            d4_t Rn;                                      // -R (B operand of the rank-16 update), by row group
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
#ifdef HPF_FACTOR_STAMPS
            {
                asm volatile("" : "+v"(ct[0]), "+v"(ct[NT - 1]));
                const long long g3_ = __builtin_amdgcn_s_memtime();
                if (T == 0) gown += g1_ - g0_;            // wave 0: the block it owns (inversion of the diagonal tile + images)
                gwait += g2_ - g1_;
                gmf += g3_ - g2_;
            }
#endif
        }
    }
#pragma unroll
    for (int st = 4 * NB16; st < B / 4; ++st) {
        {
            const int tP = st >> 2, rg = st & 3, j0 = 4 * (st & 3), buf = st & 1;
            const bool incol = jj >= j0 && jj < j0 + 4;
#ifdef HPF_FACTOR_STAMPS
            const long long g0_ = __builtin_amdgcn_s_memtime();
#endif
            if (wv == tP) {
                // pivot block -> LDS scratch (the 16 lanes that hold it), lane-parallel cofactor inverse, W -> LDS; the pivot
                // columns (64 x 4 panel) follow while the inverse's LDS reads are in flight
                if (incol) pv[buf][lg * 4 + (jj - j0)] = ct[tP][rg];
                bool weak;
                const double wji = inv4_cofactor_lane(pv[buf], lane, piv_limit, weak);
                if (incol) {
#pragma unroll
                    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) panel[buf][(16 * tr + lg + 4 * reg) * 4 + (jj - j0)] = ct[tr][reg];
                }
                // static pivot order under watch: a pivot block whose inverse amplifies by more than piv_limit (or is singular)
                // marks the scenario; hpf_solve repeats marked scenarios with partial pivoting (hpf.h, hpf_stat.flags bit 3)
                if (weak && lane < 16) atomicOr(pivflag + s, 1);
                if (lane < 16) wl[buf][(lane & 3) * 4 + (lane >> 2)] = wji;        // lane (i, j) holds W[j][i]
            }
#ifdef HPF_FACTOR_STAMPS
            const long long g1_ = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef HPF_FACTOR_STAMPS
            const long long g2_ = __builtin_amdgcn_s_memtime();
#endif
            double aop[NT];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                const double v = panel[buf][(16 * tr + jj) * 4 + lg];
                aop[tr] = (tr == tP && incol) ? 0.0 : -v;            // pivot rows are not updated by the rank-4 MFMA
            }
            const double aw = jj < 4 ? wl[buf][jj * 4 + lg] : 0.0;    // W padded to 16 x 4
            const d4_t z = {0.0, 0.0, 0.0, 0.0};
            const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, ct[tP][rg], z, 0, 0, 0);
            double rfin = d[0];
            if (wv == tP && incol) {
                rfin = wl[buf][lg * 4 + (jj - j0)];                   // W itself in the pivot block; zeroed pivot columns
#pragma unroll                                                        // make the update produce -A_iP W
                for (int tr = 0; tr < NT; ++tr) ct[tr] = d4_t{0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) ct[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[tr], rfin, ct[tr], 0, 0, 0);
            ct[tP][rg] = rfin;
#ifdef HPF_FACTOR_STAMPS
            {
                // wave 0: owner work (steps it owns), barrier wait and post-barrier section (read panel, MFMAs; the next stamp
                // waits for the accumulators only where the next step's owner reads them)
                const long long g3_ = __builtin_amdgcn_s_memtime();
                if (tP == 0) gown += g1_ - g0_;
                gwait += g2_ - g1_;
                gmf += g3_ - g2_;
            }
#endif
        }
    }

    }
    HPF_STAMP(st4);
    // ---- D. inverse (tile layout) and w = A^-1 y ------------------------------------------------------------------------
    {
        double* Zk = Zall + ((size_t)s * n + k) * CT;
        if (!cleaf && !slback) {
            double zv[NT * 4];
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) zv[e] = ct[e >> 2][e & 3];
            TileIO<B>::store(Zk, wv, lg, jj, zv);
        }
        if (wv == tcB && jj == jjB) {
            double* wk = wall + ((size_t)s * n + k) * B;
#pragma unroll
            for (int tr = 0; tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * tr + lg + 4 * reg;
                    if (row < B) wk[row] = ct[tr][reg];
                }
        }
    }

    HPF_STAMP(st5);
    // ---- E. push: Schur complement of this bus for its parent, C = G A^-1 H and G w in column B (schur_tiles on the own
    //      tile column; gl / hl were staged in A1) ---------------------------------------------------------------------------
    // element-wise form of  sign * Gs A^-1 Hs  for harmonic-diagonal Gs, Hs (LDS, 2x2 per harmonic); column B: Gs w if rhs, else 0
    auto schur_tiles = [&](const double* gsrc, const double* hsrc, const double sign, const bool rhs, double (&cv)[NT * 4]) {
        const int ti = lg & 1, tcn = jj & 1;
        double ha = hsrc[p * 4 + 2 * tcn + tcn];            // H[tcn][tcn]
        double hb = hsrc[p * 4 + 2 * (tcn ^ 1) + tcn];      // H[tcn^1][tcn]
        if (col == B) {                                     // right-hand-side column: G w
            ha = rhs ? 1.0 : 0.0;
            hb = 0.0;
        }
        if (col > B) {
            ha = 0.0;
            hb = 0.0;
        }
        ha *= sign;
        hb *= sign;
#pragma unroll
        for (int tr = 0; tr < NT; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = 8 * tr + 2 * reg + (lg >> 1);
                const double ga = gsrc[q * 4 + 2 * ti + ti];          // G[ti][ti]
                const double gb = gsrc[q * 4 + 2 * ti + (ti ^ 1)];    // G[ti][ti^1]
                const double own = ct[tr][reg];
                const double rowp = xor16_f64(own);
                const double colp = xor1_f64(own);
                const double both = xor1_f64(rowp);
                cv[tr * 4 + reg] = fma(gb, fma(both, hb, rowp * ha), ga * fma(colp, hb, own * ha));
            }
    };
    if (B > 52) {
        // large blocks: the Schur complement goes out pair of row groups by pair of row groups (nothing of it is held: with the whole image of
        // the push in registers next to the inverse's the leaf-only kernel needs 216 registers = ONE workgroup per CU)
        if (par >= 0 && !cmp_c) {
            double* Ck = Call + ((size_t)s * n + k) * CT;
            const int ti = lg & 1, tcn = jj & 1;
            double ha = hl[p * 4 + 2 * tcn + tcn], hb = hl[p * 4 + 2 * (tcn ^ 1) + tcn];
            if (col == B) {
                ha = 1.0;
                hb = 0.0;
            }
            if (col > B) {
                ha = 0.0;
                hb = 0.0;
            }
            const bool st_ok = TileIO<B>::ok(wv, jj);
            auto one = [&](const int e) {
                const int tr = e >> 2, reg = e & 3;
                const int q = 8 * tr + 2 * reg + (lg >> 1);
                const double ga = gl[q * 4 + 2 * ti + ti], gb = gl[q * 4 + 2 * ti + (ti ^ 1)];
                const double own = ct[tr][reg];
                const double rowp = xor16_f64(own);
                const double colp = xor1_f64(own);
                const double both = xor1_f64(rowp);
                return fma(gb, fma(both, hb, rowp * ha), ga * fma(colp, hb, own * ha));
            };
#pragma unroll
            for (int pp = 0; pp < TileIO<B>::NP; ++pp) {
                const double a = one(2 * pp), b2 = one(2 * pp + 1);
                if (st_ok) *reinterpret_cast<double2*>(Ck + TileIO<B>::off2(pp, wv, lg, jj)) = double2{a, b2};
            }
            const double z = one(TileIO<B>::NE - 1);
            if (st_ok) Ck[TileIO<B>::off1(wv, lg, jj)] = z;
        }
    } else if (par >= 0 && !cmp_c && (!lazy_leaf || wv == tcB)) {            // (lazy leaf: only the right-hand-side column G w is needed)
        double* Ck = Call + ((size_t)s * n + k) * CT;
        double cv[NT * 4];
        schur_tiles(gl, hl, 1.0, true, cv);
        if (!lazy_leaf) {
            TileIO<B>::store(Ck, wv, lg, jj, cv);
        } else if (col == B) {                                   // lazy leaf: only G w, by rows, at the head of the slot
#pragma unroll
            for (int e = 0; e < NT * 4; ++e)
                if (16 * (e >> 2) + 4 * (e & 3) < B) Ck[16 * (e >> 2) + 4 * (e & 3) + lg] = cv[e];
        }
    }
    if (!LEAF && cmp_v) {
        // compress step, the three more pushes of a bus eliminated before its pending child c:  A(c,k) A^-1 A(k,c) (+ A(c,k) w) for c's
        // diagonal block and right-hand side, and the dense fill  Gd = -A(p,k) A^-1 A(k,c) = A'(p,c),  Hd = -A(c,k) A^-1 A(k,p) = A'(c,p)
        // -- Gd in MFMA A-operand order [block step][row tile][lane] (c multiplies it from the left), Hd as a tile image.  One row
        // group at a time (partner values formed once, three results, stored at once: nothing is held across row groups).
        double* Fk = T.cF + ((size_t)s * T.n_comp + cmp_ci) * 3 * CT;
        const int ti = lg & 1, tcn = jj & 1;
        const bool mcol = col < B;
        const double ha = mcol ? hl[p * 4 + 2 * tcn + tcn] : 0.0, hb = mcol ? hl[p * 4 + 2 * (tcn ^ 1) + tcn] : 0.0;
        double ha2 = mcol ? lds_.hl2[p * 4 + 2 * tcn + tcn] : 0.0, hb2 = mcol ? lds_.hl2[p * 4 + 2 * (tcn ^ 1) + tcn] : 0.0;
        if (col == B) ha2 = 1.0;                            // right-hand-side column of c's Schur complement: A(c,k) w
        const bool st_ok = TileIO<B>::ok(wv, jj);
        double* Ga = Fk + CT + ((size_t)(col >> 2) * NT) * 64 + (col & 3) * 16 + lg;
#pragma unroll
        for (int e = 0; e < TileIO<B>::NE; ++e) {
            const int tr = e >> 2, reg = e & 3;
            const int q = 8 * tr + 2 * reg + (lg >> 1);
            const double ga = gl[q * 4 + 2 * ti + ti], gb = gl[q * 4 + 2 * ti + (ti ^ 1)];
            const double ga2 = lds_.gl2[q * 4 + 2 * ti + ti], gb2 = lds_.gl2[q * 4 + 2 * ti + (ti ^ 1)];
            const double own = ct[tr][reg];
            const double rowp = xor16_f64(own);
            const double colp = xor1_f64(own);
            const double both = xor1_f64(rowp);
            const double uh = fma(colp, hb, own * ha), vh = fma(both, hb, rowp * ha);          // (A^-1 H)[own row | partner row]
            const double uh2 = fma(colp, hb2, own * ha2), vh2 = fma(both, hb2, rowp * ha2);    // (A^-1 H2)
            const double fcc = fma(gb2, vh2, ga2 * uh2);
            const double hd = -fma(gb2, vh, ga2 * uh);
            const double gd = -fma(gb, vh2, ga * uh2);
            if (st_ok) {
                const size_t o = TileIO<B>::off(tr, reg, wv, lg, jj);
                Fk[o] = fcc;
                Fk[2 * CT + o] = hd;
            }
            if (mcol) Ga[tr * 64 + 4 * reg] = gd;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (mcol) {                                          // (rows B.. of the A operands: zero)
#pragma unroll
            for (int e = TileIO<B>::NE; e < NT * 4; ++e) Ga[(e >> 2) * 64 + 4 * (e & 3)] = 0.0;
        }
    }
    if (!LEAF && cmp_c && par >= 0) {
        // dense push of a bus whose coupling with its parent is the dense pair (Gd, Hd):  C = Gd A^-1 Hd,  column B = Gd w.
        //   P = Gd [A^-1 | w]:  the own tile column of A^-1 is the B operand as it stands (row group 4 st of the accumulators), the A
        //   operands come from v's image;  C = P Hd:  P goes through LDS (A operand: rows of P), Hd's tile image is the B operand.
        const double* Fk = T.cF + ((size_t)s * T.n_comp + cmp_ci) * 3 * CT;
        const double* Ga = Fk + CT + lane;
        d4_t pt[NT];
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) pt[tr] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < B / 4; ++st) {
            double aop[NT];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) aop[tr] = Ga[((size_t)st * NT + tr) * 64];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) pt[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[tr], ct[st >> 2][st & 3], pt[tr], 0, 0, 0);
        }
        double hd[NT * 4];
        TileIO<B>::load(Fk + 2 * CT, wv, lg, jj, hd);
        if (col >= B) {
#pragma unroll
            for (int e = 0; e < NT * 4; ++e) hd[e] = 0.0;
        }
        constexpr int PS = B + 1;                            // row stride of P in LDS
        constexpr int TRP = NT < 4 ? NT : 4;                 // row tiles of P per pass (B = 100: two passes)
        double* const pl = lds_.pbuf;
        double cv[NT * 4];
#pragma unroll
        for (int tr0 = 0; tr0 < NT; tr0 += TRP) {
            __syncthreads();                                 // (the Gauss-Jordan panels / the previous pass are done with)
#pragma unroll
            for (int tr = tr0; tr < tr0 + TRP && tr < NT; ++tr)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (16 * tr + 4 * reg < B && col < B) pl[(16 * (tr - tr0) + 4 * reg + lg) * PS + col] = pt[tr][reg];
            __syncthreads();
#pragma unroll
            for (int tr = tr0; tr < tr0 + TRP && tr < NT; ++tr) {
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
                const int prow = 16 * (tr - tr0) + jj;
#pragma unroll
                for (int st = 0; st < B / 4; ++st) {
                    const double a = (16 * tr + jj < B) ? pl[prow * PS + 4 * st + lg] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, hd[st], acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) cv[tr * 4 + reg] = col == B ? pt[tr][reg] : acc[reg];
            }
        }
        TileIO<B>::store(Call + ((size_t)s * n + k) * CT, wv, lg, jj, cv);
    }
    if (tstamp) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(tstamp + 1, (unsigned long long)wall_clock64());
    }
#ifdef HPF_FACTOR_STAMPS
    if ((ablate & 16) && tid == 0 && dbg) {
        st6 = __builtin_amdgcn_s_memtime();
        long long* o = dbg + ((size_t)s * n + k) * 8;
        o[0] = st1 - st0;   // assembly (+ linear children)
        // packed sub-phases of the assembly (16 bits each, units of 16 cycles): roles up to barrier 1 | reduction + Y_N
        o[1] = (((sa - st0) >> 4) & 0xffff) | ((((sc - sa) >> 4) & 0xffff) << 16) | (1ll << 62);
        // wave 0 before barrier 1: node record | base diagonal | linear children of wave 0 | wait at the barrier
        o[4] = (((sd1 - st0) >> 4) & 0xffff) | ((((sd2 - sd1) >> 4) & 0xffff) << 16) | ((((sd3 - sd2) >> 4) & 0xffff) << 32) |
               ((((sa - sd3) >> 4) & 0xffff) << 48);
        o[2] = ((gown >> 4) & 0xfffff) | (((gwait >> 4) & 0xfffff) << 20) | (((gmf >> 4) & 0xfffff) << 40);   // GJ split
        o[3] = st4 - st3;   // MFMA Gauss-Jordan
        o[5] = st6 - st5;   // Schur push
        o[6] = n_den + (sleaf ? 100 : 0);
        o[7] = (k >= M.m) | ((lin_end - lin_beg) << 1) | ((((se1 - sd2) >> 4) & 0xffff) << 16) | ((((se2 ? se2 - se1 : 0) >> 4) & 0xffff) << 32) |
               ((long long)(lzC.x >= 0) << 48) | ((long long)(lzC.y >= 0) << 49) | ((long long)lazy << 50);
    }
#endif
}

// LEAF: every bus of the launch is a constant-inverse leaf (elimination level 0 of the contracted tree): the general path
// (assembly, child sums, Gauss-Jordan) is compiled out and with it most of the register budget -> more workgroups per CU.
template <int B, bool LEAF>
__global__ __launch_bounds__(64 * ((B + 16) / 16), (B > 52 ? (LEAF ? HPF_Q100L_OCC : HPF_Q100_OCC) : (B > 28 ? (LEAF ? 6 : HPF_Q_OCC) : 5))) void k_factor_q(
    Model M, TreeDev T, const int* __restrict__ nodes, int b, int N, int Nc, const int* __restrict__ active,
    const cplx* __restrict__ Uall, const cplx* __restrict__ Eall, const double* __restrict__ fall, double* __restrict__ Zall,
    double* __restrict__ wall, const double* __restrict__ linAall, double* __restrict__ Call, double* __restrict__ Hall,
    const cplx* __restrict__ I0all, const double* __restrict__ chG, const double* __restrict__ chH,
    const double* __restrict__ chD, const double* __restrict__ chy, const double* __restrict__ Minv,
    double* __restrict__ lfK, double* __restrict__ lfS, long long* __restrict__ dbg, int ablate, int s0,
    int* __restrict__ pivflag, double piv_limit, unsigned long long* __restrict__ tstamp, int leaf_count, unsigned leaf_S) {
    constexpr int NT = (B + 16) / 16, RP = 16 * NT > 64 ? 16 * NT : 64;
    FqLds<B> lds;
    if constexpr (LEAF) {                    // (the arrays a leaf-only launch never touches cost nothing as separate objects)
        __shared__ cplx ynl[(B / 2) * (B / 2)];
        __shared__ double tab[(B / 2) * 8], dgb[RP * 3], cc[NT][RP * 3], panel[2][NT * 64], wl[2][16], pv[2][16], gl[NT * 32], hl[NT * 32];
        __shared__ double slb[B <= 52 ? 2 * B * 10 + 200 + 4 : 1];
        lds.ynl = ynl; lds.tab = tab; lds.dgb = dgb; lds.cc = cc; lds.panel = panel; lds.wl = wl; lds.pv = pv; lds.gl = gl; lds.hl = hl; lds.slb = slb;
        lds.gl2 = lds.hl2 = lds.pbuf = nullptr;
    } else {
        __shared__ __attribute__((aligned(16))) double smem[factor_q_lds<B>()];
        lds = FqLds<B>::carve(smem);
    }
    int bx = blockIdx.x, by = blockIdx.y;
#if HPF_LEAF_XCD
    if constexpr (LEAF) {
        // Leaf-only launches (every workgroup does the same work: no imbalance) as a 1-D grid of 8 ceil(count / 8) S workgroups: id -> (leaf, scenario)
        // with the leaf's low three bits fastest, then the scenario, then the leaf's high bits.  The hardware deals consecutive ids round-robin over
        // the 8 XCDs, so ALL scenarios of leaf L run on XCD L mod 8, one after the other: the leaf's per-model image (23 KB at b = 52, 80 KB at
        // b = 100 -- as much as the Schur complement the workgroup writes) is fetched into that L2 once and hit by the other scenarios.
        const unsigned id = blockIdx.x, t = id >> 3, S_ = gridDim.y == 1 ? leaf_S : 1u;
        by = (int)(t % S_);
        bx = (int)((t / S_) * 8u + (id & 7u));
        if (bx >= leaf_count) return;
    }
#endif
    factor_q_body<B, LEAF>(lds, bx, by, M, T, nodes, b, N, Nc, active, Uall, Eall, fall, Zall, wall, linAall, Call, Hall, I0all, chG,
                           chH, chD, chy, Minv, lfK, lfS, dbg, ablate, s0, pivflag, piv_limit, tstamp);
}

// root -> leaves: x_k = w_k - D_k^{-1} (A(k,parent) x_parent), inverse in tile layout; wave wv multiplies its tile column,
// the partial row sums meet in LDS.  Constant-inverse leaves keep no inverse in HBM:  D^-1 t = S^-1 (M t - Mc K (Mr t))  with
// the per-model M (tile layout, L2 / Infinity Cache), the Woodbury core K and S^-1 left by the factor kernel.
template <int B>
__device__ __forceinline__ void back_q_body(
    const int bx_, const int by_, const Model& M, const TreeDev& T, const int* __restrict__ nodes, int b, int N, int Nc, const int* __restrict__ active,
    const double* __restrict__ Zall, const double* __restrict__ wall, double* __restrict__ xall, double* __restrict__ step,
    const double* __restrict__ Hall, const double* __restrict__ Minv, const double* __restrict__ lfK,
    const double* __restrict__ lfS, int s0) {
    constexpr int NT = (B + 16) / 16;
    constexpr size_t CT = (size_t)NT * NT * 256;
    const int s = active ? active[by_ + s0] : (int)by_ + s0;   // slot -> scenario (active list; -1: frozen / empty slot)
    if (s < 0) return;
    const int4 kp = reinterpret_cast<const int4*>(nodes)[bx_];          // Tree::d_bdesc: (bus, parent, leaf slot + 1, 0)
    const int k = kp.x, par = kp.y, cleaf = kp.z;
    // compress steps (tree_build_into): role 1 = bus eliminated before its pending child c: x_k = w_k - D_k^-1 (A(k,p) x_p + A(k,c) x_c), it
    // comes AFTER c; role 2 = such a child: its coupling block with the parent p is the dense Hd = A'(k,p) the compress step left
    const int crole = kp.w >> 28, cinfo = kp.w & 0x0fffffff;
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, jj = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = M.n, c = M.c, Hn = M.Hn;
    double* xs = xall + (size_t)s * n * B;
    constexpr int RP = 16 * NT > 64 ? 16 * NT : 64;
    __shared__ double part[NT][RP];
    __shared__ double mcl[RP * 2], zl[RP];
    double x = 0.0;
    if (tid < B) x = wall[((size_t)s * n + k) * B + tid];
    if (par >= 0) {
        double zr[NT * 4];
        if (cleaf) {                                              // per-model image, same tile-image layout
            TileIO<B>::load(Minv + (size_t)(cleaf - 1) * CT, wv, lg, jj, zr);
        } else {
            TileIO<B>::load(Zall + ((size_t)s * n + k) * CT, wv, lg, jj, zr);
        }
        const int col = 16 * wv + jj, p = col >> 1, t1 = col & 1;
        double tv = 0.0;
        if (crole == 2) {                                         // t = Hd x_p: a dense product, row sums like the inverse's below
            double hr[NT * 4];
            TileIO<B>::load(T.cF + (((size_t)s * T.n_comp + cinfo) * 3 + 2) * CT, wv, lg, jj, hr);
            const double xpc = col < b ? xs[(size_t)par * B + col] : 0.0;
#pragma unroll
            for (int e = 0; e < NT * 4; ++e)
                if (16 * (e >> 2) + 4 * (e & 3) < B) {
                    const double r = row_sum16(hr[e] * xpc);
                    if (jj == 0) part[wv][16 * (e >> 2) + lg + 4 * (e & 3)] = r;
                }
            __syncthreads();
            if (tid < B) {
                double acc = part[0][tid];
#pragma unroll
                for (int w2 = 1; w2 < NT; ++w2) acc += part[w2][tid];
                zl[tid] = acc;
            }
            __syncthreads();
            tv = col < b ? zl[col] : 0.0;
            __syncthreads();                                      // (part / zl are reused below)
        } else if (col < b) {
            const double* hk = Hall + ((size_t)s * n + k) * Hn * 4 + p * 4 + t1 * 2;
            const double* xp = xs + (size_t)par * B;
            tv = fma(hk[1], xp[2 * p + 1], hk[0] * xp[2 * p]);
            if (crole == 1) {
                const double* h2 = T.cH2 + ((size_t)s * T.n_comp + cinfo) * Hn * 4 + p * 4 + t1 * 2;
                const double* xc = xs + (size_t)T.comp_child[cinfo] * B;
                tv += fma(h2[1], xc[2 * p + 1], h2[0] * xc[2 * p]);
            }
        }
        if (cleaf && tid < 2) zl[tid] = tv;                        // t[0], t[1]
        const double tvs = (cleaf && col < 2) ? 0.0 : tv;          // leaf image: columns 0, 1 hold the border, not the inverse
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < NT * 4; ++e)
            if (16 * (e >> 2) + 4 * (e & 3) < B) {
                const double r = row_sum16(zr[e] * tvs);
                if (jj == 0) part[wv][16 * (e >> 2) + lg + 4 * (e & 3)] = r;
                if (cleaf && wv == 0 && jj < 2) mcl[(16 * (e >> 2) + lg + 4 * (e & 3)) * 2 + jj] = zr[e];      // M[:, 0:2]
            }
        __syncthreads();
        double acc = 0.0;
        if (tid < B) {
            acc = part[0][tid];
#pragma unroll
            for (int w2 = 1; w2 < NT; ++w2) acc += part[w2][tid];
        }
        if (cleaf) {                                              // block-uniform
            // acc = S[row] = sum over columns >= 2 of image[row][col] t[col]:  rows 0, 1 -> (Lr t_h), rows >= 2 -> (Ahh^-1 t_h)
            double t0 = 0.0, t1v = 0.0;
            if (tid < B) {
                t0 = zl[0];
                t1v = zl[1];
            }
            __syncthreads();
            if (tid < 2) zl[tid] = acc + (tid == 0 ? t0 : t1v);   // [I Lr] t
            __syncthreads();
            if (tid < B) {
                const double* kk = lfK + ((size_t)s * n + k) * 12;
                const double u0 = fma(kk[1], zl[1], kk[0] * zl[0]), u1 = fma(kk[3], zl[1], kk[2] * zl[0]);     // (c0 + D)^-1 [I Lr] t
                const double c0r = tid < 2 ? (tid == 0 ? 1.0 : 0.0) : mcl[tid * 2];
                const double c1r = tid < 2 ? (tid == 1 ? 1.0 : 0.0) : mcl[tid * 2 + 1];
                acc = fma(c1r, u1, fma(c0r, u0, tid < 2 ? 0.0 : acc));
            }
            __syncthreads();
            if (tid < B) zl[tid] = acc;                           // Drect^-1 t
            __syncthreads();
            if (tid < B) {
                const double* si = lfS + (((size_t)s * n + k) * Hn + (tid >> 1)) * 4 + 2 * (tid & 1);
                acc = (tid >> 1) < Hn ? fma(si[1], zl[tid | 1], si[0] * zl[tid & ~1]) : acc;                    // S_q^-1 from the left
            }
        }
        if (tid < B) x -= acc;
    }
    if (tid < B) {
        xs[(size_t)k * B + tid] = x;
        if (step && tid < b) {                 // (nullptr: the update kernel reads the bus-major image xs instead)
            const int kst = (tid >> 1) * n + k;
            double* st = step + (size_t)s * N;
            if (tid & 1) {
                if (kst >= c) st[Nc + kst - c] = x;
            } else {
                if (kst >= 1) st[kst - 1] = x;
            }
        }
    }
}

template <int B>
__global__ __launch_bounds__(64 * ((B + 16) / 16)) void k_back_q(
    Model M, TreeDev T, const int* __restrict__ nodes, int b, int N, int Nc, const int* __restrict__ active,
    const double* __restrict__ Zall, const double* __restrict__ wall, double* __restrict__ xall, double* __restrict__ step,
    const double* __restrict__ Hall, const double* __restrict__ Minv, const double* __restrict__ lfK,
    const double* __restrict__ lfS, int s0) {
    back_q_body<B>(blockIdx.x, blockIdx.y, M, T, nodes, b, N, Nc, active, Zall, wall, xall, step, Hall, Minv, lfK, lfS, s0);
}

template <int B, bool LEAF>
int launch_factor_q2(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    ScopedTimer t(h, LEAF ? T_SOLVE : T_GJ);            // T_GJ: the general kernel k_factor_q<B, false> alone (roofline numerator: hpf_kernel_model)
    unsigned long long* ts = nullptr;                   // device-clock stamps of this launch (timing leg, general kernel only)
    if (!LEAF && h->timing && h->d_tstamp && h->ts_next < hpf_handle::TS_CAP) ts = h->d_tstamp + 2 * (size_t)(h->ts_next++);
    constexpr int NT = (B + 16) / 16;
    dim3 grid((unsigned)count, (unsigned)h->cur_S);
#if HPF_LEAF_XCD
    if (LEAF) grid = dim3((unsigned)(((count + 7) / 8) * 8) * (unsigned)h->cur_S, 1, 1);        // (leaf-major id decode in the kernel)
#endif
    hipLaunchKernelGGL((k_factor_q<B, LEAF>), grid, dim3(64 * NT), 0, h->cur_stream, h->M, T, nodes,
                       2 * h->Hn, h->N, h->Nc, active, h->d_U, h->d_E, h->d_fb, h->d_Z, h->d_w, h->d_linA, h->d_C, h->d_H,
                       h->d_I0, h->d_chG, h->d_chH, h->d_chD, h->d_chy, active_tree(h).d_Minv, h->d_lfK, h->d_lfS, h->d_dbg, h->debug_ablate, h->cur_s0,
                       h->d_pivflag, h->piv_limit, ts, count, (unsigned)h->cur_S);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

template <int B>
int launch_factor_q(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active, bool all_leaves) {
    return all_leaves ? launch_factor_q2<B, true>(h, T, nodes, count, active) : launch_factor_q2<B, false>(h, T, nodes, count, active);
}

template <int B>
int launch_back_q(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    constexpr int NT = (B + 16) / 16;
    hipLaunchKernelGGL((k_back_q<B>), dim3((unsigned)count, (unsigned)h->cur_S), dim3(64 * NT), 0, h->cur_stream, h->M, T, nodes,
                       2 * h->Hn, h->N, h->Nc, active, h->d_Z, h->d_w, h->d_x, (double*)nullptr, h->d_H, active_tree(h).d_Minv, h->d_lfK,
                       h->d_lfS, h->cur_s0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}
