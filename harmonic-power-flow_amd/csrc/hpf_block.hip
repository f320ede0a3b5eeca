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
#include "hpf_internal.hpp"

using namespace hpf;

namespace {

struct TreeDev {
    const int* parent;
    const int* child_ptr;
    const int* child;
    const int* e_up;
    const int* e_dn;
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
                                                     double* __restrict__ wall) {
    const int s = blockIdx.y;
    if (active && !active[s]) return;
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
    const int ldr = b | 1;
    double* Rm = lds;                        // [b][ldr]
    double* colbuf = Rm + (size_t)b * ldr;   // [b]
    double* rowr = colbuf + b;               // [b]
    double* rowj = rowr + b;                 // [b]
    double* ybuf = rowj + b;                 // [b]
    int* piv = (int*)(ybuf + b);             // [b]
    int* pinv = piv + b;                     // [b]
    int* pfwd = pinv + b;                    // [b]

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

    // ---- C. in-place Gauss-Jordan inversion with partial pivoting ---------------------------------------------
    for (int j = 0; j < b; ++j) {
        const int jr = j >> 4, jc = j & 15;       // owner thread-row index (tr == jc... see below)
        // column j -> LDS
        if (tc == (j & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai) {
                const int i = tr + 16 * ai;
                if (i < b) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci)
                        if (ci == (j >> 4)) colbuf[i] = a[ai][ci];
                }
            }
        }
        __syncthreads();
        // pivot row: largest |.| among rows >= j (lowest index wins ties), computed redundantly by every thread
        int r = j;
        double best = fabs(colbuf[j]);
        for (int i = j + 1; i < b; ++i) {
            const double v = fabs(colbuf[i]);
            if (v > best) {
                best = v;
                r = i;
            }
        }
        // rows r and j -> LDS
        if (tr == (r & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai)
                if (ai == (r >> 4)) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) {
                        const int cc = tc + 16 * ci;
                        if (cc < b) rowr[cc] = a[ai][ci];
                    }
                }
        }
        if (r != j && tr == (j & 15)) {
#pragma unroll
            for (int ai = 0; ai < R; ++ai)
                if (ai == (j >> 4)) {
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) {
                        const int cc = tc + 16 * ci;
                        if (cc < b) rowj[cc] = a[ai][ci];
                    }
                }
        }
        if (tid == 0) piv[j] = r;
        __syncthreads();
        const double inv = 1.0 / colbuf[r];
        // scaled pivot row values of my columns (column j itself becomes 1/pivot)
        double prow[R];
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            prow[ci] = cc < b ? (cc == j ? inv : rowr[cc] * inv) : 0.0;
        }
#pragma unroll
        for (int ai = 0; ai < R; ++ai) {
            const int i = tr + 16 * ai;
            if (i >= b) continue;
            if (i == j) {
#pragma unroll
                for (int ci = 0; ci < R; ++ci) a[ai][ci] = prow[ci];
            } else {
                // after the swap row r holds the old row j
                const double fct = (i == r) ? colbuf[j] : colbuf[i];
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    if (cc >= b) continue;
                    const double base = (i == r) ? rowj[cc] : a[ai][ci];
                    a[ai][ci] = (cc == j) ? -fct * inv : fma(-fct, prow[ci], base);
                }
            }
        }
        (void)jr;
        (void)jc;
        // the LDS buffers are rewritten only after the next barrier pair, but colbuf is rewritten first:
        __syncthreads();
    }

    // ---- D. R = (P D)^{-1} -> LDS; permutation pi with (P x)[k] = x[pi[k]] ------------------------------------
#pragma unroll
    for (int ai = 0; ai < R; ++ai) {
        const int i = tr + 16 * ai;
        if (i >= b) continue;
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            if (cc < b) Rm[(size_t)i * ldr + cc] = a[ai][ci];
        }
    }
    if (tid == 0) {
        for (int i = 0; i < b; ++i) pfwd[i] = i;
        for (int j = 0; j < b; ++j) {
            const int r = piv[j];
            const int tmp = pfwd[j];
            pfwd[j] = pfwd[r];
            pfwd[r] = tmp;
        }
        for (int i = 0; i < b; ++i) pinv[pfwd[i]] = i;
    }
    __syncthreads();

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
                                                   double* __restrict__ xall, double* __restrict__ step) {
    const int s = blockIdx.y;
    if (active && !active[s]) return;
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

size_t factor_lds_bytes(int b) {
    const int ldr = b | 1;
    return sizeof(double) * ((size_t)b * ldr + 4 * (size_t)b) + sizeof(int) * 3 * (size_t)b;
}

template <int R>
int launch_factor(hpf_handle* h, const TreeDev& T, const int* nodes, int count, const int* active) {
    const int b = 2 * h->Hn;
    const size_t lds = factor_lds_bytes(b);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_factor<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_tree_factor<R>), dim3((unsigned)count, (unsigned)h->S), dim3(256), lds, h->stream, h->M, T,
                       nodes, b, h->N, h->Nc, active, h->d_U, h->d_E, h->d_f, h->d_Z, h->d_w);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        h->last_detail = (int)e;
        return HPF_E_HIP;
    }
    return HPF_OK;
}

}  // namespace

namespace hpf {

int tree_build(hpf_handle* h, const hpf_desc* d) {
    const int n = d->n;
    const int b = 2 * d->Hn;
    if (b > 16 * 7) return HPF_E_ARG;                          // register tile limit (K <= 55)
    if (d->nnz != n + 2 * (n - 1)) return HPF_E_TOPOLOGY;
    Tree& T = h->tree;
    T.parent.assign(n, -2);
    std::vector<int> order, depth(n, 0), height(n, 0), e_up(n, -1), e_dn(n, -1);
    order.reserve(n);
    order.push_back(0);
    T.parent[0] = -1;
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const int i = order[oi];
        for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e) {
            const int j = d->col[e];
            if (j == i) continue;
            if (T.parent[j] == -2) {
                T.parent[j] = i;
                depth[j] = depth[i] + 1;
                e_dn[j] = e;                                   // entry (parent, child)
                order.push_back(j);
            }
        }
    }
    if ((int)order.size() != n) return HPF_E_TOPOLOGY;
    for (int i = 1; i < n; ++i) {
        const int p = T.parent[i];
        for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e)
            if (d->col[e] == p) e_up[i] = e;
        if (e_up[i] < 0 || e_dn[i] < 0) return HPF_E_TOPOLOGY;  // pattern not symmetric
    }
    for (int oi = n - 1; oi > 0; --oi) {
        const int i = order[oi], p = T.parent[i];
        if (height[i] + 1 > height[p]) height[p] = height[i] + 1;
    }
    int maxh = 0, maxd = 0;
    for (int i = 0; i < n; ++i) {
        maxh = height[i] > maxh ? height[i] : maxh;
        maxd = depth[i] > maxd ? depth[i] : maxd;
    }
    T.n_levels = maxh + 1;
    T.n_depths = maxd + 1;
    auto bucket = [&](const std::vector<int>& key, int nb, std::vector<int>& ptr, std::vector<int>& items) {
        ptr.assign(nb + 1, 0);
        for (int i = 0; i < n; ++i) ptr[key[i] + 1]++;
        for (int l = 0; l < nb; ++l) ptr[l + 1] += ptr[l];
        items.assign(n, 0);
        std::vector<int> pos(ptr.begin(), ptr.end() - 1);
        for (int i = 0; i < n; ++i) items[pos[key[i]]++] = i;   // ascending bus index inside a level
    };
    bucket(height, T.n_levels, T.lvl_ptr, T.lvl_nodes);
    bucket(depth, T.n_depths, T.dep_ptr, T.dep_nodes);
    T.child_ptr.assign(n + 1, 0);
    for (int i = 1; i < n; ++i) T.child_ptr[T.parent[i] + 1]++;
    for (int i = 0; i < n; ++i) T.child_ptr[i + 1] += T.child_ptr[i];
    T.child.assign(n > 1 ? n - 1 : 0, 0);
    {
        std::vector<int> pos(T.child_ptr.begin(), T.child_ptr.end() - 1);
        for (int i = 1; i < n; ++i) T.child[pos[T.parent[i]]++] = i;   // children in ascending bus index
    }
    const double bd = b;
    // exact flop count of the elimination: per bus 2 b^3 (Gauss-Jordan), (4 b^2 + 4 b) per child pulled,
    // 2 b^2 (w = D^-1 y); per non-root bus 4 b^2 (Z = D^-1 A(k,parent)) and 2 b^2 in the back sweep
    T.flops_factor = 0.0;
    for (int i = 0; i < n; ++i) {
        const int nch = T.child_ptr[i + 1] - T.child_ptr[i];
        T.flops_factor += 2.0 * bd * bd * bd + (4.0 * bd * bd + 4.0 * bd) * nch + 2.0 * bd * bd;
        if (i > 0) T.flops_factor += 4.0 * bd * bd;
    }
    T.flops_per_solve = T.flops_factor + 2.0 * bd * bd * (n - 1);
    int r;
    if ((r = upload(h, &T.d_parent, T.parent))) return r;
    if ((r = upload(h, &T.d_lvl_nodes, T.lvl_nodes))) return r;
    if ((r = upload(h, &T.d_dep_nodes, T.dep_nodes))) return r;
    if ((r = upload(h, &T.d_child_ptr, T.child_ptr))) return r;
    if ((r = upload(h, &T.d_child, T.child))) return r;
    if ((r = upload(h, &T.d_e_up, e_up))) return r;
    if ((r = upload(h, &T.d_e_dn, e_dn))) return r;
    return HPF_OK;
}

void tree_free(hpf_handle* h) {
    Tree& T = h->tree;
    void* ptrs[] = {T.d_parent, T.d_lvl_nodes, T.d_dep_nodes, T.d_child_ptr, T.d_child, T.d_e_up, T.d_e_dn};
    for (void* p : ptrs)
        if (p) hipFree(p);
}

int tree_alloc_scenarios(hpf_handle* h) {
    const size_t b = 2 * (size_t)h->Hn, S = h->S_max, n = h->n;
    hipError_t e;
    if ((e = hipMalloc((void**)&h->d_Z, sizeof(double) * S * n * b * b)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_w, sizeof(double) * S * n * b)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_x, sizeof(double) * S * n * b)) != hipSuccess) {
        h->last_detail = (int)e;
        return e == hipErrorOutOfMemory ? HPF_E_NOMEM : HPF_E_HIP;
    }
    return HPF_OK;
}

int tree_newton_step(hpf_handle* h, bool only_active) {
    Tree& T = h->tree;
    const int* active = only_active ? h->d_active : nullptr;
    const TreeDev td{T.d_parent, T.d_child_ptr, T.d_child, T.d_e_up, T.d_e_dn};
    const int b = 2 * h->Hn;
    const int R = (b + 15) / 16;
    {
    ScopedTimer t(h, T_SOLVE);
    for (int l = 0; l < T.n_levels; ++l) {
        const int cnt = T.lvl_ptr[l + 1] - T.lvl_ptr[l];
        if (cnt == 0) continue;
        const int* nodes = T.d_lvl_nodes + T.lvl_ptr[l];
        int r;
        switch (R) {
            case 1: r = launch_factor<1>(h, td, nodes, cnt, active); break;
            case 2: r = launch_factor<2>(h, td, nodes, cnt, active); break;
            case 3: r = launch_factor<3>(h, td, nodes, cnt, active); break;
            case 4: r = launch_factor<4>(h, td, nodes, cnt, active); break;
            case 5: r = launch_factor<5>(h, td, nodes, cnt, active); break;
            case 6: r = launch_factor<6>(h, td, nodes, cnt, active); break;
            case 7: r = launch_factor<7>(h, td, nodes, cnt, active); break;
            default: return HPF_E_ARG;
        }
        if (r) return r;
    }
    }
    ScopedTimer tb(h, T_BACK);
    for (int dl = 0; dl < T.n_depths; ++dl) {
        const int cnt = T.dep_ptr[dl + 1] - T.dep_ptr[dl];
        if (cnt == 0) continue;
        hipLaunchKernelGGL(k_tree_back, dim3((unsigned)cnt, (unsigned)h->S), dim3(256), 0, h->stream, h->n, h->c, h->Hn, td,
                           T.d_dep_nodes + T.dep_ptr[dl], b, h->N, h->Nc, active, h->d_Z, h->d_w, h->d_x, h->d_f);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            h->last_detail = (int)e;
            return HPF_E_HIP;
        }
    }
    return HPF_OK;
}

}  // namespace hpf
