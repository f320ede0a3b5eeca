// hpf_sparse_solve — update_harmonic_state_vec (HG:476-479: x - spsolve(J, f)) for the reference's CSR Jacobian at ANY size.
//
// The reference hands the stacked real Jacobian of build_harmonic_jacobian (HG:469-472: rows [P | Re dI | Q | Im dI], columns [theta | V]) to
// SuperLU.  Its row / column numbering is a function of (n, c, Hn) alone (hpf_assembly.hpp):  real row  Re(k) = k - 1 (k >= 1),
// Im(k) = Nc + k - c (k >= c);  column theta(k) = k - 1, V(k) = Nc + k - c;  k = q n + i the stacked index, Nc = n Hn - 1.  Re-ordered bus-major
// (local index l = 2 q + t of bus i) the matrix is a block matrix on the NETWORK GRAPH with blocks of size b = 2 Hn, and on a radial feeder
// eliminating buses leaves -> root creates no fill (hpf_block.hip).  This file does that elimination for blocks GIVEN by the caller:
//   host:   one pass over the pattern -> bus adjacency, BFS tree from bus 0 (anything else: HPF_E_TOPOLOGY -- the caller falls back to the dense
//           LU where that fits), children lists, elimination levels by height, back-sweep depths;
//   k_csr_scatter: CSR entries -> dense b x b blocks D_k (diagonal), Aup_k = A(k, parent), Adn_k = A(parent, k), right-hand side y_k; identity
//           rows / columns where a bus has no equation / unknown (slack at h = 1, Q / V of PV buses at h = 1);
//   k_csr_factor (one launch per level, one 256-thread workgroup per bus): D_k -= sum_children Adn_ch Z_ch, y_k -= Adn_ch w_ch; Gauss-Jordan with
//           partial pivoting over the whole block (hpf_gj_dense.hpp); Z_k = D_k^-1 Aup_k, w_k = D_k^-1 y_k;
//   k_csr_back (one launch per depth): x_k = w_k - Z_k x_parent, written in the reference's stacked order.
// Nothing of size N x N exists anywhere: 3 b^2 + b doubles per bus (65 MB at 1 000 buses x 26 harmonics, where the dense matrix is 21.6 GB).
// Off-diagonal blocks are treated as DENSE (the reference's are harmonic-diagonal, but this entry point takes any matrix on the pattern).
//
// MESHED patterns (round 5): the bus graph = BFS spanning tree + loop-closing edges ("ties").  With J_t = J without the ties' blocks,
// J = J_t + E_T Q^T  (T = the ties' endpoint buses, E_T = the identity columns of their m = |T| b unknowns' EQUATIONS, block row s of Q^T =
// sum over ties (T[s], j) of A(T[s], j) E_j^T), and  x = y - Z g,  y = J_t^-1 f,  Z = J_t^-1 E_T,  (I + Q^T Z) g = Q^T y  (the
// bordered Newton step of the loop, here with the tree factorised ONCE):
//   k_csr_factor also keeps S_k^-1;  the border matrix only needs Z at the endpoint buses, i.e. the (T, T) block of J_t^-1 -- a selected
//   inversion over P = the union of the endpoints' root paths, done in b x b block products (k_blk_jobs):
//     Up_c = S_p^-1 A(p, c)                                  (c in P, p its parent)
//     forward, leaves -> root, only on the root path of T[t]:   W[T[t], t] = S^-1,   W[p, t] = -Up_c W[c, t]
//     back, root -> leaves, every bus of P, every t:            X[root, t] = W[root, t],   X[k, t] = W[k, t] - Z_k X[p, t]
//   k_border_G builds I + Q^T Z (column-major, m x m) and Q^T y, rocSOLVER factors it, and ONE more right-hand-side sweep with the kept factors
//   (k_csr_fwd, k_csr_back) gives x = J_t^-1 (f - E_T g).  m <= 16 384 like the loop's bordered step.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "../../include/hpf.h"
#include "hpf_blk_jobs.hpp"
#include "hpf_gj_dense.hpp"

using namespace hpf;

namespace {

// (bus, local index) of a real row / column index of the reference's stacked ordering; the same rule for rows and columns
__host__ __device__ __forceinline__ void rc_to_bus(int r, int n, int c, int Nc, int& bus, int& l) {
    const int t = r >= Nc ? 1 : 0;
    const int k = t ? r - Nc + c : r + 1;
    const int q = k / n;
    bus = k - q * n;
    l = 2 * q + t;
}
__host__ __device__ __forceinline__ bool loc_valid(int n, int c, int i, int l) {
    const int kst = (l >> 1) * n + i;
    return (l & 1) ? kst >= c : kst >= 1;
}

// identity padding of the diagonal blocks + zero right-hand side padding: one thread per (bus, l)
__global__ void k_csr_pad(int n, int c, int b, double* __restrict__ D) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * b) return;
    const int i = t / b, l = t - i * b;
    if (!loc_valid(n, c, i, l)) D[((size_t)i * b + l) * b + l] = 1.0;
}

// one thread per real row: its entries go into the row's three possible blocks (the row's bus i: diagonal block, A(i, parent), and
// A(i, child) = Adn of that child); duplicates of a (row, column) pair add up like scipy's
__global__ void k_csr_scatter(int N, int n, int c, int Nc, int b, const int* __restrict__ indptr, const int* __restrict__ indices,
                              const double* __restrict__ data, const double* __restrict__ f, const int* __restrict__ parent,
                              double* __restrict__ D, double* __restrict__ Aup, double* __restrict__ Adn, double* __restrict__ y,
                              const int* __restrict__ tie_ptr, const int* __restrict__ tie_nb, double* __restrict__ Tie) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    int i, l;
    rc_to_bus(r, n, c, Nc, i, l);
    y[(size_t)i * b + l] = f[r];
    const int par = parent[i];
    const size_t bb = (size_t)b * b;
    for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
        int j, lc;
        rc_to_bus(indices[e], n, c, Nc, j, lc);
        double* dst;
        if (j == i) {
            dst = D + (size_t)i * bb;
        } else if (j == par) {
            dst = Aup + (size_t)i * bb;
        } else if (!Tie || parent[j] == i) {            // (radial: parent[j] == i was checked on the host)
            dst = Adn + (size_t)j * bb;
        } else {                                        // a tie (i, j): block number = position of j in the tie list of i (host: it is there)
            int t = tie_ptr[i];
            while (t + 1 < tie_ptr[i + 1] && tie_nb[t] != j) ++t;
            dst = Tie + (size_t)t * bb;
        }
        dst[(size_t)l * b + lc] += data[e];
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_csr_factor(int b, const int* __restrict__ nodes, const int* __restrict__ parent,
                                                    const int* __restrict__ child_ptr, const int* __restrict__ child,
                                                    const double* __restrict__ D, const double* __restrict__ Aup,
                                                    const double* __restrict__ Adn, const double* __restrict__ y, double* __restrict__ Z,
                                                    double* __restrict__ w, int* __restrict__ singular, double* __restrict__ Sinv) {
    const int k = nodes[blockIdx.x];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    const size_t bb = (size_t)b * b;
    extern __shared__ double lds[];
    const GjDenseLds L(lds, b);
    __shared__ int zero_piv;
    if (tid == 0) zero_piv = 0;
    // ---- A. the given diagonal block -> registers, right-hand side -> LDS ------------------------------------------------------
    double a[R][R];
    const double* Dk = D + (size_t)k * bb;
#pragma unroll
    for (int ai = 0; ai < R; ++ai) {
        const int i = tr + 16 * ai;
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int cc = tc + 16 * ci;
            a[ai][ci] = (i < b && cc < b) ? Dk[(size_t)i * b + cc] : 0.0;
        }
    }
    if (tid < b) L.ybuf[tid] = y[(size_t)k * b + tid];
    __syncthreads();
    // ---- B. children (fixed order): D_k -= Adn_ch Z_ch, y_k -= Adn_ch w_ch -------------------------------------------------------
    for (int cp = child_ptr[k]; cp < child_ptr[k + 1]; ++cp) {
        const int ch = child[cp];
        const double* G = Adn + (size_t)ch * bb;       // rows of k, columns of ch
        const double* Zc = Z + (size_t)ch * bb;        // rows of ch, columns of k
        const double* wc = w + (size_t)ch * b;
        for (int l0 = 0; l0 < b; l0 += 4) {
            double g[R][4], z[4][R];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = l0 + u < b ? l0 + u : b - 1;
                const bool on = l0 + u < b;
#pragma unroll
                for (int ai = 0; ai < R; ++ai) {
                    const int i = tr + 16 * ai;
                    g[ai][u] = (on && i < b) ? G[(size_t)i * b + l] : 0.0;
                }
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    z[u][ci] = cc < b ? Zc[(size_t)l * b + cc] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ai = 0; ai < R; ++ai)
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) a[ai][ci] = fma(-g[ai][u], z[u][ci], a[ai][ci]);
        }
        if (tid < b) {
            double acc = L.ybuf[tid];
            const double* Gr = G + (size_t)tid * b;
            for (int l = 0; l < b; ++l) acc = fma(-Gr[l], wc[l], acc);
            L.ybuf[tid] = acc;
        }
    }
    __syncthreads();
    // ---- C. (P D)^-1 -> LDS --------------------------------------------------------------------------------------------------------
    gj_dense_invert<R>(a, b, L, &zero_piv);
    if (tid == 0 && zero_piv) atomicMax(singular, k * b + zero_piv);
    // ---- D. w_k = D^-1 y, Z_k = D^-1 Aup_k (meshed patterns: D^-1 itself is kept as well) -----------------------------------------------
    if (Sinv) {
        double* Sk = Sinv + (size_t)k * bb;
#pragma unroll
        for (int ai = 0; ai < R; ++ai) {
            const int i = tr + 16 * ai;
#pragma unroll
            for (int ci = 0; ci < R; ++ci) {
                const int cc = tc + 16 * ci;
                if (i < b && cc < b) Sk[(size_t)i * b + cc] = L.Rm[(size_t)i * L.ldr + L.pinv[cc]];
            }
        }
    }
    if (tid < b) {
        double acc = 0.0;
        const double* Rrow = L.Rm + (size_t)tid * L.ldr;
        for (int kk = 0; kk < b; ++kk) acc = fma(Rrow[kk], L.ybuf[L.pfwd[kk]], acc);
        w[(size_t)k * b + tid] = acc;
    }
    if (parent[k] >= 0) {
        const double* Ak = Aup + (size_t)k * bb;
        double o[R][R];
#pragma unroll
        for (int ai = 0; ai < R; ++ai)
#pragma unroll
            for (int ci = 0; ci < R; ++ci) o[ai][ci] = 0.0;
        // Z = (P D)^-1 (P Aup) = sum_r Rm[:, pinv[r]] Aup[r, :]: the rows of Aup are walked in storage order (independent, coalesced loads, four
        // rows in flight), the permutation sits on the LDS column index of Rm
        for (int r0 = 0; r0 < b; r0 += 4) {
            double av[4][R], rv[R][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u < b ? r0 + u : b - 1;
                const bool on = r0 + u < b;
                const int pc = L.pinv[r];
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    av[u][ci] = (on && cc < b) ? Ak[(size_t)r * b + cc] : 0.0;
                }
#pragma unroll
                for (int ai = 0; ai < R; ++ai) {
                    const int i = tr + 16 * ai;
                    rv[ai][u] = i < b ? L.Rm[(size_t)i * L.ldr + pc] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ai = 0; ai < R; ++ai)
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) o[ai][ci] = fma(rv[ai][u], av[u][ci], o[ai][ci]);
        }
        double* Zk = Z + (size_t)k * bb;
#pragma unroll
        for (int ai = 0; ai < R; ++ai) {
            const int i = tr + 16 * ai;
#pragma unroll
            for (int ci = 0; ci < R; ++ci) {
                const int cc = tc + 16 * ci;
                if (i < b && cc < b) Zk[(size_t)i * b + cc] = o[ai][ci];
            }
        }
    }
}

// root -> leaves: x_k = w_k - Z_k x_parent; one wavefront per row group, the result also in the reference's stacked order
__global__ __launch_bounds__(256) void k_csr_back(int n, int c, int Nc, int b, const int* __restrict__ nodes, const int* __restrict__ parent,
                                                  const double* __restrict__ Z, const double* __restrict__ w, double* __restrict__ xb,
                                                  double* __restrict__ dx) {
    const int k = nodes[blockIdx.x];
    const int par = parent[k];
    const double* Zk = Z + (size_t)k * b * b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < b; i += 4) {
        double acc = 0.0;
        if (par >= 0) {
            const double* xp = xb + (size_t)par * b;
            for (int cc = lane; cc < b; cc += 64) acc = fma(Zk[(size_t)i * b + cc], xp[cc], acc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        }
        if (lane == 0) {
            const double x = w[(size_t)k * b + i] - acc;
            xb[(size_t)k * b + i] = x;
            if (loc_valid(n, c, k, i)) {
                const int kst = (i >> 1) * n + k;
                dx[(i & 1) ? Nc + kst - c : kst - 1] = x;
            }
        }
    }
}

// ---- meshed patterns -----------------------------------------------------------------------------------------------------------------------
// border matrix I + Q^T Z, column-major m x m: workgroup (s, t) forms block (s, t) = delta_st I + sum over ties (T[s], j) of Tie(T[s] -> j) X[j, t];
// the workgroups of column t = 0 also form the right-hand side block s of Q^T y from the first solve's bus-major solution xb
template <int R>
__global__ __launch_bounds__(256) void k_border_G(int b, int mT, const int* __restrict__ Tbus, const int* __restrict__ tie_ptr,
                                                  const int* __restrict__ tie_nb, const int* __restrict__ pidx, const double* __restrict__ Tie,
                                                  const double* __restrict__ X, const double* __restrict__ xb, double* __restrict__ Gm,
                                                  double* __restrict__ gr) {
    const int s = blockIdx.x, t = blockIdx.y;
    const int i_bus = Tbus[s];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    const size_t bb = (size_t)b * b, m = (size_t)mT * b;
    double o[R][R];
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) o[ai][ci] = (s == t && tr + 16 * ai == tc + 16 * ci) ? 1.0 : 0.0;
    for (int e = tie_ptr[i_bus]; e < tie_ptr[i_bus + 1]; ++e) {
        const double* A = Tie + (size_t)e * bb;
        const double* Bm = X + ((size_t)pidx[tie_nb[e]] * mT + t) * bb;
        for (int l0 = 0; l0 < b; l0 += 4) {
            double g[R][4], z[4][R];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = l0 + u < b ? l0 + u : b - 1;
                const bool on = l0 + u < b;
#pragma unroll
                for (int ai = 0; ai < R; ++ai) {
                    const int i = tr + 16 * ai;
                    g[ai][u] = (on && i < b) ? A[(size_t)i * b + l] : 0.0;
                }
#pragma unroll
                for (int ci = 0; ci < R; ++ci) {
                    const int cc = tc + 16 * ci;
                    z[u][ci] = cc < b ? Bm[(size_t)l * b + cc] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ai = 0; ai < R; ++ai)
#pragma unroll
                    for (int ci = 0; ci < R; ++ci) o[ai][ci] = fma(g[ai][u], z[u][ci], o[ai][ci]);
        }
    }
#pragma unroll
    for (int ai = 0; ai < R; ++ai)
#pragma unroll
        for (int ci = 0; ci < R; ++ci) {
            const int i = tr + 16 * ai, cc = tc + 16 * ci;
            if (i < b && cc < b) Gm[((size_t)s * b + i) + m * ((size_t)t * b + cc)] = o[ai][ci];
        }
    if (t == 0 && tid < b) {
        double acc = 0.0;
        for (int e = tie_ptr[i_bus]; e < tie_ptr[i_bus + 1]; ++e) {
            const double* Ar = Tie + (size_t)e * bb + (size_t)tid * b;
            const double* xj = xb + (size_t)tie_nb[e] * b;
            for (int l = 0; l < b; ++l) acc = fma(Ar[l], xj[l], acc);
        }
        gr[(size_t)s * b + tid] = acc;
    }
}

// y2 = y - E_T g  (one thread per border unknown)
__global__ void k_border_rhs(int b, int m, const int* __restrict__ Tbus, const double* __restrict__ g, double* __restrict__ y2) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= m) return;
    y2[(size_t)Tbus[u / b] * b + (u % b)] -= g[u];
}

// right-hand-side sweep with the kept factors, leaves -> root: w_k = S_k^-1 (y_k - sum_children Adn_ch w_ch)
__global__ __launch_bounds__(256) void k_csr_fwd(int b, const int* __restrict__ nodes, const int* __restrict__ child_ptr,
                                                 const int* __restrict__ child, const double* __restrict__ Sinv, const double* __restrict__ Adn,
                                                 const double* __restrict__ y, double* __restrict__ w) {
    const int k = nodes[blockIdx.x];
    extern __shared__ double ybuf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t bb = (size_t)b * b;
    for (int i = wave; i < b; i += 4) {
        double acc = 0.0;
        for (int cp = child_ptr[k]; cp < child_ptr[k + 1]; ++cp) {
            const int ch = child[cp];
            const double* Gr = Adn + (size_t)ch * bb + (size_t)i * b;
            const double* wc = w + (size_t)ch * b;
            for (int l = lane; l < b; l += 64) acc = fma(Gr[l], wc[l], acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) ybuf[i] = y[(size_t)k * b + i] - acc;
    }
    __syncthreads();
    const double* Sk = Sinv + (size_t)k * bb;
    for (int i = wave; i < b; i += 4) {
        double acc = 0.0;
        for (int l = lane; l < b; l += 64) acc = fma(Sk[(size_t)i * b + l], ybuf[l], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) w[(size_t)k * b + i] = acc;
    }
}

// one device allocation for everything a call needs, carved with 256-byte alignment (one hipMalloc + one hipFree per call: allocation calls are
// the part of a call whose duration the driver does not bound)
struct DevPool {
    char* base = nullptr;
    size_t used = 0, cap = 0;
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
    template <class T>
    void carve(T** p, size_t count) {
        *p = base ? reinterpret_cast<T*>(base + used) : nullptr;      // (sizing pass: no pointer arithmetic on a null base)
        used += pad(sizeof(T) * (count ? count : 1));
    }
    ~DevPool() {
        if (base) hipFree(base);
    }
};

template <int R>
hipError_t launch_factor(int b, int count, const int* nodes, const int* parent, const int* child_ptr, const int* child, const double* D,
                         const double* Aup, const double* Adn, const double* y, double* Z, double* w, int* singular, double* Sinv, hipStream_t st) {
    const size_t lds = gj_dense_lds_bytes(b);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_factor<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_csr_factor<R>), dim3((unsigned)count), dim3(256), lds, st, b, nodes, parent, child_ptr, child, D, Aup, Adn, y, Z, w, singular, Sinv);
    return hipGetLastError();
}

template <int R>
void launch_border_R(int b, int mT, const int* Tbus, const int* tie_ptr, const int* tie_nb, const int* pidx, const double* Tie, const double* X,
                     const double* xb, double* Gm, double* gr, hipStream_t st) {
    hipLaunchKernelGGL((k_border_G<R>), dim3((unsigned)mT, (unsigned)mT), dim3(256), 0, st, b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr);
}
void launch_border(int R, int b, int mT, const int* Tbus, const int* tie_ptr, const int* tie_nb, const int* pidx, const double* Tie, const double* X,
                   const double* xb, double* Gm, double* gr, hipStream_t st) {
    switch (R) {
        case 1: launch_border_R<1>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        case 2: launch_border_R<2>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        case 3: launch_border_R<3>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        case 4: launch_border_R<4>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        case 5: launch_border_R<5>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        case 6: launch_border_R<6>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        case 7: launch_border_R<7>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
        default: launch_border_R<8>(b, mT, Tbus, tie_ptr, tie_nb, pidx, Tie, X, xb, Gm, gr, st); break;
    }
}

struct BlasHandle {                                      // (the border system only; destroyed with the call)
    rocblas_handle h = nullptr;
    ~BlasHandle() {
        if (h) rocblas_destroy_handle(h);
    }
};

}  // namespace

extern "C" int hpf_sparse_solve(int device, int n, int c, int Hn, const int32_t* indptr, const int32_t* indices, const double* data,
                                const double* f, double* dx) {
    if (n < 1 || c < 1 || c > n || Hn < 1 || !indptr || !indices || !data || !f || !dx) return HPF_E_ARG;
    if ((long long)n * Hn >= (1ll << 29)) return HPF_E_ARG;
    const bool info = getenv("HPF_SPARSE_INFO") != nullptr;              // phase times to stderr (diagnostic; no effect on the result)
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const int b = 2 * Hn;
    if (b > 128) return HPF_E_ARG;                       // (the block inversion keeps a b x b block in the LDS of one workgroup: 132 KB at b = 128)
    const int Nc = n * Hn - 1;
    const int N = 2 * Nc - (c - 1);
    if (N < 1 || indptr[0] != 0) return HPF_E_ARG;
    // ---- host: bus adjacency of the pattern, tree from bus 0 ---------------------------------------------------------------------------
    std::vector<std::vector<int>> adj(n);
    {
        std::vector<int> stamp(n, -1), bus_of(N);      // bus of every real index: one division per index instead of one per entry
        for (int r = 0; r < N; ++r) {
            int l;
            rc_to_bus(r, n, c, Nc, bus_of[r], l);
        }
        // rows of one bus are not contiguous in the stacked order: stamp[j] = i marks "edge (i, j) already listed" only while the walk stays
        // on bus i, so an edge can be listed several times (once per row group); the lists are made unique afterwards
        for (int r = 0; r < N; ++r) {
            if (indptr[r + 1] < indptr[r]) return HPF_E_ARG;
            const int i = bus_of[r];
            for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                const int col = indices[e];
                if (col < 0 || col >= N) return HPF_E_ARG;
                const int j = bus_of[col];
                if (j != i && stamp[j] != r) {
                    if (adj[i].empty() || adj[i].back() != j) adj[i].push_back(j);
                    stamp[j] = r;
                }
            }
        }
        for (auto& a : adj) {
            std::sort(a.begin(), a.end());
            a.erase(std::unique(a.begin(), a.end()), a.end());
        }
    }
    for (int i = 0; i < n; ++i)                                      // symmetric block pattern: every edge is listed from both ends
        for (int j : adj[i])
            if (!std::binary_search(adj[j].begin(), adj[j].end(), i)) return HPF_E_TOPOLOGY;
    // BFS spanning tree from bus 0; the edges off the tree are the ties (kept with i < j)
    std::vector<int> parent(n, -2), order;
    std::vector<std::pair<int, int>> ties;
    order.reserve(n);
    parent[0] = -1;
    order.push_back(0);
    for (size_t h = 0; h < order.size(); ++h) {
        const int i = order[h];
        for (int j : adj[i]) {
            if (parent[j] == -2) {
                parent[j] = i;
                order.push_back(j);
            } else if (j != parent[i] && parent[j] != i && i < j) {
                ties.emplace_back(i, j);
            }
        }
    }
    if ((int)order.size() != n) return HPF_E_TOPOLOGY;               // not connected from bus 0
    const int n_ties = (int)ties.size();
    std::vector<int> Tbus, tie_ptr(n + 1, 0), tie_nb((size_t)2 * n_ties + 1, 0);
    if (n_ties) {
        for (const auto& e : ties) {
            tie_ptr[e.first + 1]++;
            tie_ptr[e.second + 1]++;
        }
        for (int i = 0; i < n; ++i) tie_ptr[i + 1] += tie_ptr[i];
        std::vector<int> fill(tie_ptr.begin(), tie_ptr.end() - 1);
        for (const auto& e : ties) {
            tie_nb[fill[e.first]++] = e.second;
            tie_nb[fill[e.second]++] = e.first;
        }
        for (int i = 0; i < n; ++i) {
            std::sort(tie_nb.begin() + tie_ptr[i], tie_nb.begin() + tie_ptr[i + 1]);
            if (tie_ptr[i + 1] > tie_ptr[i]) Tbus.push_back(i);
        }
    }
    const int mT = (int)Tbus.size();
    if ((long long)mT * b > 16384) return HPF_E_TOPOLOGY;            // dense border system: the same bound as the loop's bordered step
    const int m = mT * b;
    std::vector<int> child_ptr(n + 1, 0), child(n > 1 ? n - 1 : 1), height(n, 0), depth(n, 0);
    for (int i = 1; i < n; ++i) child_ptr[parent[i] + 1]++;
    for (int i = 0; i < n; ++i) child_ptr[i + 1] += child_ptr[i];
    {
        std::vector<int> fill(child_ptr.begin(), child_ptr.end() - 1);
        for (int i = 1; i < n; ++i) child[fill[parent[i]]++] = i;   // ascending bus order inside a children list
    }
    int n_levels = 0, n_depths = 0;
    for (int h = n - 1; h >= 0; --h) {                               // BFS order reversed: children before parents
        const int i = order[h];
        if (parent[i] >= 0 && height[parent[i]] < height[i] + 1) height[parent[i]] = height[i] + 1;
        if (height[i] + 1 > n_levels) n_levels = height[i] + 1;
    }
    for (int h = 1; h < n; ++h) {
        depth[order[h]] = depth[parent[order[h]]] + 1;
        if (depth[order[h]] + 1 > n_depths) n_depths = depth[order[h]] + 1;
    }
    if (n_depths < 1) n_depths = 1;
    std::vector<int> lvl_ptr(n_levels + 1, 0), lvl_nodes(n), dep_ptr(n_depths + 1, 0), dep_nodes(n);
    for (int i = 0; i < n; ++i) {
        lvl_ptr[height[i] + 1]++;
        dep_ptr[depth[i] + 1]++;
    }
    for (int l = 0; l < n_levels; ++l) lvl_ptr[l + 1] += lvl_ptr[l];
    for (int l = 0; l < n_depths; ++l) dep_ptr[l + 1] += dep_ptr[l];
    {
        std::vector<int> fl(lvl_ptr.begin(), lvl_ptr.end() - 1), fd(dep_ptr.begin(), dep_ptr.end() - 1);
        for (int i = 0; i < n; ++i) {
            lvl_nodes[fl[height[i]]++] = i;
            dep_nodes[fd[depth[i]]++] = i;
        }
    }
    // meshed: P = union of the endpoints' root paths (ordered by depth), the (bus, endpoint) pairs of the forward phase (ordered by height)
    struct Pair {
        int bus, t, pred, child;
    };
    std::vector<int> pidx(n, -1), Pbus, pair_of, pair_lvl_ptr(n_levels + 1, 0), pdep_ptr(n_depths + 1, 0);
    std::vector<Pair> pairs;
    if (n_ties) {
        std::vector<char> inP(n, 0);
        for (int t = 0; t < mT; ++t)
            for (int k = Tbus[t]; k >= 0 && !inP[k]; k = parent[k]) inP[k] = 1;
        for (int h = 0; h < n; ++h)
            if (inP[order[h]]) {
                pidx[order[h]] = (int)Pbus.size();
                Pbus.push_back(order[h]);
                pdep_ptr[depth[order[h]] + 1]++;
            }
        for (int l = 0; l < n_depths; ++l) pdep_ptr[l + 1] += pdep_ptr[l];
        std::vector<Pair> raw;
        for (int t = 0; t < mT; ++t) {
            int pred = -1, ch = -1;
            for (int k = Tbus[t]; k >= 0; k = parent[k]) {
                raw.push_back({k, t, pred, ch});
                pred = (int)raw.size() - 1;
                ch = k;
            }
        }
        std::vector<int> perm(raw.size()), newidx(raw.size());
        for (size_t q = 0; q < raw.size(); ++q) perm[q] = (int)q;
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int c2) { return height[raw[a].bus] < height[raw[c2].bus]; });
        for (size_t q = 0; q < raw.size(); ++q) newidx[perm[q]] = (int)q;
        pairs.resize(raw.size());
        pair_of.assign((size_t)Pbus.size() * mT, -1);
        for (size_t q = 0; q < raw.size(); ++q) {
            pairs[q] = raw[perm[q]];
            if (pairs[q].pred >= 0) pairs[q].pred = newidx[pairs[q].pred];
            pair_lvl_ptr[height[pairs[q].bus] + 1]++;
            pair_of[(size_t)pidx[pairs[q].bus] * mT + pairs[q].t] = (int)q;
        }
        for (int l = 0; l < n_levels; ++l) pair_lvl_ptr[l + 1] += pair_lvl_ptr[l];
    }
    const size_t nP = Pbus.size(), npairs = pairs.size();
    const size_t njobs = n_ties ? (nP - 1) + npairs + nP * (size_t)mT : 0;
    // ---- device -------------------------------------------------------------------------------------------------------------------------
    const double ms_host = ms_since(t_0);
    const auto t_1 = std::chrono::steady_clock::now();
    if (hipSetDevice(device) != hipSuccess) return HPF_E_HIP;
    const size_t nnz = (size_t)indptr[N], bb = (size_t)b * b;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return HPF_E_HIP;
        double need = 8.0 * (4.0 * (double)n * (double)bb + 3.0 * (double)n * b + 2.0 * N) + 12.0 * (double)nnz + 64.0 * 1048576.0;
        if (n_ties)
            need += 8.0 * ((double)bb * ((double)n + 2.0 * n_ties + (double)nP + (double)npairs + (double)nP * mT) + (double)m * m + 2.0 * (double)n * b) +
                    40.0 * (double)njobs;
        if (need > (double)free_b) return HPF_E_NOMEM;
    }
    int *d_indptr, *d_indices, *d_parent, *d_child_ptr, *d_child, *d_lvl, *d_dep, *d_sing;
    double *d_data, *d_f, *d_D, *d_Aup, *d_Adn, *d_y, *d_Z, *d_w, *d_xb, *d_dx;
    int *d_Tbus = nullptr, *d_tie_ptr = nullptr, *d_tie_nb = nullptr, *d_pidx = nullptr, *d_ipiv = nullptr, *d_info = nullptr;
    double *d_Tie = nullptr, *d_Sinv = nullptr, *d_Up = nullptr, *d_Wm = nullptr, *d_X = nullptr, *d_Gm = nullptr, *d_gr = nullptr, *d_y2 = nullptr, *d_w2 = nullptr;
    BlkJob* d_jobs = nullptr;
    DevPool B;
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 sizes the pool, pass 1 carves it
        B.used = 0;
        B.carve(&d_D, (size_t)n * bb);                   // (the zero-initialised block arrays and y first: one memset)
        B.carve(&d_Aup, (size_t)n * bb);
        B.carve(&d_Adn, (size_t)n * bb);
        if (n_ties) B.carve(&d_Tie, (size_t)2 * n_ties * bb);
        B.carve(&d_y, (size_t)n * b);
        B.carve(&d_sing, (size_t)1);
        const size_t zeroed = B.used;
        if (n_ties) {
            B.carve(&d_Sinv, (size_t)n * bb);
            B.carve(&d_Up, nP * bb);
            B.carve(&d_Wm, npairs * bb);
            B.carve(&d_X, nP * (size_t)mT * bb);
            B.carve(&d_Gm, (size_t)m * m);
            B.carve(&d_gr, (size_t)m);
            B.carve(&d_y2, (size_t)n * b);
            B.carve(&d_w2, (size_t)n * b);
            B.carve(&d_jobs, njobs);
            B.carve(&d_ipiv, (size_t)m);
            B.carve(&d_info, (size_t)1);
            B.carve(&d_Tbus, (size_t)mT);
            B.carve(&d_tie_ptr, (size_t)n + 1);
            B.carve(&d_tie_nb, (size_t)2 * n_ties);
            B.carve(&d_pidx, (size_t)n);
        }
        B.carve(&d_Z, (size_t)n * bb);
        B.carve(&d_w, (size_t)n * b);
        B.carve(&d_xb, (size_t)n * b);
        B.carve(&d_dx, (size_t)N);
        B.carve(&d_data, nnz);
        B.carve(&d_f, (size_t)N);
        B.carve(&d_indptr, (size_t)N + 1);
        B.carve(&d_indices, nnz);
        B.carve(&d_parent, (size_t)n);
        B.carve(&d_child_ptr, (size_t)n + 1);
        B.carve(&d_child, child.size());
        B.carve(&d_lvl, (size_t)n);
        B.carve(&d_dep, (size_t)n);
        if (pass == 0) {
            B.cap = B.used;
            if (hipMalloc((void**)&B.base, B.cap) != hipSuccess) {
                B.base = nullptr;
                return HPF_E_NOMEM;
            }
        } else if (hipMemsetAsync(B.base, 0, zeroed, nullptr) != hipSuccess) {
            return HPF_E_HIP;
        }
    }
    hipStream_t st = nullptr;                            // (the legacy default stream: every call below is ordered, the copies are synchronous)
    bool ok = hipMemcpy(d_indptr, indptr, sizeof(int) * ((size_t)N + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_indices, indices, sizeof(int) * nnz, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_data, data, sizeof(double) * nnz, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_f, f, sizeof(double) * (size_t)N, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_parent, parent.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_child_ptr, child_ptr.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_child, child.data(), sizeof(int) * child.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_lvl, lvl_nodes.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_dep, dep_nodes.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) return HPF_E_HIP;
    // meshed: the block-product jobs of the selected inversion, in launch order: Up | forward by height | back by depth
    std::vector<size_t> fwd_beg(n_levels + 1, 0), back_beg(n_depths + 1, 0);
    if (n_ties) {
        std::vector<BlkJob> jobs;
        jobs.reserve(njobs);
        for (size_t q = 1; q < nP; ++q) {                // (Pbus[0] is the root)
            const int cbus = Pbus[q];
            jobs.push_back({d_Sinv + (size_t)parent[cbus] * bb, d_Adn + (size_t)cbus * bb, nullptr, d_Up + q * bb, 1.0});
        }
        for (int l = 0; l < n_levels; ++l) {
            fwd_beg[l] = jobs.size();
            for (int q = pair_lvl_ptr[l]; q < pair_lvl_ptr[l + 1]; ++q) {
                const Pair& pr = pairs[q];
                if (pr.pred < 0)
                    jobs.push_back({nullptr, d_Sinv + (size_t)pr.bus * bb, nullptr, d_Wm + (size_t)q * bb, 1.0});
                else
                    jobs.push_back({d_Up + (size_t)pidx[pr.child] * bb, d_Wm + (size_t)pr.pred * bb, nullptr, d_Wm + (size_t)q * bb, -1.0});
            }
        }
        fwd_beg[n_levels] = jobs.size();
        for (int dl = 0; dl < n_depths; ++dl) {
            back_beg[dl] = jobs.size();
            for (int q = pdep_ptr[dl]; q < pdep_ptr[dl + 1]; ++q) {
                const int k = Pbus[q];
                for (int t = 0; t < mT; ++t) {
                    const int pq = pair_of[(size_t)q * mT + t];
                    const double* wq = pq >= 0 ? d_Wm + (size_t)pq * bb : nullptr;
                    double* xo = d_X + ((size_t)q * mT + t) * bb;
                    if (parent[k] < 0)
                        jobs.push_back({nullptr, wq, nullptr, xo, 1.0});                      // (the root is on every path: wq != nullptr)
                    else
                        jobs.push_back({d_Z + (size_t)k * bb, d_X + ((size_t)pidx[parent[k]] * mT + t) * bb, wq, xo, -1.0});
                }
            }
        }
        back_beg[n_depths] = jobs.size();
        if (jobs.size() != njobs) return HPF_E_STATE;
        ok = hipMemcpy(d_jobs, jobs.data(), sizeof(BlkJob) * njobs, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_Tbus, Tbus.data(), sizeof(int) * (size_t)mT, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_tie_ptr, tie_ptr.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_tie_nb, tie_nb.data(), sizeof(int) * (size_t)2 * n_ties, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_pidx, pidx.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) return HPF_E_HIP;
    }
    double ms_up = 0.0;
    std::chrono::steady_clock::time_point t_2 = std::chrono::steady_clock::now();
    if (info) {
        hipDeviceSynchronize();
        ms_up = ms_since(t_1);
        t_2 = std::chrono::steady_clock::now();
    }
    hipLaunchKernelGGL(k_csr_pad, dim3((unsigned)(((size_t)n * b + 255) / 256)), dim3(256), 0, st, n, c, b, d_D);
    hipLaunchKernelGGL(k_csr_scatter, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, n, c, Nc, b, d_indptr, d_indices, d_data, d_f, d_parent,
                       d_D, d_Aup, d_Adn, d_y, d_tie_ptr, d_tie_nb, d_Tie);
    if (hipGetLastError() != hipSuccess) return HPF_E_HIP;
    const int R = (b + 15) / 16;
    for (int l = 0; l < n_levels; ++l) {
        const int cnt = lvl_ptr[l + 1] - lvl_ptr[l];
        const int* nodes = d_lvl + lvl_ptr[l];
        hipError_t e = hipSuccess;
#define HPF_CSR_CASE(RR) \
    case RR: e = launch_factor<RR>(b, cnt, nodes, d_parent, d_child_ptr, d_child, d_D, d_Aup, d_Adn, d_y, d_Z, d_w, d_sing, d_Sinv, st); break;
        switch (R) {
            HPF_CSR_CASE(1)
            HPF_CSR_CASE(2)
            HPF_CSR_CASE(3)
            HPF_CSR_CASE(4)
            HPF_CSR_CASE(5)
            HPF_CSR_CASE(6)
            HPF_CSR_CASE(7)
            HPF_CSR_CASE(8)
            default: return HPF_E_ARG;
        }
#undef HPF_CSR_CASE
        if (e != hipSuccess) return HPF_E_HIP;
    }
    for (int dl = 0; dl < n_depths; ++dl) {
        const int cnt = dep_ptr[dl + 1] - dep_ptr[dl];
        hipLaunchKernelGGL(k_csr_back, dim3((unsigned)cnt), dim3(256), 0, st, n, c, Nc, b, d_dep + dep_ptr[dl], d_parent, d_Z, d_w, d_xb, d_dx);
    }
    if (hipGetLastError() != hipSuccess) return HPF_E_HIP;
    double ms_tree = 0.0, ms_sel = 0.0, ms_lu = 0.0;
    if (info) {
        hipDeviceSynchronize();
        ms_tree = ms_since(t_2);
    }
    int border_info = 0;
    if (n_ties) {
        // ---- selected inversion over P, border system, one more right-hand-side sweep ----------------------------------------------------
        auto t_3 = std::chrono::steady_clock::now();
        launch_jobs(R, b, (int)(nP - 1), d_jobs, st);
        for (int l = 0; l < n_levels; ++l) launch_jobs(R, b, (int)(fwd_beg[l + 1] - fwd_beg[l]), d_jobs + fwd_beg[l], st);
        for (int dl = 0; dl < n_depths; ++dl) launch_jobs(R, b, (int)(back_beg[dl + 1] - back_beg[dl]), d_jobs + back_beg[dl], st);
        launch_border(R, b, mT, d_Tbus, d_tie_ptr, d_tie_nb, d_pidx, d_Tie, d_X, d_xb, d_Gm, d_gr, st);
        if (hipGetLastError() != hipSuccess) return HPF_E_HIP;
        if (info) {
            hipDeviceSynchronize();
            ms_sel = ms_since(t_3);
            t_3 = std::chrono::steady_clock::now();
        }
        BlasHandle blas;
        if (rocblas_create_handle(&blas.h) != rocblas_status_success) return HPF_E_ROCSOLVER;
        if (rocsolver_dgetrf(blas.h, m, m, d_Gm, m, d_ipiv, d_info) != rocblas_status_success ||
            rocsolver_dgetrs(blas.h, rocblas_operation_none, m, 1, d_Gm, m, d_ipiv, d_gr, m) != rocblas_status_success)
            return HPF_E_ROCSOLVER;
        if (hipMemcpy(&border_info, d_info, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return HPF_E_HIP;
        if (info) ms_lu = ms_since(t_3);
        if (hipMemcpyAsync(d_y2, d_y, sizeof(double) * (size_t)n * b, hipMemcpyDeviceToDevice, st) != hipSuccess) return HPF_E_HIP;
        hipLaunchKernelGGL(k_border_rhs, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, b, m, d_Tbus, d_gr, d_y2);
        for (int l = 0; l < n_levels; ++l)
            hipLaunchKernelGGL(k_csr_fwd, dim3((unsigned)(lvl_ptr[l + 1] - lvl_ptr[l])), dim3(256), sizeof(double) * (size_t)b, st, b, d_lvl + lvl_ptr[l],
                               d_child_ptr, d_child, d_Sinv, d_Adn, d_y2, d_w2);
        for (int dl = 0; dl < n_depths; ++dl)
            hipLaunchKernelGGL(k_csr_back, dim3((unsigned)(dep_ptr[dl + 1] - dep_ptr[dl])), dim3(256), 0, st, n, c, Nc, b, d_dep + dep_ptr[dl], d_parent, d_Z,
                               d_w2, d_xb, d_dx);
        if (hipGetLastError() != hipSuccess) return HPF_E_HIP;
    }
    if (info) {
        hipDeviceSynchronize();
        fprintf(stderr, "hpf_sparse_solve: N %d, %zu entries, %d buses in %d levels / %d depths: host analysis %.2f ms, allocation + upload %.2f ms, "
                        "scatter + factor + back sweep %.2f ms\n", N, nnz, n, n_levels, n_depths, ms_host, ms_up, ms_tree);
        if (n_ties)
            fprintf(stderr, "hpf_sparse_solve: %d ties, %d endpoint buses (border %d), %zu buses on their root paths, %zu forward pairs, %zu block products: "
                            "selected inversion + border matrix %.2f ms, border LU %.2f ms, whole call %.2f ms\n", n_ties, mT, m, nP, npairs, njobs, ms_sel,
                    ms_lu, ms_since(t_0));
    }
    if (border_info) return HPF_E_SINGULAR;
    int sing = 0;
    if (hipMemcpy(&sing, d_sing, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(dx, d_dx, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost) != hipSuccess)
        return HPF_E_HIP;
    return sing ? HPF_E_SINGULAR : HPF_OK;
}
