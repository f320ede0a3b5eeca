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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "../../include/hpf.h"
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
                              double* __restrict__ D, double* __restrict__ Aup, double* __restrict__ Adn, double* __restrict__ y) {
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
        double* dst = j == i ? D + (size_t)i * bb : (j == par ? Aup + (size_t)i * bb : Adn + (size_t)j * bb);   // (else: parent[j] == i, checked on the host)
        dst[(size_t)l * b + lc] += data[e];
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_csr_factor(int b, const int* __restrict__ nodes, const int* __restrict__ parent,
                                                    const int* __restrict__ child_ptr, const int* __restrict__ child,
                                                    const double* __restrict__ D, const double* __restrict__ Aup,
                                                    const double* __restrict__ Adn, const double* __restrict__ y, double* __restrict__ Z,
                                                    double* __restrict__ w, int* __restrict__ singular) {
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
    // ---- D. w_k = D^-1 y, Z_k = D^-1 Aup_k -----------------------------------------------------------------------------------------
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
                         const double* Aup, const double* Adn, const double* y, double* Z, double* w, int* singular, hipStream_t st) {
    const size_t lds = gj_dense_lds_bytes(b);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_factor<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_csr_factor<R>), dim3((unsigned)count), dim3(256), lds, st, b, nodes, parent, child_ptr, child, D, Aup, Adn, y, Z, w, singular);
    return hipGetLastError();
}

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
    long long n_edges2 = 0;
    for (int i = 0; i < n; ++i) n_edges2 += (long long)adj[i].size();
    if (n_edges2 != 2ll * (n - 1)) return HPF_E_TOPOLOGY;          // a tree has n - 1 undirected edges, each listed from both ends
    std::vector<int> parent(n, -2), order;
    order.reserve(n);
    parent[0] = -1;
    order.push_back(0);
    for (size_t h = 0; h < order.size(); ++h) {
        const int i = order[h];
        for (int j : adj[i]) {
            if (parent[j] == -2) {
                parent[j] = i;
                order.push_back(j);
            } else if (j != parent[i]) {
                return HPF_E_TOPOLOGY;                               // a cycle, or a block pattern that is not symmetric
            }
        }
    }
    if ((int)order.size() != n) return HPF_E_TOPOLOGY;               // not connected from bus 0
    for (int i = 1; i < n; ++i) {                                    // symmetric pattern: the parent lists the child too
        const auto& a = adj[parent[i]];
        if (!std::binary_search(a.begin(), a.end(), i)) return HPF_E_TOPOLOGY;
    }
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
    // ---- device -------------------------------------------------------------------------------------------------------------------------
    const double ms_host = ms_since(t_0);
    const auto t_1 = std::chrono::steady_clock::now();
    if (hipSetDevice(device) != hipSuccess) return HPF_E_HIP;
    const size_t nnz = (size_t)indptr[N], bb = (size_t)b * b;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return HPF_E_HIP;
        const double need = 8.0 * (4.0 * (double)n * (double)bb + 3.0 * (double)n * b + 2.0 * N) + 12.0 * (double)nnz + 64.0 * 1048576.0;
        if (need > (double)free_b) return HPF_E_NOMEM;
    }
    int *d_indptr, *d_indices, *d_parent, *d_child_ptr, *d_child, *d_lvl, *d_dep, *d_sing;
    double *d_data, *d_f, *d_D, *d_Aup, *d_Adn, *d_y, *d_Z, *d_w, *d_xb, *d_dx;
    DevPool B;
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 sizes the pool, pass 1 carves it
        B.used = 0;
        B.carve(&d_D, (size_t)n * bb);                   // (the three zero-initialised block arrays and y first: one memset)
        B.carve(&d_Aup, (size_t)n * bb);
        B.carve(&d_Adn, (size_t)n * bb);
        B.carve(&d_y, (size_t)n * b);
        B.carve(&d_sing, (size_t)1);
        const size_t zeroed = B.used;
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
    double ms_up = 0.0;
    std::chrono::steady_clock::time_point t_2 = std::chrono::steady_clock::now();
    if (info) {
        hipDeviceSynchronize();
        ms_up = ms_since(t_1);
        t_2 = std::chrono::steady_clock::now();
    }
    hipLaunchKernelGGL(k_csr_pad, dim3((unsigned)(((size_t)n * b + 255) / 256)), dim3(256), 0, st, n, c, b, d_D);
    hipLaunchKernelGGL(k_csr_scatter, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, n, c, Nc, b, d_indptr, d_indices, d_data, d_f, d_parent,
                       d_D, d_Aup, d_Adn, d_y);
    if (hipGetLastError() != hipSuccess) return HPF_E_HIP;
    const int R = (b + 15) / 16;
    for (int l = 0; l < n_levels; ++l) {
        const int cnt = lvl_ptr[l + 1] - lvl_ptr[l];
        const int* nodes = d_lvl + lvl_ptr[l];
        hipError_t e = hipSuccess;
#define HPF_CSR_CASE(RR) \
    case RR: e = launch_factor<RR>(b, cnt, nodes, d_parent, d_child_ptr, d_child, d_D, d_Aup, d_Adn, d_y, d_Z, d_w, d_sing, st); break;
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
    if (info) {
        hipDeviceSynchronize();
        fprintf(stderr, "hpf_sparse_solve: N %d, %zu entries, %d buses in %d levels / %d depths: host analysis %.2f ms, allocation + upload %.2f ms, "
                        "scatter + factor + back sweep %.2f ms\n", N, nnz, n, n_levels, n_depths, ms_host, ms_up, ms_since(t_2));
    }
    int sing = 0;
    if (hipMemcpy(&sing, d_sing, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(dx, d_dx, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost) != hipSuccess)
        return HPF_E_HIP;
    return sing ? HPF_E_SINGULAR : HPF_OK;
}
