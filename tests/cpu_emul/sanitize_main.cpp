// TEST INFRASTRUCTURE: sanitizer run of the host-executable parts of the product (no GPU):
//   (1) the per-row / per-entry device functions of csrc/hpf_assembly.hpp (mismatch rows, dense and CSR Jacobian targets) on random radial
//       feeders with PV buses, nonlinear buses and both Norton modes -- built with -fsanitize=address,undefined: an index map that reads or
//       writes out of bounds aborts the run; the CSR form must equal the dense target entry for entry;
//   (2) (only with -DWITH_LIBHPF, linked against an ASan / UBSan host build of libhpf.so) hpf_create's argument validation and the
//       host-only elimination-tree planner hpf_tree_plan on random feeders of every block-size class.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "hpf_assembly.hpp"
#ifdef WITH_LIBHPF
#include "../../include/hpf.h"
#endif
using namespace hpf;

struct Feeder {
    int n, m, c, Hn, coupled;
    std::vector<int> rowptr, col, diag, dev;
    std::vector<cplx> Y, YN, IN;
};

static Feeder make(int n, int Hn, int n_pv, double frac_nl, int coupled, unsigned seed, int n_ties = 0) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> u(0.2, 1.5);
    Feeder F;
    F.n = n; F.Hn = Hn; F.coupled = coupled; F.c = 1 + n_pv;
    F.m = n - (int)(frac_nl * n);
    if (F.m < F.c) F.m = F.c;
    std::vector<std::vector<int>> adj(n);
    for (int i = 1; i < n; ++i) {
        const int p = (int)(rng() % i);
        adj[i].push_back(p);
        adj[p].push_back(i);
    }
    for (int t = 0, tries = 0; t < n_ties && tries < 1000; ++tries) {     // loop-closing lines: pairs of buses that are not adjacent yet
        const int a = 1 + (int)(rng() % (n - 1)), b2 = 1 + (int)(rng() % (n - 1));
        if (a == b2 || std::find(adj[a].begin(), adj[a].end(), b2) != adj[a].end()) continue;
        adj[a].push_back(b2);
        adj[b2].push_back(a);
        ++t;
    }
    F.rowptr.assign(1, 0);
    for (int i = 0; i < n; ++i) {
        std::vector<int> r = adj[i];
        r.push_back(i);
        std::sort(r.begin(), r.end());
        for (int j : r) {
            if (j == i) F.diag.push_back((int)F.col.size());
            F.col.push_back(j);
        }
        F.rowptr.push_back((int)F.col.size());
    }
    const int nnz = (int)F.col.size();
    F.Y.resize((size_t)Hn * nnz);
    for (auto& y : F.Y) y = {u(rng), -u(rng)};
    F.dev.assign(n, -1);
    for (int i = F.m; i < n; ++i) F.dev[i] = i & 1;
    F.YN.resize((size_t)2 * Hn * (coupled ? Hn : 1));
    for (auto& y : F.YN) y = {u(rng) * 0.1, u(rng) * 0.1};
    F.IN.resize((size_t)2 * Hn);
    for (auto& y : F.IN) y = {u(rng) * 0.01, u(rng) * 0.01};
    return F;
}

int main() {
    int checked = 0;
    for (unsigned seed = 1; seed <= 12; ++seed) {
        const int n = 5 + (int)(seed * 7 % 40), Hn = 2 + (int)(seed % 6), coupled = seed & 1, n_pv = seed % 3;
        Feeder F = make(n, Hn, n_pv, 0.4, coupled, seed);
        Model M;
        M.n = F.n; M.m = F.m; M.c = F.c; M.Hn = Hn; M.nnz = (int)F.col.size(); M.n_dev = 2; M.coupled = coupled;
        M.rowptr = F.rowptr.data(); M.col = F.col.data(); M.diag = F.diag.data(); M.Y = F.Y.data(); M.dev = F.dev.data();
        M.YN = F.YN.data(); M.IN = F.IN.data();
        const int cnt = n * Hn, Nc = cnt - 1, N = 2 * Nc - (F.c - 1);
        std::vector<cplx> U(cnt), E(cnt);
        std::vector<double> P(n, 0.1), Q(n, 0.05), f(N, 0.0), J((size_t)N * N, 0.0);
        for (int k = 0; k < cnt; ++k) polar<false>(0.3 + 0.01 * k, 0.02 * k, U[k], E[k]);
        for (int k = 1; k < cnt; ++k) store_mismatch(f.data(), Nc, F.c, k, mismatch_row<false>(M, U.data(), P.data(), Q.data(), k));
        DenseEmit em{J.data(), N, Nc, F.c};
        for (int q = 0; q < Hn; ++q)
            for (int i = 0; i < n; ++i)
                for (int e = F.rowptr[i]; e < F.rowptr[i + 1]; ++e) jac_entry(M, U.data(), E.data(), q, i, e, em);
        if (coupled)
            for (int i = F.m; i < n; ++i)
                for (int q = 0; q < Hn; ++q)
                    for (int p = 0; p < Hn; ++p)
                        if (p != q) jac_cross(M, U.data(), E.data(), q, p, i, em);
        std::vector<int> indptr(N + 1, 0);
        long long tot = 0;
        for (int r = 0; r < N; ++r) {
            indptr[r] = (int)tot;
            const JCount k = jcsr_count_row(M, Nc, r);
            tot += k.n_theta + k.n_v;
        }
        indptr[N] = (int)tot;
        std::vector<int> indices(tot, -1);
        std::vector<double> data(tot, 0.0);
        for (int r = 0; r < N; ++r) jcsr_fill_row(M, U.data(), E.data(), Nc, r, indptr[r], indices.data(), data.data());
        std::vector<double> J2((size_t)N * N, 0.0);
        for (int r = 0; r < N; ++r)
            for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
                if (indices[e] < 0 || indices[e] >= N || (e > indptr[r] && indices[e] <= indices[e - 1])) {
                    printf("seed %u: bad column in row %d\n", seed, r);
                    return 1;
                }
                J2[(size_t)indices[e] * N + r] = data[e];
            }
        if (memcmp(J.data(), J2.data(), sizeof(double) * J.size()) != 0) {
            printf("seed %u: CSR form differs from the dense target\n", seed);
            return 1;
        }
        ++checked;
#ifdef WITH_LIBHPF
        hpf_desc d;
        memset(&d, 0, sizeof d);
        d.n = F.n; d.m = F.m; d.c = F.c; d.Hn = Hn; d.nnz = M.nnz; d.n_dev = 2; d.coupled = coupled; d.solver = HPF_SOLVER_BLOCK_TREE;
        d.max_scenarios = seed % 2 ? 4 : 300;
        d.rowptr = F.rowptr.data(); d.col = F.col.data(); d.Yval = (const double*)F.Y.data(); d.dev_of_bus = F.dev.data();
        d.Y_N = (const double*)F.YN.data(); d.I_N = (const double*)F.IN.data();
        hpf_handle* h = nullptr;
        hpf_desc bad = d;
        bad.n = 0;
        if (hpf_create(&h, &bad) != HPF_E_ARG) return 2;
        bad = d;
        bad.rowptr = nullptr;
        if (hpf_create(&h, &bad) != HPF_E_ARG) return 2;
        // hpf_sparse_solve: the host-side analysis of the pattern (bus adjacency from the CSR Jacobian just built, BFS tree, levels) under the
        // sanitizers.  These feeders are radial: the analysis must accept them, so without a GPU the call ends at the first HIP call
        // (HPF_E_HIP); a ring closed by one extra entry on ONE side only (a block pattern that is not symmetric) must be refused (HPF_E_TOPOLOGY) before
        // any HIP call, a bad index with HPF_E_ARG; the ring closed from both ends is a meshed pattern and is accepted.
        {
            std::vector<double> dx(N, 0.0);
            const int rc = hpf_sparse_solve(0, n, F.c, Hn, indptr.data(), indices.data(), data.data(), f.data(), dx.data());
            if (rc == HPF_E_ARG || rc == HPF_E_TOPOLOGY) {
                printf("seed %u: hpf_sparse_solve refused a radial feeder: %d\n", seed, rc);
                return 4;
            }
            if (Hn >= 2 && n >= 4) {
                // one more entry: row of (bus n-1, harmonic position 1) gets a column of bus a, a not adjacent to it -> a cycle in the bus graph
                const int rbus = n - 1;
                int a = -1;
                for (int cand = 1; cand < n - 1 && a < 0; ++cand) {
                    bool adj = false;
                    for (int e = F.rowptr[rbus]; e < F.rowptr[rbus + 1]; ++e) adj = adj || F.col[e] == cand;
                    if (!adj) a = cand;
                }
                if (a >= 0) {
                    const int r = 1 * n + rbus - 1, cnew = 1 * n + a - 1;          // Re row of stacked k = n + rbus, theta column of k = n + a
                    std::vector<int> ip2(indptr), ix2;
                    std::vector<double> dt2;
                    for (int rr = 0; rr < N; ++rr) {
                        ip2[rr] = (int)ix2.size();
                        for (int e = indptr[rr]; e < indptr[rr + 1]; ++e) {
                            ix2.push_back(indices[e]);
                            dt2.push_back(data[e]);
                        }
                        if (rr == r) {
                            ix2.push_back(cnew);
                            dt2.push_back(1.0);
                        }
                    }
                    ip2[N] = (int)ix2.size();
                    if (hpf_sparse_solve(0, n, F.c, Hn, ip2.data(), ix2.data(), dt2.data(), f.data(), dx.data()) != HPF_E_TOPOLOGY) return 4;
                    {
                        // the same edge listed from BOTH ends: a meshed pattern (spanning tree + one tie) -- the analysis (ties, root paths, forward pairs,
                        // block-product jobs are built after the first HIP call, which fails here) must accept it
                        const int r2 = 1 * n + a - 1, c2 = 1 * n + rbus - 1;
                        std::vector<int> ip3(ip2), ix3;
                        std::vector<double> dt3;
                        for (int rr = 0; rr < N; ++rr) {
                            ip3[rr] = (int)ix3.size();
                            for (int e = ip2[rr]; e < ip2[rr + 1]; ++e) {
                                ix3.push_back(ix2[e]);
                                dt3.push_back(dt2[e]);
                            }
                            if (rr == r2) {
                                ix3.push_back(c2);
                                dt3.push_back(1.0);
                            }
                        }
                        ip3[N] = (int)ix3.size();
                        const int rc3 = hpf_sparse_solve(0, n, F.c, Hn, ip3.data(), ix3.data(), dt3.data(), f.data(), dx.data());
                        if (rc3 == HPF_E_ARG || rc3 == HPF_E_TOPOLOGY) {
                            printf("seed %u: hpf_sparse_solve refused a feeder with one tie: %d\n", seed, rc3);
                            return 4;
                        }
                    }
                    ix2[0] = N + 5;
                    if (hpf_sparse_solve(0, n, F.c, Hn, ip2.data(), ix2.data(), dt2.data(), f.data(), dx.data()) != HPF_E_ARG) return 4;
                }
            }
        }
#endif
    }
#ifdef WITH_LIBHPF
    // the elimination-tree planner on feeders of every block-size class (b = 12, 28, 52, 100) and both capacity classes of the compress steps
    const int hn_list[4] = {5, 13, 26, 40};
    for (int t = 0; t < 8; ++t) {
        const int Hn = hn_list[t % 4], n = 120 + 60 * t;
        const int ties = (t & 1) ? 1 + t / 2 : 0;              // every other feeder is meshed: tree_find_ties + the planner's plain-Gauss-Jordan mask
        Feeder F = make(n, Hn, t % 3, 0.35, 1, 100 + t, ties);
        hpf_desc d;
        memset(&d, 0, sizeof d);
        d.n = F.n; d.m = F.m; d.c = F.c; d.Hn = Hn; d.nnz = (int)F.col.size(); d.n_dev = 2; d.coupled = 1; d.solver = HPF_SOLVER_BLOCK_TREE;
        d.max_scenarios = t < 4 ? 8 : 512;
        d.rowptr = F.rowptr.data(); d.col = F.col.data(); d.Yval = (const double*)F.Y.data(); d.dev_of_bus = F.dev.data();
        d.Y_N = (const double*)F.YN.data(); d.I_N = (const double*)F.IN.data();
        const int rc = hpf_tree_plan(&d, "/tmp/hpf_sanitize_plan.txt");
        if (rc != HPF_OK) {
            printf("hpf_tree_plan failed: %d (n %d Hn %d, %d ties)\n", rc, n, Hn, ties);
            return 3;
        }
        if (ties) {                                            // every marked bus must have come out as a plain Gauss-Jordan bus (last column 1, never -1)
            FILE* fp = fopen("/tmp/hpf_sanitize_plan.txt", "r");
            char line[512];
            int marked = 0, bad = 0, meshed_line = 0;
            while (fp && fgets(line, sizeof line, fp)) {
                if (line[0] == '#') {
                    meshed_line += strstr(line, "# meshed:") != nullptr;
                    continue;
                }
                int v[10];
                if (sscanf(line, "%d %d %d %d %d %d %d %d %d %d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8, v + 9) != 10) continue;
                marked += v[9] == 1;
                bad += v[9] == -1 || (v[9] == 1 && (v[4] != 0 || v[5] != 0 || v[7] != 0 || v[8] != 0));
            }
            if (fp) fclose(fp);
            if (!meshed_line || marked < 2 || bad) {
                printf("meshed plan: %d marked buses, %d faults, meshed line %d (n %d Hn %d, %d ties)\n", marked, bad, meshed_line, n, Hn, ties);
                return 3;
            }
        }
        ++checked;
    }
#endif
    printf("sanitize_main: %d cases clean\n", checked);
    return 0;
}
