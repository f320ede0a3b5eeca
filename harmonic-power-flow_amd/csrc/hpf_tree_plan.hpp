// Host-side planner of the block elimination tree (tree_build_into): spanning tree, contraction, bus kinds (2x2 algebra, constant-inverse and lazy
// leaves, bordered buses, Gauss-Jordan buses, compress steps), per-model images, level / depth lists, node records and their upload.  Included by
// hpf_block.hip (one translation unit: it uses that file's tile layouts and upload helpers); split off in round 4 to keep the kernels and the planner apart.
#pragma once

// row update of the host-side complex Gauss-Jordan (tree_build_into): r -= f * c on split re / im rows; an AVX2 build of the same loop is
// picked at run time where the CPU has it (the set-up of a 10 000-bus model is two thousand 49 x 49 complex inversions)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) static void cplx_row_axpy_avx2(double* __restrict__ rr, double* __restrict__ ri, const double* __restrict__ cr,
                                                                const double* __restrict__ ci, double fr, double fi, int nn) {
    for (int c2 = 0; c2 < nn; ++c2) {
        rr[c2] -= fr * cr[c2] - fi * ci[c2];
        ri[c2] -= fr * ci[c2] + fi * cr[c2];
    }
}
#endif
// host threads of the model set-up (HPF_HOST_THREADS, default: the hardware's, at most 16)
static unsigned host_threads() {
    if (const char* e = getenv("HPF_HOST_THREADS")) return atoi(e) < 1 ? 1u : (unsigned)atoi(e);
    const unsigned hw = std::thread::hardware_concurrency();
    return hw < 1 ? 1u : (hw > 16 ? 16u : hw);
}
static void cplx_row_axpy(double* __restrict__ rr, double* __restrict__ ri, const double* __restrict__ cr, const double* __restrict__ ci,
                          double fr, double fi, int nn) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    static const bool has_avx2 = __builtin_cpu_supports("avx2");
    if (has_avx2) {
        cplx_row_axpy_avx2(rr, ri, cr, ci, fr, fi, nn);
        return;
    }
#endif
    for (int c2 = 0; c2 < nn; ++c2) {
        rr[c2] -= fr * cr[c2] - fi * ci[c2];
        ri[c2] -= fr * ci[c2] + fi * cr[c2];
    }
}

namespace hpf {

static void tree_free_one_fwd(Tree& T);
static int tree_build_into(hpf_handle* h, const hpf_desc* d, Tree& T, bool contract) {
    const auto t_begin = std::chrono::steady_clock::now();
    const int n = d->n;
    const int b = 2 * d->Hn;
    if (b > 16 * 7) return HPF_E_ARG;                          // register tile limit (K <= 55)
    if (d->nnz != n + 2 * (n - 1) + 2 * h->n_ties) return HPF_E_TOPOLOGY;   // (tree_find_ties counted the loop-closing lines)
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
    // buses whose whole subtree is linear: 2x2-per-harmonic algebra (multi-wave / wave-per-bus kernels only: the plain tree of a
    // b > 52 model is the one the generic pivoted kernels run on, which eliminate every bus as a full block)
    const bool use_lin = wave_block_size(b) != 0 && (contract || wave_block_size(b) <= 52);
    T.lin.assign(n, 0);
    if (use_lin) {
        for (int i = 0; i < n; ++i) T.lin[i] = (i < d->m || !d->coupled) ? 1 : 0;
        for (int oi = n - 1; oi > 0; --oi) {
            const int i = order[oi];
            if (!T.lin[i]) T.lin[T.parent[i]] = 0;
        }
    }
    // factor-once bordered step of a meshed network (h->sel_forced, set by tree_find_ties): the buses on the root paths of the ties' endpoints
    // stay PLAIN Gauss-Jordan buses -- no 2x2 algebra, no contracted chain, no constant-inverse / lazy / bordered form, no compress role --, so
    // that their inverse S_k^-1 is in the inverse slot after a sweep: the selected inversion of tree_sel_run reads it from there
    const std::vector<char>& fmask = h->sel_forced;
    const bool use_fp = contract && (int)fmask.size() == n;
    if (use_fp)
        for (int i = 0; i < n; ++i)
            if (fmask[i]) T.lin[i] = 0;                           // (closed under "parent of": the rule above stays true)
    // pass-through buses (contract): linear bus, not the root, exactly one child with nonlinear buses below it -> its block is
    // harmonic-diagonal and eliminating it FIRST only re-links that child to the grandparent (2x2-per-harmonic fill)
    std::vector<int> pass(n, 0), ndc(n, 0);
    for (int i = 1; i < n; ++i)
        if (!T.lin[i]) ndc[T.parent[i]]++;
    if (contract)
        for (int i = 1; i < n; ++i) pass[i] = (!T.lin[i] && i < d->m && ndc[i] == 1 && !(use_fp && fmask[i])) ? 1 : 0;
    auto kept = [&](int i) { return !T.lin[i] && !pass[i]; };
    std::vector<int> pard(n, -1);                              // parent in the dense tree (through chains)
    for (size_t oi = 1; oi < order.size(); ++oi) {
        const int i = order[oi];
        if (!kept(i)) continue;
        int p = T.parent[i];
        while (p >= 0 && pass[p]) p = T.parent[p];
        pard[i] = p;
    }
    // heights / depths over the dense tree (linear subtrees and chains are finished before the first dense level)
    for (size_t oi = 1; oi < order.size(); ++oi) {              // BFS order: parents before children
        const int i = order[oi];
        if (kept(i)) depth[i] = depth[pard[i]] + 1;
    }
    for (int oi = n - 1; oi > 0; --oi) {
        const int i = order[oi];
        if (!kept(i)) continue;
        const int p = pard[i];
        if (height[i] + 1 > height[p]) height[p] = height[i] + 1;
    }
    int maxh = 0, maxd = 0;
    T.n_dense = 0;
    for (int i = 0; i < n; ++i) {
        if (!kept(i)) continue;
        ++T.n_dense;
        maxh = height[i] > maxh ? height[i] : maxh;
        maxd = depth[i] > maxd ? depth[i] : maxd;
    }
    T.n_levels = T.n_dense ? maxh + 1 : 0;
    T.n_depths = T.n_dense ? maxd + 1 : 0;
    auto bucket = [&](const std::vector<int>& key, int nb, std::vector<int>& ptr, std::vector<int>& items) {
        ptr.assign(nb + 1, 0);
        for (int i = 0; i < n; ++i)
            if (kept(i)) ptr[key[i] + 1]++;
        for (int l = 0; l < nb; ++l) ptr[l + 1] += ptr[l];
        items.assign(T.n_dense, 0);
        std::vector<int> pos(ptr.begin(), ptr.end() - 1);
        for (int i = 0; i < n; ++i)
            if (kept(i)) items[pos[key[i]]++] = i;               // ascending bus index inside a level
    };
    bucket(height, T.n_levels, T.lvl_ptr, T.lvl_nodes);
    bucket(depth, T.n_depths, T.dep_ptr, T.dep_nodes);
    // network children lists: linear-subtree children first, then pass-through children, then dense children, ascending bus
    // index inside each group.  child_mid: end of the part the bus folds in 2x2-per-harmonic algebra -- the linear subtrees
    // and, at a dense bus, the tops of contracted chains; a pass-through bus folds its linear subtrees only.
    T.child_ptr.assign(n + 1, 0);
    for (int i = 1; i < n; ++i) T.child_ptr[T.parent[i] + 1]++;
    for (int i = 0; i < n; ++i) T.child_ptr[i + 1] += T.child_ptr[i];
    T.child.assign(n > 1 ? n - 1 : 0, 0);
    T.child_mid.assign(n, 0);
    {
        std::vector<int> pos(T.child_ptr.begin(), T.child_ptr.end() - 1);
        for (int i = 1; i < n; ++i)
            if (T.lin[i]) T.child[pos[T.parent[i]]++] = i;
        for (int i = 0; i < n; ++i)
            if (pass[i]) T.child_mid[i] = pos[i];
        for (int i = 1; i < n; ++i)
            if (pass[i]) T.child[pos[T.parent[i]]++] = i;
        for (int i = 0; i < n; ++i)
            if (!pass[i]) T.child_mid[i] = pos[i];
        for (int i = 1; i < n; ++i)
            if (kept(i)) T.child[pos[T.parent[i]]++] = i;
    }
    // dense children of the dense buses (through chains), ascending bus index
    std::vector<int> dchild_ptr(n + 1, 0), dchild(T.n_dense > 0 ? T.n_dense - 1 : 0, 0);
    for (int i = 1; i < n; ++i)
        if (kept(i)) dchild_ptr[pard[i] + 1]++;
    for (int i = 0; i < n; ++i) dchild_ptr[i + 1] += dchild_ptr[i];
    {
        std::vector<int> pos(dchild_ptr.begin(), dchild_ptr.end() - 1);
        for (int i = 1; i < n; ++i)
            if (kept(i)) dchild[pos[pard[i]]++] = i;
    }
    // contracted chains, bottom-up: ch (dense) -> k1 = parent(ch) -> ... -> kt, parent(kt) dense
    T.chain_ptr.assign(1, 0);
    T.chain_nodes.clear();
    T.chain_ch.clear();
    for (int i = 1; i < n; ++i) {
        if (!kept(i) || !pass[T.parent[i]]) continue;
        for (int k = T.parent[i]; pass[k]; k = T.parent[k]) T.chain_nodes.push_back(k);
        T.chain_ptr.push_back((int)T.chain_nodes.size());
        T.chain_ch.push_back(i);
    }
    T.n_chains = (int)T.chain_ch.size();
    // maximal linear subtrees, post-order (children before parents; iterative DFS)
    T.lin_ptr.assign(1, 0);
    T.lin_post.clear();
    T.n_lin_roots = 0;
    for (int r0 = 0; r0 < n; ++r0) {
        if (!T.lin[r0] || (T.parent[r0] >= 0 && T.lin[T.parent[r0]])) continue;
        std::vector<std::pair<int, int>> stack;                  // (node, next child position)
        stack.push_back({r0, T.child_ptr[r0]});
        while (!stack.empty()) {
            const int node = stack.back().first;
            if (stack.back().second < T.child_ptr[node + 1]) {
                const int ch = T.child[stack.back().second++];
                stack.push_back({ch, T.child_ptr[ch]});
            } else {
                T.lin_post.push_back(node);
                stack.pop_back();
            }
        }
        T.lin_ptr.push_back((int)T.lin_post.size());
        ++T.n_lin_roots;
    }
    // post-order of the whole tree (children lists as built above): used by the fundamental power-flow step
    T.all_ptr = {0, n};
    T.all_post.clear();
    {
        std::vector<std::pair<int, int>> stack;
        stack.push_back({0, T.child_ptr[0]});
        while (!stack.empty()) {
            const int node = stack.back().first;
            if (stack.back().second < T.child_ptr[node + 1]) {
                const int ch = T.child[stack.back().second++];
                stack.push_back({ch, T.child_ptr[ch]});
            } else {
                T.all_post.push_back(node);
                stack.pop_back();
            }
        }
    }
    // records of the level-parallel 2x2 kernels
    std::vector<int> lrec, crec, cnode, arec, lbrec, lbptr, lb2rec, lb2x, lb2ptr, lb2cptr, lb2clist;
    {
        std::vector<int> diag0(n, -1), hl(n, 0);
        for (int i = 0; i < n; ++i)
            for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e)
                if (d->col[e] == i) diag0[i] = e;
        int maxhl = -1;
        for (int oi = n - 1; oi >= 0; --oi) {
            const int i = order[oi];
            if (!T.lin[i]) continue;
            maxhl = hl[i] > maxhl ? hl[i] : maxhl;
            const int pp = T.parent[i];
            if (pp >= 0 && T.lin[pp] && hl[i] + 1 > hl[pp]) hl[pp] = hl[i] + 1;
        }
        T.n_lin_heights = maxhl + 1;
        T.lh_ptr.assign(T.n_lin_heights + 1, 0);
        auto put = [&](std::vector<int>& v, int k2) {
            const int a[8] = {k2, diag0[k2], T.parent[k2], k2 > 0 ? e_up[k2] : 0, k2 > 0 ? e_dn[k2] : 0, T.child_ptr[k2],
                              T.child_mid[k2] - T.child_ptr[k2], d->dev_of_bus[k2]};
            v.insert(v.end(), a, a + 8);
        };
        for (int hh = 0; hh < T.n_lin_heights; ++hh) {
            for (int i = 0; i < n; ++i)
                if (T.lin[i] && hl[i] == hh) {
                    put(lrec, i);
                    lrec[lrec.size() - 2] = T.child_ptr[i + 1] - T.child_ptr[i];      // every child of such a bus is linear
                }
            T.lh_ptr[hh + 1] = (int)lrec.size() / 8;
        }
        // bundles of whole all-linear subtrees for the one-launch kernels: subtrees in bus order, a bundle is closed before a height
        // of it would need a second pass of the 256 threads (one thread per (bus, harmonic) of a height)
        T.n_lin_bundles = 0;
        const char* lt_env = h->sw("HPF_LINTREE");
        if (!(lt_env && atoi(lt_env) == 0) && T.n_lin_heights > 1) {
            const int NH = T.n_lin_heights, cap = std::max(1, 256 / d->Hn);
            std::vector<int> root_of(n, -1);
            std::vector<std::vector<int>> sub;                       // buses of each subtree (top-down order)
            std::vector<int> sub_id(n, -1);
            for (int oi = 0; oi < n; ++oi) {
                const int i = order[oi];
                if (!T.lin[i]) continue;
                const int pp = T.parent[i];
                root_of[i] = (pp >= 0 && T.lin[pp]) ? root_of[pp] : i;
                if (root_of[i] == i) {
                    sub_id[i] = (int)sub.size();
                    sub.emplace_back();
                }
                sub[sub_id[root_of[i]]].push_back(i);
            }
            std::vector<std::vector<int>> byh(NH);
            auto flush = [&]() {
                bool any = false;
                for (int hh = 0; hh < NH; ++hh) any = any || !byh[hh].empty();
                if (!any) return;
                for (int hh = 0; hh < NH; ++hh) {
                    lbptr.push_back((int)lbrec.size() / 8);
                    for (int i : byh[hh]) {
                        put(lbrec, i);
                        lbrec[lbrec.size() - 2] = T.child_ptr[i + 1] - T.child_ptr[i];
                    }
                    byh[hh].clear();
                }
                lbptr.push_back((int)lbrec.size() / 8);
                ++T.n_lin_bundles;
            };
            // bundles of the one-round-trip kernels: units of whole subtrees -- a contracted chain with the subtrees hanging off its
            // buses is ONE unit (the chain walk follows its subtrees in the same workgroup) --, at most 256 * NP items (bus, harmonic)
            // and 256 / Hn chains per workgroup
            {
                const char* lb_env2 = h->sw("HPF_LINBUNDLE");
                const char* cb_env = h->sw("HPF_CHAINBUNDLE");
                const bool chb = !(cb_env && atoi(cb_env) == 0) && T.n_chains > 0;
                std::vector<int> node_chain(n, -1);
                for (int r2 = 0; r2 < T.n_chains; ++r2)
                    for (int idx = T.chain_ptr[r2]; idx < T.chain_ptr[r2 + 1]; ++idx) node_chain[T.chain_nodes[idx]] = r2;
                // units: [0, n_chains) the chains (when bundled), then the free subtrees
                std::vector<std::vector<int>> unit_bus(chb ? T.n_chains : 0);
                std::vector<int> unit_chain;
                for (int r2 = 0; chb && r2 < T.n_chains; ++r2) unit_chain.push_back(r2);
                for (const std::vector<int>& sb : sub) {
                    const int pp = T.parent[sb[0]];
                    const int r2 = (chb && pp >= 0) ? node_chain[pp] : -1;
                    if (r2 >= 0) {
                        unit_bus[r2].insert(unit_bus[r2].end(), sb.begin(), sb.end());
                    } else {
                        unit_bus.push_back(sb);
                        unit_chain.push_back(-1);
                    }
                }
                size_t maxunit = 0;
                for (const std::vector<int>& ub : unit_bus) maxunit = std::max(maxunit, ub.size());
                const size_t mi = maxunit * (size_t)d->Hn;
                T.lin_np = (lb_env2 && atoi(lb_env2) == 0) ? 0 : (mi <= 256 ? 1 : (mi <= 512 ? 2 : (mi <= 1024 ? 4 : 0)));
                T.n_lin_bundles2 = 0;
                T.chains_bundled = 0;
                if (T.lin_np) {
                    const size_t cap_items = 256 * (size_t)T.lin_np, cap_chains = (size_t)std::max(1, 256 / d->Hn);
                    std::vector<int> cur, curch, loc(n, -1), csv(n, 0);
                    lb2cptr.assign(1, 0);
                    auto flush2 = [&]() {
                        if (cur.empty() && curch.empty()) return;
                        std::stable_sort(cur.begin(), cur.end(), [&](int a, int b2) { return hl[a] < hl[b2]; });
                        const int base = (int)lb2rec.size() / 8;
                        int csum = 0;
                        for (size_t li = 0; li < cur.size(); ++li) {
                            loc[cur[li]] = (int)li;
                            csv[cur[li]] = csum;
                            csum += T.child_ptr[cur[li] + 1] - T.child_ptr[cur[li]];
                        }
                        int hh = 0;
                        lb2ptr.push_back(base);
                        for (size_t li = 0; li < cur.size(); ++li) {
                            const int i = cur[li];
                            while (hh < hl[i]) {
                                lb2ptr.push_back(base + (int)li);
                                ++hh;
                            }
                            put(lb2rec, i);
                            lb2rec[lb2rec.size() - 3] = csv[i];                                    // cbeg <- first child slot
                            lb2rec[lb2rec.size() - 2] = T.child_ptr[i + 1] - T.child_ptr[i];       // every child is linear
                            const int pp = T.parent[i];
                            int slot = -1, lp = -1;
                            if (pp >= 0 && T.lin[pp]) {
                                int ord = 0;
                                for (int cp = T.child_ptr[pp]; cp < T.child_ptr[pp + 1]; ++cp)
                                    if (T.child[cp] == i) ord = cp - T.child_ptr[pp];
                                slot = csv[pp] + ord;
                                lp = loc[pp];
                            }
                            lb2x.push_back(slot);
                            lb2x.push_back(lp);
                        }
                        while (hh < NH) {
                            lb2ptr.push_back(base + (int)cur.size());
                            ++hh;
                        }
                        lb2clist.insert(lb2clist.end(), curch.begin(), curch.end());
                        lb2cptr.push_back((int)lb2clist.size());
                        cur.clear();
                        curch.clear();
                        ++T.n_lin_bundles2;
                    };
                    for (size_t u = 0; u < unit_bus.size(); ++u) {
                        const bool isch = unit_chain[u] >= 0;
                        if ((cur.size() + unit_bus[u].size()) * (size_t)d->Hn > cap_items || (isch && curch.size() + 1 > cap_chains)) flush2();
                        cur.insert(cur.end(), unit_bus[u].begin(), unit_bus[u].end());
                        if (isch) curch.push_back(unit_chain[u]);
                    }
                    flush2();
                    T.chains_bundled = chb ? 1 : 0;
                    for (int e = 0; e < 8; ++e) lb2rec.push_back(0);           // (a bundle of chains only reads one record slot: keep it inside the array)
                    lb2x.push_back(-1);
                    lb2x.push_back(-1);
                }
            }
            std::vector<int> cnt_h(NH);
            for (const std::vector<int>& sb : sub) {
                std::fill(cnt_h.begin(), cnt_h.end(), 0);
                for (int i : sb) ++cnt_h[hl[i]];
                bool over = false;
                for (int hh = 0; hh < NH; ++hh) over = over || (int)byh[hh].size() + cnt_h[hh] > cap;
                if (over) flush();
                for (int i : sb) byh[hl[i]].push_back(i);
            }
            flush();
        }
        for (int r2 = 0; r2 < T.n_chains; ++r2) {
            const int ch = T.chain_ch[r2];
            const int a[8] = {ch, e_dn[ch], e_up[ch], T.chain_ptr[r2], T.chain_ptr[r2 + 1] - T.chain_ptr[r2], 0, 0, 0};
            crec.insert(crec.end(), a, a + 8);
        }
        for (size_t idx = 0; idx < T.chain_nodes.size(); ++idx) put(cnode, T.chain_nodes[idx]);   // folds its linear subtrees only
        // whole tree by height (fundamental power flow: every bus is a 2x2 bus at harmonic position 0)
        std::vector<int> ha(n, 0);
        int maxha = 0;
        for (int oi = n - 1; oi > 0; --oi) {
            const int i = order[oi], pp = T.parent[i];
            if (ha[i] + 1 > ha[pp]) ha[pp] = ha[i] + 1;
        }
        for (int i = 0; i < n; ++i) maxha = ha[i] > maxha ? ha[i] : maxha;
        T.n_all_heights = maxha + 1;
        T.ah_ptr.assign(T.n_all_heights + 1, 0);
        for (int hh = 0; hh < T.n_all_heights; ++hh) {
            for (int i = 0; i < n; ++i)
                if (ha[i] == hh) {
                    put(arec, i);
                    arec[arec.size() - 2] = T.child_ptr[i + 1] - T.child_ptr[i];
                }
            T.ah_ptr[hh + 1] = (int)arec.size() / 8;
        }
    }
    // node records of the multi-wave kernels (hpf_quad.hpp)
    std::vector<int> fdesc((size_t)T.n_dense * FDESC, 0), child3((size_t)(n > 1 ? n - 1 : 0) * 4, 0), bdesc((size_t)T.n_dense * 4, 0);
    for (size_t cp = 0; cp < T.child.size(); ++cp) {
        const int ch = T.child[cp];
        child3[cp * 4 + 0] = ch;
        child3[cp * 4 + 1] = e_dn[ch];
        child3[cp * 4 + 2] = e_up[ch];
    }
    // ---- constant-inverse leaves (see Tree::d_Minv) -----------------------------------------------------------------------
    std::vector<int> cleaf_of(n, -1);
    std::vector<double> minv;
    T.n_cleaf = 0;
    // lazy leaves (Tree::d_lzrec): complex constants of leaf c under parent k, harmonics q, p:
    //   C0[q][p] = y_kc[q] Ahh^-1[q][p] y_ck[p] (q, p >= 1),  Gc[q] = y_kc[q] [1; Lc][q],  Hr[p] = [1 Lr][p] y_ck[p]
    struct LazyLeaf {
        int k;
        std::vector<std::complex<double>> C0, G, H;
    };
    std::vector<LazyLeaf> lazies;
    // super-leaves: nonlinear buses whose dense children are all lazy leaves -- bordered low-rank inverse instead of Gauss-Jordan
    std::vector<int> sl_slot(n, 0);                            // 1 + slot of the bus's Z0 image in Tree::d_Minv
    std::vector<long long> sl_off(n, -1);                      // offset of [Tc | Pb | Qb] in slimg
    std::vector<double> slimg, lbimg;
    const char* sl_env = h->sw("HPF_SLEAF");
    const bool sleaf_on = !(sl_env && atoi(sl_env) == 0) && wave_block_size(b) <= 52;   // HPF_SLEAF=0: every bus with dense children takes the Gauss-Jordan path
                                                                                       // (b > 52: plain constant-inverse leaves only, no lazy leaves / super-leaves yet)
    const int sleaf_mode = sl_env ? atoi(sl_env) : 2;          // 1: nonlinear buses only
    int n_sleaf = 0;
    const char* sb_env = h->sw("HPF_SLBACK");
    const char* lb_env = h->sw("HPF_LEAFBATCH");
    const bool slback_on = !(sb_env && atoi(sb_env) == 0) && !(lb_env && atoi(lb_env) == 0);   // super-leaves keep T^-1 only; k_sleaf_back_batch
                                                                                             // rebuilds D^-1 t (needs the batched back sweep)
    std::vector<int> sb_ord(n, -1), sb_m(n, 0);
    std::vector<int> sl_nest(n, 0);                            // nesting order of a bordered bus: 0 = its dense children are leaves only
    // super-leaves whose parent rebuilds their Schur complement itself (like a lazy leaf's, with the m x m core T^-1): constants
    //   C0 = g Ahh^-1 h,  GP = g Pb (rows q >= 1),  QH = Qb h (columns q >= 1)
    struct LazySuper {
        int k, m1;
        std::vector<std::complex<double>> C0, GP, QH;
    };
    std::vector<LazySuper> slzs;
    std::vector<std::vector<int>> slz_of(n);
    const char* sz_env = h->sw("HPF_SLLAZY");
    const bool sllazy_on = !(sz_env && atoi(sz_env) == 0);     // HPF_SLLAZY=0: every super-leaf pushes its Schur complement itself
    const char* sn_env = h->sw("HPF_SLNEST");
    const bool slnest_on = !(sn_env && atoi(sn_env) == 0);     // HPF_SLNEST=0: bordered buses below a bordered bus stay on the Gauss-Jordan path
    std::vector<double> sbimg;
    std::vector<std::vector<int>> lazy_of(n);
    std::vector<int> is_lazy(n, 0);
    const char* lz_env = h->sw("HPF_LAZY");
    const bool lazy_on = !(lz_env && atoi(lz_env) == 0) && wave_block_size(b) <= 52;
    const int lazy_mode = lz_env ? atoi(lz_env) : 2;          // 1: only leaves hanging directly under their dense parent
    constexpr int LZ_MAX = 4;
    const int BWc = wave_block_size(b);
    // host twin of TileIO<B>::off (hpf_quad.hpp): offset of (row, col) in a tile image, -1 for the columns that are not stored
    auto tile_off = [&](int row, int col) -> long long {
        const int NTq = (BWc + 16) / 16, LW = (BWc + 1 - 16 * (NTq - 1)) <= 8 ? 8 : 16, RG = (NTq - 1) * 64 + 4 * LW;
        const int NP = (BWc / 4) / 2;
        const int tr = row >> 4, reg = (row & 15) >> 2, lg = row & 3, wv = col >> 4, jj = col & 15, e = tr * 4 + reg;
        if (row >= BWc || (wv == NTq - 1 && jj >= LW)) return -1;
        if (e < 2 * NP)
            return (long long)(e >> 1) * 2 * RG + (wv < NTq - 1 ? wv * 128 + (lg * 16 + jj) * 2 : (NTq - 1) * 128 + (lg * LW + jj) * 2) + (e & 1);
        return (long long)NP * 2 * RG + (wv < NTq - 1 ? wv * 64 + lg * 16 + jj : (NTq - 1) * 64 + lg * LW + jj);
    };
    std::vector<int> toff_tab((size_t)b * b, -1);                // tile_off of every (row, col) of a block, once
    if (BWc)
        for (int row = 0; row < b; ++row)
            for (int col = 0; col < b; ++col) toff_tab[(size_t)row * b + col] = (int)tile_off(row, col);
    T.toff_tab = toff_tab;
    if (contract && d->coupled && BWc) {
        typedef std::complex<double> cd;
        const int Hn = d->Hn, nnz = d->nnz;
        auto yv = [&](int q, int e) { return cd(d->Yval[((size_t)q * nnz + e) * 2], d->Yval[((size_t)q * nnz + e) * 2 + 1]); };
        std::vector<int> diag(n, -1);
        for (int i = 0; i < n; ++i)
            for (int e = d->rowptr[i]; e < d->rowptr[i + 1]; ++e)
                if (d->col[e] == i) diag[i] = e;
        // series elimination of the all-linear subtrees at the harmonics q >= 1 (current rows: complex-linear, constant):
        // dl[q][i] = y_ii - sum_children y_ic y_ci / dl[q][c]
        std::vector<cd> dl((size_t)Hn * n, cd(0.0, 0.0));
        for (int q = 1; q < Hn; ++q)
            for (size_t idx = 0; idx < T.lin_post.size(); ++idx) {          // children before parents
                const int i = T.lin_post[idx];
                cd v = yv(q, diag[i]);
                for (int cp = T.child_ptr[i]; cp < T.child_ptr[i + 1]; ++cp) {
                    const int g = T.child[cp];
                    v -= yv(q, e_dn[g]) * yv(q, e_up[g]) / dl[(size_t)q * n + g];
                }
                dl[(size_t)q * n + i] = v;
            }
        std::vector<int> chain_of(n, -1), chain_top(n, -1);
        for (int r2 = 0; r2 < T.n_chains; ++r2) {
            chain_of[T.chain_ch[r2]] = r2;
            chain_top[T.chain_nodes[T.chain_ptr[r2 + 1] - 1]] = r2;
        }
        const int NTc = (BWc + 16) / 16;
        const size_t CTc = (size_t)NTc * NTc * 256;
        // constant complex block of a nonlinear bus k in rectangular coordinates and its effective couplings
        auto build_Yc = [&](int k, std::vector<cd>& Yc, std::vector<cd>& geff, std::vector<cd>& heff) {
            Yc.assign((size_t)Hn * Hn, cd(0.0, 0.0));
            // effective coupling with the dense parent at the harmonics q >= 1: the line itself, or what the contracted chain
            // in between leaves of it (constant there: the chain buses are linear, their current rows complex-linear)
            geff.assign(Hn, cd(0.0, 0.0));                   // A'(parent, k), A'(k, parent)
            heff.assign(Hn, cd(0.0, 0.0));
            for (int q = 0; q < Hn; ++q) {
                geff[q] = yv(q, e_dn[k]);
                heff[q] = yv(q, e_up[k]);
            }
            const bool linear_k = k < d->m;                    // PQ bus: no Norton term; its fundamental (power rows) is state dependent
            if (!linear_k) {
                const double* yn = d->Y_N + (size_t)d->dev_of_bus[k] * Hn * Hn * 2;
                for (int q = 0; q < Hn; ++q)
                    for (int p2 = 0; p2 < Hn; ++p2) Yc[(size_t)q * Hn + p2] = -cd(yn[((size_t)q * Hn + p2) * 2], yn[((size_t)q * Hn + p2) * 2 + 1]);
            }
            for (int q = linear_k ? 1 : 0; q < Hn; ++q) {
                cd v = yv(q, diag[k]);
                if (q >= 1) {
                    for (int cp = T.child_ptr[k]; cp < T.child_mid[k]; ++cp) {         // linear subtrees below k
                        const int g = T.child[cp];
                        if (!pass[g]) {
                            v -= yv(q, e_dn[g]) * yv(q, e_up[g]) / dl[(size_t)q * n + g];
                            continue;
                        }
                        // top of a contracted chain below k (only at buses with dense children: super-leaves): what the chain's
                        // elimination leaves on k's diagonal -- independent of the dense bus at its lower end, which comes later
                        const int r3 = chain_top[g];
                        cd carry(0.0, 0.0);
                        for (int idx = T.chain_ptr[r3]; idx < T.chain_ptr[r3 + 1]; ++idx) {
                            const int kk = T.chain_nodes[idx];
                            cd dk = yv(q, diag[kk]) + carry;
                            for (int cp2 = T.child_ptr[kk]; cp2 < T.child_mid[kk]; ++cp2) {
                                const int g2 = T.child[cp2];
                                dk -= yv(q, e_dn[g2]) * yv(q, e_up[g2]) / dl[(size_t)q * n + g2];
                            }
                            carry = -yv(q, e_dn[kk]) * yv(q, e_up[kk]) / dk;
                        }
                        v += carry;
                    }
                    if (chain_of[k] >= 0) {                                              // contracted chain above k (k_chain_factor)
                        const int r2 = chain_of[k];
                        cd a_kc = yv(q, e_dn[k]), a_ck = yv(q, e_up[k]), carry(0.0, 0.0), dD(0.0, 0.0);
                        for (int idx = T.chain_ptr[r2]; idx < T.chain_ptr[r2 + 1]; ++idx) {
                            const int kk = T.chain_nodes[idx];
                            cd dk = yv(q, diag[kk]) + carry;
                            for (int cp = T.child_ptr[kk]; cp < T.child_mid[kk]; ++cp) {
                                const int g = T.child[cp];
                                dk -= yv(q, e_dn[g]) * yv(q, e_up[g]) / dl[(size_t)q * n + g];
                            }
                            const cd a_ku = yv(q, e_up[kk]), a_uk = yv(q, e_dn[kk]);
                            dD -= a_ck * a_kc / dk;
                            const cd n_ck = -a_ck * a_ku / dk, n_kc = -a_uk * a_kc / dk;
                            carry = -a_uk * a_ku / dk;
                            a_ck = n_ck;
                            a_kc = n_kc;
                        }
                        v += dD;
                        geff[q] = a_kc;
                        heff[q] = a_ck;
                    }
                }
                Yc[(size_t)q * Hn + q] += v;
            }
        };
        auto border_image = [&](const std::vector<cd>& Yc, std::vector<cd>& img) -> bool {
            const int Hh = Hn - 1;
            // Bordered form around the fundamental (index 0), the only place where a state-dependent 2x2 term D enters:
            //     [a00 + D  A0h]^-1        [0   0    ]   [  I   ]                     [        ]
            //     [Ah0      Ahh]      =    [0  Ahh^-1] + [ Lc_h ] (c0 + D)^-1  [ I  Lr_h ],   Lc_h = -Ahh^-1 Ah0,  Lr_h = -A0h Ahh^-1,
            // c0 = a00 - A0h Ahh^-1 Ah0.  Everything but D is constant: keep R(Ahh^-1), R(Lc_h), R(Lr_h), R(c0) as ONE b x b image
            // (additive rank-2 update on the device: no cancellation, unlike a Woodbury correction of the full inverse).
            // (in-place complex Gauss-Jordan with partial pivoting in plain re / im arithmetic: std::complex products go through the
            //  checked library routine, several times slower -- this inversion is what the set-up of a 10 000-bus model spends its time in)
            std::vector<double> ar((size_t)Hh * Hh), ai((size_t)Hh * Hh);
            std::vector<int> pv(Hh > 0 ? Hh : 1, 0);
            for (int q = 0; q < Hh; ++q)
                for (int p2 = 0; p2 < Hh; ++p2) {
                    ar[(size_t)q * Hh + p2] = Yc[(size_t)(q + 1) * Hn + p2 + 1].real();
                    ai[(size_t)q * Hh + p2] = Yc[(size_t)(q + 1) * Hn + p2 + 1].imag();
                }
            bool ok = true;
            for (int col = 0; col < Hh && ok; ++col) {
                int piv = col;
                double best = -1.0;
                for (int r2 = col; r2 < Hh; ++r2) {
                    const double xr = ar[(size_t)r2 * Hh + col], xi = ai[(size_t)r2 * Hh + col], mg = xr * xr + xi * xi;
                    if (mg > best) {
                        best = mg;
                        piv = r2;
                    }
                }
                if (best == 0.0 || !(best == best)) {
                    ok = false;
                    break;
                }
                pv[col] = piv;
                double* __restrict__ cr = &ar[(size_t)col * Hh];
                double* __restrict__ ci = &ai[(size_t)col * Hh];
                if (piv != col) {
                    double* __restrict__ qr = &ar[(size_t)piv * Hh];
                    double* __restrict__ qi = &ai[(size_t)piv * Hh];
                    for (int c2 = 0; c2 < Hh; ++c2) {
                        std::swap(cr[c2], qr[c2]);
                        std::swap(ci[c2], qi[c2]);
                    }
                }
                const cd ip = cd(1.0, 0.0) / cd(cr[col], ci[col]);
                const double pr = ip.real(), pi = ip.imag();
                cr[col] = 1.0;
                ci[col] = 0.0;
                for (int c2 = 0; c2 < Hh; ++c2) {
                    const double xr = cr[c2], xi = ci[c2];
                    cr[c2] = xr * pr - xi * pi;
                    ci[c2] = xr * pi + xi * pr;
                }
                for (int r2 = 0; r2 < Hh; ++r2) {
                    if (r2 == col) continue;
                    double* __restrict__ rr = &ar[(size_t)r2 * Hh];
                    double* __restrict__ ri = &ai[(size_t)r2 * Hh];
                    const double fr = rr[col], fi = ri[col];
                    if (fr == 0.0 && fi == 0.0) continue;
                    rr[col] = 0.0;
                    ri[col] = 0.0;
                    cplx_row_axpy(rr, ri, cr, ci, fr, fi, Hh);
                }
            }
            std::vector<cd> Bh((size_t)Hh * Hh);
            if (ok) {
                for (int col = Hh - 1; col >= 0; --col)               // undo the row exchanges: columns of the inverse, in reverse
                    if (pv[col] != col)
                        for (int r2 = 0; r2 < Hh; ++r2) {
                            std::swap(ar[(size_t)r2 * Hh + col], ar[(size_t)r2 * Hh + pv[col]]);
                            std::swap(ai[(size_t)r2 * Hh + col], ai[(size_t)r2 * Hh + pv[col]]);
                        }
                for (size_t e = 0; e < Bh.size(); ++e) Bh[e] = cd(ar[e], ai[e]);
            }
            if (!ok) return false;                                   // singular constant part: leave the bus on the general path
            img.assign((size_t)Hn * Hn, cd(0.0, 0.0));               // complex image: [c0 Lr_h; Lc_h Ahh^-1]
            cd c0 = Yc[0];
            for (int q = 0; q < Hh; ++q) {
                cd lc(0.0, 0.0), lr(0.0, 0.0);
                for (int p2 = 0; p2 < Hh; ++p2) {
                    lc -= Bh[(size_t)q * Hh + p2] * Yc[(size_t)(p2 + 1) * Hn];          // -(Ahh^-1 Ah0)[q]
                    lr -= Yc[p2 + 1] * Bh[(size_t)p2 * Hh + q];                          // -(A0h Ahh^-1)[q]
                }
                img[(size_t)(q + 1) * Hn] = lc;
                img[q + 1] = lr;
                for (int p2 = 0; p2 < Hh; ++p2) img[(size_t)(q + 1) * Hn + p2 + 1] = Bh[(size_t)q * Hh + p2];
            }
            for (int p2 = 0; p2 < Hh; ++p2) c0 += Yc[p2 + 1] * img[(size_t)(p2 + 1) * Hn];   // a00 + A0h Lc_h
            img[0] = c0;
            return true;
        };
        // the leaves' constant images (one complex (Hn-1) x (Hn-1) inversion each) are independent of each other: host threads
        struct LeafPre {
            std::vector<cd> Yc, geff, heff, img;
            bool ok = false;
        };
        std::vector<int> leaf_pos(n, -1);
        std::vector<LeafPre> pre;
        {
            std::vector<int> cand;
            for (int pos = 0; pos < T.n_dense; ++pos) {
                const int k = T.lvl_nodes[pos];
                if (k < d->m || dchild_ptr[k + 1] != dchild_ptr[k] || d->dev_of_bus[k] < 0) continue;
                if (use_fp && fmask[k]) continue;
                leaf_pos[k] = (int)cand.size();
                cand.push_back(k);
            }
            pre.resize(cand.size());
            unsigned nth = host_threads();
            if ((size_t)nth > cand.size() / 8 + 1) nth = (unsigned)(cand.size() / 8 + 1);
            auto work = [&](unsigned t0) {
                for (size_t ci = t0; ci < cand.size(); ci += nth) {
                    LeafPre& lp = pre[ci];
                    build_Yc(cand[ci], lp.Yc, lp.geff, lp.heff);
                    lp.ok = border_image(lp.Yc, lp.img);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t0 = 1; t0 < nth; ++t0) pool.emplace_back(work, t0);
            work(0);
            for (std::thread& th : pool) th.join();
        }
        // slots in elimination order, then the images of all leaves at once (host threads again: 100 KB of image per leaf at b = 100)
        std::vector<int> leaf_list;
        for (int pos = 0; pos < T.n_dense; ++pos) {
            const int k = T.lvl_nodes[pos];
            if (leaf_pos[k] < 0 || !pre[leaf_pos[k]].ok) continue;
            cleaf_of[k] = T.n_cleaf++;
            leaf_list.push_back(k);
        }
        minv.assign((size_t)T.n_cleaf * CTc, 0.0);
        const int NTRl = (BWc + 15) / 16, KSl = (BWc + 3) / 4, SZl = NTRl * KSl * 64 + 2 * BWc + 4;
        if (BWc <= 52) lbimg.assign((size_t)T.n_cleaf * SZl, 0.0);
        auto fill_leaf = [&](int k) {
            const std::vector<cd>& img = pre[leaf_pos[k]].img;
            double* Mt = &minv[(size_t)cleaf_of[k] * CTc];
            for (int row = 0; row < b; ++row)                          // (rows / columns beyond b: zeros)
                for (int col = 0; col < b; ++col) {
                    const cd z = img[(size_t)(row >> 1) * Hn + (col >> 1)];
                    const int t = row & 1, t2 = col & 1;              // R(z) = [re -im; im re]
                    const int o = toff_tab[(size_t)row * b + col];
                    if (o >= 0) Mt[o] = (t == t2) ? z.real() : (t ? z.imag() : -z.imag());
                }
            if (BWc <= 52) {   // the same constants for k_leaf_batch: [0 Lr; 0 Ahh^-1] in MFMA A-operand layout, R(Lc), R(c0)
                const int NTR = NTRl, KS = KSl, SZ = SZl;
                double* L = &lbimg[(size_t)cleaf_of[k] * SZ];
                auto Rz = [](cd z, int t, int t2) { return (t == t2) ? z.real() : (t ? z.imag() : -z.imag()); };
                for (int w2 = 0; w2 < NTR; ++w2)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lg = 0; lg < 4; ++lg)
                            for (int jj = 0; jj < 16; ++jj) {
                                const int row = 16 * w2 + jj, col = 4 * ks + lg;
                                double v = 0.0;
                                if (row < b && col < b && col >= 2) v = Rz(img[(size_t)(row >> 1) * Hn + (col >> 1)], row & 1, col & 1);
                                L[((size_t)w2 * KS + ks) * 64 + lg * 16 + jj] = v;
                            }
                double* Lc = L + (size_t)NTR * KS * 64;
                Lc[0] = 1.0; Lc[1] = 0.0; Lc[2] = 0.0; Lc[3] = 1.0;
                for (int row = 2; row < b; ++row)
                    for (int a2 = 0; a2 < 2; ++a2) Lc[row * 2 + a2] = Rz(img[(size_t)(row >> 1) * Hn], row & 1, a2);
                double* C0 = Lc + 2 * BWc;
                C0[0] = img[0].real(); C0[1] = -img[0].imag(); C0[2] = img[0].imag(); C0[3] = img[0].real();
            }
        };
        {
            unsigned nth = host_threads();
            if ((size_t)nth > leaf_list.size() / 8 + 1) nth = (unsigned)(leaf_list.size() / 8 + 1);
            auto work = [&](unsigned t0) {
                for (size_t li = t0; li < leaf_list.size(); li += nth) fill_leaf(leaf_list[li]);
            };
            std::vector<std::thread> pool;
            for (unsigned t0 = 1; t0 < nth; ++t0) pool.emplace_back(work, t0);
            work(0);
            for (std::thread& th : pool) th.join();
        }
        for (int k : leaf_list) {
            const std::vector<cd>&geff = pre[leaf_pos[k]].geff, &heff = pre[leaf_pos[k]].heff, &img = pre[leaf_pos[k]].img;
            const int pk = pard[k];                                   // dense parent, directly or through a contracted chain
            const bool direct = chain_of[k] < 0;
            if (lazy_on && (direct || lazy_mode >= 2) && pk >= (d->c > 1 ? d->c : 1) && (int)lazy_of[pk].size() < LZ_MAX) {
                LazyLeaf ll;
                ll.k = k;
                ll.C0.assign((size_t)Hn * Hn, cd(0.0, 0.0));
                ll.G.assign(Hn, cd(0.0, 0.0));
                ll.H.assign(Hn, cd(0.0, 0.0));
                // (harmonic position 0 of the borders is state dependent -- power rows of a PQ parent, chain buses -- and comes
                //  from the leaf per scenario: G0 S_c^-1 and H0 S_k^-1 next to its 2x2 core)
                for (int q = 1; q < Hn; ++q) {
                    ll.G[q] = geff[q] * img[(size_t)q * Hn];
                    ll.H[q] = img[q] * heff[q];
                    for (int p2 = 1; p2 < Hn; ++p2) ll.C0[(size_t)q * Hn + p2] = geff[q] * img[(size_t)q * Hn + p2] * heff[p2];
                }
                lazy_of[pk].push_back((int)lazies.size());
                lazies.push_back(std::move(ll));
                is_lazy[k] = 1;
            }
        }
        // ---- super-leaves: M_k = A_k - sum_c Gc_c K_c Hr_c + E0 D_k E0^T,  A_k = Yc_k - sum_c C0_c.  Border the
        //      system with z_c = K_c Hr_c x and eliminate the harmonic part of x with the constant Ahh_k^-1:
        //          M_k^-1 = [0 0; 0 Ahh^-1] + Pb T^-1 Qb,    T = Tc + blockdiag(D_k, K_1^-1, ..., K_L^-1) - (G0/H0 borders),
        //      Tc, Pb (b x m), Qb (m x b), m = 2 + 2L constant per model; T is m x m per scenario (k_factor_q, "sleaf" branch).
        //      Nested (round 2): a dense child may itself be a vector-only bordered bus c (M_c^-1 = Z0_c + Pb_c T_c^-1 Qb_c, border m_c): its
        //      Schur complement onto k is  C0_c + (g Pb_c) T_c^-1 (Qb_c h)  -- the same form with the m_c x m_c per-scenario matrix T_c in the
        //      place of K_c^-1 and its m_c/2 complex border columns (g Pb_c[:, i], Qb_c[i, :] h) in the place of the leaf's one -- so k
        //      borders ITS system with all of them: m_k = 2 + 2 L + sum m_c, T_k carries the children's T_c (not their inverses) on its
        //      diagonal.  Buses qualify bottom-up while m_k <= 10 (k_sleaf_batch: one thread per border row, 10 x 10 in LDS).
        for (int pos = 0; sleaf_on && pos < T.n_dense; ++pos) {
            const int k = T.lvl_nodes[pos];
            const int L = (int)lazy_of[k].size();
            const int LS = (int)slz_of[k].size();                      // vector-only bordered children (registered when THEY were built)
            if (k < (d->c > 1 ? d->c : 1) || L + LS == 0 || dchild_ptr[k + 1] - dchild_ptr[k] != L + LS) continue;
            if (use_fp && fmask[k]) continue;
            if (k < d->m && sleaf_mode < 2) continue;                  // linear (PQ) buses: power-row map W_k on the fundamental
            // border columns: per lazy leaf one complex column (G, H), per bordered child its m1_c columns (g Pb_c[:, i], Qb_c[i, :] h)
            std::vector<std::vector<cd>> colG, colH;
            for (int li : lazy_of[k]) {
                colG.push_back(lazies[li].G);
                colH.push_back(lazies[li].H);
            }
            for (int zi : slz_of[k]) {
                const int m1c = slzs[zi].m1;
                for (int i = 0; i < m1c; ++i) {
                    std::vector<cd> g(Hn, cd(0.0, 0.0)), hh(Hn, cd(0.0, 0.0));
                    for (int q = 1; q < Hn; ++q) {
                        g[q] = slzs[zi].GP[(size_t)q * m1c + i];
                        hh[q] = slzs[zi].QH[(size_t)i * Hn + q];
                    }
                    colG.push_back(std::move(g));
                    colH.push_back(std::move(hh));
                }
            }
            for (int zi : slz_of[k]) sl_nest[k] = std::max(sl_nest[k], 1 + sl_nest[slzs[zi].k]);
            const int NC = (int)colG.size();
            const int m1 = 1 + NC, mr = 2 * m1;
            const int pks = pard[k];
            const bool can_slz = slback_on && sllazy_on && mr <= 10 && pks >= (d->c > 1 ? d->c : 1) && (int)slz_of[pks].size() < 2;
            if (mr > 10 || (LS > 0 && !(can_slz && slnest_on))) continue;            // (a nested bus exists only in the scenario-batched, vector-only form)
            std::vector<cd> A, geff, heff, imgk;
            build_Yc(k, A, geff, heff);
            for (int li : lazy_of[k])
                for (int q = 1; q < Hn; ++q)
                    for (int p2 = 1; p2 < Hn; ++p2) A[(size_t)q * Hn + p2] -= lazies[li].C0[(size_t)q * Hn + p2];
            for (int zi : slz_of[k])
                for (int q = 1; q < Hn; ++q)
                    for (int p2 = 1; p2 < Hn; ++p2) A[(size_t)q * Hn + p2] -= slzs[zi].C0[(size_t)q * Hn + p2];
            if (!border_image(A, imgk)) continue;
            auto Ainv = [&](int q, int p2) { return imgk[(size_t)q * Hn + p2]; };          // q, p2 >= 1
            std::vector<cd> Tc((size_t)m1 * m1, cd(0.0, 0.0)), Pb((size_t)Hn * m1, cd(0.0, 0.0)), Qb((size_t)m1 * Hn, cd(0.0, 0.0));
            Tc[0] = imgk[0];
            Pb[0] = cd(1.0, 0.0);
            Qb[0] = cd(1.0, 0.0);
            for (int q = 1; q < Hn; ++q) {
                Pb[(size_t)q * m1] = imgk[(size_t)q * Hn];                                    // Lc
                Qb[q] = imgk[q];                                                              // Lr
            }
            for (int i = 0; i < NC; ++i) {
                const std::vector<cd>& Gi = colG[i];
                const std::vector<cd>& Hi = colH[i];
                cd t0c(0.0, 0.0), tc0(0.0, 0.0);
                for (int q = 1; q < Hn; ++q) {
                    t0c -= imgk[q] * Gi[q];                                                   // -Lr gh_c
                    tc0 -= Hi[q] * imgk[(size_t)q * Hn];                                      // -hh_c Lc
                    cd pbv(0.0, 0.0), qbv(0.0, 0.0);
                    for (int p2 = 1; p2 < Hn; ++p2) {
                        pbv += Ainv(q, p2) * Gi[p2];                                          // Ahh^-1 gh_c
                        qbv += Hi[p2] * Ainv(p2, q);                                          // hh_c Ahh^-1
                    }
                    Pb[(size_t)q * m1 + 1 + i] = pbv;
                    Qb[(size_t)(1 + i) * Hn + q] = qbv;
                }
                Tc[1 + i] = t0c;
                Tc[(size_t)(1 + i) * m1] = tc0;
            }
            for (int i = 0; i < NC; ++i)
                for (int j = 0; j < NC; ++j) {
                    cd v(0.0, 0.0);
                    for (int q = 1; q < Hn; ++q) v -= Qb[(size_t)(1 + i) * Hn + q] * colG[j][q];   // -hh_i Ahh^-1 gh_j
                    Tc[(size_t)(1 + i) * m1 + 1 + j] = v;
                }
            auto R = [](cd z, int t, int t2) { return (t == t2) ? z.real() : (t ? z.imag() : -z.imag()); };
            sl_slot[k] = ++T.n_cleaf;                                                         // Z0 image: [0 0; 0 Ahh^-1]
            minv.resize((size_t)T.n_cleaf * CTc, 0.0);
            double* Mt = &minv[(size_t)(T.n_cleaf - 1) * CTc];
            for (int row = 2; row < b; ++row)
                for (int col = 2; col < b; ++col) {
                    const long long o = tile_off(row, col);
                    if (o >= 0) Mt[o] = R(Ainv(row >> 1, col >> 1), row & 1, col & 1);
                }
            sl_off[k] = (long long)slimg.size();
            for (int r2 = 0; r2 < mr; ++r2)
                for (int c2 = 0; c2 < mr; ++c2) slimg.push_back(R(Tc[(size_t)(r2 >> 1) * m1 + (c2 >> 1)], r2 & 1, c2 & 1));
            for (int row = 0; row < b; ++row)
                for (int c2 = 0; c2 < mr; ++c2) slimg.push_back(R(Pb[(size_t)(row >> 1) * m1 + (c2 >> 1)], row & 1, c2 & 1));
            for (int r2 = 0; r2 < mr; ++r2)
                for (int col = 0; col < b; ++col) slimg.push_back(R(Qb[(size_t)(r2 >> 1) * Hn + (col >> 1)], r2 & 1, col & 1));
            if (can_slz) {
                LazySuper z;
                z.k = k;
                z.m1 = m1;
                z.C0.assign((size_t)Hn * Hn, cd(0.0, 0.0));
                z.GP.assign((size_t)Hn * m1, cd(0.0, 0.0));
                z.QH.assign((size_t)m1 * Hn, cd(0.0, 0.0));
                for (int q = 1; q < Hn; ++q) {
                    for (int p2 = 1; p2 < Hn; ++p2) z.C0[(size_t)q * Hn + p2] = geff[q] * Ainv(q, p2) * heff[p2];
                    for (int i = 0; i < m1; ++i) {
                        z.GP[(size_t)q * m1 + i] = geff[q] * Pb[(size_t)q * m1 + i];
                        z.QH[(size_t)i * Hn + q] = Qb[(size_t)i * Hn + q] * heff[q];
                    }
                }
                slz_of[pks].push_back((int)slzs.size());
                slzs.push_back(std::move(z));
                is_lazy[k] = 1;
            }
            if (slback_on) {                                                               // [0 0; 0 Ahh^-1] in MFMA A-operand layout
                const int NTR = (BWc + 15) / 16, KS = (BWc + 3) / 4;
                sb_ord[k] = n_sleaf;
                sb_m[k] = mr;
                // slot (SleafImg<B>): [NTR][KS][64] the image (+ the rows of Qb in the padding rows BW.. of the last row tile where
                // they fit: r = Qb v then falls out of the same MFMAs) | [NTR][3][64] Pb as a second A operand (x += Pb y: 3 rank-4 steps)
                const bool qb_rows = 16 * NTR - BWc >= 10;
                const size_t o0 = sbimg.size(), o1 = o0 + (size_t)NTR * KS * 64;
                sbimg.resize(o1 + (size_t)NTR * 3 * 64, 0.0);
                for (int w2 = 0; w2 < NTR; ++w2)
                    for (int lg = 0; lg < 4; ++lg)
                        for (int jj = 0; jj < 16; ++jj) {
                            const int row = 16 * w2 + jj;
                            for (int ks = 0; ks < KS; ++ks) {
                                const int col = 4 * ks + lg;
                                double v = 0.0;
                                if (row >= 2 && col >= 2 && row < b && col < b) v = R(Ainv(row >> 1, col >> 1), row & 1, col & 1);
                                if (qb_rows && row >= BWc && row - BWc < mr && col < b)
                                    v = R(Qb[(size_t)((row - BWc) >> 1) * Hn + (col >> 1)], (row - BWc) & 1, col & 1);
                                sbimg[o0 + ((size_t)w2 * KS + ks) * 64 + lg * 16 + jj] = v;
                            }
                            for (int kp = 0; kp < 3; ++kp) {
                                const int c2 = 4 * kp + lg;
                                if (row < b && c2 < mr)
                                    sbimg[o1 + ((size_t)w2 * 3 + kp) * 64 + lg * 16 + jj] = R(Pb[(size_t)(row >> 1) * m1 + (c2 >> 1)], row & 1, c2 & 1);
                            }
                        }
            }
            ++n_sleaf;
        }
    }
    // ---- compress steps on the Gauss-Jordan skeleton -------------------------------------------------------------
    // Leaf-first elimination has as many dependent levels as the dense tree is high, and every level costs one workgroup life whatever
    // its width.  Parallel tree contraction shortens the chain: a Gauss-Jordan bus v whose tallest dense child c is alone on v's
    // critical path is eliminated BEFORE c, as soon as its other children are done.  Its elimination pushes onto both neighbours
    // (parent p and c) and leaves the dense fill A'(p,c) = -A(p,v) D_v^-1 A(v,c), A'(c,p) = -A(c,v) D_v^-1 A(v,p): c then hangs under p
    // with a dense coupling pair (one rank-b product pair on the matrix cores when it is eliminated).  One round: a pending child is
    // not compressed itself (its pushes would be products of dense blocks).  Back sweep: x_p -> x_c -> x_v.
    std::vector<int> comp_role(n, 0), comp_idx(n, -1), comp_child(n, -1);
    T.n_comp = 0;
    T.comp_v.clear();
    T.comp_c.clear();
    {
        const char* cp_env = h->sw("HPF_COMPRESS");
        // A compress step trades a shorter chain of levels for more matrix-core work (the dense push of the pending child): it pays while the
        // levels do not fill the chip (crossover between 256 and 384 live scenarios on the headline feeder: +5..8 % per step at 384..1 024,
        // tools/groups_sweep.py).  The steps are nevertheless the default at EVERY capacity (round 5; rounds 3-4: up to 256 scenarios only), so
        // that the Newton steps -- and with them the iteration count of a solver-sensitive case -- do not depend on the capacity a handle was
        // created with; a sweep of several hundred live scenarios that wants the last 5 % passes HPF_COMPRESS=0 (hpf_create_opts).
        const bool compress_on = contract && d->coupled && BWc != 0 && BWc <= 100 && T.n_dense > 2 && (cp_env ? atoi(cp_env) != 0 : true);
        auto is_gj = [&](int k2) { return kept(k2) && cleaf_of[k2] < 0 && sl_off[k2] < 0; };
        std::vector<int> cc(n, -1), isc(n, 0), keptl;
        for (int i = 0; i < n; ++i)
            if (kept(i)) keptl.push_back(i);
        std::vector<int> gjb(n, 0);
        for (int k2 : keptl) gjb[k2] = is_gj(k2) ? 1 : 0;
        if (compress_on) {
            // choice of the steps: bottom-up over the dense tree, up[k] = the level at which everything k's parent waits for on k's
            // side is done -- k eliminated leaves first: max(children) + 1; k compressed with pending child c: the level of c, which
            // waits for its own children and for k, while k only waits for its OTHER children -- take the smaller; top-down a pending
            // child is forced to the leaf-first form (one round)
            std::vector<int> byh(keptl), up(n, 0), nrm(n, 0), bestc(n, -1), forced(n, 0);
            std::stable_sort(byh.begin(), byh.end(), [&](int a, int b2) { return height[a] < height[b2]; });
            for (int k2 : byh) {
                int m1 = -1, m2 = -1, a1 = -1;
                for (int i = dchild_ptr[k2]; i < dchild_ptr[k2 + 1]; ++i) {
                    const int u = up[dchild[i]];
                    if (u > m1) {
                        m2 = m1;
                        m1 = u;
                        a1 = dchild[i];
                    } else if (u > m2) {
                        m2 = u;
                    }
                }
                nrm[k2] = std::max(gjb[k2], m1 + 1);
                int best = nrm[k2], bc = -1;
                if (gjb[k2] && pard[k2] >= 0 && gjb[pard[k2]] && !(use_fp && fmask[k2]))
                    for (int i = dchild_ptr[k2]; i < dchild_ptr[k2 + 1]; ++i) {
                        const int c1 = dchild[i];
                        if (!gjb[c1] || (use_fp && fmask[c1])) continue;
                        const int levk = std::max(1, (c1 == a1 ? m2 : m1) + 1), levc = std::max(nrm[c1], levk + 1);
                        if (levc < best) {
                            best = levc;
                            bc = c1;
                        }
                    }
                up[k2] = best;
                bestc[k2] = bc;
            }
            for (auto it = byh.rbegin(); it != byh.rend(); ++it) {
                const int v = *it;
                if (forced[v] || bestc[v] < 0) continue;
                cc[v] = bestc[v];
                isc[bestc[v]] = 1;
                forced[bestc[v]] = 1;
            }
        }
        // elimination level of every dense bus under a compress set = longest path of the dependencies (fixpoint on a DAG):
        // k waits for its dense children except its pending child; the pending child waits for v; p waits for v's pending child
        const std::vector<int> pard0(pard);
        auto levels = [&](const std::vector<int>& cset, std::vector<int>& lev) -> int {
            lev.assign(n, 0);
            int top = 0;
            for (bool changed = true; changed;) {
                changed = false;
                for (int k2 : keptl) {
                    int l = gjb[k2];                              // (level 0 stays the leaves' own: k_leaf_batch)
                    for (int i = dchild_ptr[k2]; i < dchild_ptr[k2 + 1]; ++i) {
                        const int ch = dchild[i];
                        if (cset[k2] != ch) l = std::max(l, lev[ch] + 1);
                        if (cset[ch] >= 0) l = std::max(l, lev[cset[ch]] + 1);
                    }
                    if (pard0[k2] >= 0 && cset[pard0[k2]] == k2) l = std::max(l, lev[pard0[k2]] + 1);
                    if (l != lev[k2]) {
                        lev[k2] = l;
                        changed = true;
                    }
                    top = std::max(top, l);
                }
            }
            return top;
        };
        std::vector<int> lev;
        if (compress_on) {
            int top = levels(cc, lev);
            std::vector<int> cand;
            for (int v : keptl)
                if (cc[v] >= 0) cand.push_back(v);
            std::stable_sort(cand.begin(), cand.end(), [&](int a, int b2) { return height[a] < height[b2]; });
            for (int v : cand) {                                  // keep only the steps that shorten the chain
                const int c1 = cc[v];
                cc[v] = -1;
                std::vector<int> l2;
                if (levels(cc, l2) <= top) continue;
                cc[v] = c1;
            }
            levels(cc, lev);
            for (int v : keptl) {
                if (cc[v] < 0) continue;
                const int ci = T.n_comp++;
                comp_role[v] = 1;
                comp_role[cc[v]] = 2;
                comp_idx[v] = comp_idx[cc[v]] = ci;
                comp_child[v] = cc[v];
                T.comp_v.push_back(v);
                T.comp_c.push_back(cc[v]);
            }
        }
        if (T.n_comp > 0) {
            for (int i = 0; i < T.n_comp; ++i) pard[T.comp_c[i]] = pard0[T.comp_v[i]];
            for (int k2 : keptl) height[k2] = lev[k2];
            // back sweep: x_k needs x of its (new) dense parent; a compressed bus needs its pending child's as well, which comes later
            for (int k2 : keptl) depth[k2] = 0;
            for (bool changed = true; changed;) {
                changed = false;
                for (int k2 : keptl) {
                    const int bd2 = comp_role[k2] == 1 ? comp_child[k2] : pard[k2];
                    const int dd = bd2 < 0 ? 0 : depth[bd2] + 1;
                    if (dd != depth[k2]) {
                        depth[k2] = dd;
                        changed = true;
                    }
                }
            }
            int mh = 0, md = 0;
            for (int k2 : keptl) {
                mh = std::max(mh, height[k2]);
                md = std::max(md, depth[k2]);
            }
            T.n_levels = mh + 1;
            T.n_depths = md + 1;
            bucket(height, T.n_levels, T.lvl_ptr, T.lvl_nodes);
            bucket(depth, T.n_depths, T.dep_ptr, T.dep_nodes);
            std::fill(dchild_ptr.begin(), dchild_ptr.end(), 0);
            for (int k2 : keptl)
                if (pard[k2] >= 0) dchild_ptr[pard[k2] + 1]++;
            for (int i = 0; i < n; ++i) dchild_ptr[i + 1] += dchild_ptr[i];
            std::vector<int> pos(dchild_ptr.begin(), dchild_ptr.end() - 1);
            for (int k2 : keptl)
                if (pard[k2] >= 0) dchild[pos[pard[k2]]++] = k2;
        }
    }
    // per-parent lazy records and images; the parent's dense-child list keeps its non-lazy children first
    std::vector<int> lzrec, lz_idx(n, -1), n_lazy(n, 0), n_slz(n, 0);
    std::vector<double> lzimg;
    T.n_lazy_parents = 0;
    T.n_lazy_leaves = (int)lazies.size();
    if (!lazies.empty() || !slzs.empty()) {
        typedef std::complex<double> cd;
        const int Hn = d->Hn;
        const int NTc = (BWc + 16) / 16;
        const size_t CTc = (size_t)NTc * NTc * 256;
        auto R = [](cd z, int t, int t2) { return (t == t2) ? z.real() : (t ? z.imag() : -z.imag()); };   // R(z) = [re -im; im re]
        for (int pk = 0; pk < n; ++pk) {
            const int L = (int)lazy_of[pk].size();
            const int LS = (int)slz_of[pk].size();
            if (L == 0 && LS == 0) continue;
            n_lazy[pk] = L;
            n_slz[pk] = LS;
            lz_idx[pk] = T.n_lazy_parents++;
            const int np = (L + 1) / 2;
            const size_t off = lzimg.size();
            const size_t sl_img = (size_t)3 * 64 * NTc + (size_t)10 * BWc;      // per lazy super-leaf: A operands [3][64][NT] | QH [10][B]
            lzimg.resize(off + CTc + (size_t)np * NTc * 64 + (size_t)np * NTc * 2 * 64 + (size_t)LS * sl_img, 0.0);
            double* I0 = &lzimg[off];
            double* IA = I0 + CTc;
            double* IH = IA + (size_t)np * NTc * 64;
            for (int row = 0; row < b; ++row)
                for (int col = 0; col < b; ++col) {
                    double v = 0.0;
                    for (int li : lazy_of[pk]) v += R(lazies[li].C0[(size_t)(row >> 1) * Hn + (col >> 1)], row & 1, col & 1);
                    for (int zi : slz_of[pk]) v += R(slzs[zi].C0[(size_t)(row >> 1) * Hn + (col >> 1)], row & 1, col & 1);
                    const long long o = tile_off(row, col);
                    if (o >= 0) I0[o] = v;
                }
            for (int pr = 0; pr < np; ++pr)
                for (int lg = 0; lg < 4; ++lg) {
                    const int idx = 2 * pr + (lg >> 1), a = lg & 1;
                    if (idx >= L) continue;
                    const LazyLeaf& ll = lazies[lazy_of[pk][idx]];
                    for (int tr = 0; tr < NTc; ++tr)                     // MFMA A operand: lane (jj, lg) = R(Gc)[16 tr + jj][a], stored [pair][lane][tr]
                        for (int jj = 0; jj < 16; ++jj) {
                            const int row = 16 * tr + jj;
                            if (row < b) IA[((size_t)pr * 64 + lg * 16 + jj) * NTc + tr] = R(ll.G[row >> 1], row & 1, a);
                        }
                    for (int tc = 0; tc < NTc; ++tc)                     // rows of R(Hr): lane (lg, jj) = R(Hr)[a'][16 tc + jj] of leaf lg >> 1, [pair][tc][lane][a']
                        for (int a2 = 0; a2 < 2; ++a2)
                            for (int jj = 0; jj < 16; ++jj) {
                                const int col = 16 * tc + jj;
                                if (col < b) IH[(((size_t)pr * NTc + tc) * 64 + lg * 16 + jj) * 2 + a2] = R(ll.H[col >> 1], a2, col & 1);
                            }
                }
            {   // lazy super-leaves: A operands (rows >= 2 of g Pb; rows 0 / 1 come per scenario) and Qb h (columns >= 2)
                double* IS = IH + (size_t)np * NTc * 2 * 64;
                for (int zc = 0; zc < LS; ++zc) {
                    const LazySuper& z = slzs[slz_of[pk][zc]];
                    const int mz = 2 * z.m1;
                    double* ZA = IS + (size_t)zc * sl_img;
                    double* ZQ = ZA + (size_t)3 * 64 * NTc;
                    for (int ch = 0; ch < 3; ++ch)
                        for (int lg = 0; lg < 4; ++lg) {
                            const int i = 4 * ch + lg;
                            if (i >= mz) continue;
                            for (int tr = 0; tr < NTc; ++tr)
                                for (int jj = 0; jj < 16; ++jj) {
                                    const int row = 16 * tr + jj;
                                    if (row >= 2 && row < b) ZA[((size_t)ch * 64 + lg * 16 + jj) * NTc + tr] = R(z.GP[(size_t)(row >> 1) * z.m1 + (i >> 1)], row & 1, i & 1);
                                }
                        }
                    for (int j = 0; j < mz; ++j)
                        for (int col = 2; col < b; ++col) ZQ[(size_t)j * BWc + col] = R(z.QH[(size_t)(j >> 1) * Hn + (col >> 1)], j & 1, col & 1);
                }
            }
            int rec[8] = {(int)off, L, -1, -1, -1, -1, 0, 0};
            for (int i = 0; i < L; ++i) rec[2 + i] = lazies[lazy_of[pk][i]].k;
            lzrec.insert(lzrec.end(), rec, rec + 8);
            std::stable_partition(dchild.begin() + dchild_ptr[pk], dchild.begin() + dchild_ptr[pk + 1], [&](int ch) { return !is_lazy[ch]; });
        }
    }
    // elimination level 0: lazy leaves first (k_leaf_batch takes them 16 scenarios at a time), the other leaves behind them
    T.n_lazy_level0 = 0;
    if (T.n_levels > 0) {
        auto first = T.lvl_nodes.begin() + T.lvl_ptr[0], last = T.lvl_nodes.begin() + T.lvl_ptr[1];
        auto mid = std::stable_partition(first, last, [&](int k2) { return is_lazy[k2] != 0; });
        T.n_lazy_level0 = (int)(mid - first);
    }
    T.lvl_nbatch.assign(T.n_levels > 0 ? T.n_levels : 1, 0);     // factor sweep: vector-only super-leaves of a level first
    std::vector<int> is_slz(n, 0);
    for (const LazySuper& z : slzs) is_slz[z.k] = 1;
    for (int l = 1; l < T.n_levels; ++l) {
        auto first = T.lvl_nodes.begin() + T.lvl_ptr[l], last = T.lvl_nodes.begin() + T.lvl_ptr[l + 1];
        auto mid = std::stable_partition(first, last, [&](int k2) { return is_slz[k2] != 0 && sb_ord[k2] >= 0; });
        T.lvl_nbatch[l] = (int)(mid - first);
    }
    T.dep_nleaf.assign(T.n_depths > 0 ? T.n_depths : 1, 0);      // back sweep: the leaves of a depth first
    for (int dl = 0; dl < T.n_depths; ++dl) {
        auto first = T.dep_nodes.begin() + T.dep_ptr[dl], last = T.dep_nodes.begin() + T.dep_ptr[dl + 1];
        auto mid = std::stable_partition(first, last, [&](int k2) { return cleaf_of[k2] >= 0; });
        auto mid2 = std::stable_partition(mid, last, [&](int k2) { return sb_ord[k2] >= 0; });   // batched super-leaves next
        T.dep_nleaf[dl] = (int)(mid2 - first);
    }
    const long long sl_base = (long long)lzimg.size();          // super-leaf constants ride behind the lazy images
    lzimg.insert(lzimg.end(), slimg.begin(), slimg.end());
    T.lvl_all_leaf.assign(T.n_levels > 0 ? T.n_levels : 1, 1);
    T.plain_gj.assign(n, 0);
    for (int pos = 0; pos < T.n_dense; ++pos) {
        const int k = T.lvl_nodes[pos];
        int* r = &fdesc[(size_t)pos * FDESC];
        int diag_e = -1;
        for (int e = d->rowptr[k]; e < d->rowptr[k + 1]; ++e)
            if (d->col[e] == k) diag_e = e;
        if (diag_e < 0) return HPF_E_ARG;
        r[0] = k;
        r[1] = pard[k];
        r[2] = diag_e;
        r[3] = d->dev_of_bus[k];
        r[4] = e_dn[k];
        r[5] = e_up[k];
        r[6] = T.child_ptr[k];
        r[7] = T.child_mid[k] - T.child_ptr[k];
        r[8] = dchild_ptr[k];
        r[9] = dchild_ptr[k + 1] - dchild_ptr[k] - n_lazy[k] - n_slz[k];      // children whose Schur complement is read from HBM
        for (int i = 0; i < 4 && i < r[9]; ++i) r[10 + i] = dchild[dchild_ptr[k] + i];
        r[14] = ((k > 0 && pass[T.parent[k]]) ? 1 : 0) | (is_lazy[k] ? 2 : 0);   // bit 0: linked to its dense parent through a contracted chain; bit 1: lazy leaf
        r[15] = lz_idx[k] >= 0 ? -(lz_idx[k] + 1) : cleaf_of[k] + 1;     // > 0: constant-inverse leaf, 1 + slot in Tree::d_Minv; < 0: -(1 + lazy record)
        if (cleaf_of[k] < 0) T.lvl_all_leaf[height[k]] = 0;
        T.plain_gj[k] = (cleaf_of[k] < 0 && !is_lazy[k] && !(sl_off[k] >= 0 && lz_idx[k] >= 0) && !comp_role[k] && !(k > 0 && pass[T.parent[k]])) ? 1 : 0;
        for (int i = 0; i < 4; ++i) {                              // first four 2x2-algebra children: (child, e_dn, e_up), no child3 hop
            const int cp = T.child_ptr[k] + i;
            const bool has = cp < T.child_mid[k];
            r[16 + 3 * i] = has ? T.child[cp] : -1;
            r[17 + 3 * i] = has ? e_dn[T.child[cp]] : 0;
            r[18 + 3 * i] = has ? e_up[T.child[cp]] : 0;
        }
        for (int i = 28; i < 36; ++i) r[i] = (i >= 30 && i < 34) ? -1 : 0;
        if (lz_idx[k] >= 0)                                        // lazy-leaf record inline: image offset, L, leaf ids[4]
            for (int i = 0; i < 6; ++i) r[28 + i] = lzrec[(size_t)lz_idx[k] * 8 + i];
        r[36] = r[37] = -1;
        r[38] = r[39] = 0;
        for (int zc = 0; zc < n_slz[k]; ++zc) {                    // lazy super-leaf children: bus, border unknowns
            const LazySuper& z = slzs[slz_of[k][zc]];
            r[36 + zc] = z.k;
            r[38] |= (2 * z.m1) << (8 * zc);
        }
        if (sl_off[k] >= 0 && lz_idx[k] >= 0) {                    // super-leaf: Z0 image slot, offset of [Tc | Pb | Qb]
            r[14] |= 4;
            if (sb_ord[k] >= 0) r[14] |= 8;                        // its back sweep is k_sleaf_back_batch's: T^-1 instead of the inverse
            r[34] = sl_slot[k];
            r[35] = (int)(sl_base + sl_off[k]);
            r[39] = sb_ord[k] >= 0 ? sb_ord[k] : 0;                // its image in Tree::d_sbimg (k_sleaf_batch)
        }
        for (int i = 40; i < FDESC; ++i) r[i] = 0;
        if (comp_role[k]) {                                        // compress step: role, slot, (v:) pending child c, entries (v,c), (c,v), c behind a chain
            r[40] = comp_role[k];
            r[41] = comp_idx[k];
            if (comp_role[k] == 1) {
                const int cb = comp_child[k];
                r[42] = cb;
                r[43] = e_dn[cb];
                r[44] = e_up[cb];
                r[45] = pass[T.parent[cb]] ? 1 : 0;
            }
        }
        const int kb = T.dep_nodes[pos];
        bdesc[(size_t)pos * 4 + 0] = kb;
        bdesc[(size_t)pos * 4 + 1] = pard[kb];
        bdesc[(size_t)pos * 4 + 2] = cleaf_of[kb] + 1;
        bdesc[(size_t)pos * 4 + 3] = comp_role[kb] ? ((comp_role[kb] << 28) | comp_idx[kb]) : 0;
    }
    {
        int nc = 0, nb = 0, nn = 0;
        for (int i = 0; i < n; ++i) {
            if (!kept(i)) continue;
            if (cleaf_of[i] >= 0) ++nc;
            if (sl_off[i] >= 0 && lz_idx[i] >= 0) {
                ++nb;
                if (sl_nest[i] > 0) ++nn;
            }
        }
        const int cs[8] = {T.n_dense, T.n_dense - nc - nb, nc, T.n_lazy_leaves, nb, nn, T.n_levels, T.n_depths};
        for (int i = 0; i < 8; ++i) T.census[i] = cs[i];
    }
    const double bd = b;
    // FP64 flop count of the dense part of the elimination (one scenario, one Newton step).  Gauss-Jordan bus: 2 b^3 (block
    // inversion) + 2 b^2 (w = D^-1 y) + b^2 per dense child (Schur complement subtracted) + 8 b^2 (push G D^-1 H) if not the
    // root.  Constant-inverse leaf: 4 b^2 (rank-2 update) + 4 b^2 (S^-1 row scaling) + 2 b^2 (w) + 8 b^2 (push).  Back sweep:
    // 2 b^2 per non-root dense bus.  The 2x2 work of the linear subtrees and chains (~60 flop per bus and harmonic) is not counted.
    // Algorithmic HBM bytes of the factor sweep: every Schur complement once out and once in, the inverse of a Gauss-Jordan bus
    // once out (tile image, padded rows skipped: TB bytes), plus the per-scenario operands of a bus (voltages, mismatch rows,
    // w, A(k,parent), 2x2 results of the folded children); model data shared by all scenarios (Y, Y_N, leaf images) is not counted.
    T.flops_factor = 0.0;
    T.bytes_factor = 0.0;
    T.bytes_back = 0.0;
    int n_dense_nonroot = 0;
    const int BWf = wave_block_size(b);
    // bytes of a block's tile image (TileIO: the last tile column is stored 8 wide when it holds at most 8 columns)
    const int NTf = (BWf + 16) / 16, LWf = (BWf + 1 - 16 * (NTf - 1)) <= 8 ? 8 : 16;
    const double TB = BWf ? (double)((BWf + 3) / 4) * ((NTf - 1) * 64 + 4 * LWf) * 8.0 : 8.0 * bd * bd;
    T.flops_gj = 0.0;
    T.bytes_gj = 0.0;
    T.n_gj_launches = 0;
    for (int l = 0; l < T.n_levels; ++l) {         // launches of the general kernel k_factor_q<B, false> per sweep (default switches)
        const int cntl = T.lvl_ptr[l + 1] - T.lvl_ptr[l];
        const int nb = (l == 0) ? (T.lvl_all_leaf[0] ? T.n_lazy_level0 : 0) : T.lvl_nbatch[l];
        const bool leafk = (l == 0 && nb > 0) || (nb == 0 && T.lvl_all_leaf[l]);
        if (cntl - nb > 0 && !leafk) ++T.n_gj_launches;
    }
    for (int i = 0; i < n; ++i) {
        if (!kept(i)) continue;
        const int nch = dchild_ptr[i + 1] - dchild_ptr[i];
        const bool leaf = cleaf_of[i] >= 0;
        const int nlz = n_lazy[i];                                   // lazy leaves: 2x2 core + G w column in, rank-2 MFMA update
        const bool sl = sl_off[i] >= 0 && lz_idx[i] >= 0;             // bordered bus: m x m inversion, rank-4 MFMAs per 4 border unknowns, S^-1, w
        double msl = 2.0 + 2.0 * nlz;
        for (int zi : slz_of[i]) msl += 2.0 * slzs[zi].m1;                 // nested bordered children: their borders join this bus's
        double fl = 0.0, by = 0.0;
        if (sl)
            fl += 2.0 * msl * msl * msl + 2.0 * bd * bd * 4.0 * (double)((2 + 2 * nlz + 3) / 4) + 2.0 * bd * msl * 3.0 +
                  4.0 * bd * bd + 2.0 * bd * bd;
        else
            fl += leaf ? 10.0 * bd * bd : 2.0 * bd * bd * bd + 2.0 * bd * bd + bd * bd * (nch - nlz - n_slz[i]);
        if (nlz && !sl) fl += 4.0 * bd * bd * nlz + 4.0 * bd * bd;
        const bool slb = sl && sb_ord[i] >= 0;                        // super-leaf that keeps T^-1 (+ W^-1, S^-1) instead of its inverse
        const int nsz = n_slz[i];                                     // lazy super-leaf children: T^-1, borders, G w in; rank-m rebuild
        if (nsz) fl += nsz * (2.0 * bd * bd * 12.0 + 2.0 * bd * 30.0);
        by += TB * (nch - nlz - nsz) + nlz * (32.0 + 8.0 * bd) + nsz * (8.0 * 104 + 64.0 + 8.0 * bd) +
              (leaf ? 0.0 : (slb ? 8.0 * 104 + 32.0 * d->Hn : TB)) +
              8.0 * (4.0 * bd + bd + bd + 2.0 * bd) +
              48.0 * d->Hn * (T.child_mid[i] - T.child_ptr[i]);
        if (i > 0) {
            fl += is_lazy[i] ? 8.0 * bd : 8.0 * bd * bd;
            by += is_lazy[i] ? 8.0 * bd : TB;
            // back sweep: inverse of a Gauss-Jordan bus in (leaves rebuild it from the shared image), w, A(k,parent), x of the
            // parent in, x out; leaves also their 2x2 core and S^-1
            T.bytes_back += (leaf ? 32.0 + 32.0 * d->Hn : (slb ? 8.0 * 104 + 32.0 * d->Hn : TB)) + 8.0 * (bd + 2.0 * bd + bd + bd);
            ++n_dense_nonroot;
        }
        T.flops_factor += fl;
        T.bytes_factor += by;
        const int hl2 = height[i];
        const bool batched = hl2 == 0 ? (T.lvl_all_leaf[0] != 0) : (is_slz[i] != 0 && sb_ord[i] >= 0);
        const bool leafkern = hl2 > 0 && !batched && T.lvl_nbatch[hl2] == 0 && T.lvl_all_leaf[hl2];
        if (!batched && !leafkern) {                                   // goes through k_factor_q<B, false>
            T.flops_gj += fl;
            T.bytes_gj += by;
        }
    }
    // compress steps: three more tile images out of v and into c (c's extra Schur complement, Gd, Hd), three more element-wise pushes,
    // the dense push of c (two rank-b products), Hd once more and a dense matrix-vector product in the back sweep
    if (T.n_comp > 0) {
        const double nc2 = (double)T.n_comp;
        T.flops_factor += nc2 * (3.0 * 8.0 * bd * bd + 4.0 * bd * bd * bd + bd * bd);
        T.bytes_factor += nc2 * 6.0 * TB;
        T.flops_gj += nc2 * (3.0 * 8.0 * bd * bd + 4.0 * bd * bd * bd + bd * bd);
        T.bytes_gj += nc2 * 6.0 * TB;
        T.bytes_back += nc2 * (TB + 8.0 * bd + 32.0 * d->Hn);
    }
    T.flops_per_solve = T.flops_factor + 2.0 * bd * bd * (n_dense_nonroot + T.n_comp);
    if (h->sw("HPF_TREE_INFO")) {
        int sl_nl = 0, sl_lin = 0, sl_lvl[4] = {0, 0, 0, 0};
        for (int i = 1; i < n; ++i)
            if (kept(i) && n_lazy[i] > 0 && dchild_ptr[i + 1] - dchild_ptr[i] == n_lazy[i]) {
                (i >= d->m ? sl_nl : sl_lin)++;
                sl_lvl[height[i] < 3 ? height[i] : 3]++;
            }
        fprintf(stderr, "hpf tree: buses whose dense children are all lazy leaves: %d nonlinear + %d linear (levels 1/2/3+: %d/%d/%d), %d built as super-leaves\n",
                sl_nl, sl_lin, sl_lvl[1], sl_lvl[2], sl_lvl[3], n_sleaf);
    }
    if (h->sw("HPF_TREE_INFO"))
        fprintf(stderr, "hpf tree (%s): %d buses, %d dense in %d levels, %d chains, %d constant-inverse leaves, %d lazy under %d parents\n",
                contract ? "contracted" : "plain", n, T.n_dense, T.n_levels, T.n_chains, T.n_cleaf, T.n_lazy_leaves, T.n_lazy_parents);
    // host-side plan of the dense tree (tools/tree_plan.py): one line per dense bus -- to the file hpf_tree_plan names, or env HPF_TREE_DUMP
    if (const char* dump_path = h->plan_path ? h->plan_path : h->sw("HPF_TREE_DUMP")) {
        if (FILE* fp = fopen(dump_path, contract ? "w" : "a")) {
            h->plan_written = true;
            fprintf(fp, "# %s tree: k pard height depth kind(0 gauss-jordan, 1 constant-inverse leaf, 2 bordered) vector_only hbm_children via_chain compress_role "
                        "plain_gauss_jordan_on_a_root_path_of_a_tie_endpoint\n", contract ? "contracted" : "plain");
            if (h->n_ties > 0)
                fprintf(fp, "# meshed: %d loop-closing lines, %d endpoint buses, border %d unknowns, bordered step %s\n", h->n_ties, h->n_tb, h->m_border,
                        h->mesh_sel ? "factor-once" : "virtual sweeps");
            for (int pos = 0; pos < T.n_dense; ++pos) {
                const int k = T.lvl_nodes[pos];
                const int kind = cleaf_of[k] >= 0 ? 1 : ((sl_off[k] >= 0 && lz_idx[k] >= 0) ? 2 : 0);
                fprintf(fp, "%d %d %d %d %d %d %d %d %d %d\n", k, pard[k], height[k], depth[k], kind, is_lazy[k],
                        dchild_ptr[k + 1] - dchild_ptr[k] - n_lazy[k] - n_slz[k], (k > 0 && pass[T.parent[k]]) ? 1 : 0, comp_role[k],
                        (use_fp && fmask[k]) ? (T.plain_gj[k] ? 1 : -1) : 0);
            }
            fclose(fp);
        }
    }
    T.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (h->sw("HPF_TREE_INFO")) fprintf(stderr, "hpf tree (%s): planned on the host in %.1f ms\n", contract ? "contracted" : "plain", T.plan_ms);
    if (h->plan_only) return HPF_OK;            // hpf_tree_plan: host-only, nothing goes to a device
    int r;
    if ((r = upload(h, &T.d_parent, T.parent))) return r;
    if ((r = upload(h, &T.d_lvl_nodes, T.lvl_nodes))) return r;
    if ((r = upload(h, &T.d_dep_nodes, T.dep_nodes))) return r;
    if ((r = upload(h, &T.d_child_ptr, T.child_ptr))) return r;
    if ((r = upload(h, &T.d_child, T.child))) return r;
    if ((r = upload(h, &T.d_e_up, e_up))) return r;
    if ((r = upload(h, &T.d_e_dn, e_dn))) return r;
    if ((r = upload(h, &T.d_child_mid, T.child_mid))) return r;
    if ((r = upload(h, &T.d_lin, T.lin))) return r;
    if ((r = upload(h, &T.d_lin_ptr, T.lin_ptr))) return r;
    if ((r = upload(h, &T.d_lin_post, T.lin_post))) return r;
    if ((r = upload(h, &T.d_all_ptr, T.all_ptr))) return r;
    if ((r = upload(h, &T.d_all_post, T.all_post))) return r;
    if ((r = upload(h, &T.d_fdesc, fdesc))) return r;
    if ((r = upload(h, &T.d_child3, child3))) return r;
    if ((r = upload(h, &T.d_bdesc, bdesc))) return r;
    {
        // back sweep of the bordered buses: a nested bus before the bordered children below it -> groups by nesting order, highest first
        std::vector<int> bsleaf;
        int max_nest = 0;
        for (int i = 0; i < n; ++i) max_nest = std::max(max_nest, sl_nest[i]);
        T.bsleaf_ptr.assign(1, 0);
        for (int nest = max_nest; nest >= 0; --nest) {
            for (int pos = 0; pos < T.n_dense; ++pos) {
                const int kb = T.dep_nodes[pos];
                if (sb_ord[kb] < 0 || sl_off[kb] < 0 || lz_idx[kb] < 0 || sl_nest[kb] != nest) continue;
                const int rec[8] = {kb, pard[kb], sb_ord[kb], (int)(sl_base + sl_off[kb]), sb_m[kb], 0, 0, 0};
                bsleaf.insert(bsleaf.end(), rec, rec + 8);
            }
            T.bsleaf_ptr.push_back((int)bsleaf.size() / 8);
        }
        T.n_bsleaf = (int)bsleaf.size() / 8;
        if ((r = upload(h, &T.d_bsleaf, bsleaf))) return r;
        // the same records by back-sweep depth (k_level_back: a bordered bus needs its dense parent's x only, i.e. the previous depth)
        std::vector<int> bsd;
        T.bsl_dep_ptr.assign(1, 0);
        for (int dl = 0; dl < T.n_depths; ++dl) {
            for (int pos = T.dep_ptr[dl]; pos < T.dep_ptr[dl + 1]; ++pos) {
                const int kb = T.dep_nodes[pos];
                if (sb_ord[kb] < 0 || sl_off[kb] < 0 || lz_idx[kb] < 0) continue;
                const int rec[8] = {kb, pard[kb], sb_ord[kb], (int)(sl_base + sl_off[kb]), sb_m[kb], 0, 0, 0};
                bsd.insert(bsd.end(), rec, rec + 8);
            }
            T.bsl_dep_ptr.push_back((int)bsd.size() / 8);
        }
        if ((r = upload(h, &T.d_bsleaf_dep, bsd))) return r;
        if ((r = upload(h, &T.d_sbimg, sbimg))) return r;
    }
    {
        std::vector<int> bleaf;
        T.bleaf_dep_ptr.assign(1, 0);
        for (int dl = 0; dl < T.n_depths; ++dl) {                // (bdesc is in depth order)
            for (int pos = T.dep_ptr[dl]; pos < T.dep_ptr[dl + 1]; ++pos)
                if (bdesc[(size_t)pos * 4 + 2] > 0 && bdesc[(size_t)pos * 4 + 1] >= 0) bleaf.insert(bleaf.end(), &bdesc[(size_t)pos * 4], &bdesc[(size_t)pos * 4] + 4);
            T.bleaf_dep_ptr.push_back((int)bleaf.size() / 4);
        }
        T.n_bleaf = (int)bleaf.size() / 4;
        if ((r = upload(h, &T.d_bleaf, bleaf))) return r;
    }
    if ((r = upload(h, &T.d_dchild, dchild))) return r;
    if ((r = upload(h, &T.d_comp_child, T.comp_c))) return r;
    if ((r = upload(h, &T.d_chain_ptr, T.chain_ptr))) return r;
    if ((r = upload(h, &T.d_chain_nodes, T.chain_nodes))) return r;
    if ((r = upload(h, &T.d_chain_ch, T.chain_ch))) return r;
    if ((r = upload(h, &T.d_Minv, minv))) return r;
    if ((r = upload(h, &T.d_lzrec, lzrec))) return r;
    if ((r = upload(h, &T.d_lzimg, lzimg))) return r;
    if ((r = upload(h, &T.d_lbimg, lbimg))) return r;
    if ((r = upload(h, &T.d_lrec, lrec))) return r;
    if ((r = upload(h, &T.d_lb2rec, lb2rec))) return r;
    if ((r = upload(h, &T.d_lb2x, lb2x))) return r;
    if ((r = upload(h, &T.d_lb2ptr, lb2ptr))) return r;
    if ((r = upload(h, &T.d_lb2cptr, lb2cptr))) return r;
    if ((r = upload(h, &T.d_lb2clist, lb2clist))) return r;
    if ((r = upload(h, &T.d_lbrec, lbrec))) return r;
    if ((r = upload(h, &T.d_lbptr, lbptr))) return r;
    if ((r = upload(h, &T.d_crec, crec))) return r;
    if ((r = upload(h, &T.d_cnode, cnode))) return r;
    if ((r = upload(h, &T.d_arec, arec))) return r;
    return HPF_OK;
}

// host-only: runs the tree planning of a radial model exactly as hpf_create would for a handle of d->max_scenarios scenarios (the compress
// steps are a default of handles of up to 256) and writes the plan to `path`; returns before anything would go to a device
int tree_plan_dump(const hpf_desc* d, const char* path) {
    hpf_handle tmp;
    tmp.n = d->n; tmp.m = d->m; tmp.c = d->c; tmp.Hn = d->Hn; tmp.nnz = d->nnz; tmp.coupled = d->coupled;
    tmp.S_max = d->max_scenarios;
    tmp.plan_path = path;
    tmp.plan_only = true;
    if (const char* es = getenv("HPF_ENV_SWITCHES")) tmp.env_switches = atoi(es) != 0;     // (the same opt-in as hpf_create)
    Tree T;
    int r = tree_find_ties(&tmp, d);                     // (meshed models: the loop-closing lines and the buses the factor-once bordered step keeps plain)
    if (r) return r;
    r = tree_build_into(&tmp, d, T, true);
    tree_free_one_fwd(T);
    if (r) return r;
    return tmp.plan_written ? HPF_OK : HPF_E_ARG;       // (the file could not be opened)
}

int tree_build(hpf_handle* h, const hpf_desc* d) {
    int r = tree_build_into(h, d, h->tree, false);
    if (r) return r;
    h->has_ctree = wave_block_size(2 * d->Hn) != 0;
    if (h->has_ctree) r = tree_build_into(h, d, h->ctree, true);
    return r;
}

}  // namespace hpf
