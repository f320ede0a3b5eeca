// Internal declarations shared by the translation units of libhpf.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdint.h>

#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/hpf.h"
#include "hpf_assembly.hpp"

namespace hpf {

enum { T_MISMATCH = 0, T_JACOBIAN = 1, T_SOLVE = 2, T_UPDATE = 3, T_BACK = 4, T_GJ = 5, T_GJ_DEV = 6, T_COUNT = 7 };

struct TimedSpan {
    int which;
    hipEvent_t e0, e1;
};

// Feeder-tree description for the BLOCK_TREE solver (host copies + device copies).
struct Tree {
    int n_levels = 0;                 // elimination levels (by height, leaves first)
    int n_depths = 0;                 // back-substitution levels (by depth, root first)
    std::vector<int> parent;          // [n], -1 for the root (bus 0)
    std::vector<int> lvl_ptr;         // [n_levels+1] into lvl_nodes
    std::vector<int> lvl_nodes;       // nodes grouped by height
    std::vector<int> dep_ptr;         // [n_depths+1] into dep_nodes
    std::vector<int> dep_nodes;       // nodes grouped by depth
    std::vector<int> child_ptr;       // [n+1]
    std::vector<int> child;           // children lists: all-linear-subtree children first, then dense children
    std::vector<int> child_mid;       // [n] end of the linear children inside a node's child list
    std::vector<int> lin;             // [n] 1: the whole subtree of the bus is linear -> harmonic-diagonal 2x2 algebra
    std::vector<int> lin_ptr;         // [n_lin_roots+1] into lin_post
    std::vector<int> lin_post;        // post-order node lists of the maximal linear subtrees
    std::vector<int> all_ptr, all_post;   // the whole tree as one post-order list (fundamental power flow)
    int n_lin_roots = 0;
    int n_dense = 0;
    int* d_parent = nullptr;
    int* d_lvl_nodes = nullptr;
    int* d_dep_nodes = nullptr;
    int* d_child_ptr = nullptr;
    int* d_child = nullptr;
    int* d_e_up = nullptr;            // [n] CSR position of entry (i, parent(i))
    int* d_e_dn = nullptr;            // [n] CSR position of entry (parent(i), i)
    int* d_child_mid = nullptr;
    int* d_lin = nullptr;
    int* d_lin_ptr = nullptr;
    int* d_lin_post = nullptr;
    int* d_all_ptr = nullptr;
    int* d_all_post = nullptr;
    int* d_fdesc = nullptr;           // [n_dense][16] node records of the multi-wave factor kernel, elimination-level order:
                                      //   k, parent, diag entry, device, e_dn, e_up, lin child begin, count, dense child begin,
                                      //   count, first four dense children (everything a block needs behind ONE scalar load)
    int* d_child3 = nullptr;          // [n-1][4] per child-list position: child, e_dn[child], e_up[child], 0
    int* d_bdesc = nullptr;           // [n_dense][4] (k, dense parent, constant-inverse slot + 1, 0) in back-substitution (depth) order
    // contraction of pass-through buses (linear bus, exactly one dense child): eliminated in 2x2-per-harmonic algebra before
    // the dense levels; `parent` stays the network parent, the dense tree links a chain's bottom bus to the chain's top parent
    int n_chains = 0;
    std::vector<int> chain_ptr, chain_nodes, chain_ch;   // chains bottom-up: nodes k1..kt, ch = the dense bus below k1
    int* d_dchild = nullptr;          // dense children lists of the dense buses (through chains), indexed by the node records
    int* d_chain_ptr = nullptr;
    int* d_chain_nodes = nullptr;
    int* d_chain_ch = nullptr;
    // constant-inverse leaves (contracted tree only): nonlinear buses without dense children.  In rectangular coordinates their
    // block is  R(y_kk I - Y_N - series terms of the linear neighbourhood)  + a 2x2 state-dependent term at the fundamental, so
    // the inverse of the harmonic part and its borders are computed ONCE per model (host, complex) and kept as a b x b image
    // [c0 Lr; Lc Ahh^-1] in accumulator-tile layout; the device adds the rank-2 term  [I; Lc] (c0 + D)^-1 [I Lr]
    // level-parallel kernels of the 2x2 algebra (contracted tree): all-linear-subtree buses grouped by their height inside the
    // subtree, one record of 8 ints per bus (k, diagonal entry, parent, e_up, e_dn, first child position, children, 0); chains:
    // 8 ints per chain (ch, e_dn[ch], e_up[ch], first node record, nodes, 0, 0, 0) and the same node records, bottom-up
    int n_lin_heights = 0;
    std::vector<int> lh_ptr;          // [n_lin_heights+1] into the records
    int* d_lrec = nullptr;
    // ... and the same records once more, grouped into bundles of whole subtrees (k_lin_tree_factor / k_lin_tree_back: one launch
    // for all heights): records sorted by (bundle, height), lb_ptr [n_lin_bundles][n_lin_heights + 1] offsets into them
    // ... and for the one-round-trip kernels (k_lin_bundle_factor / _back): bundles of at most 256 * lin_np items (bus, harmonic),
    // records with cbeg = first child slot, d_lb2x[record] = (own slot | -1, local index of the parent | -1)
    int n_lin_bundles2 = 0, lin_np = 0;
    int* d_lb2rec = nullptr;
    int* d_lb2x = nullptr;
    int* d_lb2ptr = nullptr;
    int chains_bundled = 0;           // the contracted chains ride in the bundle launches (d_lb2cptr [bundle + 1] into d_lb2clist: chain ids)
    int* d_lb2cptr = nullptr;
    int* d_lb2clist = nullptr;
    int n_lin_bundles = 0;
    int* d_lbrec = nullptr;
    int* d_lbptr = nullptr;
    int* d_crec = nullptr;            // chain headers
    int* d_cnode = nullptr;           // chain node records
    int n_all_heights = 0;            // the whole tree by height (fundamental power flow)
    std::vector<int> ah_ptr;
    int* d_arec = nullptr;
    std::vector<int> lvl_all_leaf;    // [n_levels] 1: every bus of the elimination level is a constant-inverse leaf
    std::vector<int> toff_tab;        // [b][b] offset of (row, col) of a block in a tile image (TileIO<B>::off), -1: not stored
    std::vector<char> plain_gj;       // [n] 1: dense bus on the plain Gauss-Jordan path (its inverse S_k^-1 is in its inverse slot after a sweep)
    int n_cleaf = 0;
    double* d_Minv = nullptr;         // [n_cleaf][NT*NT*256]
    // lazy leaves: constant-inverse leaves directly under their dense parent never write their Schur complement; the parent
    // rebuilds the sum from per-model images and the leaves' 2x2 cores (k_factor_q, "lazy" phase)
    int n_lazy_parents = 0, n_lazy_leaves = 0;
    int n_lazy_level0 = 0;            // the first n_lazy_level0 buses of elimination level 0 are lazy leaves (k_leaf_batch)
    std::vector<int> lvl_nbatch;      // [n_levels] super-leaves of a level whose parent rebuilds their Schur complement: they come first
                                      // in the level's records and go through k_sleaf_batch (16 scenarios per workgroup)
    std::vector<int> dep_nleaf;       // [n_depths] constant-inverse leaves of a back-sweep depth (they come first in the depth's records)
    std::vector<int> dep_nskip;       // [n_depths] leaves + batched super-leaves of a depth: k_back_q starts behind them
    int census[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // hpf_tree_census
    std::vector<int> bsleaf_ptr;      // back sweep of the bordered buses in groups by nesting order (nested ones first): offsets into d_bsleaf
    int n_bsleaf = 0;
    int* d_bsleaf = nullptr;          // [n_bsleaf][8] back-sweep records of the super-leaves (k_sleaf_back_batch)
    double* d_sbimg = nullptr;        // [n_bsleaf][SleafImg<B>::SZ] their [0 0; 0 Ahh^-1] images (+ Qb rows) and Pb in MFMA A-operand layout
    std::vector<int> bsl_dep_ptr, bleaf_dep_ptr;   // [n_depths + 1] the same records by back-sweep depth (d_bsleaf_dep: depth order; d_bleaf is in depth order): k_level_back
    int* d_bsleaf_dep = nullptr;
    int n_bleaf = 0;
    int* d_bleaf = nullptr;           // [n_bleaf][4] back-sweep records of ALL constant-inverse leaves: one k_leaf_back_batch launch after the last depth
    double* d_lbimg = nullptr;        // [leaf slot][LeafBatchImg::SZ]: the leaf images in MFMA A-operand layout (16 scenarios per workgroup)
    int* d_lzrec = nullptr;           // [n_lazy_parents][8]: image offset (doubles), L, leaf ids[4] (-1: none), 0, 0
    double* d_lzimg = nullptr;        // per parent: sum of constant parts (tile layout) | A operands [pair][tr][64] | row factors [pair][tc][2][64]
    // compress steps on the Gauss-Jordan skeleton (contracted tree, HPF_COMPRESS != 0): bus v = comp_v[i] is eliminated BEFORE its
    // pending child c = comp_c[i] (its tallest dense child, alone on v's critical path); c then hangs under v's parent p with a dense
    // coupling pair, the elimination levels / back-sweep depths are the longest paths of the new dependencies
    int n_comp = 0;
    std::vector<int> comp_v, comp_c;
    int* d_comp_child = nullptr;      // [n_comp] c of compress step i (k_back_q: x_v needs x_c)
    double plan_ms = 0.0;             // host time of tree_build_into up to the uploads
    double flops_per_solve = 0.0;     // factor sweep + back sweep
    double flops_factor = 0.0;        // factor sweep only (k_tree_factor, all levels)
    double bytes_back = 0.0;          // algorithmic HBM bytes of the dense back sweep, one scenario and step
    double bytes_factor = 0.0;        // algorithmic HBM bytes of the factor sweep, one scenario and step (see hpf_solve_bytes)
    double flops_gj = 0.0, bytes_gj = 0.0;   // the share of the buses that go through the general kernel k_factor_q<B, false>
    int n_gj_launches = 0;            // launches of that kernel per sweep (and scenario group)
};

}  // namespace hpf

struct hpf_handle {
    // Build switches (A/B runs, diagnostics: "HPF_LAZY=0 HPF_SLEAF=1 ..."): the option string of hpf_create_opts, and -- ONLY when the process
    // opts in with HPF_ENV_SWITCHES=1 (the test-suite, tools/) -- the process environment.  Without that opt-in nothing a handle computes
    // depends on environment variables.
    std::string opts;
    bool env_switches = false;
    const char* sw(const char* name) const {
        const size_t ln = strlen(name);
        size_t pos = 0;
        while ((pos = opts.find(name, pos)) != std::string::npos) {
            const bool starts = pos == 0 || opts[pos - 1] == ' ' || opts[pos - 1] == ',' || opts[pos - 1] == ';';
            if (starts && pos + ln < opts.size() && opts[pos + ln] == '=') return opts.c_str() + pos + ln + 1;   // (atoi / the readers stop at the separator)
            pos += ln;
        }
        return env_switches ? getenv(name) : nullptr;
    }
    hpf::Model M{};                   // device pointers
    int n = 0, m = 0, c = 0, Hn = 0, nnz = 0, n_dev = 0, coupled = 0, solver = 0, device = 0;
    int S_max = 0, S = 0;
    int S_alloc = 0;                  // scenario slots allocated = S_max (+ 1 + m_border virtual slots of the bordered Newton step)
    // meshed networks on the block-tree path (hpf_block.hip, "bordered Newton step"): BFS spanning tree + loop-closing lines
    int n_ties = 0, n_tb = 0, m_border = 0;      // tie lines, their distinct endpoint buses, border unknowns = n_tb * 2 Hn
    int *d_tb_bus = nullptr;          // [n_tb] endpoint buses
    int *d_tb_ptr = nullptr;          // [n_tb + 1] into d_tb_adj
    int *d_tb_adj = nullptr;          // per (endpoint i, tie (i,j)): j, CSR entry (i,j), 0
    double *d_bM = nullptr, *d_brhs = nullptr;   // border system (m x m column-major, m)
    int *d_bipiv = nullptr, *d_binfo = nullptr;
    double *d_bM0 = nullptr, *d_brhs0 = nullptr;   // kept copy of the border system (residual check of the unpivoted LU) | copy of its right-hand side + 2 check words
    // factor-once form of the bordered step (tree_sel_build / tree_sel_run, hpf_block.hip): the tree is swept ONCE per Newton step and scenario
    // for y = J_t^-1 f; the border matrix comes from a selected inversion over P = the union of the endpoints' root paths, whose buses the planner
    // keeps as plain Gauss-Jordan buses (sel_forced) so that S_k^-1 sits in their inverse slot
    bool mesh_sel = false;
    std::vector<char> sel_forced;     // [n] 1: bus on a root path of a tie endpoint (empty: not used)
    std::vector<int> tb_bus_host;     // [n_tb] endpoint buses (host copy of d_tb_bus)
    int sel_nP = 0, sel_npairs = 0, sel_R = 0;
    int *d_sel_P = nullptr;           // [nP][4]: bus, index of its parent in P (-1: root), CSR entries (bus, parent), (parent, bus)
    int *d_sel_pidx = nullptr;        // [n] index in P or -1
    int *d_sel_toff = nullptr;        // [b][b] offset of (row, col) in a tile image of the inverse slot
    double *d_sel_S = nullptr, *d_sel_Z = nullptr, *d_sel_Up = nullptr;   // [nP][b][b] S^-1, S^-1 A(k, parent), S_parent^-1 A(parent, k): dense, row-major
    double *d_sel_W = nullptr;        // [npairs][b][b] forward blocks W[k, t]
    double *d_sel_X = nullptr;        // [nP][n_tb][b][b] (J_t^-1 E_T) restricted to P
    double *d_sel_tie = nullptr;      // [2 n_ties][Hn][4] the ties' coupling blocks of the current state
    void *d_sel_jobs = nullptr;       // BlkJob list: forward by height, back by depth
    std::vector<size_t> sel_fwd_beg, sel_back_beg, sel_hl_ptr;
    bool tree_back_only = false;      // tree_newton_step skips its factor part (set around that second pass only)
    int *d_sel_hl = nullptr, *d_sel_slot = nullptr, *d_sel_cptr = nullptr, *d_sel_clist = nullptr;   // P by height | endpoint number | children in P
    double *d_sel_dw = nullptr;       // [nP][b] corrections of the forward vectors
    int sel_cap = 1;                  // scenarios whose selected inversion / border solve run as one batch (every d_sel_* / d_bB buffer holds that many)
    double *d_sel_bM = nullptr, *d_sel_rhs = nullptr, *d_sel_g = nullptr;   // per batch position: border matrix (column-major, untouched by the solve), Q^T y, solution g
    unsigned long long* d_sel_res = nullptr;   // [sel_cap][2] residual check words (k_border_check)
    int* d_sel_info = nullptr;        // [sel_cap] weak / zero pivot of the block Gauss-Jordan solve, info of the unpivoted LU
    std::vector<unsigned long long> sel_res_host;
    std::vector<int> sel_info_host;
    double border_piv_limit = 1e3;    // ... amplification of a 4 x 4 pivot block's inverse beyond which that border system goes to the pivoted LU
    bool border_gj_mfma = true;       // ... its diagonal blocks inverted on the matrix cores (k_blk_invert_mfma; HPF_BORDER_GJ_MFMA=0: VALU)
    bool border_gj = false;           // border system by block Gauss-Jordan on the b x b grid (n_tb <= HPF_BORDER_GJ, default 96) instead of rocSOLVER's LU
    double *d_bB = nullptr;           // [n_tb][n_tb + 1][b][b] the border system in block layout, right-hand side in block column n_tb
    void *d_bgj_jobs = nullptr;       // its block-product jobs: per step the row scaling, then the elimination
    std::vector<size_t> bgj_beg;
    int border_pivoting = 0;          // option "border_pivoting": 1 = every border system through the pivoted LU (A/B, tests)
    int border_repivots = 0;          // border systems that went through the pivoted LU after the residual check
    std::vector<int> host_act;        // the slot list as the host last saw it (the bordered step walks the running scenarios)
    int N = 0, Nc = 0, Nf = 0;
    bool loads_set = false, state_set = false, mismatch_valid = false;
    int last_detail = 0;
    const char* plan_path = nullptr;  // hpf_tree_plan: file the planning run writes (instead of env HPF_TREE_DUMP) ...
    bool plan_only = false;           // ... and it stops before the uploads
    bool plan_written = false;
    int gj_mode = 1;                  // BLOCK_TREE block inversion: 0 pivoted wave Gauss-Jordan (VALU, uncontracted tree), 1 MFMA static 4x4 blocks, NT waves per bus (hpf_quad.hpp)
    double piv_limit = 1e10;          // static pivot order: amplification of a 4x4 pivot block's inverse beyond which a scenario is repeated with partial pivoting
    int fuse_back = 1;                // HPF_FUSEBACK (read by hpf_create): 0 = the back sweep's batched launches after the last depth instead of inside the depths' launches
    int fuse_back_max = 32;           // HPF_FUSEBACK_MAX (read by hpf_create): largest scenario group that takes the fused back sweep
    int border_slot_cap = 1024;       // HPF_BORDER_SLOTS (read by hpf_create): cap of the virtual scenario slots of a meshed handle's bordered step
    int fuse_levels = 1;              // HPF_FUSELEVEL (read by hpf_create): 0 = separate launches for the batched and the per-scenario workgroups of a level
    int leafbatch = 1;                // HPF_LEAFBATCH (read by hpf_create): 0 = one workgroup per (leaf, scenario) instead of 16 scenarios per workgroup
    int debug_ablate = 0;             // HPF_DEBUG_ABLATE: timing-only ablation of factor-kernel phases (results invalid)

    // model (device)
    int* d_rowrec = nullptr;          // [n][8] row records of the mismatch kernel (Model::rowrec)
    int *d_rowptr = nullptr, *d_col = nullptr, *d_diag = nullptr, *d_erow = nullptr, *d_dev = nullptr;
    hpf::cplx *d_Y = nullptr, *d_YN = nullptr, *d_IN = nullptr, *d_YNt = nullptr;
    // per-scenario state (device)
    double *d_P = nullptr, *d_Q = nullptr, *d_Vm = nullptr, *d_Va = nullptr;
    hpf::cplx *d_U = nullptr, *d_E = nullptr;
    double* d_f = nullptr;            // [S][N]  mismatch, overwritten by the Newton step during a solve
    hpf::cplx* d_I0 = nullptr;            // [S][n]  network current of the power rows, kept by the harmonic mismatch kernel
    unsigned long long* d_errbits = nullptr;   // [S]   ||f||_inf as the bit pattern of the double (hpf_mismatch: k_err_reduce)
    unsigned long long* d_errpart = nullptr;   // [S][errpart_stride] partial maxima of |f|, one per wavefront of the mismatch launch (k_mismatch)
    int errpart_stride = 0;
    double* d_err = nullptr;          // [S]
    int* d_niter = nullptr;           // [S]
    int* d_active = nullptr;          // [S]
    int* d_nactive = nullptr;         // [1]
    int* d_pivflag = nullptr;         // [S] bit0: a static 4x4 pivot block exceeded piv_limit (k_factor_q), bit1: repeated with partial pivoting,
                                      //     bit2: exactly zero pivot met by the pivoted wave Gauss-Jordan (k_factor_w)
    int* d_mask = nullptr;            // [S] scenarios of a repeat pass
    double *d_Vm0 = nullptr, *d_Va0 = nullptr;   // [S][Hn*n] state at the entry of hpf_solve (repeat with partial pivoting starts from it)
    double *d_Vmp = nullptr, *d_Vap = nullptr;   // [S][Hn*n] option "keep_previous_state": the state each scenario's LAST Newton step started from
    double *d_swapVm = nullptr, *d_swapVa = nullptr;   // [S][Hn*n] hpf_jacobian(_csr)_last: the current state while the kept one stands in its place
    int queue_chunk = 4;              // option "queue_chunk": iterations between two harvest / refill rounds of hpf_solve_queue
    int keep_prev = 0;
    bool prev_valid = false;          // d_Vmp / d_Vap belong to the last hpf_solve (set_state / set_loads invalidate them)
    double* d_hist = nullptr;         // [S][hist_cap]
    int hist_cap = 0;
    hpf_stat* d_stats = nullptr;      // [S]
    // dense solver
    double* d_J = nullptr;            // [S_alloc][Nmax*Nmax]
    size_t J_elems_per_scen = 0;
    int J_scen_alloc = 0;
    int* d_ipiv = nullptr;
    int* d_info = nullptr;
    bool info64 = false;              // the last dense factorisation went through rocSOLVER's 64-bit entry points (int64 pivots / info)
    // CSR export of the Jacobian (hpf_jacobian_csr): indptr [N+1] (built once per handle), column indices / values of one scenario
    int* d_jptr = nullptr;
    int* d_jcol = nullptr;
    double* d_jval = nullptr;
    long long jnnz = 0;
    // block-tree solver
    hpf::Tree tree;                   // elimination tree as the network gives it (single-wave / generic kernels, pf)
    hpf::Tree ctree;                  // the same with pass-through buses contracted (multi-wave kernels, gj_mode 1)
    bool has_ctree = false;
    int auto_repivot = 1;             // hpf_solve repeats scenarios flagged by the static-pivot monitor with partial pivoting
    int* h_act[2] = {nullptr, nullptr};          // pinned copies of d_active (hpf_solve looks at chunk c - 1 while chunk c runs)
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    double *trace_Vm = nullptr, *trace_Va = nullptr;   // hpf_set_trace: caller's [S][trace_cap][Hn*n] arrays (per-iteration states)
    int trace_cap = 0;
    double* d_Z = nullptr;            // [S][n][b*b]
    double* d_w = nullptr;            // [S][n][b]
    double* d_x = nullptr;            // [S][n][b]  Newton step, bus-major
    long long* d_dbg = nullptr;       // [S][n][8] diagnostic phase stamps of the factor kernel (HPF_DEBUG_ABLATE & 16)
    double* d_C = nullptr;            // [S][n][(B+1)*B] Schur complements pushed by dense children (MFMA mode)
    double* d_H = nullptr;            // [S][n][Hn][4] A(k,parent) blocks of the dense buses, kept for the back sweep
    double* d_chG = nullptr;          // [S][n][Hn][4] contracted tree: A'(top parent, ch) of a chain's bottom dense bus ch
    double* d_chH = nullptr;          // [S][n][Hn][4]                  A'(ch, top parent)
    double* d_chD = nullptr;          // [S][n][Hn][4] harmonic-diagonal addend to the diagonal block of ch
    double* d_chy = nullptr;          // [S][n][Hn][2] addend to the right-hand side of ch
    double* d_fb = nullptr;           // [S][n][Bst] bus-major image of the harmonic mismatch (zeros where there is no equation)
    double* d_lfK = nullptr;          // [S][n][12]    constant-inverse leaves: (c0 + D)^-1 (back sweep, lazy rebuild), G0 S_c^-1, H0 S_parent^-1 (lazy rebuild)
    double* d_lfS = nullptr;          // [S][n][Hn][4] constant-inverse leaves: S_q^-1 (polar <- rectangular), row-major
    double* d_chZ = nullptr;          // [S][n][Hn][4] D_k^-1 A'(k, ch) of the chain buses k (back substitution)
    double* d_linA = nullptr;         // [S][n][Hn][4] inverse 2x2 blocks of the all-linear-subtree buses
    double* d_F = nullptr;            // [S][n_comp][3][tile image] compress steps: A(c,v) D_v^-1 A(v,c) | Gd = A'(p,c) (MFMA A-operand order) | Hd = A'(c,p)
    double* d_H2 = nullptr;           // [S][n_comp][Hn][4] A(v,c) of the compressed buses (back sweep)

    hipStream_t own_stream = nullptr, stream = nullptr;
    // launch context: stream and scenario slice the launch helpers currently target (scenario groups run as independent
    // pipelines on their own streams so that the latency-bound upper tree levels of one group overlap the others)
    hipStream_t cur_stream = nullptr;
    int cur_s0 = 0, cur_S = 0;
    int n_groups = 4;
    hipStream_t gstream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // groups 1..7 (group 0: the handle's stream, group_stream)
    hipEvent_t fork_ev = nullptr, join_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    rocblas_handle blas = nullptr;
    double setup_ms[4] = {0, 0, 0, 0}; // hpf_create: total | tree planning on the host | tree uploads | per-scenario allocation
    int timing = 0;                   // hpf_timing_enable: 1 HIP-event spans + device stamps, 2 device stamps only (no event packets between kernels)
    std::vector<hpf::TimedSpan> spans;
    double t_ms[hpf::T_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    int64_t t_n[hpf::T_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    static constexpr int TS_CAP = 16384;         // launches of the general factor kernel a timing leg can stamp
    unsigned long long* d_tstamp = nullptr;      // [TS_CAP][2] first workgroup start / last workgroup end (wall_clock64) per launch
    int ts_next = 0;
};

namespace hpf {

// Stream of scenario group g: group 0 runs on the handle's stream itself, so a step of G groups keeps G hardware queues busy, not G + 1 (the
// runtime maps streams onto four queues; a fifth busy stream shares one and serialises two groups: 1.25 instead of 0.90 ms per step at G = 4)
inline hipStream_t group_stream(const hpf_handle* h, int g) { return g == 0 ? h->stream : h->gstream[g]; }

struct ScopedTimer {
    hpf_handle* h;
    int which;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ScopedTimer(hpf_handle* h_, int w) : h(h_), which(w) {
        if (h->timing == 1) {
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, h->cur_stream);
        }
    }
    ~ScopedTimer() {
        if (h->timing == 1) {
            hipEventRecord(e1, h->cur_stream);
            h->spans.push_back({which, e0, e1});
        }
    }
};

// block-tree solver (hpf_block.hip)
int tree_find_ties(hpf_handle* h, const hpf_desc* d);    // spanning tree + loop-closing lines of the pattern (before any allocation)
int tree_build(hpf_handle* h, const hpf_desc* d);
int tree_plan_dump(const hpf_desc* d, const char* path);  // host-only planning run (hpf_tree_plan), no device touched
hpf::Tree& active_tree(hpf_handle* h);
bool tree_levels_fused(hpf_handle* h);                   // every elimination level of the current mode is one k_level launch
void tree_free(hpf_handle* h);
int tree_alloc_scenarios(hpf_handle* h);
int tree_fund_step(hpf_handle* h, bool only_active);     // fundamental pf Newton step on the tree (2x2 blocks)
int tree_newton_step(hpf_handle* h, bool only_active);   // assembles, eliminates, back-substitutes -> d_f holds the step
int tree_newton_step_bordered(hpf_handle* h, bool only_active);   // the same for a network with loop-closing lines (h->n_ties > 0)
int ensure_blas(hpf_handle* h);                                   // rocBLAS handle on first use (dense LU, border system)
int border_slots(const hpf_handle* h);                            // virtual scenario slots of the bordered step (behind the S_max real ones)
int tree_sel_build(hpf_handle* h, const hpf_desc* d);             // factor-once bordered step: P, forward pairs, block-product jobs, buffers (after tree_build)

}  // namespace hpf
