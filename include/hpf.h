/*
 * libhpf — MI355X (gfx950) harmonic power-flow Newton-Raphson hot path, C ABI.
 *
 * This is the drop-in boundary for ONE path of pweigmann/harmonic-power-flow: the NR loop of
 * `hpf()` in `Harmonic Power Flow/hcne_generalized.py` (HG).  The reference has no FFI of its own (its hot path
 * sits behind plain Python functions, HG:360-560); each entry point below names the reference function it
 * replaces.  The Python host (`harmonic-power-flow_amd/`) binds these symbols with `ctypes` and re-exports the
 * reference's own call shapes (`hpf`, `pf`, `harmonic_mismatch`, `build_harmonic_jacobian`, ...).
 *
 * Conventions
 *   - return 0 on success; < 0 invalid argument / wrong state; > 0 device-side failure (HIP, rocSOLVER, singular
 *     pivot): see hpf_strerror().  Nothing throws across the ABI.
 *   - all host buffers are caller-owned and copied in/out; `*_dev` entry points take device pointers instead.
 *   - the handle owns all device memory, one HIP stream and the rocBLAS handle; one handle per (process, device).
 *     A handle is not thread-safe; distinct handles are independent.
 *   - complex128 arrays are interleaved (re, im) doubles, NumPy layout.
 *   - all arithmetic is FP64.  Stacked index k = q*n + i (harmonic position q, bus i), HG:139-143.
 *   - scenarios: S independent load cases (P, Q per bus) share topology, admittances and Norton data; every
 *     per-scenario array is [S][...] with the scenario index slowest.
 */
#ifndef HPF_H
#define HPF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpf_handle hpf_handle;

enum {
    HPF_SOLVER_DENSE = 0,      /* dense real FP64 Jacobian, rocSOLVER getrf/getrs (any topology).  N*N < 2^31 (N <= 46 340): the batched
                                  32-bit entry points; larger systems (8 N^2 bytes per scenario: 21.6 GB at N = 51 998) go through
                                  rocSOLVER's 64-bit entry points one scenario after the other; HPF_E_NOMEM when they do not fit */
    HPF_SOLVER_BLOCK_TREE = 1  /* bus-major 2Hn x 2Hn block elimination along the feeder tree.  Radial networks directly; meshed
                                  networks as BFS spanning tree + k loop-closing lines, solved as a bordered system on top of the
                                  same tree factorisation (m = 2Hn x number of distinct endpoint buses of the loop-closing lines;
                                  two sweeps of the tree per scenario and Newton step + a selected inversion over the endpoints' root
                                  paths; uncoupled models and HPF_MESH_SEL=0: m + 2 right-hand sides as virtual scenarios in chunks
                                  of up to 1 024; m x m border system on rocSOLVER); m <= 16 384 and 2Hn <= 100, else HPF_E_TOPOLOGY */
};

enum {
    HPF_OK = 0,
    HPF_E_ARG = -1,        /* null pointer / out-of-range argument */
    HPF_E_STATE = -2,      /* call order violated (e.g. solve before loads were set) */
    HPF_E_TOPOLOGY = -3,   /* BLOCK_TREE: network not connected from bus 0 / pattern not symmetric / border of the loop-closing lines
                              beyond the stated bound */
    HPF_E_NOMEM = -4,
    HPF_E_HIP = 1,         /* HIP runtime error (hpf_last_error_detail has the hipError_t) */
    HPF_E_ROCSOLVER = 2,   /* rocBLAS / rocSOLVER status != success */
    HPF_E_SINGULAR = 3     /* a factorisation met an exactly zero pivot */
};

/* Model description (host pointers, copied by hpf_create).  Produced by the Python ingest from the reference's CSV
 * formats: bus/line CSVs -> (n, m, c) HG:113-128; per-harmonic admittances HG:132-171 stored as one CSR pattern
 * shared by all harmonics; Norton equivalents HG:278-310 per device type. */
typedef struct hpf_desc {
    int32_t n;                 /* buses                                                            HG:126 */
    int32_t m;                 /* 0-based index of first nonlinear bus (n if none)                 HG:122-125 */
    int32_t c;                 /* number of PV buses + 1                                           HG:127 */
    int32_t Hn;                /* harmonics incl. fundamental (K+1)                                HG:584 */
    int32_t nnz;               /* stored entries of the shared admittance pattern (full diagonal present) */
    int32_t n_dev;             /* nonlinear device types                                           HG:285 */
    int32_t coupled;           /* 1: Y_N is Hn x Hn per device; 0: Y_N is a length-Hn vector       HG:301-308 */
    int32_t solver;            /* HPF_SOLVER_*                                                             */
    int32_t device;            /* HIP device ordinal                                                       */
    int32_t max_scenarios;     /* capacity S_max >= 1                                                      */
    const int32_t* rowptr;     /* [n+1]                                                                    */
    const int32_t* col;        /* [nnz], ascending inside a row                                            */
    const double*  Yval;       /* [Hn][nnz] complex128                                                     */
    const int32_t* dev_of_bus; /* [n]: device type of a nonlinear bus, -1 for linear buses                 */
    const double*  Y_N;        /* coupled: [n_dev][Hn][Hn] complex128 (row = harmonic of injected current,
                                  col = harmonic of voltage, HG:304,432); uncoupled: [n_dev][Hn]           */
    const double*  I_N;        /* [n_dev][Hn] complex128                                                   */
} hpf_desc;

/* Per-scenario result record, also the payload of the multi-GPU statistics gather (24 bytes). */
typedef struct hpf_stat {
    int32_t n_iter;            /* harmonic NR iterations performed                       HG:542 */
    int32_t flags;             /* bit0 converged (err <= thresh), bit1 hit max_iter, bit2 non-finite mismatch,
                                  bit3 BLOCK_TREE: a static 4x4 pivot block amplified beyond the limit during the solve,
                                  bit4 the scenario was repeated with partial pivoting (its result is the repeat's),
                                  bit5 the pivoted elimination met an exactly zero pivot (hpf_solve returns HPF_E_SINGULAR) */
    double  err;               /* final ||f||_inf                                        HG:389 */
    double  thd_max;           /* max over buses of THD_F                                HG:566-568 */
} hpf_stat;

int  hpf_create(hpf_handle** out, const hpf_desc* d);
/* hpf_create with build switches for THIS handle: `options` = "NAME=value NAME=value ..." (separators: space, comma, semicolon; NULL or "" = none)
 * with the HPF_* names listed at hpf_set_option below ("Switches read by hpf_create").  The process environment is consulted for the same names
 * ONLY when the process opts in with HPF_ENV_SWITCHES=1 (A/B tooling, the test-suite): without it nothing a handle computes depends on
 * environment variables.  hpf_create(out, d) = hpf_create_opts(out, d, NULL). */
int  hpf_create_opts(hpf_handle** out, const hpf_desc* d, const char* options);
int  hpf_destroy(hpf_handle* h);
const char* hpf_strerror(int code);
int  hpf_last_error_detail(const hpf_handle* h);      /* hipError_t / rocblas_status / pivot index of the last >0 code */
int  hpf_version(void);

/* Sizes: N = 2*n*Hn - 1 - c unknowns of the harmonic NR (HG:388,397); Nf = 2*n - 1 - c of the fundamental NR. */
int  hpf_num_unknowns(const hpf_handle* h);
int  hpf_num_unknowns_fund(const hpf_handle* h);
/* Scenario counts, so that a host binding sizes the per-batch output arrays below from the LIBRARY and not from a mirror of its own: S = the
 * current batch (set by hpf_set_loads / hpf_set_state; 0 when the handle holds no batch, e.g. right after hpf_create or hpf_solve_queue) and the
 * capacity S_max of hpf_create.  Every per-batch output ([S][...] below) is written for exactly hpf_num_scenarios() scenarios; per-batch calls on a
 * handle without a batch return HPF_E_STATE. */
int  hpf_num_scenarios(const hpf_handle* h);
int  hpf_max_scenarios(const hpf_handle* h);
/* BLOCK_TREE: number of elimination levels of the dense tree (= factor-kernel launches per Newton step and scenario group;
 * pass-through buses are contracted first in the default mode) / of back-substitution levels; 0 for DENSE. */
int  hpf_tree_levels(const hpf_handle* h);
int  hpf_tree_depths(const hpf_handle* h);

/* Loads P,Q [S][n] in p.u. (buses.P / buses.Q of HG:197,372).  Sets the active scenario count S. */
int  hpf_set_loads(hpf_handle* h, int n_scen, const double* P, const double* Q);
/* Voltages Vm,Va [S][Hn*n] (the V DataFrame of HG:174-184, signed magnitudes allowed).  NULL,NULL -> the reference's
 * initial values (1 p.u. at h=1, 0.1 p.u. above, angle 0; init_voltages HG:174-184) for all S scenarios. */
int  hpf_set_state(hpf_handle* h, int n_scen, const double* Vm, const double* Va);
int  hpf_get_state(hpf_handle* h, double* Vm, double* Va);      /* raw: signed, un-wrapped; host applies HG:545-549 */

/* harmonic_mismatch (HG:360-390) incl. current_balance / current_injections (HG:313-357) for all S scenarios.
 * f [S][N] (may be NULL), err [S] (may be NULL). */
int  hpf_mismatch(hpf_handle* h, double* f, double* err);
/* build_harmonic_jacobian (HG:401-473) of scenario `scen`, written as a dense column-major N x N matrix
 * (parity / debugging; the solver consumes the device copy directly). */
int  hpf_jacobian(hpf_handle* h, int scen, double* J_colmajor);
/* The Jacobian the reference's hpf() returns (HG:537,560): the one of the LAST iteration of the last hpf_solve, i.e. built at the
 * state scenario `scen`'s last Newton step started from (needs option "keep_previous_state" = 1 before hpf_solve; the current state
 * is left untouched). */
int  hpf_jacobian_last(hpf_handle* h, int scen, double* J_colmajor);
/* build_harmonic_jacobian (HG:401-473) in the form the reference returns it from build_harmonic_jacobian and from hpf() (HG:469-472,
 * HG:560): the stacked real matrix [[dP/dth dP/dV] [Re dI/dth Re dI/dV] [dQ/dth dQ/dV] [Im dI/dth Im dI/dV]] as CSR -- indptr [N+1],
 * indices [nnz] (ascending inside a row), data [nnz] = the three arrays of scipy.sparse.csr_matrix.  Assembled on the device entry by
 * entry straight into the CSR arrays (no dense N x N anywhere: 14.7 MB at the 1 000-bus x 26-harmonic feeder where the dense copy is
 * 21.6 GB).  The pattern is a property of the model (admittance pattern + Norton coupling: what the reference's block_diag / lil
 * construction stores; entries that only vanish by cancellation at a particular state stay stored); hpf_jacobian_nnz sizes the arrays.
 * indptr / indices may be NULL on repeated calls (values only).  HPF_E_ARG if nnz would not fit 32-bit indices.
 * hpf_jacobian_csr_last: at the state the scenario's last Newton step started from, like hpf_jacobian_last. */
int  hpf_jacobian_nnz(hpf_handle* h, int64_t* nnz);
int  hpf_jacobian_csr(hpf_handle* h, int scen, int32_t* indptr, int32_t* indices, double* data);
int  hpf_jacobian_csr_last(hpf_handle* h, int scen, int32_t* indptr, int32_t* indices, double* data);
/* fund_mismatch + build_jacobian of the fundamental power flow (HG:195-223) for scenario `scen`: f [Nf], J [Nf*Nf]. */
int  hpf_fund_mismatch(hpf_handle* h, double* f, double* err);
int  hpf_fund_jacobian(hpf_handle* h, int scen, double* J_colmajor);

/* pf (HG:244-275): fundamental NR from the current state for all S scenarios; leaves the harmonic rows untouched.
 * n_iter [S], err [S] may be NULL.  err_hist [S][max_iter] (HG:264, may be NULL). */
int  hpf_fund_pf(hpf_handle* h, double thresh, int max_iter, int* n_iter, double* err, double* err_hist);

/* The NR loop of hpf (HG:530-542) from the current state: initial mismatch, then while err > thresh and
 * n_iter < max_iter: Jacobian -> solve -> update -> mismatch.  Scenarios that satisfy the stop rule freeze.
 * n_iter [S], err [S], err_hist [S][max_iter+1] (initial + one per iteration; unused tail = NaN) may be NULL.
 * BLOCK_TREE (static pivot order on the matrix cores): every 4x4 pivot block is watched; a scenario in which one amplifies
 * rounding errors by more than the limit (option "pivot_growth_limit_log10", default 10), or whose mismatch turns non-finite,
 * is repeated from the state the call was entered with, with partial pivoting over the whole bus block (flags bit3 / bit4).
 * Returns HPF_E_SINGULAR (detail = scenario; outputs are still written) if a pivoted elimination met an exactly zero pivot --
 * DENSE: rocSOLVER info > 0, BLOCK_TREE: the pivoted wave Gauss-Jordan. */
int  hpf_solve(hpf_handle* h, double thresh, int max_iter, int* n_iter, double* err, double* err_hist);

/* A sweep of n_total scenarios through the handle's S_max slots -- the reference's counterpart is one hpf() call per load case (HG:511:
 * other buses.P / buses.Q, HG:197,372).  P, Q [n_total][n] in p.u.  Every scenario: the reference's start (HG:174-184), pf (HG:244-275) with
 * (thresh_f, max_iter_f), the harmonic NR of hpf_solve with (thresh, max_iter).  Radial BLOCK_TREE handles keep all loads and pf seeds in HBM
 * and, between chunks of Newton iterations (option "queue_chunk", default 4), harvest the scenarios that met the stop rule and put the next
 * pending scenarios into the freed slots, so the handle stays full until the queue drains; every scenario's result is bit-identical to its
 * solve alone (the arithmetic of a scenario does not depend on its slot).  Other handles (DENSE, meshed networks, pivoted mode) run waves of
 * S_max scenarios.  Outputs (host, may be NULL; Vm and Va together): stats [n_total], raw voltages Vm, Va [n_total][Hn*n] (stacked order,
 * signed / un-wrapped like hpf_get_state).  In the queued mode a scenario flagged by the static-pivot monitor (flags bit 3) or whose mismatch
 * turned non-finite (flags bit 2) is reported, not repeated: solve it again with hpf_solve (which repeats exactly those with partial pivoting).  Afterwards the handle holds no batch: set loads and state before per-batch calls. */
int  hpf_solve_queue(hpf_handle* h, int n_total, const double* P, const double* Q, double thresh_f, int max_iter_f, double thresh,
                     int max_iter, hpf_stat* stats, double* Vm, double* Va);

/* Per-iteration state dump for trajectory diffing against the oracle / the reference (the reference's analogue is the JSON log of
 * every iterate, hcne_based_on_fuchs.py:186,370-372): while set, hpf_solve writes the voltages after iteration k (k = 0: the
 * state it was entered with) of every scenario to Vm_traj / Va_traj [S][cap][Hn*n] (caller-owned host arrays, stacked order
 * q*n + i; frozen scenarios repeat their last state; iterations >= cap are not recorded).  The solve then synchronises with the
 * host after every iteration.  NULL, NULL, 0 switches it off. */
int  hpf_set_trace(hpf_handle* h, double* Vm_traj, double* Va_traj, int cap);

/* One unconditional NR iteration (Jacobian -> solve -> update -> mismatch) for all S scenarios, repeated `iters`
 * times, no host synchronisation inside (throughput measurement; update_harmonic_state_vec HG:476-479 +
 * update_harmonic_voltages HG:482-485).  Requires a valid mismatch (hpf_mismatch or hpf_solve first). */
int  hpf_iterate(hpf_handle* h, int iters);

/* update_harmonic_state_vec (HG:476-479) as a standalone, stateless call like the reference's: dx = J^-1 f for a dense
 * column-major N x N Jacobian supplied by the caller (rocSOLVER LU, partial pivoting); the caller forms x - dx.  N * N >= 2^31 goes through
 * rocSOLVER's 64-bit entry points; HPF_E_NOMEM (before anything is allocated) when 8 N^2 bytes do not fit the device's free memory. */
int  hpf_dense_solve(int device, int N, const double* J_colmajor, const double* f, double* dx);
/* The same call for the Jacobian in the form the reference passes it (HG:478: the scipy CSR matrix build_harmonic_jacobian returns, HG:469-472),
 * at every size: indptr [N + 1], indices [nnz], data [nnz] of the N x N matrix in the reference's stacked row / column order, N = 2 n Hn - 1 - c
 * (n buses, c = PV buses + 1, Hn harmonics: the numbering is a function of these three alone), f [N] -> dx [N] = J^-1 f.  The entries are
 * scattered on the device into bus-major 2 Hn x 2 Hn blocks and eliminated along the feeder tree (partial pivoting inside a bus block; off-diagonal
 * blocks may be dense); no N x N array exists on host or device (65 MB of blocks at 1 000 buses x 26 harmonics, where the dense matrix is
 * 21.6 GB).  Duplicate (row, column) entries add up like scipy's.  A MESHED bus graph (spanning tree + loop-closing lines) is solved as a bordered
 * system: the tree part is factorised once, a selected inversion over the root paths of the lines' endpoint buses gives the m x m border matrix
 * (m = 2 Hn x number of distinct endpoint buses <= 16 384; rocSOLVER LU), one more right-hand-side sweep the solution (1 000 buses x 26 harmonics
 * + 20 lines: 27 ms, 1e-12 of the step from SuperLU).  HPF_E_TOPOLOGY when the bus graph is not connected from bus 0, the block pattern is not
 * symmetric (a block (i, j) without (j, i)) or the border exceeds that bound -- use hpf_dense_solve where it fits --, HPF_E_ARG for 2 Hn > 128 or an
 * inconsistent CSR, HPF_E_SINGULAR when a bus block or the border system has no pivot.  Stateless like the reference's function: no handle.
 * (env HPF_SPARSE_INFO=1 prints its phase times to stderr.) */
int  hpf_sparse_solve(int device, int n, int c, int Hn, const int32_t* indptr, const int32_t* indices, const double* data, const double* f,
                      double* dx);

/* Per-scenario statistics after hpf_solve; `thd_max` from get_THD (HG:563-572) evaluated on device. */
int  hpf_get_stats(hpf_handle* h, hpf_stat* stats /* [S] host */);
int  hpf_get_stats_dev(hpf_handle* h, void* stats_dev /* [S] hpf_stat, device memory of the caller (RCCL gather) */);

/* Diagnostics: with env HPF_DEBUG_ABLATE & 16 the BLOCK_TREE factor kernel records shader-cycle stamps per (scenario, bus):
 * out[(s*n + k)*8 + 0..5] = assembly, packed sub-phases, packed Gauss-Jordan split, MFMA Gauss-Jordan, packed wave-0 roles,
 * Schur push (tools/stamps.py decodes them; -DHPF_FACTOR_STAMPS build only); [6] dense children,
 * [7] nonlinear bus.  Timing-only; never read by any kernel. */
int  hpf_debug_stamps(hpf_handle* h, long long* out, int count);

/* Options.  "block_pivoting" (BLOCK_TREE only): 0 (default) inverts the 2Hn x 2Hn bus blocks on the FP64 matrix cores with
 * a static pivot order (4x4 blocks = two harmonics, lane-parallel cofactor inverse), after contracting pass-through buses and
 * with per-model constant inverses for nonlinear leaf buses; 1 uses wave-level Gauss-Jordan with partial pivoting over the
 * whole block on the uncontracted tree (slower, for networks whose bus blocks are not block-diagonally dominant).  Env
 * HPF_GJ_MODE=0 selects the pivoted variant process-wide.
 * "pivot_growth_limit_log10" (0..300, default 10): the static-pivot monitor flags a scenario when |a_ij W_ji| of a pivot block
 * (a lower bound of its condition number) exceeds 10^value; "auto_repivot" (default 1): hpf_solve repeats flagged scenarios
 * with partial pivoting (0: they are only reported in hpf_stat.flags).
 * "keep_previous_state" (default 0): hpf_solve keeps per scenario the state its last Newton step started from (hpf_jacobian_last).
 * "border_pivoting" (meshed BLOCK_TREE handles, default 0): the m x m border system of a bordered Newton step is factored WITHOUT pivoting
 * first (rocSOLVER's pivoted LU spends a third of such a step in tiny pivot-search kernels) and its solution checked against a kept copy of
 * the system; a relative residual above 1e-10, a zero pivot or a non-finite entry repeats it with partial pivoting (hpf_tree_census[11]
 * counts those).  1 = always the pivoted LU.
 * "queue_chunk" (1..16, default 4): Newton iterations between two harvest / refill rounds of hpf_solve_queue.
 * "scenario_groups" (1..8, default 4; at least 32 running scenarios per group): independent scenario pipelines on separate HIP streams -- group 0
 * on the handle's own stream (hpf_set_stream), the others on streams of the handle.  The runtime maps streams onto FOUR hardware queues: with a fifth
 * stream busy at the same time (the application's own work during a solve) two groups share a queue and serialise (1.25 instead of 0.90 ms per
 * step at the benchmark shape) -- such an application sets 3.
 * Switches read by hpf_create (diagnostics, A/B runs; from the option string of hpf_create_opts, and from the environment only with
 * HPF_ENV_SWITCHES=1; HPF_HOST_THREADS -- host threads of the tree planner, no effect on results -- is always read from the environment;
 * HPF_TREE_DUMP takes a path: environment only): HPF_DEBUG_ABLATE (timing-only ablation of factor-kernel phases: results INVALID), HPF_GJ_MODE=0
 * (the pivoted variant for every solve of the handle), HPF_LAZY=0 builds the elimination tree without lazy leaves (every
 * leaf writes its Schur complement; 1: only leaves directly under their dense parent), HPF_SLEAF=0 sends the nonlinear buses
 * whose dense children are all lazy leaves (super-leaves: bordered low-rank inverse) through Gauss-Jordan like every other bus,
 * HPF_LEAFBATCH=0 runs the lazy leaves one workgroup per (leaf, scenario) instead of 16 scenarios per workgroup on the matrix
 * cores, HPF_SLBACK=0 lets the super-leaves store their inverse for the per-scenario back sweep instead of keeping T^-1 only,
 * HPF_SLLAZY=0 lets every super-leaf push its Schur complement itself (no vector-only bordered buses, hence no nested ones: the
 * faster build up to about 24 live scenarios), HPF_SLNEST=0 keeps bordered buses below bordered buses on the
 * Gauss-Jordan path, HPF_FUSELEVEL=0 launches the scenario-batched
 * and the per-scenario workgroups of an elimination level separately (k_leaf_batch / k_sleaf_batch + k_factor_q instead of k_level),
 * HPF_LINBUNDLE=0 / HPF_LINTREE=0 run the 2x2 algebra of the linear subtrees height by height in one launch / in one launch per
 * height, HPF_CHAINBUNDLE=0 gives the contracted chains their own launches, HPF_COMPRESS=0 eliminates the Gauss-Jordan buses strictly
 * leaves first (no compress steps: one elimination level per unit of tree height; 5 - 8 % faster per step from about 384 live scenarios on, whose
 * levels fill the chip -- the steps are the default at every capacity so that a handle's Newton steps do not depend on its capacity), HPF_TREE_INFO=1 prints the tree statistics to stderr,
 * HPF_QUEUE_INFO=1 prints the phase times of hpf_solve_queue to stderr, HPF_BORDER_SLOTS=n caps the virtual scenario slots a meshed handle
 * allocates for its bordered step (default 1 024, at least 16: the m + 1 right-hand sides run in chunks of that many; only with HPF_MESH_SEL=0),
 * HPF_MESH_SEL=0 runs the bordered step of a meshed handle in its form of rounds 2 - 4 (the m unit right-hand sides as virtual scenarios through
 * the tree kernels, the tree re-factorised for each) instead of the factor-once form (one sweep + a selected inversion over the tie endpoints'
 * root paths, whose buses the planner then keeps as plain Gauss-Jordan buses; coupled models), HPF_MESH_BATCH_GB=x bounds the memory of
 * the per-scenario buffers of that form (default 48, and not more than half of the device's free memory: as many scenarios per batch as fit, at least one), HPF_BORDER_GJ=n solves border systems of up to n endpoint buses (default 96) by a
 * block Gauss-Jordan elimination with the library's own block-product kernel and larger ones by rocSOLVER's LU (0: always rocSOLVER;
 * HPF_BORDER_GJ_MFMA=0 inverts its diagonal blocks on the vector units instead of the matrix cores, HPF_BORDER_PIVLIM=x sets the amplification of
 * a 4 x 4 pivot block beyond which such a system goes to the pivoted LU, default 1e3; HPF_BORDER_INFO=1 prints every border solve's residual),
 * HPF_FUSEBACK=0 launches the back sweep's scenario-batched workgroups (bordered buses, leaves) after the last depth instead of inside the
 * depths' launches (k_level_back: groups of up to HPF_FUSEBACK_MAX = 32 scenarios, blocks of 52),
 * HPF_GROUPS=n presets "scenario_groups".  Every switch selects a path with the same Newton steps (tests/test_gpu_robustness.py). */
int  hpf_set_option(hpf_handle* h, const char* name, int value);

/* Stream plumbing: run on a caller stream (e.g. torch's current stream) instead of the handle's own; NULL restores. */
int  hpf_set_stream(hpf_handle* h, void* hip_stream);
int  hpf_sync(hpf_handle* h);

/* Kernel timing with HIP events on the handle's stream, accumulated since the last reset.
 * which: 0 mismatch kernel, 1 Jacobian assembly kernels (DENSE only; BLOCK_TREE assembles inside the factor kernel),
 * 2 linear solve (DENSE: getrf+getrs, one span per step; BLOCK_TREE: one span per launch of a factor kernel OTHER than the
 *   general one: k_leaf_batch, k_sleaf_batch, the leaf-only k_factor_q<B,true>, the pivoted / generic kernels),
 * 3 state update, 4 back-substitution sweep (BLOCK_TREE only; one span per Newton step and scenario group),
 * 5 BLOCK_TREE: one span per launch of the dominant factor kernel: k_level<B> (b <= 52: one launch per elimination level, every dense bus)
 *   where hpf_tree_census reports fused levels, else the general kernel k_factor_q<B,false>,
 * 6 the same launches on the DEVICE clock: last workgroup end - first workgroup start (wall_clock64 stamps written by the kernel
 *   while timing is enabled) -- what rocprofv3 --kernel-trace reports as the kernel's duration; a HIP-event span additionally
 *   holds the event packets and the queue gaps around a ~25 us kernel.
 * Returns total milliseconds in *ms and the number of timed spans in *launches. */
int  hpf_timing_enable(hpf_handle* h, int on);   /* 1: HIP-event spans (classes 0..5) + device stamps (6); 2: device stamps only --
                                                    no event packets between the kernels, the launches run exactly as untimed; 0: off */
int  hpf_timing_get(hpf_handle* h, int which, double* ms, int64_t* launches);
int  hpf_timing_reset(hpf_handle* h);
/* FP64 flop count of the span `which == 2` for ONE scenario and ONE Newton step (roofline numerator):
 * DENSE 2/3 N^3 + 2 N^2; BLOCK_TREE (b = 2 Hn) per Gauss-Jordan bus 2 b^3 + 2 b^2 + b^2 per dense child (+ 8 b^2 non-root),
 * per constant-inverse leaf 10 b^2 (+ 8 b^2 push unless lazy: then only G w), per lazy leaf 4 b^2 + 4 b^2 per parent for the
 * rebuild, per super-leaf (m = 2 + 2 L border unknowns) 2 m^3 + 8 b^2 ceil(m/4) + 6 b^2 + 6 b m. */
double hpf_solve_flops(const hpf_handle* h);
/* Algorithmic HBM bytes of the same span (one scenario, one Newton step): BLOCK_TREE every Schur complement once out and once
 * in (lazy leaves: 2x2 core + G w instead), every Gauss-Jordan / super-leaf inverse once out, per-scenario bus operands; DENSE
 * the Jacobian out and through getrf. */
double hpf_solve_bytes(const hpf_handle* h);
/* ... and of the span `which == 4` (BLOCK_TREE back sweep: Gauss-Jordan inverses in, w, A(k,parent), x in / out); 0 for DENSE. */
double hpf_back_bytes(const hpf_handle* h);
/* Roofline model of ONE kernel class (timing span `which`, see hpf_timing_get): algorithmic HBM bytes and FP64 flops of all its
 * launches of one Newton step for ONE scenario, and the number of launches per step and scenario group.  which == 5: the
 * general multi-wave factor kernel k_factor_q<B,false> alone (Gauss-Jordan buses, non-batched super-leaves; the buses of the
 * scenario-batched kernels k_leaf_batch / k_sleaf_batch and of the leaf-only instantiation are NOT in it); which == 2: the
 * whole factor sweep (= hpf_solve_bytes / hpf_solve_flops); which == 4: the back sweep.  Other classes: HPF_E_ARG. */
int  hpf_kernel_model(const hpf_handle* h, int which, double* bytes, double* flops, int* launches);
/* Census of the BLOCK_TREE elimination tree (diagnostic; which kernel takes which bus).  counts[0..8]: buses with a dense b x b
 * block (the rest lives in the 2x2 algebra of the linear subtrees / contracted chains), Gauss-Jordan buses (k_factor_q<B,false>),
 * constant-inverse leaves, of which lazy (vector-only, k_leaf_batch), bordered buses (super-leaves, m x m core), of which nested
 * (bordered children below them), elimination levels, back-sweep depths, tie lines of a meshed network, [9] 1 if every elimination
 * level is ONE launch (k_level: scenario-batched and per-scenario workgroups in one grid; timing class 5 then covers it) -- blocks of 52
 * in the default mode; smaller blocks only when every level has scenario-batched workgroups (levels without them run k_factor_q's own grid),
 * [10] compress steps (Gauss-Jordan buses eliminated before their tallest dense child: levels counts the shortened chain),
 * [11] border systems of a meshed network that were repeated with the pivoted LU since hpf_create (option "border_pivoting"),
 * [12] border unknowns of a meshed network (2 Hn x distinct endpoint buses of the loop-closing lines), [13] buses on the endpoints' root paths
 * (kept as plain Gauss-Jordan buses by the factor-once bordered step; 0: virtual-sweep form), [14] form of the bordered step: 0 virtual sweeps,
 * 1 factor-once with rocSOLVER's LU of the border system, 2 factor-once with the block Gauss-Jordan solve.
 * HPF_E_STATE for DENSE. */
int  hpf_tree_census(const hpf_handle* h, int* counts, int n_counts);
/* Wall-clock milliseconds hpf_create spent: ms[0] total, [1] planning the elimination trees on the host (classification of the buses,
 * per-model constant images: complex inversions on host threads, env HPF_HOST_THREADS), [2] uploading them, [3] allocating the
 * per-scenario state.  (The host-side ingest and admittance build happen before hpf_create, in the Python layer.) */
int  hpf_setup_times(const hpf_handle* h, double* ms, int n_ms);
/* Number of scenario groups (independent pipelines on separate HIP streams) a Newton step of `live` running scenarios is split into:
 * option "scenario_groups" bounded by a minimum group size; 1 for DENSE and for meshed networks. */
int  hpf_scenario_groups(const hpf_handle* h, int live);
/* Host-only planning run of the BLOCK_TREE elimination tree of a model (no device is touched, no handle, the process environment
 * is not modified): builds the contracted tree exactly as hpf_create would for a handle of d->max_scenarios scenarios (compress steps
 * are the default) and writes one line per dense bus (bus, dense parent, elimination level, back-sweep depth, kind, ...)
 * to `path` (replaced if it exists; tools/tree_plan.py reads it).  Returns the planning status: HPF_OK when the plan was written,
 * HPF_E_TOPOLOGY as hpf_create would return it, HPF_E_ARG when the file cannot be written.  A meshed model (spanning tree + loop-closing lines):
 * a comment line with the lines / endpoint buses / border size, and a last column that marks the buses on the endpoints' root paths, which the
 * factor-once bordered step keeps as plain Gauss-Jordan buses (1; -1 would be a planner fault).  (env HPF_TREE_DUMP=<file> makes hpf_create itself
 * write the same dump.) */
int  hpf_tree_plan(const hpf_desc* d, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* HPF_H */
