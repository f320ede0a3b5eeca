"""`DeviceModel`: owner of one libhpf handle (one network x harmonic set x Norton data on one GPU) and the thin
NumPy-facing wrappers around the C ABI.  All heavy work happens in HIP kernels / rocSOLVER; this file only marshals
arrays."""
import ctypes as C

import numpy as np

from . import _lib


def _dp(a):
    return a.ctypes.data_as(_lib.c_dbl_p)


def _ip(a):
    return a.ctypes.data_as(_lib.c_int_p)


def is_radial(n, rowptr, col):
    """True if the admittance pattern is a tree spanning all buses (n-1 undirected edges, connected from bus 0)."""
    nnz = len(col)
    if nnz != n + 2 * (n - 1):
        return False
    seen = np.zeros(n, dtype=bool)
    seen[0] = True
    stack = [0]
    while stack:
        i = stack.pop()
        for j in col[rowptr[i]:rowptr[i + 1]]:
            if not seen[j]:
                seen[j] = True
                stack.append(int(j))
    return bool(seen.all())


class DeviceModel:
    """One libhpf handle of capacity `max_scenarios`.  The capacity is NOT a build parameter of the block tree (it was in rounds 3-4): every
    handle eliminates the Gauss-Jordan skeleton with compress steps (fewer, wider elimination levels), so the Newton steps of a
    scenario -- and the iteration count of a solver-sensitive case -- are bit-identical in handles of any capacity and any batch size.  A sweep of
    several hundred LIVE scenarios runs 5 - 8 % faster per step without the steps: `options="HPF_COMPRESS=0"` (rounding-level different Newton
    steps, like every other tree-build switch).  `tree_census()["compress_steps"]` reports what a handle uses."""

    def __init__(self, n, m, c, harmonics, rowptr, col, Yval, dev_of_bus, Y_N, I_N, n_dev, coupled,
                 solver="auto", device=0, max_scenarios=1, assembly_only=False, options=None):
        """options: build switches of THIS handle, "HPF_LAZY=0 HPF_SLEAF=1 ..." (hpf_create_opts; include/hpf.h lists them).  The process
        environment is consulted for the same names only when HPF_ENV_SWITCHES=1 is set (A/B tooling, the test-suite)."""
        lib = _lib.load()
        self.lib = lib
        self.n, self.m, self.c = int(n), int(m), int(c)
        self.harmonics = list(harmonics)
        self.Hn = len(self.harmonics)
        self.coupled = bool(coupled)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.Yval = np.ascontiguousarray(Yval, dtype=np.complex128)
        self.dev_of_bus = np.ascontiguousarray(dev_of_bus, dtype=np.int32)
        self.Y_N = np.ascontiguousarray(Y_N, dtype=np.complex128)
        self.I_N = np.ascontiguousarray(I_N, dtype=np.complex128)
        N = 2 * self.n * self.Hn - 1 - self.c
        if solver == "auto":
            # radial feeders: block-tree elimination; meshed networks (spanning tree + loop-closing lines) of 32 buses and more: the block-tree
            # path's bordered step (tried first, dense if the library refuses the topology: border beyond 16 384 unknowns) -- 0.3 - 1.1 ms per
            # iteration where the dense LU takes 2 - 250 (tools/mesh_vs_dense.py: N = 478 ... 15 598; rounds 2 - 4 drew the line at N > 8 192);
            # smaller networks (the reference's net1 - net3): dense rocSOLVER LU
            if is_radial(self.n, self.rowptr, self.col):
                solver = "block_tree" if self.n >= 32 else "dense"
            else:
                solver = "block_tree_or_dense" if self.n >= 32 else "dense"
        self._solver_request = solver
        if solver == "block_tree_or_dense":
            solver = "block_tree"
        self.solver = solver
        # (assembly_only: a handle used for hpf_mismatch / hpf_jacobian_csr alone never allocates the dense Jacobian)
        if solver == "dense" and not assembly_only and 8 * N * N * int(max_scenarios) > 240e9:
            raise ValueError("dense solver: N = %d unknowns x %d scenarios need %.0f GB for the Jacobians alone; radial feeders and feeders "
                             "with loop-closing lines of this size use solver='block_tree'" % (N, max_scenarios, 8e-9 * N * N * max_scenarios))
        d = _lib.hpf_desc()
        d.n, d.m, d.c, d.Hn, d.nnz = self.n, self.m, self.c, self.Hn, len(self.col)
        d.n_dev, d.coupled = int(n_dev), int(self.coupled)
        d.solver = {"dense": _lib.SOLVER_DENSE, "block_tree": _lib.SOLVER_BLOCK_TREE}[solver]
        d.device, d.max_scenarios = int(device), int(max_scenarios)
        d.rowptr, d.col = _ip(self.rowptr), _ip(self.col)
        d.Yval = self.Yval.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
        d.dev_of_bus = _ip(self.dev_of_bus)
        d.Y_N = self.Y_N.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
        d.I_N = self.I_N.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
        self._h = C.c_void_p()
        opts = options.encode() if options else None
        rc = lib.hpf_create_opts(C.byref(self._h), C.byref(d), opts)
        if rc == -3 and self._solver_request == "block_tree_or_dense" and 8 * N * N * int(max_scenarios) <= 240e9:
            # too many loop-closing lines for the bordered block-tree step: the dense GPU path (still no CPU path anywhere)
            self.solver = "dense"
            d.solver = _lib.SOLVER_DENSE
            rc = lib.hpf_create_opts(C.byref(self._h), C.byref(d), opts)
        _lib.check(rc, None, "hpf_create")
        self.S_max = int(lib.hpf_max_scenarios(self._h))
        self.N = lib.hpf_num_unknowns(self._h)
        self.Nf = lib.hpf_num_unknowns_fund(self._h)
        self.n_levels = lib.hpf_tree_levels(self._h)
        self.n_depths = lib.hpf_tree_depths(self._h)

    # -- lifetime ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.hpf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, code, where):
        _lib.check(code, self._h, where)

    @property
    def S(self):
        """Scenarios of the current batch -- asked from the library (hpf_num_scenarios), never mirrored here: the per-batch entry points of the
        C ABI write exactly that many rows, so every output array below is sized from it (a host-side copy that disagreed with the library's
        count was a heap overrun in round 4, DESIGN_LOG.md)."""
        return int(self.lib.hpf_num_scenarios(self._h)) if getattr(self, "_h", None) else 0

    def _batch(self, where):
        S = self.S
        if S < 1:
            raise _lib.HpfError(-2, 0, where)          # HPF_E_STATE: no batch in the handle (fresh handle, or after solve_queue)
        return S

    # -- state ---------------------------------------------------------------------------------------------
    def set_loads(self, P, Q):
        P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
        Q = np.ascontiguousarray(np.atleast_2d(Q), dtype=np.float64)
        assert P.shape == Q.shape and P.shape[1] == self.n
        self._chk(self.lib.hpf_set_loads(self._h, P.shape[0], _dp(P), _dp(Q)), "hpf_set_loads")

    def set_state(self, Vm=None, Va=None, n_scen=None):
        if Vm is None:
            S = n_scen or self.S or 1
            self._chk(self.lib.hpf_set_state(self._h, S, None, None), "hpf_set_state")
            return
        Vm = np.ascontiguousarray(np.atleast_2d(Vm), dtype=np.float64)
        Va = np.ascontiguousarray(np.atleast_2d(Va), dtype=np.float64)
        assert Vm.shape == Va.shape and Vm.shape[1] == self.n * self.Hn
        self._chk(self.lib.hpf_set_state(self._h, Vm.shape[0], _dp(Vm), _dp(Va)), "hpf_set_state")

    def get_state(self):
        Vm = np.empty((self._batch("hpf_get_state"), self.n * self.Hn))
        Va = np.empty_like(Vm)
        self._chk(self.lib.hpf_get_state(self._h, _dp(Vm), _dp(Va)), "hpf_get_state")
        return Vm, Va

    # -- kernels -------------------------------------------------------------------------------------------
    def mismatch(self, fund=False, want_f=True):
        N = self.Nf if fund else self.N
        S = self._batch("hpf_mismatch")
        f = np.empty((S, N)) if want_f else None
        err = np.empty(S)
        fn = self.lib.hpf_fund_mismatch if fund else self.lib.hpf_mismatch
        self._chk(fn(self._h, _dp(f) if want_f else None, _dp(err)), "hpf_mismatch")
        return f, err

    def jacobian(self, scen=0, fund=False):
        N = self.Nf if fund else self.N
        J = np.empty((N, N), order="F")
        fn = self.lib.hpf_fund_jacobian if fund else self.lib.hpf_jacobian
        self._chk(fn(self._h, int(scen), J.ctypes.data_as(_lib.c_dbl_p)), "hpf_jacobian")
        return J

    def jacobian_last(self, scen=0):
        """Jacobian of the last iteration of the last solve (HG:537,560); needs set_option("keep_previous_state", 1) before it."""
        J = np.empty((self.N, self.N), order="F")
        self._chk(self.lib.hpf_jacobian_last(self._h, int(scen), J.ctypes.data_as(_lib.c_dbl_p)), "hpf_jacobian_last")
        return J

    def jacobian_nnz(self):
        nnz = C.c_int64()
        self._chk(self.lib.hpf_jacobian_nnz(self._h, C.byref(nnz)), "hpf_jacobian_nnz")
        return int(nnz.value)

    def jacobian_csr(self, scen=0, last=False):
        """build_harmonic_jacobian (HG:401-473) as the reference returns it: scipy CSR of the stacked real matrix (HG:469-472), assembled
        on the device straight into the CSR arrays (hpf_jacobian_csr; no dense N x N).  last=True: at the state the scenario's last
        Newton step started from (what hpf() returns, HG:537,560; needs set_option("keep_previous_state", 1) before the solve)."""
        import scipy.sparse as sp
        nnz = self.jacobian_nnz()
        indptr = np.empty(self.N + 1, dtype=np.int32)
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz, dtype=np.float64)
        fn = self.lib.hpf_jacobian_csr_last if last else self.lib.hpf_jacobian_csr
        self._chk(fn(self._h, int(scen), _ip(indptr), _ip(indices), _dp(data)), "hpf_jacobian_csr")
        J = sp.csr_matrix((data, indices, indptr), shape=(self.N, self.N))
        J._hpf_dims = (self.n, self.c, self.Hn)        # the numbering of its rows / columns (api.update_harmonic_state_vec reads it)
        return J

    def fund_pf(self, thresh=1e-6, max_iter=30):
        S = self._batch("hpf_fund_pf")
        n_iter = np.zeros(S, dtype=np.int32)
        err = np.empty(S)
        hist = np.empty((S, max(max_iter, 1)))
        self._chk(self.lib.hpf_fund_pf(self._h, float(thresh), int(max_iter), _ip(n_iter), _dp(err), _dp(hist)),
                  "hpf_fund_pf")
        return n_iter, err, hist[:, :max_iter]

    def solve(self, thresh=1e-4, max_iter=50, trace=False):
        """hpf_solve -> (n_iter [S], err [S], err_hist [S][max_iter+1]); with trace=True additionally the per-iteration states
        (Vm_traj, Va_traj) [S][max_iter+1][Hn*n] (entry k = state after iteration k; frozen scenarios repeat their last state)."""
        S = self._batch("hpf_solve")
        n_iter = np.zeros(S, dtype=np.int32)
        err = np.empty(S)
        hist = np.empty((S, max_iter + 1))
        if trace:
            Vt = np.full((S, max_iter + 1, self.n * self.Hn), np.nan)
            At = np.full_like(Vt, np.nan)
            self._chk(self.lib.hpf_set_trace(self._h, _dp(Vt), _dp(At), max_iter + 1), "hpf_set_trace")
        try:
            self._chk(self.lib.hpf_solve(self._h, float(thresh), int(max_iter), _ip(n_iter), _dp(err), _dp(hist)),
                      "hpf_solve")
        finally:
            if trace:
                self.lib.hpf_set_trace(self._h, None, None, 0)
        if trace:
            return n_iter, err, hist, Vt, At
        return n_iter, err, hist

    def solve_queue(self, P, Q, thresh_f=1e-6, max_iter_f=30, thresh=1e-4, max_iter=50, want_voltages=False):
        """hpf_solve_queue: every row of P, Q [n_scen][n] is one scenario (reference start, pf, harmonic NR); the handle's S_max slots are
        refilled with pending scenarios as running ones meet the stop rule.  -> records (n_iter, flags, err, thd_max) [n_scen]
        [, raw Vm, Va [n_scen][Hn*n]].  Leaves the handle without a batch (set_loads / set_state before the per-batch calls)."""
        P = np.ascontiguousarray(np.atleast_2d(P), dtype=np.float64)
        Q = np.ascontiguousarray(np.atleast_2d(Q), dtype=np.float64)
        assert P.shape == Q.shape and P.shape[1] == self.n
        n_scen = P.shape[0]
        st = (_lib.hpf_stat * n_scen)()
        Vm = Va = None
        if want_voltages:
            Vm = np.empty((n_scen, self.n * self.Hn))
            Va = np.empty_like(Vm)
        self._chk(self.lib.hpf_solve_queue(self._h, n_scen, _dp(P), _dp(Q), float(thresh_f), int(max_iter_f), float(thresh), int(max_iter),
                                           st, _dp(Vm) if want_voltages else None, _dp(Va) if want_voltages else None), "hpf_solve_queue")
        rec = np.frombuffer(st, dtype=[("n_iter", "<i4"), ("flags", "<i4"), ("err", "<f8"), ("thd_max", "<f8")]).copy()
        return (rec, Vm, Va) if want_voltages else rec

    def iterate(self, iters):
        self._chk(self.lib.hpf_iterate(self._h, int(iters)), "hpf_iterate")

    def sync(self):
        self._chk(self.lib.hpf_sync(self._h), "hpf_sync")

    def stats(self):
        st = (_lib.hpf_stat * self._batch("hpf_get_stats"))()
        self._chk(self.lib.hpf_get_stats(self._h, st), "hpf_get_stats")
        return np.array([(s.n_iter, s.flags, s.err, s.thd_max) for s in st],
                        dtype=[("n_iter", "i4"), ("flags", "i4"), ("err", "f8"), ("thd_max", "f8")])

    def stats_to_device(self, data_ptr):
        self._chk(self.lib.hpf_get_stats_dev(self._h, C.c_void_p(int(data_ptr))), "hpf_get_stats_dev")

    def set_option(self, name, value):
        self._chk(self.lib.hpf_set_option(self._h, name.encode(), int(value)), "hpf_set_option")

    def set_stream(self, stream_ptr):
        self._chk(self.lib.hpf_set_stream(self._h, C.c_void_p(int(stream_ptr)) if stream_ptr else None),
                  "hpf_set_stream")

    def timing(self, on=True):
        self._chk(self.lib.hpf_timing_enable(self._h, int(on)), "hpf_timing_enable")

    def timing_reset(self):
        self._chk(self.lib.hpf_timing_reset(self._h), "hpf_timing_reset")

    def timing_get(self):
        out = {}
        for which, name in enumerate(("mismatch", "jacobian", "solve", "update", "back", "gj", "gj_dev")):
            ms = C.c_double()
            cnt = C.c_int64()
            self._chk(self.lib.hpf_timing_get(self._h, which, C.byref(ms), C.byref(cnt)), "hpf_timing_get")
            out[name] = (ms.value, cnt.value)
        return out

    def setup_times(self):
        """Milliseconds hpf_create spent: total, tree planning on the host, tree uploads, per-scenario allocation (hpf_setup_times)."""
        ms = (C.c_double * 4)()
        self._chk(self.lib.hpf_setup_times(self._h, ms, 4), "hpf_setup_times")
        return dict(zip(("create_ms", "tree_plan_ms", "tree_upload_ms", "alloc_ms"), (float(v) for v in ms)))

    def scenario_groups(self, live):
        """Scenario groups (streams) a step of `live` running scenarios is split into (hpf_scenario_groups)."""
        return int(self.lib.hpf_scenario_groups(self._h, int(live)))

    def solve_flops(self):
        return float(self.lib.hpf_solve_flops(self._h))

    def tree_census(self):
        """Which kernel takes which bus of the block tree (hpf_tree_census)."""
        out = (C.c_int32 * 15)()
        self._chk(self.lib.hpf_tree_census(self._h, out, 15), "hpf_tree_census")
        names = ("dense_buses", "gauss_jordan", "const_leaves", "lazy_leaves", "bordered", "nested_bordered", "levels", "depths", "ties",
                 "fused_levels", "compress_steps", "border_repivots", "border_unknowns", "root_path_buses", "bordered_form")
        return dict(zip(names, (int(v) for v in out)))

    def solve_bytes(self):
        """Algorithmic HBM bytes of the linear-solve span of one scenario and one Newton step (hpf_solve_bytes)."""
        return float(self.lib.hpf_solve_bytes(self._h))

    def back_bytes(self):
        return float(self.lib.hpf_back_bytes(self._h))

    def kernel_model(self, which):
        """(algorithmic bytes, flops) of one kernel class per scenario and Newton step, launches per step and scenario group
        (hpf_kernel_model; which: "gj" = the general factor kernel k_factor_q<B,false>, "solve" = whole factor sweep, "back")."""
        by, fl, ln = C.c_double(), C.c_double(), C.c_int32()
        idx = {"solve": 2, "back": 4, "gj": 5}[which]
        self._chk(self.lib.hpf_kernel_model(self._h, idx, C.byref(by), C.byref(fl), C.byref(ln)), "hpf_kernel_model")
        return by.value, fl.value, ln.value
