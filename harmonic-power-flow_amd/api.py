"""The reference's call shapes (`Harmonic Power Flow/hcne_generalized.py`, HG) on top of the HIP library.

Same names, argument meaning, return objects and printed warnings as the reference: `init_network` HG:113,
`build_admittance_matrices` HG:132, `import_Norton_Equivalents` HG:278, `pf` HG:244, `hpf` HG:511,
`harmonic_mismatch` HG:360, `build_harmonic_jacobian` HG:401, `update_harmonic_state_vec` HG:476, `get_THD` HG:563,
plus the `solve` convenience wrapper the north-star asks for.  The reference reads module globals (`buses, m, n, c,
HARMONICS, base_*`); here they live in an explicit `Settings` object, with a module-level
default (`settings`) and a "current network" context filled by `init_network`, so reference-style scripts keep
working unchanged.

Everything numerical runs on the GPU through `DeviceModel`; nothing here falls back to the CPU.
"""
import atexit
import collections
import contextlib
import hashlib
import os

import numpy as np
import pandas as pd
import scipy.sparse as sp

import ctypes as C

from . import _lib, admittance, ingest
from .device import DeviceModel
from .settings import Settings

settings = Settings()          # module-level defaults, HG:578-593
_ctx = {"buses": None, "lines": None}


class AdmittanceSet:
    """Per-harmonic admittance matrices in device layout (shared CSR pattern).  `to_frame()` gives the reference's
    dense `Y_all` DataFrame."""

    def __init__(self, rowptr, col, Yval, harmonics, n):
        self.rowptr, self.col, self.Yval, self.harmonics, self.n = rowptr, col, Yval, list(harmonics), n
        self._key = None              # digest of the three arrays (handle cache), computed once per object

    def key(self):
        if self._key is None:
            self._key = _digest(self.rowptr, self.col, self.Yval)
        return self._key

    def to_frame(self):
        return admittance.to_dense_frame(self.rowptr, self.col, self.Yval, self.harmonics, self.n)

    def dense(self, h):
        q = self.harmonics.index(h)
        return sp.csr_matrix((self.Yval[q], self.col, self.rowptr), shape=(self.n, self.n)).toarray()


def _as_admittance(Y, harmonics, n):
    if isinstance(Y, AdmittanceSet):
        return Y
    rowptr, col, Yval = admittance.from_dense_frame(Y, harmonics, n)
    return AdmittanceSet(rowptr, col, Yval, harmonics, n)


def init_network(filename_buses, filename_lines, from_csv=True, settings=None):
    """HG:113-128."""
    st = settings or globals()["settings"]
    buses, lines, m, n, c = ingest.init_network(filename_buses, filename_lines, from_csv, st)
    _ctx["buses"], _ctx["lines"] = buses, lines
    return buses, lines, m, n, c


_ADMITTANCES = collections.OrderedDict()      # digest of the inputs -> AdmittanceSet: repeated hpf() calls on one network (HG:511) build it once


def build_admittance_matrices(buses, lines, harmonics):
    """HG:132-171 -> AdmittanceSet (call `.to_frame()` for the reference's DataFrame).  The result is a function of the line table, the bus
    shunts and the harmonics alone; the last four are kept (by a digest of exactly those inputs), so the reference's sweep -- hpf() per load
    case -- builds the matrices of a network once.  Treat an AdmittanceSet as read-only."""
    harmonics = list(harmonics)
    key = _digest(lines.fromID.to_numpy(), lines.toID.to_numpy(), lines.R.to_numpy(dtype=float), lines.X.to_numpy(dtype=float),
                  lines.G.to_numpy(dtype=float), lines.B.to_numpy(dtype=float), buses["X_sh"].to_numpy(dtype=float),
                  np.asarray(harmonics, dtype=np.int64))
    hit = _ADMITTANCES.pop(key, None)
    if hit is None:
        rowptr, col, Yval = admittance.build_admittance_csr(buses, lines, harmonics)
        hit = AdmittanceSet(rowptr, col, Yval, harmonics, len(buses))
    _ADMITTANCES[key] = hit
    while len(_ADMITTANCES) > 4:
        _ADMITTANCES.popitem(last=False)
    return hit


def import_Norton_Equivalents(buses, coupled, settings=None, ne_dir=None):
    """HG:278-310."""
    return ingest.import_Norton_Equivalents(buses, coupled, settings or globals()["settings"], ne_dir)


def init_voltages(buses, harmonics):
    """HG:174-184."""
    idx = pd.MultiIndex.from_product([list(harmonics), buses.index.values], names=["harmonic", "bus"])
    V = pd.DataFrame(np.zeros((len(harmonics) * len(buses), 2)), index=idx, columns=["V_m", "V_a"])
    V.iloc[:len(buses), 0] = 1
    V.iloc[len(buses):, 0] = 0.1
    return V


_FRAME_INDEX = {}


def _frame(Vm, Va, harmonics, n):
    key = (tuple(harmonics), int(n))
    idx = _FRAME_INDEX.get(key)                          # (a MultiIndex is immutable: the result frames of repeated calls share it; 0.5 ms per build)
    if idx is None:
        if len(_FRAME_INDEX) > 8:
            _FRAME_INDEX.clear()
        idx = _FRAME_INDEX[key] = pd.MultiIndex.from_product([list(harmonics), list(range(n))], names=["harmonic", "bus"])
    return pd.DataFrame({"V_m": np.asarray(Vm), "V_a": np.asarray(Va)}, index=idx)


def _device_model(buses, Y, NE, coupled, harmonics, solver="auto", max_scenarios=1, device=0, assembly_only=False, options=None):
    m, n, c = ingest.network_constants(buses)
    Hn = len(harmonics)
    if NE is None:
        dev = np.full(n, -1, dtype=np.int32)
        dev[m:] = 0
        Y_N = np.zeros((1, Hn, Hn) if coupled else (1, Hn), dtype=np.complex128)
        I_N = np.zeros((1, Hn), dtype=np.complex128)
        n_dev = 1
    else:
        dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, coupled, Hn)
    return DeviceModel(n, m, c, harmonics, Y.rowptr, Y.col, Y.Yval, dev, Y_N, I_N, n_dev, coupled,
                       solver=solver, device=device, max_scenarios=max_scenarios, assembly_only=assembly_only, options=options)


# ---- handle cache -------------------------------------------------------------------------------------------------------------------------
# The reference's sweep is repeated hpf() calls (HG:511) with other buses.P / buses.Q; its functions are stateless, ours own a device handle whose
# creation (tree planning on the host, uploads, allocation: 15 ms at 1 000 buses x 26 harmonics) costs more than the solve.  The reference call
# shapes therefore BORROW a handle from a small LRU keyed on everything hpf_create consumes -- network constants, admittance pattern and values,
# Norton arrays, harmonics, solver, and the HPF_* environment switches hpf_create reads -- and set loads / state / options themselves on every
# call, so a borrowed handle and a fresh one give bit-identical results.  Not thread-safe (neither is the reference: module globals).
_HANDLES = collections.OrderedDict()
_HANDLE_CACHE = {"size": 4, "hits": 0, "misses": 0}


def _digest(*arrays):
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str((a.dtype.str, a.shape)).encode())
        h.update(a.view(np.uint8).reshape(-1).data)
    return h.digest()


def handle_cache(size=None):
    """Size of the LRU of device handles behind hpf / pf / harmonic_mismatch / build_harmonic_jacobian (default 4; 0 switches it off: every
    call creates and destroys its own handle, as before round 5).  Returns the statistics dict (size, hits, misses)."""
    if size is not None:
        _HANDLE_CACHE["size"] = max(int(size), 0)
        while len(_HANDLES) > _HANDLE_CACHE["size"]:
            _HANDLES.popitem(last=False)[1].close()
    return dict(_HANDLE_CACHE, held=len(_HANDLES))


def close_all():
    """Destroy every cached device handle (also run at interpreter exit)."""
    while _HANDLES:
        _HANDLES.popitem(last=False)[1].close()


atexit.register(close_all)


@contextlib.contextmanager
def _borrow_model(buses, Y, NE, coupled, harmonics, solver="auto", assembly_only=False, device=0):
    """A DeviceModel of one scenario for the duration of a reference-shaped call: from the cache, or created (and then kept)."""
    if _HANDLE_CACHE["size"] <= 0:
        dm = _device_model(buses, Y, NE, coupled, harmonics, solver=solver, device=device, assembly_only=assembly_only)
        try:
            yield dm
        finally:
            dm.close()
        return
    m, n, c = ingest.network_constants(buses)
    if NE is None:
        ne_key = b"none"
    else:
        dev, Y_N, I_N, n_dev = ingest.norton_arrays(buses, NE, coupled, len(harmonics))
        ne_key = _digest(dev, Y_N, I_N)
    # (the environment is part of a handle's build only under HPF_ENV_SWITCHES=1 -- hpf.h -- but keying on it always is harmless)
    env = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("HPF_")))
    key = (n, m, c, tuple(harmonics), bool(coupled), solver, bool(assembly_only), int(device), Y.key(), ne_key, env)
    dm = _HANDLES.pop(key, None)
    if dm is None:
        _HANDLE_CACHE["misses"] += 1
        dm = _device_model(buses, Y, NE, coupled, harmonics, solver=solver, device=device, assembly_only=assembly_only)
    else:
        _HANDLE_CACHE["hits"] += 1
    try:
        yield dm
    except BaseException:
        dm.close()                                     # (a failed call may leave the handle in any state: not kept)
        raise
    _HANDLES[key] = dm
    while len(_HANDLES) > _HANDLE_CACHE["size"]:
        _HANDLES.popitem(last=False)[1].close()


def pf(Y, buses, thresh_f=1e-6, max_iter_f=30, plt_convergence=False, settings=None, verbose=True, _model=None):
    """HG:244-275 -> (V, err_t, n_iter_f): fundamental Newton-Raphson on the device from the reference's start
    (1 p.u. / 0.1 p.u., angle 0)."""
    st = settings or globals()["settings"]
    n = len(buses)
    Y = _as_admittance(Y, st.HARMONICS, n)
    with (contextlib.nullcontext(_model) if _model is not None else _borrow_model(buses, Y, None, False, Y.harmonics, solver="dense")) as dm:
        dm.set_loads(buses["P"].to_numpy(dtype=float), buses["Q"].to_numpy(dtype=float))
        dm.set_state(None, None, n_scen=1)
        n_iter, err, hist = dm.fund_pf(thresh_f, max_iter_f)
        Vm, Va = dm.get_state()
    n_iter_f = int(n_iter[0])
    err_t = {i: float(hist[0, i]) for i in range(n_iter_f)}
    V = _frame(Vm[0], Va[0], Y.harmonics, n)
    if plt_convergence:
        import matplotlib.pyplot as plt
        plt.plot(list(err_t.keys()), list(err_t.values()))
    if verbose:
        print(V.loc[Y.harmonics[0]])
        if n_iter_f < max_iter_f:
            print("Fundamental power flow converged after " + str(n_iter_f) + " iterations.")
        elif n_iter_f == max_iter_f:
            print("Warning! Maximum of " + str(n_iter_f) + " iterations reached.")
    return V, err_t, n_iter_f


def _postprocess(Vm, Va):
    """HG:545-549."""
    Vm, Va = Vm.copy(), Va.copy()
    neg = Vm < 0
    Va[neg] += np.pi
    Va = Va % (2 * np.pi)
    Vm[neg] = -Vm[neg]
    return Vm, Va


def hpf(buses, lines, coupled, thresh_h=1e-4, max_iter_h=50, plt_convergence=False, settings=None, ne_dir=None,
        solver="auto", verbose=True, return_jacobian=True, details=None, extra_iters=0):
    """HG:511-560 -> (V, err_h, n_iter_h, J).

    `J` is the Jacobian of the last iteration as scipy CSR like the reference (HG:537,560), at every size: the device writes
    the CSR arrays directly (hpf_jacobian_csr_last; 1.2 M entries at 1 000 buses x 26 harmonics).  return_jacobian=False skips it;
    J is None only when the loop took no iteration (the reference would raise NameError there, HG:560).  `details`, if a dict, receives err_hist, the pf seed, n_iter_f, solver name and device stats.
    `extra_iters` (not in the reference, default 0 = the reference's behaviour): Newton iterations taken AFTER the stop rule.
    The reference stops at err_h <= 1e-4, up to 4e-7 p.u. away from the fixed point, and where exactly below the
    threshold the last iterate lands depends on the rounding of the linear solver; one or two more iterations put the result on
    the fixed point itself, which does not (n_iter_h and err_h still report the reference's stop)."""
    st = settings or globals()["settings"]
    harmonics = st.HARMONICS
    n = len(buses)
    Y = build_admittance_matrices(buses, lines, harmonics)                       # HG:523
    NE = import_Norton_Equivalents(buses, coupled, st, ne_dir)                    # HG:528
    with _borrow_model(buses, Y, NE, coupled, harmonics, solver=solver) as dm:
        dm.set_loads(buses["P"].to_numpy(dtype=float), buses["Q"].to_numpy(dtype=float))
        dm.set_state(None, None, n_scen=1)
        nf, ef, hf = dm.fund_pf(st.thresh_f, st.max_iter_f)                       # HG:525 (pf with its defaults)
        seed = dm.get_state()
        if verbose:
            Vs = _frame(seed[0][0], seed[1][0], harmonics, n)
            print(Vs.loc[harmonics[0]])
            if int(nf[0]) < st.max_iter_f:
                print("Fundamental power flow converged after " + str(int(nf[0])) + " iterations.")
            else:
                print("Warning! Maximum of " + str(int(nf[0])) + " iterations reached.")
        # (block-tree: hpf_solve itself watches the static pivot order and repeats a flagged scenario with partial pivoting)
        want_J = bool(return_jacobian)
        dm.set_option("keep_previous_state", 1 if want_J else 0)      # (explicit both ways: the handle may be a borrowed one)
        n_iter, err, hist = dm.solve(thresh_h, max_iter_h)                        # HG:530-542
        stats = dm.stats()
        # a meshed network on the block-tree path has no pivoted repeat of its own (hpf.h): with solver="auto" a scenario the static-pivot monitor
        # flagged, or whose mismatch turned non-finite, is solved again on the dense rocSOLVER path where that fits
        retry_dense = (solver == "auto" and dm.solver == "block_tree" and dm.tree_census()["ties"] > 0 and
                       bool(stats["flags"][0] & (4 | 8)) and not (stats["flags"][0] & 16) and 8.0 * dm.N * dm.N <= 64e9)
        if details is not None:
            details["repeated_with_pivoting"] = bool(stats["flags"][0] & 16)
        if verbose and (stats["flags"][0] & 16):
            print("Warning! Static-pivot block elimination was flagged; the solve was repeated with partial pivoting.")
        if extra_iters > 0 and np.isfinite(err[0]):
            dm.mismatch(want_f=False)
            dm.iterate(int(extra_iters))
            dm.sync()
        Vm_raw, Va_raw = dm.get_state()
        n_iter_h = int(n_iter[0])
        J = None
        if want_J and n_iter_h > 0:
            # the reference returns the Jacobian built in its last iteration, i.e. at the state before the last update: the solve
            # kept that state on the device
            J = dm.jacobian_csr(0, last=True)
        if details is not None:
            details.update(err_hist=hist[0, :n_iter_h + 1].copy(), seed=(seed[0][0].copy(), seed[1][0].copy()),
                           n_iter_f=int(nf[0]), err_f=hf[0, :int(nf[0])].copy(), solver=dm.solver,
                           tree=(dm.tree_census() if dm.solver == "block_tree" else None),
                           stats=stats, Vm_raw=Vm_raw[0].copy(), Va_raw=Va_raw[0].copy(), N=dm.N)
    if retry_dense:
        if verbose:
            print("Warning! The bordered block-tree step was flagged (static pivot order / non-finite mismatch); solving again with the dense LU.")
        return hpf(buses, lines, coupled, thresh_h=thresh_h, max_iter_h=max_iter_h, plt_convergence=plt_convergence, settings=settings, ne_dir=ne_dir,
                   solver="dense", verbose=verbose, return_jacobian=return_jacobian, details=details, extra_iters=extra_iters)
    Vm, Va = _postprocess(Vm_raw[0], Va_raw[0])                                   # HG:545-549
    V = _frame(Vm, Va, harmonics, n)
    err_h = float(err[0])
    if plt_convergence:
        import matplotlib.pyplot as plt
        plt.plot(range(n_iter_h), hist[0, 1:n_iter_h + 1])
    if verbose:
        print(V)
        if n_iter_h < max_iter_h:
            print("Harmonic power flow converged after " + str(n_iter_h) + " iterations.")
        elif n_iter_h == max_iter_h:
            print("Warning! Maximum of " + str(n_iter_h) + " iterations reached.")
    return V, err_h, n_iter_h, J


def _state_arrays(V):
    return (np.ascontiguousarray(V["V_m"].to_numpy(dtype=float)), np.ascontiguousarray(V["V_a"].to_numpy(dtype=float)))


def harmonic_mismatch(V, Y, buses, NE, settings=None):
    """HG:360-390 -> (f, err_h) for the voltages in DataFrame `V`."""
    st = settings or globals()["settings"]
    n = len(buses)
    harmonics = list(dict.fromkeys(V.index.get_level_values(0)))
    Y = _as_admittance(Y, harmonics, n)
    coupled = np.asarray(next(iter(NE.values()))[1]).shape[0] > 1 if NE else False
    with _borrow_model(buses, Y, NE, coupled, harmonics, solver="dense", assembly_only=True) as dm:
        dm.set_loads(buses["P"].to_numpy(dtype=float), buses["Q"].to_numpy(dtype=float))
        dm.set_state(*_state_arrays(V))
        f, err = dm.mismatch()
    return f[0], float(err[0])


def build_harmonic_jacobian(V, Y, NE, coupled, buses=None):
    """HG:401-473 -> real scipy CSR matrix in the reference's row/column order."""
    buses = buses if buses is not None else _ctx["buses"]
    if buses is None:
        raise ValueError("pass buses= or call init_network first (the reference reads the global `buses`)")
    n = len(buses)
    harmonics = list(dict.fromkeys(V.index.get_level_values(0)))
    Y = _as_admittance(Y, harmonics, n)
    with _borrow_model(buses, Y, NE, coupled, harmonics, solver="dense", assembly_only=True) as dm:
        dm.set_loads(buses["P"].to_numpy(dtype=float), buses["Q"].to_numpy(dtype=float))
        dm.set_state(*_state_arrays(V))
        J = dm.jacobian_csr(0)
    return J


def harmonic_state_vector(V, c=None, buses=None):
    """HG:393-398."""
    if c is None:
        b = buses if buses is not None else _ctx["buses"]
        c = ingest.network_constants(b)[2]
    return np.append(V.V_a.to_numpy()[1:], V.V_m.to_numpy()[c:])


def _dims_from_context(N):
    """(n, c, Hn) of the network the last init_network call loaded, if a Jacobian of N unknowns can belong to it (N = 2 n Hn - 1 - c)."""
    b = _ctx["buses"]
    if b is None:
        return None
    n = len(b)
    c = ingest.network_constants(b)[2]
    Hn, rem = divmod(N + 1 + c, 2 * n)
    return (n, c, Hn) if rem == 0 and Hn >= 1 else None


def update_harmonic_state_vec(J, x, f, device=0, dims=None):
    """HG:476-479: x - spsolve(J, f), the linear solve on the GPU.

    J sparse (what `build_harmonic_jacobian` / `hpf` return, and what the reference passes, HG:478): `hpf_sparse_solve` -- the CSR entries are
    scattered on the device into bus-major 2Hn x 2Hn blocks and eliminated along the feeder tree; no N x N array exists on host or device, so the
    1 000-bus x 26-harmonic Jacobian (N = 51 998, 1.2 M entries) is solved like the 46-unknown one.  The row / column numbering of the reference's
    stacked matrix is a function of (n, c, Hn): taken from `dims=(n, c, Hn)`, else from the matrix itself (the Jacobians this package returns
    carry it), else from the network of the last `init_network` call.  A meshed network's Jacobian (spanning tree + loop-closing lines) takes the
    same route -- tree part factorised once, dense border system of the lines' endpoint buses.  Dense arrays, and sparse matrices the sparse route
    refuses (bus graph not connected from bus 0, block pattern not symmetric, border beyond 16 384 unknowns), go to the dense rocSOLVER LU
    (`hpf_dense_solve`), which is bounded by 8 N^2 bytes of host and device memory."""
    lib = _lib.load()
    fv = np.ascontiguousarray(f, dtype=np.float64)
    N = fv.size
    dp = C.POINTER(C.c_double)
    if sp.issparse(J):
        if J.shape != (N, N):
            raise ValueError("J must be %d x %d" % (N, N))
        d = dims or getattr(J, "_hpf_dims", None) or _dims_from_context(N)
        if d is not None and 2 * d[0] * d[2] - 1 - d[1] == N and 2 * d[2] <= 128:
            Jc = J.tocsr()
            if Jc.nnz >= 2 ** 31:
                raise ValueError("hpf_sparse_solve takes 32-bit CSR indices (nnz < 2^31)")
            indptr = np.ascontiguousarray(Jc.indptr, dtype=np.int32)
            indices = np.ascontiguousarray(Jc.indices, dtype=np.int32)
            data = np.ascontiguousarray(Jc.data, dtype=np.float64)
            dx = np.empty(N)
            ip = C.POINTER(C.c_int32)
            rc = lib.hpf_sparse_solve(int(device), int(d[0]), int(d[1]), int(d[2]), indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip),
                                      data.ctypes.data_as(dp), fv.ctypes.data_as(dp), dx.ctypes.data_as(dp))
            if rc == 0:
                return np.asarray(x, dtype=np.float64) - dx
            if rc != -3:                                   # (HPF_E_TOPOLOGY: a pattern the sparse route does not take -> the dense LU below)
                raise RuntimeError("hpf_sparse_solve failed: %s (code %d)" % (lib.hpf_strerror(rc).decode(), rc))
        if 8.0 * N * N > 16e9:
            raise ValueError("update_harmonic_state_vec: a sparse Jacobian of %d unknowns %s; the dense fallback would need %.1f GB on the host -- "
                             "solve such a network with hpf() (bordered block-tree step)"
                             % (N, "that hpf_sparse_solve refuses (HPF_E_TOPOLOGY)" if d is not None else "whose (n, c, Hn) is unknown (pass dims=)", 8e-9 * N * N))
    Jd = np.asfortranarray(J.toarray() if hasattr(J, "toarray") else np.asarray(J), dtype=np.float64)
    if Jd.shape != (N, N):
        raise ValueError("J must be %d x %d" % (N, N))
    dx = np.empty(N)
    rc = lib.hpf_dense_solve(int(device), N, Jd.ctypes.data_as(dp), fv.ctypes.data_as(dp), dx.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError("hpf_dense_solve failed: %s (code %d)" % (lib.hpf_strerror(rc).decode(), rc))
    return np.asarray(x, dtype=np.float64) - dx


def get_THD(V):
    """HG:563-572 -> DataFrame[THD_F, THD_R] per bus (harmonic labels >= 3 are the non-fundamental rows).  One pass over the [Hn][n]
    magnitudes; the per-bus sums run over the harmonics in ascending order like the reference's Python `sum` (HG:567-570), so the values
    are the reference's to the last bit."""
    harmonics = list(dict.fromkeys(V.index.get_level_values(0)))
    n = len(V) // len(harmonics)
    Vm = V["V_m"].to_numpy(dtype=float).reshape(len(harmonics), n)
    hs = np.zeros(n)
    for q in range(1, len(harmonics)):               # sequential over harmonics, vectorised over buses
        hs = hs + Vm[q] ** 2
    tot = Vm[0] ** 2
    for q in range(1, len(harmonics)):
        tot = tot + Vm[q] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        thd_f = np.sqrt(hs) / Vm[0]
        thd_r = np.sqrt(hs) / np.sqrt(tot)
    return pd.DataFrame({"THD_F": thd_f, "THD_R": thd_r})


def solve(filename_buses, filename_lines, coupled=True, settings=None, ne_dir=None, solver="auto", verbose=False, extra_iters=0):
    """Convenience wrapper (= init_network + hpf + get_THD) -> dict(V, err_h, n_iter_h, THD, details)."""
    st = settings or globals()["settings"]
    buses, lines, m, n, c = init_network(filename_buses, filename_lines, settings=st)
    details = {}
    V, err_h, n_iter_h, J = hpf(buses, lines, coupled, st.thresh_h, st.max_iter_h, settings=st, ne_dir=ne_dir,
                                solver=solver, verbose=verbose, details=details, extra_iters=extra_iters)
    # converged = the stop rule err_h <= thresh_h was met (flags bit 0) -- not "the loop ended": a NaN mismatch ends it too
    return {"V": V, "err_h": err_h, "n_iter_h": n_iter_h, "THD": get_THD(V), "details": details,
            "converged": bool(details["stats"]["flags"][0] & 1)}
