"""libhpf.so loads on a machine without a GPU and exports every symbol include/hpf.h declares (no compute)."""
import os
import re

from conftest import REPO


def test_header_symbols_exported_and_bound():
    from harmonic_power_flow_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "hpf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hpf_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert lib.hpf_version() >= 100
    assert lib.hpf_strerror(0) == b"success"
    assert b"loop-closing" in lib.hpf_strerror(-3)


def test_struct_layouts_match_header():
    import ctypes as C
    from harmonic_power_flow_amd import _lib
    assert C.sizeof(_lib.hpf_stat) == 24
    assert C.sizeof(_lib.hpf_desc) == 10 * 4 + 6 * 8


def test_create_rejects_bad_arguments_without_gpu():
    """Argument validation happens on the host before any HIP call."""
    import ctypes as C
    from harmonic_power_flow_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    d = _lib.hpf_desc()
    assert lib.hpf_create(C.byref(h), None) == -1
    assert lib.hpf_create(C.byref(h), C.byref(d)) == -1        # n = 0
    assert lib.hpf_destroy(None) == -1
    assert lib.hpf_solve(None, 1e-4, 50, None, None, None) == -1


def test_dense_solver_memory_limit_is_refused_on_the_host():
    """Dense systems beyond N*N = 2^31 run through rocSOLVER's 64-bit entry points since round 3 (GPU test
    test_dense_solver_beyond_32_bit_offsets_matches_the_block_tree_step); what the Python host still refuses before any HIP call is a
    dense model whose Jacobians cannot fit the GPU (8 N^2 bytes per scenario)."""
    import numpy as np
    import pytest
    from harmonic_power_flow_amd.device import DeviceModel
    n, Hn = 1000, 26                                            # N = 51 998: 21.6 GB per scenario
    rows = [[i - 1, i, i + 1] for i in range(n)]
    rows[0], rows[-1] = [0, 1], [n - 2, n - 1]
    rowptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
    col = np.concatenate(rows).astype(np.int32)
    Yval = np.zeros((Hn, len(col)), dtype=np.complex128)
    dev = np.full(n, -1, dtype=np.int32)
    dev[650:] = 0
    YN, IN = np.zeros((1, Hn, Hn), dtype=np.complex128), np.zeros((1, Hn), dtype=np.complex128)
    with pytest.raises(ValueError, match="block_tree"):
        DeviceModel(n, 650, 1, list(range(1, 2 * Hn, 2)), rowptr, col, Yval, dev, YN, IN, 1, True, solver="dense", max_scenarios=16)


def test_every_environment_switch_of_the_library_is_documented_in_the_header():
    """hpf_create reads diagnostic / A-B switches from the environment (HPF_*): each of them must be described in include/hpf.h."""
    import glob
    import re
    src = "".join(open(f).read() for f in glob.glob(os.path.join(REPO, "harmonic-power-flow_amd", "csrc", "*")))
    envs = set(re.findall(r'(?:getenv|sw)\("(HPF_[A-Z_0-9]+)"\)', src))
    hdr = open(os.path.join(REPO, "include", "hpf.h")).read()
    assert envs and not [e for e in sorted(envs) if e not in hdr]


def test_sparse_solve_validates_and_classifies_the_pattern_on_the_host():
    """hpf_sparse_solve (update_harmonic_state_vec for the reference's CSR Jacobian): argument checks and the bus-graph analysis of the pattern run
    on the host before any HIP call -- inconsistent CSR -> HPF_E_ARG; a bus graph that is not connected from bus 0, or a block pattern that is not
    symmetric -> HPF_E_TOPOLOGY; a ring (spanning tree + one loop-closing line) passes the analysis (round 5: bordered elimination) and ends at the first
    HIP call where there is no GPU."""
    import ctypes as C
    import numpy as np
    import scipy.sparse as sp
    from harmonic_power_flow_amd import _lib
    lib = _lib.load()
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)

    def call(n, c, Hn, J, indptr=None):
        J = J.tocsr()
        ptr = np.ascontiguousarray(J.indptr if indptr is None else indptr, dtype=np.int32)
        idx, data = np.ascontiguousarray(J.indices, dtype=np.int32), np.ascontiguousarray(J.data, dtype=float)
        f, dx = np.ones(J.shape[0]), np.empty(J.shape[0])
        return lib.hpf_sparse_solve(0, n, c, Hn, ptr.ctypes.data_as(ip), idx.ctypes.data_as(ip), data.ctypes.data_as(dp), f.ctypes.data_as(dp),
                                    dx.ctypes.data_as(dp))
    n, c, Hn = 4, 1, 3
    Nc = n * Hn - 1
    N = 2 * Nc - (c - 1)

    def bus_of(r):
        return ((r - Nc + c) if r >= Nc else (r + 1)) % n
    ring = np.zeros((N, N))                                     # 0 - 1 - 2 - 3 - 0: one loop-closing line
    for r in range(N):
        for cc in range(N):
            d = abs(bus_of(r) - bus_of(cc))
            if d in (0, 1, n - 1):
                ring[r, cc] = 1.0
    assert call(n, c, Hn, sp.csr_matrix(ring)) not in (-1, -3)
    one_way = ring.copy()                                       # the blocks (3, 0) without the blocks (0, 3): not a symmetric block pattern
    for r in range(N):
        for cc in range(N):
            if bus_of(r) == 3 and bus_of(cc) == 0:
                one_way[r, cc] = 0.0
    assert call(n, c, Hn, sp.csr_matrix(one_way)) == -3
    assert call(n, c, Hn, sp.identity(N, format="csr")) == -3   # no coupling at all: not connected from bus 0
    assert call(0, 1, 1, sp.identity(3, format="csr")) == -1
    assert call(n, n + 1, Hn, sp.identity(N, format="csr")) == -1
    assert call(3, 1, 70, sp.identity(2 * 3 * 70 - 2, format="csr")) == -1          # 2 Hn > 128
    bad = sp.identity(N, format="csr").indptr.copy()
    bad[3] = 1                                                  # decreasing row pointer
    assert call(n, c, Hn, sp.identity(N, format="csr"), indptr=bad) == -1
    assert lib.hpf_num_scenarios(None) == -1 and lib.hpf_max_scenarios(None) == -1
    one = np.ones(1)
    assert lib.hpf_dense_solve(0, 0, one.ctypes.data_as(dp), one.ctypes.data_as(dp), one.ctypes.data_as(dp)) == -1
