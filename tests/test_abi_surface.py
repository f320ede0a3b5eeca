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


def test_dense_solver_size_limit_is_refused_on_the_host():
    """rocSOLVER addresses the dense Jacobian with 32-bit element offsets: N*N >= 2^31 must be refused before any HIP
    call (it would fault the GPU), by the C ABI and by the Python host."""
    import ctypes as C
    import numpy as np
    import pytest
    from harmonic_power_flow_amd import _lib
    from harmonic_power_flow_amd.device import DeviceModel
    n, Hn = 1000, 26                                            # N = 51 998
    rows = [[i - 1, i, i + 1] for i in range(n)]
    rows[0], rows[-1] = [0, 1], [n - 2, n - 1]
    rowptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
    col = np.concatenate(rows).astype(np.int32)
    Yval = np.zeros((Hn, len(col)), dtype=np.complex128)
    dev = np.full(n, -1, dtype=np.int32)
    dev[650:] = 0
    YN, IN = np.zeros((1, Hn, Hn), dtype=np.complex128), np.zeros((1, Hn), dtype=np.complex128)
    with pytest.raises(ValueError, match="block_tree"):
        DeviceModel(n, 650, 1, list(range(1, 2 * Hn, 2)), rowptr, col, Yval, dev, YN, IN, 1, True, solver="dense")
    d = _lib.hpf_desc()
    d.n, d.m, d.c, d.Hn, d.nnz, d.n_dev, d.coupled, d.solver, d.device, d.max_scenarios = n, 650, 1, Hn, len(col), 1, 1, 0, 0, 1
    dp = lambda a: a.view(np.float64).ctypes.data_as(_lib.c_dbl_p)
    d.rowptr, d.col = rowptr.ctypes.data_as(_lib.c_int_p), col.ctypes.data_as(_lib.c_int_p)
    d.Yval, d.dev_of_bus, d.Y_N, d.I_N = dp(Yval), dev.ctypes.data_as(_lib.c_int_p), dp(YN), dp(IN)
    h = C.c_void_p()
    assert _lib.load().hpf_create(C.byref(h), C.byref(d)) == -1


def test_every_environment_switch_of_the_library_is_documented_in_the_header():
    """hpf_create reads diagnostic / A-B switches from the environment (HPF_*): each of them must be described in include/hpf.h."""
    import glob
    import re
    src = "".join(open(f).read() for f in glob.glob(os.path.join(REPO, "harmonic-power-flow_amd", "csrc", "*")))
    envs = set(re.findall(r'getenv\("(HPF_[A-Z_0-9]+)"\)', src))
    hdr = open(os.path.join(REPO, "include", "hpf.h")).read()
    assert envs and not [e for e in sorted(envs) if e not in hdr]
