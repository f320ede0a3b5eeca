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
    envs = set(re.findall(r'getenv\("(HPF_[A-Z_0-9]+)"\)', src))
    hdr = open(os.path.join(REPO, "include", "hpf.h")).read()
    assert envs and not [e for e in sorted(envs) if e not in hdr]
