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
    assert b"radial" in lib.hpf_strerror(-3)


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
