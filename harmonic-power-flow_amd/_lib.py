"""ctypes binding of libhpf.so (include/hpf.h).  There is NO CPU fallback: if the HIP library is missing or a GPU
is not present, every product entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPF_LIB_PATH") or os.path.join(_HERE, "libhpf.so")   # HPF_LIB_PATH: diagnostic builds

SOLVER_DENSE = 0
SOLVER_BLOCK_TREE = 1

c_int_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)


class hpf_desc(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("Hn", C.c_int32), ("nnz", C.c_int32),
                ("n_dev", C.c_int32), ("coupled", C.c_int32), ("solver", C.c_int32), ("device", C.c_int32),
                ("max_scenarios", C.c_int32),
                ("rowptr", c_int_p), ("col", c_int_p), ("Yval", c_dbl_p), ("dev_of_bus", c_int_p),
                ("Y_N", c_dbl_p), ("I_N", c_dbl_p)]


class hpf_stat(C.Structure):
    _fields_ = [("n_iter", C.c_int32), ("flags", C.c_int32), ("err", C.c_double), ("thd_max", C.c_double)]


# every symbol declared in include/hpf.h: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "hpf_create": (C.c_int, [C.POINTER(_H), C.POINTER(hpf_desc)]),
    "hpf_create_opts": (C.c_int, [C.POINTER(_H), C.POINTER(hpf_desc), C.c_char_p]),
    "hpf_destroy": (C.c_int, [_H]),
    "hpf_strerror": (C.c_char_p, [C.c_int]),
    "hpf_last_error_detail": (C.c_int, [_H]),
    "hpf_version": (C.c_int, []),
    "hpf_num_unknowns": (C.c_int, [_H]),
    "hpf_num_unknowns_fund": (C.c_int, [_H]),
    "hpf_num_scenarios": (C.c_int, [_H]),
    "hpf_max_scenarios": (C.c_int, [_H]),
    "hpf_tree_levels": (C.c_int, [_H]),
    "hpf_tree_depths": (C.c_int, [_H]),
    "hpf_set_loads": (C.c_int, [_H, C.c_int, c_dbl_p, c_dbl_p]),
    "hpf_set_state": (C.c_int, [_H, C.c_int, c_dbl_p, c_dbl_p]),
    "hpf_get_state": (C.c_int, [_H, c_dbl_p, c_dbl_p]),
    "hpf_mismatch": (C.c_int, [_H, c_dbl_p, c_dbl_p]),
    "hpf_jacobian": (C.c_int, [_H, C.c_int, c_dbl_p]),
    "hpf_jacobian_last": (C.c_int, [_H, C.c_int, c_dbl_p]),
    "hpf_jacobian_nnz": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "hpf_jacobian_csr": (C.c_int, [_H, C.c_int, c_int_p, c_int_p, c_dbl_p]),
    "hpf_jacobian_csr_last": (C.c_int, [_H, C.c_int, c_int_p, c_int_p, c_dbl_p]),
    "hpf_fund_mismatch": (C.c_int, [_H, c_dbl_p, c_dbl_p]),
    "hpf_fund_jacobian": (C.c_int, [_H, C.c_int, c_dbl_p]),
    "hpf_fund_pf": (C.c_int, [_H, C.c_double, C.c_int, c_int_p, c_dbl_p, c_dbl_p]),
    "hpf_solve": (C.c_int, [_H, C.c_double, C.c_int, c_int_p, c_dbl_p, c_dbl_p]),
    "hpf_solve_queue": (C.c_int, [_H, C.c_int, c_dbl_p, c_dbl_p, C.c_double, C.c_int, C.c_double, C.c_int, C.POINTER(hpf_stat), c_dbl_p, c_dbl_p]),
    "hpf_set_trace": (C.c_int, [_H, c_dbl_p, c_dbl_p, C.c_int]),
    "hpf_iterate": (C.c_int, [_H, C.c_int]),
    "hpf_get_stats": (C.c_int, [_H, C.POINTER(hpf_stat)]),
    "hpf_get_stats_dev": (C.c_int, [_H, C.c_void_p]),
    "hpf_debug_stamps": (C.c_int, [_H, C.POINTER(C.c_longlong), C.c_int]),
    "hpf_set_option": (C.c_int, [_H, C.c_char_p, C.c_int]),
    "hpf_set_stream": (C.c_int, [_H, C.c_void_p]),
    "hpf_sync": (C.c_int, [_H]),
    "hpf_timing_enable": (C.c_int, [_H, C.c_int]),
    "hpf_timing_get": (C.c_int, [_H, C.c_int, c_dbl_p, C.POINTER(C.c_int64)]),
    "hpf_timing_reset": (C.c_int, [_H]),
    "hpf_dense_solve": (C.c_int, [C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]),
    "hpf_sparse_solve": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p]),
    "hpf_solve_flops": (C.c_double, [_H]),
    "hpf_solve_bytes": (C.c_double, [_H]),
    "hpf_back_bytes": (C.c_double, [_H]),
    "hpf_kernel_model": (C.c_int, [_H, C.c_int, c_dbl_p, c_dbl_p, c_int_p]),
    "hpf_tree_census": (C.c_int, [_H, c_int_p, C.c_int]),
    "hpf_setup_times": (C.c_int, [_H, c_dbl_p, C.c_int]),
    "hpf_scenario_groups": (C.c_int, [_H, C.c_int]),
    "hpf_tree_plan": (C.c_int, [C.POINTER(hpf_desc), C.c_char_p]),
}

_lib = None


class HpfError(RuntimeError):
    def __init__(self, code, detail, where):
        self.code, self.detail = code, detail
        msg = load().hpf_strerror(code).decode()
        super().__init__("%s failed: %s (code %d, detail %d)" % (where, msg, code, detail))


def load():
    """Load libhpf.so and bind every symbol of include/hpf.h.  Raises if the library is absent — the product has
    no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libhpf.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)            # AttributeError if the ABI drifted from include/hpf.h
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, handle, where):
    if code != 0:
        detail = load().hpf_last_error_detail(handle) if handle else 0
        raise HpfError(code, detail, where)
