"""Build libhpf.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["hpf_lib.hip", "hpf_block.hip", "hpf_csr_solve.hip"]
HEADERS = ["hpf_assembly.hpp", "hpf_internal.hpp", "hpf_gj.hpp", "hpf_gj_mfma.hpp", "hpf_gj_dense.hpp", "hpf_quad.hpp", "hpf_lin2x2.hpp", "hpf_leafbatch.hpp", "hpf_tree_plan.hpp", os.path.join("..", "..", "include", "hpf.h")]
OUT = os.path.join(HERE, "libhpf.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, stamps=False):
    """stamps=True builds the diagnostic variant libhpf_stamps.so (in-kernel phase stamps, tools/stamps.py)."""
    out = os.path.join(HERE, "libhpf_stamps.so") if stamps else OUT
    if not stamps and not force and not needs_build():
        return OUT
    extra = os.environ.get("HPF_CFLAGS", "").split()          # experiments: e.g. HPF_CFLAGS=-DHPF_Q_OCC=5 HPF_BUILD_OUT=tools/bin/x.so
    out = os.environ.get("HPF_BUILD_OUT", out)
    cmd = [os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-Wno-unused-value", "-fPIC", "-shared"] + (["-DHPF_FACTOR_STAMPS"] if stamps else []) + extra + \
          [os.path.join(CSRC, f) for f in SOURCES] + \
          ["-o", out, "-L" + os.path.join(ROCM, "lib"), "-lrocsolver", "-lrocblas",
           "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    print(build_lib(force=True, verbose=True, stamps="--stamps" in sys.argv))
