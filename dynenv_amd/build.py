"""Build the HIP C-ABI library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libdynenv_hip.so")
SRC = os.path.join(PKG, "csrc", "dynenv_capi.hip")
DEPS = [os.path.join(PKG, "csrc", f) for f in
        ("dynenv_capi.hip", "driving_kernels.hip", "driving_partial.hip", "robocup_kernels.hip", "robocup_partial.hip", "arranger_kernels.hip", "driving_dev.h",
         "robocup_dev.h", "dev_common.h")] + \
       [os.path.join(ROOT, "include", f) for f in ("dynenv.h", "dynenv_math.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, out=None, defines=(), extra=()):
    """hipcc --offload-arch=gfx950, FMA contraction off (bit-exact parity with the oracle); -fno-optimize-sibling-calls: see
    DE_OOL in csrc/dev_common.h (out-of-line device functions without callee-saved registers).
    `out` / `defines` build an instrumented variant next to the product library (e.g. -DDRV_PROFILE)."""
    if out is None and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-optimize-sibling-calls", "-fPIC", "-shared",
           "-std=c++17", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
           "-o", out or LIB, SRC] + ["-D" + d for d in defines] + list(extra)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out or LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
