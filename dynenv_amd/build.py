"""Build the HIP C-ABI library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libdynenv_hip.so")
SRC = os.path.join(PKG, "csrc", "dynenv_capi.hip")      # host code + RoboCup + arranger kernels: -O3
SRC_DRV = os.path.join(PKG, "csrc", "driving_tu.hip")    # the Driving kernels, a translation unit of their own: -Os (csrc/driving_host.h)
DEPS = [os.path.join(PKG, "csrc", f) for f in
        ("dynenv_capi.hip", "driving_tu.hip", "driving_kernels.hip", "driving_partial.hip", "robocup_kernels.hip", "robocup_partial.hip",
         "arranger_kernels.hip", "driving_dev.h", "driving_host.h", "robocup_dev.h", "dev_common.h")] + \
       [os.path.join(ROOT, "include", f) for f in ("dynenv.h", "dynenv_math.h")]
UNITS = ((SRC, ("-O3",)), (SRC_DRV, ("-Os",)))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, out=None, defines=(), extra=(), unit_flags=None):
    """hipcc --offload-arch=gfx950, FMA contraction off (bit-exact parity with the oracle); -fno-optimize-sibling-calls: see
    DE_OOL in csrc/dev_common.h (out-of-line device functions without callee-saved registers).  Two translation units, each compiled
    to an object with its own optimisation level, then linked.
    `out` / `defines` build an instrumented variant next to the product library (e.g. -DDRV_PROFILE); `extra` goes to both units,
    `unit_flags` = {source basename: flags} replaces a unit's own flags (tools/flag_sweep.py)."""
    if out is None and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    target = out or LIB
    common = ["--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-optimize-sibling-calls", "-fPIC", "-std=c++17",
              "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")] + ["-D" + d for d in defines]
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dynenv_build_")   # (objects of concurrent builds of one target never collide)
    try:
        objs, procs = [], []
        for src, flags in UNITS:
            if unit_flags and os.path.basename(src) in unit_flags:
                flags = tuple(unit_flags[os.path.basename(src)])
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            cmd = [hipcc, "-c"] + common + list(flags) + list(extra) + ["-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        failed = [(cmd, p.returncode) for cmd, p in procs if p.wait() != 0]   # (every compile is waited for before anything is raised)
        if failed:
            raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target + ".tmp"] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(target + ".tmp", target)   # (a reader never sees a half-written library)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if os.path.exists(target + ".tmp"):
            os.remove(target + ".tmp")
    return target


if __name__ == "__main__":
    build(force=True, verbose=True)
