#!/usr/bin/env python3
"""Compiler-flag A/B of the whole library (all semantics-preserving backend / mid-end switches: every variant must produce the same
digests).   python3 tools/flag_sweep.py build   (here)      python3 tools/flag_sweep.py run [workloads...]   (GPU box)
Variants are interleaved pass by pass; whole-episode means of tools/episode_time.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHI = ["-mllvm", "-phi-node-folding-threshold=12", "-mllvm", "-two-entry-phi-node-folding-threshold=8"]
VARIANTS = {   # name: {translation unit: flags replacing that unit's own (dynenv_amd/build.py UNITS)}; round 5's sweeps are in profiles/r05_flag_sweep_*.txt
    "base": {},
    "drv_O3": {"driving_tu.hip": ["-O3"]},
    "drv_O2": {"driving_tu.hip": ["-O2"]},
    "rc_O2": {"dynenv_capi.hip": ["-O2"]},
    "rc_Os": {"dynenv_capi.hip": ["-Os"]},
}


def lib(name):
    return os.path.join(ROOT, "dynenv_amd", "libdynenv_hip_flag_%s.so" % name)


def build():
    procs = []
    for name, extra in VARIANTS.items():
        cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); from dynenv_amd import build as b; b.build(force=True, out=%r, unit_flags=%r)" % (ROOT, lib(name), extra)]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE)))
        if len(procs) % 4 == 0:
            for _, p in procs[-4:]:
                p.wait()
    for name, p in procs:
        if p.wait() != 0:
            print("variant %s does not build: %s" % (name, p.stderr.read().decode()[-300:]))
    print("built", [n for n in VARIANTS if os.path.exists(lib(n))])


def run(workloads, passes=2):
    names = [n for n in VARIANTS if os.path.exists(lib(n))]
    for w in workloads:
        times, digests = {n: [] for n in names}, {}
        for p in range(passes):
            for n in (names if p % 2 == 0 else names[::-1]):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "episode_time.py"), w, "2"], env=dict(os.environ, DYNENV_HIP_LIB=lib(n)),
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                line = [ln for ln in r.stdout.decode().splitlines() if "ms/step" in ln]
                if not line:
                    times[n].append(float("nan"))
                    continue
                times[n] += [float(x) for x in line[0].split("ms/step")[0].split(":")[-1].split()]
                digests[n] = line[0].split("digest")[-1].strip()
        base = sum(times["base"]) / len(times["base"])
        print("== %s (base %.4f ms/step)" % (w, base))
        for n in names:
            m = sum(times[n]) / len(times[n])
            print("  %-22s %.4f  (%+.2f %%)  min %.4f  %s" % (n, m, 100 * (m / base - 1), min(times[n]), "" if digests.get(n) == digests.get("base") else "DIGEST DIFFERS " + str(digests.get(n))))
        sys.stdout.flush()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] or ["driving", "robocup"])
