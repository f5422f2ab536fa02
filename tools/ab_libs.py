#!/usr/bin/env python3
"""A/B of library builds on one GPU box: whole-episode mean step times (tools/episode_time.py) of each library, variants interleaved
pass by pass (the first name is the base).  A variant whose digest differs from the base's computes something else (a timing-only
experiment, or a bug).  Usage (GPU box): python3 tools/ab_libs.py robocup[,driving,...] base=dynenv_amd/libdynenv_hip.so x=dynenv_amd/libdynenv_hip_x.so [passes]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workloads = sys.argv[1].split(",")
libs = [a.split("=", 1) for a in sys.argv[2:] if "=" in a]
passes = int([a for a in sys.argv[2:] if "=" not in a][0]) if [a for a in sys.argv[2:] if "=" not in a] else 2
names = [n for n, _ in libs]
path = dict(libs)
for w in workloads:
    times, digests = {n: [] for n in names}, {}
    for p in range(passes):
        for n in (names if p % 2 == 0 else names[::-1]):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "episode_time.py"), w, "2"], env=dict(os.environ, DYNENV_HIP_LIB=os.path.join(ROOT, path[n])),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            line = [ln for ln in r.stdout.decode().splitlines() if "ms/step" in ln]
            if not line:
                print("  %s failed: %s" % (n, r.stderr.decode()[-300:]))
                times[n].append(float("nan"))
                continue
            times[n] += [float(x) for x in line[0].split("ms/step")[0].split(":")[-1].split()]
            digests[n] = line[0].split("digest")[-1].strip()
    base = sum(times[names[0]]) / len(times[names[0]])
    print("== %s (base %.4f ms/step)" % (w, base))
    for n in names:
        m = sum(times[n]) / len(times[n])
        print("  %-22s %.4f  (%+.2f %%)  min %.4f  %s" % (n, m, 100 * (m / base - 1), min(times[n]), "" if digests.get(n) == digests.get(names[0]) else "DIGEST DIFFERS " + str(digests.get(n))))
    sys.stdout.flush()
