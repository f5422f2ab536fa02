#!/usr/bin/env python3
"""The differential fuzz of tests/test_kat_general.py at full size, as a report (CPU only; ~3 min):
   python3 tools/kat_general_fuzz.py > profiles/r05_kat_general_fuzz.txt
narrowphase: 10^5 configurations per shape pair, cold and with Chipmunk's warm-started GJK; scenes: Driving and RoboCup populations
of two seeds each against the oracle; how much the outcome depends on the ORDER of the arbiters (unknowable here: Chipmunk's BB-tree
traversal) and on rounding-level noise in the inputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_fuzz as kf  # noqa: E402
import kat_worlds as kw  # noqa: E402
import oracle_lib as ol  # noqa: E402
import test_kat_general as tk  # noqa: E402


def pct(x):
    return "50 %% %.1e  90 %% %.1e  99 %% %.1e  max %.1e" % tuple(np.percentile(x, [50, 90, 99, 100]))


def main():
    ol.build()
    print("== narrowphase: oracle/cp_lite.c (SAT, closest-point formulas) vs tests/kat_general.py (GJK / EPA -> ContactPoints), tolerance 1e-9")
    for warm in (False, True):
        for pair in kf.PAIRS:
            A, B = kf.generate(pair, 100000, 20261004)
            r = kf.compare(pair, A, B, kf.oracle_collide(A, B), warm=warm)
            print("%-16s %s  n %d  colliding %d  agree %d  largest deviation among them %.1e  GJK / EPA iterations <= %d / %d" %
                  (pair, "warm" if warm else "cold", r["n"], r["colliding"], r["agree"], r["worst"], r["gjk_iters"], r["epa_iters"]))
            for c, n in sorted(r["classes"].items()):
                print("      %6d  %s   e.g. samples %s" % (n, c, r["examples"][c]))
    print()
    print("== scenes: full kinematic state after 30 (Driving) / 50 (RoboCup) substeps, oracle vs kat_general; deviation scaled by max(1, |value|)")
    for seed in (7, 8):
        n = 2000
        scenes = tk.driving_scenes(n, seed)
        ora = ol.OracleEnv(env_type=1, num_envs=n, n_players=10, seed=5, threads=8)
        ora.reset()
        acts = np.ones((n, 10, 2), np.int32)
        dev = tk.run_driving_on(ora.set_state, lambda: ora.step(acts), ora.get_state, scenes, ora.get_state(0))
        cl = tk.tally(scenes, dev, kw.driving_expected, 2)
        print("Driving seed %d: %d scenes  %s   first touches %d  separations %d  re-touches inside collision_persistence %d  EPA calls %d" %
              (seed, n, pct(dev), tk.events(scenes, "begin"), tk.events(scenes, "separate"), tk.events(scenes, "retouch"),
               sum(w.stats.get("epa_calls", 0) for _, _, w in scenes)))
        print("      " + "  ".join("%s: %d" % (k, len(v)) for k, v in sorted(cl.items())))
    for seed in (11, 12):
        n = 500
        scenes = tk.robocup_scenes(n, seed)
        ora = ol.OracleEnv(env_type=0, num_envs=n, n_players=5, seed=3, flags=ol.FLAG_USE_OBS_REWARDS, threads=8)
        ora.reset()
        a = np.zeros((n, 10, 4), np.int32)
        a[..., 3] = 3
        dev = tk.run_robocup_on(ora.set_state, lambda: ora.step(a), ora.get_state, scenes, ora.get_state(0))
        cl = tk.tally(scenes, dev, lambda sc: kw.robocup_expected(sc), 0)
        print("RoboCup seed %d: %d scenes  %s   first touches %d  separations %d  re-touches %d" %
              (seed, n, pct(dev), tk.events(scenes, "begin"), tk.events(scenes, "separate"), tk.events(scenes, "retouch")))
        print("      " + "  ".join("%s: %d (deviations %s)" % (k, len(v), ", ".join("%.1e" % d for d in sorted(v)[-8:]) if k != "agree" else "<= 1e-9")
                                   for k, v in sorted(cl.items())))
    print()
    print("== what nobody here can know: Chipmunk finds colliding pairs in BB-tree order; both restatements use ascending shape ids.")
    print("   kat_general with the pair order REVERSED against itself (same scenes):")
    for name, gen, exp, last, cnt in (("Driving", lambda r: kw.driving_scene(r), kw.driving_expected, 2, 400),
                                      ("RoboCup", lambda r: kw.robocup_scene(r), lambda sc, **k: kw.robocup_expected(sc, **k), 0, 150)):
        rng = np.random.default_rng(21)
        d, nz = [], []
        for _ in range(cnt):
            sc = gen(rng)
            if name == "RoboCup" and not kw.valid_robocup_start(sc):
                continue
            e, w, ok = exp(sc)
            if not ok:
                continue
            e2 = exp(sc, order="reversed")[0]
            d.append(kw.deviation(list(e[last]), list(e2[last])))
            e3 = exp(tk._nudged(sc, np.random.default_rng(1)))[0]
            nz.append(kw.deviation(list(e[last]), list(e3[last])))
        d, nz = np.array(d), np.array(nz)
        print("   %s: %d scenes  order reversed: %s  share > 1e-9: %.0f %%  share > 1e-4: %.0f %%" % (name, len(d), pct(d), 100 * (d > 1e-9).mean(), 100 * (d > 1e-4).mean()))
        print("   %s              velocities nudged by 1e-15 relative: %s  share > 1e-9: %.1f %%  share > 1e-4: %.1f %%" %
              (" " * len(name), pct(nz), 100 * (nz > 1e-9).mean(), 100 * (nz > 1e-4).mean()))


if __name__ == "__main__":
    main()
