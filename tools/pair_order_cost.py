#!/usr/bin/env python3
"""What the unknowable pair order costs a CONSUMER.  Chipmunk hands colliding pairs to its arbiter list in BB-tree
order; the oracle and the kernels in ascending shape ids.  Trajectories with coupled contacts diverge between any two
orders; what a trainer sees are episode statistics.  So: the same 4096 environments, the same actions, one whole episode, three runs
of the CPU oracle - as shipped; with the pair order REVERSED (oracle test mode: the other extreme); with every velocity nudged by a
relative 1e-15 before every substep (what another rounding of the same arithmetic does) - and the per-environment differences in
episode reward, crashes and goals against the shipped run, beside the population means.  CPU only (the oracle); ~2 minutes.

   python3 tools/pair_order_cost.py [envs] > profiles/r06_pair_order_cost.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

E = 4096
THREADS = os.cpu_count() or 8
L = None


def _lib():
    global L
    if L is None:
        ol.build()
        L = ol.lib()
        L.oracle_test_modes.argtypes = [C.c_int, C.c_double]
    return L


def episode(kind, reverse, nudge):
    L = _lib()
    L.oracle_test_modes(int(reverse), float(nudge))
    try:
        if kind == "driving":
            env = ol.OracleEnv(env_type=1, num_envs=E, n_players=10, seed=42, threads=THREADS)
            hi, steps = (3, 3), 600
        else:
            env = ol.OracleEnv(env_type=0, num_envs=E, n_players=5, seed=42, flags=ol.ROBOCUP_DEFAULT_FLAGS, threads=THREADS)
            hi, steps = (5, 3, 3, 7), 240
        env.reset()
        rng = np.random.default_rng(7)
        contact_steps = 0
        for s in range(steps):
            a = np.stack([rng.integers(0, k, (E, 10)) for k in hi], -1).astype(np.int32)
            env.step_noobs(a)
            if s % 20 == 19:
                contact_steps += sum(1 for e in range(0, E, 64) if env.active_contacts(e) > 0)
        r, p, o, g = env.episode_stats()
        return r.sum(1), g.copy(), contact_steps
    finally:
        L.oracle_test_modes(0, 0.0)


def report(kind):
    base_r, base_g, cs = episode(kind, 0, 0.0)
    again_r, again_g, _ = episode(kind, 0, 0.0)
    assert np.array_equal(base_r, again_r) and np.array_equal(base_g, again_g), "the shipped oracle must repeat itself bit for bit"
    what = "cars that reached their goal / crashed" if kind == "driving" else "goals of the two teams"
    print("== %s: %d environments x one episode, random actions; episode reward = sum over the 10 agents (mean %.3f, std over environments %.3f); "
          "%s: %s" % (kind, E, base_r.mean(), base_r.std(), what, base_g.sum(0).tolist()))
    print("   %-34s %8s %8s %8s %8s %8s | %10s %10s %10s %10s | %12s %14s" %
          ("run against the shipped order", ">1e-9", ">1e-4", ">1e-2", ">1", ">10", "median", "p90", "p99", "max", "counts differ", "mean reward"))
    for name, rev, nudge in (("pair order reversed", 1, 0.0), ("velocities nudged by 1e-15", 0, 1e-15)):
        r, g, _ = episode(kind, rev, nudge)
        d = np.abs(r - base_r)
        fr = [float((d > t).mean()) for t in (1e-9, 1e-4, 1e-2, 1.0, 10.0)]
        gd = float((g != base_g).any(1).mean())
        se = float(np.sqrt((r.var() + base_r.var()) / E))
        print("   %-34s %7.2f%% %7.2f%% %7.2f%% %7.2f%% %7.2f%% | %10.3g %10.3g %10.3g %10.3g | %11.2f%% %8.3f (%+.3f, s.e. of the difference of two independent populations %.3f); totals %s" %
              ((name,) + tuple(100 * x for x in fr) + tuple(float(np.percentile(d, q)) for q in (50, 90, 99)) + (float(d.max()), 100 * gd, r.mean(), r.mean() - base_r.mean(), se, g.sum(0).tolist())))
    sys.stdout.flush()


def main(envs=4096):
    global E
    E = int(envs)
    print("tools/pair_order_cost.py: per-environment |difference of the episode reward| against the shipped (ascending shape id) pair order - share of the "
          "environments above a threshold, quantiles - the share whose crash / goal counts differ, and the population mean")
    for kind in ("driving", "robocup"):
        report(kind)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
