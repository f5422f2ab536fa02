"""The CPU-only tools of round 6 on a small population, so that they do not rot: tools/pair_order_cost.py (the oracle's test modes - pair
order reversed, velocities nudged - against the shipped order) leaves the oracle in its shipped mode, and the shipped mode repeats itself."""
import ctypes as C
import os
import sys

import numpy as np

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _driving_rewards(l, reverse, nudge, E=48, steps=120):
    l.oracle_test_modes(int(reverse), float(nudge))
    try:
        env = ol.OracleEnv(env_type=1, num_envs=E, n_players=10, seed=5, threads=4)
        env.reset()
        rng = np.random.default_rng(3)
        tot = np.zeros((E, 10))
        for _ in range(steps):
            r, _d = env.step_noobs(rng.integers(0, 3, (E, 10, 2)).astype(np.int32))
            tot += r
        return tot
    finally:
        l.oracle_test_modes(0, 0.0)


def test_oracle_test_modes_are_off_by_default_and_switch_back(oracle_built):
    l = ol.lib()
    l.oracle_test_modes.argtypes = [C.c_int, C.c_double]
    a = _driving_rewards(l, 0, 0.0)
    nudged = _driving_rewards(l, 0, 1e-12)
    b = _driving_rewards(l, 0, 0.0)
    assert np.array_equal(a, b), "the shipped mode must repeat itself after a test mode was used"
    assert not np.array_equal(a, nudged), "the nudge mode must act"
    # the reversed pair order finds the same pairs: with no coupled contacts in this short run it changes nothing beyond rounding
    rev = _driving_rewards(l, 1, 0.0)
    assert np.allclose(a, rev, rtol=0, atol=1e-6)


def test_pair_order_cost_tool_runs(oracle_built, capsys):
    import pair_order_cost as poc
    poc.main(32)
    out = capsys.readouterr().out
    assert "== driving: 32 environments" in out and "== robocup: 32 environments" in out and "pair order reversed" in out
    l = ol.lib()
    assert np.array_equal(_driving_rewards(l, 0, 0.0, 8, 20), _driving_rewards(l, 0, 0.0, 8, 20))


def test_own_side_line_excusal_accepts_the_degenerate_row_and_nothing_else():
    """tools/reference_step_fuzz.py (Partial populations) excuses exactly one thing: a zero-length line at the robot's own position on one
    side only (a penalized robot on its side line).  A real extra line, or any other difference, stays a failure."""
    import test_oracle_golden_contacts as tc
    T, L = tc.RCP_TAIL, tc.RCP_OFF_LINE
    a = np.zeros(793, np.float32)
    a[L:L + 5] = [0.3, 0.6, -0.8, 1.0, 0.0]                    # one ordinary line on both sides
    a[T + 5], a[T + 6] = 1, 1
    b = a.copy()
    b[L + 5:L + 10] = [-0.9998, 0.85, -0.52, 0.0, -1.0]        # + a zero-length line at the robot (normalized distance -1) on one side
    b[T + 5], b[T + 6] = 2, 2
    assert tc._own_line_excused(a, b, 1e-9) and tc._own_line_excused(b, a, 1e-9)
    c = b.copy()
    c[L + 5] = 0.2                                             # the extra line is a real one
    assert not tc._own_line_excused(a, c, 1e-9)
    d = b.copy()
    d[T + 1] = 1                                               # ... or something else differs as well
    assert not tc._own_line_excused(a, d, 1e-9)
    e = b.copy()
    e[3] += 0.5                                                # ... or another block's values do
    assert not tc._own_line_excused(a, e, 1e-9)
