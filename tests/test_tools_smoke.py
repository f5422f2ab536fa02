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
