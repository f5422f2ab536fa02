#!/usr/bin/env python3
"""Golden fixture for RoboCup's info['Full State'] from the reference's OWN Python (build container only).

The reference puts `self.getFullState()` - the agent=None variant, RoboCupEnvironment.py:1149-1161 - into info['Full State']
(:511): `[robots f32[A, 6] = (normalize(x, standardNorm, 0), normalize(y, ...), cos a, sin a, team, fallen | penalized),
ball f32[3] = (normalize(bx, ...), normalize(by, ...), ballOwned)]`, and its trainer reads the robots' last column from it
(models/train.py:272).  Same approach and stand-ins as gen_golden_robocup.py (scripted states of the imported reference
environment); the fixture holds the states, the two arrays the reference's method returned, and what the trainer's
expression evaluates to.  Fixture: tests/golden/robocup_fullstate.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_robocup as g1  # noqa: E402


def main():
    gg.install_standins()
    rng = np.random.RandomState(31)
    rec = g1.Rec()
    robots, balls, finished = [], [], []
    for trial in range(36):
        n = [5, 5, 3, 1, 2, 4][trial % 6]
        env, rc, cut = g1.make_env(n, 2300 + trial, False)
        g1.randomize(env, rng, trial)
        before = g1.dump(env)
        state = env.getFullState()  # what step() stores as info['Full State'] (:511)
        assert len(state) == 2 and state[0].dtype == np.float32 and state[1].dtype == np.float32
        R = len(env.agents)
        rb = np.zeros((10, 6), np.float32)
        rb[:R] = state[0]
        robots.append(rb)
        balls.append(state[1].copy())
        # models/train.py:272 evaluated on the reference's own info dict
        info = [{"Full State": state}]
        fin = np.zeros(10, np.float32)
        fin[:R] = np.array([[agent[-1] for agent in s["Full State"][0]] for s in info], np.float32)[0]
        finished.append(fin)
        rec.add("fs", before, before, np.zeros(22), [n, R])
    out = rec.finish()
    out["fs_robots"] = np.array(robots)
    out["fs_ball"] = np.array(balls)
    out["fs_finished"] = np.array(finished)
    np.savez_compressed(os.path.join(HERE, "robocup_fullstate.npz"), **out)
    print("wrote", os.path.join(HERE, "robocup_fullstate.npz"))


if __name__ == "__main__":
    main()
