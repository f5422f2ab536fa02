#!/usr/bin/env python3
"""Golden fixtures for RoboCup PARTIAL observations from the reference's own getAgentVision
(DynEnv/RoboCupEnvironment.py, `def getAgentVision` ... `return (ballDets, robDets), (...), (numLandMarks, ...)`) and
processSeens, build container only.  Same stand-ins as gen_golden.py / gen_golden_robocup.py.

Every random.random()/randint() the reference makes inside getAgentVision / cutils.addNoise / addNoiseLine is served
from the Philox word the oracle uses for that draw site (header of oracle/robocup_partial.c).  Draw sites are identified
by the caller's source line (sys._getframe) and, inside the wrapped cutils functions, by the call's arguments.
"""
import os
import random as pyrandom
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_robocup as gr  # noqa: E402

RNG_OBS_NOISE = 9
CAP = dict(ball=32, rob=20, goal=16, cross=16, fcross=28, line=12)
OFF_BALL = 0
OFF_ROB = OFF_BALL + CAP["ball"] * 5
OFF_GOAL = OFF_ROB + CAP["rob"] * 7
OFF_CROSS = OFF_GOAL + CAP["goal"] * 6
OFF_FCROSS = OFF_CROSS + CAP["cross"] * 6
OFF_LINE = OFF_FCROSS + CAP["fcross"] * 8
OFF_TAIL = OFF_LINE + CAP["line"] * 5
DIM = OFF_TAIL + 6 + 2 + 9


class Tape(object):
    def __init__(self, seed, genv, episode):
        self.key = (seed, genv, episode)
        self.agent, self.tkey, self.ctx, self.counters = 0, 0, None, {}

    def word(self, kind, index, block, w):
        entity = self.agent | (kind << 4) | (index << 8) | (block << 16)
        return gg.env_rng(self.key[0], self.key[1], self.key[2], RNG_OBS_NOISE, entity, self.tkey)[w]

    def begin_agent(self, agent, tkey):
        self.agent, self.tkey, self.counters, self.ctx = agent, tkey, {}, None

    def _nth(self, key):
        n = self.counters.get(key, 0)
        self.counters[key] = n + 1
        return n

    def _frame(self):
        f = sys._getframe(2)
        return f.f_code.co_name, f.f_lineno, f

    def random(self):
        name, line, f = self._frame()
        if self.ctx is not None:
            kind, index, it = self.ctx
            block, w = next(it)
            return self.word(kind, index, block, w) * 2.0 ** -32
        assert name == "getAgentVision", (name, line)
        loc = f.f_locals
        src = SRC[line - 1]
        if "rob[0] == SightingType.Normal and random.random()" in src or "if random.random() < self.randBase * 8" in src or \
                "offset = pymunk.Vec2d(2 * random.random()" in src or ("self.ballRadius * 2 * (1 - 0.4" in src and "rob" in loc and self.counters.get("in_fpballs")):
            robs = loc["robDets"]
            idx = [k for k, r in enumerate(robs) if r is loc["rob"]][0]
            if "and random.random() < self.randBase * 10" in src:
                self.counters["in_fpballs"] = 1
                return self.word(8, idx, 0, 0) * 2.0 ** -32
            if "randBase * 8" in src:
                return self.word(8, idx, 0, 1) * 2.0 ** -32
            if "offset" in src:
                return self.word(8, idx, 0, 2 + self._nth(("off", idx))) * 2.0 ** -32
            return self.word(8, idx, 1, 0) * 2.0 ** -32
        i = loc["i"]
        if "if random.random() < self.randBase:" in src:
            return self.word(7, i, 0, 0) * 2.0 ** -32
        if src.strip().startswith("d = random.random()"):
            return self.word(7, i, 0, 2) * 2.0 ** -32
        if src.strip().startswith("a = random.random()"):
            return self.word(7, i, 0, 3) * 2.0 ** -32
        if "(random.random() - 0.5) * 2 * math.pi" in src:
            return self.word(7, i, 1, 1) * 2.0 ** -32
        if "(-1) ** int(random.random() > 0.5)" in src:
            return self.word(7, i, 1, 2 + self._nth(("coin", i))) * 2.0 ** -32
        if "(1 - 0.4 * (random.random() - 0.5))" in src:
            n = self._nth(("size", i, line))
            return self.word(7, i, 1, 0 if n == 0 else 3) * 2.0 ** -32  # 2nd random() on the field-cross line = its angle
        raise AssertionError("unexpected random.random() site %s:%d %s" % (name, line, src))

    def randint(self, lo, hi):
        name, line, f = self._frame()
        assert name == "getAgentVision", (name, line)
        src = SRC[line - 1]
        if "crossDets.append([SightingType.Normal, ball[1]" in src:
            return gg.randint_from(self.word(6, 0, 0, self._nth("mis")), lo, hi)
        i = f.f_locals["i"]
        if "c = random.randint(0, 5)" in src:
            return gg.randint_from(self.word(7, i, 0, 1), lo, hi)
        return gg.randint_from(self.word(7, i, 1, 1 + self._nth(("ri", i))), lo, hi)


SRC = None


def pack(env, agent, vis):
    (balls, robs), (goals, crosses, fcrosses, lines), (nlm, rseen, bseen) = vis
    row = np.zeros((DIM,), np.float32)
    for arr, off, f, cap in ((balls, OFF_BALL, 5, "ball"), (robs, OFF_ROB, 7, "rob"), (goals, OFF_GOAL, 6, "goal"),
                             (crosses, OFF_CROSS, 6, "cross"), (fcrosses, OFF_FCROSS, 8, "fcross"), (lines, OFF_LINE, 5, "line")):
        a = np.asarray(arr, np.float32).reshape(-1, f) if len(arr) else np.zeros((0, f), np.float32)
        assert len(a) <= CAP[cap], (cap, len(a))
        row[off:off + a.size] = a.reshape(-1)
    lens = [len(balls), len(robs), len(goals), len(crosses), len(fcrosses), len(lines)]
    row[OFF_TAIL:OFF_TAIL + 6] = lens
    row[OFF_TAIL + 6] = nlm
    row[OFF_TAIL + 7] = float(bool(bseen))
    row[OFF_TAIL + 8:OFF_TAIL + 8 + len(rseen)] = rseen
    return row


def main():
    global SRC
    gg.install_standins()
    rc = gg.ref("RoboCupEnvironment")
    cut = gg.ref("cutils")
    SRC = open(rc.__file__).read().split("\n")
    key = (42, 5, 2)
    tape = Tape(*key)
    orig_noise, orig_line = cut.addNoise, cut.addNoiseLine

    def noise(obj, noiseType, interaction, magn, rand, maxDist, misClass=False, angleNoise=False):
        n = len(obj)
        if n == 3:
            kind = 9          # centre circle: drawn by the reference, used by nothing (not returned)
        elif n == 4:
            kind = 0          # ball
        elif angleNoise:
            kind = 4          # line cross
        elif n == 6:
            kind = 1          # robot
        elif misClass:
            kind = 3          # penalty cross
        else:
            kind = 2          # goalpost
        idx = tape._nth(("noise", kind))
        sites = [(0, 0), (0, 1), (0, 2)] + ([(0, 3)] if (misClass and noiseType == cut.NoiseType.REALISTIC) else []) + \
                [(1, 0)] + ([(1, 1)] if angleNoise else [])
        tape.ctx = (kind, idx, iter(sites))
        try:
            return orig_noise(obj, noiseType, interaction, magn, rand, maxDist, misClass, angleNoise)
        finally:
            tape.ctx = None

    def noise_line(obj, noiseType, magn, rand, maxDist):
        idx = tape._nth(("noise", 5))
        tape.ctx = (5, idx, iter([(0, 0), (0, 1), (0, 2), (0, 3), (1, 0)]))
        try:
            return orig_line(obj, noiseType, magn, rand, maxDist)
        finally:
            tape.ctx = None

    rc.addNoise, rc.addNoiseLine = noise, noise_line
    rng = np.random.RandomState(23)
    recs = {k: [] for k in ("rf", "ri", "sc", "fl", "cfg", "rows")}
    scenes = [(5, 31, cut.NoiseType.REALISTIC, 3.0, 0.03), (5, 32, cut.NoiseType.REALISTIC, 30.0, 0.30),
              (5, 33, cut.NoiseType.RANDOM, 3.0, 0.03), (5, 34, cut.NoiseType.RANDOM, 25.0, 0.25),
              (5, 35, cut.NoiseType.REALISTIC, 0.0, 0.0), (2, 36, cut.NoiseType.REALISTIC, 30.0, 0.30),
              (1, 37, cut.NoiseType.REALISTIC, 20.0, 0.20), (5, 38, cut.NoiseType.REALISTIC, 12.0, 0.12),
              (5, 39, cut.NoiseType.REALISTIC, 40.0, 0.40), (5, 40, cut.NoiseType.RANDOM, 40.0, 0.40)]
    for trial, (n_players, seed, ntype, magn, rbase) in enumerate(scenes):
        rc.random.random, rc.random.randint = gg._ORIG_RANDOM, gg._ORIG_RANDINT
        pyrandom.seed(seed)
        np.random.seed(seed)
        rc.RoboCupEnvironment.canFall = False
        env = rc.RoboCupEnvironment(n_players, render=False, observationType=cut.ObservationType.PARTIAL, noiseType=ntype,
                                    noiseMagnitude=min(magn, 5))
        env.noiseMagnitude, env.randBase = magn, rbase  # beyond the constructor's 0..5 range on purpose: rare branches
        gr.randomize(env, rng, trial)
        # a scrum in front of some robots so that occlusion / nearby interactions and FP balls happen
        for r in env.agents:
            if (r.id + trial) % 3 == 0:
                c = gg.Vec2d(300 + rng.rand() * 440, 200 + rng.rand() * 340)
                for q in env.agents:
                    if q is not r and rng.rand() < 0.6:
                        p = c + gg.Vec2d((rng.rand() - 0.5) * 160, (rng.rand() - 0.5) * 160)
                        q.leftFoot.body.position = p
                        q.rightFoot.body.position = p + gg.Vec2d(0.3, -0.2)
                env.ball.shape.body.position = c + gg.Vec2d((rng.rand() - 0.5) * 120, (rng.rand() - 0.5) * 120)
                break
        rf, ri, sc, fl, _ = gr.dump(env)
        rc.random.random, rc.random.randint = tape.random, tape.randint
        rows = np.zeros((10, DIM), np.float32)
        tkey = int(env.elapsed)
        for a, agent in enumerate(env.agents):
            tape.begin_agent(a, tkey)
            rows[a] = pack(env, agent, env.getAgentVision(agent))
        recs["rf"].append(rf); recs["ri"].append(ri); recs["sc"].append(sc); recs["fl"].append(fl)
        recs["cfg"].append(np.array([n_players, int(ntype), magn, key[0], key[1], key[2], tkey], float))
        recs["rows"].append(rows)
    rc.random.random, rc.random.randint = gg._ORIG_RANDOM, gg._ORIG_RANDINT
    # processSeens: synthetic (numLandMarks, robotsSeen, ballsSeen) tuples for 5 snapshots -> the reference's obsRewards
    ps_in, ps_out = [], []
    for trial in range(12):
        n_players = [5, 2, 1][trial % 3]
        env = rc.RoboCupEnvironment(n_players, render=False, observationType=cut.ObservationType.PARTIAL,
                                    noiseType=cut.NoiseType.REALISTIC, noiseMagnitude=1)
        R = 2 * n_players
        nlm = rng.randint(0, 9, (5, R))
        rs = (rng.rand(5, R, R - 1) < [0.2, 0.6, 0.9][trial % 3]).astype("uint8")
        bs = rng.rand(5, R) < 0.5
        observations = [[(None, None, (int(nlm[t, a]), rs[t, a], bool(bs[t, a]))) for a in range(R)] for t in range(5)]
        env.obsRewards = np.array([0.0] * R)
        env.processSeens(observations)
        pad = np.zeros((10, 1 + 9 + 1))
        pad[:R, 0] = nlm.sum(0); pad[:R, 1:R] = rs.sum(0); pad[:R, 10] = bs.sum(0)
        ps_in.append(pad)
        o = np.zeros(10); o[:R] = env.obsRewards
        ps_out.append(np.concatenate([[R], o]))
    recs["seens_in"], recs["seens_out"] = ps_in, ps_out
    np.savez_compressed(os.path.join(HERE, "robocup_partial.npz"), **{k: np.array(v) for k, v in recs.items()})
    rows = np.array(recs["rows"])
    print("wrote robocup_partial.npz; mean list lengths [ball rob goal cross fcross line]:",
          rows[:, :, OFF_TAIL:OFF_TAIL + 6].reshape(-1, 6).mean(0).round(2))


if __name__ == "__main__":
    main()
