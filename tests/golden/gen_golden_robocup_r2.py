#!/usr/bin/env python3
"""Round-2 golden fixtures for RoboCup, from the reference's OWN Python (build container only; stand-ins and approach as
in gen_golden.py / gen_golden_robocup.py - read their docstrings first).

  * post_solve / separate callbacks: robotCollision (RoboCupEnvironment.py:1038-1088, fall dice served from the Philox
    words the oracle draws for the pair), separate (:1091-1103), goalpostCollision (:1106-1125)
  * fall (:735-791) with a geometric `space.point_query` stand-in (distance of the point to every capsule / circle of the
    space, < 40) and `Body.apply_force_at_world_point`; the forces are compared through one velocity_func call per body
  * the step() composition (:446-524) on the free-flight Space stand-in (positions integrate, velocity functions run,
    nothing collides, no joint is solved - the oracle runs the same way with `test_free_flight`): processAction + tick
    order, the ball logic and its rewards, team / robot reward folding, the snapshot every 10 substeps, episode sums, done
  * the reset composition (:239-336) for randomInit False / True and deterministicTurn, every random.random() /
    np.random.permutation served from the oracle's Philox words in call order

Writes tests/golden/robocup_callbacks.npz, robocup_step.npz, robocup_reset.npz.
"""
import math
import os
import random as pyrandom
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_robocup as gr  # noqa: E402

RNG_ROBO_RESET, RNG_ROBO_STEP = 7, 8


# ------------------------------------------------------------------ extra stand-in surface (fall)
class _Hit(object):
    def __init__(self, shape):
        self.shape = shape


def _reindex(self):
    """Chipmunk answers queries from each shape's cached world geometry, refreshed inside Space.step right after the
    position update (cpShapeCacheBB) - NOT when a body is teleported by `body.position = ...` [Chipmunk 7 semantics, from
    memory; SURVEY Appendix A].  The generators call this where Space.step would have."""
    for sh in self.shapes:
        if isinstance(sh, gg._Shape):
            sh._cache = (sh.body._p.x, sh.body._p.y, sh.body.angle)


def _shape_distance(sh, p):
    px, py, ang = sh._cache
    if isinstance(sh.args[0], (tuple, list)):          # Segment(body, a, b, radius)
        a0, b0, r = sh.args[0], sh.args[1], sh.args[2]
        c, s = math.cos(ang), math.sin(ang)
        ax, ay = a0[0] * c - a0[1] * s + px, a0[0] * s + a0[1] * c + py
        bx, by = b0[0] * c - b0[1] * s + px, b0[0] * s + b0[1] * c + py
        dx, dy = bx - ax, by - ay
        t = ((p[0] - ax) * dx + (p[1] - ay) * dy) / (dx * dx + dy * dy)
        t = max(0.0, min(1.0, t))
        cx, cy = ax + dx * t, ay + dy * t
        return math.hypot(p[0] - cx, p[1] - cy) - r
    r = sh.args[0]                                      # Circle(body, radius, offset)
    return math.hypot(p[0] - px, p[1] - py) - r


def _point_query(self, pos, max_dist, shape_filter):
    return [_Hit(sh) for sh in self.shapes if isinstance(sh, gg._Shape) and _shape_distance(sh, pos) < max_dist]


def _apply_force_at_world_point(self, force, point):   # cpBodyApplyForceAtWorldPoint (centre of gravity at the origin)
    self._f = gg.Vec2d(self._f.x + force[0], self._f.y + force[1])
    rx, ry = point[0] - self._p.x, point[1] - self._p.y
    self._t = self._t + (rx * force[1] - ry * force[0])


def install():
    gg.install_standins()
    pm = sys.modules["pymunk"]
    sf = types.ModuleType("pymunk.shape_filter")
    sf.ShapeFilter = lambda **k: None
    pm.shape_filter = sf
    sys.modules["pymunk.shape_filter"] = sf
    gg.Space.point_query = _point_query
    gg.Space.reindex = _reindex
    free_flight_step = gg.Space.step

    def step(self, dt):
        free_flight_step(self, dt)
        self.reindex()   # velocity functions do not move anything: same cache as right after the position update
    gg.Space.step = step
    gg.Body.apply_force_at_world_point = _apply_force_at_world_point


def velocity_update(env):
    """what Space.step would do next with every dynamic body: its velocity function (consumes the applied forces)"""
    for b in env.space.bodies:
        if b.body_type == gg.Body.DYNAMIC:
            if b.velocity_func is not None:
                b.velocity_func(b, (0.0, 0.0), 1.0, 0.01)
            else:
                gg.Body.update_velocity(b, (0.0, 0.0), 1.0, 0.01)


def spread(env, rng, trial):
    """gr.randomize + some robots moved close to robot (trial % 10) so that fall()'s radius query finds neighbours"""
    gr.randomize(env, rng, trial)
    me = env.agents[trial % len(env.agents)]
    p = me.getPos()
    for k, r in enumerate(env.agents):
        if r is me or (trial + k) % 3:
            continue
        off = gg.Vec2d((rng.rand() - 0.5) * 90, (rng.rand() - 0.5) * 90)
        r.leftFoot.body.position = p + off
        r.rightFoot.body.position = p + off + gg.Vec2d(0.3, -0.2)
    if trial % 4 == 0:
        env.ball.shape.body.position = p + gg.Vec2d((rng.rand() - 0.5) * 60, (rng.rand() - 0.5) * 60)
    env.space.reindex()


def gen_callbacks(out):
    rng = np.random.RandomState(23)
    rec = gr.Rec()
    key = (42, 9, 2)
    rc_mod = gg.ref("RoboCupEnvironment")
    dice = {"entity": 0, "elapsed": 0, "n": 0}

    def pair_random():
        blk = gg.env_rng(key[0], key[1], key[2], RNG_ROBO_STEP, dice["entity"], dice["elapsed"])
        v = blk[dice["n"]] * 2.0 ** -32
        dice["n"] += 1
        return v

    class Arb(object):
        pass
    posts = lambda env: [g.shape for g in env.goalposts]

    def with_dice(fn, *a):
        rc_mod.random.random = pair_random
        try:
            return fn(*a)
        finally:
            rc_mod.random.random = gg._ORIG_RANDOM
    try:
        # ---- robotCollision (post_solve, Robot-Robot) ------------------------------------------------------
        for trial in range(160):
            can_fall = trial % 8 != 7
            env, rc, cut = gr.make_env(5, 2000 + trial, can_fall)
            spread(env, rng, trial)
            i = trial % 10
            j = (i + 1 + trial % 8) % 10
            if trial % 13 == 0:
                j = i
            fi, fj = trial % 2, (trial // 2) % 2
            if j == i:
                fj = 1 - fi
            sa, sb = 2 * i + fi, 2 * j + fj
            if sa > sb:
                sa, sb = sb, sa
            feet = lambda s: env.agents[s // 2].rightFoot if s % 2 else env.agents[s // 2].leftFoot
            for r, cnt in ((env.agents[sa // 2], [0, 3, 2500, 9000, 30000][trial % 5]),
                           (env.agents[sb // 2], [30000, 1, 7000, 0, 12000][(trial // 5) % 5])):
                r.touchCntr = cnt
                r.touching = True
            arb = Arb()
            arb.shapes = [feet(sa), feet(sb)]
            dice.update(entity=(sa * 32 + sb) | (2 << 16), elapsed=env.elapsed, n=0)
            before = gr.dump(env)
            gr.zero_rewards(env)
            with_dice(env.robotCollision, arb, env.space, None)
            rew = gr.rewards(env)
            velocity_update(env)
            rec.add("rcol", before, gr.dump(env), rew, [sa, sb, int(can_fall), dice["n"]])
        # ---- separate (Robot-Robot and Robot-Goalpost handlers) ---------------------------------------------
        for trial in range(60):
            env, rc, cut = gr.make_env(5, 2300 + trial, True)
            gr.randomize(env, rng, trial)
            for r in env.agents:
                r.touchCntr = 1 + (trial + r.id) % 5
            i = trial % 10
            sa = 2 * i + trial % 2
            arb = Arb()
            if trial % 3 == 0:
                k = (trial // 3) % 4
                sb = 21 + k
                foot = env.agents[i].rightFoot if sa % 2 else env.agents[i].leftFoot
                arb.shapes = [foot, posts(env)[k]]
            else:
                j = (i + 1 + trial % 7) % 10
                sb = 2 * j + (trial // 2) % 2
                if sa > sb:
                    sa, sb = sb, sa
                feet = lambda s: env.agents[s // 2].rightFoot if s % 2 else env.agents[s // 2].leftFoot
                arb.shapes = [feet(sa), feet(sb)]
            before = gr.dump(env)
            gr.zero_rewards(env)
            env.separate(arb, env.space, None)
            rec.add("sep", before, gr.dump(env), gr.rewards(env), [sa, sb])
        # ---- goalpostCollision (post_solve, Robot-Goalpost) -------------------------------------------------
        for trial in range(100):
            can_fall = trial % 6 != 5
            env, rc, cut = gr.make_env(5, 2500 + trial, can_fall)
            spread(env, rng, trial)
            i = trial % 10
            sa, k = 2 * i + trial % 2, (trial // 2) % 4
            sb = 21 + k
            robot = env.agents[i]
            robot.touchCntr = [0, 2, 1500, 6000, 20000][trial % 5]
            robot.touching = bool((trial // 5) % 2)
            foot = robot.rightFoot if sa % 2 else robot.leftFoot
            arb = Arb()
            arb.shapes = [foot, posts(env)[k]]
            dice.update(entity=(sa * 32 + sb) | (3 << 16), elapsed=env.elapsed, n=0)
            before = gr.dump(env)
            gr.zero_rewards(env)
            with_dice(env.goalpostCollision, arb, env.space, None)
            rew = gr.rewards(env)
            velocity_update(env)
            rec.add("gpost", before, gr.dump(env), rew, [sa, sb, int(can_fall), dice["n"]])
        # ---- fall ----------------------------------------------------------------------------------------------
        for trial in range(120):
            env, rc, cut = gr.make_env(5, 2700 + trial, True)
            spread(env, rng, trial)
            i = trial % 10
            robot = env.agents[i]
            robot.fallCntr = trial % 4          # > 2 after the increment -> penalize
            punish = bool(trial % 2)
            before = gr.dump(env)
            gr.zero_rewards(env)
            n_near = len([h for h in env.space.point_query(robot.getPos(), 40, None)
                          if h.shape is not robot.leftFoot and h.shape is not robot.rightFoot])
            env.fall(robot, punish)
            rew = gr.rewards(env)
            velocity_update(env)
            rec.add("fall", before, gr.dump(env), rew, [i, int(punish), n_near])
    finally:
        rc_mod.random.random = gg._ORIG_RANDOM
    d = rec.finish()
    assert (d["fall_extra"][:, 2] > 0).sum() > 30, "fall(): too few trials with neighbours inside the radius"
    d["key"] = np.array(key, np.int64)
    out.update(d)


# ------------------------------------------------------------------ step composition on the free-flight Space
class StepDice(object):
    """random.random() inside step(): served by the source line of the draw site from the oracle's Philox words
    (processAction :557/:565/:577 -> words 0/1/2 of block (ROBO_STEP, robot, elapsed); getup :932 -> word 0 of block
    (ROBO_STEP, robot | 1 << 8, elapsed))"""

    def __init__(self, key, env):
        self.key, self.env = key, env

    def __call__(self):
        f = sys._getframe(1)
        line, robot = f.f_lineno, f.f_locals["robot"]
        word = {557: 0, 565: 1, 577: 2, 932: 0}[line]
        ent = robot.id | ((1 << 8) if line == 932 else 0)
        return gg.env_rng(self.key[0], self.key[1], self.key[2], RNG_ROBO_STEP, ent, int(self.env.elapsed))[word] * 2.0 ** -32


def flat_snapshots(obs, n2):
    out = np.zeros((5, 10, 66), np.float32)
    for t, snap in enumerate(obs):
        for a, o in enumerate(snap):
            (ball, robots), (selfr,), _ = o
            out[t, a, 0:4] = ball[0]
            out[t, a, 4:12] = selfr[0]
            out[t, a, 12:12 + robots.size] = robots.reshape(-1)
    return out


def gen_step(out):
    rc_mod = gg.ref("RoboCupEnvironment")
    # tag, players, canFall, steps, seed, deterministicTurn, allowHeadTurn (continuous head channel, Box(-3, 3))
    cases = [("a", 5, False, 14, 31, 0, 0), ("b", 5, True, 14, 32, 0, 0), ("c", 2, True, 10, 33, 0, 0), ("d", 5, True, 6, 34, 0, 0),
             ("e", 5, False, 6, 35, 1, 0), ("f", 3, False, 8, 36, 0, 1), ("g", 5, False, 4, 37, 1, 1)]
    for tag, n, can_fall, steps, seed, det_turn, allow_head in cases:
        cutm = gg.ref("cutils")
        pyrandom.seed(seed); np.random.seed(seed)
        rc_mod.RoboCupEnvironment.canFall = can_fall
        rc_mod.RoboCupEnvironment.deterministicTurn = bool(det_turn)
        try:
            env = rc_mod.RoboCupEnvironment(n, render=False, observationType=cutm.ObservationType.FULL,
                                            noiseType=cutm.NoiseType.REALISTIC, noiseMagnitude=0, allowHeadTurn=bool(allow_head))
        finally:
            rc_mod.RoboCupEnvironment.deterministicTurn = False
        env.deterministicTurn = bool(det_turn)   # instance attribute: survives the class reset above
        key = (42, seed, 1)
        rng = np.random.RandomState(seed)
        b = env.ball.shape.body
        if tag != "a":   # a rolling ball: progress rewards, out of field, free kicks, a goal (case d aims at the goal)
            b.velocity = gg.Vec2d(-260.0, 8.0) if tag == "d" else gg.Vec2d((rng.rand() - 0.5) * 300, (rng.rand() - 0.5) * 300)
            env.ball.lastKicked = [int(x) for x in rng.permutation(2 * n)[:2]]
        if tag == "c":
            for r in env.agents[:2]:
                r.fallen, r.fallTime, r.fallCntr = True, 30.0 + 10 * r.id, 1   # getup dice
        rc_mod.random.random = StepDice(key, env)
        env.space.reindex()
        try:
            before = gr.dump(env)
            acts, rews, dones, obss, eps = [], [], [], [], []
            arng = np.random.RandomState(seed + 7)
            for s in range(steps):
                a = np.stack([arng.randint(0, k, 2 * n) for k in (5, 3, 3, 7)], -1)
                if s % 3 == 2:
                    a[:, 0] = 0; a[:, 1] = 0   # kicks need move == turn == 0
                if allow_head:
                    a = a.astype(np.float64)
                    a[:, 3] = np.round((arng.rand(2 * n) - 0.5) * 6.0, 3) * (arng.rand(2 * n) > 0.2)
                obs, r, done, info = env.step(a.copy())
                acts.append(a.astype(np.float64)); rews.append(np.array(r, float)); dones.append(int(done)); obss.append(flat_snapshots(obs, 2 * n))
                eps.append(np.concatenate([np.array(env.episodeRewards, float), np.array(env.episodePosRewards, float)]))
            after = gr.dump(env)
        finally:
            rc_mod.random.random = gg._ORIG_RANDOM
        for nm, arrs in (("b", before), ("a", after)):
            for k, v in zip(("rf", "ri", "sc", "fl"), arrs[:4]):
                out["%s_%s_%s" % (tag, nm, k)] = v
        out[tag + "_actions"] = np.array(acts, np.float64)
        out[tag + "_rewards"] = np.array(rews)
        out[tag + "_dones"] = np.array(dones, np.int64)
        out[tag + "_obs"] = np.array(obss)
        out[tag + "_episode"] = np.array(eps)
        out[tag + "_meta"] = np.array([n, int(can_fall), key[0], key[1], key[2], det_turn, allow_head], np.int64)
        out[tag + "_goals"] = np.array(env.goals, np.int64)
    assert out["d_goals"].sum() > 0, "case d must score"


# ------------------------------------------------------------------ reset composition
class RcResetTape(object):
    def __init__(self, seed, genv, episode):
        self.k = (seed, genv, episode)
        self.n_random = 0
        self.n_perm5 = 0

    def word(self, entity):
        return gg.env_rng(self.k[0], self.k[1], self.k[2], RNG_ROBO_RESET, entity, 0)[0]

    def random(self):
        v = self.word(self.n_random) * 2.0 ** -32
        self.n_random += 1
        return v

    def permutation(self, n):
        if n == 8:
            base = 48
        else:
            assert n == 5
            base = 32 + 8 * self.n_perm5
            self.n_perm5 += 1
        p = list(range(n))
        for i in range(n - 1):
            j = i + gg.randint_from(self.word(base + i), 0, n - 1 - i)
            p[i], p[j] = p[j], p[i]
        return np.array(p)


def gen_reset(out):
    rc_mod, cut = gg.ref("RoboCupEnvironment"), gg.ref("cutils")
    cases = [(5, 42, 0, 0, 0), (5, 42, 11, 2, 0), (3, 7, 5, 0, 0), (5, 42, 0, 0, 1), (5, 9, 77, 1, 1), (2, 3, 4, 0, 1), (5, 42, 6, 0, 3),
             (5, 11, 8, 0, 1), (5, 12, 9, 0, 1), (5, 13, 10, 0, 1)]
    keys = []
    for ci, (n, seed, genv, episode, flags) in enumerate(cases):
        tape = RcResetTape(seed, genv, episode)
        orig = (pyrandom.random, np.random.permutation)
        pyrandom.random, np.random.permutation = tape.random, tape.permutation
        rc_mod.RoboCupEnvironment.randomInit = bool(flags & 1)
        rc_mod.RoboCupEnvironment.deterministicTurn = bool(flags & 2)
        rc_mod.RoboCupEnvironment.canFall = True
        try:
            env = rc_mod.RoboCupEnvironment(n, render=False, observationType=cut.ObservationType.FULL,
                                            noiseType=cut.NoiseType.REALISTIC, noiseMagnitude=0)
        finally:
            pyrandom.random, np.random.permutation = orig
            rc_mod.RoboCupEnvironment.randomInit = False
            rc_mod.RoboCupEnvironment.deterministicTurn = False
        rf, ri, sc, fl, _ = gr.dump(env)
        out["reset%d_rf" % ci], out["reset%d_ri" % ci], out["reset%d_sc" % ci], out["reset%d_fl" % ci] = rf, ri, sc, fl
        out["reset%d_obs" % ci] = flat_snapshots([env.get_full_obs()], 2 * n)[0]
        keys.append([n, seed, genv, episode, flags, tape.n_random])
    out["reset_keys"] = np.array(keys, np.int64)
    owned = [int(out["reset%d_sc" % ci][2]) for ci in range(len(cases)) if cases[ci][4] & 1]
    assert len(set(owned)) > 1, "randomInit cases must cover more than one ball ownership: %r" % owned


def main():
    install()
    cb, st, rs = {}, {}, {}
    gen_callbacks(cb)
    np.savez_compressed(os.path.join(HERE, "robocup_callbacks.npz"), **cb)
    gen_step(st)
    np.savez_compressed(os.path.join(HERE, "robocup_step.npz"), **st)
    gen_reset(rs)
    np.savez_compressed(os.path.join(HERE, "robocup_reset.npz"), **rs)
    print("wrote robocup_callbacks.npz, robocup_step.npz, robocup_reset.npz")


if __name__ == "__main__":
    main()
