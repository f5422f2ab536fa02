#!/usr/bin/env python3
"""Golden fixtures for the RoboCup game logic from the reference's OWN Python (build container only).

Same approach and stand-ins as gen_golden.py (read its docstring first): the reference's RoboCupEnvironment is imported
from /root/reference with minimal pymunk/gym/pygame/cv2 placeholders and its pure-Python methods are run on scripted
states: processAction (Robot.step/turn/kick/turnHead), tick (move timer, head clamp, kick FSM incl. joint remove/add,
getup, penalty timers, illegal defender, leave-field, approach-ball reward), isBallOutOfField + ballFreeKickProcess,
penalize + getFreePenaltySpot, the `begin` callbacks ballCollision / robotPushingDet, and getFullState/get_full_obs.
`fall()` needs pymunk's spatial query: it is pinned, with a geometric `point_query` stand-in, by gen_golden_robocup_r2.py.  Fixtures: tests/golden/robocup_unit.npz.

The getup dice (`random.random()` in tick, RoboCupEnvironment.py:932) is served from the Philox block the oracle uses.
"""
import os
import random as pyrandom
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

ROBOT_F = ("lpx", "lpy", "lvx", "lvy", "la", "lw", "rpx", "rpy", "rvx", "rvy", "ra", "rw", "head_angle", "head_moving",
           "prevx", "prevy", "initx", "inity", "penal_time", "fall_time", "move_time")
ROBOT_I = ("team", "penalized", "touching", "touch_cntr", "might_push", "fallen", "fall_cntr", "kicking", "foot",
           "joint_removed")
RNG_ROBO_STEP = 8


def make_env(n_players, seed, can_fall=False):
    rc = gg.ref("RoboCupEnvironment")
    cut = gg.ref("cutils")
    pyrandom.seed(seed)
    np.random.seed(seed)
    rc.RoboCupEnvironment.canFall = can_fall
    env = rc.RoboCupEnvironment(n_players, render=False, observationType=cut.ObservationType.FULL,
                                noiseType=cut.NoiseType.REALISTIC, noiseMagnitude=0)
    return env, rc, cut


def dump(env):
    rf, ri = [], []
    for r in env.agents:
        L, R = r.leftFoot.body, r.rightFoot.body
        ip = r.initPos if r.initPos is not None else (0.0, 0.0)
        rf.append([L._p.x, L._p.y, L._v.x, L._v.y, L.angle, L.angular_velocity, R._p.x, R._p.y, R._v.x, R._v.y, R.angle,
                   R.angular_velocity, r.headAngle, r.headMoving, r.prevPos[0], r.prevPos[1], ip[0], ip[1],
                   r.penalTime, r.fallTime, r.moveTime])
        ri.append([r.team, int(r.penalized), int(r.touching), int(r.touchCntr), int(r.mightPush), int(r.fallen),
                   int(r.fallCntr), int(r.kicking), int(r.foot) if r.foot is not None else 0, int(r.jointRemoved)])
    b = env.ball.shape.body
    lk = list(env.ball.lastKicked)
    sc = [env.elapsed, len(env.agents), int(env.ballOwned), len(lk)] + (lk + [0, 0, 0, 0])[:4] + \
         [int(env.goals[0]), int(env.goals[1]), int(env.closestID[0]), int(env.closestID[1]),
          len(env.defenders[0]), len(env.defenders[1])] + (list(env.defenders[0]) + [0] * 10)[:10] + \
         (list(env.defenders[1]) + [0] * 10)[:10]
    fl = [float(env.ballFreeCntr), float(env.gracePeriod), float(env.penalTimes[0]), float(env.penalTimes[1]),
          b._p.x, b._p.y, b._v.x, b._v.y, b.angular_velocity, env.ball.prevPos[0], env.ball.prevPos[1]]
    joints = sum(1 for r in env.agents if r.joint in env.space.shapes or r.joint in env.space.bodies)
    rfa, ria = np.zeros((10, len(ROBOT_F))), np.zeros((10, len(ROBOT_I)), np.int64)
    rfa[:len(rf)] = np.array(rf, float)
    ria[:len(ri)] = np.array(ri, np.int64)
    return rfa, ria, np.array(sc, np.int64), np.array(fl, float), joints


def rewards(env):
    return np.concatenate([np.array(env.robotRewards, float), np.array(env.robotPosRewards, float),
                           np.array(env.teamRewards, float)])


def zero_rewards(env):
    n2 = len(env.agents)
    env.robotRewards = np.array([0.0] * n2)
    env.robotPosRewards = np.array([0.0] * n2)
    env.teamRewards = np.array([0.0, 0.0])
    env.obsRewards = np.array([0.0] * n2)


class Rec(object):
    def __init__(self):
        self.d = {}

    def add(self, tag, before, after, rew, extra):
        for name, arrs in (("b", before), ("a", after)):
            for k, v in zip(("rf", "ri", "sc", "fl"), arrs[:4]):
                self.d.setdefault("%s_%s_%s" % (tag, name, k), []).append(v)
        self.d.setdefault("%s_rew" % tag, []).append(rew)
        self.d.setdefault("%s_extra" % tag, []).append(np.array(extra, float))

    def finish(self):
        return {k: np.array(v) for k, v in self.d.items()}


def randomize(env, rng, trial):
    """Put the reference env into a varied but consistent state."""
    for r in env.agents:
        L, R = r.leftFoot.body, r.rightFoot.body
        mode = (trial + r.id) % 7
        if mode == 0:
            p = gg.Vec2d(rng.rand() * 1040, rng.rand() * 740)
        elif mode == 1:   # inside own penalty box
            x = 70 + rng.rand() * 60
            p = gg.Vec2d(x if r.team > 0 else 1040 - x, 370 + (rng.rand() - 0.5) * 200)
        elif mode == 2:   # outside the carpet
            p = gg.Vec2d([-5, 1045, 500, 500][trial % 4] + rng.rand(), [300, 300, -4, 745][trial % 4] + rng.rand())
        else:
            p = gg.Vec2d(100 + rng.rand() * 840, 100 + rng.rand() * 540)
        ang = (rng.rand() - 0.5) * 6
        L.position = p
        R.position = p + gg.Vec2d((rng.rand() - 0.5) * 0.6, (rng.rand() - 0.5) * 0.6)
        L.angle = ang
        R.angle = ang + (rng.rand() - 0.5) * 0.02
        L.velocity = gg.Vec2d((rng.rand() - 0.5) * 100, (rng.rand() - 0.5) * 100)
        R.velocity = gg.Vec2d((rng.rand() - 0.5) * 100, (rng.rand() - 0.5) * 100)
        L.angular_velocity = (rng.rand() - 0.5) * 2
        R.angular_velocity = (rng.rand() - 0.5) * 2
        r.prevPos = r.getPos() + (gg.Vec2d((rng.rand() - 0.5) * 2, (rng.rand() - 0.5) * 2) if trial % 3 else gg.Vec2d(0, 0))
        r.headAngle = (rng.rand() - 0.5) * 4.2
        r.headMoving = [0.0, 0.01, -0.013][(trial + r.id) % 3]
        r.moveTime = float([0, 10, 300, 310, 400, 410, 500, 510, 990][(trial + 2 * r.id) % 9])
        r.kicking = bool((trial + r.id) % 4 == 1)
        r.foot = (trial + r.id) % 2
        r.initPos = p + gg.Vec2d(1.0, -2.0)
        r.jointRemoved = bool(r.kicking and r.moveTime <= 500 and (trial % 2 == 0))
        if r.jointRemoved:
            env.space.remove(r.joint)
        r.fallen = bool((trial + r.id) % 5 == 2)
        r.fallTime = float([5.0, 10.0, 2000.0][(trial + r.id) % 3])
        r.fallCntr = (trial + r.id) % 3
        r.penalized = bool((trial + r.id) % 6 == 3)
        r.penalTime = float([5.0, 10.0, 15000.0][(trial + r.id) % 3])
        r.touching = bool(trial % 2)
        r.touchCntr = trial % 4
        r.mightPush = bool((trial // 2) % 2)
    b = env.ball.shape.body
    b.position = gg.Vec2d(520 + (rng.rand() - 0.5) * 900, 370 + (rng.rand() - 0.5) * 600)
    b.velocity = gg.Vec2d((rng.rand() - 0.5) * 200, (rng.rand() - 0.5) * 200)
    b.angular_velocity = (rng.rand() - 0.5) * 5
    env.ball.prevPos = b.position + gg.Vec2d((rng.rand() - 0.5) * 4, 0.0)
    n2 = len(env.agents)
    k = trial % 5
    env.ball.lastKicked = [int(x) for x in rng.permutation(n2)[:min(k, 4)]]
    env.closestID = [int(rng.randint(0, n2 // 2)), n2 // 2 + int(rng.randint(0, n2 // 2))]
    env.ballOwned = [1, -1, 0][trial % 3]
    env.ballFreeCntr = [9999, 0, 5, 10, 300][trial % 5]
    env.gracePeriod = [0, 5, 10, 14999][trial % 4]
    env.penalTimes = [20000 + 10000 * (trial % 3), 20000 + 10000 * ((trial // 3) % 2)]
    env.defenders = [[int(x) for x in rng.permutation(n2 // 2)[:trial % 3]],
                     [n2 // 2 + int(x) for x in rng.permutation(n2 // 2)[:(trial // 3) % 3]]]
    env.goals = np.array([trial % 2, (trial // 2) % 3])
    env.elapsed = int(rng.randint(0, 11990))


def fresh(n_players, seed, can_fall, trial, rng):
    env, rc, cut = make_env(n_players, seed, can_fall)
    randomize(env, rng, trial)
    return env, rc, cut


def main():
    gg.install_standins()
    rng = np.random.RandomState(5)
    rec = Rec()
    key = (42, 3, 1)  # seed, genv, episode used for the getup dice
    rc_mod = gg.ref("RoboCupEnvironment")

    def getup_random():
        blk = gg.env_rng(key[0], key[1], key[2], RNG_ROBO_STEP, cur["robot"] | (1 << 8), cur["elapsed"])
        return blk[0] * 2.0 ** -32
    cur = {"robot": 0, "elapsed": 0}

    # ---- processAction (canFall False: no dice) ----------------------------------------------------------
    for trial in range(75):
        env, rc, cut = fresh(5, 100 + trial, False, trial, rng)
        rid = trial % 10
        robot = env.agents[rid]
        if trial % 3 == 0:  # make sure the "can move" path is well covered
            robot.penalized = robot.kicking = robot.fallen = False
        act = np.array([trial % 5, (trial // 5) % 3, (trial // 15) % 3, trial % 7])
        if trial % 4 == 0:
            act[0] = act[1] = 0
        before = dump(env)
        zero_rewards(env)
        env.processAction(act.copy(), robot)
        rec.add("pa", before, dump(env), rewards(env), list(act) + [rid])
    # ---- tick ----------------------------------------------------------------------------------------------
    rc_mod.random.random = getup_random
    for trial in range(180):
        can_fall = False  # a re-fall (r > 0.9) would need space.point_query; covered by GPU-vs-oracle tests only
        env, rc, cut = fresh(5, 300 + trial, can_fall, trial, rng)
        rid = trial % 10
        robot = env.agents[rid]
        cur["robot"], cur["elapsed"] = rid, env.elapsed
        before = dump(env)
        zero_rewards(env)
        env.tick(robot)
        rec.add("tick", before, dump(env), rewards(env), [rid, before[4], dump(env)[4]])
    rc_mod.random.random = gg._ORIG_RANDOM
    # ---- isBallOutOfField ----------------------------------------------------------------------------------
    for trial in range(150):
        env, rc, cut = fresh(5, 700 + trial, False, trial, rng)
        b = env.ball.shape.body
        m = trial % 10
        if m == 1: b.position = gg.Vec2d(60.0 - rng.rand() * 5, 370 + (rng.rand() - 0.5) * 150)      # goal left
        elif m == 2: b.position = gg.Vec2d(975.0 + rng.rand() * 5, 370 + (rng.rand() - 0.5) * 150)   # goal right
        elif m == 3: b.position = gg.Vec2d(60.0 - rng.rand() * 5, 100 + rng.rand() * 150)            # out at left end
        elif m == 4: b.position = gg.Vec2d(975.0 + rng.rand() * 5, 480 + rng.rand() * 150)           # out at right end
        elif m == 5: b.position = gg.Vec2d(100 + rng.rand() * 800, 64.0 - rng.rand())                # touch line top
        elif m == 6: b.position = gg.Vec2d(100 + rng.rand() * 800, 676.0 + rng.rand())               # touch line bottom
        before = dump(env)
        zero_rewards(env)
        fin = env.isBallOutOfField()
        rec.add("ball", before, dump(env), rewards(env), [int(fin)])
    # ---- penalize ------------------------------------------------------------------------------------------
    for trial in range(40):
        env, rc, cut = fresh(5, 1100 + trial, False, trial, rng)
        if trial % 4 == 0:  # crowd the penalty spots so that the free-spot search has to advance
            for k, r in enumerate(env.agents):
                x = 70 + (k % 5 + 1) * 52.5
                pos = gg.Vec2d(x if r.team > 0 else 1040 - x, 70.0 if k % 2 else 670.0)
                r.leftFoot.body.position = pos
                r.rightFoot.body.position = pos
        rid = trial % 10
        before = dump(env)
        zero_rewards(env)
        env.penalize(env.agents[rid])
        rec.add("pen", before, dump(env), rewards(env), [rid, before[4], dump(env)[4]])
    # ---- begin callbacks: ballCollision (foot, ball) and robotPushingDet (foot, foot) -------------------------
    class Arb(object):
        pass
    for trial in range(100):
        can_fall = bool(trial % 2)
        env, rc, cut = fresh(5, 1300 + trial, can_fall, trial, rng)
        i = trial % 10
        foot_i = env.agents[i].rightFoot if trial % 2 else env.agents[i].leftFoot
        arb = Arb()
        before = dump(env)
        zero_rewards(env)
        if trial % 3 == 0:
            arb.shapes = [foot_i, env.ball.shape]
            ret = env.ballCollision(arb, env.space, None)
            slots = [2 * i + (trial % 2), 20]
        else:
            j = (i + 1 + trial % 8) % 10
            if trial % 11 == 0: j = i
            foot_j = env.agents[j].leftFoot if (trial // 2) % 2 else env.agents[j].rightFoot
            if j == i: foot_j = env.agents[i].leftFoot if trial % 2 else env.agents[i].rightFoot
            sa, sb = 2 * i + (trial % 2), 2 * j + (0 if (trial // 2) % 2 else 1)
            if j == i: sb = 2 * i + (0 if trial % 2 else 1)
            if sa > sb:  # Chipmunk passes the shapes in collision order (lower slot first in our canonical order)
                foot_i, foot_j, sa, sb = foot_j, foot_i, sb, sa
            arb.shapes = [foot_i, foot_j]
            ret = env.robotPushingDet(arb, env.space, None)
            slots = [sa, sb]
        rec.add("cb", before, dump(env), rewards(env), slots + [int(bool(ret)), int(can_fall)])
    # ---- Full observation formatting -----------------------------------------------------------------------
    for trial in range(24):
        n = [5, 5, 3, 1][trial % 4]
        env, rc, cut = make_env(n, 1700 + trial, False)
        randomize(env, rng, trial)
        before = dump(env)
        obs = env.get_full_obs()
        R = len(env.agents)
        dim = 4 + 8 + (R - 1) * 6
        flat = np.zeros((10, 66), np.float32)
        for a, o in enumerate(obs):
            (ball, robots), (selfr,), _ = o
            flat[a, 0:4] = ball[0]
            flat[a, 4:12] = selfr[0]
            flat[a, 12:12 + robots.size] = robots.reshape(-1)
        rec.add("obs", before, before, np.zeros(22), [n, dim])
        rec.d.setdefault("obs_flat", []).append(flat)
    # ---- kickoff spots (reset): reference formulas with scripted random.random() ------------------------------
    spots = []
    for trial in range(8):
        draws = list(rng.rand(18))
        it = iter(draws)
        rc_mod.random.random = lambda: next(it)
        env = rc_mod.RoboCupEnvironment.__new__(rc_mod.RoboCupEnvironment)
        env.W, env.H = 1040, 740
        env.sideLength, env.fieldW, env.fieldH, env.ballRadius, env.lineWidth, env.centerCircleRadius = 70, 900, 600, 5, 5, 75
        env._create_robot_spots()
        spots.append(np.concatenate([np.array(draws), np.array(env.robotSpots, float).reshape(-1)]))
    rc_mod.random.random = gg._ORIG_RANDOM
    out = rec.finish()
    out["spots"] = np.array(spots)
    out["key"] = np.array(key, np.int64)
    np.savez_compressed(os.path.join(HERE, "robocup_unit.npz"), **out)
    print("wrote", os.path.join(HERE, "robocup_unit.npz"))


if __name__ == "__main__":
    main()
