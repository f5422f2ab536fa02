#!/usr/bin/env python3
"""The reference's OWN `step()` against the oracle on MANY random trajectories WITH collisions - the population behind the committed
fixtures of tests/golden/gen_golden_contacts.py (same generator functions, same checks as tests/test_oracle_golden_contacts.py, nothing
written to disk).  Build container only: it imports /root/reference (with the functional pymunk facade over tests/kat_general.py).

   python3 tools/reference_step_fuzz.py [n_driving] [n_robocup] [n_driving_partial] [n_robocup_partial] [n_goalposts] [n_ball_out] [n_kicks] [n_penalties] [n_driving_finish] [n_falls] [n_duels] [n_driving_aimed] > profiles/r05_reference_step_fuzz.txt      (1000 300: ~10 min)
"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_golden_contacts as gc  # noqa: E402
import oracle_lib as ol  # noqa: E402
import test_oracle_golden_contacts as tc  # noqa: E402


def driving_env(n_players, seed, offset):
    env = ol.OracleEnv(num_envs=1, n_players=n_players, seed=seed, env_id_offset=offset)
    env.reset()

    def step(a):
        o, r, d = env.step(a[None])
        return o[0, 0], r[0], d[0]
    return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))


def robocup_env(n, seed, offset, flags):
    env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, seed=seed, env_id_offset=offset, flags=flags)
    env.reset()

    def step(a):
        o, r, d = env.step(a[None])
        return o[0], r[0], d[0]
    return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))


def driving_partial_env(n_players, seed, offset, magn):
    env = ol.OracleEnv(env_type=1, num_envs=1, n_players=n_players, obs_type=1, noise_type=1, noise_magnitude=magn, seed=seed, env_id_offset=offset)
    env.reset()

    def step(a):
        o, r, d = env.step(a[None])
        return o[0, 0], r[0], d[0]
    return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))


def robocup_partial_env(n, seed, offset, flags, magn):
    env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, obs_type=1, noise_type=1, noise_magnitude=magn, seed=seed, env_id_offset=offset, flags=flags)
    env.reset()

    def step(a):
        o, r, d = env.step(a[None])
        return o[0], r[0], d[0]
    return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))


def main():
    n_drv = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_rc = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    ol.build()
    gc.install()
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2026")))
    SB = int(os.environ.get("FUZZ_SEED_BASE", "0"))  # added to every trajectory seed: a population of its own, not the default one with other lengths
    devnull = open(os.devnull, "w")
    dump_dir = os.environ.get("FUZZ_DUMP")    # keep every trajectory as <dir>/<kind>_<k>.npz: tools/hip_reference_step_fuzz.py runs the HIP path on them (GPU box)
    if dump_dir:
        os.makedirs(dump_dir, exist_ok=True)

    # FUZZ_PARTIAL_MAGN=<noise magnitude>: the SCENARIO populations (goalposts ... aimed collisions) run with Partial observations + Realistic noise
    # (getAgentVision inside the step; RoboCup: processSeens rewards), checked like the two Partial populations
    PM = float(os.environ["FUZZ_PARTIAL_MAGN"]) if "FUZZ_PARTIAL_MAGN" in os.environ else None
    FORCE_N = int(os.environ.get("FUZZ_PLAYERS", "0"))     # every population with this many players (a side), e.g. 1: the single-robot special cases

    def pick_n(opts):
        v = int(rng.choice(opts))
        return FORCE_N if FORCE_N else v
    excused = []     # Partial: rows a penalized robot's own side line decided by the last bit of libm's sin / cos (tests/test_oracle_golden_contacts.py _own_line_excused)
    rc_env, rc_kw = (robocup_partial_env, dict(partial=True, own_line_slack=excused)) if PM is not None else (robocup_env, {})

    def dump(kind, k, out):
        if PM is not None:
            kind = kind.replace("robocup_", "robocup_partial_", 1).replace("driving_", "driving_partial_", 1)
        if dump_dir:
            np.savez_compressed(os.path.join(dump_dir, "%s_%05d.npz" % (kind, k)), **out)
    t0 = time.time()
    steps = touches = crashed = dead = 0
    failures = []
    for k in range(n_drv):
        n, seed, length, bias = pick_n([2, 4, 6, 8, 10, 10, 10]), 1000 + SB + k, int(rng.integers(40, 90)), float(rng.uniform(0.3, 0.9))
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_driving(out, n, seed, length, "t", bias)
        finally:
            sys.stdout = stdout
        dump("driving", k, out)
        try:
            tc.check_trajectory(out, "t", driving_env)
        except AssertionError as e:
            failures.append(("driving", n, seed, length, str(e)[:200]))
        steps += length
        touches += int(out["t_begins_per_step"].sum())
        crashed += int(out["t_states_cars_i"][-1][:, 3].sum())
        dead += int(out["t_states_peds_i"][-1][:, 2].sum())
    print("Driving: %d trajectories of the reference's DrivingEnvironment.step() (2-10 players, 40-90 steps each, %d steps; %d first touches, %d cars crashed, "
          "%d pedestrians killed) against the oracle - rewards / states 1e-9, observations 2e-6, flags exact: %d failures  (%.0f s)"
          % (n_drv, steps, touches, crashed, dead, len([f for f in failures if f[0] == "driving"]), time.time() - t0))
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    for k in range(n_rc):
        n, can_fall, length, fw = pick_n([2, 3, 4, 5, 5]), bool(rng.random() < 0.6), int(rng.integers(12, 30)), float(rng.uniform(0.4, 0.9))
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, can_fall, length, 2000 + SB + k, fw)
        finally:
            sys.stdout = stdout
        dump("robocup", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", robocup_env)
        except AssertionError as e:
            failures.append(("robocup", n, 2000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
    print("RoboCup: %d trajectories of the reference's RoboCupEnvironment.step() (2-5 a side, canFall on in ~60 %%, 12-30 steps each, %d steps; first touches "
          "robot-robot %d, robot-ball %d, robot-post %d, ball-post %d, own feet %d) against the oracle - tolerance 1e-9 or 1000 x the fixture's own conditioning, "
          "flags exact: %d failures; %d of the %d steps were well-conditioned (twin drift <= 1e-6) and checked  (%.0f s)"
          % ((n_rc, steps) + tuple(begins) + (len([f for f in failures if f[0] == "robocup"]), checked, steps, time.time() - t0)))
    # Goalposts (both populations above report "robot-post 0"): scenes that start a robot within reach of a post, walking
    # into it, and the ball rolling into another one - goalpostCollision (RoboCupEnvironment.py:1106-1125: touch counter, the fall die
    # 0.9998 ** touchCntr, fall(punish)), its separate handler, and the default-handled ball-post bounce (Goalpost.py:4-14: e = 0.95)
    n_gp = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    for k in range(n_gp):
        n, can_fall, length = pick_n([3, 4, 5, 5]), bool(rng.random() < 0.7), int(rng.integers(8, 16))
        posts = [(70.0, 450.0), (70.0, 290.0), (970.0, 450.0), (970.0, 290.0)]
        pr, pb = [int(x) for x in rng.choice(4, 2, replace=False)]
        ru, bu = float(rng.uniform(-1.0, 1.0)), float(rng.uniform(-1.2, 1.2))          # bearing from the post, field side
        rd, bd, bv, jit = float(rng.uniform(42.0, 58.0)), float(rng.uniform(60.0, 130.0)), float(rng.uniform(200.0, 380.0)), float(rng.uniform(-0.06, 0.06))
        rsel = int(rng.integers(0, n))

        def setup(env, pr=pr, pb=pb, ru=ru, bu=bu, rd=rd, bd=bd, bv=bv, jit=jit, rsel=rsel, n=n, posts=posts):
            import math
            Vec2d = gc.Vec2d
            px, py = posts[pb]
            side = 1.0 if px < 520.0 else -1.0
            ux, uy = side * math.cos(bu), math.sin(bu)
            b = env.ball.shape.body
            b.position = Vec2d(px + bd * ux, py + bd * uy)
            b.velocity = Vec2d(-bv * math.cos(math.atan2(uy, ux) + jit), -bv * math.sin(math.atan2(uy, ux) + jit))
            env.ball.prevPos = Vec2d(px + bd * ux, py + bd * uy)
            qx, qy = posts[pr]
            side = 1.0 if qx < 520.0 else -1.0
            vx, vy = side * math.cos(ru), math.sin(ru)
            rid = (n + rsel) if qx < 520.0 else rsel      # the team whose own penalty box is on the OTHER side (no illegal-defender teleport)
            r = env.agents[rid]
            for foot in (r.leftFoot, r.rightFoot):
                foot.body.position = Vec2d(qx + rd * vx, qy + rd * vy)
                foot.body.angle = math.atan2(-vy, -vx)    # facing the post: walking forward (action 3) runs into it
            r.prevPos = r.getPos()
            # bodies moved by hand: what a pymunk user owes the space (its queries - fall()'s point_query, RoboCupEnvironment.py:742 - read the
            # shapes' CACHED world geometry, refreshed only by space.step; a state blob has no stale cache to carry to the oracle)
            for body in (b, r.leftFoot.body, r.rightFoot.body):
                env.space.reindex_shapes_for_body(body)
            return {s_: {rid: [3, 0, 0, 3]} for s_ in range(0, 10)}
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, can_fall, length, 6000 + SB + k, 0.3, setup, partial_magn=PM)
        finally:
            sys.stdout = stdout
        dump("robocup_posts", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", rc_env, **rc_kw)
        except AssertionError as e:
            failures.append(("robocup_posts", n, 6000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
    if n_gp:
        print("RoboCup at the goalposts: %d trajectories (3-5 a side, a robot walking into one post, the ball rolling into another; 8-16 steps each, %d steps; first "
              "touches robot-robot %d, robot-ball %d, ROBOT-POST %d, BALL-POST %d, own feet %d) against the oracle - same tolerances: %d failures; %d of the %d steps "
              "well-conditioned and checked  (%.0f s)" % ((n_gp, steps) + tuple(begins) + (len([f for f in failures if f[0] == "robocup_posts"]), checked, steps, time.time() - t0)))
    # Partial observations (BASELINE configs[3]): getAgentVision inside the step, noise draws served by source line
    n_dp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n_rp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    t0 = time.time()
    steps, rows = 0, np.zeros(4)
    for k in range(n_dp):
        n, seed, length, bias = pick_n([2, 4, 6, 10]), 3000 + SB + k, int(rng.integers(30, 60)), float(rng.uniform(0.3, 0.9))
        magn = float(rng.choice([0.5, 1.0, 3.0, 5.0]))
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_driving_partial(out, n, seed, length, "t", bias, magn)
        finally:
            sys.stdout = stdout
        dump("driving_partial", k, out)
        try:
            rows += tc.check_partial_trajectory(out, "t", driving_partial_env)
        except AssertionError as e:
            failures.append(("driving_partial", n, seed, length, str(e)[:200]))
        steps += length
    if n_dp:
        print("Driving, Partial observations + Realistic noise (magnitude 0.5 - 5): %d trajectories, %d steps, rows seen (cars, obstacles, pedestrians, lanes) %s - "
              "rewards / final states 1e-9, observations 3e-5, row counts exact: %d failures  (%.0f s)"
              % (n_dp, steps, [int(x) for x in rows], len([f for f in failures if f[0] == "driving_partial"]), time.time() - t0))
    t0 = time.time()
    steps = checked = 0
    for k in range(n_rp):
        n, can_fall, length, fw = pick_n([2, 3, 5]), bool(rng.random() < 0.6), int(rng.integers(10, 25)), float(rng.uniform(0.4, 0.9))
        magn = float(rng.choice([0.5, 1.0, 3.0, 5.0]))
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, can_fall, length, 4000 + SB + k, fw, partial_magn=magn)
        finally:
            sys.stdout = stdout
        dump("robocup_partial", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", robocup_partial_env, partial=True)
        except AssertionError as e:
            failures.append(("robocup_partial", n, 4000 + SB + k, length, str(e)[:200]))
        steps += length
    if n_rp:
        print("RoboCup, Partial observations + Realistic noise: %d trajectories, %d steps, %d of them well-conditioned and checked (rewards incl. processSeens, "
              "five snapshots per step: list lengths / seen tuple exact, rows 2e-6): %d failures  (%.0f s)"
              % (n_rp, steps, checked, len([f for f in failures if f[0] == "robocup_partial"]), time.time() - t0))
    # The ball leaving the field (isBallOutOfField, RoboCupEnvironment.py:622-732): over a side line, over an end line beside the goal (goal kick
    # or corner, by who kicked last), into the goal; the ball put back, the free kick's owner, grace period and counter (ballFreeKickProcess
    # :600-619), the personal rewards of lastKicked, and - in the steps after - robots of the wrong team touching an owned ball (ballCollision)
    n_out = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    kinds, goals = np.zeros(4, np.int64), np.zeros(2, np.int64)
    for k in range(n_out):
        n, can_fall, length, fw = pick_n([2, 3, 5, 5]), bool(rng.random() < 0.5), int(rng.integers(6, 14)), float(rng.uniform(0.3, 0.9))
        kind = int(rng.choice(4, p=[0.3, 0.3, 0.2, 0.2]))                    # side line | end line beside the goal | goal | a slow ball that stays in
        far, u, inset, speed = bool(rng.random() < 0.5), float(rng.random()), float(rng.uniform(4.0, 25.0)), float(rng.uniform(120.0, 380.0))
        tang = float(rng.uniform(-0.5, 0.5))
        kickers = [int(x) for x in rng.choice(2 * n, min(2 * n, int(rng.integers(0, 4))), replace=False)]
        kinds[kind] += 1

        def setup(env, kind=kind, far=far, u=u, inset=inset, speed=speed, tang=tang, kickers=kickers):
            Vec2d = gc.Vec2d
            lo, hx, hy = env.sideLength, env.W - env.sideLength, env.H - env.sideLength
            if kind == 0:      # over a side line
                x, y = lo + 60.0 + u * (hx - lo - 120.0), (hy - inset if far else lo + inset)
                vx, vy = tang * speed, (speed if far else -speed)
            else:              # towards an end line: beside the goal (1), between the posts (2), or too slow to get there (3)
                gw = env.goalWidth
                if kind == 2:
                    y = env.H / 2 + (2.0 * u - 1.0) * (gw - 25.0)
                else:
                    y = (env.H / 2 + gw + 25.0 + u * (hy - env.H / 2 - gw - 50.0)) if u < 0.5 else (env.H / 2 - gw - 25.0 - (u - 0.5) * 2.0 * (env.H / 2 - gw - lo - 50.0))
                x = hx - inset if far else lo + inset
                sp = 3.0 if kind == 3 else speed
                vx, vy = (sp if far else -sp), tang * sp * 0.3
            b = env.ball.shape.body
            b.position = Vec2d(x, y)
            b.velocity = Vec2d(vx, vy)
            env.ball.prevPos = Vec2d(x, y)
            env.ball.lastKicked = list(kickers)
            env.space.reindex_shapes_for_body(b)
            return {}
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, can_fall, length, 7000 + SB + k, fw, setup, partial_magn=PM)
        finally:
            sys.stdout = stdout
        dump("robocup_out", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", rc_env, **rc_kw)
        except AssertionError as e:
            failures.append(("robocup_out", n, 7000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
        goals += np.asarray(out["t_goals"]).reshape(-1)[-2:].astype(np.int64)
    if n_out:
        print("RoboCup, the ball leaving the field: %d trajectories (side line %d, end line beside the goal %d, into the goal %d, staying in %d; 0-3 last kickers; "
              "6-14 steps each, %d steps; goals %d / %d; first touches robot-robot %d, robot-ball %d, robot-post %d, ball-post %d, own feet %d) against the oracle - "
              "same tolerances: %d failures; %d of the %d steps well-conditioned and checked  (%.0f s)"
              % ((n_out,) + tuple(kinds) + (steps, goals[0], goals[1]) + tuple(begins) + (len([f for f in failures if f[0] == "robocup_out"]), checked, steps, time.time() - t0)))
    # Kicks (Robot.kick, tick's kick phases :875-912: the pivot joint removed at 500 ms, the foot at 150 px/s, back at -125, the foot put back on its
    # starting position, the joint re-added; ballCollision's lastKicked): the ball at a random spot in front of a robot that kicks with a random foot,
    # a second robot kicking a few steps later, everybody else walking about
    n_kick = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    kicking = 0
    for k in range(n_kick):
        n, can_fall, length, fw = pick_n([2, 3, 5, 5]), bool(rng.random() < 0.5), int(rng.integers(6, 14)), float(rng.uniform(0.2, 0.8))
        r0, r1 = [int(x) for x in rng.choice(2 * n, 2, replace=False)]
        fx, fy, foot0, foot1, s1 = float(rng.uniform(22.0, 50.0)), float(rng.uniform(-24.0, 24.0)), int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 5))
        bvx, bvy = float(rng.uniform(-40.0, 40.0)), float(rng.uniform(-40.0, 40.0))

        def setup(env, r0=r0, r1=r1, fx=fx, fy=fy, foot0=foot0, foot1=foot1, s1=s1, bvx=bvx, bvy=bvy):
            Vec2d = gc.Vec2d
            r = env.agents[r0]
            q = r.getPos() + Vec2d(fx, fy).rotated(r.leftFoot.body.angle)
            b = env.ball.shape.body
            b.position = Vec2d(q.x, q.y)
            b.velocity = Vec2d(bvx, bvy)
            env.ball.prevPos = Vec2d(q.x, q.y)
            env.space.reindex_shapes_for_body(b)
            return {0: {r0: [0, 0, foot0, 3]}, s1: {r1: [0, 0, foot1, 3]}}
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, can_fall, length, 8000 + SB + k, fw, setup, partial_magn=PM)
        finally:
            sys.stdout = stdout
        dump("robocup_kick", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", rc_env, **rc_kw)
        except AssertionError as e:
            failures.append(("robocup_kick", n, 8000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
        kicking += int(np.asarray(out["t_states_ri"])[:, :, 7].sum())
    if n_kick:
        print("RoboCup, kicks: %d trajectories (the ball 22-50 px in front of a robot that kicks with a random foot, a second robot kicking 1-4 steps later; 6-14 steps "
              "each, %d steps; robots seen mid-kick in the recorded states %d; first touches robot-robot %d, robot-ball %d, robot-post %d, ball-post %d, own feet %d) "
              "against the oracle - same tolerances: %d failures; %d of the %d steps well-conditioned and checked  (%.0f s)"
              % ((n_kick, steps, kicking) + tuple(begins) + (len([f for f in failures if f[0] == "robocup_kick"]), checked, steps, time.time() - t0)))
    # Penalties (tick :945-995, penalize :824-859, getFreePenaltySpot): three or four robots of one team inside their own penalty box (the third is an
    # illegal defender), a robot of the other team walking off the field, one more serving a penalty that ends within the trajectory (un-penalize:
    # a free penalty spot, both feet moved there), everybody else walking about - and whatever the teleported robots run into afterwards
    n_pen = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    penalized = 0
    for k in range(n_pen):
        n, can_fall, length, fw = pick_n([3, 4, 5, 5]), bool(rng.random() < 0.5), int(rng.integers(6, 14)), float(rng.uniform(0.2, 0.8))
        team = int(rng.integers(0, 2))                        # whose box gets crowded (0: ids 0..n-1, the left box)
        m = min(n, int(rng.integers(3, 5)))
        inbox = [int(x) + team * n for x in rng.choice(n, m, replace=False)]
        ys = [float(v) for v in rng.permutation(4)[:m] * 50.0 + rng.uniform(-8.0, 8.0, m)]
        xs = [float(v) for v in rng.uniform(12.0, 48.0, m)]
        leaver = int(rng.integers(0, n)) + (1 - team) * n
        edge, eu, ed = int(rng.integers(0, 4)), float(rng.random()), float(rng.uniform(6.0, 30.0))
        others = [i for i in range(2 * n) if i not in inbox and i != leaver]
        served = int(rng.choice(others)) if others and rng.random() < 0.7 else -1
        left_ms = int(rng.integers(10, 3000))

        def setup(env, team=team, inbox=inbox, xs=xs, ys=ys, leaver=leaver, edge=edge, eu=eu, ed=ed, served=served, left_ms=left_ms):
            import math
            Vec2d = gc.Vec2d

            def put(r, x, y, a):
                for foot in (r.leftFoot, r.rightFoot):
                    foot.body.position = Vec2d(x, y)
                    foot.body.angle = a
                    env.space.reindex_shapes_for_body(foot.body)
                r.prevPos = r.getPos()
            for rid, x, y in zip(inbox, xs, ys):
                X = env.sideLength + x if team == 0 else env.W - env.sideLength - x
                put(env.agents[rid], X, env.H / 2 - 75.0 + y, 0.0 if team == 0 else math.pi)
            if edge == 0: x, y, a = 100.0 + eu * (env.W - 200.0), ed, -math.pi / 2
            elif edge == 1: x, y, a = 100.0 + eu * (env.W - 200.0), env.H - ed, math.pi / 2
            elif edge == 2: x, y, a = ed, 60.0 + eu * (env.H - 120.0), math.pi
            else: x, y, a = env.W - ed, 60.0 + eu * (env.H - 120.0), 0.0
            put(env.agents[leaver], x, y, a)
            if served >= 0:
                r = env.agents[served]
                env.penalize(r)
                r.penalTime = left_ms
                r.prevPos = r.getPos()
                for foot in (r.leftFoot, r.rightFoot):
                    env.space.reindex_shapes_for_body(foot.body)
            return {s_: {leaver: [3, 0, 0, 3]} for s_ in range(0, 14)}
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, can_fall, length, 9000 + SB + k, fw, setup, partial_magn=PM)
        finally:
            sys.stdout = stdout
        dump("robocup_penalties", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", rc_env, **rc_kw)
        except AssertionError as e:
            failures.append(("robocup_penalties", n, 9000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
        penalized += int((np.asarray(out["t_states_ri"])[:, :, 1].max(0) > np.asarray(out["t_b_ri"])[:np.asarray(out["t_states_ri"]).shape[1], 1]).sum())
    if n_pen:
        print("RoboCup, penalties: %d trajectories (3-4 robots of a team in their own box, a robot walking off the field, a penalty ending; 6-14 steps each, %d "
              "steps; robots penalized on the way %d; first touches robot-robot %d, robot-ball %d, robot-post %d, ball-post %d, own feet %d) against the oracle - same "
              "tolerances: %d failures; %d of the %d steps well-conditioned and checked  (%.0f s)"
              % ((n_pen, steps, penalized) + tuple(begins) + (len([f for f in failures if f[0] == "robocup_penalties"]), checked, steps, time.time() - t0)))
    # Driving, cars reaching their goals (tick :399-405 AtGoal / finished; step :281-285 allFinished and the team reward; the time bonus): a random
    # subset of the cars starts 60-170 px in front of its goal, heading for it at 60-160 px/s with a lateral offset; the others drive about
    n_fin = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    t0 = time.time()
    steps = touches = finished = allfin = 0
    for k in range(n_fin):
        n, seed, length, bias = pick_n([2, 3, 4, 6, 10]), 10000 + SB + k, int(rng.integers(12, 30)), float(rng.uniform(0.5, 1.0))
        chosen = rng.random(n) < float(rng.choice([0.5, 0.8, 1.0, 1.0]))
        dist, lat, spd = rng.uniform(60.0, 170.0, n), rng.uniform(-28.0, 28.0, n), rng.uniform(60.0, 160.0, n)

        def setup(env, chosen=chosen, dist=dist, lat=lat, spd=spd):
            Vec2d = type(env.agents[0].goal)
            for i, car in enumerate(env.agents):
                if not chosen[i]:
                    continue
                road = min(env.roads, key=lambda r: min((r.points[0] - car.goal).length, (r.points[1] - car.goal).length))
                u = road.direction if (road.points[1] - car.goal).length < 1e-9 else -road.direction
                pos = car.goal - u * float(dist[i]) + road.normal * float(lat[i])
                body = car.shape.body
                body.position = Vec2d(pos.x, pos.y)
                body.angle = u.angle
                body.velocity = Vec2d(u.x * float(spd[i]), u.y * float(spd[i]))
                body.angular_velocity = 0.0
                car.direction = Vec2d(u.x, u.y)
                car.prevPos = Vec2d(pos.x, pos.y)
                if hasattr(env.space, "reindex_shapes_for_body"):
                    env.space.reindex_shapes_for_body(body)
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            (gc.gen_driving_partial(out, n, seed, length, "t", bias, PM, setup=setup) if PM is not None else gc.gen_driving(out, n, seed, length, "t", bias, setup=setup))
        finally:
            sys.stdout = stdout
        dump("driving_finish", k, out)
        try:
            (tc.check_partial_trajectory(out, "t", driving_partial_env) if PM is not None else tc.check_trajectory(out, "t", driving_env))
        except AssertionError as e:
            failures.append(("driving_finish", n, seed, length, str(e)[:200]))
        steps += length
        touches += int(out["t_begins_per_step"].sum()) if PM is None else 0
        fin = np.asarray(out["t_states_cars_i"][-1] if PM is None else out["t_final_cars_i"])[:, 2]
        finished += int(np.asarray(fin).sum())
        allfin += int(np.asarray(fin).all())
    if n_fin:
        print("Driving, cars reaching their goals: %d trajectories (2-10 players, a random subset 60-170 px in front of its goal; 12-30 steps each, %d steps; %d "
              "first touches on the way; cars finished at the end %d, trajectories in which every car had %d) against the oracle - rewards / states 1e-9, "
              "observations 2e-6, flags exact: %d failures  (%.0f s)" % (n_fin, steps, touches, finished, allfin, len([f for f in failures if f[0] == "driving_finish"]), time.time() - t0))
    # Falls (tick :925-944: getting up, or falling AGAIN with the die > 0.9; fall :735-792: the push on everything within 40 px, the ball's
    # lastKicked / ownership branch, the third fall's penalty): two to four robots start on the ground with one or two falls behind them and
    # 10-1500 ms to go, the ball lies next to one of them - owned by a team, in its grace period or counting down, or free - canFall on
    n_fall = int(sys.argv[10]) if len(sys.argv) > 10 else 0
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    refalls = penal = 0
    for k in range(n_fall):
        n, length, fw = pick_n([2, 3, 5, 5]), int(rng.integers(6, 14)), float(rng.uniform(0.2, 0.8))
        down = [int(x) for x in rng.choice(2 * n, min(2 * n, int(rng.integers(2, 5))), replace=False)]
        cnt, left = [int(x) for x in rng.integers(1, 3, len(down))], [float(x) for x in rng.integers(10, 1500, len(down))]
        bang, bdist = float(rng.uniform(-math.pi, math.pi)), float(rng.uniform(24.0, 46.0))
        owned, grace, freec = int(rng.integers(-1, 2)), float(rng.choice([0.0, 0.0, 700.0, 14999.0])), float(rng.choice([0.0, 300.0, 9999.0]))
        kickers = [int(x) for x in rng.choice(2 * n, min(2 * n, int(rng.integers(0, 4))), replace=False)]

        def setup(env, down=down, cnt=cnt, left=left, bang=bang, bdist=bdist, owned=owned, grace=grace, freec=freec, kickers=kickers):
            Vec2d = gc.Vec2d
            for rid, c, ms in zip(down, cnt, left):
                r = env.agents[rid]
                r.fallen, r.fallCntr, r.fallTime = True, c, ms
            p = env.agents[down[0]].getPos()
            b = env.ball.shape.body
            b.position = Vec2d(p.x + bdist * math.cos(bang), p.y + bdist * math.sin(bang))
            env.ball.prevPos = Vec2d(b.position.x, b.position.y)
            env.ball.lastKicked = list(kickers)
            env.ballOwned = owned
            if owned != 0:
                env.gracePeriod, env.ballFreeCntr = (grace, 0.0) if grace > 0 else (0.0, freec)
            env.space.reindex_shapes_for_body(b)
            return {}
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, True, length, 11000 + SB + k, fw, setup, partial_magn=PM)
        finally:
            sys.stdout = stdout
        dump("robocup_falls", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", rc_env, **rc_kw)
        except AssertionError as e:
            failures.append(("robocup_falls", n, 11000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
        ri, b0 = np.asarray(out["t_states_ri"]), np.asarray(out["t_b_ri"])
        refalls += int((ri[:, :, 6].max(0) > b0[:ri.shape[1], 6]).sum())
        penal += int((ri[:, :, 1].max(0) > b0[:ri.shape[1], 1]).sum())
    if n_fall:
        print("RoboCup, falls: %d trajectories (2-4 robots on the ground with 10-1500 ms to go and one or two falls behind them, the ball beside one of them, owned / in "
              "its grace period / counting down / free; 6-14 steps each, %d steps; robots that fell (again) on the way %d, penalized %d; first touches robot-robot %d, "
              "robot-ball %d, robot-post %d, ball-post %d, own feet %d) against the oracle - same tolerances: %d failures; %d of the %d steps well-conditioned and checked  (%.0f s)"
              % ((n_fall, steps, refalls, penal) + tuple(begins) + (len([f for f in failures if f[0] == "robocup_falls"]), checked, steps, time.time() - t0)))
    # Duels (robotPushingDet :1010-1036 might-push flags; robotCollision :1039-1088: touch counters, the fall dice 0.9999 / 0.99995 ** touchCntr,
    # fall(), the pusher's penalty when the pushed robot goes down; separate :1091-1103): one to three pairs of opposing robots 45-80 px apart,
    # facing each other within +-0.6 rad, both told to walk forward for the whole trajectory; canFall on
    n_duel = int(sys.argv[11]) if len(sys.argv) > 11 else 0
    t0 = time.time()
    steps = checked = 0
    begins = np.zeros(5, np.int64)
    fell = penal = 0
    for k in range(n_duel):
        n, length, fw = pick_n([2, 3, 5, 5]), int(rng.integers(8, 18)), float(rng.uniform(0.2, 0.8))
        pairs = min(n, int(rng.integers(1, 4)))
        ia, ib = [int(x) for x in rng.choice(n, pairs, replace=False)], [int(x) + n for x in rng.choice(n, pairs, replace=False)]
        cx, cy = rng.uniform(200.0, 840.0, pairs), (np.arange(pairs) * 180.0 + 150.0 + rng.uniform(-30.0, 30.0, pairs))
        dirs, gaps, skew = rng.uniform(-math.pi, math.pi, pairs), rng.uniform(45.0, 80.0, pairs), rng.uniform(-0.6, 0.6, (pairs, 2))
        pre = rng.integers(0, 400, (pairs, 2))       # touch counters already run up: the dice get teeth within the trajectory

        def setup(env, ia=ia, ib=ib, cx=cx, cy=cy, dirs=dirs, gaps=gaps, skew=skew, pre=pre):
            Vec2d = gc.Vec2d
            forced = {}
            for j in range(len(ia)):
                ux, uy = math.cos(float(dirs[j])), math.sin(float(dirs[j]))
                for rid, sgn, sk, tc_ in ((ia[j], -1.0, float(skew[j][0]), int(pre[j][0])), (ib[j], 1.0, float(skew[j][1]), int(pre[j][1]))):
                    r = env.agents[rid]
                    x, y = float(cx[j]) + sgn * 0.5 * float(gaps[j]) * ux, float(cy[j]) + sgn * 0.5 * float(gaps[j]) * uy
                    a = math.atan2(-sgn * uy, -sgn * ux) + sk
                    for foot in (r.leftFoot, r.rightFoot):
                        foot.body.position = Vec2d(x, y)
                        foot.body.angle = a
                        env.space.reindex_shapes_for_body(foot.body)
                    r.prevPos = r.getPos()
                    r.touchCntr = tc_
                    for s_ in range(0, 18):
                        forced.setdefault(s_, {})[rid] = [3, 0, 0, 3]
            return forced
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            gc.gen_robocup(out, "t", n, True, length, 12000 + SB + k, fw, setup, partial_magn=PM)
        finally:
            sys.stdout = stdout
        dump("robocup_duels", k, out)
        try:
            checked += tc.check_robocup_trajectory(out, "t", rc_env, **rc_kw)
        except AssertionError as e:
            failures.append(("robocup_duels", n, 12000 + SB + k, length, str(e)[:200]))
        steps += length
        begins += out["t_begins"]
        ri, b0 = np.asarray(out["t_states_ri"]), np.asarray(out["t_b_ri"])
        fell += int((ri[:, :, 6].max(0) > b0[:ri.shape[1], 6]).sum())
        penal += int((ri[:, :, 1].max(0) > b0[:ri.shape[1], 1]).sum())
    if n_duel:
        print("RoboCup, duels: %d trajectories (1-3 pairs of opposing robots walking into each other, touch counters run up to 0-400; 8-18 steps each, %d steps; robots that "
              "fell on the way %d, penalized %d; first touches robot-robot %d, robot-ball %d, robot-post %d, ball-post %d, own feet %d) against the oracle - same tolerances: "
              "%d failures; %d of the %d steps well-conditioned and checked  (%.0f s)"
              % ((n_duel, steps, fell, penal) + tuple(begins) + (len([f for f in failures if f[0] == "robocup_duels"]), checked, steps, time.time() - t0)))
    # Driving, aimed collisions (carCrash / pedHit / carHit :587-683: speed-proportional penalties, the lane-fault extra, the responsibility test
    # cos(angle to the other - own heading) against -+0.4, crash() and the crashed friction inside the same substep, ped.die()): every car is, with
    # probability 0.8, put 45-130 px from a target - another car, a pedestrian, an obstacle - heading for it within +-0.35 rad at 50-220 px/s
    n_aim = int(sys.argv[12]) if len(sys.argv) > 12 else 0
    t0 = time.time()
    steps = touches = crashed = dead = 0
    for k in range(n_aim):
        n, seed, length, bias = pick_n([2, 4, 6, 8, 10, 10]), 13000 + SB + k, int(rng.integers(8, 24)), float(rng.uniform(0.3, 0.9))
        aimed = rng.random(n) < 0.8
        what, pick, dist = rng.integers(0, 3, n), rng.random(n), rng.uniform(45.0, 130.0, n)
        off, spd, bearing = rng.uniform(-0.35, 0.35, n), rng.uniform(50.0, 220.0, n), rng.uniform(-math.pi, math.pi, n)

        def setup(env, aimed=aimed, what=what, pick=pick, dist=dist, off=off, spd=spd, bearing=bearing):
            Vec2d = type(env.agents[0].goal)
            for i, car in enumerate(env.agents):
                if not aimed[i]:
                    continue
                if what[i] == 0 and len(env.agents) > 1:
                    others = [c for c in env.agents if c is not car]
                    tp = others[int(pick[i] * len(others)) % len(others)].shape.body.position
                elif what[i] == 1 and len(env.pedestrians):
                    tp = env.pedestrians[int(pick[i] * len(env.pedestrians)) % len(env.pedestrians)].shape.body.position
                elif len(env.obstacles):
                    tp = env.obstacles[int(pick[i] * len(env.obstacles)) % len(env.obstacles)].shape.body.position
                else:
                    continue
                b = float(bearing[i])
                pos = Vec2d(tp.x - float(dist[i]) * math.cos(b), tp.y - float(dist[i]) * math.sin(b))
                h = b + float(off[i])
                body = car.shape.body
                body.position = Vec2d(pos.x, pos.y)
                body.angle = h
                body.velocity = Vec2d(math.cos(h) * float(spd[i]), math.sin(h) * float(spd[i]))
                body.angular_velocity = 0.0
                car.direction = Vec2d(math.cos(h), math.sin(h))
                car.prevPos = Vec2d(pos.x, pos.y)
        out = {}
        stdout, sys.stdout = sys.stdout, devnull
        try:
            (gc.gen_driving_partial(out, n, seed, length, "t", bias, PM, setup=setup) if PM is not None else gc.gen_driving(out, n, seed, length, "t", bias, setup=setup))
        finally:
            sys.stdout = stdout
        dump("driving_aimed", k, out)
        try:
            (tc.check_partial_trajectory(out, "t", driving_partial_env) if PM is not None else tc.check_trajectory(out, "t", driving_env))
        except AssertionError as e:
            failures.append(("driving_aimed", n, seed, length, str(e)[:200]))
        steps += length
        touches += int(out["t_begins_per_step"].sum()) if PM is None else 0
        crashed += int((out["t_states_cars_i"][-1] if PM is None else out["t_final_cars_i"])[:, 3].sum())
        dead += int((out["t_states_peds_i"][-1] if PM is None else out["t_final_peds_i"])[:, 2].sum())
    if n_aim:
        print("Driving, aimed collisions: %d trajectories (2-10 players, 80 %% of the cars heading for another car / a pedestrian / an obstacle from 45-130 px at 50-220 px/s; "
              "8-24 steps each, %d steps; %d first touches, %d cars crashed, %d pedestrians killed) against the oracle - rewards / states 1e-9, observations 2e-6, flags "
              "exact: %d failures  (%.0f s)" % (n_aim, steps, touches, crashed, dead, len([f for f in failures if f[0] == "driving_aimed"]), time.time() - t0))
    if PM is not None:
        print("Partial observations (noise magnitude %g): %d (step, snapshot, robot) rows excused - a penalized robot's own side line, seen as a zero-length line at the "
              "robot or not at all by the last bit of libm's sin / cos" % (PM, len(excused)))
    for f in failures:
        print("FAILURE", f)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
