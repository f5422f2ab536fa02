#!/usr/bin/env python3
"""Closing the parity gap where pymunk exists (VERDICT r5 items 1 + 2; INTEGRATION.md "closing the parity gap").

What this pipeline cannot pin is `pymunk.Space.step` itself (DynEnv/DrivingEnvironment.py:278, RoboCupEnvironment.py:482; handlers
environment_base.py:179-188, space :126-128): pymunk / Chipmunk2D are not installed here.  This tool runs the reference's OWN
`DrivingEnvironment.step()` / `RoboCupEnvironment.step()` on an engine -

    --engine kat_general   the pymunk-5 facade over tests/kat_general.py (works in this repository's build container)
    --engine pymunk        the real `pymunk` 5.x of the machine it is started on: no facade, the reference untouched

- feeds the same scene, the same actions and the same random draws (served from the oracle's Philox words, as the golden generators do)
to the CPU oracle through ctypes, and compares the two SUBSTEP BY SUBSTEP: positions, angles and velocities of every dynamic body after
every `space.step`, rewards after every env step.  Per trajectory it prints the first substep that deviates by more than 1e-9 and by more
than 1e-4 (relative to max(1, |value|)) and - the one thing nobody without pymunk can know - whether the engine's arbiter order in
that substep was the ascending-shape-id order the oracle and the kernels use (logged through `post_solve`, which Chipmunk calls per
active arbiter in solver order).  Only the engine's PUBLIC pymunk API is used, so the code path that talks to the real pymunk is the one
exercised here against the facade (tests/test_pymunk_crosscheck.py).  Needs /root/reference (or REFERENCE_ROOT) and the CPU oracle;
never the GPU.

    python3 tools/pymunk_crosscheck.py --engine kat_general --fixture a        # reproduces tests/golden/driving_contacts.npz, trajectory a
    python3 tools/pymunk_crosscheck.py --engine pymunk --driving 50 --robocup 20
"""
import argparse
import math
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

DRIVING_FIXTURES = {"a": (10, 3, 150, 0.5), "b": (10, 21, 150, 0.7), "c": (6, 8, 200, 0.6), "d": (10, 34, 120, 0.3), "e": (2, 13, 250, 0.85),
                    "f": (10, 41, 150, 0.6), "g": (10, 42, 150, 0.4), "h": (8, 43, 150, 0.7), "i": (10, 44, 200, 0.5), "j": (4, 45, 200, 0.8)}
RNG_ROBO_STEP = 8


# ------------------------------------------------------------------------------------------------ engines
def install_engine(name, kat_order="canonical"):
    """-> module-like namespace `gg` (tests/golden/gen_golden.py) with the reference importable; the engine behind `import pymunk`"""
    import gen_golden as gg
    if name == "kat_general":
        import gen_golden_contacts as gc
        import kat_general as kg
        gc.install()
        if not hasattr(kg.World, "_crosscheck_init"):
            kg.World._crosscheck_init = kg.World.__init__
        init = kg.World._crosscheck_init
        # --kat-order reversed: kat_general hands its pairs over last first - an engine with ANOTHER arbiter order than the oracle's, to
        # see this tool report one (what a run against the real pymunk will show wherever the BB-tree's order is not ascending ids)
        kg.World.__init__ = lambda self, order=kat_order, **kw: init(self, order=order, **kw)
        return gg
    try:
        import pymunk as real
        import pymunk.constraint as real_cons
    except ImportError:
        sys.exit("pymunk_crosscheck: --engine pymunk needs the `pymunk` module (5.x, the line DynEnv was written against: in-place Vec2d.rotate, "
                 "SURVEY F3) on this machine; it is not installed here.  Use --engine kat_general, or run this tool where pymunk exists "
                 "(INTEGRATION.md, \"closing the parity gap\").")
    major = int(str(getattr(real, "version", "0")).split(".")[0] or 0)
    if major != 5:
        sys.exit("pymunk_crosscheck: pymunk %s found; DynEnv needs the 5.x line (Vec2d.rotate in place, Car.py:36; pymunkoptions): pip install 'pymunk<6'" % getattr(real, "version", "?"))
    gg.install_standins()                      # gym / pygame / cv2 stand-ins + the DynEnv package shell ... and a fake pymunk, replaced right away:
    stub_draw = sys.modules["pymunk.pygame_util"]
    sys.modules["pymunk"], sys.modules["pymunk.constraint"] = real, real_cons
    real.pygame_util = stub_draw               # (environment_base.py imports it for rendering only)
    gg.Vec2d = real.Vec2d
    # the state dumps of the golden generators read `_p` / `_v`: on the real Body these are its position / velocity
    real.Body._p = property(lambda b: b.position)
    real.Body._v = property(lambda b: b.velocity)
    return gg


def name_shapes(space, sid_of):
    """kat_general finds pairs in the order of the ids it is given (the oracle's slots); the real engine has its own order"""
    if hasattr(space, "reindex"):
        space.sid_of = sid_of
        space.reindex()


class SubstepLog(object):
    """after every `space.step` of the reference's step(): the dynamic bodies' public state in the oracle's body order, and the arbiters
    that reached post_solve, in the engine's solver order, as (sid of the lower shape, sid of the higher)"""

    def __init__(self, env, bodies, sid_of):
        self.env, self.bodies, self.sid_of = env, bodies, sid_of
        self.states, self.orders, self._cur = [], [], []
        space = env.space
        inner = space.step

        def step(dt):
            self._cur = []
            r = inner(dt)
            self.states.append([[b.position[0], b.position[1], b.angle, b.velocity[0], b.velocity[1], b.angular_velocity] for b in self.bodies])
            self.orders.append(list(self._cur))
            return r
        space.step = step
        registered = dict(getattr(space, "_handlers", None) or getattr(space, "handlers"))      # pymunk 5: Space._handlers {(a, b): CollisionHandler}
        types_ = sorted(set(int(s.collision_type) for s in space.shapes if hasattr(s, "collision_type")))
        for i, a in enumerate(types_):
            for b in types_[i:]:
                key = (a, b) if (a, b) in registered else (b, a) if (b, a) in registered else None
                h = registered[key] if key is not None else space.add_collision_handler(a, b)    # (a pair the reference left to the default handler)
                self._wrap(h)

    def _wrap(self, h):
        orig = h.post_solve
        log = self

        def post_solve(arbiter, space, data):
            a, b = (log.sid_of(s) for s in arbiter.shapes)
            log._cur.append((min(a, b), max(a, b)))
            if orig is not None:
                return orig(arbiter, space, data)
        h.post_solve = post_solve


def ascending(order):
    return all(order[i] <= order[i + 1] for i in range(len(order) - 1))


# ------------------------------------------------------------------------------------------------ the oracle side
class OracleSide(object):
    def __init__(self, ol, env, cap):
        import ctypes as C
        self.env, self.l = env, ol.lib()
        self.states = np.zeros((cap, 32, 6))
        self.arbs = np.zeros((cap, 40), np.int32)
        self.l.oracle_trace_begin.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int]
        self.cap, self.C = cap, C

    def arm(self):
        self.nb = self.l.oracle_trace_begin(self.env.h, 0, self.states.ctypes.data_as(self.C.c_void_p), self.arbs.ctypes.data_as(self.C.c_void_p), self.cap)

    def order(self, k):
        n = int(self.arbs[k, 0])
        return [(min(int(c) >> 8, int(c) & 255), max(int(c) >> 8, int(c) & 255)) for c in self.arbs[k, 1:1 + n]]


def compare(tag, log, ora, rewards_ref, rewards_ora, substeps_per_step):
    ref = np.array(log.states, float)                        # [K, nb, 6]
    K, nb = ref.shape[0], ref.shape[1]
    got = ora.states[:K, :nb]
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    per = err.reshape(K, -1).max(1)
    rerr = np.abs(np.array(rewards_ora) - np.array(rewards_ref)) / np.maximum(1.0, np.abs(np.array(rewards_ref)))
    first = {}
    for thr in (1e-9, 1e-4):
        hit = np.nonzero(per > thr)[0]
        first[thr] = int(hit[0]) if len(hit) else None
    touching = sum(1 for o in log.orders if o)
    multi = sum(1 for o in log.orders if len(o) > 1)
    foreign = [k for k, o in enumerate(log.orders) if not ascending(o)]
    line = "%s: %d substeps (%d with an active arbiter, %d with more than one), largest deviation %.2e, rewards %.2e" % (tag, K, touching, multi, float(per.max()) if K else 0.0, float(rerr.max()) if rerr.size else 0.0)
    for thr in (1e-9, 1e-4):
        k = first[thr]
        if k is None:
            line += "; never above %.0e" % thr
        else:
            eo, oo = log.orders[k], ora.order(k)
            line += "; first above %.0e at substep %d (env step %d): engine order %s %s, oracle order %s" % (
                thr, k, k // substeps_per_step, eo, "= ascending shape ids" if ascending(eo) else "NOT ascending shape ids", oo)
    line += "; substeps whose engine order is not ascending: %d%s" % (len(foreign), (" (first %d)" % foreign[0]) if foreign else "")
    print(line)
    sys.stdout.flush()
    return dict(first=first, max_dev=float(per.max()) if K else 0.0, reward_dev=float(rerr.max()) if rerr.size else 0.0, foreign=len(foreign), substeps=K,
                touching=touching)


# ------------------------------------------------------------------------------------------------ Driving
def run_driving(gg, ol, n_players, seed, steps, bias, tag):
    import gen_golden_contacts as gc
    from test_oracle_golden import _state_from_npz
    env, de, cut = gg.make_driving(n_players, seed)
    sid_of = gc.driving_sids(env)
    name_shapes(env.space, sid_of)
    pseed, genv, episode = 42, seed, 1
    tape = gg.PedTape(pseed, genv, episode)
    orig_move = env.move

    def move_with_tape(ped):
        tape.arm(env.pedestrians.index(ped), env.elapsed)
        return orig_move(ped)
    env.move = move_with_tape
    de.random.randint, de.random.random = tape.randint, tape.random
    try:
        z = {"t_init_%s" % k: v for k, v in gg.dump_state(env, cut).items()}
        st0 = _state_from_npz(z, "t", "init", (pseed, genv, episode))
        oenv = ol.OracleEnv(num_envs=1, n_players=n_players, seed=pseed, env_id_offset=genv)
        oenv.reset()
        oenv.set_state(0, st0)
        ora = OracleSide(ol, oenv, steps * 10)
        ora.arm()
        log = SubstepLog(env, [c.shape.body for c in env.agents] + [p.shape.body for p in env.pedestrians], sid_of)
        A = len(env.agents)
        arng = np.random.RandomState(seed + 100)
        r_ref, r_ora = [], []
        for s in range(steps):
            a = np.where(arng.rand(A, 2) < bias, [2, 1], arng.randint(0, 3, size=(A, 2)))
            obs, r, d, info = env.step(a)
            o2, r2, d2 = oenv.step(a[None].astype(np.int32))
            r_ref.append(np.array(r, float)); r_ora.append(r2[0].copy())
        res = compare(tag, log, ora, r_ref, r_ora, 10)
        res["rewards"] = np.array(r_ref)
        return res
    finally:
        de.random.randint, de.random.random = gg._ORIG_RANDINT, gg._ORIG_RANDOM


# ------------------------------------------------------------------------------------------------ RoboCup
class Dice(object):
    """random.random() inside RoboCupEnvironment.step(), served by the source line of the draw site from the oracle's Philox words
    (tests/golden/gen_golden_contacts.py RoboDice, with the shape ids looked up instead of read off the facade)"""

    def __init__(self, gg, key, env, sid_of):
        self.gg, self.key, self.env, self.sid_of = gg, key, env, sid_of

    def __call__(self):
        f = sys._getframe(1)
        line = f.f_lineno
        if line in (557, 565, 577, 932):
            robot = f.f_locals["robot"]
            word = {557: 0, 565: 1, 577: 2, 932: 0}[line]
            ent = robot.id | ((1 << 8) if line == 932 else 0)
        else:
            a, b = (self.sid_of(sh) for sh in f.f_locals["arbiter"].shapes)
            word = {1064: 0, 1068: 1, 1121: 0}[line]
            ent = (min(a, b) * 32 + max(a, b)) | ((3 if line == 1121 else 2) << 16)
        return self.gg.env_rng(self.key[0], self.key[1], self.key[2], RNG_ROBO_STEP, ent, int(self.env.elapsed))[word] * 2.0 ** -32


def run_robocup(gg, ol, n, can_fall, steps, seed, forward, tag):
    import gen_golden_contacts as gc
    import gen_golden_robocup as gr
    from test_oracle_golden_robocup import _to_state
    env, rc_mod, cut = gr.make_env(n, seed, can_fall)
    sid_of = gc.robocup_sids(env)
    name_shapes(env.space, sid_of)
    key = (42, seed, 1)
    rc_mod.random.random = Dice(gg, key, env, sid_of)
    try:
        rf, ri, sc, fl = gr.dump(env)[:4]
        flags = (ol.FLAG_CAN_FALL if can_fall else 0) | ol.FLAG_USE_OBS_REWARDS
        oenv = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, seed=key[0], env_id_offset=key[1], flags=flags)
        oenv.reset()
        oenv.set_state(0, _to_state(rf, ri, sc, fl, key[2]))
        ora = OracleSide(ol, oenv, steps * 50)
        ora.arm()
        bodies = []
        for r in env.agents:
            bodies += [r.leftFoot.body, r.rightFoot.body]
        log = SubstepLog(env, bodies + [env.ball.shape.body], sid_of)
        arng = np.random.RandomState(seed + 7)
        r_ref, r_ora = [], []
        for s in range(steps):
            a = np.stack([arng.randint(0, k, 2 * n) for k in (5, 3, 3, 7)], -1)
            fw = arng.rand(2 * n) < forward
            a[fw, 0], a[fw, 1] = 3, 0
            if s % 5 == 4:
                a[:, 0] = 0; a[:, 1] = 0
            obs, r, done, info = env.step(a.copy())
            o2, r2, d2 = oenv.step(a[None].astype(np.int32))
            r_ref.append(np.array(r, float)); r_ora.append(r2[0].copy())
        return compare(tag, log, ora, r_ref, r_ora, 50)
    finally:
        rc_mod.random.random = gg._ORIG_RANDOM
        rc_mod.RoboCupEnvironment.canFall = True


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--engine", choices=("kat_general", "pymunk"), required=True)
    ap.add_argument("--fixture", action="append", default=[], help="a Driving trajectory of tests/golden/driving_contacts.npz (a..j): same scene, same actions; with --engine kat_general its rewards must be the committed ones bit for bit")
    ap.add_argument("--driving", type=int, default=0, help="fresh random Driving trajectories (2-10 players, 40-90 env steps)")
    ap.add_argument("--robocup", type=int, default=0, help="fresh random RoboCup trajectories (2-5 a side, 12-30 env steps)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--kat-order", choices=("canonical", "reversed"), default="canonical", help="kat_general only: the order in which it finds colliding pairs")
    args = ap.parse_args(argv)
    import oracle_lib as ol
    ol.build()
    gg = install_engine(args.engine, args.kat_order)
    print("pymunk_crosscheck: engine %s; the reference's own step() against the CPU oracle, substep by substep" % args.engine)
    results = []
    for tag in args.fixture:
        n, seed, steps, bias = DRIVING_FIXTURES[tag]
        res = run_driving(gg, ol, n, seed, steps, bias, "driving fixture " + tag)
        z = np.load(os.path.join(ROOT, "tests", "golden", "driving_contacts.npz"))
        same = np.array_equal(res["rewards"], z[tag + "_rewards"])
        print("   rewards of the %d env steps %s the committed fixture's" % (steps, "ARE bit for bit" if same else "are NOT"))
        if args.engine == "kat_general" and args.kat_order == "canonical" and not same:
            sys.exit("pymunk_crosscheck: the kat_general engine must reproduce tests/golden/driving_contacts.npz")
        results.append(res)
    rng = np.random.default_rng(args.seed)
    for k in range(args.driving):
        n, seed, length, bias = int(rng.choice([2, 4, 6, 8, 10, 10, 10])), 1000 + k, int(rng.integers(40, 90)), float(rng.uniform(0.3, 0.9))
        results.append(run_driving(gg, ol, n, seed, length, bias, "driving %d players seed %d" % (n, seed)))
    for k in range(args.robocup):
        n, can_fall, length, fw = int(rng.choice([2, 3, 4, 5, 5])), bool(rng.random() < 0.6), int(rng.integers(12, 30)), float(rng.uniform(0.4, 0.9))
        results.append(run_robocup(gg, ol, n, can_fall, length, 2000 + k, fw, "robocup %d a side seed %d canFall %d" % (n, 2000 + k, can_fall)))
    if results:
        n9 = sum(1 for r in results if r["first"][1e-9] is None)
        n4 = sum(1 for r in results if r["first"][1e-4] is None)
        print("summary: %d trajectories, %d substeps (%d with an active arbiter): %d within 1e-9 throughout, %d within 1e-4 throughout; trajectories in which the "
              "engine's arbiter order left ascending shape ids at least once: %d" % (len(results), sum(r["substeps"] for r in results), sum(r["touching"] for r in results),
                                                                                 n9, n4, sum(1 for r in results if r["foreign"])))
    return results


if __name__ == "__main__":
    main()
