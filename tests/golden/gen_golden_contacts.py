#!/usr/bin/env python3
"""Golden fixtures of the reference's OWN `DrivingEnvironment.step()` / `RoboCupEnvironment.step()` WITH collisions (round 5).

tests/golden/gen_golden.py runs the reference's step() on a free-flight `Space` stand-in: its fixtures are valid only while nothing
touches.  Here the stand-in for `pymunk.Space` is a FUNCTIONAL one: a thin pymunk-5 facade (Body / Poly / Circle / Segment / PivotJoint /
RotaryLimitJoint / collision handlers with begin, post_solve, separate / arbiter.shapes / point_query) over tests/kat_general.py - the
independent float64 restatement of Chipmunk's `Space.step` (GJK / EPA narrowphase, written from the published algorithm, not from
oracle/cp_lite.c).  So the trajectories below are computed by the reference's own Python (processAction, tick, move, the collision
callbacks carCrash / pedHit / carHit ..., friction velocity functions, rewards, observations) on top of an engine that shares no code
with the oracle, and they DO contain crashes, pushes and resting contacts.  What this pins beyond the free-flight fixtures: the
composition of callbacks and physics inside one step (a `begin` that crashes a car switches its friction function before the same
substep's velocity update; pedHit's return value hides a pair; rewards of the substep in which a contact begins).  What it cannot pin:
pymunk / Chipmunk itself (not in this pipeline) - kat_general is a second reading of it, not the thing.

Shape ids (= the order in which pairs are found and callbacks fire, which Chipmunk derives from its BB-tree and nobody here can know)
follow the oracle's slots: cars 0.., pedestrians 10.., obstacles 30.., buildings 50..; robots' feet 2 k / 2 k + 1, ball 20, posts 21...

RNG: as in gen_golden.py, the reference's `random.*` calls inside `move` are served from the oracle's Philox words.

Usage:  python tests/golden/gen_golden_contacts.py   (writes tests/golden/driving_contacts.npz, robocup_contacts.npz, driving_partial_contacts.npz, robocup_partial_contacts.npz)"""
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg  # noqa: E402  (Vec2d, install_standins, PedTape, dump_state, flat_obs, make_driving)
import kat_general as kg  # noqa: E402

Vec2d = gg.Vec2d


# ------------------------------------------------------------------ the pymunk-5 facade over kat_general
class Body(object):
    """pymunk.Body: a view over a kat_general.Body (position / velocity getters return copies, like pymunk 5)"""
    DYNAMIC, KINEMATIC, STATIC = 0, 1, 2

    def __init__(self, mass=0, moment=0, body_type=0):
        self.k = kg.Body(mass if body_type == Body.DYNAMIC else 0.0, moment, 0.0, 0.0)
        self.k.user = self
        self.body_type = body_type
        self._vf = None
        self._f, self._t = Vec2d(0.0, 0.0), 0.0

    mass = property(lambda s: s.k.m)
    moment = property(lambda s: s.k.i)

    def _set_p(self, v):
        self.k.px, self.k.py = float(v[0]), float(v[1])

    def _set_v(self, v):
        self.k.vx, self.k.vy = float(v[0]), float(v[1])

    def _set_a(self, a):
        self.k.a = float(a)
        self.k.cos, self.k.sin = math.cos(self.k.a), math.sin(self.k.a)

    def _set_w(self, w):
        self.k.w = float(w)

    position = property(lambda s: Vec2d(s.k.px, s.k.py), _set_p)
    velocity = property(lambda s: Vec2d(s.k.vx, s.k.vy), _set_v)
    _p = property(lambda s: Vec2d(s.k.px, s.k.py))   # (gen_golden.dump_state reads these)
    _v = property(lambda s: Vec2d(s.k.vx, s.k.vy))
    angle = property(lambda s: s.k.a, _set_a)
    angular_velocity = property(lambda s: s.k.w, _set_w)

    def _get_vf(self):
        return self._vf

    def _set_vf(self, f):
        self._vf = f
        self.k.vel_func = None if f is None else (lambda kb, dt: f(self, (0.0, 0.0), 1.0, dt))   # damping ** dt with damping = 1

    velocity_func = property(_get_vf, _set_vf)

    @staticmethod
    def update_velocity(body, gravity, damping, dt):   # cpBodyUpdateVelocity
        k = body.k
        k.vx = k.vx * damping + (gravity[0] + body._f.x * k.m_inv) * dt
        k.vy = k.vy * damping + (gravity[1] + body._f.y * k.m_inv) * dt
        k.w = k.w * damping + body._t * k.i_inv * dt
        body._f, body._t = Vec2d(0.0, 0.0), 0.0

    def apply_force_at_world_point(self, force, point):
        rx, ry = point[0] - self.k.px, point[1] - self.k.py
        self._f = Vec2d(self._f.x + force[0], self._f.y + force[1])
        self._t += rx * force[1] - ry * force[0]


class _ShapeBase(object):
    color = None

    def _mk(self, body, kshape_factory):
        self.body = body
        self._factory = kshape_factory
        self.k = None
        self._e, self._u, self._ct = 0.0, 0.0, 0

    def _set(self, name, v):
        setattr(self, name, v)
        if self.k is not None:
            self.k.e, self.k.u, self.k.ctype = self._e, self._u, self._ct

    elasticity = property(lambda s: s._e, lambda s, v: s._set("_e", float(v)))
    friction = property(lambda s: s._u, lambda s, v: s._set("_u", float(v)))
    collision_type = property(lambda s: s._ct, lambda s, v: s._set("_ct", int(v)))

    def realise(self, sid):
        self.k = self._factory(sid)
        self.k.e, self.k.u, self.k.ctype = self._e, self._u, self._ct
        self.k.user = self
        return self.k


class Poly(_ShapeBase):
    def __init__(self, body, vertices, transform=None, radius=0):
        xs, ys = sorted(set(abs(float(v[0])) for v in vertices)), sorted(set(abs(float(v[1])) for v in vertices))
        assert len(vertices) == 4 and len(xs) == 1 and len(ys) == 1 and radius == 0, "DynEnv only makes centred boxes (Car.py:21-22, Obstacle.py:12)"
        self._mk(body, lambda sid: kg.Shape(kg.POLY, body.k, sid, hx=xs[0], hy=ys[0]))


class Circle(_ShapeBase):
    def __init__(self, body, radius, offset=(0, 0)):
        assert offset[0] == 0 and offset[1] == 0
        self._mk(body, lambda sid: kg.Shape(kg.CIRCLE, body.k, sid, r=float(radius)))


class Segment(_ShapeBase):
    def __init__(self, body, a, b, radius):
        self._mk(body, lambda sid: kg.Shape(kg.SEGMENT, body.k, sid, r=float(radius), la=(float(a[0]), float(a[1])), lb=(float(b[0]), float(b[1]))))


class PivotJoint(object):
    def __init__(self, a, b, pivot):
        def local(body):   # cpBodyWorldToLocal
            dx, dy = pivot[0] - body.k.px, pivot[1] - body.k.py
            return (body.k.cos * dx + body.k.sin * dy, -body.k.sin * dx + body.k.cos * dy)
        self.k = kg.PivotJoint(a.k, b.k, local(a), local(b))
        self.k.user = self
    error_bias = property(lambda s: s.k.error_bias, lambda s, v: setattr(s.k, "error_bias", float(v)))


class RotaryLimitJoint(object):
    def __init__(self, a, b, lo, hi):
        self.k = kg.RotaryLimitJoint(a.k, b.k, float(lo), float(hi))
        self.k.user = self
    error_bias = property(lambda s: s.k.error_bias, lambda s, v: setattr(s.k, "error_bias", float(v)))


class _Arbiter(object):
    def __init__(self, karb):
        a, b = karb.shapes()          # the handler's declared type order
        self.shapes = (a.user, b.user)


class _Handler(object):
    begin = post_solve = separate = None


class _PointQueryInfo(object):
    def __init__(self, shape):
        self.shape = shape


class Space(object):
    """pymunk.Space over kat_general.World.  `sid_of` (set by the generator after the reference has built its scene) names every shape's
    id = the oracle's slot; shapes are handed to the world at the first step()."""

    def __init__(self):
        self.gravity = (0.0, 0.0)
        self.world = kg.World()
        self._bodies, self._shapes, self._pending = [], [], []
        self.sid_of = None
        self.handlers = {}

    def add(self, *objs):
        for o in objs:
            if isinstance(o, Body):
                self._bodies.append(o)
                self.world.add_body(o.k)
            elif isinstance(o, (PivotJoint, RotaryLimitJoint)):
                self.world.joints.append(o.k)
            else:
                self._shapes.append(o)
                self._pending.append(o)

    def remove(self, *objs):
        for o in objs:
            assert isinstance(o, (PivotJoint, RotaryLimitJoint)), "DynEnv only removes joints (RoboCupEnvironment.py:884-886)"
            js = self.world.joints
            i = js.index(o.k)     # cpArrayDeleteObj: the LAST element moves into the hole
            js[i] = js[-1]
            js.pop()

    def add_collision_handler(self, a, b):
        key = (int(a), int(b))
        if key not in self.handlers:
            h = self.handlers[key] = _Handler()
            space = self

            def call(which, default):
                def f(karb, world):
                    fn = getattr(h, which)
                    return default if fn is None else fn(_Arbiter(karb), space, None)
                return f
            self.world.handlers.append(kg.Handler(int(a), int(b), begin=call("begin", True), post_solve=call("post_solve", None), separate=call("separate", None)))
        return self.handlers[key]

    def point_query(self, point, max_distance, shape_filter):
        out = []
        for s in self.world.shapes:     # canonical id order (pymunk's order is unspecified)
            if s.kind == kg.CIRCLE:
                d = math.hypot(point[0] - s.tc[0], point[1] - s.tc[1]) - s.r
            elif s.kind == kg.SEGMENT:
                ex, ey = s.tb[0] - s.ta[0], s.tb[1] - s.ta[1]
                t = max(0.0, min(1.0, (ex * (point[0] - s.ta[0]) + ey * (point[1] - s.ta[1])) / (ex * ex + ey * ey)))
                d = math.hypot(point[0] - (s.ta[0] + ex * t), point[1] - (s.ta[1] + ey * t)) - s.r
            else:
                continue
            if d < max_distance:
                out.append(_PointQueryInfo(s.user))
        return out

    def reindex_shapes_for_body(self, body):
        """cpSpaceReindexShapesForBody: refresh the cached world geometry of a body that was moved by hand (point_query reads the cache)"""
        self.reindex()
        for s in self.world.shapes:
            if s.body is body.k:
                s.update()

    shapes = property(lambda s: s._shapes + [j.user for j in s.world.joints])   # (gen_golden_robocup.dump counts the joints in the space)
    bodies = property(lambda s: s._bodies)

    def reindex(self):
        """shapes are handed to the world (cpSpaceAddShape caches their world geometry) once the generator has named their ids"""
        for sh in self._pending:
            self.world.add_shape(sh.realise(self.sid_of(sh)))
        self._pending = []

    def step(self, dt):
        self.reindex()
        for s in self.world.shapes:
            if s.body.static:
                s.update()
        return self.world.step(dt)


def install():
    """gen_golden's stand-ins for gym / pygame / cv2 and Vec2d, with the functional pymunk facade on top"""
    gg.install_standins()
    pm = sys.modules["pymunk"]
    pm.Body, pm.Space, pm.Poly, pm.Circle, pm.Segment = Body, Space, Poly, Circle, Segment
    cons = sys.modules["pymunk.constraint"]
    cons.PivotJoint, cons.RotaryLimitJoint = PivotJoint, RotaryLimitJoint
    pm.PivotJoint, pm.RotaryLimitJoint = PivotJoint, RotaryLimitJoint
    sf = types.ModuleType("pymunk.shape_filter")
    sf.ShapeFilter = lambda **k: None
    pm.shape_filter = sf
    sys.modules["pymunk.shape_filter"] = sf


# ------------------------------------------------------------------ Driving trajectories with collisions
def driving_sids(env):
    m = {}
    for k, c in enumerate(env.agents):
        m[id(c.shape)] = k
    for k, p in enumerate(env.pedestrians):
        m[id(p.shape)] = 10 + k
    for k, o in enumerate(env.obstacles):
        m[id(o.shape)] = 30 + k
    for k, b in enumerate(env.buildings):
        m[id(b.shape)] = 50 + k
    return lambda sh: m[id(sh)]


def setup_finish(env):
    """every car 70 px in front of its goal (the end of a road, DrivingEnvironment.py:91), heading for it at 100 px / s, side by side: they
    leave the road within 100 px of the goal one after the other (tick :399-405: AtGoal, finished, not crashed) and the last one sets
    allFinished with its team reward (:281-285)"""
    Vec2d = type(env.agents[0].goal)
    per_goal = {}
    for car in env.agents:
        road = min(env.roads, key=lambda r: min((r.points[0] - car.goal).length, (r.points[1] - car.goal).length))
        to_end1 = (road.points[1] - car.goal).length < 1e-9
        u = road.direction if to_end1 else -road.direction
        k = per_goal.setdefault((car.goal.x, car.goal.y), [])
        lateral = (len(k) - 1) * 25.0    # -25, 0, 25: inside both roads' width, clear of each other (half widths 5-8)
        k.append(car)
        pos = car.goal - u * (70.0 + 3.0 * len(k)) + road.normal * lateral
        body = car.shape.body
        body.position = Vec2d(pos.x, pos.y)
        body.angle = u.angle
        body.velocity = Vec2d(u.x * 100.0, u.y * 100.0)
        body.angular_velocity = 0.0
        car.direction = Vec2d(u.x, u.y)
        car.prevPos = Vec2d(pos.x, pos.y)


def gen_driving(out, n_players, seed, steps, tag, action_bias, obs_every=1, setup=None):
    env, de, cut = gg.make_driving(n_players, seed)
    env.space.sid_of = driving_sids(env)
    if setup is not None:
        setup(env)
    pseed, genv, episode = 42, seed, 1
    tape = gg.PedTape(pseed, genv, episode)
    orig_move = env.move

    def move_with_tape(ped):
        tape.arm(env.pedestrians.index(ped), env.elapsed)
        return orig_move(ped)
    env.move = move_with_tape
    de.random.randint, de.random.random = tape.randint, tape.random
    A = len(env.agents)
    dim = 9 + (A - 1) * 7 + 80 + 40 + 40
    for k, v in gg.dump_state(env, cut).items():
        out["%s_init_%s" % (tag, k)] = v
    out["%s_key" % tag] = np.array([pseed, genv, episode], np.int64)
    arng = np.random.RandomState(seed + 100)
    acts, rews, dones, obss, states, ncontact = [], [], [], [], [], []
    for s in range(steps):
        # mostly "accelerate": cars that drive reach each other, the buildings, the pedestrians
        a = np.where(arng.rand(A, 2) < action_bias, [2, 1], arng.randint(0, 3, size=(A, 2)))
        before = len(env.space.world.log)
        obs, r, d, info = env.step(a)
        acts.append(a); rews.append(np.array(r, float)); dones.append(int(d))
        if s % obs_every == obs_every - 1 or s == steps - 1:
            obss.append(gg.flat_obs(env, obs[0], A, dim))
        if d:
            out["%s_episode_info" % tag] = np.array([info["episode_r"], info["episode_p_r"], info["episode_o_r"]], float)
            out["%s_episode_goals" % tag] = np.array(info["episode_g"], float).reshape(-1)
        ncontact.append(sum(1 for ev in env.space.world.log[before:] if ev[1] == "begin"))
        if s % 10 == 9 or s == steps - 1:
            st = gg.dump_state(env, cut)
            states.append((s, st))
    out["%s_actions" % tag] = np.array(acts, np.int64)
    out["%s_rewards" % tag] = np.array(rews)
    out["%s_dones" % tag] = np.array(dones, np.int64)
    out["%s_obs" % tag] = np.array(obss, np.float32)
    out["%s_obs_every" % tag] = np.array([obs_every], np.int64)
    out["%s_begins_per_step" % tag] = np.array(ncontact, np.int64)
    out["%s_state_steps" % tag] = np.array([s for s, _ in states], np.int64)
    for name in ("cars_f", "cars_i", "peds_f", "peds_i", "episode_r", "episode_pos_r", "scalars"):
        out["%s_states_%s" % (tag, name)] = np.array([st[name] for _, st in states])
    log = env.space.world.log
    crashed = int(sum(int(c.crashed) for c in env.agents))
    dead = int(sum(int(p.dead) for p in env.pedestrians))
    print("%s: %d players, %d steps: %d first touches (%d separations, %d re-touches), %d cars crashed, %d pedestrians dead" %
          (tag, n_players, steps, sum(1 for e in log if e[1] == "begin"), sum(1 for e in log if e[1] == "separate"),
           sum(1 for e in log if e[1] == "retouch"), crashed, dead))
    de.random.randint, de.random.random = gg._ORIG_RANDINT, gg._ORIG_RANDOM


# ------------------------------------------------------------------ RoboCup trajectories with collisions
RNG_ROBO_STEP = 8


class RoboDice(object):
    """random.random() inside RoboCupEnvironment.step(), served by the source line of the draw site from the oracle's Philox words:
    processAction :557 / :565 / :577 -> words 0 / 1 / 2 of block (ROBO_STEP, robot, elapsed); getup :932 -> word 0 of (robot | 1 << 8);
    robotCollision :1064 / :1068 -> words 0 / 1 of (pair key | 2 << 16); goalpostCollision :1121 -> word 0 of (pair key | 3 << 16), the pair
    key being lo * 32 + hi of the two shapes' ids (oracle/robocup.c pair_entity)"""

    def __init__(self, key, env):
        self.key, self.env, self.draws = key, env, 0

    def __call__(self):
        return self.draw(sys._getframe(1))

    def draw(self, f):
        line = f.f_lineno
        if line in (557, 565, 577, 932):
            robot = f.f_locals["robot"]
            word = {557: 0, 565: 1, 577: 2, 932: 0}[line]
            ent = robot.id | ((1 << 8) if line == 932 else 0)
        else:
            a, b = (sh.k.sid for sh in f.f_locals["arbiter"].shapes)
            word = {1064: 0, 1068: 1, 1121: 0}[line]
            ent = (min(a, b) * 32 + max(a, b)) | ((3 if line == 1121 else 2) << 16)
        self.draws += 1
        return gg.env_rng(self.key[0], self.key[1], self.key[2], RNG_ROBO_STEP, ent, int(self.env.elapsed))[word] * 2.0 ** -32


def robocup_sids(env):
    m = {}
    for r in env.agents:
        m[id(r.leftFoot)], m[id(r.rightFoot)] = 2 * r.id, 2 * r.id + 1
    m[id(env.ball.shape)] = 20
    for k, g in enumerate(env.goalposts):
        m[id(g.shape)] = 21 + k
    return lambda sh: m[id(sh)]


def setup_kick(env):
    """robot 0 stands right behind the ball and kicks it with its left foot (joint removed at 500 ms, foot at 150 px/s, RoboCupEnvironment.py:875-912)"""
    r = env.agents[0]
    p, ang = r.getPos(), r.leftFoot.body.angle
    q = p + Vec2d(31.0, 10.0).rotated(ang)
    env.ball.shape.body.position = q
    env.ball.prevPos = Vec2d(q.x, q.y)
    return {0: {0: [0, 0, 1, 3]}, 5: {0: [0, 0, 2, 3]}}


def setup_posts(env):
    """the ball rolls into the lower left goalpost; robot 5 (team -1: the left penalty box is not its own) walks into the upper left one"""
    b = env.ball.shape.body
    b.position = Vec2d(150.0, 300.0)
    b.velocity = Vec2d(-320.0, -30.0)
    env.ball.prevPos = Vec2d(150.0, 300.0)
    r = env.agents[5]
    for foot in (r.leftFoot, r.rightFoot):
        foot.body.position = Vec2d(118.0, 447.0)
        foot.body.angle = math.pi
    r.prevPos = r.getPos()
    return {s: {5: [3, 0, 0, 3]} for s in range(0, 12)}


def setup_unpenalize(env):
    """robots 1 and 7 (one per team) are serving penalties that end 13 and 62 substeps from now: tick's un-penalize branch - flags reset, a free
    penalty spot, both feet moved there (RoboCupEnvironment.py:948-968) - which no other fixture reaches (tools/reference_coverage.py)"""
    for rid, left in ((1, 130), (7, 620)):
        r = env.agents[rid]
        env.penalize(r)
        r.penalTime = left
        r.prevPos = r.getPos()
    return {}


def setup_trip(env):
    """seed 68, elapsed 1050: the die of robot 7's processAction is 0.99982 > 0.999 - a robot that moves falls over by itself
    (RoboCupEnvironment.py:556-560); it is told to walk forward in that step"""
    return {0: {7: [3, 0, 0, 3]}}


class _PartialRig(object):
    """RoboCup with Partial observations: one dispatcher per environment for every random draw of a step - the dice of processAction /
    tick / robotCollision / goalpostCollision (RoboDice) and the noise draws of getAgentVision / addNoise / addNoiseLine
    (gen_golden_robocup_partial.Tape, keyed by agent and by the snapshot's `elapsed`)"""

    def __init__(self, env, key, rc_mod, cut):
        import gen_golden_robocup_partial as grp
        if grp.SRC is None:
            grp.SRC = open(rc_mod.__file__).read().split("\n")   # (the draw sites are told apart by their source line's text, as in grp)
        self.grp, self.env, self.rc_mod, self.cut = grp, env, rc_mod, cut
        self.dice = RoboDice(key, env)
        rig = self

        class Tape(grp.Tape):
            def _frame(self):
                f = sys._getframe(3)
                return f.f_code.co_name, f.f_lineno, f

            def random(self):
                f = sys._getframe(1)
                if f.f_code.co_name in ("processAction", "tick", "robotCollision", "goalpostCollision"):
                    return rig.dice.draw(f)
                return grp.Tape.random(self)

            def randint(self, lo, hi):
                return grp.Tape.randint(self, lo, hi)
        self.tape = tape = Tape(*key)
        self.orig_noise, self.orig_line = cut.addNoise, cut.addNoiseLine
        orig_vision = env.getAgentVision

        def vision(agent):
            tape.begin_agent(env.agents.index(agent), int(env.elapsed))
            return orig_vision(agent)
        env.getAgentVision = vision

    def noise(self, obj, noiseType, interaction, magn, rand, maxDist, misClass=False, angleNoise=False):
        n, tape, cut = len(obj), self.tape, self.cut
        kind = 9 if n == 3 else 0 if n == 4 else 4 if angleNoise else 1 if n == 6 else 3 if misClass else 2
        idx = tape._nth(("noise", kind))
        sites = [(0, 0), (0, 1), (0, 2)] + ([(0, 3)] if (misClass and noiseType == cut.NoiseType.REALISTIC) else []) + [(1, 0)] + ([(1, 1)] if angleNoise else [])
        tape.ctx = (kind, idx, iter(sites))
        try:
            return self.orig_noise(obj, noiseType, interaction, magn, rand, maxDist, misClass, angleNoise)
        finally:
            tape.ctx = None

    def noise_line(self, obj, noiseType, magn, rand, maxDist):
        tape = self.tape
        idx = tape._nth(("noise", 5))
        tape.ctx = (5, idx, iter([(0, 0), (0, 1), (0, 2), (0, 3), (1, 0)]))
        try:
            return self.orig_line(obj, noiseType, magn, rand, maxDist)
        finally:
            tape.ctx = None

    def arm(self):
        m = self.rc_mod
        m.random.random, m.random.randint = self.tape.random, self.tape.randint
        m.addNoise, m.addNoiseLine = self.noise, self.noise_line

    def disarm(self):
        m = self.rc_mod
        m.random.random, m.random.randint = gg._ORIG_RANDOM, gg._ORIG_RANDINT
        m.addNoise, m.addNoiseLine = self.orig_noise, self.orig_line

    def pack(self, obs):
        grp, env = self.grp, self.env
        return np.array([[grp.pack(env, agent, vis) for agent, vis in zip(env.agents, snap)] for snap in obs])


def setup_goal(env):
    """the ball rolls over the left goal line between the posts: a goal (+-25, RoboCupEnvironment.py:640-660), the ball back on the centre spot"""
    b = env.ball.shape.body
    b.position = Vec2d(118.0, 418.0)     # past the goalkeeper (who stands at y = 370 +- 25)
    b.velocity = Vec2d(-290.0, 0.0)
    env.ball.prevPos = Vec2d(118.0, 418.0)
    env.ball.lastKicked = [7, 2]
    return {}


def gen_robocup(out, tag, n, can_fall, steps, seed, forward, setup=None, partial_magn=None, start_elapsed=None):
    import gen_golden_robocup as gr
    import gen_golden_robocup_r2 as g2
    import random as pyrandom
    if partial_magn is not None:
        rc_mod, cut = gg.ref("RoboCupEnvironment"), gg.ref("cutils")

        def make(n_, seed_, can_fall_):
            pyrandom.seed(seed_); np.random.seed(seed_)
            rc_mod.RoboCupEnvironment.canFall = can_fall_
            e = rc_mod.RoboCupEnvironment(n_, render=False, observationType=cut.ObservationType.PARTIAL, noiseType=cut.NoiseType.REALISTIC,
                                          noiseMagnitude=partial_magn)
            return e, rc_mod, cut
        gr_make_env = make
    else:
        gr_make_env = gr.make_env
    env, rc_mod, cut = gr_make_env(n, seed, can_fall)
    env.space.sid_of = robocup_sids(env)
    env.space.reindex()
    # a TWIN of the environment (same scene, same actions, same dice) whose velocities get a relative 1e-15 nudge before every physics
    # substep - what another rounding of the same arithmetic does (nudging once per env step is not enough: a walking robot's velocity is
    # overwritten by the game logic every substep, Robot.py's `step`, and the nudge with it).  RoboCup's contacts (two-point manifolds of
    # nearly parallel feet - the kick-off poses all share their angles -, duplicate end-cap contact points, friction 6.25) amplify that by
    # up to 1e10 within twenty substeps in a few percent of the contact phases (profiles/r05_kat_general_fuzz.txt): where the twin has
    # drifted from the trajectory, the trajectory is not determined to the test's tolerance by ANY implementation, and the fixture says
    # so (`<tag>_conditioning`: the twin's largest deviation over the steps up to each recorded state since the one before).
    twin, _, _ = gr_make_env(n, seed, can_fall)
    twin.space.sid_of = robocup_sids(twin)
    twin.space.reindex()
    forced = {}
    if setup is not None:
        forced = setup(env)
        setup(twin)
    if start_elapsed is not None:   # late in the episode: the trajectory ends with the terminal step (`finished`, info['episode_*'] :516-521)
        env.elapsed = twin.elapsed = int(start_elapsed)
    nrng = np.random.RandomState(seed + 1000)
    twin_space_step = twin.space.step

    def nudged_step(dt):
        for b in twin.space.bodies:
            if not b.k.static:
                b.k.vx *= 1.0 + 1e-15 * nrng.uniform(-1, 1); b.k.vy *= 1.0 + 1e-15 * nrng.uniform(-1, 1); b.k.w *= 1.0 + 1e-15 * nrng.uniform(-1, 1)
        twin_space_step(dt)
    twin.space.step = nudged_step
    key = (42, seed, 1)
    dice = RoboDice(key, env)
    twin_dice = RoboDice(key, twin)
    rig = twin_rig = None
    if partial_magn is not None:
        rig, twin_rig = _PartialRig(env, key, rc_mod, cut), _PartialRig(twin, key, rc_mod, cut)
        dice = rig.dice
    try:
        before = gr.dump(env)
        acts, rews, dones, obss, eps, marks, states, cond, drift = [], [], [], [], [], [], [], [], 0.0
        arng = np.random.RandomState(seed + 7)
        for s in range(steps):
            a = np.stack([arng.randint(0, k, 2 * n) for k in (5, 3, 3, 7)], -1)
            fw = arng.rand(2 * n) < forward          # walk forward (Robot.step direction 2 = action 3), no turn: the teams meet at the ball
            a[fw, 0], a[fw, 1] = 3, 0
            if s % 5 == 4:
                a[:, 0] = 0; a[:, 1] = 0             # kicks need move == turn == 0
            for rid, act in forced.get(s, {}).items():
                a[rid] = act
            if rig is None:
                rc_mod.random.random = dice
            else:
                rig.arm()
            obs, r, done, info = env.step(a.copy())
            if rig is None:
                rc_mod.random.random = twin_dice
            else:
                twin_rig.arm()
            twin.step(a.copy())
            here, t = gr.dump(env), gr.dump(twin)
            drift = max(drift, float(np.max(np.abs(t[0] - here[0]) / np.maximum(1.0, np.abs(here[0])))),
                        float(np.max(np.abs(t[3] - here[3]) / np.maximum(1.0, np.abs(here[3])))))
            acts.append(a.astype(np.float64)); rews.append(np.array(r, float)); dones.append(int(done))
            obss.append(g2.flat_snapshots(obs, 2 * n) if rig is None else rig.pack(obs))
            eps.append(np.concatenate([np.array(env.episodeRewards, float), np.array(env.episodePosRewards, float)]))
            if s % 5 == 4 or s == steps - 1:
                marks.append(s); states.append(here[:4])
                cond.append(drift)
                drift = 0.0
    finally:
        rc_mod.random.random = gg._ORIG_RANDOM
        if rig is not None:
            rig.disarm()
        rc_mod.RoboCupEnvironment.canFall = True      # the class default (RoboCupEnvironment.py:20)
    for k, v in zip(("rf", "ri", "sc", "fl"), before[:4]):
        out["%s_b_%s" % (tag, k)] = v
    for j, k in enumerate(("rf", "ri", "sc", "fl")):
        out["%s_states_%s" % (tag, k)] = np.array([st[j] for st in states])
    out[tag + "_state_steps"] = np.array(marks, np.int64)
    out[tag + "_conditioning"] = np.array(cond)
    out[tag + "_actions"] = np.array(acts, np.float64)
    out[tag + "_rewards"] = np.array(rews)
    out[tag + "_dones"] = np.array(dones, np.int64)
    out[tag + "_obs"] = np.array(obss)
    out[tag + "_episode"] = np.array(eps)
    out[tag + "_meta"] = np.array([n, int(can_fall), key[0], key[1], key[2], 0, 0], np.int64)
    if partial_magn is not None:
        out[tag + "_noise"] = np.array([1, partial_magn], float)
    out[tag + "_goals"] = np.array(env.goals, np.int64)
    log = env.space.world.log
    kinds = {}
    for e in log:
        if e[1] == "begin":
            k = "robot-robot" if e[3] < 20 else "robot-ball" if e[3] == 20 and e[2] < 20 else "ball-post" if e[2] == 20 else "robot-post"
            if e[3] < 20 and e[2] // 2 == e[3] // 2:
                k = "own feet"
            kinds[k] = kinds.get(k, 0) + 1
    out[tag + "_begins"] = np.array([kinds.get(k, 0) for k in ("robot-robot", "robot-ball", "robot-post", "ball-post", "own feet")], np.int64)
    print("%s: %d a side, canFall %d, %d steps: first touches %s, %d separations, %d re-touches, %d dice drawn, fallen %d, penalized %d, goals %s" %
          (tag, n, can_fall, steps, kinds, sum(1 for e in log if e[1] == "separate"), sum(1 for e in log if e[1] == "retouch"), dice.draws,
           sum(int(r.fallen) for r in env.agents), sum(int(r.penalized) for r in env.agents), list(env.goals)))
    print("   conditioning (twin with 1e-15 velocity nudges every substep), largest per window:", " ".join("%d:%.0e" % (m, c) for m, c in zip(marks, cond)))


# ------------------------------------------------------------------ Driving, Partial observations + Realistic noise 3 (BASELINE configs[3])
def setup_peds_beside_obstacles(env):
    """four pedestrians 16 px from an obstacle's centre (just clear of its 10 px half size + their 5 px radius): doesInteractPoly's
    proximity test - squared distance < 400, cutils.py:654-655 - makes them `Nearby`, which doubles their Realistic noise (cutils.py:514-515)"""
    Vec2d = type(env.agents[0].goal)
    for k in range(min(4, len(env.pedestrians), len(env.obstacles))):
        o = env.obstacles[k].getPos()
        env.pedestrians[k].shape.body.position = Vec2d(o.x + (16.0 if k % 2 == 0 else -16.0), o.y + (3.0 * k))


def gen_driving_partial(out, n_players, seed, steps, tag, action_bias, magn=3.0, setup=None):
    """the reference's step() with observationType PARTIAL: getAgentVision of every agent inside the step, its noise draws served by
    source line from the oracle's Philox words exactly as in gen_golden_partial.py (Tape), the pedestrians' draws as in gen_golden.py"""
    import gen_golden_partial as gp
    import random as pyrandom
    de, cut = gg.ref("DrivingEnvironment"), gg.ref("cutils")
    pyrandom.seed(seed)
    np.random.seed(seed)
    env = de.DrivingEnvironment(n_players, render=False, observationType=cut.ObservationType.PARTIAL, noiseType=cut.NoiseType.REALISTIC,
                                noiseMagnitude=magn)
    env.space.sid_of = driving_sids(env)
    if setup is not None:
        setup(env)
    pseed, genv, episode = 42, seed, 1
    ped = gg.PedTape(pseed, genv, episode)
    nearby = [0]

    class Tape(gp.Tape):    # one dispatcher for both kinds of draw sites
        def _site(self):
            f = sys._getframe(3)
            return f.f_code.co_name, f.f_lineno, f

        def random(self):
            return ped.random() if sys._getframe(1).f_code.co_name == "move" else gp.Tape.random(self)

        def randint(self, lo, hi):
            return ped.randint(lo, hi) if sys._getframe(1).f_code.co_name == "move" else gp.Tape.randint(self, lo, hi)
    tape = Tape(pseed, genv, episode)
    orig_rect, orig_lane, orig_move, orig_vision = cut.addNoiseRect, cut.addNoiseLane, env.move, env.getAgentVision

    def rect(obj, noiseType, interaction, magn_, rand, maxDist, misClass=False):
        kind = {9: 0, 8: 1, 5: 2, 7: 3}[len(obj)]
        idx = tape.counters.get(kind, 0)
        tape.counters[kind] = idx + 1
        sites = [(0, 0), (0, 1), (0, 2)] + ([(0, 3)] if (misClass and noiseType == cut.NoiseType.REALISTIC) else []) + [(1, 0)]
        tape.ctx = (kind, idx, iter(sites))
        if interaction == cut.InteractionType.Nearby and obj[0] != cut.SightingType.NoSighting:
            nearby[0] += 1
        try:
            return orig_rect(obj, noiseType, interaction, magn_, rand, maxDist, misClass)
        finally:
            tape.ctx = None

    def lane(obj, noiseType, magn_, rand, maxDist):
        idx = tape.counters.get(4, 0)
        tape.counters[4] = idx + 1
        tape.ctx = (4, idx, iter([(0, 0), (0, 1), (0, 2)]))
        try:
            return orig_lane(obj, noiseType, magn_, rand, maxDist)
        finally:
            tape.ctx = None

    def move_with_tape(p):
        ped.arm(env.pedestrians.index(p), env.elapsed)
        return orig_move(p)

    def vision_with_tape(agent):
        tape.begin_agent(env.agents.index(agent), env.elapsed)
        return orig_vision(agent)
    env.move, env.getAgentVision = move_with_tape, vision_with_tape
    de.addNoiseRect, de.addNoiseLane = rect, lane
    de.random.random, de.random.randint = tape.random, tape.randint
    A = len(env.agents)
    try:
        for k, v in gg.dump_state(env, cut).items():
            out["%s_init_%s" % (tag, k)] = v
        out["%s_key" % tag] = np.array([pseed, genv, episode], np.int64)
        arng = np.random.RandomState(seed + 100)
        acts, rews, dones, obss = [], [], [], []
        for s in range(steps):
            a = np.where(arng.rand(A, 2) < action_bias, [2, 1], arng.randint(0, 3, size=(A, 2)))
            obs, r, d, info = env.step(a)
            rows = np.zeros((A, gp.DIM), np.float32)
            for ai, ((cars, obst, peds), (selfr, lanes), _) in enumerate(obs[0]):
                row = rows[ai]
                row[0:9] = selfr[0]
                off = 9
                for arr, cap, feat in ((cars, 24, 7), (obst, 32, 6), (peds, 40, 2), (lanes, 16, 4)):
                    m = min(len(arr), cap)
                    if m:
                        row[off:off + m * feat] = np.asarray(arr, np.float32).reshape(len(arr), feat)[:m].reshape(-1)
                    off += cap * feat
                row[-4:] = [min(len(cars), 24), min(len(obst), 32), min(len(peds), 40), min(len(lanes), 16)]
            acts.append(a); rews.append(np.array(r, float)); dones.append(int(d)); obss.append(rows)
        st = gg.dump_state(env, cut)
    finally:
        de.addNoiseRect, de.addNoiseLane = orig_rect, orig_lane
        de.random.random, de.random.randint = gg._ORIG_RANDOM, gg._ORIG_RANDINT
    out["%s_actions" % tag] = np.array(acts, np.int64)
    out["%s_rewards" % tag] = np.array(rews)
    out["%s_dones" % tag] = np.array(dones, np.int64)
    out["%s_obs" % tag] = np.array(obss, np.float32)
    out["%s_cfg" % tag] = np.array([n_players, 1, magn], float)
    out["%s_nearby_rows" % tag] = np.array([nearby[0]], np.int64)   # rows whose noise took the `Nearby` multiplier (cutils.py:514-515)
    for name in ("cars_f", "cars_i", "peds_f", "peds_i", "episode_r", "episode_pos_r"):
        out["%s_final_%s" % (tag, name)] = st[name]
    log = env.space.world.log
    print("%s (Partial + Realistic %g): %d players, %d steps: %d first touches, %d cars crashed, %d pedestrians dead; rows per agent-step: cars %.1f obstacles %.1f pedestrians %.1f lanes %.1f" %
          ((tag, magn, n_players, steps, sum(1 for e in log if e[1] == "begin"), sum(int(c.crashed) for c in env.agents), sum(int(p.dead) for p in env.pedestrians)) +
           tuple(np.array(obss)[:, :, -4:].mean((0, 1)))))


def main():
    install()
    out = {}
    gen_driving_partial(out, 10, 71, 100, "a", 0.6)
    gen_driving_partial(out, 10, 72, 100, "b", 0.5)
    gen_driving_partial(out, 6, 73, 120, "c", 0.7)
    gen_driving_partial(out, 6, 74, 12, "d", 0.5, setup=setup_peds_beside_obstacles)   # pedestrians `Nearby` an obstacle: doubled noise
    np.savez_compressed(os.path.join(HERE, "driving_partial_contacts.npz"), **out)
    print("wrote", os.path.join(HERE, "driving_partial_contacts.npz"))
    out = {}
    gen_robocup(out, "a", 5, True, 35, 81, 0.8, partial_magn=3.0)
    gen_robocup(out, "b", 5, False, 20, 82, 0.7, partial_magn=3.0)
    gen_robocup(out, "c", 3, True, 25, 83, 0.5, partial_magn=3.0)
    np.savez_compressed(os.path.join(HERE, "robocup_partial_contacts.npz"), **out)
    print("wrote", os.path.join(HERE, "robocup_partial_contacts.npz"))
    out = {}
    gen_robocup(out, "a", 5, False, 40, 51, 0.6)
    gen_robocup(out, "b", 5, True, 40, 52, 0.6)
    gen_robocup(out, "c", 3, True, 50, 53, 0.5)
    gen_robocup(out, "d", 5, False, 30, 54, 0.8)
    gen_robocup(out, "e", 5, False, 12, 55, 0.0, setup_kick)
    gen_robocup(out, "f", 5, True, 20, 56, 0.2, setup_posts)
    for k, (n, can_fall, steps, seed, fw) in enumerate(((5, True, 40, 61, 0.7), (4, True, 40, 62, 0.5), (5, True, 30, 63, 0.8), (2, True, 60, 64, 0.6))):
        gen_robocup(out, "ghij"[k], n, can_fall, steps, seed, fw)
    gen_robocup(out, "k", 5, True, 15, 65, 0.5, setup_goal)
    gen_robocup(out, "l", 5, True, 10, 66, 0.6, start_elapsed=12000 - 10 * 50)   # the last ten steps of an episode, terminal step included
    gen_robocup(out, "m", 5, True, 6, 67, 0.3, setup_unpenalize)                  # two penalties expire: tick's un-penalize branch
    gen_robocup(out, "n", 5, True, 5, 68, 0.3, setup_trip, start_elapsed=1050)    # a walking robot falls over by itself (die > 0.999)
    assert out["k_goals"].sum() == 1, "case k must score"
    np.savez_compressed(os.path.join(HERE, "robocup_contacts.npz"), **out)
    print("wrote", os.path.join(HERE, "robocup_contacts.npz"))
    out = {}
    gen_driving(out, 10, 3, 150, "a", 0.5)
    gen_driving(out, 10, 21, 150, "b", 0.7)
    gen_driving(out, 6, 8, 200, "c", 0.6)
    gen_driving(out, 10, 34, 120, "d", 0.3)
    gen_driving(out, 2, 13, 250, "e", 0.85)      # BASELINE configs[0]'s player count
    for k, (n, seed, steps, bias) in enumerate(((10, 41, 150, 0.6), (10, 42, 150, 0.4), (8, 43, 150, 0.7), (10, 44, 200, 0.5), (4, 45, 200, 0.8))):
        gen_driving(out, n, seed, steps, "fghij"[k], bias)
    gen_driving(out, 10, 46, 600, "k", 0.5, obs_every=10)     # one WHOLE episode: the terminal step (done, info['episode_*']) included
    gen_driving(out, 3, 91, 16, "l", 1.0, setup=setup_finish)    # every car reaches its goal: allFinished and the team reward (:281-285, :301)
    gen_driving(out, 3, 96, 16, "m", 1.0, setup=setup_finish)    # the same with four touches on the way
    gen_driving(out, 3, 93, 16, "n", 1.0, setup=setup_finish)    # ... and with one car that does not make it: no team reward
    np.savez_compressed(os.path.join(HERE, "driving_contacts.npz"), **out)
    print("wrote", os.path.join(HERE, "driving_contacts.npz"))


if __name__ == "__main__":
    main()
