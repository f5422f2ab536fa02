#!/usr/bin/env python3
"""Generate golden fixtures for the oracle from the reference's OWN Python code (run in the build container only).

The reference (/root/reference, szemenyeim/DynEnv) is pure Python on top of pymunk/gym/pygame/cv2, none of which is
installed.  `import DynEnv` fails with an ordinary ModuleNotFoundError, so this script injects minimal stand-ins for
those four third-party modules into sys.modules — just enough for the reference's *pure-Python game logic* to import
and run:

  * `pymunk.Vec2d`  : 2-D vector with pymunk-5.x semantics (in-place `rotate`, copying `rotated`, `length`,
                      `get_length_sqrd`, `dot`, `cross`, `angle`) — the API the reference itself relies on
                      (Car.py:36,87,103,107; Road.py:28; cutils.py:370,442,562).
  * `pymunk.Body`   : dumb record (position/velocity/angle/angular_velocity/mass/velocity_func) whose
                      `position`/`velocity` getters return copies like pymunk 5; `Body.update_velocity` implements
                      cpBodyUpdateVelocity for zero gravity / unit damping.
  * `pymunk.Space`  : FREE-FLIGHT ONLY stand-in: `step(dt)` integrates positions then calls each body's
                      velocity_func.  It detects nothing, so fixtures made with it are only valid for windows in
                      which no shapes touch; tests assert that the oracle saw no contact in those windows.
  * `gym.spaces`, `pygame`, `cv2`: inert placeholders (never exercised: render=False).

Nothing of the reference's source is copied: the fixtures are inputs + the outputs the reference code computed.
What this pins: cutils.apply_friction, Road.*, Car.accelerate/turn, DrivingEnvironment.processAction/tick/move/
getFullState/step composition.  What it cannot pin: Chipmunk2D's collision detection and impulse solver.

CPython's `random` / NumPy RandomState streams are replaced by the same Philox4x32-10 draws the oracle makes
(`random.randint(5000,30000)` <- word0, `randint(-2,2)` <- word1, `random()` <- word2, `randint(1,2)` <- word3 of the
block keyed (seed, env, episode, PED_MOVE, ped, elapsed)), so pedestrian decisions are comparable draw for draw.

Usage:  python tests/golden/gen_golden.py   (writes tests/golden/*.npz, *.json)
"""
import copy
import importlib
import json
import math
import os
import random as pyrandom
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORIG_RANDINT, _ORIG_RANDOM = pyrandom.randint, pyrandom.random
REF = "/root/reference"


# ------------------------------------------------------------------ stand-ins
class Vec2d(object):
    __slots__ = ("x", "y")

    def __init__(self, x=0.0, y=None):
        if y is None:
            self.x, self.y = x[0], x[1]
        else:
            self.x, self.y = x, y

    def __len__(self): return 2
    def __getitem__(self, i): return (self.x, self.y)[i]
    def __iter__(self): return iter((self.x, self.y))
    def __eq__(self, o): return hasattr(o, "__getitem__") and len(o) == 2 and self.x == o[0] and self.y == o[1]
    def __ne__(self, o): return not self.__eq__(o)
    # Truthiness: pymunk 5's Vec2d (the version the reference's in-place `rotate` / `position.x = ...` imply) defines the Python-2 hook
    # `__nonzero__` only, which Python 3 (setup.py: python_requires >= 3.5) never calls: bool(Vec2d(0, 0)) falls back on __len__ == 2 and
    # is True (in pymunk >= 6 a NamedTuple: True as well).  So cutils.isLineInArea's `if pt1 and pt2` (:813) tests for None only and a
    # point exactly at the origin IS rotated - rounds 2-5 had a `__bool__` here that made the zero vector falsy; the oracle never did.
    # It matters: a penalized robot stands exactly ON its side line, whose intersection with the edge of its field of view is (0, 0).
    def __repr__(self): return "Vec2d(%r, %r)" % (self.x, self.y)
    def __add__(self, o): return Vec2d(self.x + o[0], self.y + o[1])
    __radd__ = __add__
    def __iadd__(self, o): self.x += o[0]; self.y += o[1]; return self
    def __sub__(self, o): return Vec2d(self.x - o[0], self.y - o[1])
    def __rsub__(self, o): return Vec2d(o[0] - self.x, o[1] - self.y)
    def __mul__(self, s): return Vec2d(self.x * s, self.y * s)
    __rmul__ = __mul__
    def __truediv__(self, s): return Vec2d(self.x / s, self.y / s)
    def __neg__(self): return Vec2d(-self.x, -self.y)
    def get_length_sqrd(self): return self.x ** 2 + self.y ** 2
    @property
    def length(self): return math.sqrt(self.x ** 2 + self.y ** 2)
    def get_length(self): return self.length
    def dot(self, o): return float(self.x * o[0] + self.y * o[1])
    def cross(self, o): return self.x * o[1] - self.y * o[0]
    @property
    def angle(self):
        if self.get_length_sqrd() == 0:
            return 0
        return math.atan2(self.y, self.x)
    def rotate(self, a):  # in place (pymunk < 6)
        c, s = math.cos(a), math.sin(a)
        x = self.x * c - self.y * s
        y = self.x * s + self.y * c
        self.x, self.y = x, y
    def rotated(self, a):
        c, s = math.cos(a), math.sin(a)
        return Vec2d(self.x * c - self.y * s, self.x * s + self.y * c)


class Body(object):
    DYNAMIC, KINEMATIC, STATIC = 0, 1, 2

    def __init__(self, mass=0, moment=0, body_type=0):
        self.mass, self.moment, self.body_type = mass, moment, body_type
        self._p, self._v = Vec2d(0.0, 0.0), Vec2d(0.0, 0.0)
        self.angle, self.angular_velocity = 0.0, 0.0
        self.velocity_func = None
        self._f, self._t = Vec2d(0.0, 0.0), 0.0

    position = property(lambda s: Vec2d(s._p.x, s._p.y), lambda s, v: setattr(s, "_p", Vec2d(v[0], v[1])))
    velocity = property(lambda s: Vec2d(s._v.x, s._v.y), lambda s, v: setattr(s, "_v", Vec2d(v[0], v[1])))

    @staticmethod
    def update_velocity(body, gravity, damping, dt):
        m_inv = 1.0 / body.mass
        body._v = Vec2d(body._v.x * damping + (gravity[0] + body._f.x * m_inv) * dt,
                        body._v.y * damping + (gravity[1] + body._f.y * m_inv) * dt)
        body.angular_velocity = body.angular_velocity * damping + body._t * (1.0 / body.moment) * dt
        body._f, body._t = Vec2d(0.0, 0.0), 0.0


class _Shape(object):
    def __init__(self, body, *a, **k):
        self.body = body
        self.args = a
        self.elasticity = 0.0
        self.friction = 0.0
        self.collision_type = 0
        self.color = None


class _Joint(object):  # identity-compared placeholder for PivotJoint / RotaryLimitJoint (never solved here)
    def __init__(self, *a, **k):
        self.args = a
        self.error_bias = 0


class _Handler(object):
    begin = post_solve = separate = None


class Space(object):
    """free-flight stand-in (see module docstring)"""

    def __init__(self):
        self.gravity = (0.0, 0.0)
        self.bodies, self.shapes, self.handlers = [], [], {}

    def add(self, *objs):
        for o in objs:
            (self.bodies if isinstance(o, Body) else self.shapes).append(o)

    def remove(self, *objs):
        for o in objs:
            for lst in (self.bodies, self.shapes):
                if o in lst:
                    lst.remove(o)

    def add_collision_handler(self, a, b):
        return self.handlers.setdefault((a, b), _Handler())

    def step(self, dt):
        for b in self.bodies:
            if b.body_type != Body.DYNAMIC:
                continue
            b._p = Vec2d(b._p.x + (b._v.x + 0.0) * dt, b._p.y + (b._v.y + 0.0) * dt)
            b.angle = b.angle + (b.angular_velocity + 0.0) * dt
        for b in self.bodies:
            if b.body_type != Body.DYNAMIC:
                continue
            if b.velocity_func is not None:
                b.velocity_func(b, self.gravity, 1.0, dt)
            else:
                Body.update_velocity(b, self.gravity, 1.0, dt)


def moment_for_poly(mass, vertices, offset=(0, 0), radius=0):
    s1 = s2 = 0.0
    n = len(vertices)
    for i in range(n):
        v1 = Vec2d(vertices[i][0] + offset[0], vertices[i][1] + offset[1])
        v2 = Vec2d(vertices[(i + 1) % n][0] + offset[0], vertices[(i + 1) % n][1] + offset[1])
        a = v2.cross(v1)
        b = v1.dot(v1) + v1.dot(v2) + v2.dot(v2)
        s1 += a * b
        s2 += a
    return (mass * s1) / (6.0 * s2)


def moment_for_circle(mass, inner_radius, outer_radius, offset=(0, 0)):
    return mass * (0.5 * (inner_radius * inner_radius + outer_radius * outer_radius) + (offset[0] ** 2 + offset[1] ** 2))


def moment_for_segment(mass, a, b, radius):
    off = ((a[0] + b[0]) * 0.5, (a[1] + b[1]) * 0.5)
    length = math.sqrt((b[0] - a[0]) ** 2 + (b[1] - a[1]) ** 2) + 2.0 * radius
    return mass * ((length * length + 4.0 * radius * radius) / 12.0 + (off[0] ** 2 + off[1] ** 2))


def install_standins():
    pm = types.ModuleType("pymunk")
    pm.Vec2d, pm.Body, pm.Space = Vec2d, Body, Space
    pm.Poly = pm.Circle = pm.Segment = _Shape
    pm.moment_for_poly, pm.moment_for_circle, pm.moment_for_segment = moment_for_poly, moment_for_circle, moment_for_segment
    pm.ShapeFilter = lambda **k: None
    pgu = types.ModuleType("pymunk.pygame_util")
    pgu.DrawOptions = lambda *a: None
    pm.pygame_util = pgu
    cons = types.ModuleType("pymunk.constraint")
    cons.PivotJoint = cons.RotaryLimitJoint = _Joint
    pm.constraint = cons
    sys.modules.update({"pymunk": pm, "pymunk.pygame_util": pgu, "pymunk.constraint": cons})

    class _Sp(object):
        def __init__(self, *a, **k):
            self.a, self.k = a, k
            self.spaces = a[0] if a else None
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    for n in ("Tuple", "MultiDiscrete", "Box", "Dict", "MultiBinary", "Space", "Discrete"):
        setattr(spaces, n, type(n, (_Sp,), {}))
    gym.spaces, gym.Space = spaces, spaces.Space
    sys.modules.update({"gym": gym, "gym.spaces": spaces})
    for n in ("pygame", "cv2"):
        sys.modules[n] = types.ModuleType(n)
    # package shell so that `from .cutils import ...` resolves without running DynEnv/__init__.py
    pkg = types.ModuleType("DynEnv")
    pkg.__path__ = [os.path.join(REF, "DynEnv")]
    sys.modules["DynEnv"] = pkg


def ref(mod):
    return importlib.import_module("DynEnv." + mod)


# ------------------------------------------------------------------ Philox (same as include/dynenv_math.h)
def philox(k0, k1, c):
    c = list(c)
    M = 0xFFFFFFFF
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> 32) ^ c[3] ^ k1) & M, p0 & M]
        k0 = (k0 + 0x9E3779B9) & M
        k1 = (k1 + 0xBB67AE85) & M
    return c


def env_rng(seed, genv, episode, purpose, entity, t):
    M = 0xFFFFFFFF
    k1 = ((seed >> 32) & M) ^ ((genv * 0x9E3779B1 + 0x7F4A7C15) & M)
    return philox(seed & M, k1, [episode & M, purpose, entity, t & M])


def randint_from(u, lo, hi):
    return lo + ((u * (hi - lo + 1)) >> 32)


RNG_PED_MOVE = 6


class PedTape(object):
    """Serves the reference's random.* calls inside move() from the Philox block of (ped, elapsed)."""

    def __init__(self, seed, genv, episode):
        self.seed, self.genv, self.episode = seed, genv, episode
        self.block = None

    def arm(self, ped_idx, elapsed):
        self.block = env_rng(self.seed, self.genv, self.episode, RNG_PED_MOVE, ped_idx, int(elapsed))

    def randint(self, lo, hi):
        word = {(5000, 30000): 0, (-2, 2): 1, (1, 2): 3}[(lo, hi)]
        return randint_from(self.block[word], lo, hi)

    def random(self):
        return self.block[2] * 2.0 ** -32


# ------------------------------------------------------------------ helpers to move state across
CAR_F = ("px", "py", "vx", "vy", "angle", "w", "dirx", "diry", "prevx", "prevy", "goalx", "goaly")
CAR_I = ("type", "team", "finished", "crashed", "lane_pos", "fric")
PED_F = ("px", "py", "vx", "vy")
PED_I = ("road", "side", "dead", "moving", "speed", "crossing", "begin_crossing")


def dump_state(env, cut):
    cars_f, cars_i, peds_f, peds_i = [], [], [], []
    for c in env.agents:
        b = c.shape.body
        cars_f.append([b._p.x, b._p.y, b._v.x, b._v.y, b.angle, b.angular_velocity, c.direction.x, c.direction.y,
                       c.prevPos[0], c.prevPos[1], c.goal[0], c.goal[1]])
        fric = 1 if b.velocity_func is cut.friction_car_crashed else 0
        cars_i.append([c.type, int(c.team), int(c.finished), int(c.crashed), int(c.position), fric])
    for p in env.pedestrians:
        b = p.shape.body
        road = [i for i, r in enumerate(env.roads) if r.direction is p.direction][0]
        peds_f.append([b._p.x, b._p.y, b._v.x, b._v.y])
        peds_i.append([road, int(p.side), int(p.dead), int(p.moving), int(p.speed), int(p.crossing), int(p.beginCrossing)])
    obst = [[o.getPos()[0], o.getPos()[1]] for o in env.obstacles]
    return dict(cars_f=np.array(cars_f, float).reshape(len(env.agents), 12),
                cars_i=np.array(cars_i, np.int64).reshape(len(env.agents), 6),
                peds_f=np.array(peds_f, float).reshape(len(env.pedestrians), 4),
                peds_i=np.array(peds_i, np.int64).reshape(len(env.pedestrians), 7),
                obst=np.array(obst, float).reshape(len(env.obstacles), 2),
                scalars=np.array([env.elapsed, int(env.allFinished)], np.int64),
                episode_r=np.array(env.episodeRewards, float), episode_pos_r=np.array(env.episodePosRewards, float))


def flat_obs(env, full_obs, A, dim):
    """reference get_full_obs() nested lists -> the dense padded [A, dim] layout of include/dynenv.h"""
    out = np.zeros((A, dim), np.float32)
    for a, o in enumerate(full_obs):
        (cars, obst, peds), (selfr, lanes), _ = o
        row = out[a]
        row[0:9] = selfr[0]
        row[9:9 + cars.size] = cars.reshape(-1)
        off = 9 + (A - 1) * 7
        row[off:off + obst.size] = obst.reshape(-1)
        off += 80
        row[off:off + peds.size] = peds.reshape(-1)
        off += 40
        row[off:off + 40] = lanes.reshape(-1)
    return out


def make_driving(n_players, seed):
    de = ref("DrivingEnvironment")
    cut = ref("cutils")
    pyrandom.seed(seed)
    np.random.seed(seed)
    env = de.DrivingEnvironment(n_players, render=False, observationType=cut.ObservationType.FULL,
                                noiseType=cut.NoiseType.REALISTIC, noiseMagnitude=0)
    return env, de, cut


def gen_pure(out):
    cut, roadm, carm = ref("cutils"), ref("Road"), ref("Car")
    rng = np.random.RandomState(7)
    # --- apply_friction, all five coefficient sets (cutils.py:78-140)
    fr = []
    sets = [("car", cut.friction_car, 5e-5, 1e-5, 0.0), ("car_crashed", cut.friction_car_crashed, 5e-4, 2e-5, 0.0),
            ("ped_dead", cut.friction_pedestrian_dead, 5e-2, 2e-4, 0.0), ("robot", cut.friction_robot, 1e-3, 1e-2, 0.0),
            ("ball", cut.friction_ball, 2.8e-2, 1e-3, 5e-2)]
    masses = {"car": [1200, 1800, 3500, 5000], "car_crashed": [1200, 5000], "ped_dead": [90], "robot": [4000], "ball": [10]}
    for name, fn, mu, mur, spin in sets:
        for m in masses[name]:
            for k in range(40):
                scale = [0.01, 0.3, 5.0, 80.0][k % 4]
                vx, vy, w = (rng.rand(3) - 0.5) * 2 * scale
                if k % 7 == 0: vx = 0.0
                if k % 11 == 0: w = 0.0
                b = Body(m, 1.0)
                b.velocity = Vec2d(vx, vy)
                b.angular_velocity = w
                fn(b, (0.0, 0.0), 1.0, 0.01)
                fr.append(dict(m=m, mu=mu, mur=mur, spin=spin, vin=[vx, vy, w], vout=[b._v.x, b._v.y, b.angular_velocity]))
    out["friction"] = fr
    # --- Road geometry + classification (Road.py)
    roads = [roadm.Road(2, 35, [Vec2d(875, 0), Vec2d(875, 1000)]), roadm.Road(1, 35, [Vec2d(0, 500), Vec2d(1750, 500)])]
    geo = []
    for r in roads:
        geo.append(dict(dir=[r.direction.x, r.direction.y], normal=[r.normal.x, r.normal.y], length=r.length,
                        dirAngle=r.direction.angle,
                        lanes=[[l[0].x, l[0].y, l[1].x, l[1].y] for l in r.Lanes],
                        walk=[[w[0].x, w[0].y, w[1].x, w[1].y] for w in r.Walkways]))
    out["road_geometry"] = geo
    pts = []
    for ri, r in enumerate(roads):
        for k in range(400):
            x, y = rng.rand() * 1900 - 100, rng.rand() * 1200 - 100
            if k % 3 == 0:  # concentrate around the road band
                if ri == 0: x = 875 + (rng.rand() - 0.5) * 200
                else: y = 500 + (rng.rand() - 0.5) * 120
            ang = (rng.rand() - 0.5) * 8
            pts.append(dict(road=ri, x=x, y=y, angle=ang, pos=int(r.isPointOnRoad(Vec2d(x, y), ang))))
    out["is_point_on_road"] = pts
    spots = []
    for ri, r in enumerate(roads):
        for lane in range(2 * r.nLanes):
            for spot in range(5):
                p, a = r.getSpot(lane, spot)
                spots.append(dict(road=ri, lane=lane, spot=spot, pos=[p.x, p.y], angle=a))
    out["get_spot"] = spots
    ws = []
    for ri, r in enumerate(roads):
        for side in (0, 1):
            for _ in range(10):
                l, w = rng.rand(), rng.rand() / 2 + 0.25
                p = r.getWalkSpot(side, l, w)
                ws.append(dict(road=ri, side=side, length=l, width=w, pos=[p.x, p.y]))
    out["get_walk_spot"] = ws
    # --- moments
    out["moments"] = dict(
        box=[dict(m=m, h=h, w=w, I=moment_for_poly(m, [(h, w), (-h, w), (-h, -w), (h, -w)]))
             for m, w, h in zip(carm.Car.masses, carm.Car.widths, carm.Car.lengths)])


def gen_unit_steps(out_npz):
    """processAction / tick / move on reference objects with injected states."""
    env, de, cut = make_driving(10, 11)
    rng = np.random.RandomState(3)
    # ---- processAction (Car.accelerate all branches + turn): for each car state x 9 actions
    recs_in, recs_act, recs_out = [], [], []
    for trial in range(120):
        car = env.agents[trial % 10]
        b = car.shape.body
        ang = (rng.rand() - 0.5) * 7
        b.angle = ang
        car.direction = Vec2d(1, 0)
        car.direction.rotate(ang + (rng.rand() - 0.5) * 0.2 * (trial % 2))
        speed = [0.0, 0.5, 8.0, 40.0][trial % 4] * (1 if trial % 8 < 4 else -1)
        v = Vec2d(speed, (rng.rand() - 0.5) * (trial % 3))
        v.rotate(ang)
        if trial % 4 == 0:
            v = Vec2d(0.0, 0.0)
        b.velocity = v
        car.finished = (trial % 13 == 5)
        st_in = [b._p.x, b._p.y, b._v.x, b._v.y, b.angle, b.angular_velocity, car.direction.x, car.direction.y, car.type, int(car.finished)]
        act = np.array([trial % 3, (trial // 3) % 3])
        env.processAction(act, car)
        st_out = [b._v.x, b._v.y, b.angle, car.direction.x, car.direction.y]
        recs_in.append(st_in); recs_act.append(act); recs_out.append(st_out)
    out_npz["pa_in"] = np.array(recs_in, float)
    out_npz["pa_act"] = np.array(recs_act, np.int64)
    out_npz["pa_out"] = np.array(recs_out, float)

    # ---- tick: random placements incl. goal / off-road / opposing lane / out-of-field
    env, de, cut = make_driving(10, 12)
    t_in, t_out = [], []
    for trial in range(400):
        idx = trial % 10
        car = env.agents[idx]
        b = car.shape.body
        mode = trial % 8
        if mode == 0:   # on vertical road
            p = Vec2d(875 + (rng.rand() - 0.5) * 150, rng.rand() * 1000)
        elif mode == 1:  # on horizontal road
            p = Vec2d(rng.rand() * 1750, 500 + (rng.rand() - 0.5) * 80)
        elif mode == 2:  # near a goal, beyond the road end
            g = [(875, 0), (875, 1000), (0, 500), (1750, 500)][trial % 4]
            d = [(0, -1), (0, 1), (-1, 0), (1, 0)][trial % 4]
            k = 10 + rng.rand() * 60
            p = Vec2d(g[0] + d[0] * k + (rng.rand() - 0.5) * 40 * abs(d[1]), g[1] + d[1] * k + (rng.rand() - 0.5) * 40 * abs(d[0]))
            car.goal = Vec2d(*g) if trial % 3 else car.goal
        elif mode == 3:  # way out
            p = Vec2d([-80, 1790, 875, 875][trial % 4] + rng.rand(), [500, 500, -70, 1080][trial % 4] + rng.rand())
        else:
            p = Vec2d(rng.rand() * 1900 - 100, rng.rand() * 1200 - 100)
        b.position = p
        b.angle = (rng.rand() - 0.5) * 7
        b.velocity = Vec2d((rng.rand() - 0.5) * 60, (rng.rand() - 0.5) * 60)
        car.prevPos = p + Vec2d((rng.rand() - 0.5) * 2, (rng.rand() - 0.5) * 2)
        car.finished = (trial % 9 == 4)
        car.crashed = car.finished and (trial % 2 == 0)
        b.velocity_func = cut.friction_car
        env.elapsed = int(rng.randint(0, 5990))
        env.carRewards = np.array([0.0] * 10)
        env.carPosRewards = np.array([0.0] * 10)
        rec_in = [b._p.x, b._p.y, b._v.x, b._v.y, b.angle, car.prevPos[0], car.prevPos[1], car.goal[0], car.goal[1],
                  int(car.finished), int(car.crashed), env.elapsed, idx]
        env.tick(car)
        rec_out = [env.carRewards[idx], env.carPosRewards[idx], int(car.position), int(car.finished), int(car.crashed),
                   b._v.x, b._v.y, car.prevPos[0], car.prevPos[1], 1 if b.velocity_func is cut.friction_car_crashed else 0]
        t_in.append(rec_in); t_out.append(rec_out)
    out_npz["tick_in"] = np.array(t_in, float)
    out_npz["tick_out"] = np.array(t_out, float)

    # ---- move: pedestrian FSM with the Philox tape
    env, de, cut = make_driving(10, 13)
    seed, genv, episode = 42, 5, 1
    tape = PedTape(seed, genv, episode)
    de.random.randint = tape.randint   # DrivingEnvironment does `from .cutils import *` -> module-level `random`
    de.random.random = tape.random
    m_in, m_out = [], []
    npeds = len(env.pedestrians)
    for trial in range(600):
        k = trial % npeds
        ped = env.pedestrians[k]
        b = ped.shape.body
        mode = trial % 6
        r = env.roads[[i for i, rd in enumerate(env.roads) if rd.direction is ped.direction][0]]
        if mode == 0:    # on own road
            p = r.points[0] + r.direction * (rng.rand() * r.length) + r.normal * ((rng.rand() - 0.5) * 2 * r.nLanes * 35)
        elif mode == 1:  # walkway
            p = r.getWalkSpot(int(rng.randint(0, 2)), rng.rand(), rng.rand() / 2 + 0.25)
        elif mode == 2:  # outside field
            p = Vec2d([-3, 1703, 500, 900][trial % 4] + rng.rand(), [300, 400, -2, 1001][trial % 4] + rng.rand())
        else:
            p = Vec2d(rng.rand() * 1700, rng.rand() * 1000)
        b.position = p
        b.velocity = Vec2d((rng.rand() - 0.5) * 10, (rng.rand() - 0.5) * 10)
        ped.moving = [0, 0, 5, 10, 20, 12000][trial % 6] if trial % 5 else 0
        ped.crossing = bool((trial // 2) % 2)
        ped.beginCrossing = bool((trial // 4) % 2)
        ped.side = int(trial % 2)
        ped.dead = (trial % 31 == 7)
        env.elapsed = int(rng.randint(0, 5990))
        road_idx = [i for i, rd in enumerate(env.roads) if rd.direction is ped.direction][0]
        rec_in = [b._p.x, b._p.y, b._v.x, b._v.y, road_idx, ped.side, int(ped.dead), int(ped.moving), ped.speed,
                  int(ped.crossing), int(ped.beginCrossing), env.elapsed, k]
        tape.arm(k, env.elapsed)
        env.move(ped)
        rec_out = [b._v.x, b._v.y, int(ped.side), int(ped.moving), int(ped.crossing), int(ped.beginCrossing)]
        m_in.append(rec_in); m_out.append(rec_out)
    out_npz["move_in"] = np.array(m_in, float)
    out_npz["move_out"] = np.array(m_out, float)
    out_npz["move_key"] = np.array([seed, genv, episode], np.int64)
    de.random.randint = _ORIG_RANDINT
    de.random.random = _ORIG_RANDOM


def gen_freeflight(out_npz, n_players, seed, steps, tag):
    """Whole reference step() composition on the free-flight Space stand-in."""
    env, de, cut = make_driving(n_players, seed)
    pseed, genv, episode = 42, seed, 1
    tape = PedTape(pseed, genv, episode)
    orig_move = env.move

    def move_with_tape(ped):
        tape.arm(env.pedestrians.index(ped), env.elapsed)
        return orig_move(ped)
    env.move = move_with_tape
    de.random.randint = tape.randint
    de.random.random = tape.random
    A = len(env.agents)
    dim = 9 + (A - 1) * 7 + 80 + 40 + 40
    st0 = dump_state(env, cut)
    for k, v in st0.items():
        out_npz["%s_init_%s" % (tag, k)] = v
    out_npz["%s_key" % tag] = np.array([pseed, genv, episode], np.int64)
    out_npz["%s_obs0" % tag] = flat_obs(env, env.get_full_obs(), A, dim)
    arng = np.random.RandomState(seed + 100)
    acts, rews, dones, obss = [], [], [], []
    for s in range(steps):
        a = arng.randint(0, 3, size=(A, 2))
        obs, r, d, info = env.step(a)
        acts.append(a); rews.append(np.array(r, float)); dones.append(int(d)); obss.append(flat_obs(env, obs[0], A, dim))
    out_npz["%s_actions" % tag] = np.array(acts, np.int64)
    out_npz["%s_rewards" % tag] = np.array(rews)
    out_npz["%s_dones" % tag] = np.array(dones, np.int64)
    out_npz["%s_obs" % tag] = np.array(obss, np.float32)
    st1 = dump_state(env, cut)
    for k, v in st1.items():
        out_npz["%s_final_%s" % (tag, k)] = v
    de.random.randint = _ORIG_RANDINT
    de.random.random = _ORIG_RANDOM


def main():
    install_standins()
    pure = {}
    gen_pure(pure)
    with open(os.path.join(HERE, "driving_pure.json"), "w") as f:
        json.dump(pure, f)
    unit = {}
    gen_unit_steps(unit)
    np.savez_compressed(os.path.join(HERE, "driving_unit.npz"), **unit)
    ff = {}
    gen_freeflight(ff, 10, 3, 25, "a")
    gen_freeflight(ff, 2, 5, 40, "b")
    gen_freeflight(ff, 10, 8, 25, "c")
    np.savez_compressed(os.path.join(HERE, "driving_freeflight.npz"), **ff)
    print("wrote goldens to", HERE)


if __name__ == "__main__":
    main()
