"""A first-principles float64 restatement of ONE narrow case of Chipmunk's step, used as the known answer of the round-4
KATs (tests/test_oracle_kats_r4.py on the oracle, tests/test_gpu_kats.py through the C ABI): axis-aligned boxes in a row
along x (Driving cars that are already crashed, and static obstacle boxes), touching face to face.

It is written from Chipmunk2D 7's published algorithm, NOT from oracle/cp_lite.c, and it is deliberately narrower than the
oracle (no rotation in the geometry, no friction, no broadphase), so that what it predicts can be followed by hand.  It
evaluates every a * b + c with two roundings (plain Python floats); the oracle and the kernels fuse the solver's multiply-adds
(include/dynenv_math.h): the two agree to ~1e-14 relative in these scenes, far inside the KATs' tolerances:

* cpSpaceStep order: position update (v + v_bias) -> collide (contact points, impulse carry-over by contact hash) ->
  cpArbiterPreStep (nMass, bias, bounce from the velocities BEFORE the velocity function) -> velocity function ->
  cpArbiterApplyCachedImpulse for arbiters that are not in their first step -> 10 x cpArbiterApplyImpulse in arbiter order.
* ContactPoints for two boxes whose faces x = const overlap over [lo, hi] in y: two contacts, at y = lo FIRST and y = hi second
  (shape a is the lower-index body, n = (+1, 0) from a to b; a's support edge runs up its right face, b's runs down its left
  face; the first contact pushed is lerp(e1.a, e1.b, clamp01(..e2.b..)), i.e. the LOWER end of the overlap), p1 on a's face,
  p2 on b's face, r1 = p1 - a.p, r2 = p2 - b.p, dist = (p2 - p1) . n <= 0.
* arbiters in canonical pair order (i, j), i < j; e = e_a e_b, u = 0 (Driving shapes have no friction).

Bodies: dict(m, I, hx, hy, x, y, vx, vy, w, fr, rfr) - m = inf for a static obstacle; fr / rfr = the velocity function's
friction / rotFriction (cutils.apply_friction)."""
import math

import numpy as np

DT = 0.01
SLOP = 0.1
# cpSpaceInit: collisionBias = cpfpow(1.0f - 0.1f, 60.0f) - a FLOAT 0.9 raised in double; cpSpaceStep: biasCoef = 1 - pow(collisionBias, dt)
BIAS_COEF = 1.0 - (float(np.float32(1.0) - np.float32(0.1)) ** 60.0) ** DT


def box_inertia(m, hx, hy):
    return m * ((2 * hx) ** 2 + (2 * hy) ** 2) / 12.0


def apply_friction(b):
    """cutils.apply_friction (cutils.py:102-140), spin = 0"""
    if math.isinf(b["m"]):
        return
    factor, rot = b["fr"] * b["m"], b["rfr"] * b["m"]
    x, y, th = b["vx"], b["vy"], b["w"]
    length = 1.0 / (abs(x) + abs(y) + 1e-5)
    a0, a1 = x * factor * length, y * factor * length
    x = 0.0 if abs(x) < factor else x - a0
    y = 0.0 if abs(y) < factor else y - a1
    th = 0.0 if abs(th) < rot else th - (rot if th > 0 else -rot)
    b["vx"], b["vy"], b["w"] = x, y, th


def _minv(b):
    return 0.0 if math.isinf(b["m"]) else 1.0 / b["m"]


def _iinv(b):
    return 0.0 if math.isinf(b["m"]) else 1.0 / b["I"]


class Chain:
    def __init__(self, bodies, e=0.05 * 0.05):
        self.b = [dict(x) for x in bodies]
        for x in self.b:
            x.setdefault("vbx", 0.0); x.setdefault("vby", 0.0); x.setdefault("wb", 0.0); x.setdefault("ang", 0.0)
        self.e = e
        self.cache = {}      # (i, j) -> {"lo": (jnAcc), "hi": (jnAcc)} keyed by contact hash (which face ends made the contact)
        self.active_log = []

    def contacts(self, i, j):
        a, b = self.b[i], self.b[j]
        fa, fb = a["x"] + a["hx"], b["x"] - b["hx"]          # a's right face, b's left face
        dist = fb - fa
        if dist > 0.0:
            return []
        lo, hi = max(a["y"] - a["hy"], b["y"] - b["hy"]), min(a["y"] + a["hy"], b["y"] + b["hy"])
        if hi < lo:
            return []
        out = []
        for name, y in (("lo", lo), ("hi", hi)):              # ContactPoints: the lower end of the overlap is pushed first
            out.append(dict(hash=name, r1=(fa - a["x"], y - a["y"]), r2=(fb - b["x"], y - b["y"]), dist=dist))
        return out

    def substep(self):
        B = self.b
        for x in B:                                            # cpBodyUpdatePosition
            if math.isinf(x["m"]):
                continue
            x["x"] += (x["vx"] + x["vbx"]) * DT; x["y"] += (x["vy"] + x["vby"]) * DT; x["ang"] += (x["w"] + x["wb"]) * DT
            x["vbx"] = x["vby"] = x["wb"] = 0.0
        arbs = []
        for i in range(len(B)):
            for j in range(i + 1, len(B)):
                cons = self.contacts(i, j)
                if not cons:
                    self.cache.pop((i, j), None)               # (these scenes never re-touch within collision_persistence)
                    continue
                old = self.cache.get((i, j))
                first = old is None
                for c in cons:                                  # cpArbiterUpdate: impulses carried over by contact hash
                    c["jn"] = 0.0 if first else old.get(c["hash"], 0.0)
                arbs.append((i, j, cons, first))
        for i, j, cons, first in arbs:                         # cpArbiterPreStep
            a, b = B[i], B[j]
            for c in cons:
                r1, r2 = c["r1"], c["r2"]
                r1cn, r2cn = -r1[1], -r2[1]                    # r x n, n = (1, 0)
                c["nMass"] = 1.0 / (_minv(a) + _minv(b) + _iinv(a) * r1cn * r1cn + _iinv(b) * r2cn * r2cn)
                c["bias"] = -BIAS_COEF * min(0.0, c["dist"] + SLOP) / DT
                c["jb"] = 0.0
                vrn = (b["vx"] - b["w"] * r2[1]) - (a["vx"] - a["w"] * r1[1])
                c["bounce"] = vrn * self.e
        for x in B:                                            # velocity function
            apply_friction(x)

        def push(a, b, r1, r2, j, bias=False):
            kv, kw = ("vbx", "wb") if bias else ("vx", "w")
            if not math.isinf(a["m"]):
                a[kv] -= j / a["m"]; a[kw] += (r1[0] * 0.0 - r1[1] * (-j)) / a["I"]   # a gets -j n at r1: w += cross(r1, -j n) / I
            if not math.isinf(b["m"]):
                b[kv] += j / b["m"]; b[kw] += (r2[0] * 0.0 - r2[1] * j) / b["I"]
        for i, j, cons, first in arbs:                         # cpArbiterApplyCachedImpulse
            if first:
                continue
            for c in cons:
                push(B[i], B[j], c["r1"], c["r2"], c["jn"])
        for _ in range(10):                                    # cpArbiterApplyImpulse, arbiters in canonical order
            for i, j, cons, first in arbs:
                a, b = B[i], B[j]
                for c in cons:
                    r1, r2 = c["r1"], c["r2"]
                    vbn = (b["vbx"] - b["wb"] * r2[1]) - (a["vbx"] - a["wb"] * r1[1])
                    vrn = (b["vx"] - b["w"] * r2[1]) - (a["vx"] - a["w"] * r1[1])
                    jbn = (c["bias"] - vbn) * c["nMass"]
                    old = c["jb"]; c["jb"] = max(old + jbn, 0.0)
                    push(a, b, r1, r2, c["jb"] - old, bias=True)
                    jn = -(c["bounce"] + vrn) * c["nMass"]
                    old = c["jn"]; c["jn"] = max(old + jn, 0.0)
                    push(a, b, r1, r2, c["jn"] - old)
        for i, j, cons, first in arbs:
            self.cache[(i, j)] = {c["hash"]: c["jn"] for c in cons}
        self.active_log.append([(i, j, [c["jn"] for c in cons]) for i, j, cons, first in arbs])
        return len(arbs)


CAR_M = [1200.0, 1800.0, 3500.0, 5000.0]
CAR_HX = [10.0, 15.0, 20.0, 25.0]
CAR_HY = [5.0, 6.0, 7.0, 8.0]


def crashed_car(t, x, y, vx=0.0):
    """a crashed (= finished) Driving car of type t, angle 0: friction_car_crashed (5e-4, 2e-5), cutils.py:86-87"""
    return dict(m=CAR_M[t], I=box_inertia(CAR_M[t], CAR_HX[t], CAR_HY[t]), hx=CAR_HX[t], hy=CAR_HY[t], x=x, y=y, vx=vx, vy=0.0,
                w=0.0, fr=5e-4, rfr=2e-5)


def obstacle(x, y):
    """Obstacle.py: a static 20 x 20 box, elasticity 0.05"""
    return dict(m=math.inf, I=math.inf, hx=10.0, hy=10.0, x=x, y=y, vx=0.0, vy=0.0, w=0.0, fr=0.0, rfr=0.0)


def free_travel(t, v0, substeps):
    """distance a crashed car of type t covers in `substeps` substeps of free flight from speed v0 (position first, then friction)"""
    c = crashed_car(t, 0.0, 0.0, v0)
    for _ in range(substeps):
        c["x"] += c["vx"] * DT
        apply_friction(c)
    return c["x"], c["vx"]
