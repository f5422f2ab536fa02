"""Differential fuzz of the narrowphase: oracle/cp_lite.c (SAT / closest-point formulas, through `oracle_cp_collide_batch`) against
tests/kat_general.py (Chipmunk's own route: GJK / EPA -> ClosestPoints -> support edges -> ContactPoints).  Shared by
tests/test_kat_general.py (a bounded sample in the CPU suite) and tools/kat_general_fuzz.py (the full 10^5 per shape pair, report
under profiles/).  Test infrastructure only.

A sample is a pair of shapes with the sizes DynEnv uses: car boxes (Car.py:9-11), obstacle 20 x 20 and building
800 x 450 boxes at angle 0, pedestrian circle r = 5, robot feet capsules (-10, +-10) -> (10, +-10) r = 7.5, ball / goalpost circles r = 10.
Placement: a random relative pose, then - for most samples - shape b is moved along the separation normal so that the pair ends up
between 2 px apart and 3 px deep (where real contacts live: collision_slop is 0.1, a car moves 1 px per substep), with a share of
special poses: axis-aligned, nearly parallel faces (1e-3 .. 1e-9 rad), corner to corner, parallel feet, the two feet of one robot."""
import ctypes as C
import math

import numpy as np

import kat_general as kg
import oracle_lib as ol

PAIRS = ("box_box", "box_circle", "capsule_capsule", "capsule_circle", "circle_circle")
TOL = 1e-9


def _mk(desc, sid):
    desc = [float(x) for x in desc]
    kind = int(desc[0])
    b = kg.Body(1.0, 1.0, desc[1], desc[2], desc[3])
    if kind == 0:
        return kg.Shape(kg.CIRCLE, b, sid, r=desc[4])
    if kind == 1:
        return kg.Shape(kg.SEGMENT, b, sid, r=desc[8], la=(desc[4], desc[5]), lb=(desc[6], desc[7]))
    return kg.Shape(kg.POLY, b, sid, hx=desc[4], hy=desc[5])


def _angle(rng, special):
    """a body angle: uniform, or (special) a multiple of 90 degrees, exact or off by 1e-3 .. 1e-9 rad"""
    if not special:
        return rng.uniform(-math.pi, math.pi)
    base = rng.integers(-2, 3) * (math.pi / 2.0)
    return base if rng.random() < 0.4 else base + rng.choice((-1.0, 1.0)) * 10.0 ** rng.uniform(-9.0, -3.0)


def _car(rng, ang):
    t = rng.integers(0, 4)
    return [2, 0.0, 0.0, ang, kg.CAR_HX[t], kg.CAR_HY[t], 0, 0, 0]


def _foot(rng, ang, side=None):
    yo = 10.0 if (rng.random() < 0.5 if side is None else side == 0) else -10.0
    return [1, 0.0, 0.0, ang, -10.0, yo, 10.0, yo, 7.5]


def generate(pair, n, seed):
    """-> descA, descB: float64 [n, 9]"""
    rng = np.random.default_rng(seed)
    A, B = np.zeros((n, 9)), np.zeros((n, 9))
    for i in range(n):
        special = rng.random() < 0.3
        if pair == "box_box":
            a = _car(rng, _angle(rng, special))
            k = rng.random()
            b = _car(rng, _angle(rng, special)) if k < 0.6 else ([2, 0, 0, 0.0, 10.0, 10.0, 0, 0, 0] if k < 0.85 else [2, 0, 0, 0.0, 400.0, 225.0, 0, 0, 0])
        elif pair == "box_circle":
            a = _car(rng, _angle(rng, special))
            b = [0, 0, 0, 0.0, 5.0, 0, 0, 0, 0]
        elif pair == "capsule_capsule":
            ang = _angle(rng, special)
            a = _foot(rng, ang)
            k = rng.random()
            if k < 0.25:      # parallel or nearly parallel feet of two robots
                b = _foot(rng, ang + (0.0 if rng.random() < 0.5 else 10.0 ** rng.uniform(-9.0, -2.0)) + (math.pi if rng.random() < 0.5 else 0.0))
            elif k < 0.35:    # the two feet of ONE robot: same position, slightly different angles (what the rotary limit allows)
                a = _foot(rng, ang, 0)
                b = _foot(rng, ang + rng.normal(0.0, 0.3), 1)
            else:
                b = _foot(rng, _angle(rng, special))
        elif pair == "capsule_circle":
            a = [0, 0, 0, 0.0, 10.0, 0, 0, 0, 0]
            b = _foot(rng, _angle(rng, special))
        else:
            a = [0, 0, 0, 0.0, 10.0, 0, 0, 0, 0]
            b = [0, 0, 0, 0.0, 10.0 if rng.random() < 0.7 else 5.0, 0, 0, 0, 0]
        a[1], a[2] = rng.uniform(100.0, 1600.0), rng.uniform(100.0, 900.0)
        same_robot = pair == "capsule_capsule" and a[5] == 10.0 and b[5] == -10.0 and 0.25 <= k < 0.35
        if same_robot:
            b[1], b[2] = a[1] + rng.normal(0.0, 0.05), a[2] + rng.normal(0.0, 0.05)
        else:
            reach = 1.2 * (math.hypot(a[4], a[5]) + math.hypot(b[4], b[5]) + 10.0) if a[0] == 2 or b[0] == 2 else 45.0
            if b[0] == 2 and b[4] == 400.0:
                reach = 480.0
            if special and rng.random() < 0.3 and a[0] == 2 and b[0] == 2:   # corner to corner along the diagonal
                sx, sy = rng.choice((-1.0, 1.0)), rng.choice((-1.0, 1.0))
                b[1], b[2] = a[1] + sx * (a[4] + b[4]), a[2] + sy * (a[5] + b[5])
            else:
                th, rr = rng.uniform(0.0, 2.0 * math.pi), reach * math.sqrt(rng.random())
                b[1], b[2] = a[1] + rr * math.cos(th), a[2] + rr * math.sin(th)
            if rng.random() < 0.75:   # slide b along the separation normal to a gap of -3 .. +2 px
                sa, sb = _mk(a, 0), _mk(b, 1)
                if sa.kind > sb.kind:
                    sa, sb = sb, sa
                if sa.kind == kg.CIRCLE and sb.kind != kg.POLY:
                    if sb.kind == kg.CIRCLE:
                        dx, dy = sb.tc[0] - sa.tc[0], sb.tc[1] - sa.tc[1]
                    else:
                        s1, s2, _, _, _ = kg.collide(sa, sb)
                        ex, ey = sb.tb[0] - sb.ta[0], sb.tb[1] - sb.ta[1]
                        t = max(0.0, min(1.0, (ex * (sa.tc[0] - sb.ta[0]) + ey * (sa.tc[1] - sb.ta[1])) / (ex * ex + ey * ey)))
                        dx, dy = sb.ta[0] + ex * t - sa.tc[0], sb.ta[1] + ey * t - sa.tc[1]
                    d = math.hypot(dx, dy)
                    nrm = (dx / d, dy / d) if d else (1.0, 0.0)
                    gap = d - sa.r - sb.r
                else:
                    _, _, nrm, d, _ = kg.gjk(sa, sb)
                    gap = d - sa.r - sb.r
                move = rng.uniform(-3.0, 2.0) - gap                 # along n (from the first shape in type order to the second)
                sign = 1.0 if (_mk(a, 0).kind <= _mk(b, 1).kind) else -1.0   # is `b` the second shape?
                b[1] += sign * move * nrm[0]
                b[2] += sign * move * nrm[1]
        A[i], B[i] = a, b
    return A, B


def oracle_collide(A, B):
    n = len(A)
    out = np.zeros((n, 15))
    A, B = np.ascontiguousarray(A), np.ascontiguousarray(B)
    ol.lib().oracle_cp_collide_batch(C.c_int(n), A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def _oracle_hash(cid):
    """kat_general's ((shape, vertex), (shape, vertex)) in cp_lite's packing: 1 + ((slot_a * 4 + va) << 8 | (slot_b * 4 + vb))"""
    if cid == 0:
        return 0
    (sa, va), (sb, vb) = cid
    return 1 + (((sa * 4 + va) << 8) | (sb * 4 + vb))


def compare(pair, A, B, out, warm=False):
    """-> dict(n, colliding, agree, classes {name: count}, examples {name: [index...]}, worst = largest deviation among agreeing samples)"""
    res = dict(n=len(A), colliding=0, agree=0, classes={}, examples={}, worst=0.0, gjk_iters=0, epa_iters=0)
    stats = {}

    def note(cls, i):
        res["classes"][cls] = res["classes"].get(cls, 0) + 1
        res["examples"].setdefault(cls, [])
        if len(res["examples"][cls]) < 5:
            res["examples"][cls].append(int(i))

    for i in range(len(A)):
        sa, sb = _mk(A[i], 0), _mk(B[i], 1)
        bb = sa.bb[0] <= sb.bb[2] and sb.bb[0] <= sa.bb[2] and sa.bb[1] <= sb.bb[3] and sb.bb[1] <= sa.bb[3]
        o = out[i]
        if bool(o[0]) != bb:
            note("bounding boxes disagree", i)
            continue
        if not bb:
            res["agree"] += 1
            continue
        cid = 0
        if warm:      # Chipmunk warm-starts GJK from the features of the pair's previous call: here, those of a nudged pose
            nb = B[i].copy()
            nb[1] += 0.37
            nb[2] -= 0.21
            nb[3] += 0.01
            s1w, s2w = (sa, _mk(nb, 1)) if sa.kind <= sb.kind else (_mk(nb, 1), sa)
            if s1w.kind != kg.CIRCLE or s2w.kind == kg.POLY:
                cid = kg.gjk(s1w, s2w)[4]
        s1, s2, n, cons, _ = kg.collide(sa, sb, cid, stats)
        swapped = s1 is not sa
        cnt = int(o[1])
        if cnt or cons:
            res["colliding"] += 1
        if cnt != len(cons):
            # a contact exactly at a threshold (gap 0 within rounding)?
            if s1.kind == kg.CIRCLE and s2.kind != kg.POLY:
                gap = None
            else:
                gap = kg.gjk(s1, s2)[3] - s1.r - s2.r
            dists = [(p2[0] - p1[0]) * n[0] + (p2[1] - p1[1]) * n[1] for p1, p2, _ in cons]
            od = [(o[5 + 5 * k + 2] - o[5 + 5 * k]) * o[3] + (o[5 + 5 * k + 3] - o[5 + 5 * k + 1]) * o[4] for k in range(cnt)]
            if (gap is not None and abs(gap) < TOL) or any(abs(d) < TOL for d in dists + od):
                note("contact at a threshold (|gap| < 1e-9): count differs", i)
            elif s1.kind == kg.SEGMENT and s2.kind == kg.SEGMENT and gap is not None and gap + s1.r + s2.r < 0.0:
                note("capsule cores cross", i)
            else:
                note("COUNT", i)
            continue
        if cnt == 0:
            res["agree"] += 1
            continue
        if bool(o[2]) != swapped:
            note("ORDER", i)
            continue
        dev = max(abs(o[3] - n[0]), abs(o[4] - n[1]))
        for k, (p1, p2, cid_c) in enumerate(cons):
            q = o[5 + 5 * k: 10 + 5 * k]
            dev = max(dev, abs(q[0] - p1[0]), abs(q[1] - p1[1]), abs(q[2] - p2[0]), abs(q[3] - p2[1]))
        ids_ok = all(int(o[5 + 5 * k + 4]) == _oracle_hash(c[2]) for k, c in enumerate(cons))
        if dev <= TOL and ids_ok:
            res["agree"] += 1
            res["worst"] = max(res["worst"], dev)
            continue
        # classify
        if s1.kind == kg.SEGMENT and s2.kind == kg.SEGMENT and kg.gjk(s1, s2)[3] <= 1e-9:
            note("capsule cores cross", i)
            continue
        if s1.kind == kg.SEGMENT and s2.kind == kg.SEGMENT:
            # nearly parallel cores: the closest points are ill-conditioned, the normal is only defined to the misalignment angle
            ex, ey, fx, fy = s1.tb[0] - s1.ta[0], s1.tb[1] - s1.ta[1], s2.tb[0] - s2.ta[0], s2.tb[1] - s2.ta[1]
            sin_ab = abs(ex * fy - ey * fx) / (math.hypot(ex, ey) * math.hypot(fx, fy))
            if sin_ab < 1e-6 and dev <= 40.0 * sin_ab + TOL and ids_ok:
                note("nearly parallel capsules (|sin| < 1e-6): normals differ by less than the misalignment", i)
                continue
        if s1.kind == kg.POLY and s2.kind == kg.POLY:
            # two minimum-translation axes of equal depth (within 1e-9): SAT and EPA may each pick either
            def depth_along(nx, ny):
                return max(v[0] * nx + v[1] * ny for v in s1.verts) - min(v[0] * nx + v[1] * ny for v in s2.verts)
            if abs(depth_along(o[3], o[4]) - depth_along(n[0], n[1])) < TOL and abs(o[3] * n[0] + o[4] * n[1]) < 0.999:
                note("two axes of equal depth (tie)", i)
                continue
        if s1.kind == kg.CIRCLE and s2.kind == kg.POLY:
            # the centre within 1e-7 of the polygon's boundary: the direction of a 1e-7 vector is not defined to 1e-9
            if abs(kg.gjk(s1, s2)[3]) < 1e-6:
                note("circle centre on the polygon's boundary (|d| < 1e-6)", i)
                continue
        if dev <= TOL and not ids_ok:
            note("IDS", i)
            continue
        note("VALUES", i)
    res["gjk_iters"], res["epa_iters"] = stats.get("gjk", 0), stats.get("epa", 0)
    return res
