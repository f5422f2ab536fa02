"""A second, independent restatement of `Space.step` - the differential checker for oracle/cp_lite.c and, through
`dynenv_set_state`, for the HIP kernels.  Plain Python float64, test infrastructure only.

It is written from Chipmunk2D 7's published algorithm (cpCollision.c, cpArbiter.c, cpSpaceStep.c, cpPivotJoint.c,
cpRotaryLimitJoint.c as pymunk 5.x reaches them through `space.step(0.01)`: DrivingEnvironment.py:278, RoboCupEnvironment.py:482),
NOT from oracle/cp_lite.c, and DELIBERATELY BY ANOTHER ROUTE than the oracle wherever Chipmunk's own route differs from it:

  what                     oracle/cp_lite.c (and the kernels)                  here (Chipmunk's own route)
  poly - poly              SAT: least-penetration face of either box           GJK on the Minkowski difference, EPA when the origin is inside,
  circle - poly            face distances / closest point over the edges        closest features -> ClosestPoints -> support edges -> ContactPoints
  capsule - capsule        closest points of two segments (Ericson)            the same GJK / EPA on the two segment cores
  contact ids              (slot * 4 + vertex) pairs packed in an integer       tuples ((shape, vertex), (shape, vertex))
  arbiter cache            a fixed pool searched linearly                       a dict keyed by the unordered shape pair
  arithmetic               a * b + c fused where DESIGN.md 2a says so          every operation rounded on its own (Python floats)
  sin / cos                include/dynenv_math.h                               libm

What it shares with the oracle, because Chipmunk's answer is not knowable in this pipeline: the ORDER in which colliding pairs are
found (Chipmunk: BB-tree traversal; both restatements: ascending shape ids) - `World(order=...)` can permute it to show how much
the outcome depends on it - and the hull order of a box's vertices (cpConvexHull: starts at the lowest of the leftmost, counter-
clockwise).

Semantics restated: position update with the bias velocities; exact-AABB reject; narrowphase in shape-type
order circle < segment < poly; arbiters looked up by shape pair, impulses carried over by contact id; `begin` on a first touch
(also on a re-touch inside collision_persistence = 3, where the accumulated impulses of the old contacts are STILL carried over but
cpArbiterApplyCachedImpulse is skipped); a `begin` that returns False hides the pair until it separates; `separate` in the first
step without contact; cpArbiterPreStep (bounce sampled before the velocity function); the velocity function (DynEnv's
apply_friction, cutils.py:102-140); cached impulses; 10 sweeps over arbiters then joints; post_solve.
"""
import math
import sys

DBL_MIN = sys.float_info.min
DT = 0.01
SLOP = 0.1
ITERATIONS = 10
PERSISTENCE = 3
F32_09 = 0.89999997615814208984375          # (cpFloat)(1.0f - 0.1f): cpSpaceInit / cpConstraintInit raise a FLOAT 0.9 to the 60th
COLLISION_BIAS = F32_09 ** 60.0
MAX_GJK = 30
MAX_EPA = 30

CIRCLE, SEGMENT, POLY = 0, 1, 2


# ------------------------------------------------------------------------------------------------ bodies and shapes
class Body:
    __slots__ = ("m", "i", "m_inv", "i_inv", "px", "py", "vx", "vy", "a", "w", "vbx", "vby", "wb", "cos", "sin", "fric", "static",
                 "tag", "vel_func", "user")

    def __init__(self, m, i, px, py, a=0.0, vx=0.0, vy=0.0, w=0.0, fric=None, tag=None):
        self.static = not (m > 0.0) or math.isinf(m)
        self.m, self.i = (math.inf, math.inf) if self.static else (m, i)
        self.m_inv, self.i_inv = (0.0, 0.0) if self.static else (1.0 / m, 1.0 / i)
        self.px, self.py, self.vx, self.vy, self.a, self.w = px, py, vx, vy, a, w
        self.vbx = self.vby = self.wb = 0.0
        self.cos, self.sin = math.cos(a), math.sin(a)
        self.fric = fric      # None: cpBodyUpdateVelocity (gravity 0, damping 1: nothing); else (friction, rotFriction, spin)
        self.tag = tag
        self.vel_func = None  # a callable(body, dt) instead of `fric`: the reference's own Python velocity function (tests/golden/gen_golden_contacts.py)
        self.user = None


def moment_for_box(m, hx, hy):
    """cpMomentForPoly over the four corners (+-hx, +-hy): m (w^2 + h^2) / 12 with w = 2 hx, h = 2 hy"""
    return m * ((2.0 * hx) ** 2 + (2.0 * hy) ** 2) / 12.0


def moment_for_circle(m, r_inner, r_outer):
    return m * 0.5 * (r_inner * r_inner + r_outer * r_outer)


def moment_for_segment(m, ax, ay, bx, by, r):
    """cpMomentForSegment: a box of length |b - a| + 2 r and width 2 r about its centre, moved to the body origin"""
    length = math.hypot(bx - ax, by - ay) + 2.0 * r
    ox, oy = 0.5 * (ax + bx), 0.5 * (ay + by)
    return m * ((length * length + 4.0 * r * r) / 12.0 + (ox * ox + oy * oy))


class Shape:
    """kind CIRCLE: r (centre at the body origin); SEGMENT: la, lb (local ends), r; POLY: a box with half extents hx, hy (r = 0)"""
    __slots__ = ("kind", "body", "sid", "e", "u", "ctype", "r", "la", "lb", "ln", "lverts", "lnormals", "tc", "ta", "tb", "tn", "verts",
                 "normals", "bb", "user")

    def __init__(self, kind, body, sid, e=0.0, u=0.0, ctype=0, r=0.0, la=None, lb=None, hx=0.0, hy=0.0):
        self.kind, self.body, self.sid, self.e, self.u, self.ctype, self.r = kind, body, sid, e, u, ctype, r
        self.user = None
        if kind == SEGMENT:
            self.la, self.lb = la, lb
            dx, dy = lb[0] - la[0], lb[1] - la[1]
            ln = math.hypot(dx, dy)
            self.ln = (dy / ln, -dx / ln)                       # cpvrperp(cpvnormalize(b - a))
        if kind == POLY:
            # cpConvexHull's order for Poly([(h, w), (-h, w), (-h, -w), (h, -w)]) (Car.py:21-22, Obstacle.py:12): counter-clockwise from
            # the lowest of the leftmost vertices; plane i = (vertex i, outward normal of the edge that ENDS in vertex i)
            self.lverts = [(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)]
            self.lnormals = []
            for k in range(4):
                ax, ay = self.lverts[k - 1]
                bx, by = self.lverts[k]
                dx, dy = bx - ax, by - ay
                ln = math.hypot(dx, dy)
                self.lnormals.append((dy / ln, -dx / ln))
        self.update()

    def update(self):
        """cpShapeCacheBB: world geometry + the exact bounding box"""
        b = self.body
        c, s, px, py = b.cos, b.sin, b.px, b.py
        if self.kind == CIRCLE:
            self.tc = (px, py)
            self.bb = (px - self.r, py - self.r, px + self.r, py + self.r)
        elif self.kind == SEGMENT:
            ax, ay = self.la
            bx, by = self.lb
            self.ta = (c * ax - s * ay + px, s * ax + c * ay + py)
            self.tb = (c * bx - s * by + px, s * bx + c * by + py)
            self.tn = (c * self.ln[0] - s * self.ln[1], s * self.ln[0] + c * self.ln[1])
            r = self.r
            self.bb = (min(self.ta[0], self.tb[0]) - r, min(self.ta[1], self.tb[1]) - r, max(self.ta[0], self.tb[0]) + r,
                       max(self.ta[1], self.tb[1]) + r)
        else:
            self.verts = [(c * x - s * y + px, s * x + c * y + py) for x, y in self.lverts]
            self.normals = [(c * x - s * y, s * x + c * y) for x, y in self.lnormals]
            xs = [v[0] for v in self.verts]
            ys = [v[1] for v in self.verts]
            self.bb = (min(xs), min(ys), max(xs), max(ys))


# ------------------------------------------------------------------------------------------------ cpCollision.c: GJK / EPA
def _support(shape, nx, ny):
    """-> (point, feature index) farthest along n"""
    if shape.kind == CIRCLE:
        return shape.tc, 0
    if shape.kind == SEGMENT:
        ta, tb = shape.ta, shape.tb
        return (ta, 0) if ta[0] * nx + ta[1] * ny > tb[0] * nx + tb[1] * ny else (tb, 1)
    best, idx = -math.inf, 0
    for k, v in enumerate(shape.verts):
        d = v[0] * nx + v[1] * ny
        if d > best:
            best, idx = d, k
    return shape.verts[idx], idx


def _shape_point(shape, i):
    if shape.kind == CIRCLE:
        return shape.tc, 0
    if shape.kind == SEGMENT:
        return (shape.ta, 0) if i == 0 else (shape.tb, 1)
    k = i if i < len(shape.verts) else 0
    return shape.verts[k], k


def _mpoint(a, b):
    """MinkowskiPoint: (a, b, ab = b - a, id)"""
    (pa, ia), (pb, ib) = a, b
    return (pa, pb, (pb[0] - pa[0], pb[1] - pa[1]), (ia & 0xFF) << 8 | (ib & 0xFF))


def _msupport(s1, s2, nx, ny):
    return _mpoint(_support(s1, -nx, -ny), _support(s2, nx, ny))


def _left_of(a, b, c):
    """cpCheckPointGreater(a, b, c): c lies strictly to the left of a -> b"""
    return (b[1] - a[1]) * (a[0] + b[0] - 2.0 * c[0]) > (b[0] - a[0]) * (a[1] + b[1] - 2.0 * c[1])


def _closest_t(a, b):
    dx, dy = b[0] - a[0], b[1] - a[1]
    t = (dx * (a[0] + b[0]) + dy * (a[1] + b[1])) / (dx * dx + dy * dy + DBL_MIN)
    return -max(-1.0, min(1.0, t))


def _lerp_t(a, b, t):
    ht = 0.5 * t
    return (a[0] * (0.5 - ht) + b[0] * (0.5 + ht), a[1] * (0.5 - ht) + b[1] * (0.5 + ht))


def _closest_dist(a, b):
    p = _lerp_t(a, b, _closest_t(a, b))
    return p[0] * p[0] + p[1] * p[1]


def _closest_points(v0, v1):
    """ClosestPointsNew: closest points of the two shapes from the Minkowski edge v0 -> v1 -> (pa, pb, n, d, id)"""
    t = _closest_t(v0[2], v1[2])
    p = _lerp_t(v0[2], v1[2], t)
    pa = _lerp_t(v0[0], v1[0], t)
    pb = _lerp_t(v0[1], v1[1], t)
    cid = (v0[3] & 0xFFFF) << 16 | (v1[3] & 0xFFFF)
    dx, dy = v1[2][0] - v0[2][0], v1[2][1] - v0[2][1]
    ln = math.sqrt(dx * dx + dy * dy) + DBL_MIN
    nx, ny = dy / ln, -dx / ln                                   # cpvnormalize(cpvrperp(delta))
    d = nx * p[0] + ny * p[1]
    if d <= 0.0 or (-1.0 < t < 1.0):
        return pa, pb, (nx, ny), d, cid
    d2 = math.sqrt(p[0] * p[0] + p[1] * p[1])
    inv = 1.0 / (d2 + DBL_MIN)
    return pa, pb, (p[0] * inv, p[1] * inv), d2, cid


def _epa(s1, s2, hull, iteration, stats):
    while True:
        count = len(hull)
        mini, mind = 0, math.inf
        i = count - 1
        for j in range(count):
            d = _closest_dist(hull[i][2], hull[j][2])
            if d < mind:
                mind, mini = d, i
            i = j
        v0, v1 = hull[mini], hull[(mini + 1) % count]
        p = _msupport(s1, s2, -(v1[2][1] - v0[2][1]), v1[2][0] - v0[2][0])      # cpvperp(v1.ab - v0.ab)
        duplicate = p[3] == v0[3] or p[3] == v1[3]
        if not duplicate and _left_of(v0[2], v1[2], p[2]) and iteration < MAX_EPA:
            hull2 = [p]
            for k in range(count):
                index = (mini + 1 + k) % count
                h0 = hull2[-1][2]
                h1 = hull[index][2]
                h2 = (hull[(index + 1) % count] if k + 1 < count else p)[2]
                if _left_of(h0, h2, h1):
                    hull2.append(hull[index])
            hull, iteration = hull2, iteration + 1
            if stats is not None:
                stats["epa"] = max(stats.get("epa", 0), iteration)
            continue
        return _closest_points(v0, v1)


def _gjk_recurse(s1, s2, v0, v1, iteration, stats):
    while True:
        if iteration > MAX_GJK:
            return _closest_points(v0, v1)
        if _left_of(v1[2], v0[2], (0.0, 0.0)):
            v0, v1 = v1, v0                                        # the origin is behind the axis: flip (same iteration)
            continue
        t = _closest_t(v0[2], v1[2])
        if -1.0 < t < 1.0:
            nx, ny = -(v1[2][1] - v0[2][1]), v1[2][0] - v0[2][0]  # cpvperp(v1.ab - v0.ab): towards the origin
        else:
            q = _lerp_t(v0[2], v1[2], t)
            nx, ny = -q[0], -q[1]
        p = _msupport(s1, s2, nx, ny)
        if _left_of(p[2], v0[2], (0.0, 0.0)) and _left_of(v1[2], p[2], (0.0, 0.0)):
            if stats is not None:
                stats["epa_calls"] = stats.get("epa_calls", 0) + 1
            return _epa(s1, s2, [v0, p, v1], 1, stats)               # the triangle v0, p, v1 holds the origin
        dp = p[2][0] * nx + p[2][1] * ny
        if dp <= max(v0[2][0] * nx + v0[2][1] * ny, v1[2][0] * nx + v1[2][1] * ny):   # cpCheckAxis: no progress: the edge is closest
            return _closest_points(v0, v1)
        if _closest_dist(v0[2], p[2]) < _closest_dist(p[2], v1[2]):
            v1 = p
        else:
            v0 = p
        iteration += 1
        if stats is not None:
            stats["gjk"] = max(stats.get("gjk", 0), iteration)


def gjk(s1, s2, cid=0, stats=None):
    """closest points of two convex shapes' cores -> (pa, pb, n, d, id); `cid` = the id of the pair's previous call (warm start)"""
    if cid:
        v0 = _mpoint(_shape_point(s1, (cid >> 24) & 0xFF), _shape_point(s2, (cid >> 16) & 0xFF))
        v1 = _mpoint(_shape_point(s1, (cid >> 8) & 0xFF), _shape_point(s2, cid & 0xFF))
    else:
        c1x, c1y = 0.5 * (s1.bb[0] + s1.bb[2]), 0.5 * (s1.bb[1] + s1.bb[3])
        c2x, c2y = 0.5 * (s2.bb[0] + s2.bb[2]), 0.5 * (s2.bb[1] + s2.bb[3])
        ax, ay = -(c1y - c2y), c1x - c2x                           # cpvperp(centre1 - centre2)
        v0 = _msupport(s1, s2, ax, ay)
        v1 = _msupport(s1, s2, -ax, -ay)
    return _gjk_recurse(s1, s2, v0, v1, 1, stats)


# ------------------------------------------------------------------------------------------------ support edges, ContactPoints
def _support_edge_poly(poly, nx, ny):
    """-> ((pa, ida), (pb, idb), r, n): the edge of the polygon that faces n best"""
    _, i1 = _support(poly, nx, ny)
    count = len(poly.verts)
    i0, i2 = (i1 - 1 + count) % count, (i1 + 1) % count
    n1, n2 = poly.normals[i1], poly.normals[i2]
    if nx * n1[0] + ny * n1[1] > nx * n2[0] + ny * n2[1]:
        return (poly.verts[i0], (poly.sid, i0)), (poly.verts[i1], (poly.sid, i1)), poly.r, n1
    return (poly.verts[i1], (poly.sid, i1)), (poly.verts[i2], (poly.sid, i2)), poly.r, n2


def _support_edge_segment(seg, nx, ny):
    if seg.tn[0] * nx + seg.tn[1] * ny > 0.0:
        return (seg.ta, (seg.sid, 0)), (seg.tb, (seg.sid, 1)), seg.r, seg.tn
    return (seg.tb, (seg.sid, 1)), (seg.ta, (seg.sid, 0)), seg.r, (-seg.tn[0], -seg.tn[1])


def _clamp01(x):
    return max(0.0, min(x, 1.0))


def _lerp(a, b, t):
    return (a[0] * (1.0 - t) + b[0] * t, a[1] * (1.0 - t) + b[1] * t)


def _contact_points(e1, e2, n, d):
    """ContactPoints(): clip the two support edges against each other -> [(p1, p2, id)] (0, 1 or 2)"""
    (e1a, h1a), (e1b, h1b), r1, _ = e1
    (e2a, h2a), (e2b, h2b), r2, _ = e2
    if not d <= r1 + r2:
        return []
    nx, ny = n
    d_e1_a = e1a[0] * ny - e1a[1] * nx
    d_e1_b = e1b[0] * ny - e1b[1] * nx
    d_e2_a = e2a[0] * ny - e2a[1] * nx
    d_e2_b = e2b[0] * ny - e2b[1] * nx
    e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN)
    e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN)
    out = []
    for ta, tb, ida in ((_clamp01((d_e2_b - d_e1_a) * e1_denom), _clamp01((d_e1_a - d_e2_a) * e2_denom), (h1a, h2b)),
                        (_clamp01((d_e2_a - d_e1_a) * e1_denom), _clamp01((d_e1_b - d_e2_a) * e2_denom), (h1b, h2a))):
        q1 = _lerp(e1a, e1b, ta)
        q2 = _lerp(e2a, e2b, tb)
        p1 = (nx * r1 + q1[0], ny * r1 + q1[1])
        p2 = (nx * -r2 + q2[0], ny * -r2 + q2[1])
        if (p2[0] - p1[0]) * nx + (p2[1] - p1[1]) * ny <= 0.0:
            out.append((p1, p2, ida))
    return out


def collide(a, b, cid=0, stats=None):
    """cpCollide: -> (first shape, second shape, n, [(p1, p2, id)], collision id) with the shapes in type order"""
    if a.kind > b.kind:
        a, b = b, a
    ka, kb = a.kind, b.kind
    if ka == CIRCLE and kb == CIRCLE:
        mind = a.r + b.r
        dx, dy = b.tc[0] - a.tc[0], b.tc[1] - a.tc[1]
        dsq = dx * dx + dy * dy
        if dsq < mind * mind:
            dist = math.sqrt(dsq)
            n = (dx / dist, dy / dist) if dist else (1.0, 0.0)
            return a, b, n, [((a.tc[0] + n[0] * a.r, a.tc[1] + n[1] * a.r), (b.tc[0] - n[0] * b.r, b.tc[1] - n[1] * b.r), 0)], 0
        return a, b, (0.0, 0.0), [], 0
    if ka == CIRCLE and kb == SEGMENT:
        sx, sy = b.tb[0] - b.ta[0], b.tb[1] - b.ta[1]
        t = _clamp01((sx * (a.tc[0] - b.ta[0]) + sy * (a.tc[1] - b.ta[1])) / (sx * sx + sy * sy))
        cx, cy = b.ta[0] + sx * t, b.ta[1] + sy * t
        mind = a.r + b.r
        dx, dy = cx - a.tc[0], cy - a.tc[1]
        dsq = dx * dx + dy * dy
        if dsq < mind * mind:
            dist = math.sqrt(dsq)
            n = (dx / dist, dy / dist) if dist else b.tn
            # pymunk's Segment leaves a_tangent = b_tangent = 0: the end-cap rejection never fires
            return a, b, n, [((a.tc[0] + n[0] * a.r, a.tc[1] + n[1] * a.r), (cx - n[0] * b.r, cy - n[1] * b.r), 0)], 0
        return a, b, (0.0, 0.0), [], 0
    pa, pb, n, d, cid = gjk(a, b, cid, stats)
    if ka == CIRCLE:                                               # CircleToPoly
        if d <= a.r + b.r:
            return a, b, n, [((pa[0] + n[0] * a.r, pa[1] + n[1] * a.r), (pb[0] - n[0] * b.r, pb[1] - n[1] * b.r), 0)], cid
        return a, b, (0.0, 0.0), [], cid
    if ka == SEGMENT and kb == SEGMENT:                            # SegmentToSegment (tangents 0: no end-cap rejection)
        if stats is not None and d <= 0.0:
            stats["cores_cross"] = stats.get("cores_cross", 0) + 1  # 15 px deep: the oracle's closest-point formula has no normal there
        cons = _contact_points(_support_edge_segment(a, n[0], n[1]), _support_edge_segment(b, -n[0], -n[1]), n, d)
        if stats is not None and cons:   # touching capsules that are nearly parallel: their closest points are ill-conditioned (kat_fuzz's class)
            ex, ey, fx, fy = a.tb[0] - a.ta[0], a.tb[1] - a.ta[1], b.tb[0] - b.ta[0], b.tb[1] - b.ta[1]
            sn = abs(ex * fy - ey * fx) / (math.hypot(ex, ey) * math.hypot(fx, fy))
            if sn != 0.0:                # (exactly parallel: both restatements agree, the narrowphase fuzz shows)
                stats["min_sin_touching_capsules"] = min(stats.get("min_sin_touching_capsules", 1.0), sn)
        return a, b, (n if cons else (0.0, 0.0)), cons, cid
    if ka == POLY and kb == POLY:
        cons = _contact_points(_support_edge_poly(a, n[0], n[1]), _support_edge_poly(b, -n[0], -n[1]), n, d)
        return a, b, (n if cons else (0.0, 0.0)), cons, cid
    raise NotImplementedError("segment - poly never occurs in DynEnv")


# ------------------------------------------------------------------------------------------------ arbiters, joints, the step
class Contact:
    __slots__ = ("r1", "r2", "cid", "jn", "jt", "jb", "nMass", "tMass", "bias", "bounce")


class Arbiter:
    __slots__ = ("a", "b", "n", "contacts", "e", "u", "state", "stamp", "swapped", "handler")

    def shapes(self):
        """pymunk's arbiter.shapes: the handler's declared type order"""
        return (self.b, self.a) if self.swapped else (self.a, self.b)


FIRST, NORMAL, IGNORE, CACHED = range(4)


class Handler:
    def __init__(self, type_a, type_b, begin=None, post_solve=None, separate=None):
        self.type_a, self.type_b, self.begin, self.post_solve, self.separate = type_a, type_b, begin, post_solve, separate


class PivotJoint:
    """cpPivotJoint with both anchors given in the bodies' local frames; error_bias as pymunk sets it (Robot.py:59: 0.1)"""

    def __init__(self, a, b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), error_bias=COLLISION_BIAS):
        self.a, self.b, self.anchor_a, self.anchor_b, self.error_bias = a, b, anchor_a, anchor_b, error_bias
        self.jx = self.jy = 0.0

    def prestep(self, dt):
        a, b = self.a, self.b
        self.r1 = (a.cos * self.anchor_a[0] - a.sin * self.anchor_a[1], a.sin * self.anchor_a[0] + a.cos * self.anchor_a[1])
        self.r2 = (b.cos * self.anchor_b[0] - b.sin * self.anchor_b[1], b.sin * self.anchor_b[0] + b.cos * self.anchor_b[1])
        r1, r2 = self.r1, self.r2
        m_sum = a.m_inv + b.m_inv
        k11 = m_sum + r1[1] * r1[1] * a.i_inv + r2[1] * r2[1] * b.i_inv
        k12 = -r1[0] * r1[1] * a.i_inv - r2[0] * r2[1] * b.i_inv
        k22 = m_sum + r1[0] * r1[0] * a.i_inv + r2[0] * r2[0] * b.i_inv
        det_inv = 1.0 / (k11 * k22 - k12 * k12)
        self.k = (k22 * det_inv, -k12 * det_inv, -k12 * det_inv, k11 * det_inv)
        coef = 1.0 - self.error_bias ** dt
        dx = (b.px + r2[0]) - (a.px + r1[0])
        dy = (b.py + r2[1]) - (a.py + r1[1])
        self.bias = (dx * (-coef / dt), dy * (-coef / dt))         # maxBias = inf: not clamped

    def apply_cached(self, dt_coef):
        _apply_impulses(self.a, self.b, self.r1, self.r2, self.jx * dt_coef, self.jy * dt_coef)

    def apply(self, dt):
        a, b, r1, r2 = self.a, self.b, self.r1, self.r2
        vrx = (b.vx - r2[1] * b.w) - (a.vx - r1[1] * a.w)
        vry = (b.vy + r2[0] * b.w) - (a.vy + r1[0] * a.w)
        ex, ey = self.bias[0] - vrx, self.bias[1] - vry
        jx = ex * self.k[0] + ey * self.k[1]
        jy = ex * self.k[2] + ey * self.k[3]
        ox, oy = self.jx, self.jy
        self.jx, self.jy = ox + jx, oy + jy                         # maxForce = inf: not clamped
        _apply_impulses(a, b, r1, r2, self.jx - ox, self.jy - oy)


class RotaryLimitJoint:
    def __init__(self, a, b, lo, hi, error_bias=COLLISION_BIAS):
        self.a, self.b, self.lo, self.hi, self.error_bias = a, b, lo, hi, error_bias
        self.j = 0.0

    def prestep(self, dt):
        a, b = self.a, self.b
        dist = b.a - a.a
        pdist = 0.0
        if dist > self.hi:
            pdist = self.hi - dist
        elif dist < self.lo:
            pdist = self.lo - dist
        self.i_sum = 1.0 / (a.i_inv + b.i_inv)
        self.bias = -(1.0 - self.error_bias ** dt) * pdist / dt     # maxBias = inf
        if not self.bias:
            self.j = 0.0

    def apply_cached(self, dt_coef):
        j = self.j * dt_coef
        self.a.w -= j * self.a.i_inv
        self.b.w += j * self.b.i_inv

    def apply(self, dt):
        if not self.bias:
            return
        a, b = self.a, self.b
        wr = b.w - a.w
        j = -(self.bias + wr) * self.i_sum
        old = self.j
        self.j = max(old + j, 0.0) if self.bias < 0.0 else min(old + j, 0.0)
        j = self.j - old
        a.w -= j * a.i_inv
        b.w += j * b.i_inv


def _apply_impulses(a, b, r1, r2, jx, jy):
    a.vx -= jx * a.m_inv
    a.vy -= jy * a.m_inv
    a.w += a.i_inv * (r1[0] * -jy - r1[1] * -jx)
    b.vx += jx * b.m_inv
    b.vy += jy * b.m_inv
    b.w += b.i_inv * (r2[0] * jy - r2[1] * jx)


def apply_friction(b):
    """cutils.apply_friction (cutils.py:102-140) after Body.update_velocity, which changes nothing at gravity 0, damping 1, no force"""
    friction, rot_friction, spin = b.fric
    factor, rot_factor = friction * b.m, rot_friction * b.m
    x, y, theta = b.vx, b.vy, b.w
    length = 1.0 / (abs(x) + abs(y) + 1e-5)
    a0, a1 = x * factor * length, y * factor * length
    a0 += a1 * spin * theta
    a1 -= a0 * spin * theta
    x = 0.0 if abs(x) < factor else x - a0
    y = 0.0 if abs(y) < factor else y - a1
    theta = 0.0 if abs(theta) < rot_factor else theta - (rot_factor if theta > 0 else -rot_factor)
    b.vx, b.vy, b.w = x, y, theta


class World:
    """cpSpace.  order: 'canonical' (ascending shape ids, the oracle's convention) | 'reversed' | a random.Random to shuffle with.
    drop_cache_on_separate / begin_on_recontact: switches that build the plausible WRONG readings for the KATs' sensitivity demos."""

    def __init__(self, order="canonical", warm_gjk=True, drop_cache_on_separate=False, apply_cached_on_retouch=False):
        self.bodies, self.shapes, self.joints, self.handlers = [], [], [], []
        self.arbiters = {}
        self.active = []
        self.stamp, self.prev_dt = 0, 0.0
        self.order, self.warm_gjk = order, warm_gjk
        self.gjk_ids = {}
        self.drop_cache_on_separate, self.apply_cached_on_retouch = drop_cache_on_separate, apply_cached_on_retouch
        self.log = []           # (stamp, event, sid, sid) for 'begin', 'separate', 'retouch'
        self.stats = {}

    def add_body(self, b):
        self.bodies.append(b)
        return b

    def add_shape(self, s):
        self.shapes.append(s)
        self.shapes.sort(key=lambda x: x.sid)
        return s

    def handler_for(self, ta, tb):
        for h in self.handlers:
            if (h.type_a == ta and h.type_b == tb) or (h.type_a == tb and h.type_b == ta):
                return h
        return None

    def step(self, dt=DT):
        self.stamp += 1
        prev_dt, self.prev_dt = self.prev_dt, dt
        for arb in self.active:
            arb.state = NORMAL
        self.active = []
        # cpBodyUpdatePosition
        for b in self.bodies:
            if b.static:
                continue
            b.px += (b.vx + b.vbx) * dt
            b.py += (b.vy + b.vby) * dt
            b.a += (b.w + b.wb) * dt
            b.cos, b.sin = math.cos(b.a), math.sin(b.a)
            b.vbx = b.vby = b.wb = 0.0
        for s in self.shapes:
            if not s.body.static:
                s.update()
        # colliding pairs: exact bounding boxes overlap, two different bodies, at least one of them dynamic
        pairs = []
        dyn = [s for s in self.shapes if not s.body.static]
        sta = [s for s in self.shapes if s.body.static]
        for i, a in enumerate(dyn):
            l, bt, r, t = a.bb
            for b in dyn[i + 1:]:
                bbb = b.bb
                if l <= bbb[2] and bbb[0] <= r and bt <= bbb[3] and bbb[1] <= t and a.body is not b.body:
                    pairs.append((a, b))
            for b in sta:
                bbb = b.bb
                if l <= bbb[2] and bbb[0] <= r and bt <= bbb[3] and bbb[1] <= t:
                    pairs.append((a, b) if a.sid < b.sid else (b, a))
        pairs.sort(key=lambda ab: (ab[0].sid, ab[1].sid))
        if self.order == "reversed":
            pairs.reverse()
        elif self.order != "canonical":
            self.order.shuffle(pairs)
        seen_ids = {}
        for a, b in pairs:
            key = (a.sid, b.sid)
            cid = self.gjk_ids.get(key, 0) if self.warm_gjk else 0
            s1, s2, n, cons, cid = collide(a, b, cid, self.stats)
            seen_ids[key] = cid
            if not cons:
                continue
            arb = self.arbiters.get(key)
            if arb is None:
                arb = Arbiter()
                arb.state, arb.contacts, arb.stamp = FIRST, [], 0
                self.arbiters[key] = arb
            # cpArbiterUpdate
            arb.a, arb.b = s1, s2
            fresh = []
            for p1, p2, cid_c in cons:
                c = Contact()
                c.r1 = (p1[0] - s1.body.px, p1[1] - s1.body.py)
                c.r2 = (p2[0] - s2.body.px, p2[1] - s2.body.py)
                c.cid, c.jn, c.jt = cid_c, 0.0, 0.0
                for old in arb.contacts:
                    if old.cid == cid_c:
                        c.jn, c.jt = old.jn, old.jt
                fresh.append(c)
            arb.contacts, arb.n = fresh, n
            arb.e, arb.u = s1.e * s2.e, s1.u * s2.u
            h = arb.handler = self.handler_for(s1.ctype, s2.ctype)
            arb.swapped = h is not None and s1.ctype != h.type_a
            if arb.state == CACHED:
                arb.state = FIRST
                self.log.append((self.stamp, "retouch", a.sid, b.sid))
            if arb.state == FIRST:
                self.log.append((self.stamp, "begin", a.sid, b.sid))
                if h is not None and h.begin is not None and not h.begin(arb, self):
                    arb.state = IGNORE
            if arb.state != IGNORE and not (s1.body.static and s2.body.static):
                self.active.append(arb)
            arb.stamp = self.stamp
        self.gjk_ids = seen_ids
        # cpSpaceArbiterSetFilter
        for key in sorted(self.arbiters):
            arb = self.arbiters[key]
            ticks = self.stamp - arb.stamp
            if ticks >= 1 and arb.state != CACHED:
                arb.state = CACHED
                self.log.append((self.stamp, "separate", key[0], key[1]))
                if arb.handler is not None and arb.handler.separate is not None:
                    arb.handler.separate(arb, self)
                if self.drop_cache_on_separate:
                    arb.contacts = []
            if ticks >= PERSISTENCE:
                del self.arbiters[key]
        # prestep
        bias_coef = 1.0 - COLLISION_BIAS ** dt
        for arb in self.active:
            a, b = arb.a.body, arb.b.body
            nx, ny = arb.n
            for c in arb.contacts:
                r1, r2 = c.r1, c.r2
                r1cn, r2cn = r1[0] * ny - r1[1] * nx, r2[0] * ny - r2[1] * nx
                c.nMass = 1.0 / ((a.m_inv + a.i_inv * r1cn * r1cn) + (b.m_inv + b.i_inv * r2cn * r2cn))
                r1ct, r2ct = r1[0] * nx + r1[1] * ny, r2[0] * nx + r2[1] * ny      # r x perp(n), perp(n) = (-ny, nx)
                c.tMass = 1.0 / ((a.m_inv + a.i_inv * r1ct * r1ct) + (b.m_inv + b.i_inv * r2ct * r2ct))
                dist = ((r2[0] - r1[0]) + (b.px - a.px)) * nx + ((r2[1] - r1[1]) + (b.py - a.py)) * ny
                c.bias = -bias_coef * min(0.0, dist + SLOP) / dt
                c.jb = 0.0
                vrx = (b.vx - r2[1] * b.w) - (a.vx - r1[1] * a.w)
                vry = (b.vy + r2[0] * b.w) - (a.vy + r1[0] * a.w)
                c.bounce = (vrx * nx + vry * ny) * arb.e
        for j in self.joints:
            j.prestep(dt)
        # velocity functions
        for b in self.bodies:
            if b.static:
                continue
            if b.vel_func is not None:
                b.vel_func(b, dt)
            elif b.fric is not None:
                apply_friction(b)
        # cached impulses
        dt_coef = 0.0 if prev_dt == 0.0 else dt / prev_dt
        for arb in self.active:
            if arb.state == FIRST and not self.apply_cached_on_retouch:
                continue
            nx, ny = arb.n
            for c in arb.contacts:
                jx, jy = nx * c.jn - ny * c.jt, nx * c.jt + ny * c.jn            # cpvrotate(n, (jn, jt))
                _apply_impulses(arb.a.body, arb.b.body, c.r1, c.r2, jx * dt_coef, jy * dt_coef)
        for j in self.joints:
            j.apply_cached(dt_coef)
        # sequential impulses
        for _ in range(ITERATIONS):
            for arb in self.active:
                a, b = arb.a.body, arb.b.body
                nx, ny = arb.n
                u = arb.u
                for c in arb.contacts:
                    r1, r2 = c.r1, c.r2
                    vbx = (b.vbx - r2[1] * b.wb) - (a.vbx - r1[1] * a.wb)
                    vby = (b.vby + r2[0] * b.wb) - (a.vby + r1[0] * a.wb)
                    vrx = (b.vx - r2[1] * b.w) - (a.vx - r1[1] * a.w)
                    vry = (b.vy + r2[0] * b.w) - (a.vy + r1[0] * a.w)
                    vbn = vbx * nx + vby * ny
                    vrn = vrx * nx + vry * ny
                    vrt = -vrx * ny + vry * nx
                    jbn_old = c.jb
                    c.jb = max(jbn_old + (c.bias - vbn) * c.nMass, 0.0)
                    jn_old = c.jn
                    c.jn = max(jn_old + -(c.bounce + vrn) * c.nMass, 0.0)
                    jt_max = u * c.jn
                    jt_old = c.jt
                    c.jt = max(-jt_max, min(jt_old + -vrt * c.tMass, jt_max))
                    jb = c.jb - jbn_old
                    jbx, jby = nx * jb, ny * jb
                    a.vbx -= jbx * a.m_inv
                    a.vby -= jby * a.m_inv
                    a.wb += a.i_inv * (r1[0] * -jby - r1[1] * -jbx)
                    b.vbx += jbx * b.m_inv
                    b.vby += jby * b.m_inv
                    b.wb += b.i_inv * (r2[0] * jby - r2[1] * jbx)
                    dn, dtg = c.jn - jn_old, c.jt - jt_old
                    _apply_impulses(a, b, r1, r2, nx * dn - ny * dtg, nx * dtg + ny * dn)
            for j in self.joints:
                j.apply(dt)
        for arb in self.active:
            if arb.handler is not None and arb.handler.post_solve is not None:
                arb.handler.post_solve(arb, self)
        return len(self.active)


# ------------------------------------------------------------------------------------------------ DynEnv's bodies (Appendix B)
CT_CAR, CT_PED, CT_OBST = 1, 2, 3
CT_ROBOT, CT_BALL, CT_POST = 4, 5, 6
CAR_M = (1200.0, 1800.0, 3500.0, 5000.0)      # Car.py:9-11: masses, half lengths (x), half widths (y)
CAR_HX = (10.0, 15.0, 20.0, 25.0)
CAR_HY = (5.0, 6.0, 7.0, 8.0)
FRIC_CAR, FRIC_CAR_CRASHED, FRIC_PED_DEAD = (5e-5, 1e-5, 0.0), (5e-4, 2e-5, 0.0), (5e-2, 2e-4, 0.0)    # cutils.py:78-91
FRIC_ROBOT, FRIC_BALL = (1e-3, 1e-2, 0.0), (2.8e-2, 1e-3, 5e-2)                                      # cutils.py:93-99
BUILDINGS = ((365.0, 200.0), (365.0, 800.0), (1385.0, 200.0), (1385.0, 800.0))                      # DrivingEnvironment.py:101-106


class DrivingWorld(World):
    """The physics of a Driving scene in which every car has already crashed and every pedestrian is dead: `processAction`, `tick`
    and `move` then change nothing (Car.py:58,99; DrivingEnvironment.py:385-413 `if not car.finished`, :431 `if ped.dead`) except that
    tick zeroes a car's velocity outside [-50, W + 50] x [-50, H + 50] (:414-426) - scenes stay inside - and the collision callbacks
    reduce to what they do to the physics:
      carCrash (:591-637), carHit (:670-683): crash() again, return True
      pedHit (:640-667): |v_car| > 1 -> ped.die() (velocity := 0, Pedestrian.py:40-47), True; else False (pair hidden until it parts)
      Pedestrian - Pedestrian, Pedestrian - Obstacle: ignore_collision -> False
    Shape ids are the oracle's slots (cars 0.., pedestrians 10.., obstacles 30.., buildings 50..)."""

    def __init__(self, cars, peds, obstacles, **kw):
        """cars: (type, x, y, angle, vx, vy, w); peds: (x, y, vx, vy); obstacles: (x, y)"""
        super().__init__(**kw)
        self.cars, self.peds = [], []
        for k, (t, x, y, ang, vx, vy, w) in enumerate(cars):
            b = self.add_body(Body(CAR_M[t], moment_for_box(CAR_M[t], CAR_HX[t], CAR_HY[t]), x, y, ang, vx, vy, w, FRIC_CAR_CRASHED, ("car", k)))
            self.add_shape(Shape(POLY, b, k, e=0.05, ctype=CT_CAR, hx=CAR_HX[t], hy=CAR_HY[t]))
            self.cars.append(b)
        for k, (x, y, vx, vy) in enumerate(peds):
            b = self.add_body(Body(90.0, moment_for_circle(90.0, 0.0, 5.0), x, y, 0.0, vx, vy, 0.0, FRIC_PED_DEAD, ("ped", k)))
            self.add_shape(Shape(CIRCLE, b, 10 + k, e=0.05, ctype=CT_PED, r=5.0))         # Circle(body, radius * 2), Pedestrian.py:16
            self.peds.append(b)
        for k, (x, y) in enumerate(obstacles):
            self.add_shape(Shape(POLY, Body(0.0, 0.0, x, y), 30 + k, e=0.05, ctype=CT_OBST, hx=10.0, hy=10.0))
        for k, (x, y) in enumerate(BUILDINGS):
            self.add_shape(Shape(POLY, Body(0.0, 0.0, x, y), 50 + k, e=0.05, ctype=CT_OBST, hx=400.0, hy=225.0))
        no = lambda arb, w: False
        self.handlers = [Handler(CT_PED, CT_PED, begin=no), Handler(CT_PED, CT_OBST, begin=no), Handler(CT_CAR, CT_CAR),
                         Handler(CT_CAR, CT_PED, begin=self._ped_hit), Handler(CT_CAR, CT_OBST)]

    @staticmethod
    def _ped_hit(arb, world):
        car, ped = arb.shapes()                       # arbiter.shapes[0] is the handler's first type, the car (:643-644)
        assert car.ctype == CT_CAR and ped.ctype == CT_PED
        if math.hypot(car.body.vx, car.body.vy) > 1.0:
            ped.body.vx = ped.body.vy = 0.0
            ped.body.fric = FRIC_PED_DEAD                 # die() installs friction_pedestrian_dead (Pedestrian.py:47)
            return True
        return False


class RoboCupWorld(World):
    """The physics of a RoboCup scene in which nobody acts (moveTime = 0, no kick, head action 'none'), canFall is off, no robot is in
    a penalty box or outside the field and the ball stays inside the field lines: `tick`, `isBallOutOfField` and every collision
    callback (RoboCupEnvironment.py:1010-1146) then change flags, counters and rewards only.  Robot.py:22-60: two capsule bodies
    (-10, +-10) -> (10, +-10), radius 7.5, m = 4000 each, e = 0.3, u = 2.5, both created at the robot's position and joined there by
    PivotJoint (error_bias 0.1) and RotaryLimitJoint(0, 0); Ball.py: circle radius 10, m = 10, e = 0.98, u = 3; goalposts: static
    circles of radius 10, e = 0.95, u = 0 at (70 | 970, 370 +- 80).  Shape ids: left / right foot of robot k = 2 k / 2 k + 1, ball 20,
    posts 21..24."""
    POSTS = ((70.0, 450.0), (70.0, 290.0), (970.0, 450.0), (970.0, 290.0))

    def __init__(self, robots, ball, **kw):
        """robots: ((lx, ly, la, lvx, lvy, lw), (rx, ry, ra, rvx, rvy, rw)); ball: (x, y, vx, vy, w)"""
        super().__init__(**kw)
        self.feet = []
        for k, (L, R) in enumerate(robots):
            pair = []
            for side, (x, y, ang, vx, vy, w) in enumerate((L, R)):
                yo = 10.0 if side == 0 else -10.0
                b = self.add_body(Body(4000.0, moment_for_segment(4000.0, -10.0, yo, 10.0, yo, 7.5), x, y, ang, vx, vy, w, FRIC_ROBOT,
                                       ("foot", k, side)))
                self.add_shape(Shape(SEGMENT, b, 2 * k + side, e=0.3, u=2.5, ctype=CT_ROBOT, r=7.5, la=(-10.0, yo), lb=(10.0, yo)))
                pair.append(b)
            self.feet.append(pair)
            self.joints.append(PivotJoint(pair[0], pair[1], error_bias=0.1))
            self.joints.append(RotaryLimitJoint(pair[0], pair[1], 0.0, 0.0))
        x, y, vx, vy, w = ball
        self.ball = self.add_body(Body(10.0, moment_for_circle(10.0, 0.0, 10.0), x, y, 0.0, vx, vy, w, FRIC_BALL, ("ball",)))
        self.add_shape(Shape(CIRCLE, self.ball, 20, e=0.98, u=3.0, ctype=CT_BALL, r=10.0))
        for k, (x, y) in enumerate(self.POSTS):
            self.add_shape(Shape(CIRCLE, Body(0.0, 0.0, x, y), 21 + k, e=0.95, ctype=CT_POST, r=10.0))
        self.handlers = [Handler(CT_ROBOT, CT_POST), Handler(CT_ROBOT, CT_ROBOT), Handler(CT_ROBOT, CT_BALL)]
