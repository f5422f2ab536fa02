"""Random physics scenes for the differential test of `Space.step`: the same scene is run by tests/kat_general.py (the independent
restatement) and - through set_state / step / get_state - by the oracle (tests/test_kat_general.py) and by the HIP path
(tests/test_gpu_kat_general.py, -m gpu).  Test infrastructure only.

The scenes are chosen so that DynEnv's game logic around `space.step` changes nothing but what kat_general.DrivingWorld /
RoboCupWorld restate (see their docstrings): Driving - every car crashed, every pedestrian dead; RoboCup - nobody acts, canFall off,
robots outside the penalty boxes, everything inside the field.  A scene that would leave those conditions (a ball rolling over a
line, a robot into a penalty box) is discarded by `valid_*` on kat_general's own run, before anything else sees it."""
import math

import numpy as np

import kat_general as kg
import oracle_lib as ol

W, H = 1700.0, 1000.0


# ------------------------------------------------------------------------------------------------ Driving
def driving_scene(rng, n_players=10):
    """-> dict(cars=[(type, x, y, angle, vx, vy, w)] * n_players, peds=[(x, y, vx, vy)], obst=[(x, y)]).  k cars, some dead
    pedestrians and obstacles in a cluster on one of the two roads (often against a building's wall), closing in on each other;
    the remaining cars parked far apart on the other road."""
    horizontal = rng.random() < 0.5
    if horizontal:
        cx, cy = rng.uniform(150.0, 700.0) if rng.random() < 0.5 else rng.uniform(1050.0, 1600.0), rng.uniform(430.0, 570.0)
    else:
        cx, cy = rng.uniform(770.0, 980.0), rng.uniform(80.0, 380.0) if rng.random() < 0.5 else rng.uniform(620.0, 920.0)
    k = int(rng.integers(2, 7))
    spread = rng.uniform(25.0, 70.0)
    cars, placed = [], []

    def free(shape, allow):
        for other in placed:
            if kg.collide(shape, other)[3] and rng.random() >= allow:
                return False
        return True

    tries = 0
    while len(cars) < k and tries < 400:
        tries += 1
        t = int(rng.integers(0, 4))
        kind = rng.random()
        ang = rng.uniform(-math.pi, math.pi) if kind < 0.6 else float(rng.integers(-2, 3)) * math.pi / 2.0 + (0.0 if kind < 0.8 else rng.normal(0.0, 0.02))
        x, y = cx + rng.normal(0.0, spread), cy + rng.normal(0.0, 0.6 * spread)
        sh = kg.Shape(kg.POLY, kg.Body(1.0, 1.0, x, y, ang), len(placed), hx=kg.CAR_HX[t], hy=kg.CAR_HY[t])
        if not free(sh, 0.03):
            continue
        sp = rng.uniform(0.0, 160.0) if rng.random() < 0.85 else 0.0
        dx, dy = cx - x + rng.normal(0.0, 15.0), cy - y + rng.normal(0.0, 15.0)
        dn = math.hypot(dx, dy) + 1e-9
        cars.append((t, x, y, ang, sp * dx / dn, sp * dy / dn, rng.normal(0.0, 1.5) if rng.random() < 0.5 else 0.0))
        placed.append(sh)
    # parked cars: on the other road, clear of the crossing and of each other
    if horizontal:
        spots = [(892.5, y, math.pi / 2) for y in (60.0, 135.0, 210.0, 285.0, 360.0, 640.0, 715.0, 790.0, 865.0, 940.0)]
    else:
        spots = [(x, 517.5, 0.0) for x in (60.0, 155.0, 250.0, 345.0, 440.0, 535.0, 630.0, 1120.0, 1215.0, 1310.0, 1405.0, 1500.0, 1595.0)]
    j = 0
    while len(cars) < n_players:
        x, y, ang = spots[j]
        cars.append((int(rng.integers(0, 4)), x, y, ang, 0.0, 0.0, 0.0))
        j += 1
    peds = []
    for _ in range(int(rng.integers(0, 6))):
        for _ in range(30):
            x, y = cx + rng.normal(0.0, spread), cy + rng.normal(0.0, 0.6 * spread)
            sh = kg.Shape(kg.CIRCLE, kg.Body(1.0, 1.0, x, y), 100, r=5.0)
            if free(sh, 0.05):
                sp = rng.uniform(0.0, 60.0) if rng.random() < 0.5 else 0.0
                th = rng.uniform(0.0, 2.0 * math.pi)
                peds.append((x, y, sp * math.cos(th), sp * math.sin(th)))
                placed.append(sh)
                break
    obst = []
    for _ in range(int(rng.integers(0, 4))):
        for _ in range(30):
            x, y = cx + rng.normal(0.0, 1.3 * spread), cy + rng.normal(0.0, 0.8 * spread)
            sh = kg.Shape(kg.POLY, kg.Body(1.0, 1.0, x, y), 100, hx=10.0, hy=10.0)
            if free(sh, 0.0):
                obst.append((x, y))
                placed.append(sh)
                break
    return dict(cars=cars, peds=peds, obst=obst)


def driving_world(scene, **kw):
    return kg.DrivingWorld(scene["cars"], scene["peds"], scene["obst"], **kw)


def valid_driving(world):
    """tick zeroes a car's velocity outside [-50, W + 50] x [-50, H + 50] (DrivingEnvironment.py:414-426): stay well inside"""
    return all(-40.0 < b.px < W + 40.0 and -40.0 < b.py < H + 40.0 for b in world.cars)


def driving_expected(scene, env_steps=3, **kw):
    """kat_general's run -> (per env step: cars [n, 6] (px, py, vx, vy, angle, w), peds [m, 4]), world, valid; 10 substeps per env step"""
    w = driving_world(scene, **kw)
    out, ok = [], valid_driving(w)
    for _ in range(env_steps):
        for _ in range(10):
            w.step()
            ok = ok and valid_driving(w)
        out.append((np.array([[b.px, b.py, b.vx, b.vy, b.a, b.w] for b in w.cars]), np.array([[b.px, b.py, b.vx, b.vy] for b in w.peds]).reshape(-1, 4)))
    return out, w, ok


def driving_state(template, scene):
    """the scene as a state blob (template: any DrivingState of a handle with the same player count)"""
    st = ol.DrivingState.from_buffer_copy(template)
    st.elapsed, st.all_finished, st.n_peds, st.n_obst = 0, 0, len(scene["peds"]), len(scene["obst"])
    for k, (t, x, y, ang, vx, vy, w) in enumerate(scene["cars"]):
        c = st.cars[k]
        c.px, c.py, c.vx, c.vy, c.angle, c.w = x, y, vx, vy, ang, w
        c.dirx, c.diry, c.prevx, c.prevy = math.cos(ang), math.sin(ang), x, y
        c.type, c.finished, c.crashed, c.fric, c.lane_pos = t, 1, 1, 1, 4
    for k, (x, y, vx, vy) in enumerate(scene["peds"]):
        p = st.peds[k]
        p.px, p.py, p.vx, p.vy = x, y, vx, vy
        p.road, p.side, p.dead, p.moving, p.speed, p.crossing, p.begin_crossing = 0, 0, 1, 0, 4, 0, 0
    for k, (x, y) in enumerate(scene["obst"]):
        st.obst_x[k], st.obst_y[k] = x, y
    return st


def driving_readback(st):
    cars = np.array([[c.px, c.py, c.vx, c.vy, c.angle, c.w] for c in st.cars[:st.n_cars]])
    peds = np.array([[p.px, p.py, p.vx, p.vy] for p in st.peds[:st.n_peds]]).reshape(-1, 4)
    return cars, peds


# ------------------------------------------------------------------------------------------------ RoboCup
RC_W, RC_H = 1040.0, 740.0


def robocup_scene(rng, n_robots=10):
    """-> dict(robots=[((lx, ly, la, lvx, lvy, lw), (rx, ry, ra, rvx, rvy, rw), team)] * n_robots, ball=(x, y, vx, vy, w)).
    Robots 0..4 are team +1, 5..9 team -1 (RoboCupEnvironment.py:304-318).  k robots and the ball in a cluster - in midfield or
    beside a goalpost (then only robots whose OWN penalty box is at the other end) - the rest parked on a grid in midfield."""
    near_post = rng.random() < 0.3
    if near_post:
        post = kg.RoboCupWorld.POSTS[int(rng.integers(0, 4))]
        inward = 1.0 if post[0] < 500 else -1.0
        cx, cy = post[0] + inward * rng.uniform(25.0, 60.0), post[1] + rng.normal(0.0, 25.0)
        ids = list(range(5, 10)) if post[0] < 500 else list(range(0, 5))      # the team that defends the OTHER goal
    else:
        cx, cy = rng.uniform(250.0, 790.0), rng.uniform(150.0, 590.0)
        ids = list(range(10))
    rng.shuffle(ids)
    k = int(rng.integers(2, min(5, len(ids)) + 1))
    chosen = ids[:k]
    spread = rng.uniform(20.0, 45.0)
    robots = [None] * n_robots
    centres = []
    for rid in chosen:
        for _ in range(200):
            x, y = cx + rng.normal(0.0, spread), cy + rng.normal(0.0, spread)
            if all(math.hypot(x - q[0], y - q[1]) > 33.0 for q in centres):
                break
        centres.append((x, y))
        kind = rng.random()
        ang = rng.uniform(-math.pi, math.pi) if kind < 0.7 else float(rng.integers(-2, 3)) * math.pi / 2.0
        sp = rng.uniform(0.0, 260.0) if rng.random() < 0.85 else 0.0
        dx, dy = cx - x + rng.normal(0.0, 10.0), cy - y + rng.normal(0.0, 10.0)
        dn = math.hypot(dx, dy) + 1e-9
        vx, vy = sp * dx / dn, sp * dy / dn
        mode = rng.random()
        if mode < 0.5:        # the robot moves as a whole
            L, R = (x, y, ang, vx, vy, 0.0), (x, y, ang, vx, vy, 0.0)
        elif mode < 0.8:      # the way Robot.step / turn start a move: the left foot only (Robot.py:103-125)
            L, R = (x, y, ang, vx, vy, rng.choice((-20.0, 0.0, 20.0))), (x, y, ang, 0.0, 0.0, 0.0)
        else:                 # mid-move: feet slightly apart and at slightly different angles, different velocities
            ox, oy, da = rng.normal(0.0, 0.3), rng.normal(0.0, 0.3), rng.normal(0.0, 0.02)
            L = (x, y, ang, vx, vy, rng.normal(0.0, 2.0))
            R = (x + ox, y + oy, ang + da, vx * rng.uniform(0.7, 1.0), vy * rng.uniform(0.7, 1.0), rng.normal(0.0, 2.0))
        robots[rid] = (L, R, 1 if rid < 5 else -1)
    slots = [(220.0 + 110.0 * (j % 6), 140.0 + 150.0 * (j // 6)) for j in range(24)]
    slots = [s for s in slots if math.hypot(s[0] - cx, s[1] - cy) > 170.0]
    j = 0
    for rid in range(n_robots):
        if robots[rid] is None:
            x, y = slots[j]
            j += 1
            ang = 0.0 if rid < 5 else math.pi
            robots[rid] = ((x, y, ang, 0.0, 0.0, 0.0), (x, y, ang, 0.0, 0.0, 0.0), 1 if rid < 5 else -1)
    if rng.random() < 0.8:
        for _ in range(200):
            bx, by = cx + rng.normal(0.0, spread), cy + rng.normal(0.0, spread)
            if all(math.hypot(bx - q[0], by - q[1]) > 28.0 for q in centres):
                break
        sp = rng.uniform(0.0, 400.0)
        th = rng.uniform(0.0, 2.0 * math.pi)
        ball = (bx, by, sp * math.cos(th), sp * math.sin(th), rng.normal(0.0, 3.0) if rng.random() < 0.5 else 0.0)
    else:
        ball = (520.0, 370.0, 0.0, 0.0, 0.0)
        if math.hypot(520.0 - cx, 370.0 - cy) < 150.0:
            ball = (150.0 if cx > 520 else 890.0, 150.0, 0.0, 0.0, 0.0)
    return dict(robots=robots, ball=ball)


def robocup_world(scene, **kw):
    return kg.RoboCupWorld([(L, R) for L, R, _ in scene["robots"]], scene["ball"], **kw)


def valid_robocup_now(world):
    b = world.ball
    if not (66.0 < b.px < RC_W - 66.0 and 66.0 < b.py < RC_H - 66.0):                    # isBallOutOfField :622-640
        return False
    for Lb, Rb in world.feet:
        x, y = (Lb.px + Rb.px) / 2.0, (Lb.py + Rb.py) / 2.0
        if not (1.0 < x < RC_W - 1.0 and 1.0 < y < RC_H - 1.0):                          # leaving the field penalises (:985-986)
            return False
    return True


def robocup_expected(scene, env_steps=1, **kw):
    """-> (per env step: feet [n, 2, 6] (px, py, vx, vy, angle, w), ball [5]), world, valid; 50 substeps per env step"""
    w = robocup_world(scene, **kw)
    out, ok = [], True
    for _ in range(env_steps):
        for _ in range(50):
            w.step()
            ok = ok and valid_robocup_now(w)
        out.append((np.array([[[b.px, b.py, b.vx, b.vy, b.a, b.w] for b in pair] for pair in w.feet]),
                    np.array([w.ball.px, w.ball.py, w.ball.vx, w.ball.vy, w.ball.w])))
    return out, w, ok


def in_penalty_box(x, y, team):
    rob_x = x if team > 0 else RC_W - x
    return rob_x < 70.0 + 60.0 + 2.5 and 370.0 - 110.0 < y < 370.0 + 110.0               # tick :958-972


def valid_robocup_start(scene):
    return not any(in_penalty_box((L[0] + R[0]) / 2.0, (L[1] + R[1]) / 2.0, team) for L, R, team in scene["robots"])


def robocup_state(template, scene):
    st = ol.RoboCupState.from_buffer_copy(template)
    st.elapsed, st.ball_owned, st.n_last_kicked = 0, 0, 0
    st.n_def[0] = st.n_def[1] = 0
    st.ball_free_cntr, st.grace_period = 0.0, 0.0
    bx, by, bvx, bvy, bw = scene["ball"]
    st.bpx, st.bpy, st.bvx, st.bvy, st.bw, st.bprevx, st.bprevy = bx, by, bvx, bvy, bw, bx, by
    for k, (L, R, team) in enumerate(scene["robots"]):
        r = st.robots[k]
        r.lpx, r.lpy, r.la, r.lvx, r.lvy, r.lw = L
        r.rpx, r.rpy, r.ra, r.rvx, r.rvy, r.rw = R
        r.head_angle = r.head_moving = r.penal_time = r.fall_time = r.move_time = 0.0
        r.prevx, r.prevy = (L[0] + R[0]) / 2.0, (L[1] + R[1]) / 2.0
        r.team, r.penalized, r.touching, r.touch_cntr, r.might_push, r.fallen, r.fall_cntr = team, 0, 0, 0, 0, 0, 0
        r.kicking, r.foot, r.joint_removed = 0, 0, 0
    return st


def robocup_readback(st):
    feet = np.array([[[r.lpx, r.lpy, r.lvx, r.lvy, r.la, r.lw], [r.rpx, r.rpy, r.rvx, r.rvy, r.ra, r.rw]] for r in st.robots[:st.n_robots]])
    return feet, np.array([st.bpx, st.bpy, st.bvx, st.bvy, st.bw])


# ------------------------------------------------------------------------------------------------ comparison
def deviation(a, b):
    """largest |a - b| over a list of arrays, scaled per entry by max(1, |value|)"""
    worst = 0.0
    for x, y in zip(a, b):
        if x.size:
            worst = max(worst, float(np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y)))))
    return worst
