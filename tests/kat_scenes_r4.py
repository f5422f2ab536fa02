"""Round-4 known-answer scenes for the Chipmunk-specific behaviours of `Space.step`, written once and run twice:
on the oracle (tests/test_oracle_kats_r4.py, CPU) and through the C ABI on the HIP path (tests/test_gpu_kats.py, -m gpu, where
HIP == oracle is asserted bit for bit on the same scene as well).  `sim` is anything with get_state(env) / set_state(env, st) and
step(actions) - see Sim below.  The expected values come from first principles: closed forms, or tests/kat_chain.py (an
independent float64 restatement of Chipmunk's step for boxes in a row, not derived from oracle/cp_lite.c).

  K1  two-point box-box manifold of ContactPoints, offset face contact: where p1 / p2 sit, and which comes first
  K2  the `separate` callback fires in the very substep in which the shapes stop overlapping (not when the cached arbiter
      expires collision_persistence = 3 substeps later)
  K3  a `begin` callback that returns False (pedHit at v <= 1, DrivingEnvironment.py:640-665) makes the pair invisible until it
      separates - even after the car has sped up
  K4  arbiter order, contact order and impulse carry-over (warm start by contact hash) in a chain that 10 iterations do not converge
  (the restitution target e_a e_b (v . n) sampled at cpArbiterPreStep, BEFORE the velocity function, enters K1 and K4 in closed
   form; for circles it is tests/test_gpu_kats.py::test_kat_ball_bounces_off_a_goalpost)"""
import math

import numpy as np

import kat_chain as kc
import oracle_lib as ol


class Sim:
    """adapter: the oracle (OracleEnv) or the HIP path (BatchedDynEnv) behind one step()"""

    def __init__(self, backend):
        self.b = backend
        self.hip = hasattr(backend, "step_flat")

    def get_state(self, e=0):
        return self.b.get_state(e)

    def set_state(self, e, st):
        self.b.set_state(e, st)

    def step(self, a):
        if self.hip:
            o, r, d = self.b.step_flat(a, auto_reset=False)
            return o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        o, r, d = self.b.step(a)
        return o.copy(), r.copy(), d.copy()


# ------------------------------------------------------------------------------------------------ Driving scenes
CY = 445.0   # the walkway strip between the buildings (y < 425) and the horizontal road (y > 460): nothing else is near
X_FACE = 400.0


def _place_crashed_car(c, t, x, y, vx):
    c.px, c.py, c.vx, c.vy, c.angle, c.w, c.dirx, c.diry = x, y, vx, 0.0, 0.0, 0.0, 1.0, 0.0
    c.prevx, c.prevy, c.type, c.finished, c.crashed, c.fric, c.lane_pos = x, y, t, 1, 1, 1, 4


def _arrival_x(t, v0, arrive, face_x):
    """centre x from which a crashed car of type t at speed v0 first pushes its front face past `face_x` in the position update of
    substep `arrive` (0-based), 40 % of that substep's travel deep"""
    d_before, _ = kc.free_travel(t, v0, arrive)
    d_after, _ = kc.free_travel(t, v0, arrive + 1)
    return face_x - kc.CAR_HX[t] - d_after + 0.4 * (d_after - d_before)


def k1_offset_face_contact(sim, n_players, v0=60.0, off=12.0):
    """K1.  A crashed car (type 0: 20 x 10, m = 1200) slides in +x into a static obstacle box (20 x 20) whose centre sits `off` = 12
    above the car's: the faces overlap over y in [cy + 2, cy + 5] only.  ContactPoints puts the two contacts at the ENDS OF THE
    OVERLAP (not at the car's corners), the lower one first.  The first touch is timed into the LAST substep of the env step, so
    the state read back is the state right after one solve.  The car's centre of mass lies below the patch, so the contact at
    y = cy + 2 takes the whole impulse in the first iteration and the one at cy + 5 stays at zero:
        jn = (e v_pre + v_f) / (1/m + 2^2 / I),   v_x' = v_f - jn / m,   w' = 2 jn / I        (e = 0.05 * 0.05)
    v_pre = the speed before the substep's velocity function (the restitution target is sampled at cpArbiterPreStep), v_f after.
    With the contacts at the car's corners (+-5) the centre of mass would lie between them and the spin would vanish
    (tests/test_oracle_kats_r4.py shows that).  Which contact comes first is immaterial here (the outer one is clamped to zero
    either way); K4 pins the order."""
    st = sim.get_state(0)
    st.n_peds, st.n_obst = 0, 1
    x0 = _arrival_x(0, v0, 9, X_FACE)
    _place_crashed_car(st.cars[0], 0, x0, CY, v0)
    for k in range(1, n_players):
        _place_crashed_car(st.cars[k], 0, 1500.0, 500.0 + 40.0 * k, 0.0)
    st.obst_x[0], st.obst_y[0] = X_FACE + 10.0, CY + off
    sim.set_state(0, st)
    sim.step(np.ones((1, n_players, 2), np.int32))
    g = sim.get_state(0).cars[0]
    # closed form
    m, I, e = 1200.0, kc.box_inertia(1200.0, 10.0, 5.0), 0.05 * 0.05
    x9, v_pre = kc.free_travel(0, v0, 9)          # after 9 substeps of free flight
    c = kc.crashed_car(0, 0.0, 0.0, v_pre)
    kc.apply_friction(c)
    v_f = c["vx"]
    lever = (CY + off - 10.0) - CY                # the overlap's lower end relative to the centre of mass: +2
    jn = (e * v_pre + v_f) / (1.0 / m + lever * lever / I)
    want = dict(px=x0 + x9 + v_pre * kc.DT, vx=v_f - jn / m, w=lever * jn / I)
    # the same through the independent chain restatement (two contacts, 10 iterations)
    ch = kc.Chain([kc.crashed_car(0, x0, CY, v0), kc.obstacle(X_FACE + 10.0, CY + off)])
    touched = [ch.substep() for _ in range(10)]
    assert touched == [0] * 9 + [1], touched
    assert ch.active_log[-1][0][2][1] == 0.0, "the outer contact must have stayed at zero impulse"
    return g, want, ch.b[0]


def k4_chain_against_a_wall(sim, arrive):
    """K4.  Three crashed cars in a row (types 0, 2, 1; masses 1200, 3500, 1800), faces parallel, centres on one line, the last one
    resting against a static obstacle; cars 1 | 2 and car 2 | obstacle start 0.05 deep (inside the slop: no position correction).
    Car 0 arrives at 80 px/s in substep `arrive`.  Three arbiters x two contacts share bodies: ten Gauss-Seidel iterations are far
    from converged, so the outcome depends on the ORDER (arbiters in canonical pair order, the lower contact of each manifold
    first), on the lever arms (the ends of the face overlap: +-5, +-6, +-6) and - for arrive = 8, where the step ends one substep
    after the impact - on the cached impulses being re-applied and carried over by contact hash.  Expected values: kat_chain.Chain."""
    types, v0 = (0, 2, 1), 80.0
    st = sim.get_state(0)
    st.n_peds, st.n_obst = 0, 1
    xs = [_arrival_x(types[0], v0, arrive, X_FACE), X_FACE + kc.CAR_HX[types[1]], 0.0, 0.0]
    xs[2] = xs[1] + kc.CAR_HX[types[1]] + kc.CAR_HX[types[2]] - 0.05
    xs[3] = xs[2] + kc.CAR_HX[types[2]] + 10.0 - 0.05
    for k in range(3):
        _place_crashed_car(st.cars[k], types[k], xs[k], CY, v0 if k == 0 else 0.0)
    st.obst_x[0], st.obst_y[0] = xs[3], CY
    sim.set_state(0, st)
    sim.step(np.ones((1, 3, 2), np.int32))
    ch = kc.Chain([kc.crashed_car(t, x, CY, v) for t, x, v in zip(types, xs, (v0, 0.0, 0.0))] + [kc.obstacle(xs[3], CY)])
    touched = [ch.substep() for _ in range(10)]
    assert touched == [2] * arrive + [3] * (10 - arrive), touched
    return [sim.get_state(0).cars[k] for k in range(3)], ch.b[:3]


def k3_scene(variant, v0=0.9):
    """K3 scene: car 0 (type 0) in the right lane of the horizontal road at 0.9 px/s, its nose 2 px inside a standing pedestrian.
    variant: 'with' | 'no_ped' (the pedestrian stands elsewhere) | 'no_car' (the car drives elsewhere)"""
    st = ol.DrivingState()
    st.n_cars, st.n_peds, st.n_obst, st.episode = 10, 1, 0, 1
    for i in range(10):  # the other cars are parked far apart on the vertical road
        c = st.cars[i]
        c.px, c.py, c.angle = 892.5, 40.0 + 95.0 * i, np.pi / 2
        c.dirx, c.diry = 6.123233995736766e-17, 1.0
        c.prevx, c.prevy, c.goalx, c.goaly = c.px, c.py, 875.0, 1000.0
        c.type, c.lane_pos = i % 4, 4
    p = st.peds[0]
    p.px, p.py, p.road, p.side, p.speed, p.moving = 696.0, 484.0, 1, 0, 4, 100000
    a = st.cars[0]
    a.px, a.py, a.angle, a.vx, a.vy = 683.0, 482.5, 0.0, v0, 0.0
    a.dirx, a.diry, a.goalx, a.goaly = 1.0, 0.0, 1750.0, 500.0
    if variant == "no_ped":
        p.px, p.py = 300.0, 440.0
    if variant == "no_car":
        a.px, a.py = 200.0, 517.5
    a.prevx, a.prevy = a.px, a.py
    return st


def k3_run(sim, variant, steps=30):
    """-> per step: (car 0's kinematic state + flags, the pedestrian's state, car 0's reward)"""
    sim.set_state(0, k3_scene(variant))
    acts = np.ones((1, 10, 2), np.int32)   # step 0: everybody brakes (car 0 stops dead: v = 0 <= 1 when `begin` fires)
    out = []
    for s in range(steps):
        if s == 1:
            acts[0, 0, 0] = 2              # from step 1 on car 0 accelerates: +3 px/s per step, 10 x 0.06 friction
        o, r, d = sim.step(acts)
        g = sim.get_state(0)
        c, p = g.cars[0], g.peds[0]
        out.append(((c.px, c.py, c.vx, c.vy, c.angle, c.w, c.crashed, c.finished), (p.px, p.py, p.vx, p.vy, p.dead, p.moving), float(r[0, 0])))
    return out


# ------------------------------------------------------------------------------------------------ RoboCup scene
def robot_friction(v, m=4000.0, fr=1e-3):
    """friction_robot (cutils.py:94-95) on a velocity along x with no spin"""
    factor = fr * m
    return 0.0 if abs(v) < factor else v - v * factor * (1.0 / (abs(v) + 1e-5))


def robot_travel(v0, n):
    x, v, xs = 0.0, v0, []
    for _ in range(n):
        x += v * 0.01
        xs.append(x)
        v = robot_friction(v)
    return xs


K2_PEN, K2_VA, K2_VB = 0.05, 300.0, 230.0


def k2_separate_in_substep(sim, target):
    """K2.  Robots 0 and 1 (same team, angle 0) slide in +x side by side, both feet of each at the robot's speed (the joints stay
    inert): robot 0's LEFT foot (capsule axis y + 10) runs along robot 1's RIGHT foot (axis y - 10) 14.95 apart - 0.05 inside the
    15 of two radii, less than the slop, normal straight up: no impulse, no friction, no position correction; both only lose
    4 px/s per substep to friction_robot.  Robot 0 is faster, passes, and the capsules stop overlapping once its segment start is
    g = sqrt(0.05 * 29.95) beyond the other's segment end.  The start is chosen so that this happens in substep `target` (0-based,
    50 per env step).  RoboCupEnvironment.separate (:1084-1094) clears `touching`, which robotPushingDet set at `begin`.
    -> touching flags of both robots after env steps 1 and 2, and the deviation of both robots from the closed-form slide."""
    gsep = math.sqrt(K2_PEN * (30.0 - K2_PEN))
    xa, xb = robot_travel(K2_VA, 100), robot_travel(K2_VB, 100)
    rel = [a - b for a, b in zip(xa, xb)]
    d0 = 0.5 * (rel[target - 1] + rel[target]) - (20.0 + gsep)   # robot 0 starts d0 behind robot 1
    assert 0.0 < d0 < 19.0, "both capsules must overlap side by side from the start"
    st = sim.get_state(0)
    ax, ay = 150.0, 640.0
    bx, by = ax + d0, ay + 35.0 - K2_PEN
    for r, x, y, v in ((st.robots[0], ax, ay, K2_VA), (st.robots[1], bx, by, K2_VB)):
        r.lpx = r.rpx = x; r.lpy = r.rpy = y; r.lvx = r.rvx = v; r.lvy = r.rvy = 0.0; r.la = r.ra = 0.0; r.lw = r.rw = 0.0
        r.prevx, r.prevy = x, y
    sim.set_state(0, st)
    a = np.zeros((1, 10, 4), np.int32)
    a[..., 3] = 3   # head action 3 = no head turn; nobody walks
    out = []
    for s in range(2):
        sim.step(a)
        g = sim.get_state(0)
        A, B = g.robots[0], g.robots[1]
        k = 50 * (s + 1) - 1
        dev = max(abs(A.lpx - ax - xa[k]), abs(A.rpx - ax - xa[k]), abs(B.lpx - bx - xb[k]), abs(B.rpx - bx - xb[k]),
                  abs(A.lpy - ay), abs(B.lpy - by), abs(A.la), abs(B.la))
        out.append((A.touching, B.touching, A.touch_cntr, B.touch_cntr, dev))
    return out
