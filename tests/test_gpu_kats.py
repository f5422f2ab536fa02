"""Hand-derivable physics known answers on the HIP path (through set_state / get_state of the C ABI), beside the oracle:
scenes small enough that contact point, normal and impulse follow from first principles, so that the kernels are checked
against the formula and not only against the oracle (tests/test_oracle_physics.py runs the same family on the oracle's
two-body sandbox).  -m gpu."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

BIAS = 0.061259621561307376  # 1 - collisionBias ** dt (Space defaults)
SLOP = 0.1


@pytest.fixture(scope="module")
def gpu(oracle_built):
    import torch
    import dynenv_amd
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return dynenv_amd


def ball_friction(vx, vy, w):
    """cutils.apply_friction with friction_ball's coefficients (2.8e-2, 1e-3, spin 5e-2) and m = 10, as pinned by the goldens"""
    factor, rot_factor, spin = 2.8e-2 * 10.0, 1e-3 * 10.0, 5e-2
    length = 1.0 / (abs(vx) + abs(vy) + 1e-5)
    a0 = vx * factor * length
    a1 = vy * factor * length
    a0 += a1 * spin * w
    a1 -= a0 * spin * w
    vx = 0.0 if abs(vx) < factor else vx - a0
    vy = 0.0 if abs(vy) < factor else vy - a1
    w = 0.0 if abs(w) < rot_factor else w - (rot_factor if w > 0 else -rot_factor)
    return vx, vy, w


def predict_ball_off_post(p, v, post, substeps, e=0.98 * 0.95):
    """Ball (circle r = 10, m = 10) against a static goalpost (r = 10): per substep position update, then - in the substep that
    first finds |c_post - c_ball| < 20 - the normal n = (c_post - c_ball) / |..| and the restitution target e * (v . n) taken
    BEFORE the velocity function runs, the velocity function (friction), and the solve: v_n' = -e v_n(pre-friction), v_t
    unchanged (u = u_ball * u_post = 0).  A separating contact exchanges no further impulse.  -> velocity after `substeps`."""
    p, v = np.array(p, float), np.array(v, float)
    touched = False
    hit = None
    for k in range(substeps):
        p = p + v * 0.01
        d = np.array(post) - p
        dist = np.hypot(*d)
        first = dist < 20.0 and not touched
        if first:
            n = d / dist
            target = -e * (v @ n)
        vx, vy, _ = ball_friction(v[0], v[1], 0.0)
        v = np.array([vx, vy])
        if first:
            v = v + (target - v @ n) * n
            touched = True
            hit = k
    return v, hit


def _robocup_pair(dynenv_amd, seed=3):
    flags = ol.FLAG_USE_OBS_REWARDS  # canFall off: no dice anywhere
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, 1, 5, seed=seed, flags=flags)
    ora = ol.OracleEnv(env_type=0, num_envs=1, n_players=5, seed=seed, flags=flags)
    env.reset_flat()
    ora.reset()
    return env, ora


@pytest.mark.parametrize("p,v", [((125.0, 450.0), (-100.0, 0.0)),      # head-on
                                 ((118.0, 462.0), (-90.0, -20.0)),     # oblique: normal off both axes
                                 ((70.0, 332.0), (3.0, -80.0))])       # the other post of that goal, from above
def test_kat_ball_bounces_off_a_goalpost(gpu, p, v):
    env, ora = _robocup_pair(gpu)
    st = ora.get_state(0)
    st.bpx, st.bpy, st.bvx, st.bvy, st.bw = p[0], p[1], v[0], v[1], 0.0
    st.bprevx, st.bprevy = p
    env.set_state(0, st)
    ora.set_state(0, st)
    a = np.zeros((1, 10, 4), np.int32)
    a[..., 3] = 3  # head action 3 = no head turn; nobody moves
    og, rg, dg = env.step_flat(a)
    oc, rc, dc = ora.step(a)
    sg, so = env.get_state(0), ora.get_state(0)
    assert [sg.bpx, sg.bpy, sg.bvx, sg.bvy, sg.bw] == [so.bpx, so.bpy, so.bvx, so.bvy, so.bw], "HIP == oracle, bit for bit"
    post = (70.0, 450.0) if p[1] > 370 else (70.0, 290.0)
    vexp, hit = predict_ball_off_post(p, v, post, 50)
    assert hit is not None and 5 < hit < 45, "the bounce must fall inside the step"
    np.testing.assert_allclose([sg.bvx, sg.bvy], vexp, rtol=1e-9, atol=1e-9)
    assert sg.bw == 0.0
    env.close()


@pytest.mark.parametrize("car_type,pen0", [(0, 2.0), (3, 0.9)])
def test_kat_resting_car_is_pushed_out_of_an_obstacle_geometrically(gpu, car_type, pen0):
    """A car at rest overlapping a (static) obstacle by pen_0, face to face: the begin callback crashes it, no real velocity ever
    appears, and the position correction alone acts: bias velocity = biasCoef (pen - slop) / dt, applied by the next position
    update, hence  pen - slop = (pen_0 - slop) (1 - biasCoef)^k  after k position updates (10 s - 1 after s env steps; a
    two-point manifold, symmetric: no rotation).  Runs through the Driving kernel's steady-state shortcuts."""
    dynenv_amd = gpu
    half_len = [10.0, 15.0, 20.0, 25.0][car_type]
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 1, 2, seed=5)
    ora = ol.OracleEnv(env_type=1, num_envs=1, n_players=2, seed=5)
    env.reset_flat()
    ora.reset()
    st = ora.get_state(0)
    st.n_peds, st.n_obst = 0, 1
    cx, cy = 300.0, 445.0   # the walkway strip between the buildings (y < 425) and the horizontal road (y > 460)
    c = st.cars[0]
    c.px, c.py, c.vx, c.vy, c.angle, c.w, c.dirx, c.diry = cx, cy, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0
    c.prevx, c.prevy, c.type, c.finished, c.crashed, c.fric, c.lane_pos = cx, cy, car_type, 0, 0, 0, 4
    o = st.cars[1]
    o.px, o.py, o.vx, o.vy, o.w, o.prevx, o.prevy = 1500.0, 900.0, 0.0, 0.0, 0.0, 1500.0, 900.0
    st.obst_x[0], st.obst_y[0] = cx + half_len + 10.0 - pen0, cy
    env.set_state(0, st)
    ora.set_state(0, st)
    a = np.ones((1, 2, 2), np.int32)  # coast
    for s in range(1, 5):
        env.step_flat(a)
        ora.step(a)
        g, r = env.get_state(0).cars[0], ora.get_state(0).cars[0]
        assert [g.px, g.py, g.vx, g.vy, g.angle, g.w] == [r.px, r.py, r.vx, r.vy, r.angle, r.w], "HIP == oracle, step %d" % s
        pen = (g.px + half_len) - (st.obst_x[0] - 10.0)
        np.testing.assert_allclose(pen - SLOP, (pen0 - SLOP) * (1.0 - BIAS) ** (10 * s - 1), rtol=1e-4)
        # (the two contacts are relaxed one after the other, not simultaneously: the bias impulses leave a rotation of 1e-13 rad for
        #  the short car and 1e-7 rad for the long one, which is where the 1e-4 on the closed form comes from)
        assert g.vx == 0.0 and g.vy == 0.0 and g.w == 0.0 and abs(g.angle) < 1e-6 and abs(g.py - cy) < 1e-6 and g.crashed == 1
    assert env.error_flags() == 0
    env.close()


@pytest.mark.parametrize("car_type,speed", [(0, 40.0), (2, 25.0)])
def test_kat_head_on_cars_conserve_momentum_and_mirror_each_other(gpu, car_type, speed):
    """Two identical cars on one line, nose to nose, equal and opposite speeds (a mirror-symmetric scene about the contact plane):
    whatever the solver does, (1) every impulse acts on both bodies with opposite sign, so the total momentum stays 0 up to
    rounding; (2) the scene's mirror symmetry survives - v1 = -v2, the lateral velocity and the spin of a face-on two-point
    manifold cancel; (3) the cars separate no faster than e = 0.05 * 0.05 times the approach speed allows (the velocity
    function only ever slows them); (4) both are crashed (carCrash / leaving the road).  And HIP == oracle bit for bit along the way."""
    dynenv_amd = gpu
    half_len = [10.0, 15.0, 20.0, 25.0][car_type]
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 1, 2, seed=8)
    ora = ol.OracleEnv(env_type=1, num_envs=1, n_players=2, seed=8)
    env.reset_flat()
    ora.reset()
    st = ora.get_state(0)
    st.n_peds, st.n_obst = 0, 0
    cy, gap = 445.0, 2.0   # the walkway strip (see the obstacle KAT; off the road: tick() marks both crashed at once); noses `gap` apart
    x0 = 300.0
    for k, sgn in ((0, +1.0), (1, -1.0)):
        c = st.cars[k]
        c.px, c.py = x0 - sgn * (half_len + gap / 2.0), cy
        c.vx, c.vy, c.w, c.angle = sgn * speed, 0.0, 0.0, 0.0
        c.dirx, c.diry = 1.0, 0.0
        c.prevx, c.prevy, c.type, c.finished, c.crashed, c.fric, c.lane_pos = c.px, c.py, car_type, 0, 0, 0, 4
    env.set_state(0, st)
    ora.set_state(0, st)
    a = np.ones((1, 2, 2), np.int32)  # coast: (acc, steer) = (0, 0) brakes towards rest, symmetrically
    hit = False
    for s in range(3):
        env.step_flat(a)
        ora.step(a)
        g0, g1 = env.get_state(0).cars[0], env.get_state(0).cars[1]
        r0, r1 = ora.get_state(0).cars[0], ora.get_state(0).cars[1]
        for g, r in ((g0, r0), (g1, r1)):
            assert [g.px, g.py, g.vx, g.vy, g.angle, g.w, g.crashed] == [r.px, r.py, r.vx, r.vy, r.angle, r.w, r.crashed], "HIP == oracle, step %d" % s
        scale = max(abs(g0.vx), abs(g1.vx), 1e-3)
        assert abs(g0.vx + g1.vx) <= 1e-9 * scale + 1e-12, "momentum along the line (equal masses)"
        assert abs(g0.vy + g1.vy) <= 1e-6 and abs((g0.px - x0) + (g1.px - x0)) <= 1e-6, "mirror symmetry"
        # (the two contacts are relaxed one after the other: a rotation of 1e-13 .. 1e-8 rad is left, growing with the car's length)
        assert abs(g0.vy) <= 1e-6 and abs(g0.w) <= 1e-6 and abs(g0.angle) <= 1e-6, "a face-on two-point manifold neither deflects nor spins"
        assert g0.w == -g1.w or abs(g0.w + g1.w) <= 1e-12, "... and what is left is mirrored as well"
        if (g1.px - g0.px) <= 2.0 * half_len + 1e-9:  # touching (the position correction only works down to the slop)
            hit = True
            assert g0.crashed == 1 and g1.crashed == 1
            assert g0.vx <= 0.05 * 0.05 * speed + 1e-9, "separation speed bounded by e x approach speed"
            assert (g1.px - g0.px) >= 2.0 * half_len - 2.0 * speed * 0.01 - 1e-9, "never deeper than one substep of approach"
    assert hit, "the cars must have met within three steps"
    assert env.error_flags() == 0
    env.close()


# ---- round 3: joints and the friction cone through rc_step_kernel -----------------------------------------------------------
PIVOT_BIAS = 1.0 - 0.1 ** 0.01   # Robot.py:58 joint.error_bias = 0.1  ->  biasCoef = 1 - errorBias ** dt


def _quiet_actions():
    a = np.zeros((1, 10, 4), np.int32)
    a[..., 3] = 3  # head action 3 = no head turn; nobody moves
    return a


def _same_robot(g, r):
    names = ("lpx", "lpy", "lvx", "lvy", "la", "lw", "rpx", "rpy", "rvx", "rvy", "ra", "rw")
    assert [getattr(g, n) for n in names] == [getattr(r, n) for n in names], "HIP == oracle, bit for bit"


def test_kat_pivot_joint_error_decays_with_its_error_bias(gpu):
    """A robot's feet are two bodies held together by PivotJoint(left, right, pos) with error_bias = 0.1 (Robot.py:57-58; both
    anchors are the body centres).  Pull the right foot delta away along x, everything at rest: each substep the joint gives
    the pair the relative velocity -biasCoef * error / dt (maxBias = inf; the velocity function has zeroed what was left from the
    substep before: |v| < friction * m = 4), the next position update applies it, so  error_k = delta (1 - biasCoef)^k  with
    biasCoef = 1 - 0.1^dt - after s env steps k = 50 s - 1 (the first position update still sees zero velocity).  Equal masses:
    the midpoint does not move; centre anchors: no torque, the angles stay 0."""
    env, ora = _robocup_pair(gpu)
    st = ora.get_state(0)
    r = st.robots[0]
    delta = 2.0
    r.rpx = r.lpx + delta
    mid = (r.lpx + r.rpx) / 2.0
    y0 = r.lpy
    env.set_state(0, st)
    ora.set_state(0, st)
    a = _quiet_actions()
    for s in range(1, 4):
        env.step_flat(a)
        ora.step(a)
        g = env.get_state(0).robots[0]
        _same_robot(g, ora.get_state(0).robots[0])
        np.testing.assert_allclose(g.rpx - g.lpx, delta * (1.0 - PIVOT_BIAS) ** (50 * s - 1), rtol=1e-10)
        # the relative velocity left by the last solve is the bias of the error before the last position update
        np.testing.assert_allclose(g.lvx - g.rvx, PIVOT_BIAS * (g.rpx - g.lpx) / 0.01, rtol=1e-10)
        assert abs((g.lpx + g.rpx) / 2.0 - mid) < 1e-9 and g.lpy == y0 and g.rpy == y0
        assert g.la == 0.0 and g.ra == 0.0 and g.lw == 0.0 and g.rw == 0.0
    assert env.error_flags() == 0
    env.close()


def test_kat_rotary_limit_joint_pulls_the_feet_back_to_its_limit(gpu):
    """RotaryLimitJoint(left, right, 0, 0) (Robot.py:59): the right foot turned theta past the (zero-width) limit.  The joint's
    bias is -biasCoef * pdist / dt with the DEFAULT error bias (1 - 0.1f)^60, i.e. biasCoef = BIAS, the collision constant; the
    feet have the same moment of inertia, so each takes half and the mean angle stays theta / 2:
    angle error_k = theta (1 - BIAS)^k, k = 50 s - 1.  (|w| < rotFriction * m = 40 is zeroed by the velocity function every
    substep and put back by the warm start - the closed form holds only if the cached impulse is carried over correctly.)"""
    env, ora = _robocup_pair(gpu)
    st = ora.get_state(0)
    r = st.robots[0]
    theta = 0.05
    r.ra = r.la + theta
    x0, y0 = r.lpx, r.lpy
    env.set_state(0, st)
    ora.set_state(0, st)
    a = _quiet_actions()
    for s in range(1, 4):
        env.step_flat(a)
        ora.step(a)
        g = env.get_state(0).robots[0]
        _same_robot(g, ora.get_state(0).robots[0])
        np.testing.assert_allclose(g.ra - g.la, theta * (1.0 - BIAS) ** (50 * s - 1), rtol=1e-9)
        np.testing.assert_allclose(g.lw - g.rw, BIAS * (g.ra - g.la) / 0.01, rtol=1e-9)
        np.testing.assert_allclose((g.la + g.ra) / 2.0, theta / 2.0, rtol=1e-12)
        assert (g.lpx, g.lpy, g.rpx, g.rpy) == (x0, y0, x0, y0) and g.lvx == 0.0 and g.rvx == 0.0
    assert env.error_flags() == 0
    env.close()


def test_kat_grazing_ball_slides_on_the_coulomb_cone(gpu):
    """The ball (m = 10, r = 10, I = m r^2 / 2, friction 3.0) grazes the flat side of a foot capsule (friction 2.5: u = 7.5) at
    (100, -2): far too fast tangentially for friction to stop the sliding within the cone (it would take jt = m vt / 3 against
    u jn ~ 7.5 * 1.29 m |vn|), so every iteration leaves the accumulated tangent impulse clamped at u * jn.  The scene is timed
    with the closed-form free flight so that the FIRST touch happens in the last substep of the env step: the state read back
    is the state right after that one solve, and the ball's velocity before it is the closed form.  Then, exactly:
      dp_t / dp_n = -u            (the clamp - only the arbiter touches the ball)
      I dw = r x dp = 10 dp_x      (the impulse acts at the contact point, straight below the centre)
      dp(ball) + dp(both feet) = 0 (the pivot between the feet is internal)
    and jn ~ (1 + e) m |vn| with e = 0.98 * 0.3 up to the recoil of the 400x heavier foot."""
    env, ora = _robocup_pair(gpu)
    st = ora.get_state(0)
    r = st.robots[0]
    assert r.team == 1 and r.la == 0.0
    thr = r.lpy + 10.0 + 17.5   # the left foot's axis is y = py + 10 for x in [px - 10, px + 10]; touch at centre distance 17.5

    def flight(p0, v0, n):
        p, v, w, traj = np.array(p0, float), np.array(v0, float), 0.0, []
        for _ in range(n):
            p = p + v * 0.01
            traj.append((p.copy(), v.copy()))
            vx, vy, w = ball_friction(v[0], v[1], w)
            v = np.array([vx, vy])
        return traj, v, w
    v0 = (100.0, -2.0)
    d50 = flight((0.0, 0.0), v0, 50)[0][49][0]
    p0 = np.array([r.lpx - d50[0], thr - 0.01 - d50[1]])
    traj, v_pre, w_pre = flight(p0, v0, 50)
    assert traj[48][0][1] > thr > traj[49][0][1] and abs(traj[49][0][0] - r.lpx) < 1e-9, "first touch in substep 49, above the capsule's middle"
    st.bpx, st.bpy, st.bvx, st.bvy, st.bw = p0[0], p0[1], v0[0], v0[1], 0.0
    st.bprevx, st.bprevy = p0
    env.set_state(0, st)
    ora.set_state(0, st)
    a = _quiet_actions()
    env.step_flat(a)
    ora.step(a)
    sg, so = env.get_state(0), ora.get_state(0)
    assert [sg.bpx, sg.bpy, sg.bvx, sg.bvy, sg.bw] == [so.bpx, so.bpy, so.bvx, so.bvy, so.bw], "HIP == oracle, bit for bit"
    _same_robot(sg.robots[0], so.robots[0])
    np.testing.assert_allclose([sg.bpx, sg.bpy], traj[49][0], rtol=0, atol=1e-9)   # nothing moved it before the touch
    m, inertia = 10.0, 10.0 * 10.0 * 10.0 / 2.0
    dp = m * (np.array([sg.bvx, sg.bvy]) - v_pre)
    assert dp[1] > 0.0 and dp[0] < 0.0, "pushed away from the foot, slowed along it"
    np.testing.assert_allclose(dp[0] / dp[1], -7.5, rtol=1e-12)
    np.testing.assert_allclose(inertia * (sg.bw - w_pre), 10.0 * dp[0], rtol=1e-12)
    g = sg.robots[0]
    np.testing.assert_allclose(4000.0 * np.array([g.lvx + g.rvx, g.lvy + g.rvy]), -dp, rtol=1e-9)
    np.testing.assert_allclose(dp[1], (1.0 + 0.98 * 0.3) * m * abs(traj[49][1][1]), rtol=1e-2)
    assert sg.bvx > 0.5 * v_pre[0], "still sliding: the friction did not stop the ball"
    assert env.error_flags() == 0
    env.close()


# ---- round 4: Chipmunk-specific behaviours through the C ABI (scenes and expected values: tests/kat_scenes_r4.py) --------------
import kat_scenes_r4 as ks  # noqa: E402


def _drv_pair(dynenv_amd, n_players, seed):
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 1, n_players, seed=seed)
    ora = ol.OracleEnv(env_type=1, num_envs=1, n_players=n_players, seed=seed)
    env.reset_flat()
    ora.reset()
    return env, ora


def _car_tuple(c):
    return [c.px, c.py, c.vx, c.vy, c.angle, c.w, c.crashed, c.finished]


def test_kat_r4_offset_face_contact_two_point_manifold(gpu):
    """K1: ContactPoints puts the contacts of an offset box-box face contact at the ends of the OVERLAP (here +2 and +5 above the
    car's centre), and the inner one alone stops the car: v_x' = v_f - jn / m, w' = 2 jn / I, jn = (e v_pre + v_f) / (1/m + 4/I)."""
    env, ora = _drv_pair(gpu, 2, 5)
    g, want, chain_car = ks.k1_offset_face_contact(ks.Sim(env), 2)
    o, _, _ = ks.k1_offset_face_contact(ks.Sim(ora), 2)
    assert _car_tuple(g) == _car_tuple(o), "HIP == oracle, bit for bit"
    assert g.crashed == 1 and g.vy == 0.0 and g.py == ks.CY and g.angle == 0.0
    np.testing.assert_allclose([g.px, g.vx, g.w], [want["px"], want["vx"], want["w"]], rtol=1e-12)
    np.testing.assert_allclose([g.px, g.vx, g.w], [chain_car["x"], chain_car["vx"], chain_car["w"]], rtol=1e-12)
    assert g.w > 1.0, "an impulse 2 above the centre of mass spins the car counter-clockwise; corner contacts (+-5) would not"
    assert env.error_flags() == 0
    env.close()


@pytest.mark.parametrize("target", [48, 49, 50, 51])
def test_kat_r4_separate_fires_in_the_substep_the_shapes_part(gpu, target):
    """K2: two robots slide past each other foot against foot, inside the slop (no impulse: closed-form slide); `touching` (set by
    robotPushingDet at begin, cleared by RoboCupEnvironment.separate) must drop in the very substep the capsules part: parted
    in substep 48 / 49 -> clear after env step 1; in 50 / 51 -> still set after step 1, clear after step 2."""
    dynenv_amd = gpu
    env, ora = _robocup_pair(dynenv_amd)
    out = ks.k2_separate_in_substep(ks.Sim(env), target)
    assert out == ks.k2_separate_in_substep(ks.Sim(ora), target), "HIP == oracle"
    _same_robot(env.get_state(0).robots[0], ora.get_state(0).robots[0])
    _same_robot(env.get_state(0).robots[1], ora.get_state(0).robots[1])
    (ta1, tb1, _, _, dev1), (ta2, tb2, _, _, dev2) = out
    assert dev1 < 1e-9 and dev2 < 1e-9, "the slide must stay the closed form"
    assert (ta1, tb1) == ((0, 0) if target < 50 else (1, 1)) and (ta2, tb2) == (0, 0)
    assert env.error_flags() == 0
    env.close()


def test_kat_r4_begin_returning_false_hides_the_pair_until_it_separates(gpu):
    """K3: pedHit returns False for a car at |v| <= 1 (DrivingEnvironment.py:649, 664-665): Chipmunk then ignores the pair until the
    shapes separate.  The car accelerates THROUGH the pedestrian (24 px/s while inside it): it moves and is rewarded exactly as if
    the pedestrian were not there, the pedestrian exactly as if the car were not there, nobody dies, nobody crashes."""
    dynenv_amd = gpu
    runs = {}
    for variant in ("with", "no_ped", "no_car"):
        env, ora = _drv_pair(dynenv_amd, 10, 3)
        runs[variant] = ks.k3_run(ks.Sim(env), variant)
        if variant == "with":
            assert runs[variant] == ks.k3_run(ks.Sim(ora), variant), "HIP == oracle, every step"
        assert env.error_flags() == 0
        env.close()
    a, no_ped, no_car = runs["with"], runs["no_ped"], runs["no_car"]
    assert a[0][0][2] == 0.0
    assert [x[0] for x in a] == [x[0] for x in no_ped] and [x[2] for x in a] == [x[2] for x in no_ped]
    assert [x[1] for x in a] == [x[1] for x in no_car]
    assert all(x[1][4] == 0 for x in a) and all(x[0][6] == 0 for x in a)
    v = [x[0][2] for x in a]
    np.testing.assert_allclose(v[1:], [2.4 * k for k in range(1, len(v))], rtol=1e-5)
    assert v[10] > 20.0 and a[10][0][0] + 10.0 > 691.0 and a[10][0][0] - 10.0 < 701.0, "the car is inside the pedestrian at 24 px/s"
    assert a[-1][0][0] - 10.0 > 701.0


@pytest.mark.parametrize("arrive", [9, 8])
def test_kat_r4_unconverged_chain_order_levers_and_warm_start(gpu, arrive):
    """K4: three cars and a wall, six coupled contacts that ten iterations leave far from converged: arbiter order, contact order,
    lever arms and - arrive = 8 - cached impulses re-applied and carried over by hash decide the outcome; tests/kat_chain.py says
    what it is (1e-11 for one solve; 1 % when a second, slightly rotated substep follows).  Leaving the cached impulses out would
    change car 0's speed by 90 %, the other contact order flips the sign of its spin (tests/test_oracle_kats_r4.py)."""
    env, ora = _drv_pair(gpu, 3, 5)
    cars, want = ks.k4_chain_against_a_wall(ks.Sim(env), arrive)
    ocars, _ = ks.k4_chain_against_a_wall(ks.Sim(ora), arrive)
    for c, o in zip(cars, ocars):
        assert _car_tuple(c) == _car_tuple(o), "HIP == oracle, bit for bit"
    got = np.array([[c.px, c.vx, c.w] for c in cars])
    exp = np.array([[b["x"], b["vx"], b["w"]] for b in want])
    if arrive == 9:
        np.testing.assert_allclose(got, exp, rtol=1e-11)
    else:
        np.testing.assert_allclose(got[:, 0], exp[:, 0], rtol=1e-9)
        np.testing.assert_allclose(got[:2, 1], exp[:2, 1], rtol=1e-2)
    assert abs(got[0, 2]) > 1e-3, "ten iterations must not have converged"
    assert env.error_flags() == 0
    env.close()
