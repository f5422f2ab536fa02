"""Differential tests of the oracle's `Space.step` (oracle/cp_lite.c - PARITY UNPINNED against pymunk, which this pipeline cannot
run) against tests/kat_general.py, a second restatement written from Chipmunk 7's published algorithm by ANOTHER ROUTE (GJK / EPA
closest features -> ContactPoints instead of SAT; feature-tuple contact ids; dict arbiter cache; unfused arithmetic; libm sin / cos).
CPU suite; tests/test_gpu_kat_general.py runs the same scenes through `dynenv_set_state` on the HIP path.

  (a) narrowphase: 10^5 random configurations per shape pair (rotated box-box incl. near-parallel faces and corner-corner,
      box-circle, capsule-capsule incl. parallel feet, capsule-circle, circle-circle): contact count, order, normal, both points
      and the contact ids to 1e-9; every disagreement must fall into one of the catalogued classes below;
  (b) 2000 random Driving scenes x 30 substeps and 500 RoboCup scenes x 50 substeps, full kinematic state to 1e-9, with the same
      rule: a scene above 1e-9 must be explained by a measured property of the scene (capsule cores crossing; ill-conditioning:
      a 1e-15 relative nudge of the inputs moves the independent restatement itself by more than the deviation / 1000);
  (c) KATs: re-contact inside collision_persistence with a third body (K5), handler argument order (K6), and the two plausible
      wrong readings of the re-contact rule shown to be caught by (b)'s population.
"""
import math

import numpy as np
import pytest

import kat_fuzz as kf
import kat_general as kg
import kat_worlds as kw
import oracle_lib as ol

# classes of narrowphase disagreement that are properties of the configuration, not of either implementation
EXPLAINED = {
    "contact at a threshold (|gap| < 1e-9): count differs",
    "two axes of equal depth (tie)",
    "capsule cores cross",
    "nearly parallel capsules (|sin| < 1e-6): normals differ by less than the misalignment",
    "circle centre on the polygon's boundary (|d| < 1e-6)",
}
N_PAIR = 100000


@pytest.fixture(scope="module", autouse=True)
def _built(oracle_built):
    return oracle_built


def test_constants_and_moments_agree_with_the_oracle():
    l = ol.lib()
    l.oracle_bias_coef.restype = __import__("ctypes").c_double
    assert abs(l.oracle_bias_coef(0) - (1.0 - kg.COLLISION_BIAS ** kg.DT)) < 1e-17
    assert abs(l.oracle_bias_coef(1) - (1.0 - 0.1 ** kg.DT)) < 1e-17
    for t in range(4):
        m, hx, hy = kg.CAR_M[t], kg.CAR_HX[t], kg.CAR_HY[t]
        assert abs(l.oracle_moment_for_box(m, hx, hy) / kg.moment_for_box(m, hx, hy) - 1.0) < 1e-15
    assert abs(l.oracle_moment_for_circle(10.0, 0.0, 10.0) - kg.moment_for_circle(10.0, 0.0, 10.0)) < 1e-12
    for yo in (10.0, -10.0):
        assert abs(l.oracle_moment_for_segment(4000.0, -10.0, yo, 10.0, yo, 7.5) / kg.moment_for_segment(4000.0, -10.0, yo, 10.0, yo, 7.5) - 1.0) < 1e-15


@pytest.mark.parametrize("pair", kf.PAIRS)
def test_narrowphase_fuzz_gjk_epa_against_the_oracle(pair):
    A, B = kf.generate(pair, N_PAIR, 20261004)
    res = kf.compare(pair, A, B, kf.oracle_collide(A, B))
    assert res["colliding"] > 0.4 * N_PAIR, "the generator must produce contacts"
    unexplained = {c: n for c, n in res["classes"].items() if c not in EXPLAINED}
    assert not unexplained, (unexplained, {c: res["examples"][c] for c in unexplained})
    assert res["agree"] + sum(res["classes"].values()) == N_PAIR
    assert res["agree"] >= 0.99 * N_PAIR, res["classes"]
    if pair in ("box_circle", "capsule_circle", "circle_circle"):
        assert res["agree"] == N_PAIR, res["classes"]


@pytest.mark.parametrize("pair", ("box_box", "box_circle", "capsule_capsule"))
def test_narrowphase_fuzz_with_gjk_warm_started_like_chipmunk(pair):
    """Chipmunk starts GJK from the closest features of the pair's previous call (the collision id); the answer must not depend on it"""
    A, B = kf.generate(pair, 20000, 7)
    res = kf.compare(pair, A, B, kf.oracle_collide(A, B), warm=True)
    unexplained = {c: n for c, n in res["classes"].items() if c not in EXPLAINED}
    assert not unexplained, (unexplained, {c: res["examples"][c] for c in unexplained})
    assert res["agree"] >= 0.99 * 20000


# ------------------------------------------------------------------------------------------------ scenes
def _nudged(scene, rng, eps=1e-15):
    """the scene with every VELOCITY changed by a relative 1e-15 (one rounding error).  Positions and angles stay: exact zeros in them
    are structural (two feet created at one point with one angle keep the rotary limit joint inactive, quirk C14)"""
    def f(t, lo):
        return tuple(v * (1.0 + eps * rng.uniform(-1.0, 1.0)) if k >= lo else v for k, v in enumerate(t))
    if "cars" in scene:
        return dict(cars=[f(c, 4) for c in scene["cars"]], peds=[f(p, 2) for p in scene["peds"]], obst=scene["obst"])
    return dict(robots=[(f(L, 3), f(R, 3), t) for L, R, t in scene["robots"]], ball=f(scene["ball"], 2))


def classify(scene, dev, world, expected_fn, last):
    """a scene's deviation (oracle or HIP against kat_general) -> 'agree' | a catalogued class | 'UNEXPLAINED'"""
    if dev <= 1e-9:
        return "agree"
    if world.stats.get("cores_cross", 0):
        return "capsule cores cross"
    e1 = expected_fn(scene)[0]
    e2 = expected_fn(_nudged(scene, np.random.default_rng(1)))[0]
    own = kw.deviation(list(e1[last]), list(e2[last]))
    return "ill-conditioned scene" if dev <= 1e3 * own else "UNEXPLAINED"


def driving_scenes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        sc = kw.driving_scene(rng)
        exp, world, ok = kw.driving_expected(sc)
        if ok:
            out.append((sc, exp, world))
    return out


def robocup_scenes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        sc = kw.robocup_scene(rng)
        if not kw.valid_robocup_start(sc):
            continue
        exp, world, ok = kw.robocup_expected(sc)
        if ok:
            out.append((sc, exp, world))
    return out


def run_driving_on(sim_set, sim_step, sim_get, scenes, template):
    """set every scene on environment i, three env steps, -> deviations [n] from kat_general (largest over the three read-backs)"""
    for i, (sc, _, _) in enumerate(scenes):
        sim_set(i, kw.driving_state(template, sc))
    dev = np.zeros(len(scenes))
    for s in range(3):
        sim_step()
        for i, (_, exp, _) in enumerate(scenes):
            dev[i] = max(dev[i], kw.deviation(list(kw.driving_readback(sim_get(i))), list(exp[s])))
    return dev


def run_robocup_on(sim_set, sim_step, sim_get, scenes, template):
    for i, (sc, _, _) in enumerate(scenes):
        sim_set(i, kw.robocup_state(template, sc))
    sim_step()
    return np.array([kw.deviation(list(kw.robocup_readback(sim_get(i))), list(exp[0])) for i, (_, exp, _) in enumerate(scenes)])


def tally(scenes, dev, expected_fn, last):
    classes = {}
    for (sc, _, world), d in zip(scenes, dev):
        c = classify(sc, d, world, expected_fn, last)
        classes.setdefault(c, []).append(float(d))
    return classes


def events(scenes, what):
    return sum(sum(1 for ev in w.log if ev[1] == what) for _, _, w in scenes)


def test_driving_scenes_full_state_against_the_oracle():
    n = 2000
    scenes = driving_scenes(n, 7)
    ora = ol.OracleEnv(env_type=1, num_envs=n, n_players=10, seed=5, threads=8)
    ora.reset()
    acts = np.ones((n, 10, 2), np.int32)
    dev = run_driving_on(ora.set_state, lambda: ora.step(acts), ora.get_state, scenes, ora.get_state(0))
    assert ora.overflow() == 0
    classes = tally(scenes, dev, kw.driving_expected, 2)
    assert "UNEXPLAINED" not in classes, classes["UNEXPLAINED"]
    assert len(classes["agree"]) >= 0.995 * n, {k: len(v) for k, v in classes.items()}
    # what the population exercised: first touches, separations, re-touches inside collision_persistence, EPA on rotated boxes
    assert events(scenes, "begin") > 3 * n and events(scenes, "separate") > n and events(scenes, "retouch") > 300
    assert sum(w.stats.get("epa_calls", 0) for _, _, w in scenes) > 20 * n


def test_robocup_scenes_full_state_against_the_oracle():
    n = 500
    scenes = robocup_scenes(n, 11)
    ora = ol.OracleEnv(env_type=0, num_envs=n, n_players=5, seed=3, flags=ol.FLAG_USE_OBS_REWARDS, threads=8)   # canFall off
    ora.reset()
    a = np.zeros((n, 10, 4), np.int32)
    a[..., 3] = 3                                              # head action 3 = none
    dev = run_robocup_on(ora.set_state, lambda: ora.step(a), ora.get_state, scenes, ora.get_state(0))
    assert ora.overflow() == 0
    classes = tally(scenes, dev, lambda sc: kw.robocup_expected(sc), 0)
    assert "UNEXPLAINED" not in classes, classes["UNEXPLAINED"]
    assert len(classes["agree"]) >= 0.96 * n, {k: len(v) for k, v in classes.items()}
    assert events(scenes, "begin") > 2 * n and events(scenes, "separate") > n and events(scenes, "retouch") > 100
    # error bit 4 stays down (the population's one "capsule cores cross" scene is kat_general's EPA answering -20 for the two exactly
    # parallel feet of ONE robot, 20 px apart - a degenerate Minkowski difference; the cores do not touch)
    assert ora.degenerate() == 0


def _feet_scene(ora, env_idx, b_angle, dx, dy):
    """robot 5's feet at robot 0's left-foot position + (dx, dy), turned to b_angle; robot 0 at angle 0 -> the state blob"""
    st = ora.get_state(env_idx)
    A, B = st.robots[0], st.robots[5]
    for f in ("l", "r"):
        setattr(A, f + "a", 0.0)
        setattr(B, f + "a", b_angle)
        setattr(B, f + "px", A.lpx + dx)
        setattr(B, f + "py", A.lpy + dy)
    return st


def test_crossed_capsule_cores_get_chipmunks_normal_and_degenerate_ones_are_reported():
    """VERDICT r5 item 5, and what it turned up: a foot laid ACROSS another foot has no closest-point normal (the crossing point is computed
    twice: d = 0 or rounding noise).  Chipmunk's EPA gives the minimum-translation axis there; so do the oracle and the kernels since
    round 6 (oracle/cp_lite.c cores_crossing_normal) - the capsule-capsule fuzz has no "cores cross" class any more - and what is left
    for error bit 4 is the measure-zero case in which even that normal's sign is a convention: cores exactly collinear / exactly touching.
    Sticky until reset / set_state."""
    import math
    ora = ol.OracleEnv(env_type=0, num_envs=4, n_players=5, seed=3, flags=ol.FLAG_USE_OBS_REWARDS, threads=1)
    ora.reset()
    a = np.zeros((4, 10, 4), np.int32)
    a[..., 3] = 3
    # 1: B turned by 90 degrees, its left core (the vertical segment x = Bx - 10, By - 10 .. By + 10) crossing A's left core (y = Ay + 10) 1 px
    #    from B's lower end: the minimum translation pushes B up by ~1 px, and after the step B has moved UP, not sideways
    ora.set_state(1, _feet_scene(ora, 1, math.pi / 2, 10.0, 19.0))
    # 2: B beside A, same angle, B's right core 1e-3 px above A's left core (capsules 15 px deep): close, not crossing, nothing to report
    ora.set_state(2, _feet_scene(ora, 2, 0.0, 0.0, 20.0 + 1e-3))
    # 3: B's right core exactly ON A's left core (collinear, overlapping): every candidate distance is 0 - reported
    ora.set_state(3, _feet_scene(ora, 3, 0.0, 5.0, 20.0))
    y0 = ora.get_state(1).robots[5].lpy
    l = ol.lib()
    l.oracle_cp_cores_cross.restype = C_long = __import__("ctypes").c_long
    before = l.oracle_cp_cores_cross()
    ora.step(a)
    assert l.oracle_cp_cores_cross() > before
    assert [ora.degenerate_env(i) for i in range(4)] == [0, 0, 0, 16]
    assert ora.get_state(1).robots[5].lpy > y0 + 0.01, "the crossing foot is pushed back along the minimum-translation axis (up)"
    assert ora.degenerate() == 16
    ora.set_state(3, ora.get_state(0))
    assert ora.degenerate() == 0


def test_scene_outcomes_do_not_depend_on_the_gjk_warm_start():
    """Chipmunk starts a pair's GJK from the closest features of its previous step (the collision id cached in the BB-tree pair);
    kat_general does the same by default.  Started cold in every step the scenes must come out the same (to 1e-9, ill-conditioned
    scenes aside): the cache is an accelerator, not a semantic."""
    scenes = driving_scenes(300, 5)
    dev = np.array([kw.deviation(list(exp[2]), list(kw.driving_expected(sc, warm_gjk=False)[0][2])) for sc, exp, _ in scenes])
    assert (dev <= 1e-9).mean() >= 0.99 and np.median(dev) < 1e-12, (np.sort(dev)[-5:],)
    scenes = robocup_scenes(100, 5)
    dev = np.array([kw.deviation(list(exp[0]), list(kw.robocup_expected(sc, warm_gjk=False)[0][0])) for sc, exp, _ in scenes])
    assert (dev <= 1e-9).mean() >= 0.95 and np.median(dev) < 1e-12, (np.sort(dev)[-5:],)


def test_capsule_cores_cross_rarely_in_play_and_never_degenerately():
    """Round 5 claimed that capsule cores never cross in play - its counter waited for a closest distance of exactly 0, and a crossing comes
    out as ~1e-15.  They do (the two feet of a robot that has been knocked about: 1 environment in 4096 per episode); since round 6 those
    calls take Chipmunk's minimum-translation normal.  What stays impossible in play is the DEGENERATE case (error bit 4)."""
    import ctypes as C
    l = ol.lib()
    l.oracle_cp_cores_cross.restype = C.c_long
    before = l.oracle_cp_cores_cross()
    E = 256
    env = ol.OracleEnv(env_type=0, num_envs=E, n_players=5, seed=42, flags=ol.ROBOCUP_DEFAULT_FLAGS, threads=8)
    env.reset()
    rng = np.random.default_rng(0)
    contacts = 0
    for s in range(240):
        env.step_noobs(np.stack([rng.integers(0, k, (E, 10)) for k in (5, 3, 3, 7)], -1).astype(np.int32))
        contacts += sum(env.active_contacts(e) for e in range(0, E, 16))
    assert contacts > 50, "the episode must contain contacts"
    assert l.oracle_cp_cores_cross() - before < 200 and env.degenerate() == 0


def test_wrong_readings_of_the_recontact_rule_would_be_caught():
    """Chipmunk keeps a separated pair's arbiter - contacts and accumulated impulses included - for collision_persistence = 3 steps.
    On a re-touch inside that window `begin` fires again, the old impulses ARE carried over by contact id, and
    cpArbiterApplyCachedImpulse is skipped (the arbiter is in its first-contact state).  Two plausible wrong readings - forget the
    impulses at `separate`; apply them on the re-touch - move the outcome of every scene whose re-touching contact still carried an
    impulse (a pair usually parts BECAUSE its impulse has gone to zero; with a third body pushing it does not) far beyond the 1e-9 at
    which the oracle agrees with the right reading (test_driving_scenes_...; K5 freezes one such scene)."""
    scenes = [s for s in driving_scenes(600, 33) if any(ev[1] == "retouch" for ev in s[2].log)]
    assert len(scenes) > 50
    for variant in (dict(drop_cache_on_separate=True), dict(apply_cached_on_retouch=True)):
        moved = sum(kw.deviation(list(exp[2]), list(kw.driving_expected(sc, **variant)[0][2])) > 1e-6 for sc, exp, _ in scenes)
        assert moved >= 5, (variant, moved, len(scenes))


# K5: frozen from the fuzz population (driving_scene(default_rng(33)), scene 31): cars 0 and 1 touch in substep 10, car 2 hits car 0
# in substep 12, cars 0 | 1 part in substep 14 and touch again in substep 15 - inside the persistence window, with a third body.
K5_SCENE = {'cars': [(1, 1544.4120503260542, 470.4683168673731, 1.5093479788450166, 101.24452194237047, 27.335516176848227, 1.6293065599236982),
                     (0, 1569.443454243659, 485.0360461127719, 0.9842063131329395, -0.0, 0.0, 0.0),
                     (1, 1519.1451356605219, 488.99995923270654, 3.135232723450839, 119.09322118218591, -23.052525498809487, 0.0),
                     (1, 1596.7966454338477, 493.53427924740305, -2.2774933724830184, -156.75509115512293, 7.916301057588406, 1.369630347824736),
                     (0, 892.5, 60.0, 1.5707963267948966, 0.0, 0.0, 0.0), (2, 892.5, 135.0, 1.5707963267948966, 0.0, 0.0, 0.0),
                     (1, 892.5, 210.0, 1.5707963267948966, 0.0, 0.0, 0.0), (2, 892.5, 285.0, 1.5707963267948966, 0.0, 0.0, 0.0),
                     (2, 892.5, 360.0, 1.5707963267948966, 0.0, 0.0, 0.0), (3, 892.5, 640.0, 1.5707963267948966, 0.0, 0.0, 0.0)],
            'peds': [(1493.3940026554483, 459.99519469552894, 9.321199999357635, 10.489392815254257)],
            'obst': [(1610.436704345423, 458.56864823223157)]}
K5_LOG = [(5, 'begin', 1, 3), (10, 'begin', 0, 1), (12, 'begin', 0, 2), (14, 'separate', 0, 1), (15, 'retouch', 0, 1), (15, 'begin', 0, 1),
          (28, 'separate', 0, 2)]


def k5_check(sim_set, sim_step, sim_get, template):
    exp, world, ok = kw.driving_expected(K5_SCENE)
    assert ok and world.log == K5_LOG, world.log
    sim_set(0, kw.driving_state(template, K5_SCENE))
    for s in range(3):
        sim_step()
        assert kw.deviation(list(kw.driving_readback(sim_get(0))), list(exp[s])) < 1e-10, s
    got = kw.driving_readback(sim_get(0))
    for variant in (dict(drop_cache_on_separate=True), dict(apply_cached_on_retouch=True)):
        wrong = kw.driving_expected(K5_SCENE, **variant)[0][2]
        assert kw.deviation(list(got), list(wrong)) > 1e-3, (variant, "would have gone unnoticed")


def test_k5_recontact_inside_the_persistence_window_with_a_third_body():
    ora = ol.OracleEnv(env_type=1, num_envs=1, n_players=10, seed=5)
    ora.reset()
    acts = np.ones((1, 10, 2), np.int32)
    k5_check(ora.set_state, lambda: ora.step(acts), ora.get_state, ora.get_state(0))


def k6_scene(template):
    """K6: a crashed car (type 0) slides at 30 px/s into a LIVE pedestrian standing on the walkway strip.  Chipmunk orders an arbiter's
    shapes by shape type (circle before polygon: the pedestrian first) and swaps them for the callback when the first shape's
    collision type is not the handler's first declared type - `arbiter.shapes[0]` is the CAR in pedHit (handler (Car, Pedestrian),
    DrivingEnvironment.py:69,643-644).  Read the other way round, pedHit would take the pedestrian's velocity (0) for the car's, return
    False, and the pair would be ignored: the pedestrian would stay alive and unmoved."""
    st = ol.DrivingState.from_buffer_copy(template)
    st.elapsed, st.all_finished, st.n_peds, st.n_obst = 0, 0, 1, 0
    for k, y in zip(range(st.n_cars), (60.0, 135.0, 210.0, 285.0, 360.0, 640.0, 715.0, 790.0, 865.0, 940.0)):   # parked on the vertical road
        c = st.cars[k]
        c.px, c.py, c.vx, c.vy, c.angle, c.w = 892.5, y, 0.0, 0.0, math.pi / 2, 0.0
        c.dirx, c.diry, c.prevx, c.prevy = 0.0, 1.0, c.px, c.py
        c.type, c.finished, c.crashed, c.fric, c.lane_pos = 0, 1, 1, 1, 4
    c = st.cars[0]
    c.px, c.py, c.vx, c.vy, c.angle, c.dirx, c.diry = 300.0, 445.0, 30.0, 0.0, 0.0, 1.0, 0.0
    c.prevx, c.prevy = c.px, c.py
    p = st.peds[0]
    p.px, p.py, p.vx, p.vy = 316.0, 446.0, 0.0, 0.0            # the car's front face (x = 310) reaches the circle (r = 5) in 4 substeps
    p.road, p.side, p.dead, p.moving, p.speed, p.crossing, p.begin_crossing = 1, 0, 0, 100000, 4, 0, 0
    return st


def k6_check(sim_set, sim_step, sim_get, template):
    sim_set(0, k6_scene(template))
    sim_step()
    g = sim_get(0)
    car, ped = g.cars[0], g.peds[0]
    assert ped.dead == 1 and ped.moving == 0, "pedHit must have been handed the car first"
    assert ped.vx > 1.0 and car.vx < 29.0 and car.crashed == 1 and car.type == 0, (ped.vx, car.vx)
    # the same collision in the independent restatement (the pedestrian's velocity is zeroed by die() at `begin`, then the solve)
    w = kg.DrivingWorld([(0, 300.0, 445.0, 0.0, 30.0, 0.0, 0.0)], [(316.0, 446.0, 0.0, 0.0)], [])
    w.peds[0].fric = None                                      # alive until the hit (cpBodyUpdateVelocity: nothing); die() installs the friction
    for _ in range(10):
        w.step()
    assert w.peds[0].fric == kg.FRIC_PED_DEAD
    np.testing.assert_allclose([car.px, car.vx, car.w, ped.px, ped.py, ped.vx, ped.vy],
                               [w.cars[0].px, w.cars[0].vx, w.cars[0].w, w.peds[0].px, w.peds[0].py, w.peds[0].vx, w.peds[0].vy], rtol=1e-11, atol=1e-11)


def test_k6_handler_argument_order_the_car_comes_first():
    ora = ol.OracleEnv(env_type=1, num_envs=1, n_players=10, seed=5)
    ora.reset()
    acts = np.ones((1, 10, 2), np.int32)
    k6_check(ora.set_state, lambda: ora.step(acts), ora.get_state, ora.get_state(0))
