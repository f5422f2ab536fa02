"""The oracle (oracle/*.c) against fixtures computed by the reference's own Python (tests/golden/gen_golden.py).

Pins: cutils.apply_friction, Road geometry/classification/spots, Car.accelerate/turn via processAction,
DrivingEnvironment.tick / move / getFullState and the composition of step() in collision-free windows.
Tolerances: the reference used glibc sin/cos/atan2, the oracle the deterministic header (<=1 ulp apart), so fp64
values agree to ~1e-12 relative; flags and integer state must be exact; f32 observations to 1e-6.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-11
ATOL = 1e-11


@pytest.fixture(scope="module")
def pure(oracle_built):
    with open(os.path.join(G, "driving_pure.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def unit(oracle_built):
    return np.load(os.path.join(G, "driving_unit.npz"))


def test_friction_all_coefficient_sets(pure):
    l = ol.lib()
    for rec in pure["friction"]:
        v = np.array(rec["vin"], np.float64)
        l.oracle_apply_friction(float(rec["m"]), v.ctypes.data_as(C.c_void_p), rec["mu"], rec["mur"], rec["spin"])
        np.testing.assert_allclose(v, rec["vout"], rtol=1e-14, atol=0)


def test_road_geometry(pure):
    l = ol.lib()
    for ri, g in enumerate(pure["road_geometry"]):
        out = np.zeros(6 + 20 + 8)
        l.oracle_road_geometry(ri, out.ctypes.data_as(C.c_void_p))
        nl = len(g["lanes"])
        exp = g["dir"] + g["normal"] + [g["length"], g["dirAngle"]]
        np.testing.assert_allclose(out[:6], exp, rtol=1e-15, atol=1e-30)
        np.testing.assert_allclose(out[6:6 + 4 * nl].reshape(nl, 4), g["lanes"], rtol=1e-15, atol=1e-13)
        np.testing.assert_allclose(out[26:34].reshape(2, 4), g["walk"], rtol=1e-15, atol=1e-13)


def test_is_point_on_road(pure):
    l = ol.lib()
    bad = 0
    for r in pure["is_point_on_road"]:
        bad += l.oracle_road_is_point_on_road(r["road"], r["x"], r["y"], r["angle"]) != r["pos"]
    assert bad == 0


def test_get_spot_and_walk_spot(pure):
    l = ol.lib()
    for r in pure["get_spot"]:
        out = np.zeros(3)
        l.oracle_road_get_spot(r["road"], r["lane"], r["spot"], out.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(out[:2], r["pos"], rtol=1e-14, atol=1e-12)
        assert out[2] == r["angle"]  # exact: these four angles seed every car's heading
    for r in pure["get_walk_spot"]:
        out = np.zeros(2)
        l.oracle_road_get_walk_spot(r["road"], r["side"], r["length"], r["width"], out.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(out, r["pos"], rtol=1e-14, atol=1e-12)


def test_moments(pure):
    l = ol.lib()
    for r in pure["moments"]["box"]:
        assert l.oracle_moment_for_box(float(r["m"]), float(r["h"]), float(r["w"])) == r["I"]
    assert l.oracle_moment_for_circle(90.0, 0.0, 5.0) == 90.0 * 0.5 * 25.0
    # moment_for_segment(m,a,b,r) = m((L+2r)^2+4r^2)/12 + m|mid|^2  
    got = l.oracle_moment_for_segment(4000.0, -10.0, 10.0, 10.0, 10.0, 7.5)
    assert got == 4000.0 * (((20.0 + 15.0) ** 2 + 4 * 7.5 ** 2) / 12.0 + 100.0)


def _blank_state(n_cars=10, n_peds=0, n_obst=0):
    st = ol.DrivingState()
    st.n_cars, st.n_peds, st.n_obst, st.episode = n_cars, n_peds, n_obst, 1
    for i in range(n_cars):
        st.cars[i].px = 100.0 + 150 * i
        st.cars[i].py = 100.0
        st.cars[i].dirx = 1.0
        st.cars[i].lane_pos = 4
    return st


def test_process_action_all_branches(unit):
    env = ol.OracleEnv(num_envs=1, n_players=10)
    env.reset()
    pin, act, pout = unit["pa_in"], unit["pa_act"], unit["pa_out"]
    for i in range(len(pin)):
        st = _blank_state()
        c = st.cars[0]
        c.px, c.py, c.vx, c.vy, c.angle, c.w, c.dirx, c.diry = pin[i][:8]
        c.type, c.finished = int(pin[i][8]), int(pin[i][9])
        c.prevx, c.prevy = c.px, c.py
        env.set_state(0, st)
        a = np.array(act[i], np.int32)
        env.l.oracle_drv_process_action(env.h, 0, 0, a.ctypes.data_as(C.c_void_p))
        o = env.get_state(0).cars[0]
        np.testing.assert_allclose([o.vx, o.vy, o.angle, o.dirx, o.diry], pout[i], rtol=RTOL, atol=ATOL)


def test_tick(unit):
    env = ol.OracleEnv(num_envs=1, n_players=10)
    env.reset()
    tin, tout = unit["tick_in"], unit["tick_out"]
    for i in range(len(tin)):
        r = tin[i]
        idx = int(r[12])
        st = _blank_state()
        c = st.cars[idx]
        c.px, c.py, c.vx, c.vy, c.angle, c.prevx, c.prevy, c.goalx, c.goaly = r[:9]
        c.finished, c.crashed = int(r[9]), int(r[10])
        st.elapsed = int(r[11])
        env.set_state(0, st)
        rew = env.l.oracle_drv_tick(env.h, 0, idx)
        prew = env.l.oracle_drv_pos_reward(env.h, 0, idx)
        o = env.get_state(0).cars[idx]
        e = tout[i]
        np.testing.assert_allclose([rew, prew], e[:2], rtol=1e-10, atol=1e-12, err_msg="trial %d" % i)
        assert [o.lane_pos, o.finished, o.crashed, o.fric] == [int(e[2]), int(e[3]), int(e[4]), int(e[9])], i
        np.testing.assert_allclose([o.vx, o.vy, o.prevx, o.prevy], e[5:9], rtol=RTOL, atol=ATOL)


def test_move_pedestrian_fsm(unit):
    seed, genv, episode = [int(x) for x in unit["move_key"]]
    env = ol.OracleEnv(num_envs=1, n_players=10, seed=seed, env_id_offset=genv)
    env.reset()
    min_, mout = unit["move_in"], unit["move_out"]
    for i in range(len(min_)):
        r = min_[i]
        k = int(r[12])
        st = _blank_state(n_peds=20)
        st.episode = episode
        p = st.peds[k]
        p.px, p.py, p.vx, p.vy = r[:4]
        p.road, p.side, p.dead, p.moving, p.speed, p.crossing, p.begin_crossing = [int(x) for x in r[4:11]]
        st.elapsed = int(r[11])
        env.set_state(0, st)
        env.l.oracle_drv_move(env.h, 0, k)
        o = env.get_state(0).peds[k]
        e = mout[i]
        np.testing.assert_allclose([o.vx, o.vy], e[:2], rtol=RTOL, atol=ATOL, err_msg="trial %d" % i)
        assert [o.side, o.moving, o.crossing, o.begin_crossing] == [int(x) for x in e[2:6]], i


def _state_from_npz(z, tag, which, key):
    cf, ci = z["%s_%s_cars_f" % (tag, which)], z["%s_%s_cars_i" % (tag, which)]
    pf, pi = z["%s_%s_peds_f" % (tag, which)], z["%s_%s_peds_i" % (tag, which)]
    ob, sc = z["%s_%s_obst" % (tag, which)], z["%s_%s_scalars" % (tag, which)]
    st = ol.DrivingState()
    st.elapsed, st.all_finished = int(sc[0]), int(sc[1])
    st.n_cars, st.n_peds, st.n_obst, st.episode = len(cf), len(pf), len(ob), int(key[2])
    for i in range(len(cf)):
        for n, v in zip(ol.CAR_F, cf[i]):
            setattr(st.cars[i], n, float(v))
        for n, v in zip(ol.CAR_I, ci[i]):
            setattr(st.cars[i], n, int(v))
    for i in range(len(pf)):
        for n, v in zip(ol.PED_F, pf[i]):
            setattr(st.peds[i], n, float(v))
        for n, v in zip(ol.PED_I, pi[i]):
            setattr(st.peds[i], n, int(v))
    for i in range(len(ob)):
        st.obst_x[i], st.obst_y[i] = float(ob[i][0]), float(ob[i][1])
    er, epr = z["%s_%s_episode_r" % (tag, which)], z["%s_%s_episode_pos_r" % (tag, which)]
    for i in range(len(er)):
        st.episode_r[i], st.episode_pos_r[i] = float(er[i]), float(epr[i])
    return st


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_step_composition_free_flight(oracle_built, tag):
    """Reference DrivingEnvironment.step() (processAction + tick + move + integrate + friction + rewards + Full obs)
    on scenes built by the reference's own constructor, in windows without contact."""
    z = np.load(os.path.join(G, "driving_freeflight.npz"))
    key = z["%s_key" % tag]
    st0 = _state_from_npz(z, tag, "init", key)
    env = ol.OracleEnv(num_envs=1, n_players=st0.n_cars, seed=int(key[0]), env_id_offset=int(key[1]))
    env.reset()
    env.set_state(0, st0)
    obs0 = np.zeros((1, 1, env.A, env.D), np.float32)
    env.l.oracle_reset  # (not called: the scene comes from the reference)
    acts = z["%s_actions" % tag]
    max_active = 0
    for s in range(len(acts)):
        o, r, d = env.step(acts[s][None].astype(np.int32))
        max_active = max(max_active, env.active_contacts(0))
        np.testing.assert_allclose(r[0], z["%s_rewards" % tag][s], rtol=1e-9, atol=1e-11, err_msg="step %d" % s)
        assert int(d[0]) == int(z["%s_dones" % tag][s])
        np.testing.assert_allclose(o[0, 0], z["%s_obs" % tag][s], rtol=0, atol=2e-6, err_msg="step %d" % s)
    assert max_active == 0, "golden window is only valid without contacts"
    got = ol.state_to_dict(env.get_state(0))
    np.testing.assert_allclose(got["cars_f"], z["%s_final_cars_f" % tag], rtol=1e-10, atol=1e-10)
    np.testing.assert_array_equal(got["cars_i"], z["%s_final_cars_i" % tag])
    np.testing.assert_allclose(got["peds_f"], z["%s_final_peds_f" % tag], rtol=1e-10, atol=1e-10)
    np.testing.assert_array_equal(got["peds_i"], z["%s_final_peds_i" % tag])
    np.testing.assert_allclose(got["episode_r"], z["%s_final_episode_r" % tag], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(got["episode_pos_r"], z["%s_final_episode_pos_r" % tag], rtol=1e-9, atol=1e-11)
