"""The Driving oracle against round-2 fixtures from the reference's own Python (tests/golden/gen_golden_driving_r2.py):
the `begin` collision callbacks carCrash / pedHit / carHit (DrivingEnvironment.py:591-683) and the reset composition
(DrivingEnvironment.__init__ / _setup_scene :58-115, :527-584) under the oracle's Philox draws."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _state(z, pre, t=None, episode=1):
    g = (lambda k: z[pre + k][t]) if t is not None else (lambda k: z[pre + k])
    cf, ci, pf, pi, ob, sc = g("cars_f"), g("cars_i"), g("peds_f"), g("peds_i"), g("obst"), g("scalars")
    st = ol.DrivingState()
    st.elapsed, st.all_finished = int(sc[0]), int(sc[1])
    st.n_cars, st.n_peds, st.n_obst, st.episode = len(cf), len(pf), len(ob), episode
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
    return st


def test_begin_callbacks_car_crash_ped_hit_car_hit(oracle_built):
    z = np.load(os.path.join(G, "driving_callbacks.npz"))
    env = ol.OracleEnv(num_envs=1, n_players=10)
    env.reset()
    n = len(z["cb_kind"])
    seen = {0: [0, 0], 1: [0, 0], 2: [0, 0]}
    for t in range(n):
        env.set_state(0, _state(z, "cb_b_", t))
        rew = np.zeros(10)
        sa, sb = [int(x) for x in z["cb_slots"][t]]
        ret = env.l.oracle_drv_begin(env.h, 0, sa, sb, rew.ctypes.data_as(C.c_void_p))
        msg = "trial %d kind %d slots %d,%d" % (t, int(z["cb_kind"][t]), sa, sb)
        assert ret == int(z["cb_ret"][t]), msg
        np.testing.assert_allclose(rew, z["cb_rew"][t], rtol=1e-12, atol=1e-13, err_msg=msg)
        got = ol.state_to_dict(env.get_state(0))
        np.testing.assert_array_equal(got["cars_i"], z["cb_a_cars_i"][t], err_msg=msg)   # finished, crashed, friction
        np.testing.assert_array_equal(got["peds_i"], z["cb_a_peds_i"][t], err_msg=msg)   # dead, moving
        np.testing.assert_array_equal(got["cars_f"], z["cb_a_cars_f"][t], err_msg=msg)   # untouched by callbacks
        np.testing.assert_array_equal(got["peds_f"], z["cb_a_peds_f"][t], err_msg=msg)   # die() zeroes the velocity
        seen[int(z["cb_kind"][t])][int(np.any(z["cb_rew"][t] != 0.0))] += 1
    for k, (zero, nonzero) in seen.items():
        assert zero > 5 and nonzero > 5, "callback kind %d: both the punished and the unpunished branch must occur" % k


@pytest.mark.parametrize("case", range(6))
def test_reset_composition(oracle_built, case):
    z = np.load(os.path.join(G, "driving_reset.npz"))
    A, seed, genv, episode = [int(x) for x in z["reset_keys"][case]]
    env = ol.OracleEnv(num_envs=1, n_players=A, seed=seed, env_id_offset=genv)
    for _ in range(episode + 1):  # the k-th reset of a handle draws with episode word k - 1
        obs = env.reset()
    st = env.get_state(0)
    assert st.episode == episode + 1
    got = ol.state_to_dict(st)
    pre = "reset%d_" % case
    n_ped, n_obst_raw, n_obst = [int(x) for x in z[pre + "counts"]]
    assert [st.n_cars, st.n_peds, st.n_obst] == [A, n_ped, n_obst]
    assert n_obst <= n_obst_raw
    np.testing.assert_allclose(got["cars_f"], z[pre + "cars_f"], rtol=1e-13, atol=1e-11)
    np.testing.assert_array_equal(got["cars_i"], z[pre + "cars_i"])
    np.testing.assert_allclose(got["peds_f"], z[pre + "peds_f"], rtol=1e-13, atol=1e-11)
    np.testing.assert_array_equal(got["peds_i"], z[pre + "peds_i"])
    np.testing.assert_allclose(got["obst"].T, z[pre + "obst"], rtol=1e-13, atol=1e-11)
    np.testing.assert_allclose(obs[0, 0], z[pre + "obs"], rtol=0, atol=2e-6)
