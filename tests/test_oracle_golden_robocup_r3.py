"""RoboCup info['Full State'] = getFullState(agent=None) (RoboCupEnvironment.py:511, :1149-1161): the oracle's restatement
against arrays the reference's OWN method returned on scripted states (tests/golden/gen_golden_robocup_r3.py ->
robocup_fullstate.npz), and the reference trainer's expression over it (models/train.py:272)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_golden_robocup import _to_state

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def z(oracle_built):
    return np.load(os.path.join(HERE, "golden", "robocup_fullstate.npz"))


def test_global_state_matches_the_references_getFullState_none(z):
    assert len(z["fs_robots"]) >= 30
    seen_down = seen_owned = 0
    for t in range(len(z["fs_robots"])):
        n, R = int(z["fs_extra"][t][0]), int(z["fs_extra"][t][1])
        env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, seed=1, flags=0)
        env.reset()
        env.set_state(0, _to_state(z["fs_b_rf"][t], z["fs_b_ri"][t], z["fs_b_sc"][t], z["fs_b_fl"][t], 1))
        out = np.zeros(R * 6 + 3, np.float32)
        env.l.oracle_rc_global_state(env.h, 0, out.ctypes.data_as(C.c_void_p))
        robots, ball = out[:R * 6].reshape(R, 6), out[R * 6:]
        # positions, team, flags: the same IEEE operations -> exact; cos / sin: dynenv_math.h against libm (<= 1 ulp in fp64)
        np.testing.assert_array_equal(robots[:, [0, 1, 4, 5]], z["fs_robots"][t][:R][:, [0, 1, 4, 5]], err_msg="trial %d" % t)
        np.testing.assert_allclose(robots[:, 2:4], z["fs_robots"][t][:R][:, 2:4], rtol=0, atol=2e-7, err_msg="trial %d" % t)
        np.testing.assert_array_equal(ball, z["fs_ball"][t], err_msg="trial %d" % t)
        # models/train.py:272: agentFinished = [[agent[-1] for agent in s['Full State'][0]] for s in state]
        state = [{"Full State": [robots, ball]}]
        fin = np.array([[agent[-1] for agent in s["Full State"][0]] for s in state], np.float32)
        assert fin.shape == (1, R)
        np.testing.assert_array_equal(fin[0], z["fs_finished"][t][:R])
        seen_down += int(fin.sum() > 0)
        seen_owned += int(ball[2] != 0)
    assert seen_down > 10 and seen_owned > 10
