"""The RoboCup oracle against round-2 fixtures from the reference's own Python (tests/golden/gen_golden_robocup_r2.py):
robotCollision / separate / goalpostCollision (RoboCupEnvironment.py:1038-1125), fall (:735-791), the step() composition
(:446-524) on a free-flight Space, and the reset composition (:239-336) incl. randomInit / deterministicTurn."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_golden_robocup import _check, _to_state

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cb(oracle_built):
    return np.load(os.path.join(G, "robocup_callbacks.npz"))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _run_cb(z, tag, flags_of, call, vel=True):
    key = [int(x) for x in z["key"]]
    envs = {}
    n = len(z[tag + "_rew"])
    for t in range(n):
        extra = z[tag + "_extra"][t]
        flags = flags_of(extra)
        if flags not in envs:
            envs[flags] = ol.OracleEnv(env_type=0, num_envs=1, n_players=5, seed=key[0], env_id_offset=key[1], flags=flags)
            envs[flags].reset()
        env = envs[flags]
        env.set_state(0, _to_state(z[tag + "_b_rf"][t], z[tag + "_b_ri"][t], z[tag + "_b_sc"][t], z[tag + "_b_fl"][t], key[2]))
        rew = np.zeros(22)
        call(env, extra, rew)
        if vel:  # the generator ran every body's velocity function once (forces of fall() -> velocities)
            env.l.oracle_velocity_update(env.h, 0)
        msg = "%s trial %d" % (tag, t)
        _check(env.get_state(0), z[tag + "_a_rf"][t], z[tag + "_a_ri"][t], z[tag + "_a_sc"][t], z[tag + "_a_fl"][t], msg)
        np.testing.assert_allclose(rew, z[tag + "_rew"][t], rtol=1e-10, atol=1e-12, err_msg=msg)


def test_robot_collision_post_solve(cb):
    """touch counters, the two fall dice (r > thresh ** touchCntr), fall() with its pushes, the pushing penalty"""
    def call(env, extra, rew):
        env.l.oracle_rc_callback(env.h, 0, int(extra[0]), int(extra[1]), 1, _vp(rew))
    _run_cb(cb, "rcol", lambda x: ol.FLAG_CAN_FALL if int(x[2]) else 0, call)
    fallen_before = cb["rcol_b_ri"][:, :, 5].sum(1)
    fallen_after = cb["rcol_a_ri"][:, :, 5].sum(1)
    assert (fallen_after > fallen_before).sum() > 20, "the fall branch of robotCollision must be exercised"
    assert (cb["rcol_extra"][:, 3] == 2).sum() > 50 and (cb["rcol_extra"][:, 3] == 0).sum() > 10  # dice drawn / early outs


def test_separate(cb):
    def call(env, extra, rew):
        env.l.oracle_rc_callback(env.h, 0, int(extra[0]), int(extra[1]), 2, _vp(rew))
    _run_cb(cb, "sep", lambda x: ol.FLAG_CAN_FALL, call, vel=False)


def test_goalpost_collision_post_solve(cb):
    def call(env, extra, rew):
        env.l.oracle_rc_callback(env.h, 0, int(extra[0]), int(extra[1]), 1, _vp(rew))
    _run_cb(cb, "gpost", lambda x: ol.FLAG_CAN_FALL if int(x[2]) else 0, call)
    fallen_before = cb["gpost_b_ri"][:, :, 5].sum(1)
    fallen_after = cb["gpost_a_ri"][:, :, 5].sum(1)
    assert (fallen_after > fallen_before).sum() > 10


def test_fall(cb):
    """radius-40 query, push forces on the neighbours' feet and the ball (through one velocity function call), lastKicked /
    free-kick bookkeeping, fall counters, the third fall's penalty"""
    def call(env, extra, rew):
        env.l.oracle_rc_fall(env.h, 0, int(extra[0]), int(extra[1]), _vp(rew))
    _run_cb(cb, "fall", lambda x: ol.FLAG_CAN_FALL, call)
    moved = np.abs(cb["fall_a_rf"][:, :, 2:4] - cb["fall_b_rf"][:, :, 2:4]).max((1, 2)) > 1.0
    assert moved.sum() > 30, "fall() must have pushed neighbours in a good part of the trials"


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "g"])
def test_step_composition_free_flight(oracle_built, tag):
    z = np.load(os.path.join(G, "robocup_step.npz"))
    n, can_fall, seed, genv, episode, det_turn, allow_head = [int(x) for x in z[tag + "_meta"]]
    flags = (ol.FLAG_CAN_FALL if can_fall else 0) | ol.FLAG_USE_OBS_REWARDS | \
            (ol.FLAG_DETERMINISTIC_TURN if det_turn else 0) | (ol.FLAG_ALLOW_HEAD_TURN if allow_head else 0)
    env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, seed=seed, env_id_offset=genv, flags=flags)
    env.reset()
    env.set_state(0, _to_state(z[tag + "_b_rf"], z[tag + "_b_ri"], z[tag + "_b_sc"], z[tag + "_b_fl"], episode))
    env.l.oracle_set_free_flight(env.h, 1)
    acts = z[tag + "_actions"]
    R = 2 * n
    nonzero = 0
    for s in range(len(acts)):
        # with allowHeadTurn the 4th action is the reference's continuous head channel (Box(-3, 3), :339-342)
        o, r, d = env.step(acts[s][None].astype(np.int32), head=acts[s][None, :, 3] if allow_head else None)
        np.testing.assert_allclose(r[0], z[tag + "_rewards"][s], rtol=1e-9, atol=1e-11, err_msg="step %d rewards" % s)
        assert int(d[0]) == int(z[tag + "_dones"][s])
        np.testing.assert_allclose(o[0], z[tag + "_obs"][s][:, :R, :env.D], rtol=0, atol=2e-6, err_msg="step %d obs (5 snapshots)" % s)
        er, ep, _, goals = env.episode_stats()
        np.testing.assert_allclose(np.concatenate([er[0], ep[0]]), z[tag + "_episode"][s], rtol=1e-9, atol=1e-10)
        nonzero += int(np.any(z[tag + "_rewards"][s] != 0.0))
    assert nonzero >= len(acts) // 2 or tag in "efg", "the composition fixture must carry rewards"
    _check(env.get_state(0), z[tag + "_a_rf"], z[tag + "_a_ri"], z[tag + "_a_sc"], z[tag + "_a_fl"], "final state " + tag)
    np.testing.assert_array_equal(goals[0], z[tag + "_goals"])


@pytest.mark.parametrize("case", range(10))
def test_reset_composition(oracle_built, case):
    z = np.load(os.path.join(G, "robocup_reset.npz"))
    n, seed, genv, episode, flags, n_random = [int(x) for x in z["reset_keys"][case]]
    oflags = ol.FLAG_CAN_FALL | (ol.FLAG_RANDOM_INIT if flags & 1 else 0) | (ol.FLAG_DETERMINISTIC_TURN if flags & 2 else 0)
    env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, seed=seed, env_id_offset=genv, flags=oflags)
    for _ in range(episode + 1):
        obs = env.reset()
    pre = "reset%d_" % case
    sc = z[pre + "sc"].copy()
    _check(env.get_state(0), z[pre + "rf"], z[pre + "ri"], sc, z[pre + "fl"], "reset case %d" % case)
    assert n_random in (18, 23, 24)  # default: 18 draws; randomInit: 20 spots + 2 ball + ownership (+1 if owned)
    for t in range(5):  # environment_base.py:217-222: nTimeSteps copies of the first observation
        np.testing.assert_allclose(obs[0, t], z[pre + "obs"][:2 * n, :env.D], rtol=0, atol=2e-6)
