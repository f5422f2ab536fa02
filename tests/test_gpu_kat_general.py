"""The differential scenes of tests/test_kat_general.py through `dynenv_set_state` / `dynenv_step` / `dynenv_get_state` on the HIP path
(-m gpu): the kernels against tests/kat_general.py - the independent GJK / EPA restatement of `Space.step` - to 1e-9 on the full
kinematic state, with the same rule for scenes above 1e-9 (explained by a measured property of the scene, or the test fails), and
bit-identical to the oracle on the same scenes.  Plus K5 (re-contact inside collision_persistence with a third body) and K6
(handler argument order) through the C ABI."""
import numpy as np
import pytest

import kat_worlds as kw
import oracle_lib as ol
import test_kat_general as tk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(oracle_built):
    import torch
    import dynenv_amd
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return dynenv_amd


def _same_bits(a, b):
    return all(np.array_equal(x.view(np.uint64), y.view(np.uint64)) for x, y in zip(a, b))


def test_driving_scenes_full_state_on_the_hip_path(gpu):
    n = 2000
    scenes = tk.driving_scenes(n, 7)
    env = gpu.BatchedDynEnv(gpu.DynEnvType.DRIVE, n, 10, seed=5)
    env.reset_flat()
    ora = ol.OracleEnv(env_type=1, num_envs=n, n_players=10, seed=5, threads=16)
    ora.reset()
    template = ora.get_state(0)
    acts = np.ones((n, 10, 2), np.int32)
    for i, (sc, _, _) in enumerate(scenes):
        ora.set_state(i, kw.driving_state(template, sc))
    dev = np.zeros(n)
    for i, (sc, _, _) in enumerate(scenes):
        env.set_state(i, kw.driving_state(template, sc))
    for s in range(3):
        env.step_flat(acts, auto_reset=False)
        ora.step(acts)
        for i, (_, exp, _) in enumerate(scenes):
            got = kw.driving_readback(env.get_state(i))
            dev[i] = max(dev[i], kw.deviation(list(got), list(exp[s])))
            assert _same_bits(got, kw.driving_readback(ora.get_state(i))), (i, s, "HIP != oracle")
    assert env.error_flags() == 0
    classes = tk.tally(scenes, dev, kw.driving_expected, 2)
    assert "UNEXPLAINED" not in classes, classes["UNEXPLAINED"]
    assert len(classes["agree"]) >= 0.995 * n, {k: len(v) for k, v in classes.items()}
    env.close()


def test_robocup_scenes_full_state_on_the_hip_path(gpu):
    n = 500
    scenes = tk.robocup_scenes(n, 11)
    flags = ol.FLAG_USE_OBS_REWARDS                              # canFall off: no dice anywhere
    env = gpu.BatchedDynEnv(gpu.DynEnvType.ROBO_CUP, n, 5, seed=3, flags=flags)
    env.reset_flat()
    ora = ol.OracleEnv(env_type=0, num_envs=n, n_players=5, seed=3, flags=flags, threads=16)
    ora.reset()
    template = ora.get_state(0)
    a = np.zeros((n, 10, 4), np.int32)
    a[..., 3] = 3
    for i, (sc, _, _) in enumerate(scenes):
        st = kw.robocup_state(template, sc)
        ora.set_state(i, st)
        env.set_state(i, st)
    env.step_flat(a, auto_reset=False)
    ora.step(a)
    dev = np.zeros(n)
    for i, (_, exp, _) in enumerate(scenes):
        got = kw.robocup_readback(env.get_state(i))
        dev[i] = kw.deviation(list(got), list(exp[0]))
        assert _same_bits(got, kw.robocup_readback(ora.get_state(i))), (i, "HIP != oracle")
    assert env.error_flags() == ora.degenerate() == 0   # (no capsule cores within 1e-6 px of each other: error bit 4 stays down on both sides)
    classes = tk.tally(scenes, dev, lambda sc: kw.robocup_expected(sc), 0)
    assert "UNEXPLAINED" not in classes, classes["UNEXPLAINED"]
    assert len(classes["agree"]) >= 0.96 * n, {k: len(v) for k, v in classes.items()}
    env.close()


def _driving_one(gpu):
    env = gpu.BatchedDynEnv(gpu.DynEnvType.DRIVE, 1, 10, seed=5)
    env.reset_flat()
    acts = np.ones((1, 10, 2), np.int32)
    return env, (lambda: env.step_flat(acts, auto_reset=False))


def test_k5_recontact_inside_the_persistence_window_with_a_third_body_hip(gpu):
    env, step = _driving_one(gpu)
    tk.k5_check(env.set_state, step, env.get_state, env.get_state(0))
    env.close()


def test_k6_handler_argument_order_the_car_comes_first_hip(gpu):
    env, step = _driving_one(gpu)
    tk.k6_check(env.set_state, step, env.get_state, env.get_state(0))
    env.close()
