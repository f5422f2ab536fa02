"""The drop-in boundary on the GPU (round 2): lazy compat observations / infos equal the eager forms element for element,
info['Full State'] with Partial observations equals the oracle's noise-free state rows, RoboCup's class switches
(randomInit, deterministicTurn) and the continuous head channel against the oracle, malformed actions, checkpoint / replay
across an episode end, refresh_obs, the device guard."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(oracle_built):
    import torch
    import dynenv_amd
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return dynenv_amd, torch, oracle_built


def _acts(rng, env):
    E, A = env.num_envs, env.n_agents
    if env.action_dim == 2:
        return rng.integers(0, 3, (E, A, 2)).astype(np.int32)
    return np.stack([rng.integers(0, k, (E, A)) for k in (5, 3, 3, 7)], -1).astype(np.int32)


def _same(a, b):
    if isinstance(a, (list, tuple)):
        assert type(a) is type(b) and len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and a.dtype == b.dtype
        np.testing.assert_array_equal(a, b)
    else:
        assert a == b


CONFIGS = [("DRIVE", 10, {}), ("DRIVE", 10, "partial"), ("ROBO_CUP", 5, {}), ("ROBO_CUP", 5, "partial")]


@pytest.mark.parametrize("kind,n,obs", CONFIGS)
def test_lazy_compat_equals_eager_element_for_element(gpu, kind, n, obs):
    dynenv_amd, torch, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType
    kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3) if obs else {}
    E = 6
    env = dynenv_amd.BatchedDynEnv(getattr(DynEnvType, kind), E, n, seed=5, **kw)
    rng = np.random.default_rng(1)
    lazy = env.reset()
    for s in range(3):
        lazy, rew, dones, infos = env.step(_acts(rng, env))
    eager = env._compat_obs(lazy._dense, lazy._counts)      # the round-1 eager builder on the same host copy
    assert lazy.shape == eager.shape == (E, env.n_time_steps, env.n_agents, 3) and lazy.dtype == object
    full = np.asarray(lazy)
    assert full.dtype == object and full.shape == eager.shape
    for idx in np.ndindex(*eager.shape):
        _same(full[idx], eager[idx])
        _same(lazy[idx], eager[idx])
    # the reference's consumers slice it like this (models/train.py:67-68)
    assert lazy[..., :-1].shape == eager[..., :-1].shape and lazy[..., -1].shape == eager[..., -1].shape
    _same(lazy[..., -1][2, 0, 1], eager[..., -1][2, 0, 1])
    _same(lazy[1:4, :, ::2][1, 0, 1, 1], eager[1:4, :, ::2][1, 0, 1, 1])
    _same(lazy[-1][0][-1][0], eager[-1][0][-1][0])
    assert len(lazy) == E and sum(1 for _ in lazy) == E
    with pytest.raises(IndexError):
        lazy[E]
    # infos: a lazy sequence of dicts with the reference's keys
    assert len(infos) == E and set(infos[0]) >= {"Full State", "Recon States"}
    assert infos[E - 1] is infos[-1]
    env.close()


@pytest.mark.parametrize("kind,n", [("DRIVE", 10), ("DRIVE", 4), ("ROBO_CUP", 5), ("ROBO_CUP", 2)])
def test_full_state_with_partial_observations_is_the_oracles_true_state(gpu, kind, n):
    """info['Full State'] / info['Recon States'] come from getFullState - the noise-free state - in every observation mode
    (DrivingEnvironment.py:306-307, RoboCupEnvironment.py:511-512): the Partial handle's rows equal the Full observation of an
    oracle that runs the same seed and actions."""
    dynenv_amd, torch, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType
    E = 5
    robocup = kind == "ROBO_CUP"
    flags = ol.ROBOCUP_DEFAULT_FLAGS if robocup else 0
    env = dynenv_amd.BatchedDynEnv(getattr(DynEnvType, kind), E, n, seed=9, flags=flags, observationType=ObservationType.PARTIAL,
                                   noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
    ora = ol.OracleEnv(env_type=0 if robocup else 1, num_envs=E, n_players=n, seed=9, flags=flags)  # Full observations
    env.reset()
    ora.reset()
    rng = np.random.default_rng(2)
    A = env.n_agents
    for s in range(4):
        a = _acts(rng, env)
        obs, rew, dones, infos = env.step(a)
        oc, _, _ = ora.step(a)
        np.testing.assert_array_equal(env.full_state_obs().cpu().numpy(), oc[:, -1], err_msg="step %d" % s)
        for e in (0, E - 1):
            recon = infos[e]["Recon States"]
            assert len(recon) == A
            for ag in range(A):
                if robocup:
                    np.testing.assert_array_equal(recon[ag][0], oc[e, -1, ag, 0:4].reshape(1, 4))
                    np.testing.assert_array_equal(recon[ag][1], oc[e, -1, ag, 4:12].reshape(1, 8))
                    np.testing.assert_array_equal(recon[ag][2], oc[e, -1, ag, 12:12 + (A - 1) * 6].reshape(A - 1, 6))
                else:
                    np.testing.assert_array_equal(recon[ag][0], oc[e, -1, ag, 0:9].reshape(1, 9))
                    np.testing.assert_array_equal(recon[ag][1], oc[e, -1, ag, 9:9 + (A - 1) * 7].reshape(A - 1, 7))
            full = infos[e]["Full State"]
            if robocup:  # getFullState(None): [robots [A, 6], ball [3]] (contents: test_robocup_full_state_is_getFullState_none)
                assert len(full) == 2 and full[0].shape == (A, 6) and full[1].shape == (3,)
            else:        # [cars [A, 7], obstacles, pedestrians, lanes]
                assert len(full) == 4 and full[0].shape == (A, 7)
        locs = env.env_method("get_agent_locs", indices=[1])[0]   # the true poses, not the noisy self rows
        for ag in range(A):
            np.testing.assert_array_equal(locs[ag][0], oc[1, -1, ag, 4:10] if robocup else oc[1, -1, ag, 0:4])
    env.close()


@pytest.mark.parametrize("obs", [None, "partial"])
def test_robocup_full_state_is_getFullState_none(gpu, obs):
    """info['Full State'] of RoboCup is getFullState(agent=None) (RoboCupEnvironment.py:511, :1149-1161): [robots [A, 6] in field
    coordinates, ball [3]] - not an agent's 'Recon States' triple - bit-identical to the oracle (which the reference's own
    method pins, tests/test_oracle_golden_robocup_r3.py), and the reference trainer's line over it (models/train.py:272) gives the
    fallen | penalized flag of every robot."""
    dynenv_amd, torch, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType
    E, n = 12, 5
    kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3) if obs else {}
    flags = ol.ROBOCUP_DEFAULT_FLAGS
    env = dynenv_amd.BatchedDynEnv(DynEnvType.ROBO_CUP, E, n, seed=17, flags=flags, **kw)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=n, seed=17, flags=flags)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(4)
    A = env.n_agents
    assert int(env._lib.dynenv_global_state_dim(env._h)) == 6 * A + 3
    downs = 0
    for s in range(40):
        a = _acts(rng, env)
        new_obs, rew, dones, state = env.step(a)
        ora.step(a)
        if s % 4 and s < 36:
            continue
        g = ora.global_state()
        np.testing.assert_array_equal(env.global_state().cpu().numpy(), g, err_msg="step %d" % s)
        for e in (0, 5, E - 1):
            full = state[e]["Full State"]
            assert len(full) == 2 and full[0].shape == (A, 6) and full[1].shape == (3,) and full[0].dtype == np.float32
            np.testing.assert_array_equal(full[0], g[e, :6 * A].reshape(A, 6))
            np.testing.assert_array_equal(full[1], g[e, 6 * A:])
            st = ora.get_state(e)
            np.testing.assert_array_equal(full[0][:, 4], [st.robots[r].team for r in range(A)])
            assert full[1][2] == st.ball_owned
        # models/train.py:272, verbatim
        agentFinished = torch.tensor([[agent[-1] for agent in s_['Full State'][0]] for s_ in state]).bool()
        assert tuple(agentFinished.shape) == (E, A)
        exp = [[bool(ora.get_state(e).robots[r].fallen or ora.get_state(e).robots[r].penalized) for r in range(A)] for e in range(E)]
        assert agentFinished.tolist() == exp
        downs += int(agentFinished.sum())
        # models/train.py:270-271 on 'Recon States' keeps working beside it
        fullStates = [[x[0::2] for x in s_['Recon States']] for s_ in state]
        assert len(fullStates) == E and len(fullStates[0]) == A and len(fullStates[0][0]) == 2
    assert downs > 0, "some robot must have fallen or been penalized in 40 steps with the fall dice on"
    env.close()


def test_driving_full_state_feeds_the_trainers_line(gpu):
    """the same consumer line on Driving (DrivingEnvironment.py:306, :697-711: cars [A, 7] with `finished` last)"""
    dynenv_amd, torch, _ = gpu
    E, n = 6, 10
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, n, seed=17)
    ora = ol.OracleEnv(env_type=1, num_envs=E, n_players=n, seed=17)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(4)
    for s in range(30):
        a = _acts(rng, env)
        new_obs, rew, dones, state = env.step(a)
        oc, _, _ = ora.step(a)
    agentFinished = torch.tensor([[agent[-1] for agent in s_['Full State'][0]] for s_ in state]).bool()
    assert tuple(agentFinished.shape) == (E, n)
    assert agentFinished.tolist() == [[bool(ora.get_state(e).cars[c].finished) for c in range(n)] for e in range(E)]
    cars = state[2]["Full State"][0]
    assert cars.shape == (n, 7)
    np.testing.assert_array_equal(cars, oc[2, -1][:, [0, 1, 2, 3, 4, 5, 8]])
    env.close()


def test_replay_with_the_continuous_head_channel_across_an_episode_end(gpu, tmp_path):
    """a use_continuous_actions=True RoboCup run (float head turns, RoboCupEnvironment.py:339-342) recorded to a replay file and
    re-run on a fresh handle: the head channel travels as float64, so rewards and dones reproduce bit for bit - also after the
    lock-step reset in the middle of the recording"""
    dynenv_amd, torch, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType, make_dyn_env
    from dynenv_amd.replay import ReplayRecorder, replay
    E, n = 6, 5
    venv, _ = make_dyn_env(DynEnvType.ROBO_CUP, E, n, False, ObservationType.FULL, NoiseType.REALISTIC, 0, True, seed=11)
    venv.reset_flat()
    rng = np.random.default_rng(8)
    S = venv.steps_per_episode

    def act():
        a = _acts(rng, venv).astype(np.float64)
        a[..., 3] = (rng.random((E, 2 * n)) - 0.5) * 5.9   # fractional head turns: an int8 recording would change the run
        return a
    for s in range(S - 5):
        venv.step_flat(act())
    rec = ReplayRecorder(venv)
    seen_done = 0
    for s in range(14):
        rec.step(act())
        seen_done += int(venv.last_done)
    assert seen_done == 1
    path = str(tmp_path / "head.npz")
    rec.save(path)
    z = np.load(path)
    assert z["heads"].dtype == np.float64 and z["heads"].shape == (14, E, 2 * n) and np.abs(z["heads"] - np.round(z["heads"])).max() > 0.1
    final = venv.rewards.cpu().numpy().copy()
    env2, nsteps = replay(path)           # asserts the digest of every step
    assert nsteps == 14 and env2.allow_head_turn
    np.testing.assert_array_equal(env2.rewards.cpu().numpy(), final)
    for e in range(E):
        assert ol.rc_state_to_dict(env2.get_state(e)).keys() == ol.rc_state_to_dict(venv.get_state(e)).keys()
        a, b = ol.rc_state_to_dict(env2.get_state(e)), ol.rc_state_to_dict(venv.get_state(e))
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    venv.close(); env2.close()


@pytest.mark.parametrize("flags", [ol.FLAG_RANDOM_INIT | ol.FLAG_CAN_FALL, ol.FLAG_DETERMINISTIC_TURN | ol.FLAG_CAN_FALL,
                                   ol.FLAG_RANDOM_INIT | ol.FLAG_DETERMINISTIC_TURN, 0])
def test_robocup_class_switches_against_the_oracle(gpu, flags):
    """randomInit (random robot spots, ball position and ownership), deterministicTurn (head preset + forced head action),
    and canFall = useObsRewards = False (flags = 0 is an explicit choice, not "defaults")."""
    dynenv_amd, torch, _ = gpu
    E, n = 16, 5
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, E, n, seed=21, flags=flags)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=n, seed=21, flags=flags)
    assert env.cfg.flags == flags
    np.testing.assert_array_equal(env.reset_flat().cpu().numpy(), ora.reset())
    rng = np.random.default_rng(3)
    for s in range(12):
        a = _acts(rng, env)
        og, rg, dg = env.step_flat(a)
        oc, rc, dc = ora.step(a)
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="rewards step %d" % s)
        np.testing.assert_array_equal(og.cpu().numpy(), oc, err_msg="obs step %d" % s)
    if flags & ol.FLAG_RANDOM_INIT:
        owned = {env.get_state(e).ball_owned for e in range(E)}
        assert len(owned) > 1, "randomInit draws the ball ownership"
    env.close()


def test_robocup_continuous_head_channel(gpu):
    """make_dyn_env(use_continuous_actions=True) -> allowHeadTurn: the 4th action is the reference's Box(-3, 3) head turn and
    reaches Robot.turnHead as a float (dynenv_step_head), bit-identical to the oracle"""
    dynenv_amd, torch, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType, make_dyn_env
    E, n = 8, 5
    venv, name = make_dyn_env(DynEnvType.ROBO_CUP, E, n, False, ObservationType.FULL, NoiseType.REALISTIC, 0, True, seed=4)
    assert venv.allow_head_turn and len(venv.action_space.spaces) == 2
    flags = ol.ROBOCUP_DEFAULT_FLAGS | ol.FLAG_ALLOW_HEAD_TURN
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=n, seed=4, flags=flags)
    np.testing.assert_array_equal(venv.reset_flat().cpu().numpy(), ora.reset())
    rng = np.random.default_rng(5)
    for s in range(10):
        a = _acts(rng, venv).astype(np.float64)
        a[..., 3] = np.round((rng.random((E, 2 * n)) - 0.5) * 6.0, 3) * (rng.random((E, 2 * n)) > 0.2)
        og, rg, dg = venv.step_flat(a, validate=True)
        oc, rc, dc = ora.step(a.astype(np.int32), head=a[..., 3])
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="rewards step %d" % s)
        np.testing.assert_array_equal(og.cpu().numpy(), oc, err_msg="obs step %d" % s)
    heads = {venv.get_state(0).robots[r].head_angle for r in range(2 * n)}
    assert any(abs(h) > 1e-6 and abs(h * 720 / np.pi - round(h * 720 / np.pi)) > 1e-6 for h in heads), "fractional head turns must show"
    with pytest.raises(Exception):
        bad = _acts(rng, venv).astype(np.float64)
        bad[..., 3] = 7.5
        venv.step_flat(bad, validate=True)
    venv.close()


@pytest.mark.parametrize("kind", ["DRIVE", "ROBO_CUP"])
def test_malformed_actions_raise_a_flag_and_move_nothing(gpu, kind):
    """the reference raises on an action outside the action space; the batched kernels ignore that agent's action for the step
    (the oracle does the same) and report error bit 1 - never an out-of-range index"""
    dynenv_amd, torch, _ = gpu
    robocup = kind == "ROBO_CUP"
    E, n = 4, 5 if robocup else 10
    flags = ol.ROBOCUP_DEFAULT_FLAGS if robocup else 0
    env = dynenv_amd.BatchedDynEnv(getattr(dynenv_amd.DynEnvType, kind), E, n, seed=8, flags=flags)
    ora = ol.OracleEnv(env_type=0 if robocup else 1, num_envs=E, n_players=n, seed=8, flags=flags)
    env.reset_flat(); ora.reset()
    rng = np.random.default_rng(6)
    a = _acts(rng, env)
    a[1, 2, 0] = 9
    a[2, 0, 1] = -3
    if robocup:
        a[3, 4, 2] = 5
    og, rg, dg = env.step_flat(a)           # validate=False: straight into the kernel
    oc, rc, dc = ora.step(a)
    np.testing.assert_array_equal(rg.cpu().numpy(), rc)
    np.testing.assert_array_equal(og.cpu().numpy(), oc)
    assert env.error_flags() & 2
    with pytest.raises(Exception):
        env.step_flat(a, validate=True)
    env.close()


def test_crossed_capsule_cores_on_both_sides(gpu):
    """a foot laid across another (dynenv_set_state) takes Chipmunk's minimum-translation normal on both sides, bit for
    bit; cores exactly collinear - where even that normal's sign is a convention - raise error bit 4 on the HIP path and in the oracle,
    sticky until set_state / reset, and the compat step() raises; feet that merely overlap raise nothing.  The 4096 x 240 soaks assert
    the flag stays down in play."""
    import math
    from test_kat_general import _feet_scene
    dynenv_amd, torch, _ = gpu
    flags = ol.FLAG_USE_OBS_REWARDS
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, 4, 5, seed=3, flags=flags)
    ora = ol.OracleEnv(env_type=0, num_envs=4, n_players=5, seed=3, flags=flags)
    env.reset_flat(); ora.reset()
    a = np.zeros((4, 10, 4), np.int32)
    a[..., 3] = 3
    scenes = {1: _feet_scene(ora, 1, math.pi / 2, 10.0, 19.0), 2: _feet_scene(ora, 2, 0.0, 0.0, 20.0 + 1e-3), 3: _feet_scene(ora, 3, 0.0, 5.0, 20.0)}
    for sim in (ora, env):
        for i, st in scenes.items():
            sim.set_state(i, st)
    assert env.error_flags() == 0
    for _ in range(2):
        og, rg, _d = env.step_flat(a, auto_reset=False)
        oc, rc, _d = ora.step(a)
        np.testing.assert_array_equal(og.cpu().numpy(), oc)
        np.testing.assert_array_equal(rg.cpu().numpy(), rc)
        for i in range(4):
            assert bytes(env.get_state(i)) == bytes(ora.get_state(i)), i
    assert [ora.degenerate_env(i) for i in range(4)] == [0, 0, 0, 16]
    assert env.error_flags() == ora.degenerate() == 16      # sticky
    env.set_state(3, ora.get_state(0))
    assert env.error_flags() == 0                           # set_state clears the environment's flags
    env.close()
    venv, _ = dynenv_amd.make_dyn_env(dynenv_amd.DynEnvType.ROBO_CUP, 2, 5, False, dynenv_amd.ObservationType.FULL, dynenv_amd.NoiseType.REALISTIC, 0.0, False)
    venv.reset()
    venv.set_state(1, scenes[3])
    with pytest.raises(Exception, match="error bit 4"):
        venv.step(np.zeros((2, 10, 4), np.int64) + np.array([0, 0, 0, 3]))
    venv.close()


@pytest.mark.parametrize("kind,n", [("DRIVE", 10), ("ROBO_CUP", 5)])
def test_checkpoint_and_replay_across_an_episode_end(gpu, kind, n, tmp_path):
    """restore() puts the host's episode position where the device state is: a checkpoint taken near the end of an episode and
    restored into a FRESH handle resets at the same step as the original run and stays bit-identical after the reset"""
    dynenv_amd, torch, _ = gpu
    robocup = kind == "ROBO_CUP"
    E = 8
    mk = lambda: dynenv_amd.BatchedDynEnv(getattr(dynenv_amd.DynEnvType, kind), E, n, seed=13)
    env = mk()
    env.reset_flat()
    rng = np.random.default_rng(7)
    S = env.steps_per_episode
    acts = [_acts(rng, env) for _ in range(S + 12)]
    for s in range(S - 6):
        env.step_flat(acts[s])
    ck = env.checkpoint()
    ref = []
    for s in range(S - 6, S + 12):
        o, r, d = env.step_flat(acts[s])
        ref.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), bool(env.last_done)))
    assert [x[2] for x in ref].count(True) == 1 and ref[5][2], "the episode ends 6 steps after the checkpoint"
    env2 = mk()
    env2.restore(ck)
    assert env2._episode_step == S - 6
    for k, s in enumerate(range(S - 6, S + 12)):
        o, r, d = env2.step_flat(acts[s])
        assert bool(env2.last_done) == ref[k][2], "step %d" % s
        np.testing.assert_array_equal(r.cpu().numpy(), ref[k][1], err_msg="rewards step %d" % s)
        np.testing.assert_array_equal(o.cpu().numpy(), ref[k][0], err_msg="obs step %d" % s)
    env.close(); env2.close()


def test_refresh_obs_and_device_guard(gpu):
    dynenv_amd, torch, _ = gpu
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 4, 10, seed=3)
    dev = torch.cuda.current_device()
    obs0 = env.reset_flat().clone()
    env.obs.zero_()
    np.testing.assert_array_equal(env.refresh_obs().cpu().numpy(), obs0.cpu().numpy())
    st = env.get_state(2)
    env.set_state(2, st)
    np.testing.assert_array_equal(env.refresh_obs().cpu().numpy(), obs0.cpu().numpy())
    assert torch.cuda.current_device() == dev
    with pytest.raises(dynenv_amd._capi.DynEnvError):
        dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 4, 10, flags=4)      # RoboCup's switches mean nothing to Driving
    with pytest.raises(dynenv_amd._capi.DynEnvError):
        dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, 4, 5, flags=64)   # unknown bit
    env.close()


@pytest.mark.parametrize("E,steps", [(4096, 420), (8192, 300), (4000, 420), (3100, 420)])
def test_simd_isolation_changes_the_schedule_and_nothing_else(gpu, monkeypatch, E, steps):
    """Driving on a 256-CU device.  3073 .. 4096 environments (round 5: any E in that range, the launch is 4096 regular blocks + spares
    whatever E is): those that were slowest in the previous step get a SIMD of their own;
    8192: they start first (which block steps which environment).  A handle created with DYNENV_NO_ISOLATION=1
    must produce the same observations, rewards, dones and states, bit for bit, and no placeholder may ever give up waiting."""
    dynenv_amd, torch, _ = gpu
    A = 10
    iso = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=11)
    monkeypatch.setenv("DYNENV_NO_ISOLATION", "1")
    ref = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=11)
    monkeypatch.delenv("DYNENV_NO_ISOLATION")
    assert ref.debug_counters()["isolated_next"] == -1
    on = iso.debug_counters()["isolated_next"] >= 0
    if not on:
        pytest.skip("the scheduling is only switched on for 4096 or more environments on a 256-CU device")
    iso.reset_flat(); ref.reset_flat()
    g = torch.Generator(device="cuda").manual_seed(5)
    seen = 0
    for s in range(steps):
        a = torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32)
        o1, r1, d1 = iso.step_flat(a, auto_reset=False)
        o2, r2, d2 = ref.step_flat(a, auto_reset=False)
        if s % 20 == 19 or s == steps - 1:
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), "step %d" % s
            seen = max(seen, iso.debug_counters()["isolated_next"])
    assert seen > 0, "no environment was ever isolated in %d steps" % steps
    c = iso.debug_counters()
    assert c["isolation_timeouts"] == 0
    if E <= 4096:  # one launch at a time on this device: the placement the isolation relies on must have been observed throughout
        assert c["isolation_mode"] == 1 and c["placement_validated"] == 1 and c["placement_invalid_launches"] == 0, c
    for e in (0, 1, 1023, 1024, 2048, min(4095, E - 1), E - 1):
        s1, s2 = iso.get_state(e), ref.get_state(e)
        assert bytes(s1) == bytes(s2), "state of environment %d" % e
    assert iso.error_flags() == 0 and ref.error_flags() == 0
    iso.close(); ref.close()


@pytest.mark.parametrize("E,steps", [(4096, 300), (6000, 260)])
def test_partial_vision_deadline_changes_the_schedule_and_nothing_else(gpu, monkeypatch, E, steps):
    """Driving with Partial observations: with a forecast of the launch's end (scheduling mode 3, or 2 with more environments than
    one residency round) every environment runs its vision passes until then and leaves the rest to the deferred launch; a handle
    created with DYNENV_NO_ISOLATION=1 has no forecast and follows the static rule (all ten passes fused, or none from five
    contact-path substeps on).  Who computes a pass must not matter: observations, rewards, dones and states bit for bit."""
    dynenv_amd, torch, _ = gpu
    from dynenv_amd import NoiseType, ObservationType
    A = 10
    kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3, seed=13)
    dyn = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, **kw)
    monkeypatch.setenv("DYNENV_NO_ISOLATION", "1")
    ref = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, **kw)
    monkeypatch.delenv("DYNENV_NO_ISOLATION")
    assert ref.debug_counters()["isolation_mode"] == 0 and dyn.debug_counters()["isolation_mode"] in (2, 3)
    o1, o2 = dyn.reset_flat(), ref.reset_flat()
    assert torch.equal(o1, o2)
    g = torch.Generator(device="cuda").manual_seed(6)
    for s in range(steps):
        a = torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32)
        o1, r1, d1 = dyn.step_flat(a, auto_reset=False)
        o2, r2, d2 = ref.step_flat(a, auto_reset=False)
        if s % 10 == 9 or s == steps - 1:
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), "step %d" % s
    assert dyn.debug_counters()["contact"] > 0
    for e in (0, 1, 1023, 2048, E - 1):
        assert bytes(dyn.get_state(e)) == bytes(ref.get_state(e)), "state of environment %d" % e
    assert dyn.error_flags() == 0 and ref.error_flags() == 0
    dyn.close(); ref.close()


def test_two_handles_on_two_streams_isolation_validates_itself(gpu, monkeypatch):
    """Two 4096-environment Driving handles stepped concurrently on two streams: the block -> SIMD placement that isolation
    assumes (one launch owning the device) no longer holds.  Every launch records where its blocks ran and the next ones only
    isolate while the record checks out, so: results are bit-identical to a DYNENV_NO_ISOLATION pair, no placeholder ever gives up
    waiting, and the pair is not slower than the pair without isolation (a stale placement guess would park wave slots).  The host
    side watches the device's count of launches that did not validate and drops to the plain launch while they keep failing
    (isolation_pauses), so the second pass runs the very launch DYNENV_NO_ISOLATION gives."""
    import time
    dynenv_amd, torch, _ = gpu
    E, A, steps = 4096, 10, 300
    mk = lambda seed: dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=seed)
    probe = mk(1)
    mode = probe.debug_counters()["isolation_mode"]
    probe.close()
    if mode != 1:
        pytest.skip("SIMD isolation is only switched on for 4096 environments on a 256-CU device")
    g = torch.Generator(device="cuda").manual_seed(9)
    acts = [torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(steps)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(pair):
        for h in pair:
            h.reset_flat()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in acts:
            for h, st in zip(pair, streams):
                with torch.cuda.stream(st):
                    h.step_flat(a, auto_reset=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    iso = [mk(11), mk(12)]
    monkeypatch.setenv("DYNENV_NO_ISOLATION", "1")
    ref = [mk(11), mk(12)]
    monkeypatch.delenv("DYNENV_NO_ISOLATION")
    # alternating passes (the device's clocks drift by several per cent over the first seconds of load: whichever pair ran
    # first looked 6 % slower); every pass is the same episode again from a reset, the scheduler state carries over
    t = [(run(iso), run(ref)) for _ in range(3)]
    t_iso, t_ref = min(x for x, _ in t), min(y for _, y in t)
    for a, b in zip(iso, ref):
        assert torch.equal(a.obs, b.obs) and torch.equal(a.rewards, b.rewards)
        for e in (0, 1024, 4095):
            assert bytes(a.get_state(e)) == bytes(b.get_state(e))
        c = a.debug_counters()
        assert c["isolation_timeouts"] == 0, c
        # either the launches really overlapped and the host paused isolation, or (a host too slow to overlap them) they validated
        assert c["isolation_pauses"] >= 1 or c["placement_invalid_launches"] < steps // 4, c
        assert a.error_flags() == 0 and b.error_flags() == 0
    assert t_iso <= 1.10 * t_ref, "two concurrent handles: %.1f ms with isolation, %.1f ms without" % (t_iso * 1e3, t_ref * 1e3)
    print("two handles x 4096 on two streams: %.3f ms per step pair with isolation (%s), %.3f without" %
          (t_iso * 1e3 / steps, iso[0].debug_counters(), t_ref * 1e3 / steps))
    for h in iso + ref:
        h.close()


def test_partial_observation_rows_beyond_the_layouts_capacity_are_reported(gpu):
    """VERDICT r4 item 6a.  The reference's observation lists have no cap (DrivingEnvironment.py:816-890); the dense layout holds 24 cars /
    32 obstacles / 40 pedestrians / 16 lanes per agent, SURVEY Appendix E's worst case is 39 cars.  Reaching it takes 15 of ~30 objects
    misclassified or falsely detected in one pass at 0.4 % each - no input can force it - so the overflow path is exercised with a
    test build of the same sources whose lists are LIMITED to 2 / 3 / 2 / 3 rows inside the same layout
    (dynenv_amd/libdynenv_hip_testcaps.so, __graft_entry__.build): error bit 3 comes up, step_flat keeps going, the compat step()
    raises; the product library on the same episode reports nothing."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "dynenv_amd", "libdynenv_hip_testcaps.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import dynenv_amd as d
from dynenv_amd import _capi
env = d.BatchedDynEnv(d.DynEnvType.DRIVE, 64, 10, observationType=d.ObservationType.PARTIAL, noiseType=d.NoiseType.REALISTIC, noiseMagnitude=3, seed=9)
env.reset()
a = np.ones((64, 10, 2), np.int64)
env.step_flat(a.astype(np.int32))
flags = env.error_flags()
counts = env.obs[..., -4:].max().item()
raised = False
try:
    env.step(a)
except _capi.DynEnvError as e:
    raised = "rows were dropped" in str(e)
print("RESULT", flags, int(counts), int(raised))
''' % root
    out = {}
    for name, extra in (("testcaps", {"DYNENV_HIP_LIB": lib}), ("product", {})):
        env = dict({k: v for k, v in os.environ.items() if k != "DYNENV_HIP_LIB"}, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out[name] = [int(x) for x in [ln for ln in r.stdout.decode().splitlines() if ln.startswith("RESULT")][0].split()[1:]]
    flags, counts, raised = out["testcaps"]
    assert flags & 8 and raised and counts <= 3, out      # reported, raised by step(), and no list longer than its limit
    flags, counts, raised = out["product"]
    assert flags == 0 and not raised and counts > 3, out
