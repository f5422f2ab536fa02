"""The HIP path against the trajectories of the reference's OWN `step()` with collisions (tests/golden/gen_golden_contacts.py; see
tests/test_oracle_golden_contacts.py for what they are and for the tolerances), through the C ABI.  -m gpu."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import test_oracle_golden_contacts as tc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu(oracle_built):
    import torch
    import dynenv_amd
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return dynenv_amd


@pytest.mark.parametrize("tag", tc.TAGS)
def test_reference_driving_step_with_collisions_on_the_hip_path(gpu, tag):
    z = np.load(os.path.join(G, "driving_contacts.npz"))
    envs = []

    def make_env(n_players, seed, offset):
        env = gpu.BatchedDynEnv(gpu.DynEnvType.DRIVE, 1, n_players, seed=seed, env_id_offset=offset)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, d = env.step_flat(a[None], auto_reset=False)
            return o[0, 0].cpu().numpy(), r[0].cpu().numpy(), int(d[0])

        def stats():
            r, p, o, g = env.episode_stats()
            return r[0].cpu().numpy(), p[0].cpu().numpy(), o[0].cpu().numpy(), g[0].cpu().numpy()
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0)), stats
    tc.check_trajectory(z, tag, make_env)
    assert envs[0].error_flags() == 0
    envs[0].close()


@pytest.mark.parametrize("tag", tc.RC_TAGS)
def test_reference_robocup_step_with_collisions_on_the_hip_path(gpu, tag):
    z = np.load(os.path.join(G, "robocup_contacts.npz"))
    envs = []

    def make_env(n, seed, offset, flags):
        env = gpu.BatchedDynEnv(gpu.DynEnvType.ROBO_CUP, 1, n, seed=seed, env_id_offset=offset, flags=flags)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, d = env.step_flat(a[None], auto_reset=False)
            return o[0].cpu().numpy(), r[0].cpu().numpy(), int(d[0])
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))
    assert tc.check_robocup_trajectory(z, tag, make_env) >= tc.RC_MIN_STEPS[tag]
    assert envs[0].error_flags() == 0
    envs[0].close()


@pytest.mark.parametrize("tag", tc.P_TAGS)
def test_reference_step_with_partial_observations_and_collisions_on_the_hip_path(gpu, tag):
    """BASELINE configs[3] as the reference runs it (Partial observations + Realistic noise 3 inside step(), with crashes)"""
    z = np.load(os.path.join(G, "driving_partial_contacts.npz"))
    envs = []

    def make_env(n_players, seed, offset, magn):
        env = gpu.BatchedDynEnv(gpu.DynEnvType.DRIVE, 1, n_players, observationType=gpu.ObservationType.PARTIAL, noiseType=gpu.NoiseType.REALISTIC,
                                noiseMagnitude=magn, seed=seed, env_id_offset=offset)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, d = env.step_flat(a[None], auto_reset=False)
            return o[0, 0].cpu().numpy(), r[0].cpu().numpy(), int(d[0])
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))
    assert (tc.check_partial_trajectory(z, tag, make_env) > 50).all()
    assert envs[0].error_flags() == 0
    envs[0].close()


@pytest.mark.parametrize("tag", tc.RCP_TAGS)
def test_reference_robocup_step_with_partial_observations_on_the_hip_path(gpu, tag):
    z = np.load(os.path.join(G, "robocup_partial_contacts.npz"))
    envs = []

    def make_env(n, seed, offset, flags, magn):
        env = gpu.BatchedDynEnv(gpu.DynEnvType.ROBO_CUP, 1, n, observationType=gpu.ObservationType.PARTIAL, noiseType=gpu.NoiseType.REALISTIC,
                                noiseMagnitude=magn, seed=seed, env_id_offset=offset, flags=flags)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, d = env.step_flat(a[None], auto_reset=False)
            return o[0].cpu().numpy(), r[0].cpu().numpy(), int(d[0])
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))
    assert tc.check_robocup_trajectory(z, tag, make_env, partial=True) >= tc.RCP_MIN_STEPS[tag]
    assert envs[0].error_flags() == 0
    envs[0].close()
