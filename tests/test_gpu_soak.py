"""Whole episodes at full batch size against the oracle, bit for bit (tools/soak_parity.py): rewards and dones of every step
of every environment, observations every 25th step, episode statistics.  The exact shortcuts of the step kernels are only
exercised in their full variety at this scale."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,E", [("driving", 4096), ("robocup", 4096), ("driving_partial", 4096), ("robocup_partial", 4096)])
def test_full_episode_full_batch_parity(oracle_built, cfg, E):
    import soak_parity
    soak_parity.run(cfg, E, 20261003)


def test_rank7_shard_of_the_8_gpu_configuration(oracle_built):
    """BASELINE configs[4] = 32768 Driving environments over 8 GPUs: rank 7 owns global environments [28672, 32768).  Its whole
    shard, a whole episode, against the oracle run on the same global ids (RNG streams are keyed by global environment id, so
    this is what that rank computes in the 8-GPU job) - and it is NOT the result of rank 0's ids."""
    import numpy as np
    import soak_parity
    soak_parity.run("driving", 4096, 20261003, env_id_offset=28672)
    import oracle_lib as ol
    a = ol.OracleEnv(num_envs=4, n_players=10, seed=20261003, env_id_offset=28672)
    b = ol.OracleEnv(num_envs=4, n_players=10, seed=20261003, env_id_offset=0)
    assert not np.array_equal(a.reset(), b.reset())


def test_all_32768_environments_of_the_8_gpu_configuration_on_one_gpu(oracle_built):
    """BASELINE configs[4]'s whole population - 32768 Driving environments, the global ids the eight ranks own together - as ONE handle
    on this box's one GPU (eight residency rounds, the slow environments of the previous step started first: scheduling mode 2), a whole
    episode, every environment against the oracle.  With the shard-count invariance of tests/test_distributed_gloo.py this is what
    the eight ranks compute between them."""
    import soak_parity
    soak_parity.run("driving", 32768, 20261003)


def test_crossing_feet_in_play_take_the_same_normal_on_both_sides(oracle_built):
    """Round 6: capsule cores that CROSS are reached in play (the two feet of a robot that has been knocked about) and take Chipmunk's
    minimum-translation normal on both sides (oracle/cp_lite.c cores_crossing_normal, rc_narrowphase).  An episode that contains such
    substeps - 4096 environments, seed 1, a pool of 16 action tensors repeated, the first crossing at step 138 - stays bit-identical
    between the HIP path and the oracle through them, with no error flag (crossing is not the degenerate case bit 4 reports)."""
    import ctypes as C
    import numpy as np
    import oracle_lib as ol
    from dynenv_amd import BatchedDynEnv, DynEnvType
    E, seed, steps = 4096, 1, 150
    lib = ol.lib()
    lib.oracle_cp_cores_cross.restype = C.c_long
    before = lib.oracle_cp_cores_cross()
    env = BatchedDynEnv(DynEnvType.ROBO_CUP, E, 5, seed=seed, flags=ol.ROBOCUP_DEFAULT_FLAGS)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=5, seed=seed, flags=ol.ROBOCUP_DEFAULT_FLAGS, threads=16)
    assert np.array_equal(env.reset_flat().cpu().numpy(), ora.reset())
    rng = np.random.default_rng(seed)
    pool = [np.stack([rng.integers(0, k, (E, 10)) for k in (5, 3, 3, 7)], -1).astype(np.int32) for _ in range(16)]
    for s in range(steps):
        og, rg, dg = env.step_flat(pool[s & 15], auto_reset=False)
        if s >= 136:
            oc, rc, dc = ora.step(pool[s & 15])
            assert np.array_equal(og.cpu().numpy(), oc), "observations of step %d" % s
        else:
            rc, dc = ora.step_noobs(pool[s & 15])
        assert np.array_equal(rg.cpu().numpy(), rc), "rewards of step %d" % s
    assert lib.oracle_cp_cores_cross() > before, "the episode must contain crossing capsule cores"
    assert env.error_flags() == 0 and ora.degenerate() == 0
    env.close()
