"""Exact checkpoint / deterministic replay: restoring a checkpoint taken in the middle of an episode —
with live contacts in the cache — and re-applying the same actions reproduces observations, rewards and dones bit for bit,
on the same handle, on a fresh handle, and through a replay file."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _actions(rng, E, A, hi):
    import torch
    return torch.tensor(np.stack([rng.integers(0, k, (E, A)) for k in hi], -1).astype(np.int32), device="cuda")


@pytest.mark.parametrize("cfg", ["driving", "driving_partial", "robocup", "robocup_partial"])
def test_checkpoint_restore_is_exact_mid_episode(cfg, tmp_path):
    import torch
    from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType
    from dynenv_amd.replay import ReplayRecorder, replay
    E = 64
    kw = {}
    if cfg.startswith("robocup"):
        et, n, hi, warm = DynEnvType.ROBO_CUP, 5, [5, 3, 3, 7], 60
        if cfg == "robocup_partial":
            kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
    else:
        et, n, hi, warm = DynEnvType.DRIVE, 10, [3, 3], 250
        if cfg == "driving_partial":
            kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
    env = BatchedDynEnv(et, E, n, seed=11, **kw)
    env.reset_flat()
    rng = np.random.default_rng(3)
    for _ in range(warm):
        env.step_flat(_actions(rng, E, env.n_agents, hi))
    if not cfg.startswith("robocup"):
        assert env.debug_counters()["slot_sum"] > 0, "the checkpoint should be taken with live contacts in the cache"
    rec = ReplayRecorder(env)
    acts = [_actions(rng, E, env.n_agents, hi) for _ in range(40)]
    ref = []
    for a in acts:
        o, r, d = rec.step(a)
        ref.append((o.clone(), r.clone(), d.clone()))
    path = os.path.join(str(tmp_path), "run.npz")
    rec.save(path)
    # same handle
    env.restore(rec.checkpoint)
    for a, (o0, r0, d0) in zip(acts, ref):
        o, r, d = env.step_flat(a)
        assert torch.equal(o, o0) and torch.equal(r, r0) and torch.equal(d, d0)
    # fresh handle, different seed at construction (the checkpoint carries the seed)
    env2 = BatchedDynEnv(et, E, n, seed=999, **kw)
    env2.restore(rec.checkpoint)
    for a, (o0, r0, d0) in zip(acts, ref):
        o, r, d = env2.step_flat(a)
        assert torch.equal(o, o0) and torch.equal(r, r0) and torch.equal(d, d0)
    # replay file
    _, n_steps = replay(path)
    assert n_steps == 40
    # a checkpoint of another configuration is refused
    env3 = BatchedDynEnv(et, E // 2, n, seed=11, **kw)
    with pytest.raises(Exception):
        env3.restore(rec.checkpoint)
    for x in (env, env2, env3):
        x.close()
