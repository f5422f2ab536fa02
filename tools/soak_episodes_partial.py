#!/usr/bin/env python3
"""Three whole episodes on ONE handle with Partial observations (resets in between: the deferred-vision lists and the scheduling
forecast persist across them) against the CPU oracle: rewards and dones of every step, observations every 20th, bit for bit.
Usage (GPU box): python tools/soak_episodes_partial.py [driving|robocup] [envs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "robocup"
E, seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4096, 2027
ol.build()
kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
okw = dict(obs_type=1, noise_type=1, noise_magnitude=3.0)
if what == "robocup":
    env = BatchedDynEnv(DynEnvType.ROBO_CUP, E, 5, seed=seed, flags=ol.ROBOCUP_DEFAULT_FLAGS, **kw)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=5, seed=seed, flags=ol.ROBOCUP_DEFAULT_FLAGS, threads=16, **okw)
    hi, steps = [5, 3, 3, 7], 240
else:
    env = BatchedDynEnv(DynEnvType.DRIVE, E, 10, seed=seed, **kw)
    ora = ol.OracleEnv(env_type=1, num_envs=E, n_players=10, seed=seed, threads=16, **okw)
    hi, steps = [3, 3], 600
rng = np.random.default_rng(seed)
t0 = time.time()
for ep in range(3):
    assert np.array_equal(env.reset_flat().cpu().numpy(), ora.reset()), "reset of episode %d" % ep
    for s in range(steps):
        a = np.stack([rng.integers(0, k, (E, env.n_agents)) for k in hi], -1).astype(np.int32)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        if s % 20 == 19 or s == steps - 1:
            oc, rc, dc = ora.step(a)
            assert np.array_equal(og.cpu().numpy(), oc), "observations, episode %d step %d" % (ep, s)
        else:
            rc, dc = ora.step_noobs(a)
        rgn = rg.cpu().numpy()
        assert np.array_equal(rgn, rc), "rewards, episode %d step %d: %d entries differ" % (ep, s, int((rgn != rc).sum()))
        assert np.array_equal(dg.cpu().numpy().astype(bool), np.asarray(dc).astype(bool)), "dones, episode %d step %d" % (ep, s)
    print("episode %d of %s Partial: %d envs x %d steps bit-identical to the oracle (%.0f s)" % (ep, what, E, steps, time.time() - t0), flush=True)
assert env.error_flags() == 0
print("soak OK")
