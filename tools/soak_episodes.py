#!/usr/bin/env python3
"""Three whole episodes on ONE handle (resets in between, the SIMD-isolation lists persist across them) against the CPU oracle:
rewards and dones of every 5th step, observations every 50th, bit for bit.  Usage (GPU box): python tools/soak_episodes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
from dynenv_amd import BatchedDynEnv, DynEnvType  # noqa: E402

E, A, seed = 4096, 10, 2026
env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=seed)
ora = ol.OracleEnv(env_type=1, num_envs=E, n_players=A, seed=seed, threads=16)
rng = np.random.default_rng(seed)
t0 = time.time()
for ep in range(3):
    assert np.array_equal(env.reset_flat().cpu().numpy(), ora.reset()), "reset of episode %d" % ep
    for s in range(600):
        a = np.stack([rng.integers(0, 3, (E, A)) for _ in range(2)], -1).astype(np.int32)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        oc, rc, dc = ora.step(a)
        if s % 5 == 4 or s == 599:
            assert np.array_equal(rg.cpu().numpy(), rc), "rewards, episode %d step %d" % (ep, s)
            assert np.array_equal(dg.cpu().numpy().astype(bool), np.asarray(dc).astype(bool)), "dones, episode %d step %d" % (ep, s)
        if s % 50 == 49 or s == 599:
            assert np.array_equal(og.cpu().numpy(), oc), "observations, episode %d step %d" % (ep, s)
    c = env.debug_counters()
    print("episode %d OK: isolated in the last step %d, placeholder timeouts %d, split solves %d (%.0f s)" % (
        ep, c["isolated_next"], c["isolation_timeouts"], c["split"], time.time() - t0), flush=True)
assert env.error_flags() == 0
print("soak OK: 3 episodes x 600 steps x %d environments on one handle, bit-identical to the oracle" % E)
