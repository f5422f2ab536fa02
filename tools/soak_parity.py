#!/usr/bin/env python3
"""Full-size soak: every environment of a 4096-env batch against the CPU oracle for a whole episode — rewards and dones
of every step, observations every 25th step, episode statistics at the end; bit for bit.  This is the test of the exact
shortcuts (quiescent / steady replay / light mode / bias-only / tangent-free solver, lane-parallel RoboCup ticks) at a
diversity the unit-sized parity tests cannot reach.  Usage (GPU box):  python tools/soak_parity.py [driving|robocup|...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(cfg, E, seed, players=None, env_id_offset=0):
    import numpy as np
    import torch
    import oracle_lib as ol
    from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType
    part = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
    opart = dict(obs_type=1, noise_type=1, noise_magnitude=3.0)
    if cfg.startswith("robocup"):
        et, oet, n, hi, steps, flags = DynEnvType.ROBO_CUP, 0, 5, [5, 3, 3, 7], 240, ol.ROBOCUP_DEFAULT_FLAGS
    else:
        et, oet, n, hi, steps, flags = DynEnvType.DRIVE, 1, 10, [3, 3], 600, 0
    if players is not None:
        n = players
    kw, okw = (part, opart) if cfg.endswith("partial") else ({}, {})
    # env_id_offset: the shard of a bigger job (rank g of a sharded run owns global environments [g E, (g + 1) E): RNG streams are keyed by global id)
    env = BatchedDynEnv(et, E, n, seed=seed, flags=flags, env_id_offset=env_id_offset, **kw)
    ora = ol.OracleEnv(env_type=oet, num_envs=E, n_players=n, seed=seed, flags=flags, threads=16, env_id_offset=env_id_offset, **okw)
    assert np.array_equal(env.reset_flat().cpu().numpy(), ora.reset()), "reset"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for s in range(steps):
        a = np.stack([rng.integers(0, k, (E, env.n_agents)) for k in hi], -1).astype(np.int32)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        if s % 25 == 24 or s == steps - 1:
            oc, rc, dc = ora.step(a)
        else:
            oc = None
            rc, dc = ora.step_noobs(a)
        rgn = rg.cpu().numpy()
        if not np.array_equal(rgn, rc):
            bad = np.argwhere(rgn != rc)
            raise SystemExit("%s: rewards differ at step %d in %d entries, first (env, agent) = %s: %r vs %r"
                             % (cfg, s, len(bad), bad[0], rgn[tuple(bad[0])], rc[tuple(bad[0])]))
        assert np.array_equal(dg.cpu().numpy(), dc), "%s: dones differ at step %d" % (cfg, s)
        if oc is not None:
            ogn = og.cpu().numpy()
            if not np.array_equal(ogn, oc):
                bad = np.argwhere(ogn != oc)
                raise SystemExit("%s: observations differ at step %d in %d entries, first %s" % (cfg, s, len(bad), bad[0]))
    assert dc.all(), "episode should have ended"
    for g, o in zip([x.cpu().numpy() for x in env.episode_stats()], ora.episode_stats()):
        assert np.array_equal(g, o), cfg + ": episode statistics differ"
    assert env.error_flags() == 0, env.error_flags()   # incl. bit 4 (capsule cores touching: a fallback normal) and bit 5 (non-finite solve)
    assert ora.degenerate() == 0 and ora.overflow() == 0
    extra = ""
    if cfg == "driving":
        c = env.debug_counters()
        tot = c["fast"] + c["quiescent"] + c["contact"] + c["steady"]
        extra = " paths: fast %.3f quiescent %.3f replay %.3f (light %.3f) full %.3f" % (
            c["fast"] / tot, c["quiescent"] / tot, (c["steady"] + c["light"]) / tot, c["light"] / tot, (c["contact"] - c["light"]) / tot)
    print("soak OK: %s nPlayers=%d, %d envs (global ids from %d) x %d steps bit-identical to the oracle (%.0f s)%s" % (cfg, n, E, env_id_offset, steps, time.time() - t0, extra), flush=True)
    env.close(); ora.close()


if __name__ == "__main__":
    # usage: soak_parity.py [cfg[:nPlayers[:envs]] ...]
    cfgs = sys.argv[1:] or ["driving", "robocup", "driving_partial", "robocup_partial"]
    for c in cfgs:
        parts = c.split(":")
        players = int(parts[1]) if len(parts) > 1 else None
        E = int(parts[2]) if len(parts) > 2 else (4096 if not parts[0].endswith("partial") else 1024)
        run(parts[0], E, 20261003, players)
