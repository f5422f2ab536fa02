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


_CFGS = {
    # name: (oracle env_type, n_players, action highs, steps before the checkpoint, flags, Partial?)
    "driving": (1, 10, [3, 3], 230, 0, False),
    "driving_partial": (1, 10, [3, 3], 230, 0, True),
    "robocup": (0, 5, [5, 3, 3, 7], 60, None, False),
    "robocup_partial": (0, 5, [5, 3, 3, 7], 60, None, True),
}


@pytest.mark.parametrize("cfg", sorted(_CFGS))
def test_checkpoint_restore_continues_like_the_oracle(cfg, oracle_built):
    """f4 against the ORACLE, not against the HIP path itself: the oracle runs alongside from the reset and never sees the
    checkpoint.  The HIP handle is checkpointed at step k with live contacts in its cache, stepped on (so that its device
    state moves away), restored - on the same handle and into a fresh handle built with another seed - and must then produce
    the oracle's observations, rewards, dones and state blobs of steps k+1.. bit for bit."""
    import torch
    ol = oracle_built
    from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType
    oet, n, hi, k, flags, partial = _CFGS[cfg]
    if flags is None:
        flags = ol.ROBOCUP_DEFAULT_FLAGS
    E, m = 48, 30
    kw, okw = {}, {}
    if partial:
        kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
        okw = dict(obs_type=1, noise_type=1, noise_magnitude=3.0)
    et = DynEnvType.DRIVE if oet == 1 else DynEnvType.ROBO_CUP
    env = BatchedDynEnv(et, E, n, seed=21, flags=flags, **kw)
    ora = ol.OracleEnv(env_type=oet, num_envs=E, n_players=n, seed=21, flags=flags, threads=8, **okw)
    assert np.array_equal(env.reset_flat().cpu().numpy(), ora.reset())
    rng = np.random.default_rng(17)
    A = env.n_agents

    def draw():
        return np.stack([rng.integers(0, h, (E, A)) for h in hi], -1).astype(np.int32)

    for s in range(k):
        a = draw()
        og, rg, dg = env.step_flat(torch.tensor(a, device="cuda"))
        oc, rc, dc = ora.step(a)
        if s % 10 == 9 or s == k - 1:
            assert np.array_equal(rg.cpu().numpy(), rc) and np.array_equal(og.cpu().numpy(), oc), "before the checkpoint, step %d" % s
    if oet == 1:
        assert env.debug_counters()["slot_sum"] > 0 and max(ora.active_contacts(e) for e in range(E)) > 0, \
            "the checkpoint should be taken with live contacts"
    ck = env.checkpoint()
    acts = [draw() for _ in range(m)]
    want = []
    for a in acts:  # the oracle's continuation: the reference for everything below
        oc, rc, dc = ora.step(a)
        want.append((oc.copy(), rc.copy(), dc.copy()))
    to_dict = ol.state_to_dict if oet == 1 else ol.rc_state_to_dict
    want_states = [to_dict(ora.get_state(e)) for e in (0, E // 2, E - 1)]

    def check(h, what):
        for s, (a, (oc, rc, dc)) in enumerate(zip(acts, want)):
            og, rg, dg = h.step_flat(torch.tensor(a, device="cuda"))
            assert np.array_equal(rg.cpu().numpy(), rc), "%s: rewards, step k+%d" % (what, s + 1)
            assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc), "%s: dones, step k+%d" % (what, s + 1)
            assert np.array_equal(og.cpu().numpy(), oc), "%s: observations, step k+%d" % (what, s + 1)
        for e, w in zip((0, E // 2, E - 1), want_states):
            got = to_dict(h.get_state(e))
            for f in w:
                np.testing.assert_array_equal(got[f], w[f], err_msg="%s: state of environment %d, field %s" % (what, e, f))
        assert h.error_flags() == 0

    check(env, "uninterrupted run")       # (moves the handle's device state m steps past the checkpoint)
    env.restore(ck)
    check(env, "restored on the same handle")
    env2 = BatchedDynEnv(et, E, n, seed=4242, flags=flags, **kw)
    env2.restore(ck)
    check(env2, "restored into a fresh handle")
    for x in (env, env2):
        x.close()
    ora.close()


@pytest.mark.parametrize("E", [4096, 8192, 3600])
def test_checkpoint_restore_with_the_isolation_scheduler_running(E, monkeypatch):
    """the SIMD-isolation lists are scheduling scratch, not state.  Run until environments are being isolated
    (4096) / started first (8192), checkpoint, go on for n = 1, 2, 3 more steps (every phase of the three-list rotation), restore
    on the same handle and into a fresh one, and compare every later step with a handle that never schedules
    (DYNENV_NO_ISOLATION) and never saw a checkpoint.  A list that kept ids across the restore would step some environments
    twice and others not at all."""
    import torch
    from dynenv_amd import BatchedDynEnv, DynEnvType
    A = 10
    iso = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=11)
    if iso.debug_counters()["isolation_mode"] == 0:
        pytest.skip("the scheduler is only switched on for 4096 or more environments on a 256-CU device")
    monkeypatch.setenv("DYNENV_NO_ISOLATION", "1")
    ref = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=11)
    monkeypatch.delenv("DYNENV_NO_ISOLATION")
    fresh = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=77)
    iso.reset_flat(); ref.reset_flat(); fresh.reset_flat()
    g = torch.Generator(device="cuda").manual_seed(5)
    draw = lambda: torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32)
    s = 0
    while True:
        a = draw()
        iso.step_flat(a, auto_reset=False); ref.step_flat(a, auto_reset=False)
        s += 1
        if s % 10 == 0 and s >= 200 and iso.debug_counters()["isolated_next"] > 0:
            break
        assert s < 500, "no environment was ever isolated"
    for n in (1, 2, 3):
        ck = iso.checkpoint()
        assert len(ck) == len(ref.checkpoint()), "scheduling scratch must not be part of a checkpoint"
        pre = [draw() for _ in range(n)]
        post = [draw() for _ in range(8)]
        for a in pre:
            iso.step_flat(a, auto_reset=False)      # the scheduler moves on: lists filled / rotated n more times
        iso.restore(ck)
        fresh.restore(ck)
        assert iso.debug_counters()["isolated_next"] == 0 and fresh.debug_counters()["isolated_next"] == 0
        for j, a in enumerate(pre + post):
            o0, r0, d0 = ref.step_flat(a, auto_reset=False)
            for h, what in ((iso, "same handle"), (fresh, "fresh handle")):
                o, r, d = h.step_flat(a, auto_reset=False)
                assert torch.equal(r, r0) and torch.equal(d, d0) and torch.equal(o, o0), "n=%d, %s, step %d after the restore" % (n, what, j)
        for e in (0, 1023, 1024, min(4095, E - 1), E - 1):
            assert bytes(iso.get_state(e)) == bytes(ref.get_state(e)) == bytes(fresh.get_state(e)), "state of environment %d" % e
    assert iso.debug_counters()["isolation_timeouts"] == 0 and fresh.debug_counters()["isolation_timeouts"] == 0
    for x in (iso, ref, fresh):
        assert x.error_flags() == 0
        x.close()


@pytest.mark.parametrize("kind", ["driving", "robocup"])
def test_abi2_checkpoint_of_a_partial_handle_keeps_what_its_error_bit_reported(kind):
    """ABI 2 kept "invalid action" and, for Partial observations, "rows dropped" in ONE bit (1) of the per-environment error word,
    which is part of the checkpointed arrays; ABI 3 moved the second to bit 3.  An ABI 2 blob with that bit set loads as BOTH (bits 1 | 3)
    on a Partial handle - nothing that was reported goes unreported -, an ABI 3 blob as it is."""
    import struct
    import dynenv_amd as d
    kw = dict(observationType=d.ObservationType.PARTIAL, noiseType=d.NoiseType.REALISTIC, noiseMagnitude=3)
    et, n, k = (d.DynEnvType.DRIVE, 10, 2) if kind == "driving" else (d.DynEnvType.ROBO_CUP, 5, 4)
    env = d.BatchedDynEnv(et, 8, n, seed=3, **kw)
    env.reset_flat()
    a = np.zeros((8, 10, k), np.int32)
    if k == 4:
        a[..., 3] = 3
    a[5, 1, 0] = 9                         # outside the action space: error bit 1
    env.step_flat(a, auto_reset=False)
    assert env.error_flags() == 2
    blob3 = env.checkpoint()
    assert bytes(blob3[:8]) == b"DYNCKPT2" and struct.unpack_from("<i", blob3, 8)[0] == 3
    blob2 = blob3.copy()
    struct.pack_into("<i", blob2, 8, 2)    # the header of a library of ABI 2; no array and no layout changed between the two
    other = d.BatchedDynEnv(et, 8, n, seed=3, **kw)
    other.reset_flat()
    other.restore(blob3)
    assert other.error_flags() == 2
    other.restore(blob2)
    assert other.error_flags() == 2 | 8
    full = d.BatchedDynEnv(et, 8, n, seed=3)       # Full observations: bit 1 never meant anything else
    full.reset_flat()
    full.step_flat(a, auto_reset=False)
    b = full.checkpoint()
    struct.pack_into("<i", b, 8, 2)
    full.restore(b)
    assert full.error_flags() == 2
    for e in (env, other, full):
        e.close()
