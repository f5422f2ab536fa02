"""GPU parity tests proper: the HIP path (through the C ABI, via dynenv_amd) against the CPU oracle on the same seeds.

Bar (BASELINE.json north_star): reward/done bit-exact, float positions/velocities <= 1e-4 relative.  Because both
sides use the deterministic math header with FMA contraction off, we assert the stronger property: EVERYTHING is
bit-identical (observations, rewards, dones and the full state blob).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import dynenv_amd
    from dynenv_amd import _capi
    ol.build()
    return dynenv_amd, _capi, torch


def _assert_state_equal(sg, so, msg=""):
    dg, do = ol.state_to_dict(sg), ol.state_to_dict(so)
    for k in do:
        np.testing.assert_array_equal(dg[k], do[k], err_msg="%s field %s" % (msg, k))


def test_detmath_bitwise_device_vs_host(gpu):
    _, _capi, _ = gpu
    lib = _capi.load()
    rng = np.random.default_rng(1)
    n = 1 << 18
    x = np.concatenate([(rng.random(n // 2) - 0.5) * 2000.0, (rng.random(n // 2) - 0.5) * 8.0])
    y = np.concatenate([(rng.random(n // 2) - 0.5) * 2000.0, (rng.random(n // 2) - 0.5) * 1e-3])
    x[:8] = [0.0, -0.0, np.pi / 2, -np.pi / 2, -np.pi, np.pi, 1e-10, 90.0]
    y[:8] = [1.0, -1.0, 0.0, -0.0, -0.0, 0.0, 1e300, -0.0]
    dev = np.zeros((n, 5))
    host = np.zeros((n, 5))
    rc = lib.dynenv_math_selftest(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), n,
                                  dev.ctypes.data_as(C.c_void_p), 0)
    assert rc == 0, lib.dynenv_last_error()
    ol.lib().oracle_math(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), n, host.ctypes.data_as(C.c_void_p))
    assert np.array_equal(dev.view(np.uint64), host.view(np.uint64)), \
        "sincos/atan2/sqrt/div must be bit-identical on gfx950 and the host (columns: %s)" % \
        np.nonzero((dev.view(np.uint64) != host.view(np.uint64)).any(0))[0]


@pytest.mark.parametrize("A,E,seed", [(10, 64, 42), (2, 8, 5), (7, 16, 9)])
def test_reset_parity(gpu, A, E, seed):
    dynenv_amd, _, _ = gpu
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=seed)
    ora = ol.OracleEnv(num_envs=E, n_players=A, seed=seed)
    og = env.reset_flat().cpu().numpy()
    oc = ora.reset()
    np.testing.assert_array_equal(og, oc)
    np.testing.assert_array_equal(env.counts().cpu().numpy(), ora.counts())
    for e in range(0, E, max(1, E // 8)):
        _assert_state_equal(env.get_state(e), ora.get_state(e), "env %d" % e)
    env.close()


@pytest.mark.parametrize("A,E,seed,steps", [(10, 64, 42, 120), (2, 8, 5, 100), (10, 32, 1234, 600)])
def test_step_parity_random_actions(gpu, A, E, seed, steps):
    """configs[0] (nPlayers=2) and configs[1] (nPlayers=10) at oracle-sized batches; the 600-step case crosses a
    whole episode incl. the terminal step."""
    dynenv_amd, _, _ = gpu
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=seed)
    ora = ol.OracleEnv(num_envs=E, n_players=A, seed=seed, threads=8)
    env.reset_flat()
    ora.reset()
    rng = np.random.default_rng(seed)
    contacts = 0
    for s in range(steps):
        a = rng.integers(0, 3, size=(E, A, 2)).astype(np.int32)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        oc, rc, dc = ora.step(a)
        contacts += sum(ora.active_contacts(e) > 0 for e in range(E))
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="rewards step %d" % s)
        np.testing.assert_array_equal(dg.cpu().numpy(), dc, err_msg="dones step %d" % s)
        np.testing.assert_array_equal(og.cpu().numpy(), oc, err_msg="obs step %d" % s)
        if s % 20 == 19 or s == steps - 1:
            for e in range(0, E, max(1, E // 4)):
                _assert_state_equal(env.get_state(e), ora.get_state(e), "step %d env %d" % (s, e))
    assert env.error_flags() == 0 and ora.overflow() == 0
    if steps >= 600:
        assert contacts > 0, "the long run must exercise the contact solver"
        assert dc.all()
        sg = [x.cpu().numpy() for x in env.episode_stats()]
        so = ora.episode_stats()
        for g, o in zip(sg, so):
            np.testing.assert_array_equal(g, o)
    env.close()


def _scenario(kind):
    """Hand-built collision scenes (set_state on both sides): analytically meaningful contact cases."""
    st = ol.DrivingState()
    st.n_cars, st.n_peds, st.n_obst, st.episode = 10, 2, 2, 1
    for i in range(10):  # park the unused cars far apart on the vertical road's right lane
        c = st.cars[i]
        c.px, c.py, c.angle = 892.5, 40.0 + 95.0 * i, np.pi / 2
        c.dirx, c.diry = 6.123233995736766e-17, 1.0
        c.prevx, c.prevy, c.goalx, c.goaly = c.px, c.py, 875.0, 1000.0
        c.type, c.lane_pos = i % 4, 4
    for k in range(2):
        p = st.peds[k]
        p.px, p.py, p.road, p.side, p.speed, p.moving = 700.0 + 30 * k, 440.0, 1, 0, 4, 100000
    st.obst_x[0], st.obst_y[0] = 200.0, 440.0
    st.obst_x[1], st.obst_y[1] = 1500.0, 560.0
    a, b = st.cars[0], st.cars[1]
    if kind == "head_on":  # two cars on the horizontal road driving into each other
        a.px, a.py, a.angle, a.vx, a.vy = 400.0, 517.5, 0.0, 60.0, 0.0
        a.dirx, a.diry = 1.0, 0.0
        b.px, b.py, b.angle, b.vx, b.vy = 480.0, 517.5, -np.pi, -45.0, 0.0
        b.dirx, b.diry = -1.0, -1.2246467991473532e-16
    elif kind == "oblique":  # rotated box hits a moving box off-centre -> spin + 1 or 2 contact points
        a.px, a.py, a.angle, a.vx, a.vy = 400.0, 510.0, 0.3, 50.0, 10.0
        a.dirx, a.diry = np.cos(0.3), np.sin(0.3)
        b.px, b.py, b.angle, b.vx, b.vy = 455.0, 530.0, 2.0, -30.0, -5.0
        b.dirx, b.diry = np.cos(2.0), np.sin(2.0)
    elif kind == "obstacle":  # car slides into a static obstacle box and comes to rest against it
        a.px, a.py, a.angle, a.vx, a.vy = 160.0, 442.0, 0.0, 90.0, 0.0
        a.dirx, a.diry = 1.0, 0.0
    elif kind == "building":  # car leaves the road into a building wall (rests there => persistent contacts)
        a.px, a.py, a.angle, a.vx, a.vy = 500.0, 440.0, -np.pi / 2 + 0.2, 5.0, -40.0
        a.dirx, a.diry = np.cos(a.angle), np.sin(a.angle)
    elif kind == "ped_fast":  # fast car kills a pedestrian (pedHit returns True: solved) and crashes
        a.px, a.py, a.angle, a.vx, a.vy = 660.0, 482.5, 0.0, 55.0, 0.0
        a.dirx, a.diry = 1.0, 0.0
        st.peds[0].py = 484.0
    elif kind == "ped_slow":  # slow car touches a pedestrian: pedHit returns False -> ignored until separation
        a.px, a.py, a.angle, a.vx, a.vy = 683.0, 482.5, 0.0, 0.9, 0.0
        a.dirx, a.diry = 1.0, 0.0
        st.peds[0].py = 484.0
        st.peds[0].vx = -1.0
    elif kind == "pileup":  # three cars in a row + the building: chained arbiters share bodies (multi-level solve)
        a.px, a.py, a.angle, a.vx, a.vy = 300.0, 480.0, 0.0, 70.0, -8.0
        a.dirx, a.diry = 1.0, 0.0
        b.px, b.py, b.angle, b.vx, b.vy = 345.0, 478.0, 0.1, 0.0, 0.0
        b.dirx, b.diry = np.cos(0.1), np.sin(0.1)
        c = st.cars[2]
        c.px, c.py, c.angle, c.vx, c.vy = 388.0, 474.0, -0.4, -70.0, -3.0
        c.dirx, c.diry = np.cos(-0.4), np.sin(-0.4)
    for c in (st.cars[0], st.cars[1], st.cars[2]):
        c.prevx, c.prevy = c.px, c.py
    return st


@pytest.mark.parametrize("kind", ["head_on", "oblique", "obstacle", "building", "ped_fast", "ped_slow", "pileup"])
def test_collision_scenarios(gpu, kind):
    dynenv_amd, _, _ = gpu
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 2, 10, seed=3)
    ora = ol.OracleEnv(num_envs=2, n_players=10, seed=3)
    env.reset_flat()
    ora.reset()
    st = _scenario(kind)
    for e in range(2):
        env.set_state(e, st)
        ora.set_state(e, st)
    acts = np.ones((2, 10, 2), np.int32)  # coast
    touched = 0
    for s in range(40):
        if s == 10 and kind != "ped_slow":
            acts[:, 0, 0] = 2  # accelerate into whatever we are resting against
        og, rg, dg = env.step_flat(acts, auto_reset=False)
        oc, rc, dc = ora.step(acts)
        touched += ora.active_contacts(0)
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="%s rewards step %d" % (kind, s))
        np.testing.assert_array_equal(og.cpu().numpy(), oc, err_msg="%s obs step %d" % (kind, s))
        _assert_state_equal(env.get_state(0), ora.get_state(0), "%s step %d" % (kind, s))
    if kind != "ped_slow":
        assert touched > 0, "scenario %s never produced an active contact" % kind
    if kind == "pileup":  # chained arbiters with moving bodies: the split-lane sweeps (drv_solve_general_split) must have run here
        assert env.debug_counters()["split"] > 0, "the pile-up never took the general multi-level solver"
    assert env.error_flags() == 0
    env.close()


def test_full_size_properties(gpu):
    """configs[1] at BASELINE size (4096 envs): size-independent properties over a whole episode + auto-reset."""
    dynenv_amd, _, torch = gpu
    E, A = 4096, 10
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=42)
    obs0 = env.reset_flat().clone()
    st0 = [env.get_state(e) for e in (0, 1000, 4095)]
    g = torch.Generator(device="cuda").manual_seed(0)
    total = torch.zeros((E, A), dtype=torch.float64, device="cuda")
    for s in range(600):
        a = torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32)
        obs, rew, done = env.step_flat(a, auto_reset=False)
        total += rew
        assert bool(done.all()) == (s == 599)
    assert torch.isfinite(obs).all() and torch.isfinite(total).all()
    ep_r, ep_p, _, goals = env.episode_stats()
    # the per-step rewards telescope into the episode accumulator (DrivingEnvironment.py:300-304)
    assert torch.allclose(ep_r, total, rtol=0, atol=1e-9)
    assert bool((ep_p >= 0).all())
    assert int(goals.sum(1).max()) <= A
    assert env.error_flags() == 0
    # spot-check three envs of the big batch against the oracle run on the same global env ids (shard invariance)
    for st, e in zip(st0, (0, 1000, 4095)):
        ora = ol.OracleEnv(num_envs=1, n_players=A, seed=42, env_id_offset=e)
        ora.reset()
        _assert_state_equal(st, ora.get_state(0), "env %d at reset" % e)
    # lock-step auto reset returns the next episode's first observation
    env.reset_flat()
    assert not torch.equal(env.obs, obs0)
    env.close()


def test_vec_env_compat_surface(gpu):
    """make_dyn_env drop-in: same call shape and return structure as the reference's SubprocVecEnv path."""
    dynenv_amd, _, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType, make_dyn_env
    venv, name = make_dyn_env(DynEnvType.DRIVE, 3, 2, False, ObservationType.FULL, NoiseType.REALISTIC, 0, False)
    assert name == "Driving" and venv.num_envs == 3
    obs = venv.reset()
    assert obs.shape == (3, 1, 2, 3) and obs.dtype == object
    cars, obst, peds = obs[0, 0, 0, 0]
    selfr, lanes = obs[0, 0, 0, 1]
    assert cars.shape == (1, 7) and selfr.shape == (1, 9) and lanes.shape == (8, 5) and obst.shape[1] == 4 and peds.shape[1] == 2
    assert obs[0, 0, 0, 2] == (1, 1, 1)
    acts = np.ones((3, 2, 2), np.int64)
    obs, rew, dones, infos = venv.step(acts)
    assert rew.shape == (3, 2) and rew.dtype == np.float64 and dones.shape == (3,) and len(infos) == 3
    assert set(infos[0]) >= {"Full State", "Recon States"} and infos[0]["Full State"][0].shape == (2, 7)
    with pytest.raises(Exception):
        venv.step(np.ones((3, 2, 3), np.int64))
    assert venv.get_attr("stepNum")[0] == 600.0
    assert len(venv.env_method("get_agent_locs")[0]) == 2
    assert venv.action_space.spaces[0].nvec.tolist() == [3, 3]
    venv.close()


# ------------------------------------------------------------------------------------------------ RoboCup (configs[2])
def _assert_rc_state_equal(sg, so, msg=""):
    dg, do = ol.rc_state_to_dict(sg), ol.rc_state_to_dict(so)
    for k in do:
        if k == "scalars":  # defenders are a set on the device (reported ascending); compare everything else exactly
            continue
        np.testing.assert_array_equal(dg[k], do[k], err_msg="%s field %s" % (msg, k))
    assert [sg.elapsed, sg.ball_owned, sg.n_last_kicked, sg.episode] == [so.elapsed, so.ball_owned, so.n_last_kicked, so.episode], msg
    assert list(sg.last_kicked)[:sg.n_last_kicked] == list(so.last_kicked)[:so.n_last_kicked], msg
    assert list(sg.goals) == list(so.goals) and list(sg.closest) == list(so.closest), msg
    for t in range(2):
        assert sorted(list(sg.defenders[t])[:sg.n_def[t]]) == sorted(list(so.defenders[t])[:so.n_def[t]]), msg + " defenders"


def _rc_actions(rng, E, A):
    return np.stack([rng.integers(0, 5, (E, A)), rng.integers(0, 3, (E, A)), rng.integers(0, 3, (E, A)),
                     rng.integers(0, 7, (E, A))], -1).astype(np.int32)


@pytest.mark.parametrize("n,E,seed", [(5, 32, 42), (2, 8, 7), (1, 4, 3)])
def test_robocup_reset_parity(gpu, n, E, seed):
    dynenv_amd, _, _ = gpu
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, E, n, seed=seed)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=n, seed=seed, flags=ol.ROBOCUP_DEFAULT_FLAGS)
    og = env.reset_flat().cpu().numpy()
    oc = ora.reset()
    assert og.shape == oc.shape == (E, 5, 2 * n, 12 + (2 * n - 1) * 6)
    np.testing.assert_array_equal(og, oc)
    for e in range(0, E, max(1, E // 4)):
        _assert_rc_state_equal(env.get_state(e), ora.get_state(e), "env %d" % e)
    env.close()


@pytest.mark.parametrize("n,E,seed,steps,flags", [(5, 32, 42, 60, None), (2, 8, 7, 60, None), (5, 16, 11, 240, None),
                                                   (5, 16, 5, 60, 0)])
def test_robocup_step_parity(gpu, n, E, seed, steps, flags):
    """configs[2]: RoboCup nPlayers=5, Full obs, 50 substeps (canFall=True like the reference default; flags=0 turns
    the dice off = the RNG-free configuration of SURVEY Appendix D)."""
    dynenv_amd, _, _ = gpu
    fl = ol.ROBOCUP_DEFAULT_FLAGS if flags is None else flags
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, E, n, seed=seed, flags=fl if fl else 8)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=n, seed=seed, flags=fl if fl else 8, threads=8)
    env.reset_flat()
    ora.reset()
    rng = np.random.default_rng(seed)
    contacts = 0
    for s in range(steps):
        a = _rc_actions(rng, E, 2 * n)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        oc, rc, dc = ora.step(a)
        contacts += sum(ora.active_contacts(e) > 0 for e in range(E))
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="rewards step %d" % s)
        np.testing.assert_array_equal(dg.cpu().numpy(), dc, err_msg="dones step %d" % s)
        np.testing.assert_array_equal(og.cpu().numpy(), oc, err_msg="obs step %d" % s)
        if s % 10 == 9 or s == steps - 1:
            for e in range(0, E, max(1, E // 4)):
                _assert_rc_state_equal(env.get_state(e), ora.get_state(e), "step %d env %d" % (s, e))
    assert env.error_flags() == 0 and ora.overflow() == 0
    if steps >= 240:
        assert dc.all() and contacts > 0
        for g, o in zip([x.cpu().numpy() for x in env.episode_stats()], ora.episode_stats()):
            np.testing.assert_array_equal(g, o)
    env.close()


# ------------------------------------------------------------------------------------------------ Driving Partial (configs[3])
@pytest.mark.parametrize("noise_type,magn,E,steps", [(1, 3.0, 32, 40), (0, 3.0, 8, 20), (1, 0.0, 8, 20), (1, 5.0, 16, 60)])
def test_driving_partial_obs_parity(gpu, noise_type, magn, E, steps):
    """configs[3]: Driving nPlayers=10, Partial obs, Realistic noise magnitude 3 (+ RANDOM noise, no noise, max noise):
    the ragged getAgentVision rows (dense padded + counts) are bit-identical to the oracle."""
    dynenv_amd, _, _ = gpu
    from dynenv_amd import NoiseType, ObservationType
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, 10, observationType=ObservationType.PARTIAL,
                                   noiseType=NoiseType(noise_type), noiseMagnitude=magn, seed=42)
    ora = ol.OracleEnv(env_type=1, num_envs=E, n_players=10, obs_type=1, noise_type=noise_type, noise_magnitude=magn,
                       seed=42, threads=8)
    assert env.obs_dim == ora.D == 517
    og = env.reset_flat().cpu().numpy()
    oc = ora.reset()
    np.testing.assert_array_equal(og, oc, err_msg="reset obs")
    rng = np.random.default_rng(1)
    rows = np.zeros(4)
    for s in range(steps):
        a = rng.integers(0, 3, size=(E, 10, 2)).astype(np.int32)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        oc, rc, dc = ora.step(a)
        og = og.cpu().numpy()
        bad = np.argwhere(og != oc)
        assert len(bad) == 0, "obs step %d first mismatch at %s: %r vs %r" % (s, bad[0], og[tuple(bad[0])], oc[tuple(bad[0])])
        np.testing.assert_array_equal(rg.cpu().numpy(), rc)
        rows += oc[..., -4:].reshape(-1, 4).sum(0)
    assert env.error_flags() == 0 and ora.l.oracle_obs_overflow(ora.h) == 0
    assert (rows > 0).all()
    env.close()


def _crowded_view_scene(st, rng):
    """Everything within sight of car 0 at the road crossing and nothing hidden: twenty pedestrians in a ring around it, twenty
    obstacles in a wider ring, the other nine cars (crashed, standing) along the two roads behind those - a pedestrian is only
    occluded by something nearer than itself."""
    import kat_scenes_r4 as ks
    cx, cy = 875.0 + float(rng.uniform(-20, 20)), 500.0 + float(rng.uniform(-20, 20))
    ks._place_crashed_car(st.cars[0], 0, cx, cy, 0.0)
    st.n_peds, st.n_obst = 20, 20
    for i in range(20):
        a, r = 2 * np.pi * (i + float(rng.uniform(0, 0.5))) / 20, float(rng.uniform(40.0, 70.0))
        p = st.peds[i]
        p.px, p.py, p.vx, p.vy = cx + r * np.cos(a), cy + r * np.sin(a), 0.0, 0.0
        p.road, p.side, p.dead, p.moving, p.speed, p.crossing, p.begin_crossing = i & 1, 0, 1, 0, 4, 0, 0
        a, r = 2 * np.pi * (i + 0.5 + float(rng.uniform(0, 0.4))) / 20, float(rng.uniform(100.0, 150.0))
        st.obst_x[i], st.obst_y[i] = cx + r * np.cos(a), cy + r * np.sin(a)
    spots = [(cx + d, cy) for d in (-420.0, -330.0, -240.0, 240.0, 330.0, 420.0)] + [(cx, cy + d) for d in (-330.0, -240.0, 240.0)]
    for k in range(1, st.n_cars):
        x, y = spots[k - 1]
        ks._place_crashed_car(st.cars[k], int(rng.integers(0, 4)), x + float(rng.uniform(-5, 5)), y + float(rng.uniform(-5, 5)), 0.0)


@pytest.mark.parametrize("noise_type", [1, 0])
def test_driving_partial_crowded_view_parity(gpu, noise_type):
    """A view with 55+ of the 56 possible rows drawing noise: the HIP pass then has fewer than ten lanes left to carry the
    random-false-positive trials' Philox blocks and falls back to drawing them separately (driving_partial.hip, phase 5) - a
    path random scenes never take (the oracle's instrumentation confirms it was taken here)."""
    dynenv_amd, _, _ = gpu
    from dynenv_amd import NoiseType, ObservationType
    E, A = 48, 10
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, observationType=ObservationType.PARTIAL,
                                   noiseType=NoiseType(noise_type), noiseMagnitude=3.0, seed=77)
    ora = ol.OracleEnv(env_type=1, num_envs=E, n_players=A, obs_type=1, noise_type=noise_type, noise_magnitude=3.0, seed=77, threads=8)
    env.reset_flat()
    ora.reset()
    rng = np.random.default_rng(31)
    for e in range(E):
        st = ora.get_state(e)
        _crowded_view_scene(st, rng)
        env.set_state(e, st)
        ora.set_state(e, st)
    acts = np.ones((E, A, 2), np.int32)
    fp_rows = 0
    for s in range(6):
        og, rg, dg = env.step_flat(acts, auto_reset=False)
        oc, rc, dc = ora.step(acts)
        og = og.cpu().numpy()
        bad = np.argwhere(og != oc)
        assert len(bad) == 0, "obs step %d first mismatch at %s: %r vs %r" % (s, bad[0], og[tuple(bad[0])], oc[tuple(bad[0])])
        np.testing.assert_array_equal(rg.cpu().numpy(), rc)
    ora.l.oracle_obs_noise_draws_max.restype = C.c_int
    assert ora.l.oracle_obs_noise_draws_max(ora.h) >= 55, ora.l.oracle_obs_noise_draws_max(ora.h)
    assert env.error_flags() == 0 and ora.l.oracle_obs_overflow(ora.h) == 0
    env.close()


def test_partial_compat_view(gpu):
    dynenv_amd, _, _ = gpu
    from dynenv_amd import DynEnvType, NoiseType, ObservationType, make_dyn_env
    venv, name = make_dyn_env(DynEnvType.DRIVE, 2, 10, False, ObservationType.PARTIAL, NoiseType.REALISTIC, 3, False)
    obs = venv.reset()
    cars, obst, peds = obs[0, 0, 0, 0]
    selfr, lanes = obs[0, 0, 0, 1]
    assert cars.shape[1] == 7 and obst.shape[1] == 6 and peds.shape[1] == 2 and selfr.shape == (1, 9) and lanes.shape[1] == 4
    obs, rew, dones, infos = venv.step(np.ones((2, 10, 2), np.int64))
    assert obs.shape == (2, 1, 10, 3)
    venv.close()


# ------------------------------------------------------------------------------------------------ RoboCup Partial (a17)
@pytest.mark.parametrize("n,E,seed,noise,magn,steps", [(5, 16, 42, 1, 3.0, 40), (5, 8, 7, 0, 5.0, 25), (2, 8, 9, 1, 5.0, 25),
                                                       (1, 4, 3, 1, 2.0, 15), (5, 8, 11, 1, 0.0, 15)])
def test_robocup_partial_parity(gpu, n, E, seed, noise, magn, steps):
    """RoboCup Partial observation (getAgentVision at the five snapshots of every step) + the
    sighting-based observation rewards (processSeens): rows, rewards, dones and episode statistics identical to the oracle."""
    dynenv_amd, _, _ = gpu
    from dynenv_amd import NoiseType, ObservationType
    fl = ol.ROBOCUP_DEFAULT_FLAGS
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.ROBO_CUP, E, n, observationType=ObservationType.PARTIAL,
                                   noiseType=NoiseType(noise), noiseMagnitude=magn, seed=seed, flags=fl)
    ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=n, obs_type=1, noise_type=noise, noise_magnitude=magn, seed=seed,
                       flags=fl, threads=8)
    og = env.reset_flat().cpu().numpy()
    oc = ora.reset()
    assert og.shape == oc.shape and og.shape[-1] == 793
    np.testing.assert_array_equal(og, oc, err_msg="reset observations")
    rng = np.random.default_rng(seed)
    seen_rewards = False
    for s in range(steps):
        a = _rc_actions(rng, E, 2 * n)
        og, rg, dg = env.step_flat(a, auto_reset=False)
        oc, rc, dc = ora.step(a)
        np.testing.assert_array_equal(og.cpu().numpy(), oc, err_msg="obs step %d" % s)
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="rewards step %d" % s)
        np.testing.assert_array_equal(dg.cpu().numpy(), dc, err_msg="dones step %d" % s)
    assert env.error_flags() == 0 and ora.overflow() == 0
    for g, o in zip([x.cpu().numpy() for x in env.episode_stats()], ora.episode_stats()):
        np.testing.assert_array_equal(g, o)
    obs_r = env.episode_stats()[2].cpu().numpy()
    assert (obs_r > 0).any(), "observation rewards should have been paid"
    env.close()


def _resting_chain_scene(st, rng):
    """2..4 crashed cars in a row (random types, lateral offsets, overlaps above and below the collision slop), optionally a dead
    pedestrian wedged in and an obstacle at the end: resting contacts that share bodies - the two- and three-level solves a launch's
    slowest environments run (drv_solve_bias_multi / drv_solve_general_split), in every arrangement of who is shape a and shape b."""
    import kat_scenes_r4 as ks
    n = int(rng.integers(2, 5))
    st.n_peds, st.n_obst = 0, 0
    x, cy = 300.0 + float(rng.uniform(0, 50)), ks.CY
    order = list(rng.permutation(n))         # which car index stands where: canonical pair order != spatial order
    prev_hx = None
    for pos, k in enumerate(order):
        t = int(rng.integers(0, 4))
        hx = [10.0, 15.0, 20.0, 25.0][t]
        if prev_hx is not None:
            x += prev_hx + hx - float(rng.choice([0.03, 0.5, 1.5, 3.0]))
        ks._place_crashed_car(st.cars[k], t, x, cy + float(rng.uniform(-3.0, 3.0)), float(rng.choice([0.0, 0.0, 0.0, 2.0])) if pos == 0 else 0.0)
        prev_hx = hx
    for k in range(n, st.n_cars):
        ks._place_crashed_car(st.cars[k], 0, 1500.0, 480.0 + 45.0 * k, 0.0)
    if rng.random() < 0.6:
        st.n_obst = 1
        st.obst_x[0], st.obst_y[0] = x + prev_hx + 10.0 - float(rng.choice([0.03, 0.8, 2.0])), cy + float(rng.uniform(-6.0, 6.0))
    if rng.random() < 0.3:   # a dead pedestrian (a circle, dynamic, shape a of its pair) leaning on the first car
        st.n_peds = 1
        p = st.peds[0]
        c = st.cars[order[0]]
        p.px, p.py, p.vx, p.vy = c.px - [10.0, 15.0, 20.0, 25.0][c.type] - 5.0 + float(rng.choice([0.05, 0.7])), c.py, 0.0, 0.0
        p.road, p.side, p.dead, p.moving, p.speed, p.crossing, p.begin_crossing = 1, 0, 1, 0, 4, 0, 0


def test_resting_chains_parity(gpu):
    dynenv_amd, _, _ = gpu
    E, A = 96, 5
    env = dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=9)
    ora = ol.OracleEnv(num_envs=E, n_players=A, seed=9, threads=8)
    env.reset_flat()
    ora.reset()
    rng = np.random.default_rng(2024)
    for e in range(E):
        st = ora.get_state(e)
        _resting_chain_scene(st, rng)
        env.set_state(e, st)
        ora.set_state(e, st)
    acts = np.ones((E, A, 2), np.int32)
    multi = 0
    for s in range(12):
        og, rg, dg = env.step_flat(acts, auto_reset=False)
        oc, rc, dc = ora.step(acts)
        np.testing.assert_array_equal(rg.cpu().numpy(), rc, err_msg="rewards step %d" % s)
        bad = np.argwhere(og.cpu().numpy() != oc)
        assert len(bad) == 0, "observations differ at step %d, first (env, t, agent, column) = %s" % (s, bad[0])
        for e in range(E):
            _assert_state_equal(env.get_state(e), ora.get_state(e), "step %d env %d" % (s, e))
        multi += sum(ora.active_contacts(e) >= 2 for e in range(E))
    assert multi > 200, "the scenes must keep several arbiters active at once (%d)" % multi
    assert env.error_flags() == 0 and ora.overflow() == 0
    env.close()
