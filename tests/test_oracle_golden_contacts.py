"""The oracle against trajectories of the reference's OWN `DrivingEnvironment.step()` WITH collisions (tests/golden/gen_golden_contacts.py:
the reference's Python - processAction, tick, move, carCrash / pedHit / carHit, the friction velocity functions, rewards, getFullState -
on a functional pymunk facade over tests/kat_general.py, the independent GJK / EPA restatement of Chipmunk).  Unlike the free-flight
fixtures these contain crashes, pushes, resting contacts and a killed pedestrian: they pin the COMPOSITION of callbacks and physics inside
a step.  Tolerances: rewards / states 1e-9 (relative to max(1, |value|)), observations 2e-6 (float32), flags and dones exact."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_golden import _state_from_npz

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = list("abcdefghijklmn")   # k: one whole episode, terminal step included; l, m: every car reaches its goal (allFinished, team reward); n: all but one


def _close(got, want, tol, msg):
    got, want = np.asarray(got, float), np.asarray(want, float)
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.size == 0 or float(err.max()) <= tol, "%s: deviation %.3e at %s" % (msg, err.max(), np.unravel_index(err.argmax(), err.shape))


def check_trajectory(z, tag, make_env):
    """make_env(n_players, seed, env_id_offset) -> (set_state(st), step(actions [A, 2]) -> (obs [A, D], rewards [A], done), get_state())"""
    key = z["%s_key" % tag]
    st0 = _state_from_npz(z, tag, "init", key)
    made = make_env(st0.n_cars, int(key[0]), int(key[1]))
    set_state, step, get_state = made[:3]
    set_state(st0)
    acts, marks = z["%s_actions" % tag], list(z["%s_state_steps" % tag])
    every = int(z["%s_obs_every" % tag][0]) if "%s_obs_every" % tag in z else 1   # (the whole-episode fixture keeps every 10th observation)
    n_obs = 0
    for s in range(len(acts)):
        o, r, d = step(acts[s].astype(np.int32))
        _close(r, z["%s_rewards" % tag][s], 1e-9, "%s: rewards of step %d" % (tag, s))
        assert int(d) == int(z["%s_dones" % tag][s])
        if s % every == every - 1 or s == len(acts) - 1:
            np.testing.assert_allclose(o, z["%s_obs" % tag][n_obs], rtol=0, atol=2e-6, err_msg="%s: observation of step %d" % (tag, s))
            n_obs += 1
        if int(d):   # the terminal step: info['episode_r'], ['episode_p_r'], ['episode_o_r'], ['episode_g'] (DrivingEnvironment.py:309-316)
            assert s == len(acts) - 1 == 599 and len(made) == 4
            er, ep, eo, g = made[3]()
            want = z["%s_episode_info" % tag]
            _close(er, want[0], 1e-9, tag + ": episode_r"); _close(ep, want[1], 1e-9, tag + ": episode_p_r"); _close(eo, want[2], 0.0, tag + ": episode_o_r")
            np.testing.assert_array_equal(np.asarray(g, float), z["%s_episode_goals" % tag])
        if s in marks:
            k = marks.index(s)
            got = ol.state_to_dict(get_state())
            _close(got["cars_f"], z["%s_states_cars_f" % tag][k], 1e-9, "%s: cars after step %d" % (tag, s))
            _close(got["peds_f"], z["%s_states_peds_f" % tag][k], 1e-9, "%s: pedestrians after step %d" % (tag, s))
            np.testing.assert_array_equal(got["cars_i"], z["%s_states_cars_i" % tag][k], err_msg="%s: car flags after step %d" % (tag, s))
            np.testing.assert_array_equal(got["peds_i"], z["%s_states_peds_i" % tag][k], err_msg="%s: pedestrian flags after step %d" % (tag, s))
            _close(got["episode_r"], z["%s_states_episode_r" % tag][k], 1e-9, "%s: episode rewards after step %d" % (tag, s))
            _close(got["episode_pos_r"], z["%s_states_episode_pos_r" % tag][k], 1e-9, "%s: positive episode rewards after step %d" % (tag, s))
            if "%s_states_scalars" % tag in z:   # elapsed, allFinished
                np.testing.assert_array_equal(got["scalars"][:2], z["%s_states_scalars" % tag][k], err_msg="%s: elapsed / allFinished after step %d" % (tag, s))


def test_the_fixtures_contain_what_they_are_for():
    z = np.load(os.path.join(G, "driving_contacts.npz"))
    crashed = sum(int(z["%s_states_cars_i" % t][-1][:, 3].sum()) for t in TAGS)      # CAR_I: type team finished crashed lane_pos fric
    dead = sum(int(z["%s_states_peds_i" % t][-1][:, 2].sum()) for t in TAGS)         # PED_I: road side dead ...
    begins = sum(int(z["%s_begins_per_step" % t].sum()) for t in TAGS)
    assert crashed >= 15 and dead >= 2 and begins >= 30, (crashed, dead, begins)
    # l, m: all three cars finished without a crash, allFinished set, and the step in which the last one arrived carries the team reward
    # (maxTime - elapsed) / 100 ~ 59 for every car (DrivingEnvironment.py:281-285, :301); n: one car short, no such step
    for t in "lm":
        assert z[t + "_states_scalars"][-1][1] == 1 and z[t + "_states_cars_i"][-1][:, 2].all() and not z[t + "_states_cars_i"][-1][:, 3].any()
        assert (z[t + "_rewards"] > 50).all(1).sum() == 1 and (z[t + "_rewards"].sum(0) > 115).all(), "one step with the team reward for everybody, on top of each car's own arrival reward"
    assert z["n_states_scalars"][-1][1] == 0 and not (z["n_rewards"] > 50).all(1).any()


@pytest.mark.parametrize("tag", TAGS)
def test_reference_step_with_collisions_against_the_oracle(oracle_built, tag):
    z = np.load(os.path.join(G, "driving_contacts.npz"))

    def make_env(n_players, seed, offset):
        env = ol.OracleEnv(num_envs=1, n_players=n_players, seed=seed, env_id_offset=offset)
        env.reset()

        def step(a):
            o, r, d = env.step(a[None])
            return o[0, 0], r[0], d[0]

        def stats():
            r, p, o, g = env.episode_stats()
            return r[0], p[0], o[0], g[0]
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0)), stats
    check_trajectory(z, tag, make_env)


# ------------------------------------------------------------------------------------------------ RoboCup
RC_TAGS = list("abcdefghijklmn")   # l: the last ten steps of an episode, terminal step included; m: two penalties expire; n: a walking robot trips
RC_MIN_STEPS = {"a": 30, "b": 40, "c": 50, "d": 30, "e": 12, "f": 20, "g": 25, "h": 40, "i": 30, "j": 60, "k": 15, "l": 10, "m": 6, "n": 5}   # steps of each trajectory that are well-conditioned (and checked)


def _rc_check_state(st, rf, ri, sc, fl, msg, tol=1e-9):
    n = int(sc[1])
    got_f = np.array([[getattr(st.robots[i], k) for k in ol.ROBOT_F] for i in range(n)])
    got_i = np.array([[getattr(st.robots[i], k) for k in ol.ROBOT_I] for i in range(n)])
    _close(got_f, rf[:n], tol, msg + ": robot floats")
    np.testing.assert_array_equal(got_i, ri[:n], err_msg=msg + ": robot flags")
    assert [st.elapsed, st.ball_owned, st.n_last_kicked] == [int(sc[0]), int(sc[2]), int(sc[3])], msg
    assert list(st.last_kicked)[:st.n_last_kicked] == [int(x) for x in sc[4:4 + int(sc[3])]], msg + ": lastKicked"
    assert [st.goals[0], st.goals[1], st.closest[0], st.closest[1]] == [int(x) for x in sc[8:12]], msg + ": goals / closest"
    assert [st.n_def[0], st.n_def[1]] == [int(sc[12]), int(sc[13])], msg + ": defenders"
    _close([st.ball_free_cntr, st.grace_period, st.penal_times[0], st.penal_times[1], st.bpx, st.bpy, st.bvx, st.bvy, st.bw, st.bprevx,
            st.bprevy], fl, tol, msg + ": ball / timers")


RCP_TAIL = 793 - 17   # oracle/robocup_partial.h: the row's last 17 entries are list lengths and the seen tuple (exact)


RCP_OFF_LINE = 32 * 5 + 20 * 7 + 16 * 6 + 16 * 6 + 28 * 8   # oracle/robocup_partial.h: the line block [12][5] of a Partial row


def _own_line_excused(got, want, tol):
    """A penalized robot stands exactly ON its side line (its penalty spot), so that line passes through the robot's own position and
    cutils.isLineInArea (:751-828) returns either nothing or a zero-length line AT the robot - which of the two is decided by the last
    bit of libm's sin / cos of the field-of-view edges (the oracle's are within 2 ulp of glibc's, tests/test_detmath.py, not identical).
    True if two rows differ by exactly that: one more line on one side, that line degenerate (normalized distance -1), everything else equal."""
    gt, wt = got[RCP_TAIL:], want[RCP_TAIL:]
    d = gt - wt
    if not (abs(d[5]) == 1 and d[6] == d[5] and not np.delete(d, [5, 6]).any()):
        return False
    lng, sht = (got, want) if d[5] > 0 else (want, got)
    nl = int(lng[RCP_TAIL + 5])
    a, b = lng[RCP_OFF_LINE:RCP_OFF_LINE + 5 * nl].reshape(nl, 5), sht[RCP_OFF_LINE:RCP_OFF_LINE + 5 * (nl - 1)].reshape(nl - 1, 5)
    for j in range(nl):
        if a[j][0] < -0.999 and np.allclose(np.delete(a, j, 0), b, rtol=2e-6, atol=max(2e-6, 10 * tol)):
            return np.allclose(got[:RCP_OFF_LINE], want[:RCP_OFF_LINE], rtol=2e-6, atol=max(2e-6, 10 * tol))
    return False


def check_robocup_trajectory(z, tag, make_env, partial=False, own_line_slack=None):
    """make_env(n, seed, offset, flags[, noise magnitude]) -> (set_state(st), step(actions [R, 4]) -> (obs [5, R, D], rewards [R], done), get_state())
    own_line_slack (a list, Partial only; tools/reference_step_fuzz.py): rows that differ by `_own_line_excused` are appended to it as (step, snapshot,
    robot) instead of failing; such a robot's rewards (processSeens: 0.0025 x the mean landmark count of the five snapshots) get 0.0026 of slack"""
    from test_oracle_golden_robocup import _to_state
    n, can_fall, seed, genv, episode, _, _ = [int(x) for x in z[tag + "_meta"]]
    flags = (ol.FLAG_CAN_FALL if can_fall else 0) | ol.FLAG_USE_OBS_REWARDS
    set_state, step, get_state = make_env(n, seed, genv, flags, float(z[tag + "_noise"][1])) if partial else make_env(n, seed, genv, flags)
    set_state(_to_state(z[tag + "_b_rf"], z[tag + "_b_ri"], z[tag + "_b_sc"], z[tag + "_b_fl"], episode))
    acts, marks, R = z[tag + "_actions"], list(z[tag + "_state_steps"]), 2 * n
    # The fixture carries its own conditioning: per window of steps (up to each recorded state), how far a twin of the reference run had
    # drifted whose velocities got a relative 1e-15 nudge before every physics substep.  RoboCup's contact phases amplify rounding by orders
    # of magnitude now and then (nearly parallel feet, duplicate end-cap contacts under friction 6.25; profiles/r05_kat_general_fuzz.txt):
    # the tolerance of a window is 1e-9 or 1000 x the twin's largest drift in it, and where the twin is off by more than 1e-6 the
    # trajectory pins nothing any more: the check ends.
    cond = list(z[tag + "_conditioning"])
    checked = 0
    mine = []    # this trajectory's excused rows
    for s in range(len(acts)):
        k = min(i for i, m in enumerate(marks) if m >= s)
        if cond[k] > 1e-6:
            break
        tol = max(1e-9, 1e3 * cond[k])
        o, r, d = step(acts[s].astype(np.int32))
        want = z[tag + "_obs"][s][:, :R, :o.shape[-1]]
        if partial and own_line_slack is not None:
            o, want, r = np.array(o), np.array(want), np.array(r)
            for t in range(o.shape[0]):
                for a in range(R):
                    if not np.array_equal(o[t, a, RCP_TAIL:], want[t, a, RCP_TAIL:]) and _own_line_excused(o[t, a], want[t, a], tol):
                        own_line_slack.append((s, t, a)); mine.append((s, t, a))
                        o[t, a] = want[t, a]
            slack = sorted(set(a for (s_, _, a) in mine))          # (episode sums carry a deviation on)
            for a in slack:
                if abs(r[a] - z[tag + "_rewards"][s][a]) <= 0.0026:
                    r[a] = z[tag + "_rewards"][s][a]
        _close(r, z[tag + "_rewards"][s], tol, "%s: rewards of step %d" % (tag, s))
        assert int(d) == int(z[tag + "_dones"][s])
        if partial:   # list lengths and the seen tuple exact, rows to float32 accuracy (as tests/test_oracle_golden_robocup_partial.py)
            np.testing.assert_array_equal(o[..., RCP_TAIL:], want[..., RCP_TAIL:], err_msg="%s: list lengths / seen tuple of step %d" % (tag, s))
            np.testing.assert_allclose(o[..., :RCP_TAIL], want[..., :RCP_TAIL], rtol=2e-6, atol=max(2e-6, 10 * tol), err_msg="%s: Partial observation of step %d" % (tag, s))
        else:
            np.testing.assert_allclose(o, want, rtol=0, atol=max(2e-6, 10 * tol), err_msg="%s: observation (5 snapshots) of step %d" % (tag, s))
        if s in marks:
            st = get_state()
            _rc_check_state(st, z[tag + "_states_rf"][k], z[tag + "_states_ri"][k], z[tag + "_states_sc"][k], z[tag + "_states_fl"][k],
                            "%s: state after step %d" % (tag, s), tol)
            # episodeRewards / episodePosRewards (info['episode_r'] / ['episode_p_r'] of the terminal step, RoboCupEnvironment.py:516-521)
            if not mine:
                _close(list(st.episode_r)[:R] + list(st.episode_pos_r)[:R], z[tag + "_episode"][s], max(tol, 1e-9) * (s + 1), "%s: episode sums after step %d" % (tag, s))
        checked = s + 1
    return checked


def test_the_robocup_fixtures_contain_what_they_are_for():
    z = np.load(os.path.join(G, "robocup_contacts.npz"))
    assert list(z["l_dones"]) == [0] * 9 + [1] and int(z["l_states_sc"][-1][0]) == 12000, "fixture l ends with the episode's terminal step"
    # ROBOT_I[1] = penalized, [5] = fallen.  m: robots 1 and 7 start penalized and are back in play at the first recorded state (tick's
    # un-penalize branch, RoboCupEnvironment.py:948-968); n: robot 7 falls in a step without any robot-robot touch (processAction :556-560)
    assert z["m_b_ri"][[1, 7], 1].all() and not z["m_states_ri"][0][:, 1].any()
    assert z["n_states_ri"][-1][7, 5] == 1 and z["n_begins"][0] == 0
    begins = sum(z[t + "_begins"] for t in RC_TAGS)     # robot-robot, robot-ball, robot-post, ball-post, own feet
    assert begins[0] >= 30 and begins[1] >= 6 and begins[2] >= 5 and begins[3] >= 1, begins
    kicking = sum(int(z[t + "_states_ri"][:, :, 7].sum()) for t in RC_TAGS)          # ROBOT_I[7] = kicking (pivot joint removed mid-kick)
    assert kicking >= 100, kicking
    fallen = sum(int(z[t + "_states_ri"][:, :, 5].max(0).sum()) for t in RC_TAGS)   # ROBOT_I[5] = fallen
    assert fallen >= 2, fallen


@pytest.mark.parametrize("tag", RC_TAGS)
def test_reference_robocup_step_with_collisions_against_the_oracle(oracle_built, tag):
    z = np.load(os.path.join(G, "robocup_contacts.npz"))

    def make_env(n, seed, offset, flags):
        env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, seed=seed, env_id_offset=offset, flags=flags)
        env.reset()

        def step(a):
            o, r, d = env.step(a[None])
            return o[0], r[0], d[0]
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))
    assert check_robocup_trajectory(z, tag, make_env) >= RC_MIN_STEPS[tag]


# ------------------------------------------------------------------------------------------------ Driving, Partial observations (configs[3])
P_TAGS = ["a", "b", "c", "d"]   # d: pedestrians within 20 px of an obstacle's centre: the `Nearby` noise multiplier (cutils.py:514-515) on 36 rows


def check_partial_trajectory(z, tag, make_env):
    """make_env(n_players, seed, offset, noise_magnitude) -> (set_state, step(actions [A, 2]) -> (obs [A, 517], rewards [A], done), get_state)"""
    key = z["%s_key" % tag]
    n_players, _, magn = z["%s_cfg" % tag]
    st0 = _state_from_npz(z, tag, "init", key)
    set_state, step, get_state = make_env(int(n_players), int(key[0]), int(key[1]), float(magn))
    set_state(st0)
    acts = z["%s_actions" % tag]
    rows = np.zeros(4)
    for s in range(len(acts)):
        o, r, d = step(acts[s].astype(np.int32))
        _close(r, z["%s_rewards" % tag][s], 1e-9, "%s: rewards of step %d" % (tag, s))
        assert int(d) == int(z["%s_dones" % tag][s])
        want = z["%s_obs" % tag][s]
        np.testing.assert_array_equal(o[:, -4:], want[:, -4:], err_msg="%s: row counts of step %d" % (tag, s))
        np.testing.assert_allclose(o, want, rtol=0, atol=3e-5, err_msg="%s: Partial observation of step %d" % (tag, s))   # (the tolerance of test_oracle_golden_partial.py)
        rows += o[:, -4:].sum(0)
    got = ol.state_to_dict(get_state())
    _close(got["cars_f"], z["%s_final_cars_f" % tag], 1e-9, tag + ": final cars")
    _close(got["peds_f"], z["%s_final_peds_f" % tag], 1e-9, tag + ": final pedestrians")
    np.testing.assert_array_equal(got["cars_i"], z["%s_final_cars_i" % tag])
    np.testing.assert_array_equal(got["peds_i"], z["%s_final_peds_i" % tag])
    return rows


@pytest.mark.parametrize("tag", P_TAGS)
def test_reference_step_with_partial_observations_and_collisions_against_the_oracle(oracle_built, tag):
    """BASELINE configs[3] as the reference runs it: `step()` with ObservationType.PARTIAL, NoiseType.REALISTIC, noiseMagnitude 3 -
    getAgentVision of every agent inside the step, on top of physics with crashes"""
    z = np.load(os.path.join(G, "driving_partial_contacts.npz"))

    def make_env(n_players, seed, offset, magn):
        env = ol.OracleEnv(env_type=1, num_envs=1, n_players=n_players, obs_type=1, noise_type=1, noise_magnitude=magn, seed=seed, env_id_offset=offset)
        env.reset()

        def step(a):
            o, r, d = env.step(a[None])
            return o[0, 0], r[0], d[0]
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))
    rows = check_partial_trajectory(z, tag, make_env)
    assert (rows > 50).all(), rows      # cars, obstacles, pedestrians, lanes seen over the trajectory
    assert tag != "d" or int(z["d_nearby_rows"][0]) >= 30


RCP_TAGS = ["a", "b", "c"]
RCP_MIN_STEPS = {"a": 20, "b": 20, "c": 25}


@pytest.mark.parametrize("tag", RCP_TAGS)
def test_reference_robocup_step_with_partial_observations_against_the_oracle(oracle_built, tag):
    """SURVEY section 8 a17 / f3 as the reference runs it: `RoboCupEnvironment.step()` with ObservationType.PARTIAL + Realistic noise 3 -
    getAgentVision of every agent at the five snapshots inside the step, processSeens' observation rewards folded into the step's rewards"""
    z = np.load(os.path.join(G, "robocup_partial_contacts.npz"))

    def make_env(n, seed, offset, flags, magn):
        env = ol.OracleEnv(env_type=0, num_envs=1, n_players=n, obs_type=1, noise_type=1, noise_magnitude=magn, seed=seed, env_id_offset=offset, flags=flags)
        env.reset()

        def step(a):
            o, r, d = env.step(a[None])
            return o[0], r[0], d[0]
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))
    assert check_robocup_trajectory(z, tag, make_env, partial=True) >= RCP_MIN_STEPS[tag]
