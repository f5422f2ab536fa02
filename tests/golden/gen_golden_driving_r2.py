#!/usr/bin/env python3
"""Round-2 golden fixtures for Driving, from the reference's OWN Python (build container only; stand-ins and approach
as in gen_golden.py - read its docstring first).

  * `begin` collision callbacks carCrash / pedHit / carHit (DrivingEnvironment.py:591-683) called with a fake arbiter
    (an object with `.shapes`) over all lane-position x crashed x finished x speed x bearing branches; rewards,
    crashed/finished flags, the friction function, the pedestrian's dead/moving/velocity and the return value.
  * the reset composition: DrivingEnvironment.__init__ -> _setup_scene (:58-115, :527-584) with every `np.random.*` /
    `random.*` call served, in call order, from the Philox words the oracle's drv_reset draws for that call site
    (the partial Fisher-Yates behind `np.random.permutation(30)[:nPlayers]` is restated here: a permutation is an
    input of the composition, not a result of it).  What is compared: which spot/lane/road every car got, goals, types,
    the pedestrian and obstacle placements, the obstacle on-road filter, the counts, and the first observation.

Writes tests/golden/driving_callbacks.npz and tests/golden/driving_reset.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

RNG_RESET_AGENT, RNG_RESET_PERM, RNG_RESET_PED, RNG_RESET_OBST, RNG_RESET_COUNTS = 1, 2, 3, 4, 5


class Arb(object):
    pass


def gen_callbacks(out):
    env, de, cut = gg.make_driving(10, 21)
    rng = np.random.RandomState(17)
    LP = cut.LanePosition
    recs = {k: [] for k in ("kind", "slots", "ret", "rew")}
    states = {"b": [], "a": []}

    def scramble(trial):
        for i, car in enumerate(env.agents):
            b = car.shape.body
            b.position = gg.Vec2d(200 + rng.rand() * 1300, 150 + rng.rand() * 700)
            sp = [0.0, 0.6, 0.99, 1.01, 3.0, 12.0, 45.0][(trial + 3 * i) % 7]
            th = rng.rand() * 2 * np.pi
            b.velocity = gg.Vec2d(sp * np.cos(th), sp * np.sin(th))
            b.angle = (rng.rand() - 0.5) * 6
            b.angular_velocity = (rng.rand() - 0.5) * 0.5
            car.crashed = bool((trial + i) % 5 == 1)
            car.finished = car.crashed or bool((trial + 2 * i) % 7 == 3)
            car.position = [LP.InRightLane, LP.InOpposingLane, LP.OverRoad, LP.OffRoad, LP.InRightLane][(trial // 2 + i) % 5]
            b.velocity_func = cut.friction_car_crashed if car.crashed else cut.friction_car
        for k, ped in enumerate(env.pedestrians):
            b = ped.shape.body
            b.position = gg.Vec2d(200 + rng.rand() * 1300, 150 + rng.rand() * 700)
            b.velocity = gg.Vec2d((rng.rand() - 0.5) * 8, (rng.rand() - 0.5) * 8)
            ped.moving = [0, 10, 5000][(trial + k) % 3]
            ped.dead = bool((trial + k) % 11 == 4)
            if ped.dead:
                b.velocity_func = cut.friction_pedestrian_dead
            else:
                b.velocity_func = None
        env.elapsed = int(rng.randint(0, 5990))

    n_peds, n_obst = len(env.pedestrians), len(env.obstacles)
    for trial in range(360):
        scramble(trial)
        kind = trial % 3
        i = trial % 10
        car = env.agents[i]
        arb = Arb()
        if kind == 0:                       # carCrash, shapes in canonical order (lower slot first)
            j = (i + 1 + (trial // 3) % 9) % 10
            lo, hi = min(i, j), max(i, j)
            a, b2 = env.agents[lo], env.agents[hi]
            # aim half of the pairs at each other so that both bearing tests (< -0.4, > 0.4) fire
            if trial % 2 == 0:
                dp = a.shape.body.position - b2.shape.body.position
                if dp.length > 0:
                    u = dp / dp.length
                    a.shape.body.velocity = -u * max(a.shape.body.velocity.length, 0.5) if trial % 4 == 0 else u * a.shape.body.velocity.length
                    b2.shape.body.velocity = u * max(b2.shape.body.velocity.length, 0.5) if trial % 8 < 4 else -u * b2.shape.body.velocity.length
            arb.shapes = [a.shape, b2.shape]
            slots = [lo, hi]
            fn = env.carCrash
        elif kind == 1:                     # pedHit: shapes[0] = car, shapes[1] = pedestrian
            k = (trial // 3) % n_peds
            ped = env.pedestrians[k]
            if trial % 2 == 0:              # drive at the pedestrian
                dp = car.shape.body.position - ped.shape.body.position
                sp = car.shape.body.velocity.length
                car.shape.body.velocity = -(dp / dp.length) * sp
            arb.shapes = [car.shape, ped.shape]
            slots = [i, 10 + k]
            fn = env.pedHit
        else:                               # carHit: obstacle or building
            if (trial // 3) % 2 == 0 and n_obst:
                k = (trial // 6) % n_obst
                arb.shapes = [car.shape, env.obstacles[k].shape]
                slots = [i, 30 + k]
            else:
                k = (trial // 6) % 4
                arb.shapes = [car.shape, env.buildings[k].shape]
                slots = [i, 50 + k]
            fn = env.carHit
        before = gg.dump_state(env, cut)
        env.carRewards = np.array([0.0] * 10)
        ret = fn(arb, env.space, None)
        after = gg.dump_state(env, cut)
        states["b"].append(before); states["a"].append(after)
        recs["kind"].append(kind); recs["slots"].append(slots); recs["ret"].append(int(bool(ret)))
        recs["rew"].append(np.array(env.carRewards, float))
    for w in ("b", "a"):
        for key in ("cars_f", "cars_i", "peds_f", "peds_i", "obst", "scalars"):
            out["cb_%s_%s" % (w, key)] = np.array([s[key] for s in states[w]])
    out["cb_kind"] = np.array(recs["kind"], np.int64)
    out["cb_slots"] = np.array(recs["slots"], np.int64)
    out["cb_ret"] = np.array(recs["ret"], np.int64)
    out["cb_rew"] = np.array(recs["rew"])
    assert out["cb_ret"].min() == 0 and out["cb_ret"].max() == 1  # pedHit returns False for slow cars


class ResetTape(object):
    """Serves the constructor's draws from the oracle's Philox words, by call site (= call order) (oracle/driving.c
    drv_reset).  np.random.randint(lo, hi, n) is exclusive at hi; random.randint(lo, hi) inclusive."""

    def __init__(self, seed, genv, episode, n_players):
        self.k = (seed, genv, episode)
        self.A = n_players
        self.np_randint_calls = 0
        self.np_rand_calls = 0
        self.py_randint_calls = 0
        self.n_ped = self.n_obst_raw = None
        self.ped_speed_idx = 0

    def blk(self, purpose, entity, t=0):
        return gg.env_rng(self.k[0], self.k[1], self.k[2], purpose, entity, t)

    def np_randint(self, lo, hi, n):
        c = self.np_randint_calls
        self.np_randint_calls += 1
        if c < 4:     # _create_agents: roadSel, endSel, teams, types  <- words 0..3 of (RESET_AGENT, i)
            assert n == self.A
            return np.array([gg.randint_from(self.blk(RNG_RESET_AGENT, i)[c], lo, hi - 1) for i in range(n)])
        if c < 6:     # createRandomPedestrians: roadIds, sideIds <- words 0, 1 of (RESET_PED, i, 0)
            assert n == self.n_ped
            return np.array([gg.randint_from(self.blk(RNG_RESET_PED, i)[c - 4], lo, hi - 1) for i in range(n)])
        assert n == self.n_obst_raw  # createRandomObstacles
        return np.array([gg.randint_from(self.blk(RNG_RESET_OBST, i)[c - 6], lo, hi - 1) for i in range(n)])

    def np_rand(self, n):
        c = self.np_rand_calls
        self.np_rand_calls += 1
        purpose = RNG_RESET_PED if c < 2 else RNG_RESET_OBST
        return np.array([self.blk(purpose, i)[2 + (c % 2)] * 2.0 ** -32 for i in range(n)])

    def np_permutation(self, n):
        assert n == 30
        spots = list(range(30))
        for i in range(self.A):
            j = i + gg.randint_from(self.blk(RNG_RESET_PERM, i)[0], 0, 29 - i)
            spots[i], spots[j] = spots[j], spots[i]
        return np.array(spots)

    def py_randint(self, lo, hi):
        if (lo, hi) == (10, 20):
            c = self.py_randint_calls
            self.py_randint_calls += 1
            v = gg.randint_from(self.blk(RNG_RESET_COUNTS, 0)[0 if c == 0 else 1], lo, hi)
            if c == 0:
                self.n_ped = v
            else:
                self.n_obst_raw = v
            return v
        assert (lo, hi) == (3, 6)  # Pedestrian.py:33 speed  <- word 0 of (RESET_PED, i, 1)
        v = gg.randint_from(self.blk(RNG_RESET_PED, self.ped_speed_idx, 1)[0], lo, hi)
        self.ped_speed_idx += 1
        return v


def gen_reset(out):
    de, cut, pedm = gg.ref("DrivingEnvironment"), gg.ref("cutils"), gg.ref("Pedestrian")
    import random as pyrandom
    cases = [(10, 42, 0, 0), (10, 42, 7, 3), (2, 5, 1, 0), (7, 99, 4095, 12), (10, 7, 123, 1), (6, 1, 2, 2)]
    keys = []
    for ci, (A, seed, genv, episode) in enumerate(cases):
        tape = ResetTape(seed, genv, episode, A)
        orig = (np.random.randint, np.random.rand, np.random.permutation, pyrandom.randint)
        np.random.randint = lambda lo, hi=None, size=None: tape.np_randint(lo, hi, size)
        np.random.rand = lambda n: tape.np_rand(n)
        np.random.permutation = lambda n: tape.np_permutation(n)
        pyrandom.randint = tape.py_randint  # de.random / pedm.random are this same module object
        try:
            env = de.DrivingEnvironment(A, render=False, observationType=cut.ObservationType.FULL,
                                        noiseType=cut.NoiseType.REALISTIC, noiseMagnitude=0)
        finally:
            np.random.randint, np.random.rand, np.random.permutation, pyrandom.randint = orig
        st = gg.dump_state(env, cut)
        for k, v in st.items():
            out["reset%d_%s" % (ci, k)] = v
        dim = 9 + (A - 1) * 7 + 80 + 40 + 40
        out["reset%d_obs" % ci] = gg.flat_obs(env, env.get_full_obs(), A, dim)
        out["reset%d_counts" % ci] = np.array([tape.n_ped, tape.n_obst_raw, len(env.obstacles)], np.int64)
        keys.append([A, seed, genv, episode])
    out["reset_keys"] = np.array(keys, np.int64)


def main():
    gg.install_standins()
    cb = {}
    gen_callbacks(cb)
    np.savez_compressed(os.path.join(HERE, "driving_callbacks.npz"), **cb)
    rs = {}
    gen_reset(rs)
    np.savez_compressed(os.path.join(HERE, "driving_reset.npz"), **rs)
    print("wrote driving_callbacks.npz, driving_reset.npz")


if __name__ == "__main__":
    main()
