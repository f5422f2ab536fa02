"""Driving PARTIAL observation of the oracle (oracle/driving_partial.c) against the reference's own getAgentVision
(fixtures: tests/golden/gen_golden_partial.py): visibility radius, building occlusion, pedestrian interactions, REALISTIC
and RANDOM noise incl. false negatives / misclassification swap / random false positives / FP pedestrians, normalisation.
Noise magnitudes beyond the reference's 0..5 range are used on purpose so that the rare branches fire."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_golden import _state_from_npz  # noqa: F401  (same blob helpers)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _state(z, i, key):
    st = ol.DrivingState()
    cf, ci, pf, pi, ob, sc = (z["%s_%d" % (k, i)] for k in ("cars_f", "cars_i", "peds_f", "peds_i", "obst", "scalars"))
    st.elapsed, st.all_finished = int(sc[0]), int(sc[1])
    st.n_cars, st.n_peds, st.n_obst, st.episode = len(cf), len(pf), len(ob), int(key[2])
    for a in range(len(cf)):
        for n, v in zip(ol.CAR_F, cf[a]):
            setattr(st.cars[a], n, float(v))
        for n, v in zip(ol.CAR_I, ci[a]):
            setattr(st.cars[a], n, int(v))
    for a in range(len(pf)):
        for n, v in zip(ol.PED_F, pf[a]):
            setattr(st.peds[a], n, float(v))
        for n, v in zip(ol.PED_I, pi[a]):
            setattr(st.peds[a], n, int(v))
    for a in range(len(ob)):
        st.obst_x[a], st.obst_y[a] = float(ob[a][0]), float(ob[a][1])
    return st


def test_agent_vision_matches_reference(oracle_built):
    z = np.load(os.path.join(G, "driving_partial.npz"))
    key = [int(x) for x in z["key"]]
    total_rows = np.zeros(4)
    for i in range(int(z["n"][0])):
        n_players, ntype, magn, elapsed = z["cfg_%d" % i]
        env = ol.OracleEnv(env_type=1, num_envs=1, n_players=int(n_players), obs_type=1, noise_type=int(ntype),
                           noise_magnitude=float(magn), seed=key[0], env_id_offset=key[1])
        env.reset()
        st = _state(z, i, key)
        env.set_state(0, st)
        exp = z["rows_%d" % i]
        for a in range(int(n_players)):
            out = np.zeros(env.D, np.float32)
            env.l.oracle_drv_vision(env.h, 0, a, out.ctypes.data_as(C.c_void_p))
            np.testing.assert_array_equal(out[-4:], exp[a][-4:], err_msg="scene %d agent %d row counts" % (i, a))
            np.testing.assert_allclose(out, exp[a], rtol=0, atol=3e-5, err_msg="scene %d agent %d" % (i, a))
            total_rows += out[-4:]
        assert env.l.oracle_obs_overflow(env.h) == 0
    assert (total_rows > 20).all(), "fixtures must exercise every block (cars, obstacles, pedestrians, lanes)"
