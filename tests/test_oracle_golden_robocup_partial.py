"""The RoboCup Partial-observation oracle (oracle/robocup_partial.c) against rows computed by the reference's own
getAgentVision (tests/golden/gen_golden_robocup_partial.py): detections, interactions, noise in both modes,
misclassification swaps, random false positives, false-positive balls near robots, polar/line conversion, list lengths and
the (numLandMarks, robotsSeen, ballsSeen) tuple.  Transcendentals differ from CPython's libm by <= 1 ulp (dynenv_math.h),
hence the float32-level tolerance."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_golden_robocup import _to_state

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_robocup_partial_rows_match_reference(oracle_built):
    z = np.load(os.path.join(G, "robocup_partial.npz"))
    lib = ol.lib()
    lib.oracle_rc_partial_obs.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    n_scenes = z["cfg"].shape[0]
    worst = 0.0
    for s in range(n_scenes):
        n_players, ntype, magn, seed, genv, episode, tkey = z["cfg"][s]
        env = ol.OracleEnv(env_type=0, num_envs=1, n_players=int(n_players), obs_type=1, noise_type=int(ntype),
                           noise_magnitude=float(magn), seed=int(seed), env_id_offset=int(genv), flags=ol.ROBOCUP_DEFAULT_FLAGS)
        env.reset()
        env.set_state(0, _to_state(z["rf"][s], z["ri"][s], z["sc"][s], z["fl"][s], int(episode)))
        R = 2 * int(n_players)
        D = env.D
        out = np.zeros((R, D), np.float32)
        lib.oracle_rc_partial_obs(env.h, 0, int(tkey), out.ctypes.data_as(C.c_void_p))
        want = z["rows"][s][:R]
        assert want.shape[1] == D
        tail = D - 17
        # list lengths, numLandMarks, ballsSeen, robotsSeen: exact
        np.testing.assert_array_equal(out[:, tail:], want[:, tail:], err_msg="scene %d: counts / seen tuple" % s)
        np.testing.assert_allclose(out[:, :tail], want[:, :tail], rtol=2e-6, atol=2e-6, err_msg="scene %d rows" % s)
        worst = max(worst, float(np.abs(out[:, :tail] - want[:, :tail]).max()))
        env.close()
    assert worst < 2e-6


def test_process_seens_matches_reference(oracle_built):
    z = np.load(os.path.join(G, "robocup_partial.npz"))
    lib = ol.lib()
    lib.oracle_rc_process_seens.argtypes = [C.c_double, C.POINTER(C.c_double), C.c_int, C.c_double]
    lib.oracle_rc_process_seens.restype = C.c_double
    for inp, out in zip(z["seens_in"], z["seens_out"]):
        R = int(out[0])
        for a in range(R):
            rs = (C.c_double * 9)(*inp[a, 1:10])
            got = lib.oracle_rc_process_seens(float(inp[a, 0]), rs, R - 1, float(inp[a, 10]))
            assert got == out[1 + a], (R, a, got, out[1 + a])
