"""include/dynenv_math.h (host build, through the oracle library): accuracy vs glibc/numpy, edge cases, Philox KATs."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol


@pytest.fixture(scope="module", autouse=True)
def _built(oracle_built):
    return oracle_built


def _ulps(a, b):
    return np.abs(a - b) / np.maximum(np.spacing(np.abs(b)), 5e-324)


def test_sincos_atan2_sqrt_within_2_ulp_of_libm():
    rng = np.random.default_rng(0)
    n = 400000
    x = np.concatenate([(rng.random(n // 2) - 0.5) * 2000.0, (rng.random(n // 2) - 0.5) * 8.0])
    y = np.concatenate([(rng.random(n // 2) - 0.5) * 2000.0, (rng.random(n // 2) - 0.5) * 1e-3])
    out = np.zeros((n, 5))
    ol.lib().oracle_math(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p))
    assert _ulps(out[:, 0], np.sin(x)).max() <= 2.0
    assert _ulps(out[:, 1], np.cos(x)).max() <= 2.0
    assert _ulps(out[:, 2], np.arctan2(y, x)).max() <= 2.0
    assert np.array_equal(out[:, 3], np.sqrt(np.abs(x)))  # correctly rounded
    assert np.array_equal(out[:, 4], x / y)


def test_special_angles_used_by_the_scene_constants():
    x = np.array([0.0, -0.0, 90.0, -90.0, -0.0, 0.0, 1.0])
    y = np.array([90.0, -90.0, 0.0, -0.0, -0.0, 0.0, 1.0])
    # atan2(y, x) as used by Road.getSpot's spotDir.angle (Road.py:114): math.atan2 semantics incl. signed zeros
    xs = np.array([0.0, -0.0, 90.0, -90.0, -90.0, 90.0])
    ys = np.array([90.0, -90.0, 0.0, -0.0, 0.0, -0.0])
    out = np.zeros((len(xs), 5))
    ol.lib().oracle_math(xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), len(xs), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out[:, 2], np.arctan2(ys, xs))
    ang = np.array([np.pi / 2, -np.pi / 2, -np.pi, np.pi, 0.0, 2 * np.pi / 180])
    out = np.zeros((len(ang), 5))
    ol.lib().oracle_math(ang.ctypes.data_as(C.c_void_p), ang.ctypes.data_as(C.c_void_p), len(ang), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out[:, 0], np.sin(ang)) and np.array_equal(out[:, 1], np.cos(ang))


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    l = ol.lib()
    cases = [((0, 0), (0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
             ((0xffffffff, 0xffffffff), (0xffffffff,) * 4, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
             ((0xa4093822, 0x299f31d0), (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
              (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    l.oracle_philox.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    for key, ctr, exp in cases:
        c = np.array(ctr, np.uint32)
        o = np.zeros(4, np.uint32)
        l.oracle_philox(key[0], key[1], c.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        assert tuple(int(v) for v in o) == exp
