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


def test_fused_sincos_of_the_observation_code_within_2_ulp_of_libm():
    """dm_sincos_f (the vision passes' variant: Horner chains of fma) against glibc, and against the unfused dm_sincos: the two
    may differ in the last place, not more."""
    rng = np.random.default_rng(3)
    n = 400000
    x = np.ascontiguousarray(np.concatenate([(rng.random(n // 2) - 0.5) * 2000.0, (rng.random(n // 2) - 0.5) * 8.0]))
    out = np.zeros((n, 2))
    ol.lib().oracle_math_f(x.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p))
    assert _ulps(out[:, 0], np.sin(x)).max() <= 2.0
    assert _ulps(out[:, 1], np.cos(x)).max() <= 2.0
    ref = np.zeros((n, 5))
    ol.lib().oracle_math(x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), n, ref.ctypes.data_as(C.c_void_p))
    assert _ulps(out[:, 0], ref[:, 0]).max() <= 2.0 and _ulps(out[:, 1], ref[:, 1]).max() <= 2.0
    ang = np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, 2 * np.pi / 180])
    o = np.zeros((len(ang), 2))
    ol.lib().oracle_math_f(ang.ctypes.data_as(C.c_void_p), len(ang), o.ctypes.data_as(C.c_void_p))
    assert _ulps(o[:, 0], np.sin(ang)).max() <= 1.0 and _ulps(o[:, 1], np.cos(ang)).max() <= 1.0


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


def _fma(a, b, c):
    """a * b + c with one rounding, elementwise: exact rational arithmetic, then Fraction -> float (correctly rounded)"""
    from fractions import Fraction

    def one(x, y, z):
        if not (np.isfinite(x) and np.isfinite(y) and np.isfinite(z)):
            return x * y + z
        r = Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z))
        if r == 0:
            return x * y + z  # the sign of a zero result follows the IEEE rule of the unfused expression here
        try:
            return float(r)
        except OverflowError:
            return np.inf if r > 0 else -np.inf
    a, b, c = np.broadcast_arrays(np.asarray(a, float), np.asarray(b, float), np.asarray(c, float))
    return np.array([one(x, y, z) for x, y, z in zip(a.ravel(), b.ravel(), c.ravel())]).reshape(a.shape)


def _fdlibm_atan_pos(ax):
    """fdlibm __atan for ax >= 0 in its branching form (one arm per range), the polynomial's and the last range's multiply-adds
    fused as include/dynenv_math.h fuses them; evaluated with numpy (IEEE double) and an exact fma"""
    hi_t = np.array([4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00])
    lo_t = np.array([2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17])
    aT = [3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
          9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
          4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02]
    idx = np.where(ax < 0.4375, -1, np.where(ax < 0.6875, 0, np.where(ax < 1.1875, 1, np.where(ax < 2.4375, 2, 3))))
    t = np.where(idx == -1, ax, np.where(idx == 0, (2.0 * ax - 1.0) / (2.0 + ax), np.where(idx == 1, (ax - 1.0) / (ax + 1.0),
                 np.where(idx == 2, (ax - 1.5) / _fma(1.5, ax, 1.0), -1.0 / ax))))
    z = t * t
    w = z * z
    s1 = z * _fma(w, _fma(w, _fma(w, _fma(w, _fma(w, aT[10], aT[8]), aT[6]), aT[4]), aT[2]), aT[0])
    s2 = w * _fma(w, _fma(w, _fma(w, _fma(w, aT[9], aT[7]), aT[5]), aT[3]), aT[1])
    hi, lo = hi_t[np.maximum(idx, 0)], lo_t[np.maximum(idx, 0)]
    r = np.where(idx < 0, t - t * (s1 + s2), hi - (_fma(t, s1 + s2, -lo) - t))
    r = np.where(ax < 3.7252902984619140625e-09, ax, r)
    return np.where(ax >= 1.0e300, hi_t[3] + lo_t[3], r)


def _fdlibm_atan2(y, x):
    pi, pi_lo = 3.1415926535897931160e+00, 1.2246467991473531772e-16
    aq = np.abs(y / x)
    z = np.where(aq > 1.0e300, 1.5707963267948966, np.where((x < 0.0) & (aq < 1.0e-300), 0.0, _fdlibm_atan_pos(aq)))
    sx, sy = np.signbit(x), np.signbit(y)
    r = np.where(~sx, np.where(sy, -z, z), np.where(~sy, pi - (z - pi_lo), (z - pi_lo) - pi))
    r = np.where(x == 0.0, np.where(sy, -1.5707963267948966, 1.5707963267948966), r)
    return np.where(y == 0.0, np.where(sx, np.where(sy, -pi, pi), y), r)


def test_atan2_is_the_branching_fdlibm_evaluation_bit_for_bit():
    """dm_atan2 selects numerator, denominator, hi and lo instead of branching over the five ranges (one division instead of
    five arms on a diverging wavefront); it must stay the same arithmetic: bit patterns against the branching form, over the
    simulators' ranges, the range boundaries +- a few ulp, extreme magnitudes, zeros and infinities.  (The first range's
    t - t s: the header computes 0 - (fma(t, s, -0) - t), and fma(t, s, -0) = t s rounded once = the product.)"""
    rng = np.random.default_rng(5)
    n = 4000
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e300, -1e300, 1e-300, -1e-300, 5e-324, -5e-324, 1e-310, 0.4375, 0.6875,
                   1.1875, 2.4375, 3.7252902984619140625e-09, 2.0 ** -28, 1e308, 90.0, -90.0, 1e-9, 1e150, 1e-150])
    gx, gy = np.meshgrid(sp, sp)
    sgn = lambda: rng.choice([-1.0, 1.0], n)
    bx = (rng.random(n) + 0.5) * sgn()
    by = rng.choice([0.4375, 0.6875, 1.1875, 2.4375], n) * (1 + rng.integers(-3, 4, n) * 2.2e-16) * bx * sgn()
    x = np.concatenate([gx.ravel(), (rng.random(n) - 0.5) * 2000.0, (rng.random(n) - 0.5) * 8.0, np.exp((rng.random(n) - 0.5) * 1400) * sgn(), rng.choice(sp, n), bx])
    y = np.concatenate([gy.ravel(), (rng.random(n) - 0.5) * 2000.0, (rng.random(n) - 0.5) * 1e-3, np.exp((rng.random(n) - 0.5) * 1400) * sgn(), (rng.random(n) - 0.5) * 8.0, by])
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    out = np.zeros((len(x), 5))
    ol.lib().oracle_math(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p))
    with np.errstate(all="ignore"):
        ref = _fdlibm_atan2(y, x)
    got = out[:, 2]
    same = (got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), [(y[i], x[i], got[i], ref[i]) for i in np.flatnonzero(~same)[:5]]
