/* dynenv_math.h — deterministic fp64 elementary functions + Philox4x32-10.
 *
 * Why this exists: the DynEnv hot path compares fp64 quantities against thresholds
 * (`abs(x) < factor` cutils.py:123, `cos(...) < -0.4` DrivingEnvironment.py:627,660,
 * `relAngle < 0` Road.py:95-96).  glibc's and the ROCm device library's sin/cos/atan2 differ
 * in the last ulp, which flips those flags between host and gfx950.  Every routine here is
 * built from IEEE-754 +,-,*,/, sqrt and explicit fma only (all correctly rounded on both targets when FMA
 * contraction is off: build with -ffp-contract=off), so the HIP kernels and the CPU oracle
 * produce bit-identical results.  Accuracy against glibc is pinned in tests/test_detmath.py
 * (<= 2 ulp over the ranges the environments use).
 *
 * Polynomials/reduction constants are the classic fdlibm (Sun, freely distributable) minimax
 * sets for __kernel_sin/__kernel_cos/atan.
 *
 * Usable from C99 (gcc), C++ and HIP device code.
 */
#ifndef DYNENV_MATH_H
#define DYNENV_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DM_FN __host__ __device__ __forceinline__
#ifndef DM_EXPERIMENT_CONTRACT_FAST /* (timing experiments only: what contraction everywhere would buy; never a product build) */
#pragma clang fp contract(off)
#endif
#else
#define DM_FN static inline
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC optimize("fp-contract=off")
#endif
#endif

#define DM_PI 3.141592653589793
#define DM_TWO_PI 6.283185307179586
#define DM_PI_2 1.5707963267948966

DM_FN double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); } /* one rounding; see "fused multiply-add" below */
DM_FN double dm_abs(double x) { return x < 0.0 ? -x : x; } /* -0.0 -> -0.0 stays harmless */
DM_FN double dm_min(double a, double b) { return a < b ? a : b; }
DM_FN double dm_max(double a, double b) { return a > b ? a : b; }
DM_FN double dm_clamp(double x, double lo, double hi) { return dm_min(dm_max(x, lo), hi); }

DM_FN double dm_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_sqrt(x); /* f64 sqrt is correctly rounded on gfx950 (checked by tests -m gpu) */
#else
  return __builtin_sqrt(x);
#endif
}

/* round-half-even to integer (v_rndne_f64 on gfx950, rint/roundsd on the host: both exact) */
DM_FN double dm_rint(double x) { return __builtin_rint(x); }

DM_FN double dm_kernel_sin(double r) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double v = z * r;
  double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return r + v * (S1 + z * p);
}

DM_FN double dm_kernel_cos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + z * p);
}

/* sin and cos of x, |x| < ~1e6 rad (angles in DynEnv stay within a few hundred rad). */
DM_FN void dm_sincos(double x, double* s, double* c) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double pio2_1t = 6.07710050650619224932e-11; /* pi/2 - pio2_1 */
  const double pio2_2 = 6.07710050630396597660e-11;  /* second 33 bits */
  const double pio2_2t = 2.02226624879595063154e-21; /* pi/2 - pio2_1 - pio2_2 */
  double fn = dm_rint(x * invpio2);
  double r;
  {
    /* two-stage Cody-Waite (fdlibm __ieee754_rem_pio2 medium case, always taking the 2nd stage) */
    double t = x - fn * pio2_1; /* exact product for |fn| < 2^20 */
    double w = fn * pio2_2;
    double r1 = t - w;
    double w2 = fn * pio2_2t - ((t - r1) - w);
    r = r1 - w2;
    (void)pio2_1t;
  }
  int64_t n = (int64_t)fn;
  double sr = dm_kernel_sin(r);
  double cr = dm_kernel_cos(r);
  switch ((int)(n & 3)) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
  }
}

/* The same with the polynomials and the second reduction stage as chains of dm_fma (9 fewer instructions of ~45): for the
 * OBSERVATION code only (getAgentVision's headings and noise angles: float32 outputs, hundreds of thousands of calls per step).
 * Every body's rotation goes through dm_sincos above, which stays unfused - fused, RoboCup's joint iterations reach their
 * bit-exact fixed point later and the step is 1.5-2 % slower (measured over four seeds). */
DM_FN void dm_sincos_f(double x, double* s, double* c) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double fn = dm_rint(x * invpio2);
  const double t = dm_fma(-fn, pio2_1, x); /* (the product is exact for |fn| < 2^20 anyway) */
  const double w = fn * pio2_2;
  const double r1 = t - w;
  const double r = r1 - dm_fma(fn, pio2_2t, -((t - r1) - w));
  const double z = r * r;
  const double sr = dm_fma(z * r, dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, S6, S5), S4), S3), S2), S1), r);
  const double hz = 0.5 * z, w1 = 1.0 - hz;
  const double cr = w1 + dm_fma(z, z * dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, dm_fma(z, C6, C5), C4), C3), C2), C1), (1.0 - w1) - hz);
  switch ((int)((int64_t)fn & 3)) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
  }
}

DM_FN double dm_sin(double x) { double s, c; dm_sincos(x, &s, &c); return s; }
DM_FN double dm_cos(double x) { double s, c; dm_sincos(x, &s, &c); return c; }

/* atan of ax >= 0 (fdlibm's __atan: five ranges, odd minimax polynomial).  Written without branches: the ranges only differ in
 * the quotient they reduce to and in the (hi, lo) they add back, so numerator, denominator, hi and lo are SELECTED and there is
 * one division - on a 64-lane wavefront whose lanes fall into all five ranges the branching form executes every arm, four
 * divisions of ~12 instructions each and fifteen exec-mask regions.  Operation for operation the same arithmetic per range (the
 * first range's ax / 1.0 is exact, and so are its hi = lo = 0 in the last line), so the results are bit-identical to the
 * branching form; tests/test_detmath.py holds a numpy restatement of the branching form and compares bit patterns.
 * The polynomial is a Horner chain of dm_fma (11 instead of 20 instructions, and the usual gain in accuracy: still within 2 ulp
 * of glibc, test_detmath.py); the restatement fuses the same operations with an exact rational fma.  atan2 only feeds the
 * observations (headings and bearing angles) - dm_sincos, which every body's rotation goes through, stays unfused: with its
 * multiply-adds fused RoboCup's joint iterations reach their bit-exact fixed point later and the step is 1.5-2 % slower. */
/* the arithmetic of one range: A..D, hi, lo as selected (or looked up: DM_ATAN_TAB below) for ax */
DM_FN double dm_atan_core(double ax, double nA, double nB, double dA, double dB, double hi, double lo) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  const double num = dm_fma(ax, nA, nB), den = dm_fma(ax, dA, dB); /* (ax = inf: 0 * inf, discarded by the last select below) */
  const double t = num / den;
  const double z = t * t;
  const double w = z * z;
  const double s1 = z * dm_fma(w, dm_fma(w, dm_fma(w, dm_fma(w, dm_fma(w, aT10, aT8), aT6), aT4), aT2), aT0);
  const double s2 = w * dm_fma(w, dm_fma(w, dm_fma(w, dm_fma(w, aT9, aT7), aT5), aT3), aT1);
  double r = hi - (dm_fma(t, s1 + s2, -lo) - t);   /* first range: 0 - ((t s - 0) - t) = t - t s, exactly */
  if (ax < 3.7252902984619140625e-09) r = ax;      /* |x| < 2^-28 */
  if (ax >= 1.0e300) r = 1.57079632679489655800e+00 + 6.12323399573676603587e-17; /* huge (incl. inf): atanhi3 + atanlo3 */
  return r;
}
/* the five ranges' {nA, nB, dA, dB, hi, lo}, for callers that look the row up instead of selecting it (the device's observation
 * code keeps a copy in LDS: half of dm_atan2's vector instructions were the select chains); row = number of range bounds <= ax:
 *                       [0, .4375)  [.4375, .6875)  [.6875, 1.1875)  [1.1875, 2.4375)  [2.4375, inf)
 *   numerator           ax          2 ax - 1        ax - 1           ax - 1.5          -1
 *   denominator         1           2 + ax          ax + 1           1 + 1.5 ax        ax                      */
#define DM_ATAN_TAB                                                                                          \
  {1.0, 0.0, 0.0, 1.0, 0.0, 0.0,                                                                             \
   2.0, -1.0, 1.0, 2.0, 4.63647609000806093515e-01, 2.26987774529616870924e-17,                              \
   1.0, -1.0, 1.0, 1.0, 7.85398163397448278999e-01, 3.06161699786838301793e-17,                              \
   1.0, -1.5, 1.5, 1.0, 9.82793723247329054082e-01, 1.39033110312309984516e-17,                              \
   0.0, -1.0, 1.0, 0.0, 1.57079632679489655800e+00, 6.12323399573676603587e-17}
DM_FN int dm_atan_row(double ax) { return !(ax < 0.4375) + !(ax < 0.6875) + !(ax < 1.1875) + !(ax < 2.4375); }
DM_FN double dm_atan_pos(double ax) {
  const double atanhi0 = 4.63647609000806093515e-01, atanhi1 = 7.85398163397448278999e-01,
               atanhi2 = 9.82793723247329054082e-01, atanhi3 = 1.57079632679489655800e+00;
  const double atanlo0 = 2.26987774529616870924e-17, atanlo1 = 3.06161699786838301793e-17,
               atanlo2 = 1.39033110312309984516e-17, atanlo3 = 6.12323399573676603587e-17;
  const int r0 = ax < 0.4375, r1 = ax < 0.6875, r2 = ax < 1.1875, r3 = ax < 2.4375;
  /* the quotient's two halves as ax * A + B with selected A, B (products by 0, 1, 2 are exact; 1.5 ax is the one rounded
   * product of the branching form), so that no arm is left for the compiler to branch around */
  const double nA = r0 ? 1.0 : (r1 ? 2.0     : (r2 ? 1.0     : (r3 ? 1.0     : 0.0)));
  const double nB = r0 ? 0.0 : (r1 ? -1.0    : (r2 ? -1.0    : (r3 ? -1.5    : -1.0)));
  const double dA = r0 ? 0.0 : (r1 ? 1.0     : (r2 ? 1.0     : (r3 ? 1.5     : 1.0)));
  const double dB = r0 ? 1.0 : (r1 ? 2.0     : (r2 ? 1.0     : (r3 ? 1.0     : 0.0)));
  const double hi = r0 ? 0.0 : (r1 ? atanhi0 : (r2 ? atanhi1 : (r3 ? atanhi2 : atanhi3)));
  const double lo = r0 ? 0.0 : (r1 ? atanlo0 : (r2 ? atanlo1 : (r3 ? atanlo2 : atanlo3)));
  return dm_atan_core(ax, nA, nB, dA, dB, hi, lo);
}
DM_FN double dm_atan(double x) {
  const double r = dm_atan_pos(__builtin_fabs(x));
  return x < 0.0 ? -r : r;
}

/* math.atan2 semantics incl. signed zeros (Road.py:315 spotDir.angle relies on atan2(-0.0,-90) == -pi) */
DM_FN int dm_signbit(double x) {
  union { double d; uint64_t u; } q;
  q.d = x;
  return (int)(q.u >> 63);
}

/* Branch-free as well: the quadrant logic is two selects, and so are the zero operands (0 / 0 would be a NaN, and for x = -0 the
 * general expression pi - (pi/2 - pi_lo) lands one ulp above pi/2).  A NaN operand propagates through the division.  One corner
 * differs from the branching form this replaces: a quotient that underflows to -0 (or x = +-inf with y < 0) now gives the -0 / -pi
 * of math.atan2, where the old form - whose |q| kept the sign of a zero - gave +0 / +pi. */
/* quadrant logic and zero operands of dm_atan2, given z = atan |y / x| */
DM_FN double dm_atan2_finish(double y, double x, double aq, double z) {
  const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16;
  if (x < 0.0 && aq < 1.0e-300) z = 0.0;
  {
    double r = dm_signbit(x) ? pi - (z - pi_lo) : z; /* (z - pi_lo) - pi == -(pi - (z - pi_lo)) exactly */
    if (x == 0.0) r = DM_PI_2;
    if (y == 0.0) r = dm_signbit(x) ? pi : 0.0;
    return dm_signbit(y) ? -r : r;
  }
}
DM_FN double dm_atan2(double y, double x) {
  const double aq = __builtin_fabs(y / x);
  return dm_atan2_finish(y, x, aq, dm_atan_pos(aq)); /* aq >= 1e300: atanhi3 + atanlo3 == pi/2 */
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 counter-based RNG (Salmon et al., SC'11).  Replaces the reference's serial
 * CPython `random` (MT19937) and NumPy legacy RandomState streams: a serial
 * stream cannot be reproduced by thousands of environments stepping in parallel, a counter
 * keyed by (seed, global env id | episode, purpose, entity, time) can, and is invariant to how
 * environments are sharded over GPUs.
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t v[4]; } dm_u32x4;

DM_FN dm_u32x4 dm_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  int r;
  for (r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  {
    dm_u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
  }
}

/* uniform integer in [lo, hi] (inclusive, like random.randint) from one 32-bit word */
DM_FN int32_t dm_randint(uint32_t u, int32_t lo, int32_t hi) {
  uint32_t n = (uint32_t)(hi - lo + 1);
  return lo + (int32_t)(((uint64_t)u * n) >> 32);
}
/* uniform double in [0,1) with 32-bit resolution (exact conversion) */
DM_FN double dm_unit(uint32_t u) { return (double)u * 2.3283064365386962890625e-10; }

/* ---- fused multiply-add, and the contact / joint solver's arithmetic built on it (round 4) -----------------------------------
 * fma(a, b, c) = a * b + c with ONE rounding is an IEEE-754 operation, correctly rounded on gfx950 (v_fma_f64) and on the host
 * (vfmadd with -mfma, libm's fma() otherwise): explicit calls give bit-identical results on both sides, which compiler
 * contraction (-ffp-contract=fast) would not.  The sequential-impulse solver is the dependent fp64 instruction stream a launch's
 * slowest environment spends its time in; with its multiply-adds fused it needs 23 instead of 38 instructions per contact for
 * the position-correction half of cpArbiterApplyImpulse.  The fused forms below ARE the arithmetic contract of `Space.step`'s
 * solver in this repository: oracle/cp_lite.c and the kernels call the same functions.  Against the unfused evaluation the
 * results differ in the last place of single operations - inside what the reference itself leaves open (pymunk ships Chipmunk
 * built with GCC in GNU C mode, whose default -ffp-contract=fast fuses exactly such expressions wherever the target has an
 * FMA, e.g. on aarch64) and far inside north_star's 1e-4. */
DM_FN double dms_dot(double ax, double ay, double bx, double by) { return dm_fma(ax, bx, ay * by); }      /* a.x b.x + a.y b.y */
DM_FN double dms_cross(double ax, double ay, double bx, double by) { return dm_fma(ax, by, -(ay * bx)); } /* a.x b.y - a.y b.x */
/* velocity of the point r of a body: v + perp(r) w, perp(r) = (-r.y, r.x) */
DM_FN double dms_point_vx(double vx, double ry, double w) { return dm_fma(-ry, w, vx); }
DM_FN double dms_point_vy(double vy, double rx, double w) { return dm_fma(rx, w, vy); }
/* cpvrotate(n, j) = (n.x j.x - n.y j.y, n.x j.y + n.y j.x) */
DM_FN double dms_rotate_x(double nx, double ny, double jx, double jy) { return dm_fma(nx, jx, -(ny * jy)); }
DM_FN double dms_rotate_y(double nx, double ny, double jx, double jy) { return dm_fma(nx, jy, ny * jx); }
/* a polygon vertex in world coordinates, cpTransformPoint: (c x - s y) + px, (s x + c y) + py, both multiply-adds fused */
DM_FN double dms_xform_x(double c, double s, double x, double y, double px) { return dm_fma(c, x, dm_fma(-s, y, px)); }
DM_FN double dms_xform_y(double c, double s, double x, double y, double py) { return dm_fma(s, x, dm_fma(c, y, py)); }
/* k_scalar_body: m_inv + i_inv (r x n)^2 */
DM_FN double dms_k_scalar(double m_inv, double i_inv, double rx, double ry, double nx, double ny) {
  const double rcn = dms_cross(rx, ry, nx, ny);
  return dm_fma(i_inv * rcn, rcn, m_inv);
}
/* accumulate-and-clamp of cpArbiterApplyImpulse: max(acc + c * nMass, 0) */
DM_FN double dms_acc_clamp0(double c, double nMass, double acc) {
  const double s = dm_fma(c, nMass, acc);
  return (s > 0.0) ? s : 0.0;
}

/* b ** n for small non-negative integer n by square-and-multiply (deterministic on host and device; the reference's
 * `0.9999 ** touchCntr` dice thresholds, RoboCupEnvironment.py:1065,1069,1117) */
DM_FN double dm_powi(double b, int n) {
  double r = 1.0;
  while (n > 0) {
    if (n & 1) r = r * b;
    b = b * b;
    n >>= 1;
  }
  return r;
}

/* RNG purposes (counter word c1) */
#define DM_RNG_RESET_AGENT 1u
#define DM_RNG_RESET_PERM 2u
#define DM_RNG_RESET_PED 3u
#define DM_RNG_RESET_OBST 4u
#define DM_RNG_RESET_COUNTS 5u
#define DM_RNG_PED_MOVE 6u
#define DM_RNG_ROBO_RESET 7u
#define DM_RNG_ROBO_STEP 8u
#define DM_RNG_OBS_NOISE 9u

DM_FN dm_u32x4 dm_env_rng(uint64_t seed, uint32_t genv, uint32_t episode, uint32_t purpose, uint32_t entity,
                           uint32_t t) {
  return dm_philox((uint32_t)seed, (uint32_t)(seed >> 32) ^ (genv * 0x9E3779B1u + 0x7F4A7C15u), episode, purpose,
                   entity, t);
}

#endif /* DYNENV_MATH_H */
