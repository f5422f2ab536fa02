// dev_common.h — shared device helpers for the gfx950 kernels (wave64, one environment per wavefront).
// No compatibility layers: this is CDNA4-only code (64-wide ballots, v_readlane, LDS tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dynenv.h"
#include "dynenv_math.h"

#define DE_WAVE 64
#define DE_DEV __device__ __forceinline__
// Out-of-line device function.  `static` (internal linkage) together with -fno-optimize-sibling-calls (no `tail` marker on its
// calls) lets LLVM's interprocedural register allocation drop the callee-saved-register convention for it (TargetFrameLowering's
// no-CSR optimisation: local linkage, norecurse, never a tail call): the function saves nothing on entry, and a caller keeps
// only what it really has live across the call.  Without either of the two, every call of a function that needs all 128 VGPRs
// stores and reloads the 48 callee-saved ones through scratch (12 KB per wave and call: 3/4 of the step kernels' HBM traffic).
#define DE_OOL static __device__ __noinline__
#define DE_HD __host__ __device__ __forceinline__

struct V2 {
  double x, y;
};
DE_HD V2 v2(double x, double y) { V2 r; r.x = x; r.y = y; return r; }
DE_HD V2 vadd(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
DE_HD V2 vsub(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
DE_HD V2 vneg(V2 a) { return v2(-a.x, -a.y); }
DE_HD V2 vmul(V2 a, double s) { return v2(a.x * s, a.y * s); }
DE_HD double vdot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
DE_HD double vcross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
DE_HD V2 vperp(V2 a) { return v2(-a.y, a.x); }
DE_HD V2 vrotate(V2 a, V2 b) { return v2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// fused forms (include/dynenv_math.h "fused multiply-add": the solver's and the polygon narrowphase's arithmetic)
DE_HD double vdot_f(V2 a, V2 b) { return dms_dot(a.x, a.y, b.x, b.y); }
DE_HD V2 vrotate_f(V2 n, V2 j) { return v2(dms_rotate_x(n.x, n.y, j.x, j.y), dms_rotate_y(n.x, n.y, j.x, j.y)); }
DE_HD double vlensq(V2 a) { return vdot(a, a); }
DE_HD V2 vlerp(V2 a, V2 b, double t) { return vadd(vmul(a, 1.0 - t), vmul(b, t)); }
DE_HD double vlen(V2 v) { return dm_sqrt(v.x * v.x + v.y * v.y); }  // Vec2d.length
DE_HD V2 vrot_angle(V2 v, double a) {                                // Vec2d.rotate(angle)
  double s, c;
  dm_sincos(a, &s, &c);
  return v2(v.x * c - v.y * s, v.x * s + v.y * c);
}
DE_HD double fmax_cp(double a, double b) { return (a > b) ? a : b; }
DE_HD double fmin_cp(double a, double b) { return (a < b) ? a : b; }
DE_HD double fclamp_cp(double f, double lo, double hi) { return fmin_cp(fmax_cp(f, lo), hi); }
DE_HD double fclamp01_cp(double f) { return fmax_cp(0.0, fmin_cp(f, 1.0)); }

// ---- wave64 cross-lane helpers -------------------------------------------------------------
DE_DEV uint64_t wave_ballot(bool p) { return __ballot(p); }
// (not threadIdx.x & 63: a function that reads the work-item id makes every caller keep v31, the ABI's packed id, alive for it)
DE_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// The lane id computed afresh (two instructions, no input): a value the compiler cannot tie to an earlier copy, so a kernel that
// calls it after every call of a 128-VGPR function keeps nothing lane-derived alive - and spilled - across that call.
// (workgroup = one wave: the id within the wave is threadIdx.x)
DE_DEV int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
DE_DEV uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// broadcast from a wave-uniform lane index (v_readlane_b32 pairs for doubles)
DE_DEV int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
DE_DEV double bcast_d(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  src = __builtin_amdgcn_readfirstlane(src);
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}
// value of `v` in lane `src` (any lane, per-lane index; source lanes must be active): ds_bpermute_b32, no LDS storage involved
DE_DEV int lane_read_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
DE_DEV double lane_read_d(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(src << 2, u.i[1]);
  return u.d;
}
DE_DEV int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
DE_DEV uint64_t uniform_u64(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
template <typename T>
DE_DEV T* uniform_ptr(T* p) { return (T*)uniform_u64((uint64_t)p); }
DE_DEV double uniform_d(double v) { return __longlong_as_double((long long)uniform_u64((uint64_t)__double_as_longlong(v))); }

// Out-of-line transcendental wrappers for device code.  sincos/atan2 are only needed when a car's angle changes or a
// collision callback fires; inlining their polynomial constants into the substep loop makes the compiler hoist dozens
// of 64-bit literals into registers for the whole loop (=> scratch spills on the hot path).  As small leaf functions
// they only touch caller-saved registers.
struct DevSC {
  double s, c;
};
DE_OOL DevSC dev_sincos(double x) {
  DevSC r;
  dm_sincos(x, &r.s, &r.c);
  return r;
}
DE_OOL DevSC dev_sincos_v(double x) {  // the observation code's variant (dm_sincos_f: fused polynomials)
  DevSC r;
  dm_sincos_f(x, &r.s, &r.c);
  return r;
}
DE_DEV DevSC dev_sincos_inl(double x) {  // for functions that must not contain a call (their live values would need callee-saved registers)
  DevSC r;
  dm_sincos(x, &r.s, &r.c);
  return r;
}
DE_OOL double dev_atan2(double y, double x) { return dm_atan2(y, x); }
// The observation code's variant: the range's six constants come from a 30-double table in LDS (byte address `tab`, 16-byte
// aligned; DEV_ATAN_TAB_INIT fills it) instead of six select chains - half of dm_atan2's vector instructions.  Same arithmetic
// (dm_atan_core), same constants: bit-identical to dm_atan2.
__device__ const double g_atanTab[30] = DM_ATAN_TAB;
#define DEV_ATAN_TAB_INIT(dst, lane) do { if ((lane) < 30) (dst)[lane] = g_atanTab[lane]; } while (0)
DE_DEV int dev_lds_addr(const void* p) { return (int)(size_t)(const __attribute__((address_space(3))) void*)p; }
DE_OOL double dev_atan2_t(double y, double x, int tab) {
  typedef double __attribute__((ext_vector_type(2))) d2;
  const double aq = __builtin_fabs(y / x);
  typedef const __attribute__((address_space(3))) d2* lds_d2p;
  const lds_d2p row = (lds_d2p)(size_t)(unsigned)(tab + 48 * dm_atan_row(aq));
  const d2 n = row[0], d = row[1], hl = row[2];
  return dm_atan2_finish(y, x, aq, dm_atan_core(aq, n.x, n.y, d.x, d.y, hl.x, hl.y));
}
DE_DEV double dev_cos(double x) { return dev_sincos(x).c; }

#define DE_DBL_MIN 2.2250738585072014e-308
// Chipmunk space defaults reached through pymunk.Space() (environment_base.py:126-128); values pinned against
// libm pow() by tests/test_oracle_physics.py::test_bias_constants_match_device_header
#define DE_COLLISION_SLOP 0.1
#define DE_CONTACT_BIAS_COEF 0.061259621561307376 /* 1 - pow(pow(1-0.1f,60), 0.01) */
#define DE_PIVOT_BIAS_COEF 0.022762779044189331   /* 1 - pow(0.1, 0.01)  (Robot.py:59 error_bias=0.1) */
#define DE_JOINT_BIAS_COEF DE_CONTACT_BIAS_COEF   /* default errorBias = pow(1-0.1f,60) */
#define DE_DT 0.01
