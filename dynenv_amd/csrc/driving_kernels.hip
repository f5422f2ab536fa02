// driving_kernels.hip — hand-written gfx950 kernels for the batched DynEnv Driving step.
//
// Replaces, for E environments at once, the body of DrivingEnvironment.step (reference DrivingEnvironment.py:248-322):
// 10 physics substeps {processAction, tick, pedestrian move | pymunk Space.step | bookkeeping} + Full observation
// + per-agent rewards, all fused in ONE launch.  One wavefront per environment; see driving_dev.h for the lane roles
// and DESIGN.md for the roofline discussion.  fp64 throughout (the reference computes in Python float / C double);
// FMA contraction is OFF so results are bit-identical to the CPU oracle.
//
// Register discipline: the LDS tile is the HOME of every per-body quantity; each phase of a substep loads the few
// values it needs, computes, and stores back.  Nothing but a handful of scalars stays live across phases, so the
// rare contact path (narrowphase + solver) cannot push the common path into scratch.
#include "driving_dev.h"

__constant__ DrvConst C;

#include "driving_host.h" /* DRV_WAVES_PER_SIMD, kernel declarations */

// ------------------------------------------------------------------------------------------------
// LDS tile of one environment (10 KiB budget => 16 environments per CU)
// ------------------------------------------------------------------------------------------------
struct DrvMailbox {  // narrowphase result of the pair that maps to slot s (written by the detecting lane)
  double p1x[DRV_NS][2], p1y[DRV_NS][2], p2x[DRV_NS][2], p2y[DRV_NS][2], nx[DRV_NS], ny[DRV_NS];
  int hash[DRV_NS][2], count[DRV_NS], flag[DRV_NS]; /* flag: bit0 touched, bit1 newly allocated */
};
struct DrvObsStage {  // f32 staging of the observation rows (only used after the last substep)
  float carRow[DRV_MAXA][8];
  float goal[DRV_MAXA][2];
  float shared[DRV_MAXO * 4 + DRV_MAXP * 2 + DRV_LANE_ROWS * 5];
};
// -DDRV_PROFILE: per-environment cycle counters for the step kernel and the stages of the contact path, dumped by
// dynenv_debug_counters() (tools/contact_profile.py).  Compiled out of the product build.
#ifdef DRV_PROFILE
#define DRV_PROF(...) __VA_ARGS__
#else
#define DRV_PROF(...)
#endif
#define DRV_CLIST 128  // capacity of the per-substep candidate list (485 pairs exist; more than 128 AABB overlaps sets EI_ERR)
struct __align__(16) DrvLds {
  // dynamic bodies (lane l = body l): home location of the state
  double px[DRV_NB], py[DRV_NB], vx[DRV_NB], vy[DRV_NB], ang[DRV_NB], w[DRV_NB], vbx[DRV_NB], vby[DRV_NB], wb[DRV_NB];
  double minv[DRV_NB], iinv[DRV_NB];
  double rc[16], rs[16], rotAng[16];  // cars: cos/sin of rotAng (recomputed only when the angle changes)
  // cars
  double dirx[16], diry[16], prevx[16], prevy[16], goalx[16], goaly[16];
  double cosRel0[16];  // dm_cos(road0.dirAngle - rotAng), cached with rc/rs.  (Road 1 has dirAngle 0: dm_cos(0 - a) is dm_cos(a) = rc bit for
                       // bit - dm_sincos negates fn and r exactly with its argument, the cosine kernel is even and the quadrant cases pair up)
  double dprev[16];                 // |prevPos - goal|
  double cmass[16], cpower[16], chx[16], chy[16];  // per-car constants (Car.py:9-12) copied out of constant memory
  double aabb[DRV_MAXA][4];
  int flags[DRV_NB], moving[DRV_NB];
  double ox[DRV_MAXO], oy[DRV_MAXO];
  // contact cache slots (lane s = slot s)
  int s_pair[DRV_NS], s_meta[DRV_NS], s_hash0[DRV_NS], s_hash1[DRV_NS];
  double s_jn0[DRV_NS], s_jt0[DRV_NS], s_jn1[DRV_NS], s_jt1[DRV_NS];
  double rewAcc[16], posAcc[16];  // the step's reward / positive-reward accumulators of the cars (:252-254): in LDS, not carried
                                  // through the substep loop in registers (two doubles less to save around every call)
  // Per-lane values the step carries from substep to substep live HERE, not in the kernel's registers: whatever the kernel holds in
  // VGPRs across the call of the solver (124 VGPRs) is spilled around it, 256 B of scratch per register and call.
  int lastCand[64];        // candidate mask of the previous substep (-1: unknown), persisted in S.lastcand
  int stepErr;             // error bits of this step (1: candidate list / contact cache overflow, 2: malformed action)
  unsigned char act[16];   // cars: acc | steer << 2 (each 0..2), consumed by the first substep
  int still[DRV_NB];  // bit0: body had exactly zero v, w, v_bias, w_bias when positions were integrated; bit1: frozen
  unsigned short clist[DRV_CLIST];  // contact path: dense list of candidate pair ids in canonical order
  // What the contact path derives from the STRUCTURE of the contact set alone - which pairs are candidates, which arbiters are active -
  // is kept from call to call within a launch: a pile that is being relaxed keeps its structure for hundreds of substeps.
  int clistN;                       // >= 0: clist holds the full list of the current candidate masks (that many pairs); -1: rebuild
  int sLvMeta;                      // maxLevel & 0xFF | period << 8 of the schedule below
  unsigned long long sActive;       // active-arbiter mask the cached level schedule was computed for (~0: none)
  unsigned char sLevel[DRV_NS];     // level of slot s in that schedule
  union {
    DrvMailbox mb;
    DrvObsStage ob;
  } u;
};

// One tile per workgroup (= one wavefront = one environment).  File scope so that the out-of-line contact path
// addresses it with ds_* instructions instead of flat pointers.
static_assert(sizeof(DrvLds) <= 10240, "16 one-wave workgroups per CU - one residency round of 4096 environments on 256 CUs - need <= 160 KB / 16 of LDS each");
__shared__ DrvLds g_L;


// flag-word accessors (layout in driving_dev.h)
#define CF_FIN(f) (((f) >> 4) & 1)
#define CF_CRASHED(f) (((f) >> 5) & 1)
#define CF_FRIC(f) (((f) >> 6) & 1)
#define CF_LP(f) (((f) >> 8) & 7)
#define CF_SET_LP(f, lp) (((f) & ~(7 << 8)) | ((lp) << 8))
#define CF_CRASH_BITS ((1 << 4) | (1 << 5) | (1 << 6)) /* Car.crash(): finished, crashed, friction_car_crashed */
#define PF_ROAD(f) ((f)&1)
#define PF_SIDE(f) (((f) >> 1) & 1)
#define PF_DEAD(f) (((f) >> 2) & 1)
#define PF_CROSSING(f) (((f) >> 3) & 1)
#define PF_BEGIN(f) (((f) >> 4) & 1)
#define PF_SPEED(f) (((f) >> 8) & 15)

// ------------------------------------------------------------------------------------------------
// game logic (mirrors oracle/driving.c, which cites the reference lines)
// ------------------------------------------------------------------------------------------------
// The two roads are fixed by the reference (DrivingEnvironment.py:110-115).  Their derived constants are spelled as
// literals here (instead of reading DrvConst) so that the substep loop carries no loop-invariant SGPRs for them; the
// host checks at dynenv_create() that these literals equal the values it computes with the Road.__init__ arithmetic.
#define DRV_COS_PI_2 6.123233995736766e-17 /* dm_cos(pi/2) == math.cos(math.pi/2) */
template <int R>
struct RoadK;
template <>
struct RoadK<0> {  // Road(2, 35, [(875,0),(875,1000)])
  static constexpr double p0x = 875.0, p0y = 0.0, dirx = 0.0, diry = 1.0, lat = 2.0 * 35.0 + 5.0, length = 1000.0;
  static constexpr double dirAngle = 1.5707963267948966, cosDir0 = DRV_COS_PI_2, nx = -1.0, ny = DRV_COS_PI_2;
};
template <>
struct RoadK<1> {  // Road(1, 35, [(0,500),(1750,500)])
  static constexpr double p0x = 0.0, p0y = 500.0, dirx = 1.0, diry = 0.0, lat = 1.0 * 35.0 + 5.0, length = 1750.0;
  static constexpr double dirAngle = 0.0, cosDir0 = 1.0, nx = DRV_COS_PI_2, ny = 1.0;
};

// Car.py:9-12 (masses, lengths = half extent x, widths = half extent y, powers) and cpMomentForPoly of the car's box, as
// literals; Pedestrian.py:11-14.  dynenv_create compares them with the table build_consts computes (car_literals_ok).
struct CarK {
  static constexpr double carMass0 = 1200.0, carMass1 = 1800.0, carMass2 = 3500.0, carMass3 = 5000.0;
  static constexpr double carHx0 = 10.0, carHx1 = 15.0, carHx2 = 20.0, carHx3 = 25.0;
  static constexpr double carHy0 = 5.0, carHy1 = 6.0, carHy2 = 7.0, carHy3 = 8.0;
  static constexpr double carPower0 = 3.0, carPower1 = 4.0, carPower2 = 3.0, carPower3 = 4.0;
  static constexpr double carInertia0 = 50000.0, carInertia1 = 156600.0, carInertia2 = 0x1.ff8e555555555p+18, carInertia3 = 0x1.185ad55555555p+20;
  static constexpr double pedMass = 90.0, pedInertia = 90.0 * (0.5 * (0.0 * 0.0 + 5.0 * 5.0) + 0.0);
};

// Road.isPointOnRoad (Road.py:74-97) with cos(road.dirAngle - angle) supplied by the caller (cached)
template <int R>
DE_DEV int road_pos(V2 point, double cosRel) {
  V2 pt = vsub(point, v2(RoadK<R>::p0x, RoadK<R>::p0y));
  double dist = vcross(v2(RoadK<R>::dirx, RoadK<R>::diry), pt);
  if (dm_abs(dist) >= RoadK<R>::lat) return LP_OffRoad;
  int pos = LP_OverRoad;
  double dirDist = vdot(v2(RoadK<R>::dirx, RoadK<R>::diry), pt);
  if (dirDist >= -10.0 && dirDist <= RoadK<R>::length + 10.0) {
    double relAngle = cosRel * dist;
    pos = relAngle < 0.0 ? LP_InRightLane : LP_InOpposingLane;
  }
  return pos;
}

DE_DEV bool drv_is_off_road(V2 point) {  // DrivingEnvironment.py:509-520 (angle argument is always 0)
  int position = LP_OffRoad;
  int rp = road_pos<0>(point, RoadK<0>::cosDir0);
  if (rp < position) position = rp;
  rp = road_pos<1>(point, RoadK<1>::cosDir0);
  if (rp < position) position = rp;
  return position >= LP_OverRoad;
}
DE_DEV bool drv_is_out(V2 p) { return p.x <= 0.0 || p.y <= 0.0 || p.x >= DRV_W || p.y >= DRV_H; }

// cutils.py:102-140 apply_friction (Body.update_velocity with g=0, damping=1, f=t=0 folded to `+ 0.0`)
DE_DEV void apply_friction(double& vx, double& vy, double& w, double m, double friction, double rotFriction,
                           double spin) {
  vx = vx * 1.0 + (0.0 + 0.0) * DE_DT;
  vy = vy * 1.0 + (0.0 + 0.0) * DE_DT;
  w = w * 1.0 + 0.0;
  double factor = friction * m;
  double rotFactor = rotFriction * m;
  double x = vx, y = vy;
  double length = 1.0 / (dm_abs(x) + dm_abs(y) + 1e-5);
  double theta = w;
  double a0 = x * factor * length;
  double a1 = y * factor * length;
  a0 += a1 * spin * theta;
  a1 -= a0 * spin * theta;
  if (dm_abs(x) < factor) x = 0.0; else x -= a0;
  if (dm_abs(y) < factor) y = 0.0; else y -= a1;
  if (dm_abs(theta) < rotFactor) theta = 0.0; else theta -= (theta > 0.0 ? rotFactor : -rotFactor);
  vx = x; vy = y; w = theta;
}

// velocity_func of body `lane` on the LDS tile (friction_car / friction_car_crashed / friction_pedestrian_dead /
// default cpBodyUpdateVelocity): parameters are selected per lane so that ONE instance of the arithmetic serves all
DE_DEV void velocity_update(DrvLds& L, int lane, bool isCar, bool isPed) {
  if (!(isCar || isPed)) return;
  double vx = L.vx[lane], vy = L.vy[lane], w = L.w[lane];
  const int f = L.flags[lane];
  double m, fr, rfr;
  bool dflt = false;
  if (isCar) { m = L.cmass[lane]; fr = CF_FRIC(f) ? 5e-4 : 5e-5; rfr = CF_FRIC(f) ? 2e-5 : 1e-5; }
  else { m = 90.0; fr = 5e-2; rfr = 2e-4; dflt = !PF_DEAD(f); }
  if (dflt) { vx = vx * 1.0 + (0.0 + 0.0) * DE_DT; vy = vy * 1.0 + (0.0 + 0.0) * DE_DT; w = w * 1.0 + 0.0; }
  else apply_friction(vx, vy, w, m, fr, rfr, 0.0);
  L.vx[lane] = vx; L.vy[lane] = vy; L.w[lane] = w;
}
// The same as a leaf function, for the substeps without a solve (fast path, replay): inlined into the kernel's substep loop, its
// friction coefficients - 64-bit literals - are hoisted out of the loop into registers that are then saved around every call.
DE_OOL void velocity_update_ool(int nCarPed) {
  const int lane = lane_id(), A = uniform_i(nCarPed) & 0xFF, nPed = uniform_i(nCarPed) >> 8;
  velocity_update(g_L, lane, lane < A, lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed);
}

// refresh the cached rotation of car `lane` (cpBodySetAngle -> cpvforangle) and the two road-relative cosines
// INL: the inline form for drv_light_substep, which must stay a LEAF function - a function that calls saves its return
// address through a VGPR that it stores to and reloads from scratch on every call (256 B per wave and substep, and the
// reload sits right in front of the return).
struct CarRot { double c, s, cosRel0; };
template <bool INL = false>
DE_DEV CarRot car_refresh_rot(DrvLds& L, int lane, double ang) {
  const DevSC sc = INL ? dev_sincos_inl(ang) : dev_sincos(ang);
  CarRot r;
  r.c = sc.c; r.s = sc.s;
  r.cosRel0 = INL ? dev_sincos_inl(RoadK<0>::dirAngle - ang).c : dev_cos(RoadK<0>::dirAngle - ang);
  L.rc[lane] = r.c; L.rs[lane] = r.s; L.rotAng[lane] = ang; L.cosRel0[lane] = r.cosRel0;
  static_assert(RoadK<1>::dirAngle == 0.0, "cos(road1.dirAngle - angle) is read from rc");
  return r;
}

// ------------------------------------------------------------------------------------------------
// narrowphase (mirrors oracle/cp_lite.c)
// ------------------------------------------------------------------------------------------------
struct BoxW {
  V2 v[4], n[4];
};
DE_DEV void box_world(BoxW& b, V2 p, double rc, double rs, double hx, double hy) {
  const double lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
  const double nx[4] = {-1.0, 0.0, 1.0, 0.0}, ny[4] = {0.0, -1.0, 0.0, 1.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b.v[i] = v2(dms_xform_x(rc, rs, lx[i], ly[i], p.x), dms_xform_y(rc, rs, lx[i], ly[i], p.y));  // as cpShapeCacheBB in oracle/cp_lite.c
    b.n[i] = v2(rc * nx[i] - rs * ny[i], rs * nx[i] + rc * ny[i]);
  }
}

struct Contacts {
  int count;
  V2 n, p1[2], p2[2];
  int hash[2];
};

struct EdgeW {
  V2 ap, bp, n;
  int ah, bh;
};

// select element `i` of a 4-vector without dynamic register indexing (no scratch)
DE_DEV V2 sel4(const V2* a, int i) {
  V2 r = a[0];
  r = (i == 1) ? a[1] : r;
  r = (i == 2) ? a[2] : r;
  r = (i == 3) ? a[3] : r;
  return r;
}

// A box by its parameters.  The narrowphase derives vertices and normals from these on demand (same arithmetic as
// box_world) instead of holding two expanded boxes (64 doubles) in registers, which spilled to scratch.
struct BoxP {
  V2 p;
  double c, s, hx, hy;
};
DE_DEV V2 boxp_vertex(const BoxP& b, int i) {
  const double lx = (i == 1 || i == 2) ? b.hx : -b.hx;
  const double ly = (i >= 2) ? b.hy : -b.hy;
  return v2(dms_xform_x(b.c, b.s, lx, ly, b.p.x), dms_xform_y(b.c, b.s, lx, ly, b.p.y));
}
DE_DEV V2 boxp_normal(const BoxP& b, int i) {
  const double nx = (i == 0) ? -1.0 : ((i == 2) ? 1.0 : 0.0);
  const double ny = (i == 1) ? -1.0 : ((i == 3) ? 1.0 : 0.0);
  return v2(b.c * nx - b.s * ny, b.s * nx + b.c * ny);
}

// ---- quad-cooperative narrowphase: 4 adjacent lanes (a DPP quad) work on one pair, lane q owning axis / vertex /
// edge q.  Every reduction replays the reference's sequential loop (index ascending, strict comparison) on the four
// per-lane values, so results are bit-identical to the scalar code in oracle/cp_lite.c.  All four lanes of a quad
// follow the same control flow (every branch condition is quad-uniform), so DPP reads never hit an inactive lane.
template <int S> DE_DEV double quad_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, S * 0x55, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, S * 0x55, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// for (i = 0..3) if (v_i > best) { best = v_i; idx = i; }   starting from best = -inf, idx = 0
DE_DEV void quad_argmax_first(double v, double& best, int& idx) {
  best = -INFINITY; idx = 0;
  double m;
  m = quad_bcast<0>(v); if (m > best) { best = m; idx = 0; }
  m = quad_bcast<1>(v); if (m > best) { best = m; idx = 1; }
  m = quad_bcast<2>(v); if (m > best) { best = m; idx = 2; }
  m = quad_bcast<3>(v); if (m > best) { best = m; idx = 3; }
}

DE_DEV int poly_support_index(const BoxP& p, V2 n, int q) {
  double mx;
  int index;
  quad_argmax_first(vdot_f(boxp_vertex(p, q), n), mx, index);
  return index;
}

DE_DEV EdgeW support_edge_poly(const BoxP& p, int slot, V2 n, int q) {
  int i1 = poly_support_index(p, n, q);
  int i0 = (i1 + 3) & 3;
  int i2 = (i1 + 1) & 3;
  int h = slot * 4;
  V2 n1 = boxp_normal(p, i1), n2 = boxp_normal(p, i2);
  EdgeW e;
  V2 vm = boxp_vertex(p, i1);
  if (vdot(n, n1) > vdot(n, n2)) {
    e.ap = boxp_vertex(p, i0); e.ah = h + i0; e.bp = vm; e.bh = h + i1; e.n = n1;
  } else {
    e.ap = vm; e.ah = h + i1; e.bp = boxp_vertex(p, i2); e.bh = h + i2; e.n = n2;
  }
  return e;
}

DE_DEV int hash_pair(int a, int b) { return 1 + ((a << 8) | b); }

DE_DEV void contact_points(const EdgeW& e1, const EdgeW& e2, V2 n, Contacts& out) {  // radius 0 edges
  double d_e1_a = vcross(e1.ap, n), d_e1_b = vcross(e1.bp, n);
  double d_e2_a = vcross(e2.ap, n), d_e2_b = vcross(e2.bp, n);
  double e1_denom = 1.0 / (d_e1_b - d_e1_a + DE_DBL_MIN);
  double e2_denom = 1.0 / (d_e2_b - d_e2_a + DE_DBL_MIN);
  out.n = n;
  out.count = 0;
  {
    V2 p1 = vadd(vmul(n, 0.0), vlerp(e1.ap, e1.bp, fclamp01_cp((d_e2_b - d_e1_a) * e1_denom)));
    V2 p2 = vadd(vmul(n, -0.0), vlerp(e2.ap, e2.bp, fclamp01_cp((d_e1_a - d_e2_a) * e2_denom)));
    double dist = vdot(vsub(p2, p1), n);
    if (dist <= 0.0) { out.p1[0] = p1; out.p2[0] = p2; out.hash[0] = hash_pair(e1.ah, e2.bh); out.count = 1; }
  }
  {
    V2 p1 = vadd(vmul(n, 0.0), vlerp(e1.ap, e1.bp, fclamp01_cp((d_e2_a - d_e1_a) * e1_denom)));
    V2 p2 = vadd(vmul(n, -0.0), vlerp(e2.ap, e2.bp, fclamp01_cp((d_e1_b - d_e2_a) * e2_denom)));
    double dist = vdot(vsub(p2, p1), n);
    if (dist <= 0.0) {
      int h = hash_pair(e1.bh, e2.ah);
      if (out.count == 0) { out.p1[0] = p1; out.p2[0] = p2; out.hash[0] = h; }
      else { out.p1[1] = p1; out.p2[1] = p2; out.hash[1] = h; }
      out.count += 1;
    }
  }
}

DE_DEV double sat_max_sep(const BoxP& a, const BoxP& b, int q, int& best) {
  const V2 n = boxp_normal(a, q);
  const double d0 = vdot_f(n, boxp_vertex(a, q));
  double minv = INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double d = vdot_f(n, boxp_vertex(b, j)) - d0;
    if (d < minv) minv = d;
  }
  double maxsep;
  quad_argmax_first(minv, maxsep, best);
  return maxsep;
}

DE_DEV void poly_to_poly(const BoxP& p1, int slot1, const BoxP& p2, int slot2, int q, Contacts& out) {
  out.count = 0;
  int ia, ib;
  double sa = sat_max_sep(p1, p2, q, ia);
  if (sa > 0.0) return;
  double sb = sat_max_sep(p2, p1, q, ib);
  if (sb > 0.0) return;
  V2 n;
  if (sa >= sb) n = boxp_normal(p1, ia); else n = vneg(boxp_normal(p2, ib));
  const EdgeW e1 = support_edge_poly(p1, slot1, n, q);
  const EdgeW e2 = support_edge_poly(p2, slot2, vneg(n), q);
  contact_points(e1, e2, n, out);
}

DE_DEV void circle_to_poly(V2 c, double r, const BoxP& poly, int q, Contacts& out) {
  out.count = 0;
  double maxsep;
  int best;
  const V2 vq = boxp_vertex(poly, q);
  quad_argmax_first(vdot(boxp_normal(poly, q), vsub(c, vq)), maxsep, best);
  if (maxsep > r) return;
  if (maxsep <= 0.0) {
    V2 fn = boxp_normal(poly, best);
    V2 n = vneg(fn);
    V2 pb = vsub(c, vmul(fn, maxsep));
    out.n = n; out.p1[0] = vadd(c, vmul(n, r)); out.p2[0] = pb; out.hash[0] = 0; out.count = 1;
  } else {
    // lane q: closest point on edge (v[q-1], v[q]); then for (i = 0..3) if (dsq_i < bestd) take it
    const V2 a = boxp_vertex(poly, (q + 3) & 3), b = vq;
    const V2 d = vsub(b, a);
    const double t = fclamp01_cp(vdot(d, vsub(c, a)) / vlensq(d));
    const V2 pt = vadd(a, vmul(d, t));
    const double dsq = vlensq(vsub(pt, c));
    double bestd = INFINITY;
    V2 bestp = c;
    double m;
    m = quad_bcast<0>(dsq); if (m < bestd) { bestd = m; bestp = v2(quad_bcast<0>(pt.x), quad_bcast<0>(pt.y)); }
    m = quad_bcast<1>(dsq); if (m < bestd) { bestd = m; bestp = v2(quad_bcast<1>(pt.x), quad_bcast<1>(pt.y)); }
    m = quad_bcast<2>(dsq); if (m < bestd) { bestd = m; bestp = v2(quad_bcast<2>(pt.x), quad_bcast<2>(pt.y)); }
    m = quad_bcast<3>(dsq); if (m < bestd) { bestd = m; bestp = v2(quad_bcast<3>(pt.x), quad_bcast<3>(pt.y)); }
    if (bestd <= r * r) {
      double dist = dm_sqrt(bestd);
      V2 delta = vsub(bestp, c);
      V2 n = (dist != 0.0) ? vmul(delta, 1.0 / dist) : vneg(boxp_normal(poly, best));
      out.n = n; out.p1[0] = vadd(c, vmul(n, r)); out.p2[0] = bestp; out.hash[0] = 0; out.count = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// body table access (index < 30: dynamic body in LDS; >= 30: static box, zero velocity / inverse mass)
// ------------------------------------------------------------------------------------------------
struct BodyV {
  V2 p, v, vb;
  double w, wb, minv, iinv;
};
DE_DEV V2 static_pos(const DrvLds& L, int idx) {
  if (idx >= DRV_SLOT_BLD) {
    int k = idx - DRV_SLOT_BLD; /* DrivingEnvironment.py:101-106 */
    return v2((k & 2) ? 1385.0 : 365.0, (k & 1) ? 800.0 : 200.0);
  }
  return v2(L.ox[idx - DRV_SLOT_OBST], L.oy[idx - DRV_SLOT_OBST]);
}
// branch-free (round 6): every field from body slot idx, or from the all-zero slot 30 for a static body (body_rd below); the position of a
// static body from the obstacle table or the buildings' constants by selection, both loads issued
DE_DEV void body_load(const DrvLds& L, int idx, BodyV& b) {
  const bool dyn = idx < DRV_SLOT_OBST;
  const int bi = dyn ? idx : DRV_SLOT_OBST /* = the zero slot */, oi = (idx >= DRV_SLOT_OBST && idx < DRV_SLOT_BLD) ? idx - DRV_SLOT_OBST : 0;
  const double dpx = L.px[bi], dpy = L.py[bi], sox = L.ox[oi], soy = L.oy[oi];
  const int k = idx - DRV_SLOT_BLD;
  const double bx = (k & 2) ? 1385.0 : 365.0, by = (k & 1) ? 800.0 : 200.0;  // DrivingEnvironment.py:101-106
  b.p = dyn ? v2(dpx, dpy) : (idx >= DRV_SLOT_BLD ? v2(bx, by) : v2(sox, soy));
  b.v = v2(L.vx[bi], L.vy[bi]); b.w = L.w[bi];
  b.vb = v2(L.vbx[bi], L.vby[bi]); b.wb = L.wb[bi]; b.minv = L.minv[bi]; b.iinv = L.iinv[bi];
}
// the solver's per-iteration view of a body: p, minv and iinv do not change while a substep is being solved
DE_DEV void body_load_vel(const DrvLds& L, int idx, BodyV& b) {
  if (idx < DRV_SLOT_OBST) {
    b.v = v2(L.vx[idx], L.vy[idx]); b.w = L.w[idx]; b.vb = v2(L.vbx[idx], L.vby[idx]); b.wb = L.wb[idx];
  }
}
DE_DEV void body_store_vel(DrvLds& L, int idx, const BodyV& b) {  // (a static body's all-zero velocities go to slot 31, which nobody reads)
  const int wi = idx < DRV_SLOT_OBST ? idx : DRV_NB - 1;
  L.vx[wi] = b.v.x; L.vy[wi] = b.v.y; L.w[wi] = b.w; L.vbx[wi] = b.vb.x; L.vby[wi] = b.vb.y; L.wb[wi] = b.wb;
}
// The arbiter solver's arithmetic with its multiply-adds FUSED: the dms_* functions of include/dynenv_math.h, the same ones
// oracle/cp_lite.c calls (k_scalar_body_f, relative_velocity_f, apply_impulse_f ...).
DE_DEV double k_scalar_body(const BodyV& b, V2 r, V2 n) { return dms_k_scalar(b.minv, b.iinv, r.x, r.y, n.x, n.y); }
DE_DEV V2 relative_velocity(const BodyV& a, const BodyV& b, V2 r1, V2 r2) {
  V2 v1 = v2(dms_point_vx(a.v.x, r1.y, a.w), dms_point_vy(a.v.y, r1.x, a.w));
  V2 v2s = v2(dms_point_vx(b.v.x, r2.y, b.w), dms_point_vy(b.v.y, r2.x, b.w));
  return vsub(v2s, v1);
}
DE_DEV void apply_impulse(BodyV& b, V2 j, V2 r) {
  b.v = v2(dm_fma(j.x, b.minv, b.v.x), dm_fma(j.y, b.minv, b.v.y));
  b.w = dm_fma(b.iinv, dms_cross(r.x, r.y, j.x, j.y), b.w);
}
DE_DEV void apply_bias_impulse(BodyV& b, V2 j, V2 r) {
  b.vb = v2(dm_fma(j.x, b.minv, b.vb.x), dm_fma(j.y, b.minv, b.vb.y));
  b.wb = dm_fma(b.iinv, dms_cross(r.x, r.y, j.x, j.y), b.wb);
}
// v_bias + perp(r) w_bias of the two bodies at their contact points, as the difference along n (vbn of cpArbiterApplyImpulse)
DE_DEV double bias_rel_n(const BodyV& a, const BodyV& b, V2 r1, V2 r2, V2 n) {
  const V2 vb1 = v2(dms_point_vx(a.vb.x, r1.y, a.wb), dms_point_vy(a.vb.y, r1.x, a.wb));
  const V2 vb2 = v2(dms_point_vx(b.vb.x, r2.y, b.wb), dms_point_vy(b.vb.y, r2.x, b.wb));
  return vdot_f(vsub(vb2, vb1), n);
}

// ------------------------------------------------------------------------------------------------
// observation writer: Full obs of DrivingEnvironment.getFullState/get_full_obs (:686-747, :121-124)
// dense padded row per agent: [self 9 | other cars (A-1)x7 | obstacles 20x4 | pedestrians 20x2 | lanes 8x5]
// ------------------------------------------------------------------------------------------------
#define STD_NORM_X (0.5 / (DRV_W + 100.0))
#define STD_NORM_Y (0.5 / (DRV_H + 100.0))
#define STD_NORM_W (1.0 / 15.0)
#define STD_NORM_H (1.0 / 25.0)
DE_DEV double normalize_obs(double pt, double nf, double mean) { return ((pt * nf) - mean) * 2.0 * 1.0; }  // cutils.py:318-323

DE_DEV float obs_self_or_car(const DrvObsStage& O, int A, int a, int ff) {
  if (ff < 6) return O.carRow[a][ff];
  if (ff < 8) return O.goal[a][ff - 6];
  if (ff == 8) return O.carRow[a][6];
  int c = (ff - 9) / 7, kk = (ff - 9) - c * 7;
  c += (c >= a);
  return O.carRow[c][kk];
}

// all state is read from the LDS tile; the staging area aliases the (idle) mailbox.  Out of line (a leaf with a register allocation
// of its own): inlined into the step kernel's epilogue it reloaded spilled addresses between its stores, and a reload waits for
// every store issued before it (vmcnt is in order) - 2.8 % of the launch.
DE_OOL void write_full_obs_ool(int lane, int A, int nPed, int nObst, int obs_dim, float* __restrict__ out) {
  DrvLds& L = g_L;
  DrvObsStage& O = L.u.ob;
  __syncthreads();
  if (lane < A) {
    int f = L.flags[lane];
    O.carRow[lane][0] = (float)normalize_obs(L.px[lane], STD_NORM_X, 0.0);
    O.carRow[lane][1] = (float)normalize_obs(L.py[lane], STD_NORM_Y, 0.0);
    O.carRow[lane][2] = (float)L.rc[lane];  // rc/rs are cos/sin of the current angle (refreshed on every change)
    O.carRow[lane][3] = (float)L.rs[lane];
    O.carRow[lane][4] = (float)normalize_obs(L.chy[lane], STD_NORM_W, 0.5);  // c.width
    O.carRow[lane][5] = (float)normalize_obs(L.chx[lane], STD_NORM_H, 0.5);  // c.height
    O.carRow[lane][6] = (float)CF_FIN(f);
    O.goal[lane][0] = (float)normalize_obs(L.goalx[lane], STD_NORM_X, 0.0);
    O.goal[lane][1] = (float)normalize_obs(L.goaly[lane], STD_NORM_Y, 0.0);
  }
  if (lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + DRV_MAXP) {
    int k = lane - DRV_SLOT_PED;
    bool on = k < nPed;
    O.shared[DRV_MAXO * 4 + 2 * k + 0] = on ? (float)normalize_obs(L.px[lane], STD_NORM_X, 0.0) : 0.0f;
    O.shared[DRV_MAXO * 4 + 2 * k + 1] = on ? (float)normalize_obs(L.py[lane], STD_NORM_Y, 0.0) : 0.0f;
  }
  if (lane < DRV_MAXO) {
    bool on = lane < nObst;
    O.shared[4 * lane + 0] = on ? (float)normalize_obs(L.ox[lane], STD_NORM_X, 0.0) : 0.0f;
    O.shared[4 * lane + 1] = on ? (float)normalize_obs(L.oy[lane], STD_NORM_Y, 0.0) : 0.0f;
    O.shared[4 * lane + 2] = on ? (float)normalize_obs(10.0, STD_NORM_W, 0.5) : 0.0f;
    O.shared[4 * lane + 3] = on ? (float)normalize_obs(10.0, STD_NORM_H, 0.5) : 0.0f;
  }
  if (lane < DRV_LANE_ROWS * 5) O.shared[DRV_MAXO * 4 + DRV_MAXP * 2 + lane] = C.laneRows[lane];
  __syncthreads();
  const int carsEnd = 9 + (A - 1) * 7;
  if (((carsEnd | obs_dim) & 3) == 0) {
    // vector path (A in {2,6,10}): one float4 per lane; the 160-float tail is identical for every agent of the env
    // the per-agent part (own row, then the other cars in order): lane = float index, which fixes the source column and - but for
    // the skip of the agent itself - the source car; one LDS read and one 4-byte store per agent (a wave's stores are 256 contiguous
    // bytes), instead of assembling float4s through four index computations (a division by 7 each) per agent
    for (int f = lane; f < carsEnd; f += DE_WAVE) {
      const bool own = f < 9;
      const int p = own ? 0 : (f - 9) / 7, col = own ? (f < 6 ? f : (f == 8 ? 6 : f - 6)) : (f - 9) - p * 7;
      const bool fromGoal = f == 6 || f == 7;
      for (int a = 0; a < A; ++a) {
        const int k = own ? a : p + (p >= a ? 1 : 0);
        out[(size_t)a * obs_dim + f] = fromGoal ? O.goal[a][col] : O.carRow[k][col];
      }
    }
    const int nvec = (obs_dim - carsEnd) >> 2;
    for (int q = lane; q < nvec; q += DE_WAVE) {
      const float4 v = *reinterpret_cast<const float4*>(&O.shared[q << 2]);
      for (int a = 0; a < A; ++a) *reinterpret_cast<float4*>(out + (size_t)a * obs_dim + carsEnd + (q << 2)) = v;
    }
  } else {
    for (int idx = lane; idx < A * obs_dim; idx += DE_WAVE) {
      int a = idx / obs_dim, ff = idx - a * obs_dim;
      out[idx] = ff < carsEnd ? obs_self_or_car(O, A, a, ff) : O.shared[ff - carsEnd];
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// HBM <-> LDS (field-major rows: each field of one env is one coalesced 256-byte segment)
// ------------------------------------------------------------------------------------------------
// Every global load of an environment is ISSUED before the first one is waited for (addresses are clamped into the arrays, the
// values of lanes without that object are replaced by the defaults afterwards): two memory round trips per launch - the
// environment's scalars, then everything else - instead of one per `if` of the straightforward form (13, ~1.2 k cycles each).
// The four car types' constants are wave-uniform (scalar loads) and selected by the type bits instead of indexed per lane.
DE_DEV void load_env(const DrvState& S, DrvLds& L, int e, int lane, int A, int nPed, int nObst, uint64_t occ) {
  const size_t E = (size_t)S.E;
  const int bl = lane & (DRV_NB - 1), cl = lane & 15, ol = lane < DRV_MAXO ? lane : DRV_MAXO - 1, sl = lane < DRV_NS ? lane : DRV_NS - 1;
  const double* b = S.body + (size_t)e * DRV_NB + bl;
  const double g_px = b[BF_PX * E * DRV_NB], g_py = b[BF_PY * E * DRV_NB], g_vx = b[BF_VX * E * DRV_NB], g_vy = b[BF_VY * E * DRV_NB];
  const double g_ang = b[BF_ANG * E * DRV_NB], g_w = b[BF_W * E * DRV_NB];
  const double g_vbx = b[BF_VBX * E * DRV_NB], g_vby = b[BF_VBY * E * DRV_NB], g_wb = b[BF_WB * E * DRV_NB];
  const int g_f = S.flags[(size_t)e * DRV_NB + bl], g_aux = S.aux[(size_t)e * DRV_NB + bl];
  const double* c = S.carx + (size_t)e * 16 + cl;
  const double g_prevx = c[CF_PREVX * E * 16], g_prevy = c[CF_PREVY * E * 16], g_gx = c[CF_GOALX * E * 16], g_gy = c[CF_GOALY * E * 16];
  const double g_dirx = c[CF_DIRX * E * 16], g_diry = c[CF_DIRY * E * 16];
  const double g_ox = S.obst[(size_t)e * DRV_MAXO + ol], g_oy = S.obst[E * DRV_MAXO + (size_t)e * DRV_MAXO + ol];
  int g_pair = 0xFFFF, g_meta = 0, g_h0 = 0, g_h1 = 0;
  double g_jn0 = 0.0, g_jt0 = 0.0, g_jn1 = 0.0, g_jt1 = 0.0;
  if (occ != 0ull) {  // (wave-uniform: most environments have an empty contact cache and skip these eight loads)
    const size_t o = (size_t)e * DRV_NS + sl;
    g_pair = S.s_pair[o]; g_meta = S.s_meta[o];
    g_h0 = (int)S.s_hash[o]; g_h1 = (int)S.s_hash[E * DRV_NS + o];
    g_jn0 = S.s_imp[o]; g_jt0 = S.s_imp[E * DRV_NS + o]; g_jn1 = S.s_imp[2 * E * DRV_NS + o]; g_jt1 = S.s_imp[3 * E * DRV_NS + o];
  }
  if (lane < DRV_NB) {
    const bool used = lane < A || (lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed);
    L.px[lane] = used ? g_px : 0.0; L.py[lane] = used ? g_py : 0.0;
    L.vx[lane] = used ? g_vx : 0.0; L.vy[lane] = used ? g_vy : 0.0;
    const double ang = used ? g_ang : 0.0;
    L.ang[lane] = ang; L.w[lane] = used ? g_w : 0.0;
    L.vbx[lane] = used ? g_vbx : 0.0; L.vby[lane] = used ? g_vby : 0.0;
    L.wb[lane] = used ? g_wb : 0.0;
    const int f = used ? g_f : 0;
    L.flags[lane] = f;
    L.moving[lane] = used ? g_aux : 0;
    const int ty = f & 3;
    // the four car types' constants (Car.py:9-12 and cpMomentForPoly of the box) as literals selected by the type bits: no
    // memory access that depends on the flag word just loaded (dynenv_create checks them against the computed table, CarK)
#define DRV_BY_TYPE(ARR) (ty == 3 ? CarK::ARR##3 : (ty == 2 ? CarK::ARR##2 : (ty == 1 ? CarK::ARR##1 : CarK::ARR##0)))
    double minv = 0.0, iinv = 0.0;
    if (lane < A) { minv = 1.0 / DRV_BY_TYPE(carMass); iinv = 1.0 / DRV_BY_TYPE(carInertia); }
    else if (used) { minv = 1.0 / CarK::pedMass; iinv = 1.0 / CarK::pedInertia; }
    L.minv[lane] = minv; L.iinv[lane] = iinv;
    if (lane < 16) { L.rc[lane] = 1.0; L.rs[lane] = 0.0; L.rotAng[lane] = 0.0; }
    if (lane < A) {
      L.dirx[lane] = g_dirx; L.diry[lane] = g_diry;
      L.prevx[lane] = g_prevx; L.prevy[lane] = g_prevy; L.goalx[lane] = g_gx; L.goaly[lane] = g_gy;
      L.dprev[lane] = vlen(vsub(v2(g_prevx, g_prevy), v2(g_gx, g_gy)));
      L.cmass[lane] = DRV_BY_TYPE(carMass); L.cpower[lane] = DRV_BY_TYPE(carPower);
      L.chx[lane] = DRV_BY_TYPE(carHx); L.chy[lane] = DRV_BY_TYPE(carHy);
      car_refresh_rot(L, lane, ang);
    }
#undef DRV_BY_TYPE
  }
  if (lane < DRV_MAXO) {
    L.ox[lane] = lane < nObst ? g_ox : 0.0;
    L.oy[lane] = lane < nObst ? g_oy : 0.0;
  }
  if (lane < DRV_NS) {
    const bool on = (occ >> lane) & 1ull;
    L.s_pair[lane] = on ? g_pair : 0xFFFF;
    L.s_meta[lane] = on ? g_meta : 0;
    L.s_hash0[lane] = on ? g_h0 : 0; L.s_hash1[lane] = on ? g_h1 : 0;
    L.s_jn0[lane] = on ? g_jn0 : 0.0; L.s_jt0[lane] = on ? g_jt0 : 0.0;
    L.s_jn1[lane] = on ? g_jn1 : 0.0; L.s_jt1[lane] = on ? g_jt1 : 0.0;
  }
}

DE_DEV void store_env(const DrvState& S, const DrvLds& L, int e, int lane, int A, int nPed, uint64_t occ) {
  const size_t E = (size_t)S.E;
  if (lane < A || (lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed)) {
    double* b = S.body + (size_t)e * DRV_NB + lane;
    b[BF_PX * E * DRV_NB] = L.px[lane]; b[BF_PY * E * DRV_NB] = L.py[lane]; b[BF_VX * E * DRV_NB] = L.vx[lane];
    b[BF_VY * E * DRV_NB] = L.vy[lane]; b[BF_ANG * E * DRV_NB] = L.ang[lane]; b[BF_W * E * DRV_NB] = L.w[lane];
    b[BF_VBX * E * DRV_NB] = L.vbx[lane]; b[BF_VBY * E * DRV_NB] = L.vby[lane]; b[BF_WB * E * DRV_NB] = L.wb[lane];
    S.flags[(size_t)e * DRV_NB + lane] = L.flags[lane];
    if (lane < A) {
      double* c = S.carx + (size_t)e * 16 + lane;
      c[CF_DIRX * E * 16] = L.dirx[lane]; c[CF_DIRY * E * 16] = L.diry[lane];
      c[CF_PREVX * E * 16] = L.prevx[lane]; c[CF_PREVY * E * 16] = L.prevy[lane];
    } else {
      S.aux[(size_t)e * DRV_NB + lane] = L.moving[lane];
    }
  }
  if (lane < DRV_NS && ((occ >> lane) & 1ull)) {
    size_t o = (size_t)e * DRV_NS + lane;
    S.s_pair[o] = L.s_pair[lane]; S.s_meta[o] = L.s_meta[lane];
    S.s_hash[o] = (uint32_t)L.s_hash0[lane]; S.s_hash[E * DRV_NS + o] = (uint32_t)L.s_hash1[lane];
    S.s_imp[o] = L.s_jn0[lane]; S.s_imp[E * DRV_NS + o] = L.s_jt0[lane];
    S.s_imp[2 * E * DRV_NS + o] = L.s_jn1[lane]; S.s_imp[3 * E * DRV_NS + o] = L.s_jt1[lane];
  }
}

// ------------------------------------------------------------------------------------------------
// collision `begin` callbacks, executed wave-uniformly in canonical pair order on the LDS tile
// (DrivingEnvironment.py:587-683: carCrash / pedHit / carHit).  Returns false => arbiter ignored until separation.
// `rew` is the per-step reward accumulator held by the car's own lane.
// ------------------------------------------------------------------------------------------------
DE_DEV bool cb_begin(DrvLds& L, int i, int j, int lane) {
  // i = car, j = partner slot (car / pedestrian lane, or static >= 30); all arguments are wave-uniform
  const double v1x = L.vx[i], v1y = L.vy[i];
  const int f1 = L.flags[i];
  if (j < DRV_SLOT_PED) {  // carCrash :591-637
    const double v2x = L.vx[j], v2y = L.vy[j];
    const int f2 = L.flags[j];
    const int crashed1 = CF_CRASHED(f1), crashed2 = CF_CRASHED(f2), pos1 = CF_LP(f1), pos2 = CF_LP(f2);
    const double v1len = vlen(v2(v1x, v1y)), v2len = vlen(v2(v2x, v2y));
    const double v1l = v1len / 5.0, v2l = v2len / 5.0;
    double r1 = L.rewAcc[i], r2 = L.rewAcc[j];
    if (!crashed1) r1 -= v1l;
    if (!crashed2) r2 -= v2l;
    if (pos1 != LP_InRightLane && !crashed1) r1 -= v1l;
    if (pos2 != LP_InRightLane && !crashed2) r2 -= v2l;
    if (pos1 == LP_InRightLane && pos2 == LP_InRightLane) {
      V2 dp = v2(L.px[i] - L.px[j], L.py[i] - L.py[j]);
      double adp = dev_atan2(dp.y, dp.x);
      if (v1len > 1.0 && dev_cos(adp - dev_atan2(v1y, v1x)) < -0.4 && !crashed1) r1 -= v1l;
      if (v2len > 1.0 && dev_cos(adp - dev_atan2(v2y, v2x)) > 0.4 && !crashed2) r2 -= v2l;
    }
    if (lane == i) { L.rewAcc[i] = r1; L.flags[i] = f1 | CF_CRASH_BITS; }
    if (lane == j) { L.rewAcc[j] = r2; L.flags[j] = f2 | CF_CRASH_BITS; }
    return true;
  } else if (j < DRV_SLOT_OBST) {  // pedHit :640-667
    const double v1l = vlen(v2(v1x, v1y));
    if (v1l > 1.0) {
      V2 dp = v2(L.px[i] - L.px[j], L.py[i] - L.py[j]);
      if (lane == j) {  // Pedestrian.die (Pedestrian.py:40-47)
        L.moving[j] = 0; L.vx[j] = 0.0; L.vy[j] = 0.0; L.flags[j] = L.flags[j] | (1 << 2);
      }
      if (dev_cos(dev_atan2(dp.y, dp.x) - dev_atan2(v1y, v1x)) < -0.4 && !CF_FIN(f1)) {
        if (lane == i) { L.flags[i] = f1 | CF_CRASH_BITS; L.rewAcc[i] = L.rewAcc[i] - v1l / 5.0; }
      }
      return true;
    }
    return false;
  } else {  // carHit :670-683
    if (lane == i) {
      if (!CF_FIN(f1)) L.rewAcc[i] = L.rewAcc[i] - vlen(v2(v1x, v1y)) / 5.0;
      L.flags[i] = f1 | CF_CRASH_BITS;
    }
    return true;
  }
}

// ------------------------------------------------------------------------------------------------
// The contact path of one substep (drv_contact_path below): narrowphase -> contact cache -> callbacks -> prestep, velocity
// update and solve (drv_prestep_solve, out of line; the sweeps of a general multi-level solve in drv_solve_general_split).
// Operates on the LDS tile; returns the few scalars it changes.  First the solver's building blocks.
// ------------------------------------------------------------------------------------------------
// cpArbiterApplyCachedImpulse for the (up to two) contacts of one arbiter
DE_DEV void arb_warm_start(BodyV& a, BodyV& b, V2 n, const V2* r1, const V2* r2, const double* jn, const double* jt,
                           int count) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c == 0 || count > 1) {  // (an arbiter that is warm-started has one contact or two)
      V2 j = vrotate_f(n, v2(jn[c], jt[c]));
      j = vmul(j, 1.0);  // dt_coef = dt/prev_dt = 1
      apply_impulse(a, vneg(j), r1[c]);
      apply_impulse(b, j, r2[c]);
    }
  }
}
// one cpArbiterApplyImpulse pass over the contacts of one arbiter
DE_DEV void arb_apply_impulse(BodyV& a, BodyV& b, V2 n, const V2* r1, const V2* r2, const double* nMass,
                              const double* tMass, const double* bias, const double* bounce, double* jBias, double* jn,
                              double* jt, int count, double arb_u) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c == 0 || count > 1) {
      double vbn = bias_rel_n(a, b, r1[c], r2[c], n);
      // Bias-only contact (a resting contact that is still being pushed out of penetration): both bodies have all-zero
      // (+0) velocities, no accumulated impulse and no bounce.  Then vr = +-0, jn and jt come out as +0 again and the
      // velocity impulse is +-0, which leaves +0 velocities at +0: the velocity half of the pass is an exact no-op.
      const long long zbits = __double_as_longlong(a.v.x) | __double_as_longlong(a.v.y) | __double_as_longlong(a.w) |
                              __double_as_longlong(b.v.x) | __double_as_longlong(b.v.y) | __double_as_longlong(b.w) |
                              __double_as_longlong(jn[c]) | __double_as_longlong(jt[c]);
      if (zbits == 0ll && bounce[c] == 0.0) {
        double jbnOld = jBias[c];
        jBias[c] = dms_acc_clamp0(bias[c] - vbn, nMass[c], jbnOld);
        V2 jb = vmul(n, jBias[c] - jbnOld);
        apply_bias_impulse(a, vneg(jb), r1[c]);
        apply_bias_impulse(b, jb, r2[c]);
        continue;
      }
      V2 vr = relative_velocity(a, b, r1[c], r2[c]);
      double vrn = vdot_f(vr, n);
      double jbnOld = jBias[c];
      jBias[c] = dms_acc_clamp0(bias[c] - vbn, nMass[c], jbnOld);
      double jnOld = jn[c];
      jn[c] = dms_acc_clamp0(-(bounce[c] + vrn), nMass[c], jnOld);
      // Friction: no Driving shape sets one (Chipmunk default u = 0), so arb.u = 0 * 0 = +0 and jtMax = u * jn = +0.
      // cpfclamp(x, -0, +0) = cpfmin(cpfmax(x, -0), +0) is +0 for EVERY x (also inf / NaN): jt stays +0 and its increment
      // is +0 - +0 = +0.  The tangent speed, tMass and the clamp are therefore not evaluated; the +0 increment still goes
      // through cpvrotate exactly as in the reference.
      V2 jb = vmul(n, jBias[c] - jbnOld);
      apply_bias_impulse(a, vneg(jb), r1[c]);
      apply_bias_impulse(b, jb, r2[c]);
      V2 jj = vrotate_f(n, v2(jn[c] - jnOld, 0.0));
      apply_impulse(a, vneg(jj), r1[c]);
      apply_impulse(b, jj, r2[c]);
    }
  }
}

// The bias half of cpArbiterApplyImpulse alone (exactly the bias-only branch above), for solves in which EVERY active
// arbiter is bias-only.  Nothing in such a solve writes v, w, jn or jt, so the condition - evaluated once, after the warm
// start - holds for all 10 iterations and the loop needs neither the per-pass test nor the velocities themselves.
DE_DEV bool arb_apply_bias_only(BodyV& a, BodyV& b, V2 n, const V2* r1, const V2* r2, const double* nMass, const double* bias,
                                double* jBias, int count) {
  bool changed = false;
#define DRV_BIAS_PASS(c)                                              \
  {                                                                   \
    double vbn = bias_rel_n(a, b, r1[c], r2[c], n);                   \
    double jbnOld = jBias[c];                                         \
    jBias[c] = dms_acc_clamp0(bias[c] - vbn, nMass[c], jbnOld);       \
    changed |= jBias[c] != jbnOld;                                    \
    V2 jb = vmul(n, jBias[c] - jbnOld);                               \
    apply_bias_impulse(a, vneg(jb), r1[c]);                           \
    apply_bias_impulse(b, jb, r2[c]);                                 \
  }
  DRV_BIAS_PASS(0)   // (an active arbiter has one contact or two)
  if (count > 1) DRV_BIAS_PASS(1)
#undef DRV_BIAS_PASS
  return changed;
}
DE_DEV bool arb_is_bias_only(const BodyV& a, const BodyV& b, const double* jn, const double* jt, const double* bounce, int count) {
  long long z = __double_as_longlong(a.v.x) | __double_as_longlong(a.v.y) | __double_as_longlong(a.w) |
                __double_as_longlong(b.v.x) | __double_as_longlong(b.v.y) | __double_as_longlong(b.w);
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c == 0 || count > 1) { z |= __double_as_longlong(jn[c]) | __double_as_longlong(jt[c]); ok = ok && bounce[c] == 0.0; }
  return ok && z == 0ll;
}
DE_DEV void body_load_bias(const DrvLds& L, int idx, BodyV& b) {
  if (idx < DRV_SLOT_OBST) { b.vb = v2(L.vbx[idx], L.vby[idx]); b.wb = L.wb[idx]; }
}
DE_DEV void body_store_bias(DrvLds& L, int idx, const BodyV& b) {
  if (idx < DRV_SLOT_OBST) { L.vbx[idx] = b.vb.x; L.vby[idx] = b.vb.y; L.wb[idx] = b.wb; }
}
// Branch-free access for the sweeps (round 6, as in the RoboCup solver): a static partner reads the zeros of body slot 30 - unused,
// cleared by load_env, never written - and writes to slot 31, which nobody reads; its velocities are +0 after every finite impulse
// (x * 0 + 0), so reloading zeros is what keeping them in registers was.  Every `if (dynamic)` around a load or a store was a
// saveexec + a taken branch into an out-of-line block: ~40 cycles each for a lone wave, four per pass.
#define DRV_ZERO_SLOT 30
#define DRV_SINK_SLOT 31
static_assert(DRV_SLOT_OBST == DRV_ZERO_SLOT && DRV_NB == 32, "body slots 30 and 31 are free");
DE_DEV int body_rd(int idx) { return idx < DRV_SLOT_OBST ? idx : DRV_ZERO_SLOT; }
DE_DEV int body_wr(int idx) { return idx < DRV_SLOT_OBST ? idx : DRV_SINK_SLOT; }

// ------------------------------------------------------------------------------------------------
// The sweeps of a GENERAL multi-level solve (some arbiter with moving bodies or accumulated impulses, arbiters sharing bodies:
// a car pushing into a pile - the slowest environments of a launch), as a function of its own.  An arbiter's two impulse chains -
// velocity (v, w, jn) and position correction (v_bias, w_bias, jBias) - read and write disjoint body fields and run the same
// instruction sequence:  j += clamp((c -+ rel.n) * nMass),  impulse along n,  two body updates.  A lone wave pays per
// instruction, not per lane, so the bias chain of slot s runs on lane s + 32 beside the velocity chain on lane s: 63 instead of
// 95 fp64 instructions per contact and pass.  Same operations on the same operands in the same order as arb_apply_impulse,
// chain by chain (the two only meet in memory, in different fields): bit-identical.  The mirror lanes fetch their arbiter's
// scalars from the slot lane (ds_bpermute) and rebuild r1, r2, n from the mailbox exactly as the prestep did.
// (Inside drv_prestep_solve this form needs more registers than the function has: 16 spills on every call, measured.  Here only
// the solves that take this path pay for what their caller keeps across the call.)
// ------------------------------------------------------------------------------------------------
struct DrvSplitRet { double jn0, jn1, jb0, jb1, jt0, jt1; int pk, pair, bits; };
DE_OOL DrvSplitRet drv_solve_general_split(int lane, int myLevel_, int bodyAB, int lvl_, double nMass0, double nMass1, double bias0, double bias1,
                                           double bounce0, double bounce1, double jn0, double jn1, double jt0, double jt1, int pk, int pair,
                                           int bits3) {
  DrvLds& L = g_L;
  DrvMailbox& M = L.u.mb;
  int code = ((pk >> 30) & 1) | (((pk >> 8) & 0xFF) << 1) | (myLevel_ << 8);  // active | contact count << 1 | level << 8
  const int maxLevel = (int)(signed char)(uniform_i(lvl_) & 0xFF), period = uniform_i(lvl_) >> 8;
  const bool biasLane = lane >= 32;
  const int sl = lane & 31;  // the slot this lane works for
  code = lane_read_i(code, sl); bodyAB = lane_read_i(bodyAB, sl);
  nMass0 = lane_read_d(nMass0, sl); nMass1 = lane_read_d(nMass1, sl);
  bias0 = lane_read_d(bias0, sl); bias1 = lane_read_d(bias1, sl);
  const bool active = (code & 1) != 0 && sl < DRV_NS;
  const int count = (code >> 1) & 3, myLevel = code >> 8;
  const int bodyA = bodyAB & 0xFF, bodyB = bodyAB >> 8;
  const double nM[2] = {nMass0, nMass1};
  const double cq[2] = {biasLane ? bias0 : bounce0, biasLane ? bias1 : bounce1};
  double acc[2] = {biasLane ? 0.0 : jn0, biasLane ? 0.0 : jn1};  // jBias starts every solve at zero, jn at the warm-started value
  V2 n = v2(0.0, 0.0), r1[2], r2[2];
  r1[0] = r1[1] = r2[0] = r2[1] = v2(0.0, 0.0);
  BodyV a, b;
  a.p = b.p = a.v = b.v = a.vb = b.vb = v2(0.0, 0.0); a.w = b.w = a.wb = b.wb = a.minv = b.minv = a.iinv = b.iinv = 0.0;
  if (active) {
    body_load(L, bodyA, a);
    body_load(L, bodyB, b);
    r1[0] = vsub(v2(M.p1x[sl][0], M.p1y[sl][0]), a.p);
    r2[0] = vsub(v2(M.p2x[sl][0], M.p2y[sl][0]), b.p);
    if (count > 1) {
      r1[1] = vsub(v2(M.p1x[sl][1], M.p1y[sl][1]), a.p);
      r2[1] = vsub(v2(M.p2x[sl][1], M.p2y[sl][1]), b.p);
    }
    n = v2(M.nx[sl], M.ny[sl]);
  }
  // this lane's chain works on (v, w) - the velocity fields on a slot lane, the bias fields on its mirror lane
  double* const fX = biasLane ? L.vbx : L.vx;
  double* const fY = biasLane ? L.vby : L.vy;
  double* const fW = biasLane ? L.wb : L.w;
  const bool aDyn = bodyA < DRV_SLOT_OBST, bDyn = bodyB < DRV_SLOT_OBST;
  a.v = b.v = v2(0.0, 0.0); a.w = b.w = 0.0;  // statics: all-zero and never stored
  const int nSteps = maxLevel + 1 + period * 9;  // pipelined sweeps, see drv_prestep_solve
  int due = myLevel, passes = 0;
  const int ra = body_rd(bodyA), rb = body_rd(bodyB), wa = body_wr(bodyA), wb_ = body_wr(bodyB);
#define DRV_SPLIT_PASS(q)                                                                   \
  {                                                                                         \
    const V2 vr = relative_velocity(a, b, r1[q], r2[q]);                                    \
    const double vn = vdot_f(vr, n);                                                        \
    const double t0 = cq[q] + (biasLane ? -vn : vn);                                        \
    const double old = acc[q];                                                              \
    acc[q] = dms_acc_clamp0(biasLane ? t0 : -t0, nM[q], old);                               \
    const double dj = acc[q] - old;                                                         \
    const V2 jr = vrotate_f(n, v2(dj, 0.0));                                                \
    const V2 jl = vmul(n, dj);                                                              \
    const V2 jj = biasLane ? jl : jr;                                                       \
    apply_impulse(a, vneg(jj), r1[q]);                                                      \
    apply_impulse(b, jj, r2[q]);                                                            \
  }
  for (int t = 0; t < nSteps; ++t) {
    if (active && t == due && passes < 10) {
      a.v = v2(fX[ra], fY[ra]); a.w = fW[ra];
      b.v = v2(fX[rb], fY[rb]); b.w = fW[rb];
      DRV_SPLIT_PASS(0)
      if (count > 1) DRV_SPLIT_PASS(1)
      fX[wa] = a.v.x; fY[wa] = a.v.y; fW[wa] = a.w;
      fX[wb_] = b.v.x; fY[wb_] = b.v.y; fW[wb_] = b.w;
      due += period; ++passes;
    }
    __syncthreads();
  }
#undef DRV_SPLIT_PASS
  DrvSplitRet r;
  r.jt0 = jt0; r.jt1 = jt1; r.pk = pk; r.pair = pair; r.bits = bits3;  // the caller's own values, handed back
  r.jn0 = acc[0]; r.jn1 = acc[1];
  r.jb0 = lane_read_d(acc[0], (lane + 32) & 63); r.jb1 = lane_read_d(acc[1], (lane + 32) & 63);  // a slot lane reads its mirror lane's jBias
  return r;
}

DRV_PROF(__device__ unsigned long long g_dbgw[4096 * 12];)
DRV_PROF(__device__ unsigned long long g_dbgp[4096 * 8];)
DRV_PROF(__device__ unsigned long long g_dbgr[16];)
DRV_PROF(__device__ unsigned long long g_dbgl[4096 * 8];)  // stages of drv_light_substep, summed over the step's substeps
DRV_PROF(__device__ unsigned long long g_dbgs[4096 * 8];)  // stages of the slot update (between narrowphase and prestep), summed over the step's calls
DRV_PROF(DE_DEV int prof_any(int v) { const uint64_t m = wave_ballot(v != 0); return m ? bcast_i(v, __builtin_ctzll(m)) : 0; })
// Which half of a substep is out of line: the COMMON part (game logic, position update, broadphase: drv_light_substep, a leaf
// with nothing to save) is a function; the contact path is inlined into the kernel, which as the outermost frame never saves a
// register (round 1 had it the other way round and paid the save / restore of 47 callee-saved VGPRs per call: 3/4 of that
// kernel's HBM traffic).  Its solver is a function again (drv_prestep_solve), entered with nothing of the contact cache live.
#ifndef DRV_CONTACT_INLINE
#define DRV_CONTACT_INLINE __forceinline__
#endif

// pk: a_state | a_count << 8 | a_age << 16 | touched << 24 | freeMe << 25 | hashSame << 26 | prevInert << 27 | skipped << 28 | slotOcc << 29 | active << 30
DE_OOL int drv_prestep_solve(int lane, int nCarPed, int pk, int a_pair, int bodyA, int bodyB, int myLevel,
                                             int maxLevel_, int anyActive_, double jn0, double jn1, double jt0, double jt1) {
  DrvLds& L = g_L;
  DrvMailbox& M = L.u.mb;
  int a_state = pk & 0xFF;
  int a_count = (pk >> 8) & 0xFF, a_age = (pk >> 16) & 0xFF;
  bool touched = (pk >> 24) & 1, freeMe = (pk >> 25) & 1, hashSame = (pk >> 26) & 1, prevInert = (pk >> 27) & 1;
  bool skipped = (pk >> 28) & 1, slotOcc = (pk >> 29) & 1, active = (pk >> 30) & 1;
  const bool isCar = lane < (uniform_i(nCarPed) & 0xFF), isPed = lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + (uniform_i(nCarPed) >> 8);
  const int maxLevel = (int)(signed char)(uniform_i(maxLevel_) & 0xFF), period = uniform_i(maxLevel_) >> 8;
  const uint64_t activeMask = uniform_i(anyActive_) ? 1ull : 0ull;
DRV_PROF(const unsigned long long P0 = __builtin_amdgcn_s_memtime();)
  double jn[2] = {jn0, jn1}, jt[2] = {jt0, jt1};
  V2 n = v2(0.0, 0.0), r1[2], r2[2];
  r1[0] = r1[1] = r2[0] = r2[1] = v2(0.0, 0.0);
  // ---- prestep (cpArbiterPreStep) on velocities BEFORE the friction update -------------------------------
  double nMass[2] = {0.0, 0.0}, tMass[2] = {0.0, 0.0}, bias[2] = {0.0, 0.0}, bounce[2] = {0.0, 0.0}, jBias[2] = {0.0, 0.0};
  const double arb_e = 0.05 * 0.05, arb_u = 0.0 * 0.0;
  bool restIn = false;  // both bodies exactly at rest when the arbiter was prestepped
  if (active) {
    BodyV a, b;
    body_load(L, bodyA, a);
    body_load(L, bodyB, b);
    restIn = a.v.x == 0.0 && a.v.y == 0.0 && a.w == 0.0 && b.v.x == 0.0 && b.v.y == 0.0 && b.w == 0.0;
    r1[0] = vsub(v2(M.p1x[lane][0], M.p1y[lane][0]), a.p);
    r2[0] = vsub(v2(M.p2x[lane][0], M.p2y[lane][0]), b.p);
    if (a_count > 1) {
      r1[1] = vsub(v2(M.p1x[lane][1], M.p1y[lane][1]), a.p);
      r2[1] = vsub(v2(M.p2x[lane][1], M.p2y[lane][1]), b.p);
    }
    n = v2(M.nx[lane], M.ny[lane]);
    V2 body_delta = vsub(b.p, a.p);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c == 0 || a_count > 1) {  // (an active arbiter has one contact or two)
        nMass[c] = 1.0 / (k_scalar_body(a, r1[c], n) + k_scalar_body(b, r2[c], n));
        double dist = vdot_f(vadd(vsub(r2[c], r1[c]), body_delta), n);  // (tMass is not needed: see arb_apply_impulse)
        bias[c] = -DE_CONTACT_BIAS_COEF * fmin_cp(0.0, dist + DE_COLLISION_SLOP) / DE_DT;
        jBias[c] = 0.0;
        bounce[c] = vdot_f(relative_velocity(a, b, r1[c], r2[c]), n) * arb_e;
      }
    }
  }
  __syncthreads();
DRV_PROF(const unsigned long long P1 = __builtin_amdgcn_s_memtime();)

  // ---- velocity update (velocity_func: friction_* or default) -------------------------------------------
  velocity_update(L, lane, isCar, isPed);
DRV_PROF(int profMode = 0; const unsigned long long P2 = __builtin_amdgcn_s_memtime();)
  bool tookSplit = false;  // (wave-uniform) the sweeps ran in drv_solve_general_split: counted in EI_N_SPLIT
  if (activeMask && maxLevel == 0) {
    // No two active arbiters share a dynamic body: each lane keeps its two bodies in registers through the warm start
    // and all 10 iterations, with one LDS load and one store (same arithmetic, no LDS round trip per iteration).
    __syncthreads();
    if (active) {
      BodyV a, b;
      body_load(L, bodyA, a);
      body_load(L, bodyB, b);
      if (a_state != ARB_FIRST) arb_warm_start(a, b, n, r1, r2, jn, jt, a_count);
      if (wave_ballot(!arb_is_bias_only(a, b, jn, jt, bounce, a_count)) == 0ull) {  // all resting contacts: bias half only
DRV_PROF(profMode = 1;)
#pragma unroll 1
        // (fixed-point exit: a sweep in which no accumulated bias impulse moved added +-0 to bias velocities that start at +0 and
        //  can never be -0 - it changed nothing, and neither will any later sweep)
        for (int iter = 0; iter < 10; ++iter)
          if (wave_ballot(arb_apply_bias_only(a, b, n, r1, r2, nMass, bias, jBias, a_count)) == 0ull) break;
      } else {
DRV_PROF(profMode = 2;)
#pragma unroll 1
        for (int iter = 0; iter < 10; ++iter) arb_apply_impulse(a, b, n, r1, r2, nMass, tMass, bias, bounce, jBias, jn, jt, a_count, arb_u);
      }
      body_store_vel(L, bodyA, a);
      body_store_vel(L, bodyB, b);
    }
    __syncthreads();
  } else if (activeMask) {
    __syncthreads();
    // ---- warm start (cpArbiterApplyCachedImpulse; skipped on first contact), level by level ------------
    for (int lv = 0; lv <= maxLevel; ++lv) {
      if (active && myLevel == lv && a_state != ARB_FIRST) {
        BodyV a, b;
        body_load(L, bodyA, a);
        body_load(L, bodyB, b);
        arb_warm_start(a, b, n, r1, r2, jn, jt, a_count);
        body_store_vel(L, bodyA, a);
        body_store_vel(L, bodyB, b);
      }
      __syncthreads();
    }
    // ---- 10 sequential-impulse iterations (cpArbiterApplyImpulse) ---------------------------------------
    BodyV a, b;
    bool biasOnly = true;
    if (active) {  // statics stay all-zero; p, minv, iinv are invariant
      body_load(L, bodyA, a); body_load(L, bodyB, b);
      biasOnly = arb_is_bias_only(a, b, jn, jt, bounce, a_count);
    }
    // PIPELINED SWEEPS.  Chipmunk runs 10 sweeps over the arbiters in order; sweep k + 1 of an arbiter only depends on sweep k of
    // the arbiters it shares a body with.  Pass (arbiter a, sweep k) runs at time step  level(a) + period * k  with
    // period = 1 + the largest level difference between two arbiters sharing a dynamic body: then for two such arbiters, a before
    // b in canonical order,  (a, k) < (b, k) < (a, k + 1)  holds in time exactly as in the sequential sweep (level(a) <
    // level(b) < level(a) + period), passes of one time step never share a body, and everything else commutes - the result is
    // bit for bit the sequential one.  A chain of L resting cars (period 2) needs L + 18 time steps instead of 10 L.
    const int nSteps = maxLevel + 1 + period * 9;
    int due = myLevel, passes = 0;
DRV_PROF(profMode = wave_ballot(!biasOnly) == 0ull ? 3 : 4;)
    if (wave_ballot(!biasOnly) == 0ull) {
      // every active arbiter is a resting contact being pushed out of penetration (the pile-ups that make up the launch's
      // tail): only bias velocities move, through LDS
      const int ra = body_rd(bodyA), rb = body_rd(bodyB), wa = body_wr(bodyA), wb_ = body_wr(bodyB);
      for (int t = 0; t < nSteps; ++t) {
        if (active && t == due && passes < 10) {
          a.vb = v2(L.vbx[ra], L.vby[ra]); a.wb = L.wb[ra];
          b.vb = v2(L.vbx[rb], L.vby[rb]); b.wb = L.wb[rb];
          arb_apply_bias_only(a, b, n, r1, r2, nMass, bias, jBias, a_count);
          L.vbx[wa] = a.vb.x; L.vby[wa] = a.vb.y; L.wb[wa] = a.wb;
          L.vbx[wb_] = b.vb.x; L.vby[wb_] = b.vb.y; L.wb[wb_] = b.wb;
          due += period; ++passes;
        }
        __syncthreads();
      }
    } else [[unlikely]] {
      // the general sweeps are a function of their own (drv_solve_general_split).  NOTHING of this frame lives across the call:
      // what the verdicts below need - the packed slot state, the pair, the tangent impulses, three predicates - travels through
      // the callee's registers and comes back in its return value (a value kept across the call would be spilled, and the
      // allocator then spills it across the whole function, the bias-only loops that every pile-up runs included: measured)
      const int bits3 = (restIn ? 1 : 0) | (bias[0] == 0.0 ? 2 : 0) | (bias[1] == 0.0 ? 4 : 0);
      const DrvSplitRet sr = drv_solve_general_split(lane, myLevel, bodyA | (bodyB << 8), maxLevel_, nMass[0], nMass[1], bias[0], bias[1], bounce[0],
                                                     bounce[1], jn[0], jn[1], jt[0], jt[1], pk, a_pair, bits3);
      lane = fresh_lane();
      pk = sr.pk; a_pair = sr.pair;
      a_state = pk & 0xFF; a_count = (pk >> 8) & 0xFF; a_age = (pk >> 16) & 0xFF;
      touched = (pk >> 24) & 1; freeMe = (pk >> 25) & 1; hashSame = (pk >> 26) & 1; prevInert = (pk >> 27) & 1;
      skipped = (pk >> 28) & 1; slotOcc = (pk >> 29) & 1; active = (pk >> 30) & 1;
      restIn = (sr.bits & 1) != 0;
      bias[0] = (sr.bits & 2) ? 0.0 : 1.0; bias[1] = (sr.bits & 4) ? 0.0 : 1.0;  // (only compared with zero from here on)
      jn[0] = sr.jn0; jn[1] = sr.jn1; jBias[0] = sr.jb0; jBias[1] = sr.jb1; jt[0] = sr.jt0; jt[1] = sr.jt1;
      tookSplit = true;
    }
  }
DRV_PROF(const unsigned long long P3 = __builtin_amdgcn_s_memtime();)
  // arbiters that were active this step are NORMAL from the next step on (cpSpaceStep resets the state)
  const bool wasNormal = a_state == ARB_NORMAL;  // i.e. not a first contact in this substep
  if (active && a_state == ARB_FIRST) a_state = ARB_NORMAL;
  // steady: re-running this slot on identical inputs (same frozen positions, bodies at rest) reproduces this substep
  // bit for bit: the slot record is unchanged (same contact ids, same accumulated impulses, NORMAL before and after, or
  // ignored) and both bodies were at rest before the prestep and after the solve.  See DESIGN.md "steady replay".
  bool steady = true;
  if (slotOcc && !skipped) {
    steady = touched && !freeMe && hashSame;
    if (steady && a_state != ARB_IGNORE) {
      steady = a_state == ARB_NORMAL && wasNormal && restIn && L.s_jn0[lane] == jn[0] && L.s_jt0[lane] == jt[0] &&
               L.s_jn1[lane] == jn[1] && L.s_jt1[lane] == jt[1];
      if (steady) {
        const int i = a_pair >> 8, j = a_pair & 0xFF;
        steady = L.vx[i] == 0.0 && L.vy[i] == 0.0 && L.w[i] == 0.0;
        if (j < DRV_SLOT_OBST) steady = steady && L.vx[j] == 0.0 && L.vy[j] == 0.0 && L.w[j] == 0.0;
      }
    }
  }
  const bool allSteady = wave_ballot(!steady) == 0ull;
  // inert: touched, not first contact, and (ignored | zero bias and zero accumulated impulses on every contact)
  bool inert = true;
  if (slotOcc && skipped) inert = prevInert;
  else if (slotOcc) {
    inert = touched && !freeMe &&
            (a_state == ARB_IGNORE ||
             (a_state == ARB_NORMAL && wasNormal && bias[0] == 0.0 && bias[1] == 0.0 && jn[0] == 0.0 && jt[0] == 0.0 &&
              jn[1] == 0.0 && jt[1] == 0.0 && jBias[0] == 0.0 && jBias[1] == 0.0));
  }
  if (slotOcc) {
    if (freeMe) L.s_pair[lane] = 0xFFFF;
    L.s_meta[lane] = a_state | (a_count << 8) | (a_age << 16) | (steady ? (1 << 24) : 0) | (inert ? (1 << 25) : 0);
    if (touched) { L.s_jn0[lane] = jn[0]; L.s_jt0[lane] = jt[0]; L.s_jn1[lane] = jn[1]; L.s_jt1[lane] = jt[1]; }
  }
  const bool allInert = wave_ballot(!inert) == 0ull;
DRV_PROF(if (lane == 0 && blockIdx.x < 4096) { unsigned long long* d = g_dbgp + blockIdx.x * 8; const unsigned long long P4 = __builtin_amdgcn_s_memtime(); d[2] += (P1 - P0) + ((P4 - P3) << 32); d[3] += (P2 - P1) + ((P3 - P2) << 32); })
  return (allInert ? 2 : 0) | (allSteady ? 4 : 0) | (tookSplit ? 8 : 0) DRV_PROF(| (prof_any(profMode) << 4));
}
struct ContactRet {
  uint64_t occ;
  int err;
};

__device__ DRV_CONTACT_INLINE ContactRet drv_contact_path(int lane, int cand, int dirty, int light, int A, int nCarPed, uint64_t occ) {
  const bool isCar = lane < A, isPed = lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + (nCarPed >> 8);
  DrvLds& L = g_L;
  int err = 0;
  // The lane id is made opaque here: everything this (inlined) function derives from it - quad roles, lane masks, slot
  // predicates - is then recomputed per call (a few integer instructions) instead of being hoisted out of the kernel's substep
  // loop and kept alive, i.e. saved to and reloaded from scratch, across every call the loop makes.
  asm volatile("" : "+v"(lane));
  // ---------- slow path: narrowphase -> arbiter cache -> callbacks -> prestep -> friction -> solver -------
DRV_PROF(const unsigned long long T0 = __builtin_amdgcn_s_memtime();)
  DrvMailbox& M = L.u.mb;
  if (lane < DRV_NS) M.flag[lane] = 0;
  __syncthreads();
  // Two modes share one instance of the narrowphase.  mode 0 ("light", only when every slot was steady in the previous
  // substep): test just the DIRTY candidates - pairs that are new or have a thawed body.  If none of them touches and
  // none owns a slot, every remaining pair is unchanged and frozen, so the rest of this function would reproduce the
  // previous substep: return and let the caller replay it.  Otherwise, and in mode 1, process every candidate.
  bool lightOk = false;
  int savedN = uniform_i(L.clistN);  // (the caller resets it whenever a lane's candidate mask changes)
  int spair = lane < DRV_NS ? L.s_pair[lane] : -1;  // the slots' pairs, for the slot search below (one load, then v_readlane per occupied slot)
DRV_PROF(int profCand = 0;)
#pragma unroll 1
  for (int mode = light ? 0 : 1; mode < 2; ++mode) {
  const int bits = mode == 0 ? dirty : cand;
  bool lightBad = false;
  // ---- compact the pairs (bit i of lane j = pair (i, j)) into one dense list in canonical order, so that the
  //      narrowphase runs once over up to 64 pairs.  The full list (mode 1) of unchanged candidate masks is still there.
  int nCand = 0;
  if (mode == 1 && savedN >= 0) nCand = savedN;
  else {
  // (straight-line over the ten car bits - `bits` has none at or above A: ten ballots and masked stores, no loop-carried wait)
#pragma unroll
  for (int i = 0; i < DRV_MAXA; ++i) {
    const bool c = (bits >> i) & 1;
    const uint64_t m = wave_ballot(c);
    if (c) {
      const int idx = nCand + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (idx < DRV_CLIST) L.clist[idx] = (unsigned short)((i << 8) | lane); else err |= 1;  // overflow is reported
    }
    nCand += __popcll(m);
  }
  savedN = mode == 1 && nCand <= DRV_CLIST ? nCand : -1;  // (the dirty list of mode 0 overwrites the full one; an overflow is reported by every build)
  if (lane == 0) L.clistN = savedN;
  if (nCand > DRV_CLIST) nCand = DRV_CLIST;
  }
DRV_PROF(profCand += nCand;)
  __syncthreads();
DRV_PROF(const unsigned long long N1 = __builtin_amdgcn_s_memtime(); unsigned long long nMath = 0ull;)
  if (mode == 0) {  // a dirty pair that owns a slot rules the light mode out before any narrowphase work is spent on it
    bool owns = false;
    for (int k = lane; k < nCand; k += 64) {
      const int pr = (int)L.clist[k];
      for (uint64_t mm = occ; mm; mm &= mm - 1) owns = owns || bcast_i(spair, __builtin_ctzll(mm)) == pr;
    }
    if (wave_ballot(owns) != 0ull) continue;
  }
#pragma unroll 1
  for (int pass = 0; pass * 16 < nCand; ++pass) {  // 16 pairs per pass, one DPP quad of lanes each
DRV_PROF(nMath -= __builtin_amdgcn_s_memtime();)
    const int q = lane & 3;
    const bool isCand = pass * 16 + (lane >> 2) < nCand;
    Contacts ct;
    ct.count = 0;
    const int pr = isCand ? (int)L.clist[pass * 16 + (lane >> 2)] : 0xFFFF;
    if (isCand) {
      const int i = pr >> 8, j = pr & 0xFF;
      BoxP b1;
      b1.p = v2(L.px[i], L.py[i]); b1.c = L.rc[i]; b1.s = L.rs[i]; b1.hx = L.chx[i]; b1.hy = L.chy[i];
      if (j >= DRV_SLOT_PED && j < DRV_SLOT_OBST) {
        circle_to_poly(v2(L.px[j], L.py[j]), 5.0, b1, q, ct);
      } else {  // car or static box: one instance of the SAT + clipping code for both
        BoxP b2;
        b2.c = 1.0; b2.s = 0.0;
        if (j < DRV_SLOT_PED) { b2.p = v2(L.px[j], L.py[j]); b2.c = L.rc[j]; b2.s = L.rs[j]; b2.hx = L.chx[j]; b2.hy = L.chy[j]; }
        else { b2.p = static_pos(L, j); b2.hx = j >= DRV_SLOT_BLD ? 400.0 : 10.0; b2.hy = j >= DRV_SLOT_BLD ? 225.0 : 10.0; }
        poly_to_poly(b1, i, b2, j, q, ct);
      }
    }
DRV_PROF(nMath += __builtin_amdgcn_s_memtime();)
    const bool touch = isCand && q == 0 && ct.count > 0;  // lane 0 of the quad speaks for the pair
    // find my slot among the occupied ones
    int slot = -1;
    if (mode != 0 && wave_ballot(touch) != 0ull) {
      for (uint64_t mm = occ; mm; mm &= mm - 1) {
        int sidx = __builtin_ctzll(mm);
        if (bcast_i(spair, sidx) == pr) slot = sidx;
      }
      if (!touch) slot = -1;
    }
    if (mode == 0) { lightBad = lightBad || touch; continue; }
    const uint64_t tmask = wave_ballot(touch);
    if (tmask == 0ull) continue;
    const bool needNew = touch && slot < 0;
    const uint64_t newMask = wave_ballot(needNew);
    if (newMask) {
      const uint64_t slotBits = (1ull << DRV_NS) - 1ull;
      int rank = __popcll(newMask & ((1ull << lane) - 1ull));
      uint64_t fm = (~occ) & slotBits;
      if (needNew) {
        for (int r = 0; r < rank; ++r) fm &= fm - 1;
        if (fm) { slot = __builtin_ctzll(fm); L.s_pair[slot] = pr; }
        else err |= 1;  // contact cache overflow: pair dropped (reported through EI_ERR)
      }
      int cnt = __popcll(newMask);
      uint64_t fm2 = (~occ) & slotBits;
      for (int r = 0; r < cnt && fm2; ++r) { occ |= (fm2 & (~fm2 + 1)); fm2 &= fm2 - 1; }
      spair = lane < DRV_NS ? L.s_pair[lane] : -1;  // (slots were handed out: a later pass must see their pairs)
    }
    if (touch && slot >= 0) {
      M.flag[slot] = needNew ? 3 : 1;
      M.count[slot] = ct.count;
      M.nx[slot] = ct.n.x; M.ny[slot] = ct.n.y;
      M.p1x[slot][0] = ct.p1[0].x; M.p1y[slot][0] = ct.p1[0].y; M.p2x[slot][0] = ct.p2[0].x; M.p2y[slot][0] = ct.p2[0].y;
      M.hash[slot][0] = ct.hash[0];
      if (ct.count > 1) {
        M.p1x[slot][1] = ct.p1[1].x; M.p1y[slot][1] = ct.p1[1].y; M.p2x[slot][1] = ct.p2[1].x; M.p2y[slot][1] = ct.p2[1].y;
        M.hash[slot][1] = ct.hash[1];
      }
    }
    __syncthreads();
  }
  __syncthreads();
DRV_PROF(if (lane == 0 && blockIdx.x < 4096) { unsigned long long* d = g_dbgl + blockIdx.x * 8; d[6] += (N1 - T0) + (nMath << 32); d[7] += __builtin_amdgcn_s_memtime() - N1 - nMath; })
  if (mode == 0 && wave_ballot(lightBad) == 0ull) { lightOk = true; break; }
  }
  if (lightOk) {
    ContactRet ret;
    ret.occ = occ; ret.err = err | 8;
    return ret;
  }

DRV_PROF(const unsigned long long T1 = __builtin_amdgcn_s_memtime();)
  // ---- slot lanes: cpArbiterUpdate for touched slots; ageing / expiry for the rest ----------------------
  const bool slotOcc = lane < DRV_NS && ((occ >> lane) & 1ull);
  bool touched = false;
  int bodyA = 0, bodyB = 0;
  int a_pair = 0xFFFF, a_state = ARB_FIRST, a_count = 0, a_age = 0;
  bool hashSame = false;  // same contact ids in the same order as when the slot was last written
  bool prevSteady = false, prevInert = false;  // how this slot came out of its previous evaluation (persisted in s_meta)
  double jn[2] = {0.0, 0.0}, jt[2] = {0.0, 0.0};
  V2 n = v2(0.0, 0.0), r1[2], r2[2];
  r1[0] = r1[1] = r2[0] = r2[1] = v2(0.0, 0.0);
  if (slotOcc) {
    const int flag = M.flag[lane];
    touched = flag != 0;
    a_pair = L.s_pair[lane];
    int meta = L.s_meta[lane];
    a_state = meta & 0xFF; a_count = (meta >> 8) & 0xFF; a_age = (meta >> 16) & 0xFF;
    prevSteady = (meta >> 24) & 1; prevInert = (meta >> 25) & 1;
    if (flag & 2) { a_state = ARB_FIRST; a_count = 0; a_age = 0; prevSteady = false; prevInert = false; }
    if (touched) {
      const int i = a_pair >> 8, j = a_pair & 0xFF;
      // narrowphase order: shape type ascending => pedestrian circle first for car-ped pairs
      if (j >= DRV_SLOT_PED && j < DRV_SLOT_OBST) { bodyA = j; bodyB = i; } else { bodyA = i; bodyB = j; }
      const V2 pa = bodyA < DRV_SLOT_OBST ? v2(L.px[bodyA], L.py[bodyA]) : static_pos(L, bodyA);
      const V2 pb = bodyB < DRV_SLOT_OBST ? v2(L.px[bodyB], L.py[bodyB]) : static_pos(L, bodyB);
      const int cnt = M.count[lane];
      const int h0 = M.hash[lane][0], h1 = cnt > 1 ? M.hash[lane][1] : 0;
      const int oh0 = L.s_hash0[lane], oh1 = L.s_hash1[lane];
      // carry impulses of contacts with matching hash (later match wins, as in Chipmunk's loop)
      if (a_count > 0 && h0 == oh0) { jn[0] = L.s_jn0[lane]; jt[0] = L.s_jt0[lane]; }
      if (a_count > 1 && h0 == oh1) { jn[0] = L.s_jn1[lane]; jt[0] = L.s_jt1[lane]; }
      if (cnt > 1) {
        if (a_count > 0 && h1 == oh0) { jn[1] = L.s_jn0[lane]; jt[1] = L.s_jt0[lane]; }
        if (a_count > 1 && h1 == oh1) { jn[1] = L.s_jn1[lane]; jt[1] = L.s_jt1[lane]; }
      }
      r1[0] = vsub(v2(M.p1x[lane][0], M.p1y[lane][0]), pa);
      r2[0] = vsub(v2(M.p2x[lane][0], M.p2y[lane][0]), pb);
      if (cnt > 1) {
        r1[1] = vsub(v2(M.p1x[lane][1], M.p1y[lane][1]), pa);
        r2[1] = vsub(v2(M.p2x[lane][1], M.p2y[lane][1]), pb);
      }
      n = v2(M.nx[lane], M.ny[lane]);
      hashSame = a_count == cnt && h0 == oh0 && (cnt < 2 || h1 == oh1);
      a_count = cnt;
      L.s_hash0[lane] = h0; L.s_hash1[lane] = h1;
      if (a_state == ARB_CACHED) a_state = ARB_FIRST;
      a_age = 0;
    }
  }

DRV_PROF(const unsigned long long U1 = __builtin_amdgcn_s_memtime();)
  // ---- rank touched slots by canonical pair order ------------------------------------------------------
  // (the rank is needed by the begin callbacks and by the level computation; a call without a first contact whose active set is the
  //  previous call's needs neither - see the level cache below)
  const uint64_t touchedMask = wave_ballot(touched);
  const int nTouched = __popcll(touchedMask);
  // the cached schedule's key and values: loads issued here, used after the closure
  const unsigned long long c_active = L.sActive;
  const int c_meta = L.sLvMeta;
  const int c_level = lane < DRV_NS ? (int)L.sLevel[lane] : 0;
  int rank = 0;
  bool haveRank = false;
  const bool anyFirst = wave_ballot(touched && a_state == ARB_FIRST) != 0ull;
  if (anyFirst) {
    haveRank = true;
    if (nTouched > 1)  // (a lone touched slot has rank 0)
      for (uint64_t mm = touchedMask; mm; mm &= mm - 1) {
        int b = __builtin_ctzll(mm);
        int pk = bcast_i(a_pair, b);
        rank += (pk < a_pair) ? 1 : 0;
      }
  }

DRV_PROF(const unsigned long long U2 = __builtin_amdgcn_s_memtime();)
  // ---- begin callbacks in canonical order (first contact only) ------------------------------------------
  if (anyFirst)  // (most calls have no first contact: the loop would only skip)
    for (int k = 0; k < nTouched; ++k) {
      uint64_t who = wave_ballot(touched && rank == k);
      int b = __builtin_ctzll(who);
      int st = bcast_i(a_state, b);
      if (st != ARB_FIRST) continue;
      int pk = bcast_i(a_pair, b);
      bool keep = cb_begin(L, pk >> 8, pk & 0xFF, lane);
      if (!keep && lane == b) a_state = ARB_IGNORE;
      __syncthreads();
    }

DRV_PROF(const unsigned long long U3 = __builtin_amdgcn_s_memtime();)
  // ---- expiry of untouched slots (cpSpaceArbiterSetFilter; no `separate` handlers in Driving) -----------
  bool freeMe = false;
  if (slotOcc && !touched) {
    a_age += 1;
    if (a_state != ARB_CACHED) a_state = ARB_CACHED;
    if (a_age >= 3) freeMe = true;
  }
  const uint64_t freeMask = wave_ballot(freeMe);

  // ---- component replay.  A touched, solvable slot is CLEAN if it came out of its previous evaluation steady, is not a
  // first contact, has the same contact ids, and both its bodies are frozen (at rest, position unchanged since the
  // previous substep): evaluating it again would read the same inputs.  That only reproduces the previous outputs if
  // the whole connected component of the contact graph (arbiters sharing a dynamic body) is clean, because the
  // Gauss-Seidel sweep couples exactly those.  Dirtiness therefore spreads from every other occupied slot over shared
  // dynamic bodies until it stops; clean components are skipped: their slots stay as they are, their bodies keep zero
  // velocity and the bias velocities of the previous solve (still in L.vb*, see vbValid).
  const bool solvable = touched && a_state != ARB_IGNORE;
  bool cleanSlot = false;
  int myBodies = 0;  // bit b: dynamic body b belongs to my slot
  if (slotOcc) {
    const int pi = a_pair >> 8, pj = a_pair & 0xFF;
    myBodies = (1 << pi) | (pj < DRV_SLOT_OBST ? (1 << pj) : 0);
    cleanSlot = solvable && prevSteady && a_state == ARB_NORMAL && hashSame && (L.still[pi] & 2) &&
                (pj >= DRV_SLOT_OBST || (L.still[pj] & 2));
  }
  int dirtyBodies = 0;
  if (wave_ballot(cleanSlot) != 0ull) {  // (usually nothing is clean: no closure to compute)
    const uint64_t occLanes = wave_ballot(slotOcc);
    bool slotDirty = slotOcc && !cleanSlot && !(touched && a_state == ARB_IGNORE);
    for (int round = 0; round < DRV_NS; ++round) {
      int m = 0;
      for (uint64_t mm = occLanes & wave_ballot(slotDirty); mm; mm &= mm - 1) m |= bcast_i(myBodies, __builtin_ctzll(mm));  // (dirty slots only)
      if (m == dirtyBodies) break;
      dirtyBodies = m;
      // (if no clean slot touches a dirty body, nothing spreads: the next round would find the same set - the usual case, a
      //  frozen pair somewhere else in the scene beside the pile that is being relaxed)
      const bool spreads = cleanSlot && !slotDirty && (myBodies & dirtyBodies) != 0;
      if (wave_ballot(spreads) == 0ull) break;
      slotDirty = slotDirty || spreads;
    }
    cleanSlot = cleanSlot && (myBodies & dirtyBodies) == 0;
  }
  const bool skipped = cleanSlot;  // solved before with identical inputs: replayed
  {  // bias velocities restart from zero, except on the bodies of skipped components (they keep the previous solve's)
    int keep = 0;
    const uint64_t skipLanes = wave_ballot(skipped);
    for (uint64_t mm = skipLanes; mm; mm &= mm - 1) keep |= bcast_i(myBodies, __builtin_ctzll(mm));
    if ((isCar || isPed) && !((keep >> lane) & 1)) { L.vbx[lane] = 0.0; L.vby[lane] = 0.0; L.wb[lane] = 0.0; }
  }

DRV_PROF(const unsigned long long U4 = __builtin_amdgcn_s_memtime();)
  // ---- active arbiters: levels (arbiters sharing a dynamic body keep their canonical sequential order) --
  const bool active = solvable && !skipped;
  const uint64_t activeMask = wave_ballot(active);
  // `period` (pipelined sweeps, see drv_prestep_solve): 1 + the largest level difference between two arbiters that share a
  // dynamic body.  The first arbiter on a body has the lowest level of all arbiters on it (levels ascend along a body's
  // arbiters), so the difference to that one (bfirst) is the largest difference this arbiter has with any earlier one.
  int myLevel = 0, maxLevel = -1, period = 1;
  const int nActive = __popcll(activeMask);
  // The schedule is a function of the active arbiters' pairs alone (canonical order = ascending pair id, levels from shared bodies).
  // sActive is the active mask of the PREVIOUS full evaluation (every one records its own).  An arbiter that is active in two
  // consecutive evaluations was touched in both, so its slot was neither freed nor handed out again in between: the same active mask
  // means the same pairs, and the previous evaluation's schedule is this one's.
  const bool lvHit = uniform_u64(c_active) == activeMask;
  if (!lvHit && lane == 0) L.sActive = activeMask;
  if (nActive == 1) maxLevel = 0;  // a lone active arbiter: level 0, nothing shared
  else if (nActive > 1 && lvHit) {
    myLevel = c_level;
    maxLevel = (int)(signed char)(uniform_i(c_meta) & 0xFF);
    period = uniform_i(c_meta) >> 8;
  } else if (nActive > 1) {
    if (!haveRank && nTouched > 1)
      for (uint64_t mm = touchedMask; mm; mm &= mm - 1) {
        int b = __builtin_ctzll(mm);
        int pk = bcast_i(a_pair, b);
        rank += (pk < a_pair) ? 1 : 0;
      }
    // per body lane: level the next arbiter on this body gets | (level of the first arbiter on it + 1) << 8 (0: none yet) - one
    // v_readlane per body instead of two, and the arbiter's two bodies come over in one
    int bl = 0;
    const int bodyAB = bodyA | (bodyB << 8);
    for (int k = 0; k < nTouched; ++k) {
      uint64_t who = wave_ballot(active && rank == k);
      if (who == 0ull) continue;
      int b = __builtin_ctzll(who);
      const int ab = bcast_i(bodyAB, b), ba = ab & 0xFF, bb2 = ab >> 8;
      const int pa = ba < DRV_SLOT_OBST ? bcast_i(bl, ba) : 0, pb = bb2 < DRV_SLOT_OBST ? bcast_i(bl, bb2) : 0;
      const int la = pa & 0xFF, lb = pb & 0xFF;
      int lv = la > lb ? la : lb;
      const int fa = (pa >> 8) - 1, fb = (pb >> 8) - 1;
      const int lo = fa < 0 ? fb : (fb < 0 ? fa : (fa < fb ? fa : fb));  // lowest level of an earlier arbiter sharing a body (-1: none)
      if (lo >= 0 && lv - lo + 1 > period) period = lv - lo + 1;
      if (lane == b) myLevel = lv;
      if (lane == ba || lane == bb2) bl = (lv + 1) | (((bl >> 8) ? (bl >> 8) : lv + 1) << 8);  // static indices (>= 30) never equal a body lane (< 30)
      maxLevel = lv > maxLevel ? lv : maxLevel;
    }
    if (lane < DRV_NS) L.sLevel[lane] = (unsigned char)myLevel;
    if (lane == 0) L.sLvMeta = (maxLevel & 0xFF) | (period << 8);
  }

DRV_PROF(const unsigned long long T2 = __builtin_amdgcn_s_memtime();)
DRV_PROF(if (lane == 0 && blockIdx.x < 4096) { unsigned long long* d = g_dbgs + blockIdx.x * 8; d[0] += U1 - T1; d[1] += U2 - U1; d[2] += U3 - U2; d[3] += U4 - U3; d[4] += T2 - U4; })
  // everything per-slot from here on - prestep, solve, the steady / inert verdicts and the slot record - happens inside the
  // function: nothing of the contact cache stays live in this frame across the call
  const int solveBits = drv_prestep_solve(lane, nCarPed,
                                          (a_state & 0xFF) | ((a_count & 0xFF) << 8) | ((a_age & 0xFF) << 16) | (touched ? 1 << 24 : 0) | (freeMe ? 1 << 25 : 0) |
                                              (hashSame ? 1 << 26 : 0) | (prevInert ? 1 << 27 : 0) | (skipped ? 1 << 28 : 0) | (slotOcc ? 1 << 29 : 0) | (active ? 1 << 30 : 0),
                                          a_pair, bodyA, bodyB, myLevel, (maxLevel & 0xFF) | (period << 8), activeMask != 0ull ? 1 : 0, jn[0], jn[1], jt[0], jt[1]);
  occ &= ~freeMask;
DRV_PROF(const unsigned long long T3 = T2, T4 = T2, T5 = __builtin_amdgcn_s_memtime(); const int profModeW = (uniform_i(solveBits) >> 4) & 7;)
DRV_PROF(if (lane == 0 && blockIdx.x < 4096) { unsigned long long* d = g_dbgp + blockIdx.x * 8; d[0] += T1 - T0; d[1] += T2 - T1; d[2] += T3 - T2; d[3] += T4 - T3; d[4] += T5 - T4; d[5] += 1ull + (light ? (1ull << 16) : 0ull); d[6] += (unsigned long long)(maxLevel + 1) + ((unsigned long long)(maxLevel + 1) << (12 * profModeW)); d[7] += (unsigned long long)nTouched + ((unsigned long long)profCand << 16); })
  {
    ContactRet ret;
    ret.occ = occ; ret.err = err | (uniform_i(solveBits) & 6) | ((uniform_i(solveBits) & 8) << 1);  // bit 4: split-lane sweeps
    return ret;
  }
}

// ------------------------------------------------------------------------------------------------
// THE step kernel: grid = E blocks of one wavefront
// ------------------------------------------------------------------------------------------------
DE_OOL int drv_partial_obs_fused(PvIn in, uint64_t seed, int A, int* envi, int env_id_offset, int e, int nPedObst, int elapsed,
                                 uint32_t episode, int noiseTypeAgents, double magn, float* __restrict__ obs, int budgetCycles);  // driving_partial.hip
#ifndef DRV_DEFER_MIN_CONTACT
#define DRV_DEFER_MIN_CONTACT 5 /* without a forecast: contact-path substeps (of 10) from which an environment defers its Partial observation */
#endif
#ifndef DRV_PV_DEADLINE_PCT
#define DRV_PV_DEADLINE_PCT 110 /* fused vision passes start until this many percent of the forecast of the launch's slowest environment (round 6 sweep: 90 +2.5 %, 100 0, 105 -0.7 %, 110 -0.9 %, 115 -0.1 %, 120 +1.5 %) */
#endif
#ifndef DRV_FUSED_AGENTS
#define DRV_FUSED_AGENTS 10 /* without a forecast: agent passes a light environment runs in the step launch */
#endif
struct DrvLightRet {
  int cand, dirty, bits;
};
struct DrvSeedOnly { uint64_t seed; };
DE_OOL DrvLightRet drv_light_substep(int it_, int lane, int A_, int nPed_, int nObst_, int elapsed_,
                                                      uint32_t seedLo, uint32_t seedHi, uint32_t genv_, uint32_t episode_, int stateBits) {
  DrvLds& L = g_L;
DRV_PROF(const unsigned long long Q0 = __builtin_amdgcn_s_memtime();)
  const int lastCand = L.lastCand[lane];
  const int it = uniform_i(it_), A = uniform_i(A_), nPed = uniform_i(nPed_), nObst = uniform_i(nObst_), elapsed = uniform_i(elapsed_);
  const uint32_t genv = (uint32_t)uniform_i((int)genv_), episode = (uint32_t)uniform_i((int)episode_);
  DrvSeedOnly S;
  S.seed = ((uint64_t)(uint32_t)uniform_i((int)seedHi) << 32) | (uint64_t)(uint32_t)uniform_i((int)seedLo);
  const bool isCar = lane < A;
  const bool isPed = lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed;
  const bool isBody = isCar || isPed;
  const bool vbValid = (uniform_i(stateBits) & 1) != 0;
  bool aabbValid = (uniform_i(stateBits) & 2) != 0;
    bool turned = false;  // Car.turn rotated the body in place: geometry changed even if every velocity is zero
    // ---- ONE batch of LDS loads: everything the lane's role reads below, issued before the first use (a lone wave cannot hide an
    //      LDS round trip, ~110 cycles, and the phases below would otherwise make a dozen of them one after the other).  Indices are
    //      clamped into the arrays, the values of lanes without the role are never used.  What this function changes before a later
    //      use (velocity, angle, rotation cache) is tracked in these registers.
    const int bi = lane & (DRV_NB - 1), ci = lane & 15;
    int f = L.flags[bi];
    double px = L.px[bi], py = L.py[bi], vx = L.vx[bi], vy = L.vy[bi], ang = L.ang[bi];
    const double w = L.w[bi];
    const double vbx = vbValid ? L.vbx[bi] : 0.0, vby = vbValid ? L.vby[bi] : 0.0, wb = vbValid ? L.wb[bi] : 0.0;
    double rcv = L.rc[ci], rsv = L.rs[ci], rotAngv = L.rotAng[ci], cosRel0v = L.cosRel0[ci];
    const double goalx = L.goalx[ci], goaly = L.goaly[ci], dprev = L.dprev[ci], chx = L.chx[ci], chy = L.chy[ci];
    double rew = L.rewAcc[ci], posrew = L.posAcc[ci];
    int moving = L.moving[bi];
    if (!isBody) { f = 0; px = 0.0; py = 0.0; vx = 0.0; vy = 0.0; }
    // ---- common prefix of cars and pedestrians, ONE instruction stream for both (the two roles sit on different lanes of the
    //      wave: what each does on its own is executed one after the other): after processAction, the lane classification of the
    //      position - min over the two roads of Road.isPointOnRoad - with the car's cached cos(road - angle) or, for a pedestrian
    //      (isOffRoad, angle 0), the road's constant
    if (isCar) {
      if (it == 0) {  // processAction :357-373 -> Car.accelerate (Car.py:55-94), Car.turn (Car.py:97-108)
        const int actPk = (int)L.act[lane];
        const int acc = (actPk & 3) - 1, steer = ((actPk >> 2) - 1) * 2;
        if (!CF_FIN(f)) {
          double dirx = L.dirx[lane], diry = L.diry[lane];
          double power = (double)acc;
          double moveDir = vx * dirx + vy * diry;
          bool skip = false;
          if (acc < 0) power = (double)acc * 0.75;
          if (acc == 0) power = (moveDir == 0.0) ? 0.0 : (moveDir > 0.0 ? -2.0 : 2.0);
          else if (acc < 0 && moveDir > 0.0) skip = true;
          else if (acc > 0 && moveDir < 0.0) skip = true;
          if (!skip) {
            const double cs = rcv, sn = rsv;  // = dm_sincos(ang): the cache is refreshed on every change
            const double cpower = L.cpower[lane];
            vx = vx + cpower * power * cs;
            vy = vy + cpower * power * sn;
            if (acc == 0 && (vx * dirx + vy * diry) * moveDir < 0.0) { vx = 0.0; vy = 0.0; }
          }
          if (steer != 0) {
            const double rot = (double)steer * (DM_PI / 180.0);
            ang = ang + rot;
            const DevSC rsc = dev_sincos_inl(rot);
            const double sn = rsc.s, cs = rsc.c;
            const double dx = dirx * cs - diry * sn, dy = dirx * sn + diry * cs;
            L.dirx[lane] = dx; L.diry[lane] = dy;
            const double nvx = vx * cs - vy * sn, nvy = vx * sn + vy * cs;
            vx = nvx; vy = nvy;
            L.ang[lane] = ang;
            const CarRot cr = car_refresh_rot<true>(L, lane, ang);
            rcv = cr.c; rsv = cr.s; rotAngv = ang; cosRel0v = cr.cosRel0;
            turned = true;
          }
        }
      }
    }
DRV_PROF(const unsigned long long Q1 = __builtin_amdgcn_s_memtime();)
    const V2 pos = v2(px, py);
    int lp = LP_OffRoad;
    if (isBody) {
      int rp = road_pos<0>(pos, isCar ? cosRel0v : RoadK<0>::cosDir0);
      if (rp < lp) lp = rp;
      rp = road_pos<1>(pos, isCar ? rcv : RoadK<1>::cosDir0);
      if (rp < lp) lp = rp;
    }
DRV_PROF(const unsigned long long Q2 = __builtin_amdgcn_s_memtime();)
    if (isCar) {
      // tick :376-426
      const double dnow = vlen(vsub(pos, v2(goalx, goaly)));
      const double diff = dprev - dnow;
      if (!CF_FIN(f)) { const double d50 = diff / 50.0; rew += d50; posrew += dm_max(0.0, d50); }
      L.prevx[lane] = px; L.prevy[lane] = py; L.dprev[lane] = dnow;
      if (lp >= LP_OverRoad) {
        if (!CF_FIN(f)) {
          if (lp == LP_OverRoad && dnow < 100.0) {
            lp = LP_AtGoal;
            f |= (1 << 4) | (1 << 6);  // finished, friction_car_crashed
            rew += (double)(DRV_MAX_TIME - elapsed) / 100.0;
            posrew += (double)(DRV_MAX_TIME - elapsed) / 100.0;
          } else {
            f |= CF_CRASH_BITS;
            rew -= vlen(v2(vx, vy)) / 5.0;
          }
        }
      } else if (lp == LP_InOpposingLane) {
        if (!CF_FIN(f)) rew -= vlen(v2(vx, vy)) / 10000.0;
      }
      f = CF_SET_LP(f, lp);
      if (px >= DRV_W + 50.0 || px <= -50.0 || py >= DRV_H + 50.0 || py <= -50.0) { vx = 0.0; vy = 0.0; }  // prevPos == pos here
      L.flags[lane] = f;
      L.vx[lane] = vx; L.vy[lane] = vy;
      L.rewAcc[lane] = rew; L.posAcc[lane] = posrew;
DRV_PROF(asm volatile("" ::: "memory");)
    } else if (isPed) {
      // ======== phase 1b: pedestrian FSM (move :429-506) ====================================================
      if (!PF_DEAD(f)) {
        int crossing = PF_CROSSING(f), beginc = PF_BEGIN(f), side = PF_SIDE(f);
        const bool isOffRoad = lp >= LP_OverRoad;  // drv_is_off_road(pos)
        const bool isOut = drv_is_out(pos);
        if (moving > 0) {
          moving = (moving - DRV_TIME_DIFF > 0) ? moving - DRV_TIME_DIFF : 0;
          if (crossing) {
            if (!beginc && isOffRoad) { moving = 0; crossing = 0; vx = 0.0; vy = 0.0; }
            else if (beginc && !isOffRoad) { beginc = 0; }
          }
          if (isOut) { moving = 0; vx = 0.0; vy = 0.0; }
        } else {
          if (!crossing) {
            dm_u32x4 u = dm_env_rng(S.seed, genv, episode, DM_RNG_PED_MOVE, (uint32_t)(lane - DRV_SLOT_PED), (uint32_t)elapsed);
            const bool r1 = PF_ROAD(f) != 0;
            const V2 rdir = r1 ? v2(RoadK<1>::dirx, RoadK<1>::diry) : v2(RoadK<0>::dirx, RoadK<0>::diry);
            const V2 rnrm = r1 ? v2(RoadK<1>::nx, RoadK<1>::ny) : v2(RoadK<0>::nx, RoadK<0>::ny);
            V2 dir = rdir;
            moving = dm_randint(u.v[0], 5000, 30000);
            int speed = dm_randint(u.v[1], -2, 2);
            if (!isOffRoad) {
              crossing = 1; beginc = 0;
              if (speed == 0) speed = 2;
            } else if (isOut) {
              dir = drv_is_out(vadd(pos, rdir)) ? vneg(rdir) : rdir;
            } else if (dm_unit(u.v[2]) < 0.05) {
              crossing = 1; beginc = 1;
              dir = side ? rnrm : vneg(rnrm);
              side = side ? 0 : 1;
              speed = dm_randint(u.v[3], 1, 2);
            }
            const V2 nv = vmul(vmul(dir, (double)PF_SPEED(f)), (double)speed);
            vx = nv.x; vy = nv.y;
          } else if (isOffRoad) {
            crossing = 0; beginc = 0;
          }
        }
        L.moving[lane] = moving;
        L.flags[lane] = PEDF_PACK(PF_ROAD(f), side, 0, crossing, beginc, PF_SPEED(f));
        L.vx[lane] = vx; L.vy[lane] = vy;
      }
    }

DRV_PROF(const unsigned long long Q3 = __builtin_amdgcn_s_memtime();)
    // ======== phase 1c: cpBodyUpdatePosition for every body (one instance of the code for cars + pedestrians) ===
    // Every lane leaves this phase with the box of ITS object (bl, bb, br, bt), `live` and the still / frozen bits in registers:
    // the broadphase below reads only the ten car rows from LDS.  (Lane = object j: cars, pedestrians, obstacles, buildings.)
    double bl = 0.0, bb = 0.0, br = -1.0, bt = -1.0;
    bool live = false;
    int sj = 3;  // statics are always still and frozen
    if (isBody) {
      // bias velocities: L.vb* keep the output of the last contact solve (a replay needs it again); they count only if
      // the previous substep solved or replayed contacts (vbValid), else cpBodyUpdatePosition saw zeros
      const double npx = px + (vx + vbx) * DE_DT, npy = py + (vy + vby) * DE_DT, nang = ang + (w + wb) * DE_DT;
      const bool still = !turned && vx == 0.0 && vy == 0.0 && w == 0.0 && vbx == 0.0 && vby == 0.0 && wb == 0.0;
      L.px[lane] = npx; L.py[lane] = npy; L.ang[lane] = nang;
      // frozen: at rest and the position update was absorbed by rounding (sub-ulp bias velocities of a resting contact)
      const bool frozen = !turned && vx == 0.0 && vy == 0.0 && w == 0.0 && npx == px && npy == py && nang == ang;
      sj = (still ? 1 : 0) | (frozen ? 2 : 0);
      L.still[lane] = sj;
      live = true;
      if (isCar) {
        if (nang != rotAngv) { const CarRot cr = car_refresh_rot<true>(L, lane, nang); rcv = cr.c; rsv = cr.s; }
        if (!aabbValid || !frozen) {  // frozen: same position and rotation => same box
          BoxW bw;
          box_world(bw, v2(npx, npy), rcv, rsv, chx, chy);
          double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            l = fmin_cp(l, bw.v[k].x); r = fmax_cp(r, bw.v[k].x); b = fmin_cp(b, bw.v[k].y); t = fmax_cp(t, bw.v[k].y);
          }
          bl = l - 0.0; bb = b - 0.0; br = r + 0.0; bt = t + 0.0;
          L.aabb[lane][0] = bl; L.aabb[lane][1] = bb; L.aabb[lane][2] = br; L.aabb[lane][3] = bt;
        } else {
          bl = L.aabb[lane][0]; bb = L.aabb[lane][1]; br = L.aabb[lane][2]; bt = L.aabb[lane][3];
        }
      } else {  // pedestrian: a circle of radius 5 at the new position
        bl = npx - 5.0; bb = npy - 5.0; br = npx + 5.0; bt = npy + 5.0;
      }
    } else if (lane >= DRV_SLOT_OBST && lane < DRV_SLOT_BLD + 4) {
      live = lane >= DRV_SLOT_BLD || (lane - DRV_SLOT_OBST) < nObst;
      const V2 c = static_pos(L, lane);
      const double ex = lane >= DRV_SLOT_BLD ? 400.0 : 10.0, ey = lane >= DRV_SLOT_BLD ? 225.0 : 10.0;
      // static box AABB = min/max of (c +- e) exactly as cached by cpShapeCacheBB with rot = (1,0)
      bl = -ex + c.x; br = ex + c.x; bb = -ey + c.y; bt = ey + c.y;
    }
    aabbValid = true;
    __syncthreads();
DRV_PROF(const unsigned long long Q4 = __builtin_amdgcn_s_memtime();)

    // ======== phase 2: broadphase.  The loop runs over the cars i < j whose box is broadcast from LDS.  cand bit i <=>
    // cpBBIntersects(bb_i, bb_j) for the canonical pair (i, j); the pairs of one car are consecutive in canonical order and
    // ascend with the lane.
    int cand = 0;
    bool candMoving = false, removed = false;
    int dirty = 0;
    {
      const int carStill = (int)(wave_ballot(isCar && (sj & 1)) & 0x3FFull);
      const int carFrozen = (int)(wave_ballot(isCar && (sj & 2)) & 0x3FFull);
      // cpBBIntersects(a, b) = a.l <= b.r && b.l <= a.r && a.b <= b.t && b.b <= a.t, branch-free: four compares into scalar masks,
      // three s_and, one select per car.  Pairs that do not exist (i >= A, i >= lane, a lane without an object) are masked out
      // afterwards, whatever their rows held.
#pragma unroll
      for (int i = 0; i < DRV_MAXA; ++i) {  // rows >= A are never written but in bounds: the loads pipeline unconditionally
        const double al = L.aabb[i][0], ab = L.aabb[i][1], ar = L.aabb[i][2], at = L.aabb[i][3];
        const bool hit = (al <= br) & (bl <= ar) & (ab <= bt) & (bb <= at);
        cand |= hit ? (1 << i) : 0;
      }
      {
        const int lim = lane < A ? lane : A;  // pair (i, lane) exists for i < min(lane, A)
        cand &= live ? ((1 << lim) - 1) : 0;
      }
      if (cand) candMoving = !(sj & 1) || (cand & ~carStill) != 0;
      // clean pair: a candidate in the previous substep too, both bodies frozen since.  dirty: every other candidate.
      const int clean = (lastCand >= 0 && (sj & 2)) ? (cand & lastCand & carFrozen) : 0;
      dirty = cand & ~clean;
      removed = lastCand < 0 || (lastCand & ~cand) != 0;
    }
  L.lastCand[lane] = cand;
DRV_PROF(if (lane == 0 && blockIdx.x < 4096) { unsigned long long* d = g_dbgl + blockIdx.x * 8; const unsigned long long Q5 = __builtin_amdgcn_s_memtime(); if (it == 0) { for (int k = 0; k < 8; ++k) d[k] = 0ull; } d[0] += Q1 - Q0; d[1] += Q2 - Q1; d[2] += Q3 - Q2; d[3] += Q4 - Q3; d[4] += Q5 - Q4; d[5] += Q5 - Q0; })
  DrvLightRet ret;
  ret.cand = cand; ret.dirty = dirty; ret.bits = (candMoving ? 1 : 0) | (removed ? 2 : 0) | (cand != lastCand ? 4 : 0);
  return ret;
}
// ------------------------------------------------------------------------------------------------
// SIMD isolation of the slow environments.  A launch lasts as long as its slowest environment, and that environment's wave
// shares its SIMD with three others: their instructions delay it by ~10 % (measured: 458 k cycles alone, 502 k with neighbours).
// Which block steps which environment is free - results do not depend on it - so: the K environments that were slowest in the
// previous step (list built by that step's epilogue) take blocks G0..G0+K-1, the first wave slot of SIMD groups G0..G0+K-1 (blocks
// b, b + 1024, b + 2048, b + 3072 share a SIMD for b in 256..511 in every launch looked at - tools/placement_probe.py: a CU receives
// blocks c, c + 256, c + 512, ... in turn; its very first block (groups 0..255) and its very last (group c + 768, sometimes c + 512)
// trade SIMDs in 3-35 % of the launches, the second block of every round never does -, G0 = 256); the
// other three blocks of those groups run no environment - they sleep until
// "their" slow environment has finished and hold the slots meanwhile - and the 3 K environments that would have sat there run
// in the spare blocks behind the regular grid, which the dispatcher starts as soon as the first light environments finish.
// Every environment that is not in the list is the r-th of them in id order for exactly one block r: a bijection whatever the
// list holds.  Returns the environment of this block, or -1 for a block without one (after the placeholder's wait).
// ------------------------------------------------------------------------------------------------
// hardware placement of this wave: XCC id, shader engine / array, CU and SIMD (HW_ID bits 15:8 and 5:4; wave slot and pipe dropped)
DE_DEV unsigned drv_hw_simd_key() {
  const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg(63492) /* HW_REG_HW_ID, 32 bits */, xcc = (unsigned)__builtin_amdgcn_s_getreg(63508) /* XCC_ID */;
  return ((xcc & 0xFu) << 16) | (hw & 0xFF30u);
}
// tick / pv_par of this launch: kernel arguments the host advances per step - or, once a step of the handle has been captured into
// a hipGraph (tick_src = 1: a replayed launch has frozen arguments), two device words that a one-thread kernel in front of every
// step advances (drv_tick_advance_kernel).  An eager launch mirrors its arguments into those words, so the switch is seamless.
DE_DEV int drv_launch_tick(const DrvState& S) { return S.tick_src ? uniform_i(__atomic_load_n(&S.iso[13], __ATOMIC_RELAXED)) : S.tick; }
DE_DEV int drv_launch_pv(const DrvState& S) { return S.tick_src ? uniform_i(__atomic_load_n(&S.iso[14], __ATOMIC_RELAXED)) : S.pv_par; }
DE_DEV int drv_iso_assign(const DrvState& S, int lane, int tick) {
  const int b = blockIdx.x, E = S.E;
  if (!S.iso_on) return b;
  const int buf = tick % 3;
  if (b == 0 && lane == 0) { const int nn = (tick + 2) % 3; S.iso[nn] = 0; S.iso[3 + nn] = 0; }  // the buffer the NEXT step fills
  if (S.iso_on == 3) return b;  // timing only (Partial observations): see drv_iso_report and the fused passes in drv_step_body
  int K = uniform_i(S.iso[buf]);
  const int cap = S.iso_on == 1 ? DRV_ISO_MAX : DRV_ISO_LIST;
  K = K < cap ? K : cap;
  // mode 1: the regular grid is ALWAYS one residency round of the device (G = 4096 blocks, whatever E <= G is: the block -> SIMD
  // pattern belongs to the grid, not to the environments; blocks beyond the environments end at once), spares and validator behind it
  const int G = S.iso_on == 1 ? 4 * DRV_ISO_GROUPS : E;
  if (S.iso_on == 1) {
    // Self-validation of the placement isolation relies on (blocks g, g + 1024, g + 2048, g + 3072 of the regular grid on one
    // SIMD, for the groups G0 <= g < G0 + GSPAN it may use): every regular block records where it runs; the LAST spare block of the launch (it steps no environment) checks the
    // record of the previous launch and publishes the verdict for the next one.  Wherever the pattern does not hold - another
    // kernel sharing the device, a partitioned device, a different dispatcher - isolation stays off (K = 0): same results,
    // same bijection, no placeholders holding wave slots for nothing.
    if (b < G) { if (lane == 0) S.iso_hw[(size_t)(tick & 1) * (4 * DRV_ISO_GROUPS) + b] = drv_hw_simd_key(); }
    else if (b == G + 3 * DRV_ISO_MAX) {  // (one block behind the spares, for this alone)
      const unsigned* hw = S.iso_hw + (size_t)((tick + 1) & 1) * (4 * DRV_ISO_GROUPS);
      bool bad = false;
      for (int g = DRV_ISO_G0 + lane; g < DRV_ISO_G0 + DRV_ISO_GSPAN; g += 64) {
        const unsigned k0 = hw[g], k1 = hw[g + DRV_ISO_GROUPS], k2 = hw[g + 2 * DRV_ISO_GROUPS], k3 = hw[g + 3 * DRV_ISO_GROUPS];
        bad |= k0 == 0xFFFFFFFFu || k0 != k1 || k0 != k2 || k0 != k3;
      }
      const bool ok = wave_ballot(bad) == 0ull;
      if (lane == 0) {
        // hysteresis: one launch that did not validate keeps isolation off for the next DRV_ISO_COOLDOWN validated ones as well - a
        // device shared with another kernel validates now and then by chance, and placeholders parked on such a launch's verdict
        // hold wave slots the other kernel could use
        int cd = S.iso[12];
        if (!ok) cd = DRV_ISO_COOLDOWN; else if (cd > 0) cd -= 1;
        S.iso[12] = cd;
        S.iso[8 + (tick + 1) % 3] = (ok && cd == 0) ? 1 : 0;
        if (!ok && tick > 1) atomicAdd(&S.iso[11], 1);
      }
      return -1;
    }
    if (uniform_i(S.iso[8 + buf]) != 1) K = 0;
  }
  if (K == 0) return b < E ? b : -1;
  const int* H = S.iso + DRV_ISO_HDR + buf * DRV_ISO_LIST;
  int r;
  if (S.iso_on == 2) {
    // more environments than fit at once: the listed (slow) ones take the first blocks - they start with the launch instead of
    // wherever their id falls in the later residency rounds -, every other environment follows in id order
    if (b < K) return uniform_i(H[b]);
    r = b - K;
  } else if (b >= G) {
    const int d = b - G;
    if (d >= 3 * K) return -1;
    r = (G - 4 * K) + d;
  } else {
    const int g = b & (DRV_ISO_GROUPS - 1), pos = b / DRV_ISO_GROUPS;
    if ((unsigned)(g - DRV_ISO_G0) < (unsigned)K) {
      const int h = uniform_i(H[g - DRV_ISO_G0]);
      if (pos == 0) return h;
      // placeholder: hold this wave slot idle while the slow environment of this SIMD runs (bounded wait, then exit)
      int i = 0;
      for (; i < 96; ++i) {
        if (uniform_i(__atomic_load_n(&S.iso_done[h], __ATOMIC_RELAXED)) == tick) break;
        __builtin_amdgcn_s_sleep(127);
      }
      if (i >= 96 && lane == 0) atomicAdd(&S.iso[7], 1);  // (diagnostic: a placeholder that gave up waiting; dynenv_debug_counters)
      return -1;
    }
    r = pos * (DRV_ISO_GROUPS - K) + (g < DRV_ISO_G0 ? g : g - K);
  }
  // the r-th environment id that is not in the list: least x with x = r + #{h in H : h <= x}
  int h0 = 0x7FFFFFFF, h1 = h0, h2 = h0, h3 = h0;
  if (lane < K) h0 = H[lane];
  if (lane + 64 < K) h1 = H[lane + 64];
  if (lane + 128 < K) h2 = H[lane + 128];
  if (lane + 192 < K) h3 = H[lane + 192];
  int x = r;
  for (int it = 0; it <= DRV_ISO_LIST; ++it) {
    const int c = __popcll(wave_ballot(h0 <= x)) + __popcll(wave_ballot(h1 <= x)) + __popcll(wave_ballot(h2 <= x)) + __popcll(wave_ballot(h3 <= x));
    const int nx = r + c;
    if (nx == x) break;
    x = nx;
  }
  return x < E ? x : -1;
}
// the step's epilogue: this environment's cycles -> the list of the next step (those above 0.7 of the previous step's slowest, once
// that is long enough for isolation to matter), and the tick its placeholders are waiting for
#ifndef DRV_ISO_MIN
#define DRV_ISO_MIN 300000 /* isolation starts when the slowest environment of a step needed this many cycles */
#endif
#ifndef DRV_ISO_PCT
#define DRV_ISO_PCT 75 /* ... for the environments above this many percent of it (round 6, with DRV_ISO_MAX = 64: 70 +5.4 %, 75 +2.4 %, 80 0,
                          85 +2.1 %, 90 +3.8 % - the list overflowed below 80; with DRV_ISO_MAX = 192: 70 -0.25 %, 75 -0.6 %) */
#endif
DE_DEV void drv_iso_report(const DrvState& S, int e, int lane, unsigned long long t0, int tick) {
  if (!S.iso_on || lane != 0) return;
  const int cycles = (int)(__builtin_amdgcn_s_memtime() - t0);
  const int buf = tick % 3, nxt = (tick + 1) % 3;
  const int slowest = S.iso[3 + buf];
  // (only the few environments above the floor touch the shared words: 4096 atomics on one address serialise - 0.12 ms, measured)
  if (cycles > DRV_ISO_MIN && cycles > (slowest / 100) * DRV_ISO_PCT) atomicMax(&S.iso[3 + nxt], cycles);
  if (S.iso_on == 3) return;
  if (slowest > DRV_ISO_MIN && cycles > (slowest / 100) * DRV_ISO_PCT) {
    const int k = atomicAdd(&S.iso[nxt], 1);
    if (k < DRV_ISO_LIST) S.iso[DRV_ISO_HDR + nxt * DRV_ISO_LIST + k] = e;
  }
  __atomic_store_n(&S.iso_done[e], tick, __ATOMIC_RELAXED);
}

template <bool PARTIAL>
DE_DEV void drv_step_body(const DrvState& S, const int* __restrict__ actions, float* __restrict__ obs, double* __restrict__ rewards,
                          uint8_t* __restrict__ dones, float* __restrict__ pobs, int pvNoise, double pvMagn) {
  const unsigned long long isoT0 = __builtin_amdgcn_s_memtime();
DRV_PROF(const unsigned long long KS = isoT0;)
  DrvLds& L = g_L;
  int lane = threadIdx.x;
  const int tick = drv_launch_tick(S);
  if (!S.tick_src && blockIdx.x == 0 && lane == 0) { S.iso[13] = S.tick; S.iso[14] = S.pv_par; }  // (mirror for a later switch to tick_src = 1)
  const int e = drv_iso_assign(S, lane, tick);  // (= blockIdx.x unless the slow environments of the previous step are being isolated)
  if (e < 0) return;
DRV_PROF(if (lane < 8 && e < 4096) { g_dbgp[e * 8 + lane] = 0ull; g_dbgs[e * 8 + lane] = 0ull; })
  const int A = S.A;
  int* envi = S.envi + (size_t)e * EI_COUNT;
  int elapsed = uniform_i(envi[EI_ELAPSED]);
  int allFinished = uniform_i(envi[EI_ALLFIN]);
  const int nPed = uniform_i(envi[EI_NPED]);
  const int nObst = uniform_i(envi[EI_NOBST]);
  const uint32_t episode = (uint32_t)uniform_i(envi[EI_EPISODE]);
  uint64_t occ = (uint64_t)(uint32_t)uniform_i(envi[EI_OCC]);
  // Environments with live contacts are the long ones and the launch ends with the slowest: their waves get issue priority
  // over the (three) lighter waves they share a SIMD with, from the first instruction on.
  if (occ != 0ull) __builtin_amdgcn_s_setprio(3);
  else if (PARTIAL) __builtin_amdgcn_s_setprio(1);  // any physics goes before the neighbours' vision passes (priority 0)
  const uint32_t genv = (uint32_t)(S.env_id_offset + e);

  const bool isCar = lane < A;
  const bool isPed = lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed;
  const bool isBody = isCar || isPed;
  // (the action and candidate-mask loads are issued with load_env's batch, not after it)
  const int g_act0 = actions[((size_t)e * A + (isCar ? lane : 0)) * 2 + 0], g_act1 = actions[((size_t)e * A + (isCar ? lane : 0)) * 2 + 1];
  const int g_lastCand = S.lastcand[(size_t)e * 64 + lane];
  load_env(S, L, e, lane, A, nPed, nObst, occ);

  {
    int act0 = isCar ? g_act0 : 1, act1 = isCar ? g_act1 : 1;
    // action_space is MultiDiscrete([3, 3]) (:170-174); the reference raises on a malformed action (:365-368), here the car
    // coasts (acc = steer = 0) and the environment's error flag (bit 1, dynenv_error_flags) records it
    const bool bad = (unsigned)act0 > 2u || (unsigned)act1 > 2u;
    if (bad) { act0 = 1; act1 = 1; }
    if (lane < 16) L.act[lane] = (unsigned char)(act0 | (act1 << 2));
    const bool anyBad = wave_ballot(bad) != 0ull;
    if (lane == 0) L.stepErr = anyBad ? 2 : 0;
  }
  int finishedAt = -1;  // elapsed time at which the last car finished in this step: the team reward (:252-254) follows from it
  if (lane < 16) { L.rewAcc[lane] = 0.0; L.posAcc[lane] = 0.0; }
  bool aabbValid = false;
  // quiescent-shortcut state carried across launches: candidate mask of the previous substep (-1: unknown) and
  // whether every cached arbiter was inert when the contact path last ran
  L.lastCand[lane] = g_lastCand;
  if (lane == 0) { L.clistN = -1; L.sActive = ~0ull; L.sLvMeta = 0; }
  bool inertAll = (uniform_i(envi[EI_PAD]) & 1) != 0;
  // steady-replay state: the contact path ran (or was replayed) in the previous substep and reported every slot steady
  bool steadyAll = (uniform_i(envi[EI_PAD]) & 2) != 0;
  bool vbValid = (uniform_i(envi[EI_PAD]) & 4) != 0;
  int nFast = 0, nQuiet = 0, nContact = 0, nSlots = 0, nWhyCand = 0, nWhyMoving = 0, nWhyInert = 0, nSteady = 0, nLight = 0, nSplit = 0;  // diagnostics
  __syncthreads();

DRV_PROF(const unsigned long long K0 = __builtin_amdgcn_s_memtime(); unsigned long long tPh1 = 0, tBroad = 0, tFast = 0, tCont = 0, tBook = 0;)
  bool lightOff = false;
  for (int it = 0; it < 10; ++it) {
    lane = fresh_lane();  // per substep: nothing derived from the lane id is hoisted out of the loop and kept alive (= spilled and
                          // reloaded) across the calls of every substep, and the id itself is recomputed after them
DRV_PROF(const unsigned long long A0 = __builtin_amdgcn_s_memtime();)
    // ======== phase 1a: car game logic (processAction at substep 0, tick) ===================================
    const DrvLightRet lr = drv_light_substep(it, lane, A, nPed, nObst, elapsed, (uint32_t)S.seed, (uint32_t)(S.seed >> 32), genv, episode,
                                             (vbValid ? 1 : 0) | (aabbValid ? 2 : 0));
    const int cand = lr.cand, dirty = lr.dirty;
    const bool candMoving = (lr.bits & 1) != 0, removed = (lr.bits & 2) != 0;
    aabbValid = true;
DRV_PROF(const unsigned long long A1 = A0;)
    const uint64_t anyCand = wave_ballot(cand != 0);
    // Quiescent contact set: same candidate pairs as in the previous substep, every body in them exactly at rest and
    // every cached arbiter inert (zero bias, zero accumulated impulse, not first contact).  Then narrowphase, arbiter
    // update, warm start and all 10 solver iterations are exact no-ops (DESIGN.md "quiescent shortcut") and only the
    // velocity update remains.
    const bool candChanged = wave_ballot((lr.bits & 4) != 0) != 0ull, anyMoving = wave_ballot(candMoving) != 0ull;
    if (candChanged && lane == 0) L.clistN = -1;  // the contact path's candidate list belongs to other masks now
    const bool quiescent = inertAll && !candChanged && !anyMoving;
    // Steady replay: the contact path of the previous substep reported every slot steady, the candidate set is the
    // same and every body in it is frozen => this substep's contact path would read the same inputs and reproduce the
    // same outputs: slots unchanged, velocities stay zero, bias velocities equal to the saved ones.
    // If some pairs are dirty, the contact path first tests only those ("light" mode) and falls back to the full path
    // when one of them touches or owns a slot.
    const bool steadyOk = !quiescent && steadyAll && wave_ballot(removed) == 0ull;
    const bool anyDirty = wave_ballot(dirty != 0) != 0ull;
    bool replay = steadyOk && !anyDirty;
    // (a light-mode attempt that fails is paid on top of the full path, and what made it fail - a moving car leaning on a
    // resting pile - persists: after a failure the rest of the step goes straight to the full path.  Both give the same result.)
    const bool light = steadyOk && anyDirty && !lightOff;
    if (!(anyCand == 0ull && occ == 0ull) && !quiescent && !replay) { if (candChanged) nWhyCand++; else if (anyMoving) nWhyMoving++; else nWhyInert++; }

DRV_PROF(const unsigned long long A2 = __builtin_amdgcn_s_memtime(); bool tookContact = false;)
    if (anyCand == 0ull && occ == 0ull) nFast++; else if (quiescent) nQuiet++; else if (replay) nSteady++; else nContact++;
    nSlots += __popcll(occ);
    if ((anyCand == 0ull && occ == 0ull) || quiescent) {
      // ---------- fast path: nothing touches and the contact cache is empty (or quiescent): velocity update only
      velocity_update_ool(A | (nPed << 8));
      if (anyCand == 0ull && occ == 0ull) steadyAll = false;
      vbValid = false;  // nothing was solved: the next position update sees zero bias velocities
    } else if (!replay) {
      // ---------- contact path: narrowphase -> contact cache -> callbacks -> prestep -> friction -> solver
DRV_PROF(tookContact = true;)
      __builtin_amdgcn_s_setprio(3);  // an environment on the contact path is on the launch's critical path: issue it first
      ContactRet cr = drv_contact_path(lane, cand, dirty, light ? 1 : 0, A, A | (nPed << 8), occ);
      lane = fresh_lane();
      if (wave_ballot((cr.err & 1) != 0) != 0ull && lane == 0) L.stepErr |= 1;
      if (light && !(uniform_i(cr.err >> 3) & 1)) lightOff = true;
      if (uniform_i(cr.err >> 3) & 1) {  // light mode: no dirty pair touches => replay
        replay = true; nLight++;
      } else {
        nSplit += uniform_i(cr.err >> 4) & 1;
        occ = uniform_u64(cr.occ);
        inertAll = (uniform_i(cr.err >> 1) & 1) != 0;
        steadyAll = (uniform_i(cr.err >> 2) & 1) != 0;
        vbValid = true;
      }
    }
    if (replay) {  // L.vb* still hold the bias velocities of the solve being replayed
      velocity_update_ool(A | (nPed << 8));
      vbValid = true;
    }
    __syncthreads();

DRV_PROF(const unsigned long long A3 = __builtin_amdgcn_s_memtime(); tPh1 += A1 - A0; tBroad += A2 - A1; if (tookContact) tCont += A3 - A2; else tFast += A3 - A2;)
    // ======== bookkeeping :280-287 =========================================================================
    elapsed += 1;
    bool notDone = false;
    if (isCar) { int f = L.flags[lane]; notDone = !(CF_FIN(f) && !CF_CRASHED(f)); }
    const bool allFin = wave_ballot(notDone) == 0ull;
    if (!allFinished && allFin) {
      allFinished = 1;
      finishedAt = elapsed;
    }
  }
  lane = fresh_lane();
  const double teamReward = finishedAt >= 0 ? 0.0 + (double)(DRV_MAX_TIME - finishedAt) / 100.0 : 0.0;

DRV_PROF(const unsigned long long K1 = __builtin_amdgcn_s_memtime();)
  // ---------------- end of env step :300-322 ----------------------------------------------------------------
  // The read-modify-writes of the epilogue (episode sums, diagnostic counters) are split: their loads are issued here, the
  // state and the observation are written meanwhile, the sums are stored last - one memory round trip hidden behind the
  // observation instead of two in front of it.
  double* er = S.epr + (size_t)e * 16 + (isCar ? lane : 0);
  double* ep = S.epr + (size_t)S.E * 16 + (size_t)e * 16 + (isCar ? lane : 0);
  const double g_er = *er, g_ep = *ep;
  const int cl = lane < 10 ? lane : 0;  // lanes 0..9: one diagnostic counter each (EI_N_FAST .. EI_N_SPLIT are consecutive)
  const int g_cnt = envi[EI_N_FAST + cl];
  const int g_err = envi[EI_ERR];
  double rew = 0.0, posrew = 0.0;
  if (isCar) {
    rew = L.rewAcc[lane] + teamReward;
    posrew = L.posAcc[lane] + dm_max(0.0, teamReward);
    rewards[(size_t)e * A + lane] = rew;
  }
  if (lane == 0) {
    dones[e] = (uint8_t)(elapsed >= DRV_MAX_TIME);
    envi[EI_ELAPSED] = elapsed; envi[EI_ALLFIN] = allFinished; envi[EI_OCC] = (int)(uint32_t)occ;
    envi[EI_PAD] = (inertAll ? 1 : 0) | (steadyAll ? 2 : 0) | (vbValid ? 4 : 0);
  }
  const int errBits = uniform_i(L.stepErr);
  S.lastcand[(size_t)e * 64 + lane] = L.lastCand[lane];
  store_env(S, L, e, lane, A, nPed, occ);
  if (obs) write_full_obs_ool(lane, A, nPed, nObst, S.obs_dim, obs + (size_t)e * A * S.obs_dim);
  if (isCar) { *er = g_er + rew; *ep = g_ep + posrew; }
  {
    static_assert(EI_N_QUIET == EI_N_FAST + 1 && EI_N_CONTACT == EI_N_FAST + 2 && EI_N_SLOTS == EI_N_FAST + 3 && EI_N_WHY_CAND == EI_N_FAST + 4 &&
                  EI_N_WHY_MOVING == EI_N_FAST + 5 && EI_N_WHY_INERT == EI_N_FAST + 6 && EI_N_STEADY == EI_N_FAST + 7 && EI_N_LIGHT == EI_N_FAST + 8 && EI_N_SPLIT == EI_N_FAST + 9,
                  "the diagnostic counters are consecutive");
    const int add = lane == 0 ? nFast : lane == 1 ? nQuiet : lane == 2 ? nContact : lane == 3 ? nSlots : lane == 4 ? nWhyCand : lane == 5 ? nWhyMoving :
                    lane == 6 ? nWhyInert : lane == 7 ? nSteady : lane == 8 ? nLight : nSplit;
    if (lane < 10) envi[EI_N_FAST + lane] = g_cnt + add;
  }
  if (errBits && lane == 0) envi[EI_ERR] = g_err | errBits;
  // Partial observation of this environment, fused (see drv_partial_obs_fused): the first `fusedAgents` agent passes run
  // here, the rest is left to the deferred launch.  Without a forecast (DYNENV_NO_ISOLATION, or no environment was slow in the
  // previous step): an environment that spent the step on the contact path defers everything, a light one nothing.
  int fusedAgents = !(PARTIAL && pobs) ? 0 : nContact >= DRV_DEFER_MIN_CONTACT ? 0 : (A < DRV_FUSED_AGENTS ? A : DRV_FUSED_AGENTS);
  // With a forecast of when the launch will end (the previous step's slowest environment, drv_iso_report keeps it) every
  // environment, light or not, simply runs its passes until then and leaves the rest: the SIMDs whose four waves are all light are
  // the ones with the most vision to do (an environment on the contact path is done with its physics later and gets to fewer of
  // its passes), and they - not the slowest environment - were what the launch waited for.
  int budget = 0;
  if (PARTIAL && pobs && S.iso_on >= 2) {
    const int slowest = uniform_i(S.iso[3 + tick % 3]);
    if (slowest > 0) {
      budget = (slowest / 100) * DRV_PV_DEADLINE_PCT - (int)(__builtin_amdgcn_s_memtime() - isoT0);
      fusedAgents = budget > 0 ? A : 0;
    }
  }
  drv_iso_report(S, e, fresh_lane(), isoT0, tick);
  if (PARTIAL && fusedAgents > 0) {
    __builtin_amdgcn_s_setprio(0);  // (an environment that touched the contact path raised it: the vision passes are nobody's critical path)
    PvIn in;
    in.px = in.py = in.ang = in.ox = in.oy = in.gx = in.gy = 0.0; in.flags = 0;
    if (lane < DRV_NB && (lane < A || (lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed))) {
      in.px = L.px[lane]; in.py = L.py[lane]; in.ang = L.ang[lane]; in.flags = L.flags[lane];
    }
    if (lane < nObst) { in.ox = L.ox[lane]; in.oy = L.oy[lane]; }
    if (lane < A) { in.gx = L.goalx[lane]; in.gy = L.goaly[lane]; }
    fusedAgents = drv_partial_obs_fused(in, S.seed, S.A, S.envi, S.env_id_offset, e, nPed | (nObst << 8), elapsed, episode, pvNoise | (fusedAgents << 8), pvMagn,
                                        pobs, budget);
  }
  if (PARTIAL && fresh_lane() == 0) {
    envi[EI_DEFER_OBS] = fusedAgents;  // first agent the deferred launch has to do
    if (pobs && fusedAgents < A) {  // ... and this environment on its list
      const int pv = drv_launch_pv(S);
      const int k = atomicAdd(&S.pvq[16 * pv], 1);
      if (k < S.E) S.pvq[32 + pv * S.E + k] = e;
      else envi[EI_ERR] = envi[EI_ERR] | 4;  // only a host that replays a captured launch (frozen pv_par: the length is never cleared) gets here
    }
  }
DRV_PROF(if (lane == 0 && e < 4096) { unsigned long long* d = g_dbgw + e * 12; const unsigned long long KE = __builtin_amdgcn_s_memtime(); d[0] = KE - KS; d[1] = nContact; d[2] = __popcll(occ); d[3] = nSteady + nQuiet; d[4] = K0 - KS; d[5] = tPh1; d[6] = tBroad; d[7] = tFast; d[8] = tCont; d[9] = (K1 - K0) - tPh1 - tBroad - tFast - tCont; d[10] = KE - K1; d[11] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492); })
}

// tick_src = 1 (a step of this handle was captured into a hipGraph): what the host does between two eager launches, on the device
// (flipPv: the deferred-vision parity alternates only over steps that run the Partial + deferred pair - a list's length word is cleared by
//  the deferred launch of the OTHER parity, so a step without an observation buffer must leave the parity alone, as the eager host does)
extern "C" __global__ void drv_tick_advance_kernel(DrvState S, int flipPv) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { S.iso[13] = (S.iso[13] + 1) % (3 * (1 << 28)); if (flipPv) S.iso[14] = S.iso[14] ^ 1; }
}

extern "C" __global__ void __launch_bounds__(64, DRV_WAVES_PER_SIMD)
drv_step_kernel(DrvState S, const int* __restrict__ actions, float* __restrict__ obs, double* __restrict__ rewards,
                uint8_t* __restrict__ dones) {
  drv_step_body<false>(S, actions, obs, rewards, dones, nullptr, 0, 0.0);
}
// Partial observation: same step, then each wave writes its environment's observation (fused getAgentVision)
extern "C" __global__ void __launch_bounds__(64, DRV_WAVES_PER_SIMD)
drv_step_partial_kernel(DrvState S, const int* __restrict__ actions, double* __restrict__ rewards, uint8_t* __restrict__ dones,
                        float* __restrict__ pobs, int pvNoise, double pvMagn) {
  drv_step_body<true>(S, actions, nullptr, rewards, dones, pobs, pvNoise, pvMagn);
}

// ------------------------------------------------------------------------------------------------
// observation-only kernel (used after reset / set_state)
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64) drv_obs_kernel(DrvState S, float* __restrict__ obs) {
  DrvLds& L = g_L;
  const int e = blockIdx.x, lane = threadIdx.x, A = S.A;
  const int* envi = S.envi + (size_t)e * EI_COUNT;
  const int nPed = uniform_i(envi[EI_NPED]), nObst = uniform_i(envi[EI_NOBST]);
  load_env(S, L, e, lane, A, nPed, nObst, 0ull);
  write_full_obs_ool(lane, A, nPed, nObst, S.obs_dim, obs + (size_t)e * A * S.obs_dim);
}

// ------------------------------------------------------------------------------------------------
// reset kernel: scene re-randomisation, one thread per environment
// (environment_base.py:205-211 -> DrivingEnvironment._setup_scene :58-115, :527-584).  Not on the per-step path.
// ------------------------------------------------------------------------------------------------
DE_DEV void road_get_spot(const DrvRoad& r, int lane, int spot, V2& pos, double& angle) {  // Road.py:100-114
  int end = lane >= r.nLanes ? 1 : 0;
  V2 p = end ? r.p1 : r.p0;
  V2 spotDir = vmul(end ? vneg(r.dir) : r.dir, r.followDist);
  V2 laneDir = vmul(end ? r.normal : vneg(r.normal), r.width);
  double l = (double)(end ? lane - r.nLanes : lane) + 0.5;
  pos = vadd(vadd(p, vmul(laneDir, l)), vmul(spotDir, (double)spot));
  angle = dm_atan2(spotDir.y, spotDir.x);
}
DE_DEV V2 road_get_walk_spot(const DrvRoad& r, int side, double length, double width) {  // Road.py:117-123
  V2 w0 = r.walk[side][0], w1 = r.walk[side][1];
  V2 center = vadd(w0, vmul(vsub(w1, w0), length));
  double f = width * r.width;
  V2 off = vmul(vmul(r.normal, f), side ? 1.0 : -1.0);
  return vadd(center, off);
}

extern "C" __global__ void __launch_bounds__(64) drv_reset_kernel(DrvState S) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.E) return;
  const size_t E = (size_t)S.E;
  const int A = S.A;
  int* envi = S.envi + (size_t)e * EI_COUNT;
  const uint32_t ep = (uint32_t)envi[EI_EPISODE];
  const uint32_t genv = (uint32_t)(S.env_id_offset + e);
  // zero this env's rows
  for (int k = 0; k < DRV_NB; ++k) {
    for (int f = 0; f < BF_COUNT; ++f) S.body[(size_t)f * E * DRV_NB + (size_t)e * DRV_NB + k] = 0.0;
    S.flags[(size_t)e * DRV_NB + k] = 0;
    S.aux[(size_t)e * DRV_NB + k] = 0;
  }
  for (int k = 0; k < 16; ++k) {
    for (int f = 0; f < CF_COUNT; ++f) S.carx[(size_t)f * E * 16 + (size_t)e * 16 + k] = 0.0;
    S.epr[(size_t)e * 16 + k] = 0.0;
    S.epr[E * 16 + (size_t)e * 16 + k] = 0.0;
  }
  for (int k = 0; k < DRV_NS; ++k) { S.s_pair[(size_t)e * DRV_NS + k] = 0xFFFF; S.s_meta[(size_t)e * DRV_NS + k] = 0; }
  // cars: spots = permutation(30)[:A] (partial Fisher-Yates), road/end/team/type draws
  int spots[30];
  for (int i = 0; i < 30; ++i) spots[i] = i;
  for (int i = 0; i < A; ++i) {
    dm_u32x4 u = dm_env_rng(S.seed, genv, ep, DM_RNG_RESET_PERM, (uint32_t)i, 0);
    int j = i + dm_randint(u.v[0], 0, 29 - i);
    int t = spots[i]; spots[i] = spots[j]; spots[j] = t;
  }
  for (int i = 0; i < A; ++i) {
    dm_u32x4 u = dm_env_rng(S.seed, genv, ep, DM_RNG_RESET_AGENT, (uint32_t)i, 0);
    int roadSel = dm_randint(u.v[0], 0, 1), endSel = dm_randint(u.v[1], 0, 1);
    int team = dm_randint(u.v[2], 0, 2), type = dm_randint(u.v[3], 0, 3);
    V2 goal = endSel ? C.roads[roadSel].p1 : C.roads[roadSel].p0;
    int spotID = spots[i];
    int roadID = spotID < 20 ? 0 : 1;
    spotID -= roadID ? 20 : 0;
    int laneID = spotID / 5, spot = spotID % 5;
    V2 pos; double angle;
    road_get_spot(C.roads[roadID], laneID, spot, pos, angle);
    V2 dir = vrot_angle(v2(1.0, 0.0), angle);
    size_t b = (size_t)e * DRV_NB + i;
    S.body[BF_PX * E * DRV_NB + b] = pos.x; S.body[BF_PY * E * DRV_NB + b] = pos.y; S.body[BF_ANG * E * DRV_NB + b] = angle;
    size_t c = (size_t)e * 16 + i;
    S.carx[CF_DIRX * E * 16 + c] = dir.x; S.carx[CF_DIRY * E * 16 + c] = dir.y;
    S.carx[CF_PREVX * E * 16 + c] = pos.x; S.carx[CF_PREVY * E * 16 + c] = pos.y;
    S.carx[CF_GOALX * E * 16 + c] = goal.x; S.carx[CF_GOALY * E * 16 + c] = goal.y;
    S.flags[b] = CARF_PACK(type, team, 0, 0, 0, LP_OffRoad);
  }
  dm_u32x4 uc = dm_env_rng(S.seed, genv, ep, DM_RNG_RESET_COUNTS, 0, 0);
  int nPed = dm_randint(uc.v[0], 10, 20), nObstRaw = dm_randint(uc.v[1], 10, 20);
  for (int i = 0; i < nPed; ++i) {
    dm_u32x4 a = dm_env_rng(S.seed, genv, ep, DM_RNG_RESET_PED, (uint32_t)i, 0);
    dm_u32x4 bq = dm_env_rng(S.seed, genv, ep, DM_RNG_RESET_PED, (uint32_t)i, 1);
    int road = dm_randint(a.v[0], 0, 1), side = dm_randint(a.v[1], 0, 1);
    double len = dm_unit(a.v[2]), wid = dm_unit(a.v[3]) / 2.0 + 0.25;
    V2 p = road_get_walk_spot(C.roads[road], side, len, wid);
    size_t b = (size_t)e * DRV_NB + DRV_SLOT_PED + i;
    S.body[BF_PX * E * DRV_NB + b] = p.x; S.body[BF_PY * E * DRV_NB + b] = p.y;
    S.flags[b] = PEDF_PACK(road, side, 0, 0, 0, dm_randint(bq.v[0], 3, 6));
  }
  int nObst = 0;
  for (int i = 0; i < nObstRaw; ++i) {
    dm_u32x4 a = dm_env_rng(S.seed, genv, ep, DM_RNG_RESET_OBST, (uint32_t)i, 0);
    int road = dm_randint(a.v[0], 0, 1), side = dm_randint(a.v[1], 0, 1);
    double len = dm_unit(a.v[2]), wid = dm_unit(a.v[3]) / 2.0 + 0.25;
    V2 c = road_get_walk_spot(C.roads[road], side, len, wid);
    if (drv_is_off_road(c)) {
      S.obst[(size_t)e * DRV_MAXO + nObst] = c.x;
      S.obst[E * DRV_MAXO + (size_t)e * DRV_MAXO + nObst] = c.y;
      nObst++;
    }
  }
  envi[EI_ELAPSED] = 0; envi[EI_ALLFIN] = 0; envi[EI_NPED] = nPed; envi[EI_NOBST] = nObst;
  envi[EI_EPISODE] = (int)(ep + 1); envi[EI_OCC] = 0; envi[EI_ERR] = 0;
  envi[EI_N_FAST] = 0; envi[EI_N_QUIET] = 0; envi[EI_N_CONTACT] = 0; envi[EI_N_SLOTS] = 0; envi[EI_PAD] = 0;
  envi[EI_N_STEADY] = 0; envi[EI_N_LIGHT] = 0; envi[EI_N_SPLIT] = 0;
  envi[EI_N_WHY_CAND] = 0; envi[EI_N_WHY_MOVING] = 0; envi[EI_N_WHY_INERT] = 0;
  for (int k = 0; k < 64; ++k) S.lastcand[(size_t)e * 64 + k] = -1;
}

// episode_g = [#finished & !crashed, #crashed] (:315-316) + episode accumulators, gathered for the host mirror
extern "C" __global__ void drv_stats_kernel(DrvState S, double* ep_r, double* ep_pos_r, double* ep_obs_r, int* goals) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.E) return;
  int fin = 0, crashed = 0;
  for (int a = 0; a < S.A; ++a) {
    int f = S.flags[(size_t)e * DRV_NB + a];
    int fi = (f >> 4) & 1, cr = (f >> 5) & 1;
    fin += fi && !cr;
    crashed += cr;
    if (ep_r) ep_r[(size_t)e * S.A + a] = S.epr[(size_t)e * 16 + a];
    if (ep_pos_r) ep_pos_r[(size_t)e * S.A + a] = S.epr[(size_t)S.E * 16 + (size_t)e * 16 + a];
    if (ep_obs_r) ep_obs_r[(size_t)e * S.A + a] = 0.0;
  }
  if (goals) { goals[2 * e] = fin; goals[2 * e + 1] = crashed; }
}

extern "C" __global__ void drv_counts_kernel(DrvState S, int* counts) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.E) return;
  counts[2 * e] = S.envi[(size_t)e * EI_COUNT + EI_NOBST];
  counts[2 * e + 1] = S.envi[(size_t)e * EI_COUNT + EI_NPED];
}

// deterministic-math self test (bit-compare against the host evaluation in tests/)
extern "C" __global__ void math_selftest_kernel(const double* x, const double* y, int n, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  dm_sincos(x[i], &s, &c);
  out[5 * i + 0] = s; out[5 * i + 1] = c; out[5 * i + 2] = dm_atan2(y[i], x[i]);
  out[5 * i + 3] = dm_sqrt(dm_abs(x[i])); out[5 * i + 4] = (y[i] != 0.0) ? x[i] / y[i] : 0.0;
}
