// dynenv_capi.hip — the C ABI of include/dynenv.h on top of the gfx950 kernels (this unit: RoboCup + arranger kernels + all host code;
// driving_tu.hip: the Driving kernels).
// Host code only does allocation, constant upload and launches.  There is NO CPU fallback: without a usable HIP
// device every entry point fails with DYNENV_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "driving_host.h"   /* the Driving kernels are a translation unit of their own: driving_tu.hip */
#include "robocup_kernels.hip"
#include "arranger_kernels.hip"
#include "dynenv.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
// Every entry point runs on the handle's device and leaves the calling thread's current device as it found it (a process may
// hold handles on several devices next to torch's own current device).
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (prev >= 0) { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev); } }
};
#define ON_DEVICE(h) DeviceGuard guard_((h)->cfg.device_id); if (!guard_.ok) return fail(DYNENV_ERR_HIP, "hipSetDevice failed")
#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return fail(DYNENV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
  } while (0)

struct dynenv {
  dynenv_cfg_t cfg;
  int A, obs_dim, T, action_dim;
  DrvState S;
  RcState R;
  bool robocup;
  bool partial;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;  // checkpoint = these arrays, in allocation order
  std::vector<void*> scratch;       // scheduling scratch (SIMD-isolation lists): NOT simulation state, never checkpointed
  hipEvent_t ev_begin = nullptr, ev_main = nullptr, ev_end = nullptr;  // dynenv_set_step_events (caller-owned)
  // SIMD isolation, host side: iso_cfg = the mode the handle was created with (S.iso_on may be 0 while isolation is paused);
  // iso_seen = pinned word the device's count of launches whose placement did not validate is copied into now and then
  int iso_cfg = 0, iso_last_invalid = 0, iso_pauses = 0;
  long long steps = 0, iso_paused_until = 0;
  int* iso_seen = nullptr;
};
#define ISO_PROBE_EVERY 64     /* steps between two looks at the device's validation counter (an asynchronous 4-byte copy) */
#define ISO_PAUSE_STEPS 2048   /* steps without isolation after a probe window in which more than half of the launches did not validate */

// ---------------------------------------------------------------------------------------------- constants
static void road_init_host(DrvRoad& r, int nLanes, double width, V2 p0, V2 p1) {  // Road.py:11-33
  V2 d = vsub(p1, p0);
  r.nLanes = nLanes; r.width = width; r.p0 = p0; r.p1 = p1; r.followDist = 90.0;
  r.length = vlen(d);
  r.dir = v2(d.x / r.length, d.y / r.length);
  r.normal = vrot_angle(r.dir, DM_PI / 2.0);
  r.dirAngle = dm_atan2(r.dir.y, r.dir.x);
  r.cosDir0 = dm_cos(r.dirAngle - 0.0);
  double k = (double)(nLanes + 1) * width;
  r.walk[0][0] = vadd(p0, vmul(r.normal, k)); r.walk[0][1] = vadd(p1, vmul(r.normal, k));
  r.walk[1][0] = vsub(p0, vmul(r.normal, k)); r.walk[1][1] = vsub(p1, vmul(r.normal, k));
}

static double moment_for_box(double m, double hx, double hy) {  // cpMomentForPoly on Car.points (Car.py:21-23)
  V2 verts[4] = {v2(hx, hy), v2(-hx, hy), v2(-hx, -hy), v2(hx, -hy)};
  double sum1 = 0.0, sum2 = 0.0;
  for (int i = 0; i < 4; ++i) {
    V2 a1 = verts[i], a2 = verts[(i + 1) % 4];
    double a = vcross(a2, a1);
    double b = vdot(a1, a1) + vdot(a1, a2) + vdot(a2, a2);
    sum1 += a * b; sum2 += a;
  }
  return (m * sum1) / (6.0 * sum2);
}

static double norm_obs_host(double pt, double nf, double mean) { return ((pt * nf) - mean) * 2.0 * 1.0; }

static void build_consts(DrvConst& c) {
  memset(&c, 0, sizeof(c));
  road_init_host(c.roads[0], 2, 35.0, v2(875.0, 0.0), v2(875.0, 1000.0));  // DrivingEnvironment.py:110-115
  road_init_host(c.roads[1], 1, 35.0, v2(0.0, 500.0), v2(1750.0, 500.0));
  // lane rows of getFullState (:689-695) incl. the negative-index quirk `Lanes[i - nLanes]`
  int row = 0;
  for (int r = 0; r < 2; ++r) {
    const DrvRoad& l = c.roads[r];
    int n = l.nLanes, cnt = 2 * n + 1;
    for (int i = -n; i <= n; ++i) {
      int idx = ((i - n) % cnt + cnt) % cnt;
      double s = (double)(idx - n) * l.width;
      V2 a = vadd(l.p0, vmul(l.normal, s)), b = vadd(l.p1, vmul(l.normal, s));
      c.laneRows[row * 5 + 0] = (float)norm_obs_host(a.x, 0.5 / (DRV_W + 100.0), 0.0);
      c.laneRows[row * 5 + 1] = (float)norm_obs_host(a.y, 0.5 / (DRV_H + 100.0), 0.0);
      c.laneRows[row * 5 + 2] = (float)norm_obs_host(b.x, 0.5 / (DRV_W + 100.0), 0.0);
      c.laneRows[row * 5 + 3] = (float)norm_obs_host(b.y, 0.5 / (DRV_H + 100.0), 0.0);
      c.laneRows[row * 5 + 4] = (float)((i == n || i == -n) ? 1 : (i == 0 ? -1 : 0));
      ++row;
    }
  }
  const double masses[4] = {1200, 1800, 3500, 5000}, widths[4] = {5, 6, 7, 8}, lengths[4] = {10, 15, 20, 25},
               powers[4] = {3, 4, 3, 4};  // Car.py:9-12
  for (int t = 0; t < 4; ++t) {
    c.carMass[t] = masses[t]; c.carHx[t] = lengths[t]; c.carHy[t] = widths[t]; c.carPower[t] = powers[t];
    c.carInertia[t] = moment_for_box(masses[t], lengths[t], widths[t]);
  }
  c.pedMass = 90.0;  // Pedestrian.py:11-14
  c.pedInertia = 90.0 * (0.5 * (0.0 * 0.0 + 5.0 * 5.0) + 0.0);
}

// ---------------------------------------------------------------------------------------------- helpers
template <typename T>
static int dev_alloc(dynenv* h, T** out, size_t count) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, count * sizeof(T)));
  HIP_OK(hipMemset(p, 0, count * sizeof(T)));
  h->allocs.push_back(p);
  h->alloc_bytes.push_back(count * sizeof(T));
  *out = (T*)p;
  return 0;
}

// device memory that is not part of the simulation state (never saved / restored by dynenv_checkpoint_*)
template <typename T>
static int dev_alloc_scratch(dynenv* h, T** out, size_t count) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, count * sizeof(T)));
  HIP_OK(hipMemset(p, 0, count * sizeof(T)));
  h->scratch.push_back(p);
  *out = (T*)p;
  return 0;
}
// the SIMD-isolation scheduler starts from scratch: empty lists, placement "not validated yet", no environment "done"
static int iso_reset(dynenv* h) {
  DrvState& S = h->S;
  S.tick = 0;
  HIP_OK(hipMemset(S.iso, 0, sizeof(int) * DRV_ISO_WORDS));
  HIP_OK(hipMemset(S.iso_done, 0xFF, sizeof(int) * (size_t)S.E));
  HIP_OK(hipMemset(S.iso_hw, 0xFF, sizeof(unsigned) * 2 * 4 * DRV_ISO_GROUPS));
  HIP_OK(hipMemset(S.pvq, 0, sizeof(int) * 32));  // both deferred-observation lists empty
  S.pv_par = 0;
  return 0;
}

static int iso_reset_async(dynenv* h, hipStream_t st) {  // the same, stream-ordered (resuming isolation between two steps)
  DrvState& S = h->S;
  S.tick = 0;
  HIP_OK(hipMemsetAsync(S.iso, 0, sizeof(int) * DRV_ISO_WORDS, st));
  HIP_OK(hipMemsetAsync(S.iso_done, 0xFF, sizeof(int) * (size_t)S.E, st));
  HIP_OK(hipMemsetAsync(S.iso_hw, 0xFF, sizeof(unsigned) * 2 * 4 * DRV_ISO_GROUPS, st));
  return 0;
}

// ---------------------------------------------------------------------------------------------- RoboCup host side
static double moment_for_segment_host(double m, V2 a, V2 b, double r) {  // cpMomentForSegment
  V2 offset = vlerp(a, b, 0.5);
  V2 d = vsub(b, a);
  double length = dm_sqrt(vdot(d, d)) + 2.0 * r;
  return m * ((length * length + 4.0 * r * r) / 12.0 + vlensq(offset));
}

static int rc_create(dynenv* h) {
  const dynenv_cfg_t& cfg = h->cfg;
  RcState& R = h->R;
  memset(&R, 0, sizeof(R));
  const size_t E = (size_t)cfg.num_envs;
  R.E = (int)E; R.n = cfg.n_players > 5 ? 5 : cfg.n_players; R.R = 2 * R.n;  // environment_base.py:57, maxPlayers = 5
  R.obs_type = cfg.obs_type; R.noise_type = cfg.noise_type; R.noise_magn = cfg.noise_magnitude;
  R.obs_dim = cfg.obs_type == DYNENV_OBS_PARTIAL ? RCP_DIM : 4 + 8 + (R.R - 1) * 6;
  R.seed = cfg.seed; R.env_id_offset = cfg.env_id_offset; R.flags = cfg.flags;
  h->A = R.R; h->obs_dim = R.obs_dim; h->T = 5; h->action_dim = 4;
  int rc = 0;
  rc |= dev_alloc(h, &R.body, (size_t)(RB_COUNT + 4) * E * RC_NB);
  rc |= dev_alloc(h, &R.rob, (size_t)RR_COUNT * E * 16);
  rc |= dev_alloc(h, &R.robi, (size_t)RI_COUNT * E * 16);
  rc |= dev_alloc(h, &R.envi, E * RE_COUNT);
  rc |= dev_alloc(h, &R.envd, E * RD_COUNT);
  rc |= dev_alloc(h, &R.epr, 2 * E * 16);
  rc |= dev_alloc(h, &R.epo, E * 16);
  rc |= dev_alloc(h, &R.snap, E * 5);
  rc |= dev_alloc(h, &R.prew0, E * 16);
  // (per-step scratch of the Partial observation - who deferred what, the seen counts in transit, the scheduling forecast: rebuilt
  //  by every step, never part of a checkpoint, whose bytes stay a function of the simulation state alone)
  rc |= dev_alloc_scratch(h, &R.seenPart, (size_t)E * 5 * 10 * RCP_SEEN_STRIDE);
  rc |= dev_alloc_scratch(h, &R.deferList, (size_t)E + 1 + 8);
  rc |= dev_alloc(h, &R.s_pair, E * RC_NS);
  rc |= dev_alloc(h, &R.s_meta, E * RC_NS);
  rc |= dev_alloc(h, &R.s_hash, 2 * E * RC_NS);
  rc |= dev_alloc(h, &R.s_imp, 4 * E * RC_NS);
  uint64_t* pairTab = nullptr;
  rc |= dev_alloc(h, &pairTab, 64 * 2);
  if (rc) return DYNENV_ERR_HIP;
  R.pairTab = pairTab;
  RcConst c;
  memset(&c, 0, sizeof(c));
  c.footInertia = moment_for_segment_host(4000.0, v2(-10.0, 10.0), v2(10.0, 10.0), 7.5);  // Robot.py:34
  c.ballInertia = 10.0 * (0.5 * (0.0 * 0.0 + 10.0 * 10.0) + 0.0);                          // Ball.py:9
  {  // cpPivotJoint preStep with r1 = r2 = 0 (k_tensor + inverse), cpRotaryLimitJoint iSum: same operations, same order
    const double ma = 1.0 / ROBOT_MASS, mb = 1.0 / ROBOT_MASS, ia = 1.0 / c.footInertia, ib = 1.0 / c.footInertia;
    const double pr1x = 0.0, pr1y = 0.0, pr2x = 0.0, pr2y = 0.0;
    const double m_sum = ma + mb;
    double k11 = m_sum, k12 = 0.0, k21 = 0.0, k22 = m_sum;
    { const double r1xsq = pr1x * pr1x * ia, r1ysq = pr1y * pr1y * ia, r1nxy = -pr1x * pr1y * ia; k11 += r1ysq; k12 += r1nxy; k21 += r1nxy; k22 += r1xsq; }
    { const double r2xsq = pr2x * pr2x * ib, r2ysq = pr2y * pr2y * ib, r2nxy = -pr2x * pr2y * ib; k11 += r2ysq; k12 += r2nxy; k21 += r2nxy; k22 += r2xsq; }
    const double det = k11 * k22 - k12 * k21;
    const double det_inv = 1.0 / det;
    c.jkk0 = k22 * det_inv; c.jkk1 = -k12 * det_inv; c.jkk2 = -k21 * det_inv; c.jkk3 = k11 * det_inv;
    c.jiSum = 1.0 / (ia + ib);
    c.footMinv = ma; c.footIinv = ia; c.ballIinv = 1.0 / c.ballInertia;
  }
  {  // the Partial-observation scene by vision lane (same expressions as oracle/robocup_partial.c rcp_scene)
    const double W = RC_W, H = RC_H, s = RC_SIDE, pl = 60.0, pw = 110.0, cr = 75.0, pd = 130.0, gw = 80.0;
    int i = 33;
#define LN(ax, ay, bx, by, tx, ty) do { c.visPx[i] = ax; c.visPy[i] = ay; c.visQx[i] = bx; c.visQy[i] = by; c.visT0[i] = tx; c.visT1[i] = ty; ++i; } while (0)
    LN(s, s, s, H - s, 1, 0); LN(W - s, s, W - s, H - s, -1, 0); LN(s, s, W - s, s, 0, 1); LN(s, H - s, W - s, H - s, 0, -1);
    LN(W / 2, s, W / 2, H - s, 0, 0);
    LN(s, H / 2 - pw, s + pl, H / 2 - pw, 1, 0.37); LN(s, H / 2 + pw, s + pl, H / 2 + pw, 1, -0.37);
    LN(s + pl, H / 2 - pw, s + pl, H / 2 + pw, 0.87, 0);
    LN(W - s - pl, H / 2 - pw, W - s, H / 2 - pw, -1, 0.37); LN(W - s - pl, H / 2 + pw, W - s, H / 2 + pw, -1, -0.37);
    LN(W - s - pl, H / 2 - pw, W - s - pl, H / 2 + pw, -0.87, 0);
#undef LN
#define PT(k, x, y, tx, ty) do { c.visPx[k] = x; c.visPy[k] = y; c.visT0[k] = tx; c.visT1[k] = ty; } while (0)
    PT(10, s, H / 2 + gw, 1, -0.27); PT(11, s, H / 2 - gw, 1, 0.27); PT(12, W - s, H / 2 + gw, -1, -0.27); PT(13, W - s, H / 2 - gw, -1, 0.27);
    PT(14, 520.0, 370.0, 0, 0); PT(15, s + pd, 370.0, 1, 0); PT(16, W - (s + pd), 370.0, -1, 0);
    i = 17;
#define FC(x, y, tx, ty) do { PT(i, x, y, tx, ty); ++i; } while (0)
    FC(s, s, 1, 1); FC(s, H - s, 1, -1); FC(W - s, s, -1, 1); FC(W - s, H - s, -1, -1);
    FC(W / 2, s, 0, 1); FC(W / 2, H - s, 0, -1);
    FC(W / 2, H / 2 - cr * 2, 0, 0.5); FC(W / 2, H / 2 + cr * 2, 0, -0.5);
    FC(s, H / 2 - pw, 1, 0.37); FC(s, H / 2 + pw, 1, -0.37); FC(s + pl, H / 2 - pw, 0.87, 0.37); FC(s + pl, H / 2 + pw, 0.87, -0.37);
    FC(W - s, H / 2 - pw, -1, 0.37); FC(W - s, H / 2 + pw, -1, -0.37); FC(W - s - pl, H / 2 - pw, -0.87, 0.37); FC(W - s - pl, H / 2 + pw, -0.87, -0.37);
#undef FC
#undef PT
  }
  int p = 0;
  for (int i = 0; i <= RC_BALL; ++i)
    for (int j = i + 1; j < RC_POST + 4; ++j) c.pairs[p++] = (uint16_t)((i << 8) | j);
  for (; p < RC_NPAIR_ROUNDS * 64; ++p) c.pairs[p] = 0xFFFF;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(RC), &c, sizeof(c));
  if (e != hipSuccess) return fail(DYNENV_ERR_HIP, hipGetErrorString(e));
  {  // the pairs of each lane (lane l tests pairs l, 64 + l, ...), packed, without those of feet this handle's robots do not have
    static_assert(RC_NPAIR_ROUNDS == 5, "pairTab packs four rounds into the first word and the fifth into the second");
    uint64_t tab[64][2];
    for (int lane = 0; lane < 64; ++lane) {
      uint64_t lo = 0ull, hi = 0ull, feet = 0ull;
      for (int t = 0; t < RC_NPAIR_ROUNDS; ++t) {
        const int pr = c.pairs[t * 64 + lane], i = pr >> 8, j = pr & 0xFF;
        bool ok = pr != 0xFFFF;
        if (ok && i < RC_BALL) ok = i < 2 * R.R;  // feet 2r, 2r + 1 of robot r < R.R
        if (ok && j < RC_BALL) ok = j < 2 * R.R;
        const uint64_t v = (uint64_t)(ok ? pr : 0xFFFF);
        if (t < 4) lo |= v << (16 * t); else hi |= v;
        if (ok && j < RC_BALL && j == i + 1 && !(i & 1)) feet |= 1ull << t;
      }
      tab[lane][0] = lo; tab[lane][1] = hi | (feet << 32);
    }
    e = hipMemcpy(pairTab, tab, sizeof(tab), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(DYNENV_ERR_HIP, hipGetErrorString(e));
  }
  return 0;
}

template <typename T>
static int rows_d2h(T* dst, const T* src, size_t nfields, size_t E, size_t width, int env) {
  for (size_t f = 0; f < nfields; ++f)
    HIP_OK(hipMemcpy(dst + f * width, src + f * E * width + (size_t)env * width, sizeof(T) * width, hipMemcpyDeviceToHost));
  return 0;
}
template <typename T>
static int rows_h2d(T* dst, const T* src, size_t nfields, size_t E, size_t width, int env) {
  for (size_t f = 0; f < nfields; ++f)
    HIP_OK(hipMemcpy(dst + f * E * width + (size_t)env * width, src + f * width, sizeof(T) * width, hipMemcpyHostToDevice));
  return 0;
}

static int rc_get_state_host(dynenv* h, int32_t env, void* blob, size_t nbytes) {
  const RcState& R = h->R;
  if (env < 0 || env >= R.E || nbytes < sizeof(dynenv_robocup_state_t)) return fail(DYNENV_ERR_ARG, "bad env index / size");
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  const size_t E = (size_t)R.E;
  static thread_local double body[RB_COUNT + 4][RC_NB], rob[RR_COUNT][16], envd[RD_COUNT], epr[2][16];
  static thread_local int robi[RI_COUNT][16], envi[RE_COUNT];
  if (rows_d2h(&body[0][0], R.body, RB_COUNT + 4, E, RC_NB, env) || rows_d2h(&rob[0][0], R.rob, RR_COUNT, E, 16, env) ||
      rows_d2h(&robi[0][0], R.robi, RI_COUNT, E, 16, env) || rows_d2h(&epr[0][0], R.epr, 2, E, 16, env))
    return DYNENV_ERR_HIP;
  HIP_OK(hipMemcpy(envi, R.envi + (size_t)env * RE_COUNT, sizeof(envi), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(envd, R.envd + (size_t)env * RD_COUNT, sizeof(envd), hipMemcpyDeviceToHost));
  dynenv_robocup_state_t* st = (dynenv_robocup_state_t*)blob;
  memset(st, 0, sizeof(*st));
  st->elapsed = envi[RE_ELAPSED]; st->n_robots = R.R; st->ball_owned = envi[RE_OWNED]; st->n_last_kicked = envi[RE_NLK];
  for (int i = 0; i < 4; ++i) st->last_kicked[i] = i < envi[RE_NLK] ? envi[RE_LK0 + i] : 0;
  st->goals[0] = envi[RE_GOAL0]; st->goals[1] = envi[RE_GOAL1]; st->closest[0] = envi[RE_CLOSE0]; st->closest[1] = envi[RE_CLOSE1];
  for (int t = 0; t < 2; ++t) {  // defenders are a set on the device: reported in ascending id order
    int n = 0;
    for (int i = 0; i < DYNENV_MAX_ROBOTS; ++i) if (envi[RE_DEF0 + t] & (1 << i)) st->defenders[t][n++] = i;
    st->n_def[t] = n;
    st->penal_times[t] = envd[RD_PT0 + t];
  }
  st->episode = envi[RE_EPISODE];
  st->ball_free_cntr = envd[RD_FREECNT]; st->grace_period = envd[RD_GRACE];
  st->bpx = body[RB_PX][RC_BALL]; st->bpy = body[RB_PY][RC_BALL]; st->bvx = body[RB_VX][RC_BALL]; st->bvy = body[RB_VY][RC_BALL];
  st->bw = body[RB_W][RC_BALL]; st->bprevx = envd[RD_BPREVX]; st->bprevy = envd[RD_BPREVY];
  for (int i = 0; i < DYNENV_MAX_ROBOTS; ++i) { st->episode_r[i] = epr[0][i]; st->episode_pos_r[i] = epr[1][i]; }
  for (int i = 0; i < R.R; ++i) {
    dynenv_robot_state_t& s = st->robots[i];
    const int l = 2 * i, r = 2 * i + 1, f = robi[RI_FLAGS][i];
    s.lpx = body[RB_PX][l]; s.lpy = body[RB_PY][l]; s.lvx = body[RB_VX][l]; s.lvy = body[RB_VY][l]; s.la = body[RB_ANG][l]; s.lw = body[RB_W][l];
    s.rpx = body[RB_PX][r]; s.rpy = body[RB_PY][r]; s.rvx = body[RB_VX][r]; s.rvy = body[RB_VY][r]; s.ra = body[RB_ANG][r]; s.rw = body[RB_W][r];
    s.head_angle = rob[RR_HEAD][i]; s.head_moving = rob[RR_HEADMOV][i]; s.prevx = rob[RR_PREVX][i]; s.prevy = rob[RR_PREVY][i];
    s.initx = rob[RR_INITX][i]; s.inity = rob[RR_INITY][i]; s.penal_time = rob[RR_PENALT][i]; s.fall_time = rob[RR_FALLT][i];
    s.move_time = rob[RR_MOVET][i];
    s.team = (f & RF_TEAMPOS) ? 1 : -1; s.penalized = !!(f & RF_PENAL); s.touching = !!(f & RF_TOUCH); s.might_push = !!(f & RF_PUSH);
    s.fallen = !!(f & RF_FALLEN); s.kicking = !!(f & RF_KICK); s.foot = !!(f & RF_FOOT); s.joint_removed = !!(f & RF_JREM);
    s.touch_cntr = robi[RI_TOUCHC][i]; s.fall_cntr = robi[RI_FALLC][i];
  }
  return DYNENV_OK;
}

static int rc_set_state_host(dynenv* h, int32_t env, const void* blob, size_t nbytes) {
  const RcState& R = h->R;
  if (env < 0 || env >= R.E || nbytes < sizeof(dynenv_robocup_state_t)) return fail(DYNENV_ERR_ARG, "bad env index / size");
  const dynenv_robocup_state_t* st = (const dynenv_robocup_state_t*)blob;
  if (st->n_robots != R.R) return fail(DYNENV_ERR_ARG, "state blob does not match this handle's layout");
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  const size_t E = (size_t)R.E;
  static thread_local double body[RB_COUNT + 4][RC_NB], rob[RR_COUNT][16], envd[RD_COUNT], epr[2][16];
  static thread_local int robi[RI_COUNT][16], envi[RE_COUNT];
  memset(body, 0, sizeof(body)); memset(rob, 0, sizeof(rob)); memset(envd, 0, sizeof(envd)); memset(epr, 0, sizeof(epr));
  memset(robi, 0, sizeof(robi)); memset(envi, 0, sizeof(envi));
  int ncon = 0;
  for (int i = 0; i < R.R; ++i) {
    const dynenv_robot_state_t& s = st->robots[i];
    const int l = 2 * i, r = 2 * i + 1;
    body[RB_PX][l] = s.lpx; body[RB_PY][l] = s.lpy; body[RB_VX][l] = s.lvx; body[RB_VY][l] = s.lvy; body[RB_ANG][l] = s.la; body[RB_W][l] = s.lw;
    body[RB_PX][r] = s.rpx; body[RB_PY][r] = s.rpy; body[RB_VX][r] = s.rvx; body[RB_VY][r] = s.rvy; body[RB_ANG][r] = s.ra; body[RB_W][r] = s.rw;
    for (int k = 0; k < 2; ++k) {  // shape cache = geometry at cpSpaceAddShape time
      const int b = 2 * i + k;
      double sn, cs;
      dm_sincos(body[RB_ANG][b], &sn, &cs);
      body[RB_COUNT + 0][b] = body[RB_PX][b]; body[RB_COUNT + 1][b] = body[RB_PY][b]; body[RB_COUNT + 2][b] = cs; body[RB_COUNT + 3][b] = sn;
    }
    rob[RR_HEAD][i] = s.head_angle; rob[RR_HEADMOV][i] = s.head_moving; rob[RR_PREVX][i] = s.prevx; rob[RR_PREVY][i] = s.prevy;
    rob[RR_INITX][i] = s.initx; rob[RR_INITY][i] = s.inity; rob[RR_PENALT][i] = s.penal_time; rob[RR_FALLT][i] = s.fall_time;
    rob[RR_MOVET][i] = s.move_time;
    int f = (s.team > 0 ? RF_TEAMPOS : 0) | (s.penalized ? RF_PENAL : 0) | (s.touching ? RF_TOUCH : 0) | (s.might_push ? RF_PUSH : 0) |
            (s.fallen ? RF_FALLEN : 0) | (s.kicking ? RF_KICK : 0) | (s.foot ? RF_FOOT : 0) | (s.joint_removed ? RF_JREM : 0);
    robi[RI_FLAGS][i] = f; robi[RI_TOUCHC][i] = s.touch_cntr; robi[RI_FALLC][i] = s.fall_cntr;
    if (!s.joint_removed) envi[RE_CORDER + ncon++] = 2 * i;
    envi[RE_CORDER + ncon++] = 2 * i + 1;
  }
  body[RB_PX][RC_BALL] = st->bpx; body[RB_PY][RC_BALL] = st->bpy; body[RB_VX][RC_BALL] = st->bvx; body[RB_VY][RC_BALL] = st->bvy;
  body[RB_W][RC_BALL] = st->bw; body[RB_COUNT + 0][RC_BALL] = st->bpx; body[RB_COUNT + 1][RC_BALL] = st->bpy; body[RB_COUNT + 2][RC_BALL] = 1.0;
  envi[RE_ELAPSED] = st->elapsed; envi[RE_OWNED] = st->ball_owned; envi[RE_NLK] = st->n_last_kicked;
  for (int i = 0; i < 4; ++i) envi[RE_LK0 + i] = st->last_kicked[i];
  envi[RE_GOAL0] = st->goals[0]; envi[RE_GOAL1] = st->goals[1]; envi[RE_CLOSE0] = st->closest[0]; envi[RE_CLOSE1] = st->closest[1];
  for (int t = 0; t < 2; ++t) {
    int m = 0;
    for (int i = 0; i < st->n_def[t]; ++i) m |= 1 << st->defenders[t][i];
    envi[RE_DEF0 + t] = m;
    envd[RD_PT0 + t] = st->penal_times[t];
  }
  envi[RE_NCON] = ncon; envi[RE_EPISODE] = st->episode; envi[RE_OCC] = 0; envi[RE_ERR] = 0;
  envd[RD_FREECNT] = st->ball_free_cntr; envd[RD_GRACE] = st->grace_period; envd[RD_BPREVX] = st->bprevx; envd[RD_BPREVY] = st->bprevy;
  for (int i = 0; i < DYNENV_MAX_ROBOTS; ++i) { epr[0][i] = st->episode_r[i]; epr[1][i] = st->episode_pos_r[i]; }
  if (rows_h2d(R.body, &body[0][0], RB_COUNT + 4, E, RC_NB, env) || rows_h2d(R.rob, &rob[0][0], RR_COUNT, E, 16, env) ||
      rows_h2d(R.robi, &robi[0][0], RI_COUNT, E, 16, env) || rows_h2d(R.epr, &epr[0][0], 2, E, 16, env))
    return DYNENV_ERR_HIP;
  HIP_OK(hipMemcpy(R.envi + (size_t)env * RE_COUNT, envi, sizeof(envi), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(R.envd + (size_t)env * RD_COUNT, envd, sizeof(envd), hipMemcpyHostToDevice));
  return DYNENV_OK;
}

extern "C" {

int dynenv_abi_version(void) { return DYNENV_ABI_VERSION; }
const char* dynenv_last_error(void) { return g_err.c_str(); }

int dynenv_create(const dynenv_cfg_t* cfg, dynenv_t** out) {
  if (!cfg || !out) return fail(DYNENV_ERR_ARG, "null argument");
  if (cfg->abi_version != DYNENV_ABI_VERSION) return fail(DYNENV_ERR_ARG, "ABI version mismatch");
  if (cfg->num_envs <= 0 || cfg->n_players <= 0) return fail(DYNENV_ERR_ARG, "num_envs and n_players must be > 0");
  if (cfg->noise_magnitude < 0 || cfg->noise_magnitude > 5)
    return fail(DYNENV_ERR_ARG, "Error: The noise magnitude must be between 0 and 5!");  // environment_base.py:162-164
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(DYNENV_ERR_NO_DEVICE, "no HIP device visible: libdynenv_hip has no CPU fallback");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(DYNENV_ERR_ARG, "device_id out of range");
  DeviceGuard guard_(cfg->device_id);  // (the __constant__ tables are uploaded to this device below)
  if (!guard_.ok) return fail(DYNENV_ERR_HIP, "hipSetDevice failed");
  if (cfg->env_type != DYNENV_DRIVE && cfg->env_type != DYNENV_ROBO_CUP) return fail(DYNENV_ERR_ARG, "unknown env_type");
  const int32_t known = DYNENV_FLAG_RANDOM_INIT | DYNENV_FLAG_DETERMINISTIC_TURN | DYNENV_FLAG_CAN_FALL | DYNENV_FLAG_USE_OBS_REWARDS |
                        DYNENV_FLAG_ALLOW_HEAD_TURN;
  if (cfg->flags & ~known) return fail(DYNENV_ERR_UNSUPPORTED, "unknown bits in dynenv_cfg.flags");
  if (cfg->env_type == DYNENV_DRIVE && cfg->flags != 0)
    return fail(DYNENV_ERR_UNSUPPORTED, "dynenv_cfg.flags are RoboCup's class switches; DrivingEnvironment has none (continuous actions are broken in the reference, DrivingEnvironment.py:360-368)");
  if (cfg->obs_type != DYNENV_OBS_FULL && cfg->obs_type != DYNENV_OBS_PARTIAL)
    return fail(DYNENV_ERR_UNSUPPORTED, "Image observations are out of scope");
  dynenv* h = new dynenv();
  h->cfg = *cfg;
  h->robocup = cfg->env_type == DYNENV_ROBO_CUP;
  h->partial = false;
  if (h->robocup) {
    int rc = rc_create(h);
    if (rc) { dynenv_destroy(h); return rc; }
    *out = h;
    return DYNENV_OK;
  }
  h->A = cfg->n_players > DRV_MAXA ? DRV_MAXA : cfg->n_players;  // environment_base.py:57
  h->partial = cfg->obs_type == DYNENV_OBS_PARTIAL;
  h->obs_dim = h->partial ? PV_DIM : 9 + (h->A - 1) * 7 + DRV_MAXO * 4 + DRV_MAXP * 2 + DRV_LANE_ROWS * 5;
  h->T = 1;
  h->action_dim = 2;
  DrvState& S = h->S;
  memset(&S, 0, sizeof(S));
  const size_t E = (size_t)cfg->num_envs;
  S.E = (int)E; S.A = h->A; S.obs_dim = h->partial ? 9 + (h->A - 1) * 7 + DRV_MAXO * 4 + DRV_MAXP * 2 + DRV_LANE_ROWS * 5 : h->obs_dim; S.seed = cfg->seed; S.env_id_offset = cfg->env_id_offset;
  int rc = 0;
  rc |= dev_alloc(h, &S.body, (size_t)BF_COUNT * E * DRV_NB);
  rc |= dev_alloc(h, &S.carx, (size_t)CF_COUNT * E * 16);
  rc |= dev_alloc(h, &S.flags, E * DRV_NB);
  rc |= dev_alloc(h, &S.aux, E * DRV_NB);
  rc |= dev_alloc(h, &S.obst, 2 * E * DRV_MAXO);
  rc |= dev_alloc(h, &S.envi, E * EI_COUNT);
  rc |= dev_alloc(h, &S.epr, 2 * E * 16);
  rc |= dev_alloc(h, &S.s_pair, E * DRV_NS);
  rc |= dev_alloc(h, &S.s_meta, E * DRV_NS);
  rc |= dev_alloc(h, &S.s_hash, 2 * E * DRV_NS);
  rc |= dev_alloc(h, &S.s_imp, 4 * E * DRV_NS);
  rc |= dev_alloc(h, &S.lastcand, E * 64);
  rc |= dev_alloc_scratch(h, &S.iso, DRV_ISO_WORDS);
  rc |= dev_alloc_scratch(h, &S.iso_done, E);
  rc |= dev_alloc_scratch(h, &S.iso_hw, 2 * 4 * DRV_ISO_GROUPS);
  rc |= dev_alloc_scratch(h, &S.pvq, 32 + 2 * (size_t)S.E);
  if (rc) { dynenv_destroy(h); return DYNENV_ERR_HIP; }
  {
    // The slowest environments of the previous step get a SIMD to themselves (drv_iso_assign): only where the block -> SIMD
    // pattern it relies on was measured - 4096 environments = one residency round of a 256-CU device, four waves per SIMD.
    hipDeviceProp_t prop;
    S.iso_on = 0;
    if (hipGetDeviceProperties(&prop, cfg->device_id) == hipSuccess)
    {
      const bool dev256 = prop.multiProcessorCount * 4 == DRV_ISO_GROUPS, off = getenv("DYNENV_NO_ISOLATION") != nullptr;
      // 1: isolation - one residency round, Full observations (Partial: the fused observation makes the displaced environments
      //    too long for the second round: +1.4 %);  2: more environments than fit at once - the slow ones simply start first;
      // 3: Partial observations - nothing is rescheduled, the step's slowest environment is timed all the same: an environment on
      //    the contact path runs as many of its own vision passes as fit before that one is done (drv_step_body)
      // (round 5: mode 1 for every E in (DRV_ISO_MIN_E, 4096], not for E == 4096 alone - the launch is 4096 regular blocks + spares
      //  whatever E is, and the blocks without an environment end at once; below DRV_ISO_MIN_E a SIMD holds two waves or fewer)
      const int isoMinE = getenv("DYNENV_ISO_MIN_E") ? atoi(getenv("DYNENV_ISO_MIN_E")) : DRV_ISO_MIN_E;
      S.iso_on = off || !dev256 ? 0 : ((int)E > isoMinE && E <= 4 * DRV_ISO_GROUPS ? (h->partial ? 0 : 1) : (E > 4 * DRV_ISO_GROUPS ? 2 : 0));
      if (!off && S.iso_on == 0 && h->partial) S.iso_on = 3;
    }
    if (iso_reset(h)) { dynenv_destroy(h); return DYNENV_ERR_HIP; }
    h->iso_cfg = S.iso_on;
    if (S.iso_on == 1) {
      if (hipHostMalloc((void**)&h->iso_seen, sizeof(int), hipHostMallocDefault) != hipSuccess) { dynenv_destroy(h); return fail(DYNENV_ERR_HIP, "hipHostMalloc"); }
      *h->iso_seen = 0;
    }
  }
  DrvConst c;
  build_consts(c);
  if (!drv_literals_ok(c)) {
    dynenv_destroy(h);
    return fail(DYNENV_ERR_HIP, "internal: RoadK / CarK literals differ from the computed constants");
  }
  hipError_t e = drv_upload_consts(c);
  if (e != hipSuccess) { dynenv_destroy(h); return fail(DYNENV_ERR_HIP, hipGetErrorString(e)); }
  *out = h;
  return DYNENV_OK;
}

void dynenv_destroy(dynenv_t* h) {
  if (!h) return;
  DeviceGuard guard_(h->cfg.device_id);
  for (void* p : h->allocs) hipFree(p);
  for (void* p : h->scratch) hipFree(p);
  if (h->iso_seen) (void)hipHostFree(h->iso_seen);
  delete h;
}

int dynenv_layout(const dynenv_t* h, dynenv_layout_t* L) {
  if (!h || !L) return fail(DYNENV_ERR_ARG, "null argument");
  memset(L, 0, sizeof(*L));
  int A = h->A;
  L->num_envs = h->cfg.num_envs; L->n_agents = A; L->n_time_steps = h->T; L->obs_dim = h->obs_dim;
  L->action_dim = h->action_dim;
  if (h->robocup && h->cfg.obs_type == DYNENV_OBS_PARTIAL) {
    // ((balls, robots), (goals, crosses, line crosses, lines), (numLandMarks, robotsSeen, ballsSeen)) of getAgentVision;
    // block 6 = the tail: 6 list lengths, numLandMarks, ballsSeen, robotsSeen[9]
    const int off[7] = {RCP_OFF_BALL, RCP_OFF_ROB, RCP_OFF_GOAL, RCP_OFF_CROSS, RCP_OFF_FCROSS, RCP_OFF_LINE, RCP_OFF_TAIL};
    const int rows[7] = {RCP_CAP_BALL, RCP_CAP_ROB, RCP_CAP_GOAL, RCP_CAP_CROSS, RCP_CAP_FCROSS, RCP_CAP_LINE, 1};
    const int feat[7] = {5, 7, 6, 6, 8, 5, 17};
    L->n_blocks = 7;
    for (int i = 0; i < 7; ++i) { L->block_offset[i] = off[i]; L->block_rows[i] = rows[i]; L->block_feat[i] = feat[i]; }
    L->steps_per_episode = RC_MAX_TIME / 50;
    return DYNENV_OK;
  }
  if (h->robocup) {
    L->n_blocks = 3;  // ((ball, robots), (self,)) of RoboCupEnvironment.py:440-443
    L->block_offset[0] = 0; L->block_rows[0] = 1; L->block_feat[0] = 4;
    L->block_offset[1] = 4; L->block_rows[1] = 1; L->block_feat[1] = 8;
    L->block_offset[2] = 12; L->block_rows[2] = A - 1; L->block_feat[2] = 6;
    L->steps_per_episode = RC_MAX_TIME / 50;
    return DYNENV_OK;
  }
  if (h->partial) {  // ((cars, obstacles, pedestrians), (self, lanes)) of getAgentVision + the 4 row counts
    L->n_blocks = 6;
    L->block_offset[0] = 0; L->block_rows[0] = 1; L->block_feat[0] = 9;
    L->block_offset[1] = PV_OFF_CARS; L->block_rows[1] = PV_CAP_CARS; L->block_feat[1] = 7;
    L->block_offset[2] = PV_OFF_OBST; L->block_rows[2] = PV_CAP_OBST; L->block_feat[2] = 6;
    L->block_offset[3] = PV_OFF_PEDS; L->block_rows[3] = PV_CAP_PEDS; L->block_feat[3] = 2;
    L->block_offset[4] = PV_OFF_LANES; L->block_rows[4] = PV_CAP_LANES; L->block_feat[4] = 4;
    L->block_offset[5] = PV_DIM - 4; L->block_rows[5] = 1; L->block_feat[5] = 4;
    L->steps_per_episode = DRV_MAX_TIME / 10;
    return DYNENV_OK;
  }
  L->n_blocks = 5;
  L->block_offset[0] = 0; L->block_rows[0] = 1; L->block_feat[0] = 9;
  L->block_offset[1] = 9; L->block_rows[1] = A - 1; L->block_feat[1] = 7;
  L->block_offset[2] = 9 + (A - 1) * 7; L->block_rows[2] = DRV_MAXO; L->block_feat[2] = 4;
  L->block_offset[3] = L->block_offset[2] + DRV_MAXO * 4; L->block_rows[3] = DRV_MAXP; L->block_feat[3] = 2;
  L->block_offset[4] = L->block_offset[3] + DRV_MAXP * 2; L->block_rows[4] = DRV_LANE_ROWS; L->block_feat[4] = 5;
  L->steps_per_episode = DRV_MAX_TIME / 10;
  return DYNENV_OK;
}

int dynenv_seed(dynenv_t* h, uint64_t seed) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  h->cfg.seed = seed;
  h->S.seed = seed;
  h->R.seed = seed;
  return DYNENV_OK;
}

int dynenv_reset(dynenv_t* h, float* obs_dev, void* stream) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  hipStream_t st = (hipStream_t)stream;
  ON_DEVICE(h);
  if (h->robocup) {
    int E = h->R.E;
    hipLaunchKernelGGL(rc_reset_kernel, dim3((E + 63) / 64), dim3(64), 0, st, h->R);
    if (obs_dev) {
      hipLaunchKernelGGL(rc_obs_kernel, dim3(E), dim3(64), 0, st, h->R, obs_dev, 0);
      if (h->R.obs_type == DYNENV_OBS_PARTIAL)
        hipLaunchKernelGGL(rc_partial_obs_kernel, dim3(E), dim3(64), 0, st, h->R, obs_dev, (double*)nullptr);
    }
    HIP_OK(hipGetLastError());
    return DYNENV_OK;
  }
  int E = h->S.E;
  hipLaunchKernelGGL(drv_reset_kernel, dim3((E + 63) / 64), dim3(64), 0, st, h->S);
  if (obs_dev && h->partial)
    hipLaunchKernelGGL(drv_partial_obs_kernel, dim3(E), dim3(64), 0, st, h->S, (int)h->cfg.noise_type, h->cfg.noise_magnitude, obs_dev);
  else if (obs_dev)
    hipLaunchKernelGGL(drv_obs_kernel, dim3(E), dim3(64), 0, st, h->S, obs_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_full_obs_dim(const dynenv_t* h) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  return h->robocup ? 4 + 8 + (h->R.R - 1) * 6 : 9 + (h->A - 1) * 7 + DRV_MAXO * 4 + DRV_MAXP * 2 + DRV_LANE_ROWS * 5;
}

int dynenv_full_obs(dynenv_t* h, float* full_dev, void* stream) {
  if (!h || !full_dev) return fail(DYNENV_ERR_ARG, "null argument");
  ON_DEVICE(h);
  hipStream_t st = (hipStream_t)stream;
  if (h->robocup) hipLaunchKernelGGL(rc_obs_kernel, dim3(h->R.E), dim3(64), 0, st, h->R, full_dev, 1);
  else hipLaunchKernelGGL(drv_obs_kernel, dim3(h->S.E), dim3(64), 0, st, h->S, full_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_global_state_dim(const dynenv_t* h) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  return h->robocup ? h->R.R * 6 + 3 : 0;
}

int dynenv_global_state(dynenv_t* h, float* state_dev, void* stream) {
  if (!h || !state_dev) return fail(DYNENV_ERR_ARG, "null argument");
  if (!h->robocup)
    return fail(DYNENV_ERR_UNSUPPORTED, "Driving: getFullState(None) is made of columns of dynenv_full_obs (every car's own self block); no separate emit");
  ON_DEVICE(h);
  hipLaunchKernelGGL(rc_global_state_kernel, dim3((h->R.E + 3) / 4), dim3(64), 0, (hipStream_t)stream, h->R, state_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_step(dynenv_t* h, const int32_t* actions_dev, float* obs_dev, double* rewards_dev, uint8_t* dones_dev,
                void* stream) {
  return dynenv_step_head(h, actions_dev, nullptr, obs_dev, rewards_dev, dones_dev, stream);
}

int dynenv_step_head(dynenv_t* h, const int32_t* actions_dev, const double* head_dev, float* obs_dev, double* rewards_dev,
                     uint8_t* dones_dev, void* stream) {
  if (!h || !actions_dev || !rewards_dev || !dones_dev) return fail(DYNENV_ERR_ARG, "null argument");
  if (head_dev && !(h->robocup && (h->cfg.flags & DYNENV_FLAG_ALLOW_HEAD_TURN)))
    return fail(DYNENV_ERR_ARG, "the continuous head channel exists for RoboCup with DYNENV_FLAG_ALLOW_HEAD_TURN only");
  ON_DEVICE(h);
  hipStream_t st = (hipStream_t)stream;
  // measurement hook (dynenv_set_step_events): the step's dominant kernel bracketed by events on the launch stream
  struct StepEvents {
    dynenv* h; hipStream_t st;
    void main_done() { if (h->ev_main) (void)hipEventRecord(h->ev_main, st); }
    ~StepEvents() { if (h->ev_end) (void)hipEventRecord(h->ev_end, st); }
  } sev{h, st};
  if (h->robocup) {
    if (h->R.obs_type == DYNENV_OBS_PARTIAL && obs_dev) {  // getAgentVision at the five snapshots + processSeens fused into the launch
      HIP_OK(hipMemsetAsync(h->R.deferList, 0, sizeof(int), st));
      if (h->ev_begin) HIP_OK(hipEventRecord(h->ev_begin, st));
      hipLaunchKernelGGL(rc_step_partial_kernel, dim3(h->R.E), dim3(64), 0, st, h->R, (const int*)actions_dev, head_dev, obs_dev, rewards_dev, dones_dev);
      sev.main_done();
      const int nb = h->R.E < RC_DEFER_BLOCKS ? h->R.E : RC_DEFER_BLOCKS;  // the deferred environments are few: blocks stride over their list
      hipLaunchKernelGGL(rc_partial_obs_deferred_kernel, dim3(nb, 5, h->R.R), dim3(64), 0, st, h->R, obs_dev);
      hipLaunchKernelGGL(rc_partial_finalize_kernel, dim3(nb), dim3(64), 0, st, h->R, rewards_dev);
    }
    else if (h->R.obs_type == DYNENV_OBS_PARTIAL)
      return fail(DYNENV_ERR_ARG, "RoboCup Partial: the observation buffer is required (the processSeens rewards come out of the same pass)");
    else {
      if (h->ev_begin) HIP_OK(hipEventRecord(h->ev_begin, st));
      hipLaunchKernelGGL(rc_step_kernel, dim3(h->R.E), dim3(64), 0, st, h->R, (const int*)actions_dev, head_dev, obs_dev, rewards_dev, dones_dev);
      sev.main_done();
    }
    HIP_OK(hipGetLastError());
    return DYNENV_OK;
  }
  // Partial: getAgentVision for every agent (DrivingEnvironment.py:294) is fused into the step kernel - each wave writes its
  // environment's observation as soon as its step is done, which fills the launch's tail
  // (with isolation on, 3 x DRV_ISO_MAX spare blocks behind the E regular ones: the environments displaced from a slow
  //  environment's SIMD run there - or nothing, and the block ends at once)
  // A device this handle does not have to itself (another handle stepping on another stream, another kernel): the placement
  // does not validate, isolation holds off on the device - and what is left of it (the spare blocks, the placement record, the
  // per-environment report) still costs 0.5-0.9 % of a step.  So the host looks at the device's count of
  // launches that did not validate every ISO_PROBE_EVERY steps - an asynchronous 4-byte copy, read one window later, never
  // waited for - and when more than half of a window's launches did not validate it drops to the plain launch (mode 0: exactly
  // what DYNENV_NO_ISOLATION=1 gives) for ISO_PAUSE_STEPS steps, then starts isolation over.  Scheduling only, as ever.
  // A step that is being CAPTURED into a hipGraph will be replayed with frozen kernel arguments, and none of the host logic below
  // runs at replay: from the first captured step on, for good, the handle keeps tick / pv_par in two device words that a one-thread
  // kernel advances in front of every step (eager steps included: ~2 us each), the isolation pause logic is off (the device-side
  // validation still holds isolation off on a shared device), and a paused isolation stays paused.
  if (!h->S.tick_src) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) h->S.tick_src = 1;
  }
  if (h->S.tick_src) hipLaunchKernelGGL(drv_tick_advance_kernel, dim3(1), dim3(64), 0, st, h->S, (h->partial && obs_dev) ? 1 : 0);
  if (h->ev_begin) HIP_OK(hipEventRecord(h->ev_begin, st));  // (behind the advance kernel: the event window is the step kernel's own duration)
  if (h->iso_cfg == 1 && !h->S.tick_src) {
    h->steps += 1;
    if (h->S.iso_on == 0 && h->steps >= h->iso_paused_until) {
      if (iso_reset_async(h, st)) return DYNENV_ERR_HIP;
      h->S.iso_on = 1; h->iso_last_invalid = 0; *h->iso_seen = 0;
    } else if (h->S.iso_on == 1 && h->steps % ISO_PROBE_EVERY == 0) {
      const int seen = *(volatile int*)h->iso_seen;  // the copy issued one window ago (stale at worst)
      if (seen - h->iso_last_invalid > ISO_PROBE_EVERY / 2) {
        h->S.iso_on = 0; h->iso_paused_until = h->steps + ISO_PAUSE_STEPS; h->iso_pauses += 1;
      } else {
        h->iso_last_invalid = seen;
        HIP_OK(hipMemcpyAsync(h->iso_seen, h->S.iso + 11, sizeof(int), hipMemcpyDeviceToHost, st));
      }
    }
  }
  const unsigned stepGrid = h->S.iso_on == 1 ? 4u * DRV_ISO_GROUPS + 3u * DRV_ISO_MAX + 1u /* one residency round + spares + the placement validator */ : (unsigned)h->S.E;
  if (!h->S.tick_src) h->S.tick = (h->S.tick + 1) % (3 * (1 << 28));  // (wraps at a multiple of 3: the three isolation lists keep rotating in order)
  if (h->partial && obs_dev)
  {
    if (!h->S.tick_src) h->S.pv_par ^= 1;
    hipLaunchKernelGGL(drv_step_partial_kernel, dim3(stepGrid), dim3(64), 0, st, h->S, (const int*)actions_dev, rewards_dev, dones_dev, obs_dev,
                       (int)h->cfg.noise_type, (double)h->cfg.noise_magnitude);
    sev.main_done();
    // one block per (listed environment, agent) in turn, over a grid that fits the device at once: the step launch left a list
    const long long items = (long long)h->S.E * h->S.A;
    hipLaunchKernelGGL(drv_partial_obs_deferred_kernel, dim3((unsigned)(items < 4096 ? items : 4096)), dim3(64), 0, st, h->S,
                       (int)h->cfg.noise_type, (double)h->cfg.noise_magnitude, obs_dev);
  }
  else {
    hipLaunchKernelGGL(drv_step_kernel, dim3(stepGrid), dim3(64), 0, st, h->S, (const int*)actions_dev, h->partial ? (float*)nullptr : obs_dev,
                       rewards_dev, dones_dev);
    sev.main_done();
  }
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_set_step_events(dynenv_t* h, void* ev_begin, void* ev_main_done, void* ev_end) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  h->ev_begin = (hipEvent_t)ev_begin; h->ev_main = (hipEvent_t)ev_main_done; h->ev_end = (hipEvent_t)ev_end;
  return DYNENV_OK;
}

int dynenv_counts(dynenv_t* h, int32_t* counts_dev, void* stream) {
  if (!h || !counts_dev) return fail(DYNENV_ERR_ARG, "null argument");
  ON_DEVICE(h);
  if (h->robocup) { HIP_OK(hipMemsetAsync(counts_dev, 0, sizeof(int32_t) * 2 * h->R.E, (hipStream_t)stream)); return DYNENV_OK; }
  int E = h->S.E;
  hipLaunchKernelGGL(drv_counts_kernel, dim3((E + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->S, (int*)counts_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_episode_stats(dynenv_t* h, double* ep_r, double* ep_pos_r, double* ep_obs_r, int32_t* goals, void* stream) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  ON_DEVICE(h);
  if (h->robocup) {
    int E = h->R.E;
    hipLaunchKernelGGL(rc_stats_kernel, dim3((E + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->R, ep_r, ep_pos_r, ep_obs_r, (int*)goals);
    HIP_OK(hipGetLastError());
    return DYNENV_OK;
  }
  int E = h->S.E;
  hipLaunchKernelGGL(drv_stats_kernel, dim3((E + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->S, ep_r, ep_pos_r, ep_obs_r,
                     (int*)goals);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

size_t dynenv_state_size(const dynenv_t* h) { return (h && h->robocup) ? sizeof(dynenv_robocup_state_t) : sizeof(dynenv_driving_state_t); }

int dynenv_sync(dynenv_t* h, void* stream) {
  if (!h) return fail(DYNENV_ERR_ARG, "null handle");
  ON_DEVICE(h);
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  return DYNENV_OK;
}

// error flags raised by the kernels (bit0: contact cache overflow), OR-ed over all envs
int dynenv_error_flags(dynenv_t* h, int32_t* out) {
  if (!h || !out) return fail(DYNENV_ERR_ARG, "null argument");
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  if (h->robocup) {
    std::vector<int> ev((size_t)h->R.E * RE_COUNT);
    HIP_OK(hipMemcpy(ev.data(), h->R.envi, ev.size() * sizeof(int), hipMemcpyDeviceToHost));
    int fl = 0;
    for (int e = 0; e < h->R.E; ++e) fl |= ev[(size_t)e * RE_COUNT + RE_ERR];
    *out = fl;
    return DYNENV_OK;
  }
  std::vector<int> envi((size_t)h->S.E * EI_COUNT);
  HIP_OK(hipMemcpy(envi.data(), h->S.envi, envi.size() * sizeof(int), hipMemcpyDeviceToHost));
  int f = 0;
  for (int e = 0; e < h->S.E; ++e) f |= envi[(size_t)e * EI_COUNT + EI_ERR];
  *out = f;
  return DYNENV_OK;
}

// diagnostics: per-path substep counts since the last reset, summed over envs: out8 = {fast, quiescent, contact,
// slot-sum, contact-because-candidates-changed, -because-a-body-moved, -because-an-arbiter-was-not-inert,
// steady replays, 0...}
int dynenv_debug_counters(dynenv_t* h, int64_t* out4) {
  if (!h || !out4) return fail(DYNENV_ERR_ARG, "null argument");
  ON_DEVICE(h);
  if (h->robocup) {
    for (int k = 0; k < 16; ++k) out4[k] = 0;
#ifdef DRV_PROFILE
    HIP_OK(hipDeviceSynchronize());
    { static unsigned long long d[4096 * 12]; HIP_OK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_rcprof), sizeof(d))); FILE* f = fopen("gpurun_out/rcprof.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 12; ++q) fprintf(f, "%llu ", d[12*k+q]); fprintf(f, "\n"); } fclose(f); }
    { static unsigned long long d[4096 * 8]; HIP_OK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_rcprof2), sizeof(d))); FILE* f = fopen("gpurun_out/rcprof2.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", d[8*k+q]); fprintf(f, "\n"); } fclose(f); }
    { static unsigned long long d[4096 * 8]; HIP_OK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_rcprof3), sizeof(d))); FILE* f = fopen("gpurun_out/rcprof3.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", d[8*k+q]); fprintf(f, "\n"); } fclose(f); }
    { static unsigned long long d[4096 * 8]; HIP_OK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_rcprof4), sizeof(d))); FILE* f = fopen("gpurun_out/rcprof4.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", d[8*k+q]); fprintf(f, "\n"); } fclose(f); }
#endif
    return DYNENV_OK;
  }
  HIP_OK(hipDeviceSynchronize());
  std::vector<int> envi((size_t)h->S.E * EI_COUNT);
  HIP_OK(hipMemcpy(envi.data(), h->S.envi, envi.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int k = 0; k < 16; ++k) out4[k] = 0;
  for (int e = 0; e < h->S.E; ++e)
    for (int k = 0; k < 10; ++k) out4[k] += envi[(size_t)e * EI_COUNT + EI_N_FAST + k];
  {  // SIMD isolation: how many environments the next step isolates, placeholders that gave up waiting (should stay 0)
    int iso[DRV_ISO_HDR];
    HIP_OK(hipMemcpy(iso, h->S.iso, sizeof(iso), hipMemcpyDeviceToHost));
    const int nxt = ((h->S.tick_src ? iso[13] : h->S.tick) + 1) % 3, k = iso[nxt];
    const int cap = h->S.iso_on == 1 ? DRV_ISO_MAX : DRV_ISO_LIST;
    out4[10] = h->S.iso_on ? (k < cap ? k : cap) : -1; out4[11] = iso[7];
    // out[12]: scheduling mode (0 off, 1 SIMD isolation, 2 slow environments first, 3 timing only: Partial); out[13]: 1 = the next step found the block ->
    // SIMD placement validated (mode 1 only; 0 = isolation is holding off); out[14]: launches whose placement did not validate
    out4[12] = h->iso_cfg; out4[13] = h->S.iso_on == 1 ? iso[8 + nxt] : (h->iso_cfg == 1 ? 0 : -1); out4[14] = iso[11];
    out4[15] = h->iso_pauses;  // times the host dropped to the plain launch because the placement kept failing to validate
  }
#ifdef DRV_PROFILE
  { unsigned long long d[16]; HIP_OK(drv_prof_read(0, d, sizeof(d))); FILE* f = fopen("gpurun_out/dbgr.txt", "w"); for (int k = 0; k < 16; ++k) fprintf(f, "%llu\n", d[k]); fclose(f); }
  { static unsigned long long d[4096 * 12]; HIP_OK(drv_prof_read(1, d, sizeof(d))); FILE* f = fopen("gpurun_out/dbgw.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 12; ++q) fprintf(f, "%llu ", d[12*k+q]); fprintf(f, "\n"); } fclose(f); }
  { static unsigned long long d[4096 * 8]; HIP_OK(drv_prof_read(2, d, sizeof(d))); FILE* f = fopen("gpurun_out/dbgp.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", d[8*k+q]); fprintf(f, "\n"); } fclose(f); }
  { static unsigned long long d[4096 * 8]; HIP_OK(drv_prof_read(3, d, sizeof(d))); FILE* f = fopen("gpurun_out/dbgs.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", d[8*k+q]); fprintf(f, "\n"); } fclose(f); }
  { static unsigned long long d[4096 * 16]; HIP_OK(drv_prof_read(5, d, sizeof(d))); FILE* f = fopen("gpurun_out/dbgv.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 16; ++q) fprintf(f, "%llu ", d[16*k+q]); fprintf(f, "\n"); } fclose(f); }
  { static unsigned long long d[4096 * 8]; HIP_OK(drv_prof_read(4, d, sizeof(d))); FILE* f = fopen("gpurun_out/dbgl.txt", "w"); for (int k = 0; k < 4096; ++k) { for (int q = 0; q < 8; ++q) fprintf(f, "%llu ", d[8*k+q]); fprintf(f, "\n"); } fclose(f); }
#endif
  return DYNENV_OK;
}

// diagnostics: where the blocks of the last Driving step ran (XCC << 16 | HW_ID bits: SE 15:13, SH 12, CU 11:8, SIMD 5:4), one
// word per regular block; only recorded by handles in isolation mode 1.  Returns the number of words written (<= n), < 0 on error.
int dynenv_debug_placement(dynenv_t* h, uint32_t* out, int32_t n) {
  if (!h || !out || n < 0) return fail(DYNENV_ERR_ARG, "bad argument");
  if (h->robocup || h->S.iso_on != 1) return 0;
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  const int m = n < 4 * DRV_ISO_GROUPS ? n : 4 * DRV_ISO_GROUPS;
  int tick = h->S.tick;
  if (h->S.tick_src) HIP_OK(hipMemcpy(&tick, h->S.iso + 13, sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out, h->S.iso_hw + (size_t)(tick & 1) * (4 * DRV_ISO_GROUPS), sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
  return m;
}

int dynenv_get_state(dynenv_t* h, int32_t env, void* blob, size_t nbytes) {
  if (!h || !blob) return fail(DYNENV_ERR_ARG, "null argument");
  if (h->robocup) return rc_get_state_host(h, env, blob, nbytes);
  if (env < 0 || env >= h->S.E || nbytes < sizeof(dynenv_driving_state_t)) return fail(DYNENV_ERR_ARG, "bad env index / size");
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  const DrvState& S = h->S;
  const size_t E = (size_t)S.E;
  double body[BF_COUNT][DRV_NB], carx[CF_COUNT][16], obst[2][DRV_MAXO], epr[2][16];
  int flags[DRV_NB], aux[DRV_NB], envi[EI_COUNT];
  for (int f = 0; f < BF_COUNT; ++f)
    HIP_OK(hipMemcpy(body[f], S.body + (size_t)f * E * DRV_NB + (size_t)env * DRV_NB, sizeof(double) * DRV_NB, hipMemcpyDeviceToHost));
  for (int f = 0; f < CF_COUNT; ++f)
    HIP_OK(hipMemcpy(carx[f], S.carx + (size_t)f * E * 16 + (size_t)env * 16, sizeof(double) * 16, hipMemcpyDeviceToHost));
  for (int f = 0; f < 2; ++f) {
    HIP_OK(hipMemcpy(obst[f], S.obst + (size_t)f * E * DRV_MAXO + (size_t)env * DRV_MAXO, sizeof(double) * DRV_MAXO, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(epr[f], S.epr + (size_t)f * E * 16 + (size_t)env * 16, sizeof(double) * 16, hipMemcpyDeviceToHost));
  }
  HIP_OK(hipMemcpy(flags, S.flags + (size_t)env * DRV_NB, sizeof(flags), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(aux, S.aux + (size_t)env * DRV_NB, sizeof(aux), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(envi, S.envi + (size_t)env * EI_COUNT, sizeof(envi), hipMemcpyDeviceToHost));
  dynenv_driving_state_t* st = (dynenv_driving_state_t*)blob;
  memset(st, 0, sizeof(*st));
  st->elapsed = envi[EI_ELAPSED]; st->all_finished = envi[EI_ALLFIN]; st->n_cars = S.A;
  st->n_peds = envi[EI_NPED]; st->n_obst = envi[EI_NOBST]; st->episode = envi[EI_EPISODE];
  for (int i = 0; i < DYNENV_MAX_CARS; ++i) { st->episode_r[i] = epr[0][i]; st->episode_pos_r[i] = epr[1][i]; }
  for (int i = 0; i < S.A; ++i) {
    dynenv_car_state_t& c = st->cars[i];
    c.px = body[BF_PX][i]; c.py = body[BF_PY][i]; c.vx = body[BF_VX][i]; c.vy = body[BF_VY][i];
    c.angle = body[BF_ANG][i]; c.w = body[BF_W][i];
    c.dirx = carx[CF_DIRX][i]; c.diry = carx[CF_DIRY][i]; c.prevx = carx[CF_PREVX][i]; c.prevy = carx[CF_PREVY][i];
    c.goalx = carx[CF_GOALX][i]; c.goaly = carx[CF_GOALY][i];
    int f = flags[i];
    c.type = f & 3; c.team = (f >> 2) & 3; c.finished = (f >> 4) & 1; c.crashed = (f >> 5) & 1;
    c.fric = (f >> 6) & 1; c.lane_pos = (f >> 8) & 7;
  }
  for (int i = 0; i < st->n_peds; ++i) {
    dynenv_ped_state_t& p = st->peds[i];
    int l = DRV_SLOT_PED + i, f = flags[l];
    p.px = body[BF_PX][l]; p.py = body[BF_PY][l]; p.vx = body[BF_VX][l]; p.vy = body[BF_VY][l];
    p.road = f & 1; p.side = (f >> 1) & 1; p.dead = (f >> 2) & 1; p.crossing = (f >> 3) & 1;
    p.begin_crossing = (f >> 4) & 1; p.speed = (f >> 8) & 15; p.moving = aux[l];
  }
  for (int i = 0; i < st->n_obst; ++i) { st->obst_x[i] = obst[0][i]; st->obst_y[i] = obst[1][i]; }
  return DYNENV_OK;
}

int dynenv_set_state(dynenv_t* h, int32_t env, const void* blob, size_t nbytes) {
  if (!h || !blob) return fail(DYNENV_ERR_ARG, "null argument");
  if (h->robocup) return rc_set_state_host(h, env, blob, nbytes);
  if (env < 0 || env >= h->S.E || nbytes < sizeof(dynenv_driving_state_t)) return fail(DYNENV_ERR_ARG, "bad env index / size");
  const dynenv_driving_state_t* st = (const dynenv_driving_state_t*)blob;
  if (st->n_cars != h->S.A || st->n_peds > DRV_MAXP || st->n_obst > DRV_MAXO || st->n_peds < 0 || st->n_obst < 0)
    return fail(DYNENV_ERR_ARG, "state blob does not match this handle's layout");
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  const DrvState& S = h->S;
  const size_t E = (size_t)S.E;
  double body[BF_COUNT][DRV_NB] = {}, carx[CF_COUNT][16] = {}, obst[2][DRV_MAXO] = {}, epr[2][16] = {};
  int flags[DRV_NB] = {}, aux[DRV_NB] = {}, envi[EI_COUNT] = {};
  for (int i = 0; i < S.A; ++i) {
    const dynenv_car_state_t& c = st->cars[i];
    body[BF_PX][i] = c.px; body[BF_PY][i] = c.py; body[BF_VX][i] = c.vx; body[BF_VY][i] = c.vy;
    body[BF_ANG][i] = c.angle; body[BF_W][i] = c.w;
    carx[CF_DIRX][i] = c.dirx; carx[CF_DIRY][i] = c.diry; carx[CF_PREVX][i] = c.prevx; carx[CF_PREVY][i] = c.prevy;
    carx[CF_GOALX][i] = c.goalx; carx[CF_GOALY][i] = c.goaly;
    flags[i] = CARF_PACK(c.type & 3, c.team & 3, c.finished & 1, c.crashed & 1, c.fric & 1, c.lane_pos & 7);
  }
  for (int i = 0; i < st->n_peds; ++i) {
    const dynenv_ped_state_t& p = st->peds[i];
    int l = DRV_SLOT_PED + i;
    body[BF_PX][l] = p.px; body[BF_PY][l] = p.py; body[BF_VX][l] = p.vx; body[BF_VY][l] = p.vy;
    flags[l] = PEDF_PACK(p.road & 1, p.side & 1, p.dead & 1, p.crossing & 1, p.begin_crossing & 1, p.speed & 15);
    aux[l] = p.moving;
  }
  for (int i = 0; i < st->n_obst; ++i) { obst[0][i] = st->obst_x[i]; obst[1][i] = st->obst_y[i]; }
  for (int i = 0; i < DYNENV_MAX_CARS; ++i) { epr[0][i] = st->episode_r[i]; epr[1][i] = st->episode_pos_r[i]; }
  envi[EI_ELAPSED] = st->elapsed; envi[EI_ALLFIN] = st->all_finished; envi[EI_NPED] = st->n_peds;
  envi[EI_NOBST] = st->n_obst; envi[EI_EPISODE] = st->episode; envi[EI_OCC] = 0; envi[EI_ERR] = 0;
  for (int f = 0; f < BF_COUNT; ++f)
    HIP_OK(hipMemcpy(S.body + (size_t)f * E * DRV_NB + (size_t)env * DRV_NB, body[f], sizeof(double) * DRV_NB, hipMemcpyHostToDevice));
  for (int f = 0; f < CF_COUNT; ++f)
    HIP_OK(hipMemcpy(S.carx + (size_t)f * E * 16 + (size_t)env * 16, carx[f], sizeof(double) * 16, hipMemcpyHostToDevice));
  for (int f = 0; f < 2; ++f) {
    HIP_OK(hipMemcpy(S.obst + (size_t)f * E * DRV_MAXO + (size_t)env * DRV_MAXO, obst[f], sizeof(double) * DRV_MAXO, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(S.epr + (size_t)f * E * 16 + (size_t)env * 16, epr[f], sizeof(double) * 16, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMemcpy(S.flags + (size_t)env * DRV_NB, flags, sizeof(flags), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(S.aux + (size_t)env * DRV_NB, aux, sizeof(aux), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(S.envi + (size_t)env * EI_COUNT, envi, sizeof(envi), hipMemcpyHostToDevice));
  HIP_OK(hipMemset(S.lastcand + (size_t)env * 64, 0xFF, 64 * sizeof(int)));  // -1: quiescent shortcut state unknown
  return DYNENV_OK;
}

// ------------------------------------------------------------------------------------------------
// ragged -> padded arranger (arranger_kernels.hip); replaces InOutArranger (reference models/models.py:208-274)
// ------------------------------------------------------------------------------------------------
static int arr_types(const dynenv_arr_type_t* types, int32_t n_types, int32_t D, ArrTypes& out) {
  if (!types || n_types < 1 || n_types > DYNENV_ARR_MAX_TYPES) return fail(DYNENV_ERR_ARG, "arranger: 1..4 object types");
  out.n = n_types;
  for (int i = 0; i < n_types; ++i) {
    const dynenv_arr_type_t& t = types[i];
    if (t.feat < 1 || t.cap < 0 || t.offset < 0 || t.offset + t.feat * t.cap > D) return fail(DYNENV_ERR_ARG, "arranger: block outside the observation row");
    if (t.count_mode == DYNENV_ARR_COUNT_ROW && (t.count_index < 0 || t.count_index >= D)) return fail(DYNENV_ERR_ARG, "arranger: count_index outside the observation row");
    if (t.count_mode < 0 || t.count_mode > DYNENV_ARR_COUNT_ROW) return fail(DYNENV_ERR_ARG, "arranger: unknown count_mode");
    out.t[i] = t;
  }
  return DYNENV_OK;
}
static int arr_have_device() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DYNENV_ERR_NO_DEVICE, "no HIP device visible: libdynenv_hip has no CPU fallback");
  return DYNENV_OK;
}

int64_t dynenv_arrange_scratch_ints(int32_t E, int32_t T, int32_t A, int32_t n_types) {
  const int64_t TP = (int64_t)E * T * A, nBlocks = (TP + ARR_BLOCK - 1) / ARR_BLOCK;
  return (n_types + 1) * nBlocks + 2 * (DYNENV_ARR_MAX_TYPES + 1) /* int64 results */;
}

int dynenv_arrange_plan(const float* obs_dev, int32_t E, int32_t T, int32_t A, int32_t D, const dynenv_arr_type_t* types,
                        int32_t n_types, const int32_t* count_env_dev, int32_t* counts_dev, int32_t* obj_counts_dev,
                        int32_t* base_dev, int32_t* scratch_dev, dynenv_arr_plan_t* plan_host, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !counts_dev || !obj_counts_dev || !base_dev || !scratch_dev || !plan_host) return fail(DYNENV_ERR_ARG, "null argument");
  if (E < 1 || T < 1 || A < 1 || D < 1 || (int64_t)E * T * A > (int64_t)1 << 30) return fail(DYNENV_ERR_ARG, "arranger: bad shape");
  ArrTypes ty;
  if (int rc = arr_types(types, n_types, D, ty)) return rc;
  for (int i = 0; i < n_types; ++i)
    if (types[i].count_mode == DYNENV_ARR_COUNT_ENV && !count_env_dev) return fail(DYNENV_ERR_ARG, "arranger: count_env_dev missing");
  hipStream_t st = (hipStream_t)stream;
  const int TP = E * T * A, nBlocks = (TP + ARR_BLOCK - 1) / ARR_BLOCK;
  int32_t* blockTot = scratch_dev;
  // int64 results live behind the block totals, 8-byte aligned
  size_t off = (size_t)(n_types + 1) * nBlocks;
  off = (off + 1) & ~(size_t)1;
  int64_t* result = reinterpret_cast<int64_t*>(scratch_dev + off);
  hipLaunchKernelGGL(arr_count_kernel, dim3(nBlocks), dim3(ARR_BLOCK), 0, st, obs_dev, E, T, A, D, ty, count_env_dev, counts_dev,
                     obj_counts_dev, base_dev, blockTot);
  hipLaunchKernelGGL(arr_scan_blocks_kernel, dim3(1), dim3(ARR_BLOCK), 0, st, blockTot, nBlocks, n_types, result);
  hipLaunchKernelGGL(arr_add_offsets_kernel, dim3(nBlocks), dim3(ARR_BLOCK), 0, st, base_dev, blockTot, TP, nBlocks, n_types);
  HIP_OK(hipGetLastError());
  int64_t res[DYNENV_ARR_MAX_TYPES + 1];
  HIP_OK(hipMemcpyAsync(res, result, sizeof(int64_t) * (n_types + 1), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  plan_host->n_types = n_types; plan_host->n_time = T; plan_host->n_players = E * A; plan_host->max_count = (int32_t)res[n_types];
  for (int i = 0; i < DYNENV_ARR_MAX_TYPES; ++i) plan_host->total[i] = i < n_types ? res[i] : 0;
  return DYNENV_OK;
}

int dynenv_arrange_gather(const float* obs_dev, int32_t E, int32_t T, int32_t A, int32_t D, const dynenv_arr_type_t* types,
                          int32_t n_types, const int32_t* counts_dev, const int32_t* base_dev, int32_t max_count,
                          float* const* inputs_dev, int32_t* const* slot_dev, uint8_t* mask_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !counts_dev || !base_dev) return fail(DYNENV_ERR_ARG, "null argument");
  ArrTypes ty;
  if (int rc = arr_types(types, n_types, D, ty)) return rc;
  int capSum = 0;
  for (int i = 0; i < n_types; ++i) capSum += types[i].cap;
  const int jSpan = capSum > max_count ? capSum : (max_count > 0 ? max_count : 1);
  const long long total = (long long)E * T * A * jSpan;
  float* in[DYNENV_ARR_MAX_TYPES] = {nullptr, nullptr, nullptr, nullptr};
  int32_t* sl[DYNENV_ARR_MAX_TYPES] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n_types; ++i) { if (inputs_dev) in[i] = inputs_dev[i]; if (slot_dev) sl[i] = slot_dev[i]; }
  hipLaunchKernelGGL(arr_gather_kernel, dim3((unsigned)((total + ARR_BLOCK - 1) / ARR_BLOCK)), dim3(ARR_BLOCK), 0, (hipStream_t)stream,
                     obs_dev, E, T, A, D, ty, counts_dev, base_dev, max_count, jSpan, in[0], in[1], in[2], in[3], sl[0], sl[1], sl[2], sl[3], mask_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_arrange_scatter(const float* emb_dev, const int32_t* slot_dev, int64_t N, int32_t F, float* padded_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (N == 0) return DYNENV_OK;
  if (!emb_dev || !slot_dev || !padded_dev || N < 0 || F < 1) return fail(DYNENV_ERR_ARG, "bad argument");
  const long long total = (long long)N * F;
  hipLaunchKernelGGL(arr_scatter_kernel, dim3((unsigned)((total + ARR_BLOCK - 1) / ARR_BLOCK)), dim3(ARR_BLOCK), 0, (hipStream_t)stream,
                     emb_dev, slot_dev, (long long)N, F, padded_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_arrange_pad(const float* const* emb_dev, const int32_t* counts_dev, const int32_t* base_dev, int32_t n_types,
                       int32_t T, int32_t P, int32_t max_count, int32_t F, float* padded_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!emb_dev || !counts_dev || !base_dev || !padded_dev) return fail(DYNENV_ERR_ARG, "null argument");
  if (n_types < 1 || n_types > DYNENV_ARR_MAX_TYPES || T < 1 || P < 1 || max_count < 0 || F < 4 || (F & 3)) return fail(DYNENV_ERR_ARG, "arranger: bad shape (F must be a multiple of 4)");
  if (max_count == 0) return DYNENV_OK;
  const float* e[DYNENV_ARR_MAX_TYPES] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n_types; ++i) e[i] = emb_dev[i];
  const long long perRow = (long long)P * (F / 4);
  if (perRow > 0x7fffffffLL || T > 65535) return fail(DYNENV_ERR_ARG, "arranger: padded tensor too large for one launch");
  hipLaunchKernelGGL(arr_pad_cols_kernel, dim3((unsigned)((perRow + ARR_BLOCK - 1) / ARR_BLOCK), (unsigned)T), dim3(ARR_BLOCK), 0, (hipStream_t)stream,
                     e[0], e[1], e[2], e[3], counts_dev, base_dev, n_types, T, P, max_count, F / 4, reinterpret_cast<float4*>(padded_dev));
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_obs_pack(const float* obs_dev, int64_t n_env_time, int32_t A, int32_t D, int32_t split, float* packed_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !packed_dev || n_env_time < 0 || A < 1 || D < 1 || split < 0 || split > D) return fail(DYNENV_ERR_ARG, "bad argument");
  const long long total = (long long)n_env_time * ((long long)A * split + (D - split));
  if (total == 0) return DYNENV_OK;
  hipLaunchKernelGGL(obs_pack_kernel, dim3((unsigned)((total + ARR_BLOCK - 1) / ARR_BLOCK)), dim3(ARR_BLOCK), 0, (hipStream_t)stream,
                     obs_dev, (long long)n_env_time, A, D, split, packed_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}
int dynenv_obs_unpack(const float* packed_dev, int64_t n_env_time, int32_t A, int32_t D, int32_t split, float* obs_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !packed_dev || n_env_time < 0 || A < 1 || D < 1 || split < 0 || split > D) return fail(DYNENV_ERR_ARG, "bad argument");
  const long long total = (long long)n_env_time * A * D;
  if (total == 0) return DYNENV_OK;
  hipLaunchKernelGGL(obs_unpack_kernel, dim3((unsigned)((total + ARR_BLOCK - 1) / ARR_BLOCK)), dim3(ARR_BLOCK), 0, (hipStream_t)stream,
                     packed_dev, (long long)n_env_time, A, D, split, obs_dev, 0ll);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}
int dynenv_obs_unpack_ranks(const float* packed_dev, int64_t src_stride_floats, int32_t n_ranks, int64_t n_env_time, int32_t A,
                            int32_t D, int32_t split, float* obs_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !packed_dev || n_env_time < 0 || A < 1 || D < 1 || split < 0 || split > D || n_ranks < 1 || n_ranks > 65535 || src_stride_floats < 0)
    return fail(DYNENV_ERR_ARG, "bad argument");
  const long long total = (long long)n_env_time * A * D;
  if (total == 0) return DYNENV_OK;
  hipLaunchKernelGGL(obs_unpack_kernel, dim3((unsigned)((total + ARR_BLOCK - 1) / ARR_BLOCK), (unsigned)n_ranks), dim3(ARR_BLOCK), 0,
                     (hipStream_t)stream, packed_dev, (long long)n_env_time, A, D, split, obs_dev, (long long)src_stride_floats);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

int dynenv_obs_pack_peers(const float* obs_dev, int64_t n_env_time, int32_t A, int32_t D, float* packed_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !packed_dev || n_env_time < 0 || A < 1 || D < PEER_SELF + (A - 1) * PEER_COLS) return fail(DYNENV_ERR_ARG, "bad argument");
  const long long total = (long long)n_env_time * (A * PEER_SELF + (D - PEER_SELF - (A - 1) * PEER_COLS));
  if (total == 0) return DYNENV_OK;
  hipLaunchKernelGGL(obs_pack_peers_kernel, dim3((unsigned)((total + ARR_BLOCK - 1) / ARR_BLOCK)), dim3(ARR_BLOCK), 0, (hipStream_t)stream,
                     obs_dev, (long long)n_env_time, A, D, packed_dev);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}
int dynenv_obs_unpack_peers_ranks(const float* packed_dev, int64_t src_stride_floats, int32_t n_ranks, int64_t n_env_time, int32_t A,
                                  int32_t D, float* obs_dev, void* stream) {
  if (int rc = arr_have_device()) return rc;
  if (!obs_dev || !packed_dev || n_env_time < 0 || A < 1 || D < PEER_SELF + (A - 1) * PEER_COLS || n_ranks < 1 || n_ranks > 65535 ||
      src_stride_floats < 0)
    return fail(DYNENV_ERR_ARG, "bad argument");
  const long long total = (long long)n_env_time * A * D;
  if (total == 0) return DYNENV_OK;
  const bool vec = (D % 4) == 0 && ((uintptr_t)obs_dev % 16) == 0;
  if (vec && A <= PEER_ROWS_MAXA && D - (PEER_SELF + (A - 1) * PEER_COLS) <= 160) {
    // row job (see obs_unpack_peers_rows_kernel)
    int rowsPerBlock = (int)((n_env_time * (long long)n_ranks + 4095) / 4096);  // ~4096 blocks: the chip several times over
    if (rowsPerBlock < 1) rowsPerBlock = 1;
    if (rowsPerBlock > 32) rowsPerBlock = 32;
    const dim3 rgrid((unsigned)((n_env_time + rowsPerBlock - 1) / rowsPerBlock), (unsigned)n_ranks);
    hipLaunchKernelGGL(obs_unpack_peers_rows_kernel, rgrid, dim3(ARR_BLOCK), 0, (hipStream_t)stream, packed_dev, (long long)n_env_time, A, D,
                       obs_dev, (long long)src_stride_floats, rowsPerBlock);
    HIP_OK(hipGetLastError());
    return DYNENV_OK;
  }
  const long long threads = vec ? total / 4 : total;
  const dim3 grid((unsigned)((threads + ARR_BLOCK - 1) / ARR_BLOCK), (unsigned)n_ranks);
  if (vec)
    hipLaunchKernelGGL(obs_unpack_peers4_kernel, grid, dim3(ARR_BLOCK), 0, (hipStream_t)stream, packed_dev, (long long)n_env_time, A, D,
                       obs_dev, (long long)src_stride_floats);
  else
    hipLaunchKernelGGL(obs_unpack_peers1_kernel, grid, dim3(ARR_BLOCK), 0, (hipStream_t)stream, packed_dev, (long long)n_env_time, A, D,
                       obs_dev, (long long)src_stride_floats);
  HIP_OK(hipGetLastError());
  return DYNENV_OK;
}

// ------------------------------------------------------------------------------------------------
// exact checkpoint: every device array of the handle, bit for bit
// ------------------------------------------------------------------------------------------------
struct CkptHeader {
  char magic[8];  // "DYNCKPT2"
  int32_t abi_version, n_arrays;
  dynenv_cfg_t cfg;
  uint64_t payload_bytes;
};
static size_t ckpt_payload(const dynenv* h) {
  size_t n = 0;
  for (size_t b : h->alloc_bytes) n += b;
  return n;
}
size_t dynenv_checkpoint_size(const dynenv_t* h) { return h ? sizeof(CkptHeader) + ckpt_payload(h) : 0; }

int dynenv_checkpoint_save(dynenv_t* h, void* buf_host, size_t nbytes) {
  if (!h || !buf_host) return fail(DYNENV_ERR_ARG, "null argument");
  if (nbytes < dynenv_checkpoint_size(h)) return fail(DYNENV_ERR_ARG, "checkpoint buffer too small");
  ON_DEVICE(h);
  HIP_OK(hipDeviceSynchronize());
  CkptHeader hd;
  memset(&hd, 0, sizeof(hd));
  memcpy(hd.magic, "DYNCKPT2", 8);
  hd.abi_version = DYNENV_ABI_VERSION; hd.n_arrays = (int32_t)h->allocs.size(); hd.cfg = h->cfg;
  hd.cfg.seed = h->robocup ? h->R.seed : h->S.seed;
  hd.payload_bytes = ckpt_payload(h);
  char* out = (char*)buf_host;
  memcpy(out, &hd, sizeof(hd));
  out += sizeof(hd);
  for (size_t i = 0; i < h->allocs.size(); ++i) {
    HIP_OK(hipMemcpy(out, h->allocs[i], h->alloc_bytes[i], hipMemcpyDeviceToHost));
    out += h->alloc_bytes[i];
  }
  return DYNENV_OK;
}

int dynenv_checkpoint_load(dynenv_t* h, const void* buf_host, size_t nbytes) {
  if (!h || !buf_host) return fail(DYNENV_ERR_ARG, "null argument");
  ON_DEVICE(h);
  if (nbytes < sizeof(CkptHeader)) return fail(DYNENV_ERR_ARG, "not a checkpoint");
  CkptHeader hd;
  memcpy(&hd, buf_host, sizeof(hd));
  // (ABI 3 changed no array and no layout: checkpoints written by an ABI 2 library load)
  if (memcmp(hd.magic, "DYNCKPT2", 8) != 0 || (hd.abi_version != DYNENV_ABI_VERSION && hd.abi_version != 2)) return fail(DYNENV_ERR_ARG, "not a checkpoint of this ABI version");
  const dynenv_cfg_t& a = hd.cfg; const dynenv_cfg_t& b = h->cfg;
  if (a.env_type != b.env_type || a.num_envs != b.num_envs || a.n_players != b.n_players || a.obs_type != b.obs_type ||
      a.noise_type != b.noise_type || a.noise_magnitude != b.noise_magnitude || a.env_id_offset != b.env_id_offset || a.flags != b.flags)
    return fail(DYNENV_ERR_ARG, "checkpoint was taken from a differently configured handle");
  if (hd.n_arrays != (int32_t)h->allocs.size() || hd.payload_bytes != ckpt_payload(h) || nbytes < sizeof(hd) + hd.payload_bytes)
    return fail(DYNENV_ERR_ARG, "checkpoint layout does not match this build");
  HIP_OK(hipDeviceSynchronize());
  const char* in = (const char*)buf_host + sizeof(hd);
  // ABI 2 kept "an invalid action was seen" and, for Partial observations, "rows beyond the layout's capacity were dropped" in ONE bit
  // (1) of the per-environment error word, which is part of the checkpointed array; ABI 3 gave the second its own bit 3.  A set bit 1
  // of an ABI 2 blob of a Partial handle may mean either: it is loaded as both (nothing that was reported goes unreported).
  const bool partialHandle = h->robocup ? h->R.obs_type == DYNENV_OBS_PARTIAL : h->partial;
  const void* errArray = h->robocup ? (const void*)h->R.envi : (const void*)h->S.envi;
  const size_t errStride = h->robocup ? RE_COUNT : EI_COUNT, errWord = h->robocup ? RE_ERR : EI_ERR;
  for (size_t i = 0; i < h->allocs.size(); ++i) {
    if (hd.abi_version == 2 && partialHandle && h->allocs[i] == errArray) {
      std::vector<int> w(h->alloc_bytes[i] / sizeof(int));
      memcpy(w.data(), in, w.size() * sizeof(int));
      for (size_t e = 0; (e + 1) * errStride <= w.size(); ++e) if (w[e * errStride + errWord] & 2) w[e * errStride + errWord] |= 8;
      HIP_OK(hipMemcpy(h->allocs[i], w.data(), h->alloc_bytes[i], hipMemcpyHostToDevice));
    } else {
      HIP_OK(hipMemcpy(h->allocs[i], in, h->alloc_bytes[i], hipMemcpyHostToDevice));
    }
    in += h->alloc_bytes[i];
  }
  h->cfg.seed = a.seed;
  if (h->robocup) h->R.seed = a.seed; else h->S.seed = a.seed;
  // the scheduler's lists describe the timing of the steps this handle ran, not the state just restored: start them over
  // (a list that keeps ids from before the restore could be appended to without having been cleared - ADVICE r3)
  if (!h->robocup) {
    if (iso_reset(h)) return DYNENV_ERR_HIP;
    h->S.iso_on = h->iso_cfg; h->iso_last_invalid = 0; h->iso_paused_until = 0; h->steps = 0;
    if (h->iso_seen) *h->iso_seen = 0;
  }
  return DYNENV_OK;
}

int dynenv_math_selftest(const double* x, const double* y, int32_t n, double* out, int32_t device_id) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DYNENV_ERR_NO_DEVICE, "no HIP device visible");
  HIP_OK(hipSetDevice(device_id));
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HIP_OK(hipMalloc((void**)&dx, sizeof(double) * n));
  HIP_OK(hipMalloc((void**)&dy, sizeof(double) * n));
  HIP_OK(hipMalloc((void**)&dout, sizeof(double) * n * 5));
  HIP_OK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dy, y, sizeof(double) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(math_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n, dout);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpy(out, dout, sizeof(double) * n * 5, hipMemcpyDeviceToHost));
  hipFree(dx); hipFree(dy); hipFree(dout);
  return DYNENV_OK;
}

}  // extern "C"
