// driving_partial.hip — Driving PARTIAL observation + noise on gfx950 (BASELINE configs[3]).
//
// Replaces DrivingEnvironment.getAgentVision (reference DrivingEnvironment.py:750-977 with cutils.py isSeenInRadius /
// doesInteractPoly / addNoiseRect / addNoiseLane and Road.getCarLaneDistances) for all agents of all environments.
// One wavefront per environment, looping over the env's agents; for one agent the lanes are the OBJECTS it might see:
//   lanes 0..9 cars (the agent's own lane builds the self row), 10..29 pedestrians, 30..49 obstacles, 50..53 buildings,
//   54..59 the six lane rows; the ten random false-positive trials are evaluated in a later phase, by lanes that are idle there.
// Row positions inside the ragged lists (which also index the noise streams) come from wave ballots + popcounts.
// Mirrors oracle/driving_partial.c operation by operation (bit-identical output); see that file for the RNG keying.
#include "driving_dev.h"

#include "driving_host.h" /* PV_CAP_*, PV_LIM_*, PV_DIM, PV_OFF_*: the dense layout, shared with the host code */
#define SIGHT_NONE 0
#define SIGHT_NORMAL 3
#define SIGHT_MISCLASS 4
#define INTER_NONE 0
#define INTER_NEARBY 1
#define INTER_OCCLUDE 2
#define PV_ANGLE_NOISE (DM_PI / 180.0)

struct PvBlocker {  // an object as `elem2` of doesInteractPoly: view-blocking interval + the three decisive corners
  double posx, posy, angle2, minA, maxA, p1x, p1y, p2x, p2y, pmx, pmy;
  int seen, extreme;
};
struct PvPedSlot {  // phase 4: a visible pedestrian's inputs and running maxima, overlaid on its (unused) blocker entry
  double posx, posy, angle1;
  int seen, carPed, obsPed;
};
static_assert(sizeof(PvPedSlot) <= sizeof(PvBlocker), "pedestrian slot must fit a blocker entry");
static_assert(10 * sizeof(PvBlocker) >= (64 + 34) * sizeof(int), "index lists (ints 0..60) and the corner pool (64..97) must fit the entries of lanes 54..63");
struct PvLds {
  double px[DRV_NB], py[DRV_NB], ang[DRV_NB];
  double ox[DRV_MAXO], oy[DRV_MAXO];
  int flags[DRV_NB];
  PvBlocker blk[64];  // indexed by lane (cars 0..9, obstacles 30..49, buildings 50..53)
  float row[PV_DIM + 3];  // the agent's dense row is assembled here, then streamed out with coalesced stores
  alignas(16) double atanTab[30];  // dev_atan2_t's table
};
__shared__ PvLds g_P;
#ifdef DRV_PROFILE
#define PV_PROF(...) __VA_ARGS__
__device__ unsigned long long g_pvprof[4096 * 16];  // per environment (atomics on ONE set of counters serialise in L2 and the queue's back-pressure lands in the next phase), cycles summed over its agent passes of the run: detection | lane rows | blockers | buildings + list positions | pedestrian pairs | noise | random FPs | assembly + row out; [8] passes
#else
#define PV_PROF(...)
#endif

DE_DEV dm_u32x4 pv_rng(const DrvState& S, uint32_t genv, uint32_t episode, int elapsed, int agent, int kind, int index, int block) {
  uint32_t entity = (uint32_t)agent | ((uint32_t)kind << 4) | ((uint32_t)index << 8) | ((uint32_t)block << 16);
  return dm_env_rng(S.seed, genv, episode, DM_RNG_OBS_NOISE, entity, (uint32_t)elapsed);
}
DE_DEV double pv_normalize(double pt, double nf, double mean) { return ((pt * nf) - mean) * 2.0 * 1.0; }
DE_DEV V2 pv_rotated(V2 v, double a) {
  const DevSC sc = dev_sincos_v(a);
  return v2(v.x * sc.c - v.y * sc.s, v.x * sc.s + v.y * sc.c);
}
DE_DEV double pv_lensq(V2 v) { return v.x * v.x + v.y * v.y; }

// cutils.doesInteractPoly :643-696 with the blocker's interval precomputed (getViewBlockAngle :626-640).  Straight-line: every
// comparison is evaluated and the verdict selected - the lanes of a wave disagree on all of these conditions, so a branching form
// executes every arm anyway, behind an exec-mask region each (4 calls were 445 instructions and 34 branches).
DE_DEV int pv_interact(int seen1, V2 point1, double angle1, const PvBlocker& b, double radius) {
  const bool valid = seen1 != SIGHT_NONE && b.seen != SIGHT_NONE;
  const V2 point2 = v2(b.posx, b.posy);
  const bool nearby = radius > 0.0 && pv_lensq(vsub(point2, point1)) < radius;
  double pAngle = angle1 - b.angle2;
  const double up = pAngle - DM_TWO_PI, down = pAngle + DM_TWO_PI;
  pAngle = pAngle > DM_PI ? up : (pAngle < -DM_PI ? down : pAngle);
  const bool inside = pAngle > b.minA && pAngle < b.maxA;
  const V2 p1 = v2(b.p1x, b.p1y), p2 = v2(b.p2x, b.p2y), pm = v2(b.pmx, b.pmy);
  const V2 d1 = vsub(point1, p1), dm = vsub(point1, pm);
  const bool c21 = vcross(vsub(p2, p1), d1) < 0.0;
  const bool c2m = vcross(vsub(p2, pm), dm) < 0.0, cm1 = vcross(vsub(pm, p1), d1) < 0.0;
  const bool occl = inside && (b.extreme ? c21 : (c2m && cm1));
  return !valid ? INTER_NONE : (occl ? INTER_OCCLUDE : (nearby ? INTER_NEARBY : INTER_NONE));
}

// The observation of every agent of environment e.  `in` carries the per-lane state (from HBM in the stand-alone kernel,
// straight from the step kernel's LDS tile in the fused call); L is this wave's scratch tile.
// Returns the first agent it did NOT do: aEnd, or less when `deadline` (a s_memtime value, 0 = none) had passed before a pass.
DE_DEV int pv_env(const DrvState& S, PvLds& L, const int e, const int lane, const int nPed, const int nObst, const int elapsed,
                  const uint32_t episode, const PvIn& in, const int noiseType, const double magn, float* __restrict__ obs,
                  const int aBegin, const int aEnd, const unsigned long long deadline = 0ull) {
  const int A = S.A;
  int* envi = S.envi + (size_t)e * EI_COUNT;
  const uint32_t genv = (uint32_t)(S.env_id_offset + e);
  if (lane < DRV_NB) { L.px[lane] = in.px; L.py[lane] = in.py; L.ang[lane] = in.ang; L.flags[lane] = in.flags; }
  if (lane < DRV_MAXO) { L.ox[lane] = in.ox; L.oy[lane] = in.oy; }
  DEV_ATAN_TAB_INIT(L.atanTab, lane);
  const int atanTab = dev_lds_addr(L.atanTab);
  __syncthreads();
  const double randBase = 0.01 * magn;
  const double maxVis0 = (DRV_W * 0.4) * (DRV_W * 0.4), maxVis1 = (DRV_W * 0.6) * (DRV_W * 0.6);
  int overflow = 0;
  // lane role (fixed over agents)
  const bool isCarLane = lane < A;
  const bool isPedLane = lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed;
  const bool isObsLane = lane >= DRV_SLOT_OBST && lane < DRV_SLOT_OBST + nObst;
  const bool isBldLane = lane >= DRV_SLOT_BLD && lane < DRV_SLOT_BLD + 4;
  const bool isLaneRow = lane >= 54 && lane < 60;
  const uint64_t carLanes = wave_ballot(isCarLane), pedLanes = wave_ballot(isPedLane), obsLanes = wave_ballot(isObsLane);
  (void)carLanes;

  int a = aBegin;
#pragma unroll 1
  for (; a < aEnd; ++a) {
    if (deadline != 0ull && __builtin_amdgcn_s_memtime() >= deadline) break;
PV_PROF(const unsigned long long VT0 = __builtin_amdgcn_s_memtime();)
    float* __restrict__ grow = obs + ((size_t)e * A + a) * PV_DIM;
    float* row = L.row;
    for (int i = lane; i < PV_DIM; i += DE_WAVE) row[i] = 0.0f;
    const V2 P = v2(L.px[a], L.py[a]);
    const double ang = L.ang[a];
    const int typeA = L.flags[a] & 3;

    // ---- phase 1: detection of my object (isSeenInRadius) --------------------------------------------------
    int seen = SIGHT_NONE;
    V2 pos = v2(0.0, 0.0);
    double dc = 0.0, ds = 0.0, dw = 0.0, dh = 0.0;
    int dfin = 0;
    bool hasCorners = false;
    // the four corners as scalars, not an array: an array written by a (later unrolled) loop and read through index selects
    // ends up in scratch memory - the selects of loads become loads through a selected pointer before the loop is unrolled
    V2 cor0 = v2(0.0, 0.0), cor1 = cor0, cor2 = cor0, cor3 = cor0;
    const bool isSelf = isCarLane && lane == a;
    // one sincos call serves every lane of this pass: self -> its heading, the other objects -> heading relative to the
    // agent, lane rows -> road direction relative to the agent, and lane 63 -> the rotation by -ang that every object's
    // position needs (broadcast below)
    const bool isObjLane = isCarLane || isPedLane || isObsLane || isBldLane;
    V2 point = v2(0.0, 0.0);
    double hx = 0.0, hy = 0.0, oangle = 0.0, maxD = maxVis0, distD = maxVis1;
    if (isObjLane) {
      if (isCarLane) {
        const int f = L.flags[lane];
        point = v2(L.px[lane], L.py[lane]); oangle = L.ang[lane];
        hx = C.carHx[f & 3]; hy = C.carHy[f & 3];  // Car.points (h, w): h = length, w = width
        dw = hy; dh = hx; dfin = CF_FIN(f);
        hasCorners = true;
      } else if (isPedLane) {
        point = v2(L.px[lane], L.py[lane]);
      } else if (isObsLane) {
        point = v2(L.ox[lane - DRV_SLOT_OBST], L.oy[lane - DRV_SLOT_OBST]); hx = 10.0; hy = 10.0; dw = 10.0; dh = 10.0; hasCorners = true;
      } else {
        const int k = lane - DRV_SLOT_BLD;
        point = v2((k & 2) ? 1385.0 : 365.0, (k & 1) ? 800.0 : 200.0); hx = 400.0; hy = 225.0; hasCorners = true;
        maxD = 20000000.0; distD = 20000000.0;
      }
    }
    const int lrow = lane - 54, lroad = lrow < 4 ? 0 : 1;
    double sarg = -ang;  // lane 63 (and idle lanes)
    if (isSelf) sarg = ang;
    else if (isObjLane) sarg = oangle - ang;
    else if (isLaneRow) sarg = C.roads[lroad].dirAngle - ang;
    const DevSC sc1 = dev_sincos_v(sarg);
    const double rotC = bcast_d(sc1.c, 63), rotS = bcast_d(sc1.s, 63);
    if (isObjLane) {
      if (isSelf) {  // selfDet :755-756: absolute position, own corners, never filtered
        seen = SIGHT_NORMAL; pos = P; dc = sc1.c; ds = sc1.s;
        cor0 = vadd(v2(hx, hy), P); cor1 = vadd(v2(-hx, hy), P); cor2 = vadd(v2(-hx, -hy), P); cor3 = vadd(v2(hx, -hy), P);
      } else {
        const V2 trPt = vsub(point, P);
        const double dist = pv_lensq(trPt);
        if (dist <= maxD) {
          seen = SIGHT_NORMAL;  // Distant is unreachable: maxDist < distantDist (quirk C17)
          if (!(dist <= distD)) seen = 2;
          if (hasCorners) {
#define PV_COR(X, Y) vadd(vsub(vadd(v2((X), (Y)), point), point), trPt)
            cor0 = PV_COR(hx, hy); cor1 = PV_COR(-hx, hy); cor2 = PV_COR(-hx, -hy); cor3 = PV_COR(hx, -hy);
#undef PV_COR
          }
          pos = v2(trPt.x * rotC - trPt.y * rotS, trPt.x * rotS + trPt.y * rotC);  // trPt.rotated(-ang)
          dc = sc1.c; ds = sc1.s;
        }
      }
    }
PV_PROF(const unsigned long long VT1 = __builtin_amdgcn_s_memtime();)
    // ---- lane rows: Road.getCarLaneDistances :36-71 (rows 0..3 road 0, rows 4..5 road 1) ---------------------
    double ldist = 0.0, lc = 0.0, ls = 0.0, ltype = 0.0;
    int lseen = SIGHT_NONE;
    if (isLaneRow) {
      const int rowi = lrow;
      const int r = lroad;
      const int n = r ? 1 : 2;
      const int i = (rowi - (r ? 4 : 0)) - n;
      const V2 pt = vsub(P, C.roads[r].p0);
      const double dist = vcross(C.roads[r].dir, pt) / C.roads[r].width;
      if (!(dm_abs(dist) > 10.0)) {
        double cc = sc1.c, ss = sc1.s, distMult = 1.0, typeMult = 1.0;
        if (cc >= 0.0) { typeMult = -1.0; cc *= -1.0; ss *= -1.0; distMult = -1.0; }
        lseen = SIGHT_NORMAL;
        ldist = ((dist + 0.5) + (double)i) * C.roads[r].width * 0.1 * distMult;
        lc = cc; ls = ss;
        ltype = ((i + n) < n ? 1.0 : -1.0) * typeMult;
      }
    }
PV_PROF(const unsigned long long VT2 = __builtin_amdgcn_s_memtime();)
    // ---- phase 2: publish blockers (cars other than self, obstacles, buildings) ------------------------------
    // The four corner angles of every visible blocker: 4 n evaluations of atan2 for n blockers, pooled over the whole wave (item =
    // 4 j + corner) instead of four calls with only the blocker lanes busy - n is ~10 of 64 lanes, so one call replaces four.  The
    // corners travel through the blocker's own table entry (not yet written) and come back as (angle, squared length) in place.
    const double angle1 = (seen != SIGHT_NONE && !isSelf && (isCarLane || isPedLane || isObsLane || isBldLane)) ? dev_atan2_t(pos.y, pos.x, atanTab) : 0.0;
    const bool isBlk = (isCarLane && !isSelf) || isObsLane || isBldLane;
    const bool blkSeen = isBlk && seen != SIGHT_NONE;
    const uint64_t blkMask = wave_ballot(blkSeen);
    const uint64_t below = lanemask_lt();
    if (blkMask) {
      int* pool = reinterpret_cast<int*>(&L.blk[54]) + 64;  // (the lists of phase 4 use ints 0..60 of these entries)
      double* slot = reinterpret_cast<double*>(&L.blk[lane]);
      if (blkSeen) {
        pool[__popcll(blkMask & below)] = lane;
        slot[0] = cor0.x; slot[1] = cor0.y; slot[2] = cor1.x; slot[3] = cor1.y;
        slot[4] = cor2.x; slot[5] = cor2.y; slot[6] = cor3.x; slot[7] = cor3.y;
      }
      __syncthreads();
      const int nItems = 4 * __popcll(blkMask);
      for (int it = lane; it < nItems; it += DE_WAVE) {
        double* c = reinterpret_cast<double*>(&L.blk[pool[it >> 2]]) + 2 * (it & 3);
        const V2 cor = v2(c[0], c[1]);
        c[0] = dev_atan2_t(cor.y, cor.x, atanTab);
        c[1] = pv_lensq(cor);
      }
      __syncthreads();
    }
    if (isBlk) {
      PvBlocker b;
      b.seen = seen; b.posx = pos.x; b.posy = pos.y; b.angle2 = angle1;
      b.minA = b.maxA = 0.0; b.p1x = b.p1y = b.p2x = b.p2y = b.pmx = b.pmy = 0.0; b.extreme = 0;
      if (seen != SIGHT_NONE) {
        const double* slot = reinterpret_cast<const double*>(&L.blk[lane]);
        double ang0 = slot[0] - angle1, ang1 = slot[2] - angle1, ang2 = slot[4] - angle1, ang3 = slot[6] - angle1;
        const double dst0 = slot[1], dst1 = slot[3], dst2 = slot[5], dst3 = slot[7];
#define PV_WRAP(A) do { if ((A) > DM_PI) (A) -= DM_TWO_PI; } while (0)
        PV_WRAP(ang0); PV_WRAP(ang1); PV_WRAP(ang2); PV_WRAP(ang3);
#undef PV_WRAP
#define PV_WRAP(A) do { if ((A) < -DM_PI) (A) += DM_TWO_PI; } while (0)
        PV_WRAP(ang0); PV_WRAP(ang1); PV_WRAP(ang2); PV_WRAP(ang3);
#undef PV_WRAP
        int mn = 0, mx = 0, ci = 0;
        double amin = ang0, amax = ang0, dmin = dst0;
#define PV_STEP(I, A, D) do { if ((A) < amin) { amin = (A); mn = (I); } if ((A) > amax) { amax = (A); mx = (I); } if ((D) < dmin) { dmin = (D); ci = (I); } } while (0)
        PV_STEP(1, ang1, dst1); PV_STEP(2, ang2, dst2); PV_STEP(3, ang3, dst3);
#undef PV_STEP
#define PV_SEL(I) ((I) == 3 ? cor3 : ((I) == 2 ? cor2 : ((I) == 1 ? cor1 : cor0)))
        const V2 p1 = PV_SEL(mn), p2 = PV_SEL(mx), pm = PV_SEL(ci);
#undef PV_SEL
        b.minA = amin; b.maxA = amax; b.p1x = p1.x; b.p1y = p1.y; b.p2x = p2.x; b.p2y = p2.y; b.pmx = pm.x; b.pmy = pm.y;
        b.extreme = (ci == mn || ci == mx) ? 1 : 0;
      }
      L.blk[lane] = b;
    }
    __syncthreads();
PV_PROF(const unsigned long long VT3 = __builtin_amdgcn_s_memtime();)
    // ---- phase 3: building occlusion :782-789, then list positions ------------------------------------------
    const bool isObj = (isCarLane && !isSelf) || isPedLane || isObsLane;
    bool alive = isObj && seen != SIGHT_NONE;
    if (alive) {
      int m = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { int t = pv_interact(seen, pos, angle1, L.blk[DRV_SLOT_BLD + k], 0.0); m = t > m ? t : m; }
      if (m == INTER_OCCLUDE) alive = false;
    }
    const uint64_t aliveMask = wave_ballot(alive);
    const uint64_t carMask = aliveMask & ((1ull << DRV_SLOT_PED) - 1ull);
    const uint64_t pedMask = aliveMask & pedLanes;
    const uint64_t obsMask = aliveMask & obsLanes;
    const uint64_t laneAlive = wave_ballot(isLaneRow && lseen != SIGHT_NONE);
    const int nCars0 = __popcll(carMask), nObst0 = __popcll(obsMask), nPeds0 = __popcll(pedMask);
    const int nLanes0 = __popcll(laneAlive);
    int listIdx = 0;
    if (alive) listIdx = isCarLane ? __popcll(carMask & below) : (isPedLane ? __popcll(pedMask & below) : __popcll(obsMask & below));
    if (isLaneRow) listIdx = __popcll(laneAlive & below);
PV_PROF(const unsigned long long VT4 = __builtin_amdgcn_s_memtime();)
    // ---- phase 4: pedestrian interactions :792-801 ----------------------------------------------------------
    int pedInter = INTER_NONE;
    {
      int carPed = INTER_NONE, obsPed = INTER_NONE;
      // Every visible pedestrian against every visible car and obstacle: one (pedestrian, blocker) pair per lane and
      // round instead of one blocker per iteration with only the pedestrian lanes busy.  The pedestrians' inputs and their
      // two running maxima live in the blocker-table entries of the pedestrian lanes (never blockers themselves), the two
      // index lists in those of lanes 54..63; max is order independent, so the LDS atomics reproduce the loops exactly.
      const int nP = __popcll(pedMask), nC = __popcll(carMask), nB = nC + __popcll(obsMask);
      if (nP > 0 && nB > 0) {
        int* lists = reinterpret_cast<int*>(&L.blk[54]);  // [0..19] pedestrian lanes, [32..60] blocker lanes
        if (alive && isPedLane) {
          PvPedSlot& ps = *reinterpret_cast<PvPedSlot*>(&L.blk[lane]);
          ps.posx = pos.x; ps.posy = pos.y; ps.angle1 = angle1; ps.seen = seen; ps.carPed = INTER_NONE; ps.obsPed = INTER_NONE;
          lists[listIdx] = lane;
        }
        if (alive && isCarLane) lists[32 + listIdx] = lane;
        if (alive && isObsLane) lists[32 + nC + listIdx] = lane;
        __syncthreads();
        const int nPairs = nP * nB;
        const unsigned inv = (65536u + (unsigned)nB - 1u) / (unsigned)nB;  // p / nB == (p * inv) >> 16 for p < 640, nB <= 29
        for (int p = lane; p < nPairs; p += DE_WAVE) {
          const int pi = (int)(((unsigned)p * inv) >> 16), bi = p - pi * nB;
          const int bl = lists[32 + bi];
          PvPedSlot& ps = *reinterpret_cast<PvPedSlot*>(&L.blk[lists[pi]]);
          const int t = pv_interact(ps.seen, v2(ps.posx, ps.posy), ps.angle1, L.blk[bl], 400.0);
          if (t > INTER_NONE) atomicMax(bl < DRV_SLOT_PED ? &ps.carPed : &ps.obsPed, t);
        }
        __syncthreads();
        if (alive && isPedLane) {
          const PvPedSlot& ps = *reinterpret_cast<const PvPedSlot*>(&L.blk[lane]);
          carPed = ps.carPed; obsPed = ps.obsPed;
        }
      }
      // pedInter = max(carPedInter, obsPedInter): Python LIST comparison -> first differing pedestrian decides (C12)
      const uint64_t diff = wave_ballot(alive && isPedLane && carPed != obsPed);
      bool useObs = false;
      if (diff) {
        const int first = __builtin_ctzll(diff);
        useObs = bcast_i(obsPed, first) > bcast_i(carPed, first);
      }
      pedInter = useObs ? obsPed : carPed;
      if (alive && isPedLane && pedInter == INTER_OCCLUDE) seen = SIGHT_NONE;  // filterOcclude (row stays in the list)
    }
PV_PROF(const unsigned long long VT5 = __builtin_amdgcn_s_memtime();)
    // ---- phase 5: noise (addNoiseRect :479-542 on self / cars / pedestrians / obstacles; addNoiseLane :382-413) ---
    // Objects and lane rows live on disjoint lanes: one pair of Philox blocks, one atan2 and one sincos serve both.  The ten
    // random-FP trials of phase 6 need a pair of blocks each as well: a Philox block costs the wave the same whatever the number
    // of lanes that want one, so the trials ride on the first ten lanes that have no noise to draw here (the buildings' lanes and
    // lanes 60..63 never have: at least eight; every invisible or absent object adds one) - trial t on the t-th such lane, so lane
    // order is trial order.  Fewer than ten (a crowded view): the trials stay on lanes 0..9 with a pair of blocks of their own.
    int trial = -1;
    dm_u32x4 fu, fu1;
    fu.v[0] = fu.v[1] = fu.v[2] = fu.v[3] = 0u; fu1 = fu;
    bool trialsPooled;
    {
      const bool objNoise = (alive || isSelf) && seen != SIGHT_NONE, laneNoise = isLaneRow && lseen != SIGHT_NONE;
      const bool noisy = objNoise || laneNoise;
      const uint64_t idleMask = ~wave_ballot(noisy);
      const int idleRank = __popcll(idleMask & below);
      trialsPooled = __popcll(idleMask) >= 10;
      if (trialsPooled) { if (!noisy && idleRank < 10) trial = idleRank; }
      else if (lane < 10) trial = lane;
      const bool trialHere = trialsPooled && trial >= 0;
      const int kind = trialHere ? 5 : (isLaneRow ? 4 : (isSelf ? 0 : (isCarLane ? 1 : (isPedLane ? 2 : 3))));
      const int idx = trialHere ? trial : (isSelf ? 0 : listIdx);
      dm_u32x4 u, u1;
      u.v[0] = u.v[1] = u.v[2] = u.v[3] = 0u; u1 = u;
      double base = 0.0;
      if (noisy || trialHere) {
        u = pv_rng(S, genv, episode, elapsed, a, kind, idx, 0);
        u1 = pv_rng(S, genv, episode, elapsed, a, kind, idx, 1);
      }
      if (trialHere) { fu = u; fu1 = u1; }
      if (noisy) base = dev_atan2_t(objNoise ? ds : ls, objNoise ? dc : lc, atanTab);
      double angArg = base;
      bool applyAngle = false;
      V2 newPos = pos;
      if (objNoise) {
        const bool misClass = (isCarLane && !isSelf) || isObsLane;
        const double maxDist = isPedLane ? maxVis0 : maxVis1;
        const V2 noiseVec = v2((dm_unit(u.v[0]) - 0.5) * magn, (dm_unit(u.v[1]) - 0.5) * magn);
        if (noiseType == 0) {  // NoiseType.RANDOM
          if (dm_unit(u.v[2]) < randBase) {
            seen = SIGHT_NONE;
          } else {
            newPos = vadd(pos, noiseVec);
            const double angleDiff = (dm_unit(u1.v[0]) - 0.5) * magn * PV_ANGLE_NOISE;
            angArg = base + angleDiff;
            applyAngle = true;
          }
        } else {
          const double range = 0.25 + 3.75 * vlen(pos) / maxDist;  // C18
          double multiplier = range;
          if (pedInter == INTER_NEARBY && isPedLane) multiplier = range * 2.0;
          if (seen == 2) multiplier = range * 3.0;
          const V2 np = vadd(pos, vmul(noiseVec, multiplier));
          if (dm_unit(u.v[2]) < randBase * multiplier) {
            seen = SIGHT_NONE;
          } else {
            if (misClass && dm_unit(u.v[3]) < randBase * multiplier / 2.0) seen = SIGHT_MISCLASS;
            const double angleDiff = (dm_unit(u1.v[0]) - 0.5) * magn * PV_ANGLE_NOISE * 0.25;
            angArg = base + angleDiff;
            applyAngle = true;
            newPos = np;
          }
        }
      } else if (laneNoise) {
        const double distNoise = (dm_unit(u.v[0]) - 0.5) * magn, angleDiff = (dm_unit(u.v[1]) - 0.5) * magn;
        if (noiseType == 0) {
          if (dm_unit(u.v[2]) < randBase) lseen = SIGHT_NONE;
          ldist *= distNoise;  // C19
          angArg = base + PV_ANGLE_NOISE * angleDiff;
        } else {
          const double multiplier1 = 0.25 + 3.75 * ldist * ldist / maxVis1;
          if (dm_unit(u.v[2]) < randBase * multiplier1) lseen = SIGHT_NONE;
          ldist += distNoise * multiplier1;
          angArg = base + PV_ANGLE_NOISE * multiplier1 / 5.0 * angleDiff;
        }
      }
      if (objNoise || laneNoise) {
        const DevSC sc = dev_sincos_v(angArg);
        if (applyAngle) { dc = sc.c; ds = sc.s; pos = newPos; }
        if (laneNoise) { lc = sc.c; ls = sc.s; }
      }
    }
PV_PROF(const unsigned long long VT6 = __builtin_amdgcn_s_memtime();)
    // ---- phase 6: random false positives :824-874, one trial per lane (see phase 5 for which lanes) -------------
    int fpClass = -1;
    V2 fpPos = v2(0.0, 0.0);
    double fpc = 0.0, fps = 0.0, fpw = 0.0, fph = 0.0, fpLaneDist = 0.0, fpLaneType = 0.0;
    if (!trialsPooled && trial >= 0) {
      fu = pv_rng(S, genv, episode, elapsed, a, 5, trial, 0);
      fu1 = pv_rng(S, genv, episode, elapsed, a, 5, trial, 1);
    }
    if (trial >= 0) {
      const dm_u32x4 u = fu, u1 = fu1;
      if (dm_unit(u.v[0]) < randBase) {
        fpClass = dm_randint(u.v[1], 0, 5);
        const double d = dm_unit(u.v[2]) * maxVis1;
        const double a1 = dm_unit(u.v[3]) * 2.0 * DM_PI;
        fpPos = pv_rotated(v2(d, 0.0), a1);
        const DevSC sc = dev_sincos_v(dm_unit(u1.v[0]) * 2.0 * DM_PI);
        fpc = sc.c; fps = sc.s;
        if (fpClass <= 1) { fpw = dm_unit(u1.v[1]) * 5.0 + 5.0; fph = dm_unit(u1.v[2]) * 10.0 + 5.0; }
        else if (fpClass == 3) {
          const DevSC lsc = dev_sincos_v((dm_unit(u1.v[1]) - 0.5) * DM_PI * 2.0);
          fpc = lsc.c; fps = lsc.s;
          fpLaneDist = __builtin_floor(dm_unit(u1.v[2]) * DRV_W / 2.0);
          fpLaneType = (double)dm_randint(u1.v[3], -1, 1);
        }
      }
    }
PV_PROF(const unsigned long long VT7 = __builtin_amdgcn_s_memtime();)
    // ---- phase 7: list assembly (misclassification swap :816-821, FP pedestrians near cars :877-882, final filter) ---
    // A lane can hold a real object AND a random-FP trial (lanes 0..9, in a crowded view): two independent code paths.
    const bool realCar = alive && isCarLane, realObs = alive && isObsLane, realPed = alive && isPedLane;
    const uint64_t misCarMask = wave_ballot(realCar && seen == SIGHT_MISCLASS);   // cars -> appended to obstacles
    const uint64_t misObsMask = wave_ballot(realObs && seen == SIGHT_MISCLASS);   // obstacles -> appended to cars
    const uint64_t fpCarMask = wave_ballot(fpClass == 0), fpObsMask = wave_ballot(fpClass == 1);
    const uint64_t fpPedMask = wave_ballot(fpClass == 2), fpLaneMask = wave_ballot(fpClass == 3);
    const uint64_t outCarReal = wave_ballot(realCar && seen == SIGHT_NORMAL);
    const uint64_t outObsReal = wave_ballot(realObs && seen == SIGHT_NORMAL);
    const uint64_t outPedReal = wave_ballot(realPed && seen != SIGHT_NONE);
    const uint64_t outLaneReal = wave_ballot(isLaneRow && lseen != SIGHT_NONE);
    // FP pedestrians near the entries of the reference's carDets list (positions BEFORE the final filter):
    //   real cars keep their slot, misclassified obstacles are appended next, then the random-FP cars
    bool genA = false, genB = false;  // A: from my real object (car or misclassified obstacle), B: from my FP car
    V2 genPosA = v2(0.0, 0.0), genPosB = v2(0.0, 0.0);
    if (noiseType == 1) {
      int idxA = -1;
      if (realCar && seen == SIGHT_NORMAL) idxA = listIdx;
      else if (realObs && seen == SIGHT_MISCLASS) idxA = nCars0 + __popcll(misObsMask & below);
      if (idxA >= 0) {
        const dm_u32x4 u = pv_rng(S, genv, episode, elapsed, a, 6, idxA, 0);
        if (dm_unit(u.v[0]) < randBase * 10.0 && vlen(pos) < 250.0) {
          genA = true;
          genPosA = vadd(pos, vmul(v2(2.0 * dm_unit(u.v[1]) - 1.0, 2.0 * dm_unit(u.v[2]) - 1.0), 10.0));
        }
      }
      if (fpClass == 0) {
        const int idxB = nCars0 + __popcll(misObsMask) + __popcll(fpCarMask & below);
        const dm_u32x4 u = pv_rng(S, genv, episode, elapsed, a, 6, idxB, 0);
        if (dm_unit(u.v[0]) < randBase * 10.0 && vlen(fpPos) < 250.0) {
          genB = true;
          genPosB = vadd(fpPos, vmul(v2(2.0 * dm_unit(u.v[1]) - 1.0, 2.0 * dm_unit(u.v[2]) - 1.0), 10.0));
        }
      }
    }
    const uint64_t genReal = wave_ballot(genA && realCar), genMis = wave_ballot(genA && realObs), genFp = wave_ballot(genB);
    const int nOutCars = __popcll(outCarReal) + __popcll(misObsMask) + __popcll(fpCarMask);
    const int nOutObst = __popcll(outObsReal) + __popcll(misCarMask) + __popcll(fpObsMask);
    const int nOutPeds = __popcll(outPedReal) + __popcll(fpPedMask) + __popcll(genReal) + __popcll(genMis) + __popcll(genFp);
    const int nOutLanes = __popcll(outLaneReal) + __popcll(fpLaneMask);
    __syncthreads();  // row buffer zero-filled
    // A lane holds at most one real object and one random-FP trial, and each of them goes to exactly one of the row's blocks: one
    // store sequence per kind with the block, the position in it and the values selected, instead of one per (kind, block) pair
    // behind an exec-mask region each (13 of them).
    {
      // --- rectangles (cars / obstacles block): the real object ...
      //   cars block: surviving real cars, misclassified obstacles, random-FP cars; obstacles block: the mirror image
      const bool rCar = (realCar && seen == SIGHT_NORMAL) || (realObs && seen == SIGHT_MISCLASS);
      const bool rObs = (realObs && seen == SIGHT_NORMAL) || (realCar && seen == SIGHT_MISCLASS);
      const bool fCar = fpClass == 0, fObs = fpClass == 1;
      const int pReal = realCar ? (seen == SIGHT_NORMAL ? __popcll(outCarReal & below) : __popcll(outObsReal) + __popcll(misCarMask & below))
                                : (seen == SIGHT_NORMAL ? __popcll(outObsReal & below) : __popcll(outCarReal) + __popcll(misObsMask & below));
      const int pFp = fCar ? __popcll(outCarReal) + __popcll(misObsMask) + __popcll(fpCarMask & below)
                           : __popcll(outObsReal) + __popcll(misCarMask) + __popcll(fpObsMask & below);
#define PV_PUT_RECT(on_, car_, P_, q_, c_, s_, w_, h_, fin_)                                                               \
  do {                                                                                                                    \
    if (on_) {                                                                                                            \
      if ((P_) < ((car_) ? PV_LIM_CARS : PV_LIM_OBST)) {                                                                  \
        float* o = row + ((car_) ? PV_OFF_CARS + (P_)*7 : PV_OFF_OBST + (P_)*6);                                          \
        o[0] = (float)pv_normalize((q_).x, (5.0 * 2.0 / DRV_W), 0.0); o[1] = (float)pv_normalize((q_).y, (5.0 * 2.0 / DRV_H), 0.0); \
        o[2] = (float)(c_); o[3] = (float)(s_);                                                                           \
        o[4] = (float)pv_normalize((w_), 1.0 / 7.5, 0.5); o[5] = (float)pv_normalize((h_), 1.0 / 15.0, 0.5);              \
        if (car_) o[6] = (fin_);                                                                                          \
      } else overflow = 1;                                                                                                \
    }                                                                                                                     \
  } while (0)
      PV_PUT_RECT(rCar || rObs, rCar, pReal, pos, dc, ds, dw, dh, realCar ? (float)dfin : 0.0f);
      // ... and the random-FP trial
      PV_PUT_RECT(fCar || fObs, fCar, pFp, fpPos, fpc, fps, fpw, fph, 0.0f);
#undef PV_PUT_RECT
      // --- pedestrians block: real, random FP (class 2), then the FP pedestrians near cars in car-list order.  From the real object:
      //   a surviving pedestrian or the pedestrian generated near my car / misclassified obstacle; from the trial: a class-2 FP
      //   or the pedestrian generated near my FP car.
      const int pedBase = __popcll(outPedReal) + __popcll(fpPedMask);
      const bool pedReal = realPed && seen != SIGHT_NONE;
      const int pA = pedReal ? __popcll(outPedReal & below)
                             : pedBase + (realCar ? __popcll(genReal & below) : __popcll(genReal) + __popcll(genMis & below));
      const int pB = fpClass == 2 ? __popcll(outPedReal) + __popcll(fpPedMask & below)
                                  : pedBase + __popcll(genReal) + __popcll(genMis) + __popcll(genFp & below);
      const V2 qA = pedReal ? pos : genPosA, qB = fpClass == 2 ? fpPos : genPosB;
#define PV_PUT_PED(on_, P_, q_)                                                                                            \
  do {                                                                                                                    \
    if (on_) {                                                                                                            \
      if ((P_) < PV_LIM_PEDS) {                                                                                           \
        float* o = row + PV_OFF_PEDS + (P_)*2;                                                                            \
        o[0] = (float)pv_normalize((q_).x, (5.0 * 2.0 / DRV_W), 0.0); o[1] = (float)pv_normalize((q_).y, (5.0 * 2.0 / DRV_H), 0.0); \
      } else overflow = 1;                                                                                                \
    }                                                                                                                     \
  } while (0)
      PV_PUT_PED(pedReal || genA, pA, qA);
      PV_PUT_PED(fpClass == 2 || genB, pB, qB);
#undef PV_PUT_PED
      // --- lanes block: a visible lane row draws noise, so it never carries a trial (phase 5): one store sequence for both
      const bool lReal = isLaneRow && lseen != SIGHT_NONE, lFp = fpClass == 3;
      if (lReal || lFp) {
        const int p = lReal ? __popcll(outLaneReal & below) : __popcll(outLaneReal) + __popcll(fpLaneMask & below);
        if (p < PV_LIM_LANES) {
          float* o = row + PV_OFF_LANES + p * 4;
          o[0] = (float)(lReal ? ldist : fpLaneDist); o[1] = (float)(lReal ? lc : fpc); o[2] = (float)(lReal ? ls : fps);
          o[3] = (float)(lReal ? ltype : fpLaneType);
        } else overflow = 1;
      }
    }
    // --- self row + counts
    if (isSelf) {
      row[0] = (float)pv_normalize(pos.x, (5.0 * 2.0 / DRV_W), 5.0); row[1] = (float)pv_normalize(pos.y, (5.0 * 2.0 / DRV_H), 5.0);
      row[2] = (float)dc; row[3] = (float)ds;
      row[4] = (float)pv_normalize(C.carHy[typeA], 1.0 / 7.5, 0.5); row[5] = (float)pv_normalize(C.carHx[typeA], 1.0 / 15.0, 0.5);
      const double gx = in.gx, gy = in.gy;  // isSelf: lane == a
      row[6] = (float)pv_normalize(gx, (5.0 * 2.0 / DRV_W), 5.0); row[7] = (float)pv_normalize(gy, (5.0 * 2.0 / DRV_H), 5.0);
      row[8] = (float)CF_FIN(L.flags[a]);
      row[PV_DIM - 4] = (float)(nOutCars < PV_LIM_CARS ? nOutCars : PV_LIM_CARS);
      row[PV_DIM - 3] = (float)(nOutObst < PV_LIM_OBST ? nOutObst : PV_LIM_OBST);
      row[PV_DIM - 2] = (float)(nOutPeds < PV_LIM_PEDS ? nOutPeds : PV_LIM_PEDS);
      row[PV_DIM - 1] = (float)(nOutLanes < PV_LIM_LANES ? nOutLanes : PV_LIM_LANES);
    }
    __syncthreads();
    for (int i = lane; i < PV_DIM; i += DE_WAVE) grow[i] = row[i];
    (void)nObst0; (void)nPeds0; (void)nLanes0;
    __syncthreads();
PV_PROF(if (lane == 0) { const unsigned long long VT8 = __builtin_amdgcn_s_memtime(); atomicAdd(&g_pvprof[(e & 4095) * 16 + 0], VT1 - VT0); atomicAdd(&g_pvprof[(e & 4095) * 16 + 1], VT2 - VT1); atomicAdd(&g_pvprof[(e & 4095) * 16 + 2], VT3 - VT2); atomicAdd(&g_pvprof[(e & 4095) * 16 + 3], VT4 - VT3); atomicAdd(&g_pvprof[(e & 4095) * 16 + 4], VT5 - VT4); atomicAdd(&g_pvprof[(e & 4095) * 16 + 5], VT6 - VT5); atomicAdd(&g_pvprof[(e & 4095) * 16 + 6], VT7 - VT6); atomicAdd(&g_pvprof[(e & 4095) * 16 + 7], VT8 - VT7); atomicAdd(&g_pvprof[(e & 4095) * 16 + 8], 1ull); })
  }
  if (wave_ballot(overflow != 0) && lane == 0) envi[EI_ERR] = envi[EI_ERR] | 8;  // rows dropped (dynenv.h, error bit 3)
  return a;
}

DE_DEV PvIn pv_load_inputs(const DrvState& S, int e, int lane, int nPed, int nObst) {
  const size_t E = (size_t)S.E;
  const int A = S.A;
  PvIn in;
  in.px = in.py = in.ang = in.ox = in.oy = in.gx = in.gy = 0.0; in.flags = 0;
  if (lane < DRV_NB) {
    const bool used = lane < A || (lane >= DRV_SLOT_PED && lane < DRV_SLOT_PED + nPed);
    const double* b = S.body + (size_t)e * DRV_NB + lane;
    if (used) { in.px = b[BF_PX * E * DRV_NB]; in.py = b[BF_PY * E * DRV_NB]; in.ang = b[BF_ANG * E * DRV_NB]; in.flags = S.flags[(size_t)e * DRV_NB + lane]; }
  }
  if (lane < nObst) { in.ox = S.obst[(size_t)e * DRV_MAXO + lane]; in.oy = S.obst[E * DRV_MAXO + (size_t)e * DRV_MAXO + lane]; }
  if (lane < A) { in.gx = S.carx[CF_GOALX * E * 16 + (size_t)e * 16 + lane]; in.gy = S.carx[CF_GOALY * E * 16 + (size_t)e * 16 + lane]; }
  return in;
}

// stand-alone launch (after reset / set_state, where no step kernel ran)
extern "C" __global__ void __launch_bounds__(64, 4)  // 128 VGPRs: all 4096 environments resident in one pass
drv_partial_obs_kernel(DrvState S, int noiseType, double magn, float* __restrict__ obs) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int* envi = S.envi + (size_t)e * EI_COUNT;
  const int nPed = uniform_i(envi[EI_NPED]), nObst = uniform_i(envi[EI_NOBST]), elapsed = uniform_i(envi[EI_ELAPSED]);
  const uint32_t episode = (uint32_t)uniform_i(envi[EI_EPISODE]);
  const PvIn in = pv_load_inputs(S, e, lane, nPed, nObst);
  pv_env(S, g_P, e, lane, nPed, nObst, elapsed, episode, in, noiseType, magn, obs, 0, S.A);
}

// Fused call at the end of drv_step_kernel: the wave that has just finished environment e's step produces its Partial
// observation right away, from the state still in its LDS tile, while the waves of heavier environments are still
// stepping - the observation work of the light environments fills the launch's tail instead of a second launch.  The
// scratch tile aliases the step kernel's (no longer needed) LDS tile.
static_assert(sizeof(PvLds) <= sizeof(DrvLds), "the Partial observation tile must fit in the step kernel's LDS tile");
// The per-lane inputs come first in the argument list (an aggregate is passed in registers only while the function has 16 argument
// registers left, else through scratch), and of the state the function receives the four fields it reads, as scalars, not a
// reference to the struct (which the caller would first have to write to scratch, all of it, in every lane).
DE_OOL int drv_partial_obs_fused(PvIn in, uint64_t seed, int A, int* envi, int env_id_offset, int e, int nPedObst, int elapsed,
                                 uint32_t episode, int noiseTypeAgents, double magn, float* __restrict__ obs, int budgetCycles) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int nPed = nPedObst & 0xFF, nObst = nPedObst >> 8;  // (one register: with v31 reserved there are 31 for arguments)
  const int noiseType = noiseTypeAgents & 0xFF, nAgents = noiseTypeAgents >> 8;
  DrvState S = DrvState();  // (a local that never leaves registers: pv_env is inlined)
  S.seed = uniform_u64(seed); S.A = uniform_i(A); S.envi = uniform_ptr(envi); S.env_id_offset = uniform_i(env_id_offset);
  budgetCycles = uniform_i(budgetCycles);
  __syncthreads();  // every lane has taken what it needs out of the step tile
  // budgetCycles > 0: run passes until that many cycles from now have gone (the launch's forecast end), leave the rest
  return pv_env(S, *reinterpret_cast<PvLds*>(&g_L), uniform_i(e), lane_id(), uniform_i(nPed), uniform_i(nObst), uniform_i(elapsed),
                (uint32_t)uniform_i((int)episode), in, uniform_i(noiseType), uniform_d(magn), uniform_ptr(obs), 0, uniform_i(nAgents),
                budgetCycles > 0 ? t0 + (unsigned long long)budgetCycles : 0ull);
}

// The agent passes the step launch left over (EI_DEFER_OBS = first agent not done there): all of them for the environments
// that spent the step on the contact path and finish last - ten passes run by their one wave would sit on the launch's
// critical path, a lone wave being latency bound at ~50 k cycles per pass - and the last few of the light ones.  One
// wave per (environment, agent) spreads them over the whole chip; the step launch leaves the list of the environments concerned
// (S.pvq), so the grid is one residency round of blocks that take the items in turn - a grid of E x A blocks, most of which
// found nothing to do, took 9 us to dispatch.
extern "C" __global__ void __launch_bounds__(64, 4)
drv_partial_obs_deferred_kernel(DrvState S, int noiseType, double magn, float* __restrict__ obs) {
  const int lane = threadIdx.x, A = S.A;
  const int pv = drv_launch_pv(S);
  const int* list = S.pvq + 32 + pv * S.E;
  int n = uniform_i(S.pvq[16 * pv]);
  n = n < S.E ? n : S.E;  // (longer only if the length was not cleared: see the append in drv_step_body)
  if (blockIdx.x == 0 && lane == 0) S.pvq[16 * (pv ^ 1)] = 0;  // the next step's list starts empty (nobody reads or fills it now)
  // item j = (agent j / n, listed environment j % n): neighbouring blocks work on different environments
#pragma unroll 1
  for (int j = blockIdx.x; j < n * A; j += gridDim.x) {
    const int a = j / n, e = uniform_i(list[j - a * n]);
    const int* envi = S.envi + (size_t)e * EI_COUNT;
    if (a < uniform_i(envi[EI_DEFER_OBS])) continue;  // done in the step launch
    const int nPed = uniform_i(envi[EI_NPED]), nObst = uniform_i(envi[EI_NOBST]), elapsed = uniform_i(envi[EI_ELAPSED]);
    const uint32_t episode = (uint32_t)uniform_i(envi[EI_EPISODE]);
    const PvIn in = pv_load_inputs(S, e, lane, nPed, nObst);
    pv_env(S, g_P, e, lane, nPed, nObst, elapsed, episode, in, noiseType, magn, obs, a, a + 1);
    __syncthreads();
  }
}
