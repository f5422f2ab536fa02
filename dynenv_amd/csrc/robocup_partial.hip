// robocup_partial.hip — RoboCup PARTIAL observation on gfx950.
//
// Replaces RoboCupEnvironment.getAgentVision (detections, interactions, noise, misclassification, random false
// positives, false-positive balls near robots, polar / line conversion) with cutils.isSeenInArea / isLineInArea /
// doesInteract / addNoise / addNoiseLine / convertToPolar / normalizeLine, for every robot of an environment at each of
// the step's five snapshots.  rc_step_kernel exports the few state fields vision needs at each snapshot (RvSnap); the
// observation kernel below (its own launch bounds: a 166-VGPR callee inside rc_step_kernel cost the step kernel a wave
// per SIMD) turns them into rows and pays the sighting-based observation rewards (processSeens).  Mirrors
// oracle/robocup_partial.c operation by operation (bit-identical rows); the RNG keying is described there.
// For one agent the lanes are the things it might see:
//   lane 0 ball | 1..9 the other robots | 10..13 goalposts | 14..16 penalty crosses | 17..32 line crosses | 33..43 lines
//   | 44..53 the ten random false-positive trials.
// List positions come from ballots + popcounts; the field-cross `insert(len(crossDets), ..)` quirk is replayed serially.
#define RCP_CAP_BALL 32
#define RCP_CAP_ROB 20
#define RCP_CAP_GOAL 16
#define RCP_CAP_CROSS 16
#define RCP_CAP_FCROSS 28
#define RCP_CAP_LINE 12
#define RCP_OFF_BALL 0
#define RCP_OFF_ROB (RCP_OFF_BALL + RCP_CAP_BALL * 5)
#define RCP_OFF_GOAL (RCP_OFF_ROB + RCP_CAP_ROB * 7)
#define RCP_OFF_CROSS (RCP_OFF_GOAL + RCP_CAP_GOAL * 6)
#define RCP_OFF_FCROSS (RCP_OFF_CROSS + RCP_CAP_CROSS * 6)
#define RCP_OFF_LINE (RCP_OFF_FCROSS + RCP_CAP_FCROSS * 8)
#define RCP_OFF_TAIL (RCP_OFF_LINE + RCP_CAP_LINE * 5)
#define RCP_DIM (RCP_OFF_TAIL + 6 + 2 + 9)
#define RCP_SEEN_STRIDE 12  // per agent: lSum, bSum, rSum[9], pad

#define RV_NONE 0
#define RV_PARTIAL 1
#define RV_DISTANT 2
#define RV_NORMAL 3
#define RV_MISCLASS 4
#define RV_FOV (DM_PI / 4.0)
#define RV_MAXVIS0 ((RC_W * 0.4) * (RC_W * 0.4))
#define RV_MAXVIS1 ((RC_W * 0.8) * (RC_W * 0.8))
#define RV_STD_NORM (2.0 / RC_W)
#define RV_SIZE_NORM (10.0 / 5.0)

#ifndef RC_PARTIAL_FUNCTIONS
struct RvDetTable {  // detections every lane may need (rotPt, `is not None`)
  double px[48], py[48];
  int has[48];
};
// what getAgentVision reads of the environment, as exported by rc_step_kernel at a snapshot (or after a reset)
struct RvSnap {
  double px[21], py[21], ang[20], head[10];
  int rflags[10], owned, close0, close1, tkey;
  int pad[2];
};
// (included twice by robocup_kernels.hip: the definitions above where the step kernel needs them, the functions below - with
//  RC_PARTIAL_FUNCTIONS - behind the step kernels, so that the distance between rc_step_kernel and the out-of-line functions it
//  calls does not change with the size of the vision code: the launch's instruction working set is larger than the 64 KB
//  instruction cache and its time moves by 1-2 % with that distance)
DE_OOL int rc_partial_obs_fused(uint64_t seed, int env_id_offset, int* envi, int n, int R, int noise_type, double noise_magn, RvSnap* snap,
                                int flags, double* prew0, double* epr, int E, double* epo, int e, float* __restrict__ obs,
                                double* __restrict__ rewards, int* seenPart, int budgetCycles);
#else
struct RvLds {  // LDS of the observation kernel (2.7 KB: every environment of a 4096-env launch is resident)
  double px[RC_NB], py[RC_NB], ang[RC_NB], head[16];
  int rflags[16], owned, close0, close1, tkey;
  RvDetTable T;
  int seen[10 * RCP_SEEN_STRIDE];
  int seenSum[10 * RCP_SEEN_STRIDE];  // (fused path with a deadline: the sum over the snapshots done so far)
  alignas(16) double atanTab[30];     // dev_atan2_t's table
};
__shared__ RvLds g_V;
DE_DEV V2 rv_pos(const RvLds& V, int r) { return v2((V.px[2 * r] + V.px[2 * r + 1]) / 2.0, (V.py[2 * r] + V.py[2 * r + 1]) / 2.0); }  // Robot.getPos
DE_DEV double rv_angle(const RvLds& V, int r) { return (V.ang[2 * r] + V.ang[2 * r + 1]) / 2.0; }
DE_DEV int rv_team(const RvLds& V, int r) { return (V.rflags[r] & RF_TEAMPOS) ? 1 : -1; }

DE_DEV dm_u32x4 rv_rng(uint64_t seed, uint32_t genv, uint32_t episode, uint32_t tkey, int agent, int kind, int index, int block) {
  const uint32_t entity = (uint32_t)agent | ((uint32_t)kind << 4) | ((uint32_t)index << 8) | ((uint32_t)block << 16);
  return dm_env_rng(seed, genv, episode, DM_RNG_OBS_NOISE, entity, tkey);
}
DE_DEV V2 rv_rot(V2 v, double c, double s) { return v2(v.x * c - v.y * s, v.x * s + v.y * c); }  // Vec2d.rotated
DE_DEV double rv_scale(double val, double norm) { return ((val * norm) - 0.5) / 0.5; }
DE_DEV double rv_normalize(double pt, double nf) { return ((pt * nf) - 0.0) * 2.0 * 1.0; }
DE_DEV double rv_nas(double pt, double nf, double mean) { return (pt - mean) * nf * 1.0; }

// cutils.doesInteract (obj1 = table entry i, obj2 = (has2, p2))
DE_DEV int rv_interact(const RvDetTable& T, int i, bool has2, V2 p2, double radius, bool canOcclude) {
  if (!T.has[i] || !has2) return 0;
  const V2 p1 = v2(T.px[i], T.py[i]);
  int type = 0;
  if (vlen(vsub(p1, p2)) < radius) type = 1;
  if (canOcclude) {
    const double dist = vcross(p1, p2) / vlen(p1);
    if (dm_abs(dist) < radius && vlensq(p1) < vlensq(p2)) type = 2;
  }
  return type;
}

struct RvArgs {
  uint64_t seed;
  uint32_t genv, episode, tkey;
  int R, n, noiseType;
  double magn;
};

// One snapshot: rows of all R agents -> out[R][RCP_DIM]; seen += the snapshot's (numLandMarks, ballsSeen, robotsSeen)
// Returns overflow (bit 0) | first agent NOT done << 8 (aEnd, or less when `deadline` - a s_memtime value, 0 = none - had passed).
DE_DEV int rc_partial_vision(const RvArgs& A, RvLds& V, int lane, float* __restrict__ out, bool countSeen, int aBegin, int aEnd,
                             unsigned long long deadline = 0ull) {
  RvDetTable& T = V.T;
  int* seen = countSeen ? V.seen : nullptr;
  const int R = A.R;
  const double randBase = 0.01 * A.magn;
  const uint64_t below = lanemask_lt();
  int overflow = 0;
  DEV_ATAN_TAB_INIT(V.atanTab, lane);  // (made visible by the barrier behind the first agent's detection table)
  const int atanTab = dev_lds_addr(V.atanTab);
  // lane role (fixed over agents)
  const bool isBall = lane == 0, isRob = lane >= 1 && lane < R, isGoal = lane >= 10 && lane < 14, isCross = lane >= 14 && lane < 17;
  const bool isFc = lane >= 17 && lane < 33, isLine = lane >= 33 && lane < 44, isTrial = lane >= 44 && lane < 54;
  const bool isPoint = isBall || isRob || isGoal || isCross || isFc;
  int a = aBegin;
#pragma unroll 1
  for (; a < aEnd; ++a) {
    if (deadline != 0ull && __builtin_amdgcn_s_memtime() >= deadline) break;
    float* __restrict__ row = out + (size_t)a * RCP_DIM;
    for (int i = lane; i < RCP_DIM; i += DE_WAVE) row[i] = 0.0f;
    const V2 pos = rv_pos(V, a);
    const double angle = rv_angle(V, a), headAngle = angle + V.head[a];
    const int team = rv_team(V, a);
    const int aflags = V.rflags[a];
    // one sincos call for the three uniform angles: lane 0 -> FoV edge 1, lane 1 -> FoV edge 2, others -> -headAngle
    const DevSC sc0 = dev_sincos_v(lane == 0 ? headAngle + RV_FOV : (lane == 1 ? headAngle - RV_FOV : -headAngle));
    const V2 vec1 = v2(1.0 * bcast_d(sc0.c, 0) - 0.0 * bcast_d(sc0.s, 0), 1.0 * bcast_d(sc0.s, 0) + 0.0 * bcast_d(sc0.c, 0));
    const V2 vec2 = v2(1.0 * bcast_d(sc0.c, 1) - 0.0 * bcast_d(sc0.s, 1), 1.0 * bcast_d(sc0.s, 1) + 0.0 * bcast_d(sc0.c, 1));
    const double cR = bcast_d(sc0.c, 2), sR = bcast_d(sc0.s, 2);
    // ---- detections -------------------------------------------------------------------------------------
    int seenT = RV_NONE;
    bool has = false;
    V2 p = v2(0.0, 0.0), p2 = v2(0.0, 0.0);  // rotPt (lines: pt1, pt2)
    double size = 0.0, e3 = 0.0, e4 = 0.0, e5 = 0.0;
    int robId = -1;
    if (isPoint) {
      V2 objp;
      double maxDist = RV_MAXVIS0, radius = 5.0;
      if (isBall) { objp = v2(V.px[RC_BALL], V.py[RC_BALL]); radius = BALL_R; e3 = (double)(V.owned * team); }
      else if (isRob) {
        robId = (lane - 1) < a ? (lane - 1) : lane;
        objp = rv_pos(V, robId); maxDist = RV_MAXVIS1; radius = ROBOT_TOTAL_RADIUS;
        e3 = rv_angle(V, robId) - headAngle; e4 = (double)(team * rv_team(V, robId));
        e5 = (aflags & (RF_FALLEN | RF_PENAL)) ? 1.0 : 0.0;
      } else {
        objp = v2(RC.visPx[lane], RC.visPy[lane]); e3 = RC.visT0[lane]; e4 = RC.visT1[lane];
        if (isGoal) maxDist = RV_MAXVIS1;
        if (isFc) e5 = 0.0 - headAngle;
      }
      const V2 point = vsub(objp, pos);
      const double dist1 = vcross(vec1, point), dist2 = vcross(vec2, point);
      size = radius;
      if (dist1 < radius && dist2 > -radius) {
        if (dist1 < -radius && dist2 > radius) seenT = vlensq(point) < maxDist ? RV_NORMAL : RV_DISTANT;
        else seenT = RV_PARTIAL;
        p = rv_rot(point, cR, sR);
        has = true;
      }
    } else if (isLine) {  // cutils.isLineInArea
      const V2 q1 = vsub(v2(RC.visPx[lane], RC.visPy[lane]), pos), q2 = vsub(v2(RC.visQx[lane], RC.visQy[lane]), pos);
      e3 = RC.visT0[lane]; e4 = RC.visT1[lane];
      const double dist11 = vcross(vec1, q1), dist12 = vcross(vec1, q2);
      if (!(dist11 > 0.0 && dist12 > 0.0)) {
        const double dist21 = vcross(vec2, q1), dist22 = vcross(vec2, q2);
        if (!(dist21 < 0.0 && dist22 < 0.0)) {
          V2 pt1, pt2;
          seenT = RV_NORMAL;
          if (dist11 <= 0.0 && dist21 >= 0.0) pt1 = q1;
          else {
            const V2 d = vsub(q2, q1);
            const double i1 = vcross(q1, vec1) / (vcross(vec1, d) + 1e-7), i2 = vcross(q1, vec2) / (vcross(vec2, d) + 1e-7);
            const double inter = (i1 < 1.0 && i2 < 1.0) ? (i1 > i2 ? i1 : (i2 > i1 ? i2 : i1)) : (i1 < i2 ? i1 : (i2 < i1 ? i2 : i1));
            pt1 = vadd(q1, vmul(d, inter));
            seenT = RV_PARTIAL;
          }
          if (dist12 <= 0.0 && dist22 >= 0.0) pt2 = q2;
          else {
            const V2 d = vsub(q1, q2);
            const double i1 = vcross(q2, vec1) / vcross(vec1, d), i2 = vcross(q2, vec2) / vcross(vec2, d);
            const double inter = (i1 < 1.0 && i2 < 1.0) ? (i1 > i2 ? i1 : (i2 > i1 ? i2 : i1)) : (i1 < i2 ? i1 : (i2 < i1 ? i2 : i1));
            pt2 = vadd(q2, vmul(d, inter));
            seenT = RV_PARTIAL;
          }
          if (vlensq(pt1) > RV_MAXVIS1 || vlensq(pt2) > RV_MAXVIS1) seenT = RV_DISTANT;
          pt1 = rv_rot(pt1, cR, sR);
          pt2 = rv_rot(pt2, cR, sR);
          if (pt1.x < 0.0 || pt2.x < 0.0) seenT = RV_NONE;
          p = pt1; p2 = pt2; has = true;
        }
      }
    }
    if (lane < 48) { T.px[lane] = p.x; T.py[lane] = p.y; T.has[lane] = (isPoint && has) ? 1 : 0; }
    __syncthreads();
    // ---- interactions ---------------------------------------------------------------------------------------
    int inter = 0;
    if (isPoint) {
      for (int i = 1; i < R; ++i) {  // max over the robots (other than me) of doesInteract(rob, me, totalRadius * 2)
        if (!uniform_i(T.has[i])) continue;  // robot i is outside the field of view (None): NoInter for everybody
        if (i == lane) continue;
        const int t = rv_interact(T, i, has, p, ROBOT_TOTAL_RADIUS * 2.0, true);
        inter = t > inter ? t : inter;
      }
      if (isBall) for (int k = 0; k < 4; ++k) {  // ballPostInter: doesInteract(ball, post, ballRadius * 8, False)
        const int t = (has && T.has[10 + k]) ? ((vlen(vsub(p, v2(T.px[10 + k], T.py[10 + k]))) < BALL_R / 2.0 * 8.0) ? 1 : 0) : 0;
        inter = t > inter ? t : inter;
      }
      if (isCross) {  // ballCrossInter: doesInteract(ball, cross, ballRadius * 4, False)
        const int t = (T.has[0] && has) ? ((vlen(vsub(v2(T.px[0], T.py[0]), p)) < BALL_R / 2.0 * 4.0) ? 1 : 0) : 0;
        inter = t > inter ? t : inter;
      }
    }
    // ---- one pair of Philox blocks per lane for its draw site: point noise (kind by role), line noise (kind 5) or
    //      false-positive trial (kind 7) - the three groups live on disjoint lanes, so two calls serve them all
    int rkind = 7, rindex = lane - 44;
    if (isPoint) { rkind = isBall ? 0 : isRob ? 1 : isGoal ? 2 : isCross ? 3 : 4; rindex = isBall ? 0 : isRob ? lane - 1 : isGoal ? lane - 10 : isCross ? lane - 14 : lane - 17; }
    else if (isLine) { rkind = 5; rindex = lane - 33; }
    dm_u32x4 u0, u1;
    u0.v[0] = u0.v[1] = u0.v[2] = u0.v[3] = 0u; u1 = u0;
    if ((isPoint && inter != 2 && seenT) || (isLine && seenT) || isTrial) {
      u0 = rv_rng(A.seed, A.genv, A.episode, A.tkey, a, rkind, rindex, 0);
      u1 = rv_rng(A.seed, A.genv, A.episode, A.tkey, a, rkind, rindex, 1);
    }
    // ---- noise (cutils.addNoise on points, addNoiseLine on lines) ---------------------------------------------
    if (isPoint) {
      if (inter == 2) seenT = RV_NONE;
      else if (seenT) {
        const bool misClass = isBall || isCross, angleNoise = isFc;
        const double maxDist = (isRob || isGoal) ? RV_MAXVIS1 : RV_MAXVIS0;
        const V2 noiseVec = vmul(v2(dm_unit(u0.v[0]) - 0.5, dm_unit(u0.v[1]) - 0.5), A.magn);
        if (A.noiseType == 0) {
          if (dm_unit(u0.v[2]) < randBase) seenT = RV_NONE;
          p = vadd(p, noiseVec);
          size *= (1.0 - (dm_unit(u1.v[0]) - 0.5) * 0.2);
          if (angleNoise) e5 += (dm_unit(u1.v[1]) - 0.5) * A.magn / 10.0;
        } else {
          int st = seenT;
          const double range = 0.25 + 3.75 * vlensq(p) / maxDist;
          double multiplier = range;
          if (inter == 1) multiplier = range * 2.0;
          if (st == RV_DISTANT) multiplier = range * 3.0;
          else if (st == RV_PARTIAL) multiplier = range * 4.0;
          const V2 newPos = vadd(p, v2(noiseVec.x * multiplier / 4.0, noiseVec.y * multiplier / 4.0));
          const double diff = vlen(newPos) - vlen(p);
          if (dm_unit(u0.v[2]) < randBase * multiplier) st = RV_NONE;
          if (misClass && dm_unit(u0.v[3]) < randBase * multiplier / 2.0) st = RV_MISCLASS;
          seenT = st;
          p = newPos;
          size *= 1.0 + (dm_unit(u1.v[0]) * 0.1 * diff);
          if (angleNoise) e5 += (dm_unit(u1.v[1]) - 0.5) * A.magn * multiplier / 180.0;
        }
      }
    } else if (isLine && seenT) {
      const V2 n1 = vmul(v2(dm_unit(u0.v[0]) - 0.5, dm_unit(u0.v[1]) - 0.5), A.magn);
      const V2 n2 = vmul(v2(dm_unit(u0.v[2]) - 0.5, dm_unit(u0.v[3]) - 0.5), A.magn);
      if (A.noiseType == 0) {
        if (dm_unit(u1.v[0]) < randBase) seenT = RV_NONE;
        p = vadd(p, n1);
        p2 = vadd(p2, n2);
      } else {
        const double m1 = 0.25 + 3.75 * vlensq(p) / RV_MAXVIS1, m2 = 0.25 + 3.75 * vlensq(p2) / RV_MAXVIS1;
        const double m = (m1 + m2) * 0.5;
        if (dm_unit(u1.v[0]) < randBase * m) seenT = RV_NONE;
        p = vadd(p, v2(n1.x * m1 / 2.0, n1.y * m1 / 2.0));
        p2 = vadd(p2, v2(n2.x * m2 / 2.0, n2.y * m2 / 2.0));
      }
    }
    // ---- the third element of the observation: (numLandMarks, robotsSeen, ballsSeen) ------------------------------
    const bool robSeen = isRob && seenT != RV_NONE;
    const bool ballsSeen = wave_ballot(isBall && seenT != RV_NONE && seenT != RV_MISCLASS) != 0ull;
    // ---- misclassification swaps, filters, list positions ---------------------------------------------------------
    const bool ballKeep = isBall && seenT != RV_NONE && seenT != RV_MISCLASS, ballToCross = isBall && seenT == RV_MISCLASS;
    const bool crossKeep = isCross && seenT != RV_NONE && seenT != RV_MISCLASS, crossToBall = isCross && seenT == RV_MISCLASS;
    const bool robKeep = robSeen, goalKeep = isGoal && seenT != RV_NONE, fcKeep = isFc && seenT != RV_NONE && seenT != RV_MISCLASS;
    const bool lineKeep = isLine && seenT != RV_NONE;
    const uint64_t mBallKeep = wave_ballot(ballKeep), mBallToCross = wave_ballot(ballToCross), mCrossKeep = wave_ballot(crossKeep);
    const uint64_t mCrossToBall = wave_ballot(crossToBall), mRobKeep = wave_ballot(robKeep), mGoalKeep = wave_ballot(goalKeep);
    const uint64_t mFcKeep = wave_ballot(fcKeep), mLineKeep = wave_ballot(lineKeep);
    const int nBall0 = __popcll(mBallKeep) + __popcll(mCrossToBall), nCross0 = __popcll(mCrossKeep) + __popcll(mBallToCross);
    const int nRob0 = __popcll(mRobKeep), nGoal0 = __popcll(mGoalKeep), nFc0 = __popcll(mFcKeep), nLine0 = __popcll(mLineKeep);
    const int numLandMarks = nFc0 + nLine0 + nCross0 + nGoal0;
    if (ballToCross) {  // the misclassified ball joins the crosses with two random tags
      const dm_u32x4 u = rv_rng(A.seed, A.genv, A.episode, A.tkey, a, 6, 0, 0);
      e3 = (double)dm_randint(u.v[0], -1, 1); e4 = (double)dm_randint(u.v[1], -1, 1);
    }
    if (crossToBall) e3 = 0.0;
    // ---- random false positives (lanes 44..53 = trials 0..9) ------------------------------------------------------
    int fpClass = -1;
    if (isTrial) {
      const dm_u32x4 u = u0;
      if (dm_unit(u.v[0]) < randBase) {
        fpClass = dm_randint(u.v[1], 0, 5);
        const double d = dm_unit(u.v[2]) * dm_sqrt(RV_MAXVIS1);
        const double an = dm_unit(u.v[3]) * 2.0 * RV_FOV - RV_FOV;
        const DevSC sc = dev_sincos_v(an);
        p = v2(d * sc.c - 0.0 * sc.s, d * sc.s + 0.0 * sc.c);
        seenT = RV_NORMAL; has = true;
        const double f = 1.0 - 0.4 * (dm_unit(u1.v[0]) - 0.5);
        if (fpClass == 0) { size = BALL_R / 2.0 * 2.0 * f; e3 = 0.0; }
        else if (fpClass == 1) {
          size = ROBOT_TOTAL_RADIUS * f;
          e3 = (dm_unit(u1.v[1]) - 0.5) * 2.0 * DM_PI; e4 = (dm_unit(u1.v[2]) > 0.5) ? -1.0 : 1.0; e5 = (dm_unit(u1.v[3]) > 0.9) ? 1.0 : 0.0;
        } else if (fpClass <= 4) {
          size = 5.0 * f; e3 = (double)dm_randint(u1.v[1], -1, 1); e4 = (double)dm_randint(u1.v[2], -1, 1);
          if (fpClass == 4) e5 = dm_unit(u1.v[3]) * DM_PI * 2.0;
        }
      }
    }
    const uint64_t mFp0 = wave_ballot(fpClass == 0), mFp1 = wave_ballot(fpClass == 1), mFp2 = wave_ballot(fpClass == 2);
    const uint64_t mFp3 = wave_ballot(fpClass == 3), mFp4 = wave_ballot(fpClass == 4);
    // field crosses: fieldCrossDets.insert(len(crossDets), fp) replayed in trial order
    int fcPos = fcKeep ? __popcll(mFcKeep & below) : -1;
    {
      int nCrossCur = nCross0, nFcCur = nFc0;
      for (int i = 0; i < 10; ++i) {
        const int ci = bcast_i(fpClass, 44 + i);
        if (ci == 3) nCrossCur++;
        else if (ci == 4) {
          const int at = nCrossCur < nFcCur ? nCrossCur : nFcCur;
          if (fcPos >= at) fcPos++;
          if (lane == 44 + i) fcPos = at;
          nFcCur++;
        }
      }
    }
    // ---- false-positive balls near robots (REALISTIC): robDets = kept real robots, then the FP robots ---------------
    const int robPos = robKeep ? __popcll(mRobKeep & below) : (fpClass == 1 ? nRob0 + __popcll(mFp1 & below) : -1);
    bool genBall = false;
    V2 genP = v2(0.0, 0.0);
    double genSize = 0.0;
    if (A.noiseType == 1 && robPos >= 0 && seenT == RV_NORMAL) {
      const dm_u32x4 u = rv_rng(A.seed, A.genv, A.episode, A.tkey, a, 8, robPos, 0), u1 = rv_rng(A.seed, A.genv, A.episode, A.tkey, a, 8, robPos, 1);
      if (dm_unit(u.v[0]) < randBase * 10.0 && vlen(p) < 250.0) {
        genBall = true;  // (the `rob[0] = NoSighting` coin changes nothing that is returned)
        genP = vadd(p, vmul(v2(2.0 * dm_unit(u.v[2]) - 1.0, 2.0 * dm_unit(u.v[3]) - 1.0), ROBOT_TOTAL_RADIUS));
        genSize = BALL_R / 2.0 * 2.0 * (1.0 - 0.4 * (dm_unit(u1.v[0]) - 0.5));
      }
    }
    const uint64_t mGen = wave_ballot(genBall);
    const int nBall = nBall0 + __popcll(mFp0) + __popcll(mGen), nRob = nRob0 + __popcll(mFp1), nGoal = nGoal0 + __popcll(mFp2);
    const int nCross = nCross0 + __popcll(mFp3), nFc = nFc0 + __popcll(mFp4), nLine = nLine0;
    __threadfence_block();  // s_waitcnt: the zero fill of the row has completed before the scattered entries are issued
    // ---- conversion + placement ----------------------------------------------------------------------------------
    const int closest = (a == V.close0 || a == V.close1) ? 1 : 0;
#define RV_BALL_ROW(P_, q_, sz_, own_)                                                                                  \
  do {                                                                                                                  \
    if ((P_) < RCP_CAP_BALL) {                                                                                          \
      float* o = row + RCP_OFF_BALL + (P_)*5;                                                                           \
      o[0] = (float)rv_normalize((q_).x, RV_STD_NORM); o[1] = (float)rv_normalize((q_).y, RV_STD_NORM);                 \
      o[2] = (float)rv_nas((sz_), RV_SIZE_NORM, BALL_R / 2.0 * 2.0); o[3] = (float)(own_); o[4] = (float)closest;       \
    } else overflow = 1;                                                                                                \
  } while (0)
    if (ballKeep) RV_BALL_ROW(0, p, size, e3);
    if (crossToBall) { const int P = __popcll(mBallKeep) + __popcll(mCrossToBall & below); RV_BALL_ROW(P, p, size, 0.0); }
    if (fpClass == 0) { const int P = nBall0 + __popcll(mFp0 & below); RV_BALL_ROW(P, p, size, 0.0); }
    if (genBall) { const int P = nBall0 + __popcll(mFp0) + __popcll(mGen & below); RV_BALL_ROW(P, genP, genSize, 0.0); }
    // destination of my polar entry (goals, crosses incl. the misclassified ball, field crosses), if any
    int P = -1, off = 0, cap = 0, feat = 6;
    if (goalKeep) { P = __popcll(mGoalKeep & below); off = RCP_OFF_GOAL; cap = RCP_CAP_GOAL; }
    else if (fpClass == 2) { P = nGoal0 + __popcll(mFp2 & below); off = RCP_OFF_GOAL; cap = RCP_CAP_GOAL; }
    else if (crossKeep) { P = __popcll(mCrossKeep & below); off = RCP_OFF_CROSS; cap = RCP_CAP_CROSS; }
    else if (ballToCross) { P = __popcll(mCrossKeep); off = RCP_OFF_CROSS; cap = RCP_CAP_CROSS; }
    else if (fpClass == 3) { P = nCross0 + __popcll(mFp3 & below); off = RCP_OFF_CROSS; cap = RCP_CAP_CROSS; }
    else if (fcPos >= 0) { P = fcPos; off = RCP_OFF_FCROSS; cap = RCP_CAP_FCROSS; feat = 8; }
    // one sincos for the orientation angle of robot rows (e3) and of field-cross rows (e5); one atan2 + one sincos for the
    // bearing of polar rows and the direction of line rows: the groups live on disjoint lanes
    const V2 ldiff = vsub(p2, p);
    DevSC scA; scA.s = 0.0; scA.c = 0.0;
    if (robPos >= 0 || (P >= 0 && feat == 8)) scA = dev_sincos_v(robPos >= 0 ? e3 : e5);
    DevSC scB; scB.s = 0.0; scB.c = 0.0;
    if (P >= 0 || lineKeep) {
      // (one call for both kinds of row: operands selected, not two calls in the arms of a conditional - a wave executes both)
      const double ang = dev_atan2_t(lineKeep ? ldiff.y : p.y * (double)team, lineKeep ? ldiff.x : p.x * (double)team, atanTab);
      scB = dev_sincos_v(ang);
    }
    if (robPos >= 0) {
      if (robPos < RCP_CAP_ROB) {
        float* o = row + RCP_OFF_ROB + robPos * 7;
        o[0] = (float)rv_normalize(p.x, RV_STD_NORM); o[1] = (float)rv_normalize(p.y, RV_STD_NORM);
        o[2] = (float)rv_nas(size, RV_SIZE_NORM, ROBOT_TOTAL_RADIUS); o[3] = (float)scA.c; o[4] = (float)scA.s; o[5] = (float)e4; o[6] = (float)e5;
      } else overflow = 1;
    }
    if (P >= 0) {  // convertToPolar
      if (P < cap) {
        float* o = row + off + P * feat;
        const double dist = dm_sqrt(p.x * p.x + p.y * p.y);
        o[0] = (float)rv_scale(dist, RV_STD_NORM); o[1] = (float)scB.c; o[2] = (float)scB.s;
        o[3] = (float)((size - 5.0) * RV_SIZE_NORM); o[4] = (float)(e3 * (double)team); o[5] = (float)(e4 * (double)team);
        if (feat == 8) { o[6] = (float)scA.c; o[7] = (float)(-scA.s); }
      } else overflow = 1;
    }
    if (lineKeep) {  // normalizeLine
      const int PL = __popcll(mLineKeep & below);
      float* o = row + RCP_OFF_LINE + PL * 5;
      const double dist = dm_abs(p2.x * p.y - p2.y * p.x) / (vlen(ldiff) + 1e-7);
      o[0] = (float)rv_scale(dist, RV_STD_NORM); o[1] = (float)scB.c; o[2] = (float)scB.s; o[3] = (float)e3; o[4] = (float)e4;
    }
#undef RV_BALL_ROW
    if (lane == 0) {
      float* o = row + RCP_OFF_TAIL;
      o[0] = (float)(nBall < RCP_CAP_BALL ? nBall : RCP_CAP_BALL); o[1] = (float)(nRob < RCP_CAP_ROB ? nRob : RCP_CAP_ROB);
      o[2] = (float)(nGoal < RCP_CAP_GOAL ? nGoal : RCP_CAP_GOAL); o[3] = (float)(nCross < RCP_CAP_CROSS ? nCross : RCP_CAP_CROSS);
      o[4] = (float)(nFc < RCP_CAP_FCROSS ? nFc : RCP_CAP_FCROSS); o[5] = (float)nLine;
      o[6] = (float)numLandMarks; o[7] = ballsSeen ? 1.0f : 0.0f;
      if (seen) { seen[a * RCP_SEEN_STRIDE + 0] += numLandMarks; seen[a * RCP_SEEN_STRIDE + 1] += ballsSeen ? 1 : 0; }
    }
    if (isRob) {
      row[RCP_OFF_TAIL + 8 + (lane - 1)] = robSeen ? 1.0f : 0.0f;
      if (seen) seen[a * RCP_SEEN_STRIDE + 2 + (lane - 1)] += robSeen ? 1 : 0;
    }
    __syncthreads();  // the detection table is rewritten by the next agent
  }
  return (int)(wave_ballot(overflow != 0) != 0ull) | (a << 8);
}

// ------------------------------------------------------------------------------------------------
// The observation kernel: one wave per environment, the step's five snapshots in turn (after a reset: five views of the
// initial state with draw keys t = 0..4, environment_base.py:217-222).  With `rewards` it also applies processSeens:
// the step kernel left rewards[e][a] = robot reward + team reward and prew0 = its positive part; this kernel adds the
// observation reward in the reference's order of operations and only then updates the episode accumulators.
// ------------------------------------------------------------------------------------------------
DE_DEV RvArgs rv_args(const RcState& S, int e) {
  RvArgs va;
  va.seed = S.seed; va.genv = (uint32_t)(S.env_id_offset + e);
  va.episode = (uint32_t)uniform_i(S.envi[(size_t)e * RE_COUNT + RE_EPISODE]);
  va.R = S.R; va.n = S.n; va.noiseType = S.noise_type; va.magn = S.noise_magn;
  va.tkey = 0;
  return va;
}
// snapshot t of environment e -> its R observation rows (+ the snapshot's seen counts into V.seen)
DE_DEV int rv_snapshot(const RcState& S, RvLds& V, RvArgs va, int e, int lane, int t, float* __restrict__ obs, bool countSeen,
                       int aBegin, int aEnd, unsigned long long deadline = 0ull) {
  const RvSnap& sn = S.snap[(size_t)e * 5 + t];
  if (lane < 21) { V.px[lane] = sn.px[lane]; V.py[lane] = sn.py[lane]; }
  if (lane < 20) V.ang[lane] = sn.ang[lane];
  if (lane < 10) { V.head[lane] = sn.head[lane]; V.rflags[lane] = sn.rflags[lane]; }
  if (lane == 0) { V.owned = sn.owned; V.close0 = sn.close0; V.close1 = sn.close1; V.tkey = sn.tkey; }
  __syncthreads();
  va.tkey = (uint32_t)uniform_i(V.tkey);
  const int ov = rc_partial_vision(va, V, lane, obs + ((size_t)e * 5 + t) * S.R * RCP_DIM, countSeen, aBegin, aEnd, deadline);
  __syncthreads();
  return ov;
}
// processSeens (oracle/robocup.c rc_process_seens) on the seen counts of the five snapshots, then the episode sums in
// the reference's order of operations
DE_DEV void rv_finalize(const RcState& S, const int* seen, int e, int lane, double* __restrict__ rewards) {
  const int R = S.R;
  if (lane < R) {
    double obsRew = 0.0;
    if (S.flags & 8) {  // useObsRewards
      const int* sn = seen + lane * RCP_SEEN_STRIDE;
      double lSeens = (double)sn[0] / 5.0, rSeens = 0.0, bSeens = (double)sn[1];
      lSeens = lSeens < 0.0 ? 0.0 : (lSeens > 3.0 ? 3.0 : lSeens);
      for (int k = 0; k < R - 1; ++k) { const double r = (double)sn[2 + k]; rSeens += r < 0.0 ? 0.0 : (r > 2.0 ? 2.0 : r); }
      bSeens = bSeens < 0.0 ? 0.0 : (bSeens > 3.0 ? 3.0 : bSeens);
      obsRew += (0.0025 * (rSeens + lSeens) + 0.01 * bSeens);
    }
    double rew = rewards[(size_t)e * R + lane];
    rew += obsRew;
    double prew = S.prew0[(size_t)e * 16 + lane];
    prew += dm_max(obsRew, 0.0);
    double* er = S.epr + (size_t)e * 16 + lane;
    double* ep = S.epr + (size_t)S.E * 16 + (size_t)e * 16 + lane;
    double* eo = S.epo + (size_t)e * 16 + lane;
    *er = *er + rew;
    *ep = *ep + prew;
    *eo = *eo + obsRew;
    rewards[(size_t)e * R + lane] = rew;
  }
}
DE_DEV void rv_env(const RcState& S, RvLds& V, const int e, const int lane, float* __restrict__ obs, double* __restrict__ rewards) {
  for (int i = lane; i < 10 * RCP_SEEN_STRIDE; i += DE_WAVE) V.seen[i] = 0;
  const RvArgs va = rv_args(S, e);
  int ov = 0;
#pragma unroll 1
  for (int t = 0; t < 5; ++t) ov |= rv_snapshot(S, V, va, e, lane, t, obs, rewards != nullptr, 0, S.R) & 1;
  if (ov && lane == 0) S.envi[(size_t)e * RE_COUNT + RE_ERR] |= 8;
  if (rewards) rv_finalize(S, V.seen, e, lane, rewards);
}
// stand-alone launch: after reset / set_state (rewards == nullptr)
extern "C" __global__ void __launch_bounds__(64, 4)
rc_partial_obs_kernel(RcState S, float* __restrict__ obs, double* __restrict__ rewards) {
  rv_env(S, g_V, blockIdx.x, threadIdx.x, obs, rewards);
}
// Fused call at the end of rc_step_partial_kernel: the wave that has finished environment e's step turns its five snapshots
// into observation rows right away, while the waves of the environments with contact work are still stepping (two thirds
// of a RoboCup launch are such a tail).  The tile aliases the step kernel's LDS tile, which is no longer needed; the
// snapshots, rewards and prew0 this wave wrote to HBM are read back after a device-scope fence.
static_assert(sizeof(RvLds) <= sizeof(RcLds), "the vision tile must fit in the step kernel's LDS tile");
// What it needs of the state arrives as scalar arguments (25 registers), not as a reference to the struct: a by-reference struct
// has to exist in memory, i.e. the caller would write all 200-odd bytes of it to scratch in every lane before the call.
DE_OOL int rc_partial_obs_fused(uint64_t seed, int env_id_offset, int* envi, int n, int R, int noise_type, double noise_magn, RvSnap* snap,
                                int flags, double* prew0, double* epr, int E, double* epo, int e, float* __restrict__ obs,
                                double* __restrict__ rewards, int* seenPart, int budgetCycles) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  RcState S = RcState();  // (a local that never leaves registers: everything below is inlined)
  S.seed = uniform_u64(seed); S.env_id_offset = uniform_i(env_id_offset); S.envi = uniform_ptr(envi); S.n = uniform_i(n); S.R = uniform_i(R);
  S.noise_type = uniform_i(noise_type); S.noise_magn = uniform_d(noise_magn); S.snap = uniform_ptr(snap); S.flags = uniform_i(flags);
  S.prew0 = uniform_ptr(prew0); S.epr = uniform_ptr(epr); S.E = uniform_i(E); S.epo = uniform_ptr(epo);
  __threadfence();
  __syncthreads();
  // Passes in (snapshot, agent) order until `budgetCycles` from now have gone (<= 0: no limit).  All 5 R done: processSeens right
  // here.  Otherwise the seen counts of what was done go where the deferred launch puts its own (seenPart) and the number of
  // passes done is returned: the caller lists the environment for the deferred launch, rc_partial_finalize_kernel adds the parts up.
  budgetCycles = uniform_i(budgetCycles);
  const unsigned long long deadline = budgetCycles > 0 ? t0 + (unsigned long long)budgetCycles : 0ull;
  RvLds& V = *reinterpret_cast<RvLds*>(&g_R);
  const int lane = lane_id();
  e = uniform_i(e);
  const RvArgs va = rv_args(S, e);
  for (int i = lane; i < 10 * RCP_SEEN_STRIDE; i += DE_WAVE) V.seenSum[i] = 0;
  int ov = 0, done = 0;
#pragma unroll 1
  for (int t = 0; t < 5; ++t) {
    for (int i = lane; i < 10 * RCP_SEEN_STRIDE; i += DE_WAVE) V.seen[i] = 0;
    const int r = rv_snapshot(S, V, va, e, lane, t, uniform_ptr(obs), true, 0, S.R, deadline);
    const int aEnd = uniform_i(r >> 8);
    ov |= r & 1;
    int* part = uniform_ptr(seenPart) + ((size_t)e * 5 + t) * 10 * RCP_SEEN_STRIDE;
    for (int i = lane; i < aEnd * RCP_SEEN_STRIDE; i += DE_WAVE) { const int v = V.seen[i]; part[i] = v; V.seenSum[i] += v; }
    __syncthreads();
    done += aEnd;
    if (aEnd < S.R) break;
  }
  if (ov && lane == 0) S.envi[(size_t)e * RE_COUNT + RE_ERR] |= 8;
  if (done == 5 * S.R) rv_finalize(S, V.seenSum, e, lane, uniform_ptr(rewards));
  return done;
}

// The environments that held a contact through the step finish last; their 50 agent passes run by one lone, latency-bound
// wave (~26 us per pass) would sit on the launch's critical path.  They append themselves to deferList instead and this
// launch gives each of them one wave per (snapshot, agent); each leaves its agent's seen counts of its snapshot in
// seenPart.  The counts are integer sums, so adding the parts reproduces the sequential accumulation exactly;
// rc_partial_finalize_kernel does that and processSeens.  (The order of the list varies from run to run; nothing depends
// on it.)
#ifndef RC_DEFER_BLOCKS
#define RC_DEFER_BLOCKS 1024 /* round 6: 128 +4.4 %, 256 +1.4 %, 512 0, 1024 -0.65 %, 2048 -0.6 %, 4096 0 (RoboCup Partial step time) */
#endif
extern "C" __global__ void __launch_bounds__(64, 4)
rc_partial_obs_deferred_kernel(RcState S, float* __restrict__ obs) {
  const int t = blockIdx.y, a = blockIdx.z, lane = threadIdx.x;
  const int count = uniform_i(S.deferList[0]);
  RvLds& V = g_V;
  for (int k = blockIdx.x; k < count; k += gridDim.x) {
    const int entry = uniform_i(S.deferList[1 + k]), e = entry & 0xFFFFF;
    if (t * S.R + a < (entry >> 20)) continue;  // done in the step launch
    for (int i = lane; i < 10 * RCP_SEEN_STRIDE; i += DE_WAVE) V.seen[i] = 0;
    const int ov = rv_snapshot(S, V, rv_args(S, e), e, lane, t, obs, true, a, a + 1) & 1;
    if (ov && lane == 0) atomicOr(&S.envi[(size_t)e * RE_COUNT + RE_ERR], 8);
    int* part = S.seenPart + ((size_t)e * 5 + t) * 10 * RCP_SEEN_STRIDE + a * RCP_SEEN_STRIDE;
    if (lane < RCP_SEEN_STRIDE) part[lane] = V.seen[a * RCP_SEEN_STRIDE + lane];
    __syncthreads();
  }
}
extern "C" __global__ void __launch_bounds__(64)
rc_partial_finalize_kernel(RcState S, double* __restrict__ rewards) {
  const int lane = threadIdx.x;
  const int count = uniform_i(S.deferList[0]);
  RvLds& V = g_V;
  for (int k = blockIdx.x; k < count; k += gridDim.x) {
    const int e = uniform_i(S.deferList[1 + k]) & 0xFFFFF;
    const int* part = S.seenPart + (size_t)e * 5 * 10 * RCP_SEEN_STRIDE;
    for (int i = lane; i < 10 * RCP_SEEN_STRIDE; i += DE_WAVE) {
      int s = 0;
      for (int t = 0; t < 5; ++t) s += part[t * 10 * RCP_SEEN_STRIDE + i];
      V.seen[i] = s;
    }
    __syncthreads();
    rv_finalize(S, V.seen, e, lane, rewards);
    __syncthreads();
  }
}
#endif  // RC_PARTIAL_FUNCTIONS
