/* robocup_partial.c — ORACLE (test infrastructure): CPU restatement of RoboCupEnvironment.getAgentVision
 * (reference DynEnv/RoboCupEnvironment.py:1192-1316 detections / interactions / noise / false positives, :1536-1561
 * conversion) with cutils.isSeenInArea (:699-748), isLineInArea (:751-828), doesInteract (:546-566), addNoise (:417-469),
 * addNoiseLine (:352-379), convertToPolar (:308-315), normalize / normalizeAfterScale / scale / normalizeLine (:295-349).
 * processSeens (:1563-1575) is restated in robocup.c (rc_step).  One agent, scalar fp64.
 *
 * Output row of one (agent, snapshot) — dense, zero padded, float32, RCP_DIM floats:
 *   balls  [RCP_CAP_BALL ][5]  x, y, size, owned*team, agentIsClosest
 *   robots [RCP_CAP_ROB  ][7]  x, y, size, cos, sin, team*team_i, agent fallen|penalized
 *   goals  [RCP_CAP_GOAL ][6]  convertToPolar: dist, cos, sin, size, side*team, dir*team
 *   crosses[RCP_CAP_CROSS][6]
 *   fcross [RCP_CAP_FCROSS][8] convertToPolar + cos(a), -sin(a)
 *   lines  [RCP_CAP_LINE ][5]  dist, cos, sin, tx, ty
 *   tail: 6 list lengths, numLandMarks, ballsSeen, robotsSeen[9]   (the third element of the reference's return value)
 * (selfDets and circleDets are computed by the reference but not returned, :1561; the circle's noise draws feed nothing.)
 *
 * RNG: the reference draws from CPython's global `random` in program order, which thousands of parallel agents cannot
 * reproduce; as everywhere in this oracle each draw SITE has its own Philox word:
 *   dm_env_rng(seed, genv, episode, DM_RNG_OBS_NOISE, entity = agent | kind<<4 | index<<8 | block<<16, elapsed)
 *   kind 0 ball noise        block0 [nx, ny, falseNeg, misclass]   block1 [size]
 *   kind 1 robot noise (index = position in robDets)   kind 2 goalpost   kind 3 penalty cross (with misclass)
 *   kind 4 line cross        block0 [nx, ny, falseNeg, -]          block1 [size, angle]
 *   kind 5 line              block0 [n1x, n1y, n2x, n2y]           block1 [falseNeg]
 *   kind 6 ball->cross misclassification   block0 [randint, randint]
 *   kind 7 random false positive, index = trial  block0 [gate, class, dist, angle]  block1 [size, w1, w2, w3]
 *   kind 8 false-positive ball near robot, index = position in robDets  block0 [gate, hideRobot, ox, oy]  block1 [size]
 * tests/golden/gen_golden_robocup_partial.py serves exactly these words to the reference's own getAgentVision. */
#include <math.h>
#include <string.h>

#include "dynenv_math.h"
#include "robocup.h"
#include "robocup_partial.h"

enum { S_NONE = 0, S_PARTIAL = 1, S_DISTANT = 2, S_NORMAL = 3, S_MISCLASS = 4 };
enum { I_NONE = 0, I_NEARBY = 1, I_OCCLUDE = 2 };

#define FOV (DM_PI / 4.0)
#define MAXVIS0 ((RC_W * 0.4) * (RC_W * 0.4))
#define MAXVIS1 ((RC_W * 0.8) * (RC_W * 0.8))
#define TOTAL_RADIUS 17.5
#define BALL_RADIUS 5.0
#define PENALTY_RADIUS 5.0
#define GOALPOST_RADIUS 5.0
#define STD_NORM (2.0 / RC_W)
#define SIZE_NORM (10.0 / PENALTY_RADIUS)
#define SIDE 70.0

typedef struct { int seen, has; cpv p; double size, e3, e4, e5; } Det; /* [seen, rotPt, radius, extra...] */
typedef struct { int seen, has; cpv p1, p2; double tx, ty; } LineDet;

static cpv v_(double x, double y) { cpv r; r.x = x; r.y = y; return r; }
static cpv vsub_(cpv a, cpv b) { return v_(a.x - b.x, a.y - b.y); }
static cpv vadd_(cpv a, cpv b) { return v_(a.x + b.x, a.y + b.y); }
static cpv vmul_(cpv a, double s) { return v_(a.x * s, a.y * s); }
static double vcross_(cpv a, cpv b) { return a.x * b.y - a.y * b.x; }
static double vlensq_(cpv a) { return a.x * a.x + a.y * a.y; }
static double vlen_(cpv a) { return dm_sqrt(a.x * a.x + a.y * a.y); }
static cpv vrotated_(cpv v, double a) { /* Vec2d.rotated */
  double s, c;
  dm_sincos_f(a, &s, &c);
  return v_(v.x * c - v.y * s, v.x * s + v.y * c);
}
static double unit_(uint32_t u) { return dm_unit(u); }

typedef struct { const RoboCupEnv* e; int agent; uint32_t tkey; } VCtx; /* whose draws, and the time word of their keys */
static dm_u32x4 rng_(const VCtx* c, int kind, int index, int block) {
  uint32_t entity = (uint32_t)c->agent | ((uint32_t)kind << 4) | ((uint32_t)index << 8) | ((uint32_t)block << 16);
  return dm_env_rng(c->e->seed, c->e->genv, c->e->episode, DM_RNG_OBS_NOISE, entity, c->tkey);
}

/* cutils.isSeenInArea :699-748 (allowPartial=True: every caller that reaches the output uses the default) */
static Det seen_in_area(cpv point, cpv dir1, cpv dir2, double maxDist, double angle, double radius) {
  Det d;
  double dist1 = vcross_(dir1, point), dist2 = vcross_(dir2, point);
  memset(&d, 0, sizeof(d));
  d.size = radius;
  if (dist1 < radius && dist2 > -radius) {
    if (dist1 < -radius && dist2 > radius) d.seen = vlensq_(point) < maxDist ? S_NORMAL : S_DISTANT;
    else d.seen = S_PARTIAL;
    d.p = vrotated_(point, -angle);
    d.has = 1;
  }
  return d;
}

/* cutils.isLineInArea :751-828 */
static LineDet line_in_area(cpv p1, cpv p2, cpv dir1, cpv dir2, double maxDist, double angle) {
  LineDet L;
  double dist11 = vcross_(dir1, p1), dist12 = vcross_(dir1, p2);
  memset(&L, 0, sizeof(L));
  if (!(dist11 > 0.0 && dist12 > 0.0)) {
    double dist21 = vcross_(dir2, p1), dist22 = vcross_(dir2, p2);
    if (!(dist21 < 0.0 && dist22 < 0.0)) {
      cpv pt1, pt2;
      L.seen = S_NORMAL;
      if (dist11 <= 0.0 && dist21 >= 0.0) pt1 = p1;
      else {
        cpv d = vsub_(p2, p1);
        double inter1 = vcross_(p1, dir1) / (vcross_(dir1, d) + 1e-7);
        double inter2 = vcross_(p1, dir2) / (vcross_(dir2, d) + 1e-7);
        double inter = (inter1 < 1.0 && inter2 < 1.0) ? (inter1 > inter2 ? inter1 : (inter2 > inter1 ? inter2 : inter1))
                                                      : (inter1 < inter2 ? inter1 : (inter2 < inter1 ? inter2 : inter1));
        pt1 = vadd_(p1, vmul_(d, inter));
        L.seen = S_PARTIAL;
      }
      if (dist12 <= 0.0 && dist22 >= 0.0) pt2 = p2;
      else {
        cpv d = vsub_(p1, p2);
        double inter1 = vcross_(p2, dir1) / vcross_(dir1, d);
        double inter2 = vcross_(p2, dir2) / vcross_(dir2, d);
        double inter = (inter1 < 1.0 && inter2 < 1.0) ? (inter1 > inter2 ? inter1 : (inter2 > inter1 ? inter2 : inter1))
                                                      : (inter1 < inter2 ? inter1 : (inter2 < inter1 ? inter2 : inter1));
        pt2 = vadd_(p2, vmul_(d, inter));
        L.seen = S_PARTIAL;
      }
      if (vlensq_(pt1) > maxDist || vlensq_(pt2) > maxDist) L.seen = S_DISTANT;
      /* `if pt1 and pt2` (:813): both are set here; a Vec2d is truthy whatever its value under Python 3 (pymunk 5 defines the Python-2
       * `__nonzero__` only, tests/golden/gen_golden.py Vec2d), so the rotation is unconditional - also for a point exactly at the origin,
       * which is what a penalized robot on its side line gets from that line in every snapshot */
      pt1 = vrotated_(pt1, -angle);
      pt2 = vrotated_(pt2, -angle);
      if (pt1.x < 0.0 || pt2.x < 0.0) L.seen = S_NONE;
      L.p1 = pt1; L.p2 = pt2; L.has = 1;
    }
  }
  return L;
}

/* cutils.doesInteract :546-566 (obj None <=> !has) */
static int does_interact(const Det* o1, const Det* o2, double radius, int canOcclude) {
  int type = I_NONE;
  if (!o1->has || !o2->has) return I_NONE;
  if (vlen_(vsub_(o1->p, o2->p)) < radius) type = I_NEARBY;
  if (canOcclude) {
    double dist = vcross_(o1->p, o2->p) / vlen_(o1->p);
    if (fabs(dist) < radius && vlensq_(o1->p) < vlensq_(o2->p)) type = I_OCCLUDE;
  }
  return type;
}

/* cutils.addNoise :417-469; `ang` is obj[5] (only with angleNoise) */
static void add_noise(const VCtx* vc, int kind, int index, Det* o, int noiseType, int interaction,
                      double magn, double rand, double maxDist, int misClass, int angleNoise) {
  dm_u32x4 u0, u1;
  cpv noiseVec;
  if (interaction == I_OCCLUDE) { o->seen = S_NONE; return; }
  if (!o->seen) return;
  u0 = rng_(vc, kind, index, 0);
  u1 = rng_(vc, kind, index, 1);
  noiseVec = vmul_(v_(unit_(u0.v[0]) - 0.5, unit_(u0.v[1]) - 0.5), magn);
  if (noiseType == 0) { /* RANDOM */
    if (unit_(u0.v[2]) < rand) o->seen = S_NONE;
    o->p = vadd_(o->p, noiseVec);
    o->size *= (1.0 - (unit_(u1.v[0]) - 0.5) * 0.2);
    if (angleNoise) o->e5 += (unit_(u1.v[1]) - 0.5) * magn / 10.0;
  } else {
    int st = o->seen;
    double range = 0.25 + 3.75 * vlensq_(o->p) / maxDist, multiplier = range, diff;
    cpv newPos;
    if (interaction == I_NEARBY) multiplier = range * 2.0;
    if (st == S_DISTANT) multiplier = range * 3.0;
    else if (st == S_PARTIAL) multiplier = range * 4.0;
    newPos = vadd_(o->p, v_(noiseVec.x * multiplier / 4.0, noiseVec.y * multiplier / 4.0));
    diff = vlen_(newPos) - vlen_(o->p);
    if (unit_(u0.v[2]) < rand * multiplier) st = S_NONE;
    if (misClass && unit_(u0.v[3]) < rand * multiplier / 2.0) st = S_MISCLASS;
    o->seen = st;
    o->p = newPos;
    o->size *= 1.0 + (unit_(u1.v[0]) * 0.1 * diff);
    if (angleNoise) o->e5 += (unit_(u1.v[1]) - 0.5) * magn * multiplier / 180.0;
  }
}

/* cutils.addNoiseLine :352-379 */
static void add_noise_line(const VCtx* vc, int index, LineDet* L, int noiseType, double magn, double rand, double maxDist) {
  dm_u32x4 u0, u1;
  cpv n1, n2;
  if (!L->seen) return;
  u0 = rng_(vc, 5, index, 0);
  u1 = rng_(vc, 5, index, 1);
  n1 = vmul_(v_(unit_(u0.v[0]) - 0.5, unit_(u0.v[1]) - 0.5), magn);
  n2 = vmul_(v_(unit_(u0.v[2]) - 0.5, unit_(u0.v[3]) - 0.5), magn);
  if (noiseType == 0) {
    if (unit_(u1.v[0]) < rand) L->seen = S_NONE;
    L->p1 = vadd_(L->p1, n1);
    L->p2 = vadd_(L->p2, n2);
  } else {
    double m1 = 0.25 + 3.75 * vlensq_(L->p1) / maxDist, m2 = 0.25 + 3.75 * vlensq_(L->p2) / maxDist;
    double m = (m1 + m2) * 0.5;
    if (unit_(u1.v[0]) < rand * m) L->seen = S_NONE;
    L->p1 = vadd_(L->p1, v_(n1.x * m1 / 2.0, n1.y * m1 / 2.0));
    L->p2 = vadd_(L->p2, v_(n2.x * m2 / 2.0, n2.y * m2 / 2.0));
  }
}

static double scale_(double val, double norm) { return ((val * norm) - 0.5) / 0.5; }          /* cutils.scale :295 */
static double normalize_(double pt, double nf) { return ((pt * nf) - 0.0) * 2.0 * 1.0; }        /* :318-323 */
static double norm_after_scale_(double pt, double nf, double mean) { return (pt - mean) * nf * 1.0; } /* :326-331 */

static void polar_(const Det* d, double sizeMean, int team, float* o) { /* convertToPolar :308-315 */
  double dist = dm_sqrt(d->p.x * d->p.x + d->p.y * d->p.y);
  double ang = dm_atan2(d->p.y * (double)team, d->p.x * (double)team), s, c;
  dm_sincos_f(ang, &s, &c);
  o[0] = (float)scale_(dist, STD_NORM); o[1] = (float)c; o[2] = (float)s;
  o[3] = (float)((d->size - sizeMean) * SIZE_NORM); o[4] = (float)(d->e3 * (double)team); o[5] = (float)(d->e4 * (double)team);
}

/* the scene (RoboCupEnvironment._create_football_field :169-236, _create_goalposts :296-302) */
void rcp_scene(cpv lines[11][2], double lineT[11][2], cpv crosses[3], double crossT[3][2], cpv fcross[16], double fcrossT[16][2],
               cpv posts[4], double postT[4][2]) {
  const double W = RC_W, H = RC_H, s = SIDE, pl = PENALTY_LENGTH_, pw = PENALTY_WIDTH_, cr = 75.0, pd = 130.0, gw = 80.0;
  int i = 0;
#define LN(ax, ay, bx, by, tx, ty) do { lines[i][0] = v_(ax, ay); lines[i][1] = v_(bx, by); lineT[i][0] = tx; lineT[i][1] = ty; ++i; } while (0)
  LN(s, s, s, H - s, 1, 0); LN(W - s, s, W - s, H - s, -1, 0); LN(s, s, W - s, s, 0, 1); LN(s, H - s, W - s, H - s, 0, -1);
  LN(W / 2, s, W / 2, H - s, 0, 0);
  LN(s, H / 2 - pw, s + pl, H / 2 - pw, 1, 0.37); LN(s, H / 2 + pw, s + pl, H / 2 + pw, 1, -0.37);
  LN(s + pl, H / 2 - pw, s + pl, H / 2 + pw, 0.87, 0);
  LN(W - s - pl, H / 2 - pw, W - s, H / 2 - pw, -1, 0.37); LN(W - s - pl, H / 2 + pw, W - s, H / 2 + pw, -1, -0.37);
  LN(W - s - pl, H / 2 - pw, W - s - pl, H / 2 + pw, -0.87, 0);
#undef LN
  crosses[0] = v_(520.0, 370.0); crossT[0][0] = 0; crossT[0][1] = 0;      /* W // 2, H // 2 */
  crosses[1] = v_(s + pd, 370.0); crossT[1][0] = 1; crossT[1][1] = 0;
  crosses[2] = v_(W - (s + pd), 370.0); crossT[2][0] = -1; crossT[2][1] = 0;
  i = 0;
#define FC(x, y, tx, ty) do { fcross[i] = v_(x, y); fcrossT[i][0] = tx; fcrossT[i][1] = ty; ++i; } while (0)
  FC(s, s, 1, 1); FC(s, H - s, 1, -1); FC(W - s, s, -1, 1); FC(W - s, H - s, -1, -1);
  FC(W / 2, s, 0, 1); FC(W / 2, H - s, 0, -1);
  FC(W / 2, H / 2 - cr * 2, 0, 0.5); FC(W / 2, H / 2 + cr * 2, 0, -0.5);
  FC(s, H / 2 - pw, 1, 0.37); FC(s, H / 2 + pw, 1, -0.37); FC(s + pl, H / 2 - pw, 0.87, 0.37); FC(s + pl, H / 2 + pw, 0.87, -0.37);
  FC(W - s, H / 2 - pw, -1, 0.37); FC(W - s, H / 2 + pw, -1, -0.37); FC(W - s - pl, H / 2 - pw, -0.87, 0.37); FC(W - s - pl, H / 2 + pw, -0.87, -0.37);
#undef FC
  posts[0] = v_(s, H / 2 + gw); postT[0][0] = 1; postT[0][1] = -0.27;
  posts[1] = v_(s, H / 2 - gw); postT[1][0] = 1; postT[1][1] = 0.27;
  posts[2] = v_(W - s, H / 2 + gw); postT[2][0] = -1; postT[2][1] = -0.27;
  posts[3] = v_(W - s, H / 2 - gw); postT[3][0] = -1; postT[3][1] = 0.27;
}

int rc_partial_obs_dim(void) { return RCP_DIM; }

int rc_agent_vision(const RoboCupEnv* e, int agentIdx, int noiseType, double magn, uint32_t tkey, float* out) {
  const Robot* ag = &e->robots[agentIdx];
  VCtx vctx, *vc = &vctx;
  const int R = e->nRobots, team = ag->team;
  const double randBase = 0.01 * magn;
  cpv lines[11][2], crossP[3], fcrossP[16], postP[4];
  double lineT[11][2], crossT[3][2], fcrossT[16][2], postT[4][2];
  Det ball[RCP_CAP_BALL], rob[RCP_CAP_ROB], goal[RCP_CAP_GOAL], cross[RCP_CAP_CROSS], fcr[RCP_CAP_FCROSS];
  LineDet line[11];
  int nBall = 1, nRob = 0, nGoal = 4, nCross = 3, nFc = 16, nLine = 11, i, j, overflow = 0;
  int robRob[RC_MAX_ROBOTS], robBall = 0, robPost[4], robCross[3], robFc[16], ballPost = 0, ballCross[3];
  int robotsSeen[RC_MAX_ROBOTS], ballsSeen, numLandMarks;
  cpv pos = rc_robot_pos(ag), vec1, vec2;
  const double angle = (ag->leftBody.a + ag->rightBody.a) / 2.0, headAngle = angle + ag->headAngle;
  vctx.e = e; vctx.agent = agentIdx; vctx.tkey = tkey;
  rcp_scene(lines, lineT, crossP, crossT, fcrossP, fcrossT, postP, postT);
  memset(out, 0, sizeof(float) * RCP_DIM);
  vec1 = vrotated_(v_(1.0, 0.0), headAngle + FOV);
  vec2 = vrotated_(v_(1.0, 0.0), headAngle - FOV);
  /* ---- detections :1209-1226 */
  ball[0] = seen_in_area(vsub_(e->ballBody.p, pos), vec1, vec2, MAXVIS0, headAngle, BALL_RADIUS * 2.0);
  ball[0].e3 = (double)(e->ballOwned * team);
  for (i = 0; i < R; ++i) {
    const Robot* r = &e->robots[i];
    if (i == agentIdx) continue;
    rob[nRob] = seen_in_area(vsub_(rc_robot_pos(r), pos), vec1, vec2, MAXVIS1, headAngle, TOTAL_RADIUS);
    rob[nRob].e3 = (r->leftBody.a + r->rightBody.a) / 2.0 - headAngle;
    rob[nRob].e4 = (double)(team * r->team);
    rob[nRob].e5 = (double)((ag->fallen || ag->penalized) ? 1 : 0);
    ++nRob;
  }
  for (i = 0; i < 4; ++i) { goal[i] = seen_in_area(vsub_(postP[i], pos), vec1, vec2, MAXVIS1, headAngle, GOALPOST_RADIUS); goal[i].e3 = postT[i][0]; goal[i].e4 = postT[i][1]; }
  for (i = 0; i < 3; ++i) { cross[i] = seen_in_area(vsub_(crossP[i], pos), vec1, vec2, MAXVIS0, headAngle, PENALTY_RADIUS); cross[i].e3 = crossT[i][0]; cross[i].e4 = crossT[i][1]; }
  for (i = 0; i < 16; ++i) {
    fcr[i] = seen_in_area(vsub_(fcrossP[i], pos), vec1, vec2, MAXVIS0, headAngle, PENALTY_RADIUS);
    fcr[i].e3 = fcrossT[i][0]; fcr[i].e4 = fcrossT[i][1]; fcr[i].e5 = 0.0 - headAngle;
  }
  for (i = 0; i < 11; ++i) { line[i] = line_in_area(vsub_(lines[i][0], pos), vsub_(lines[i][1], pos), vec1, vec2, MAXVIS1, headAngle); line[i].tx = lineT[i][0]; line[i].ty = lineT[i][1]; }
  /* ---- interactions :1228-1243 */
  for (j = 0; j < nRob; ++j) {
    int m = 0;
    if (e->nPlayers > 1) for (i = 0; i < nRob; ++i) if (i != j) { int t = does_interact(&rob[i], &rob[j], TOTAL_RADIUS * 2.0, 1); m = t > m ? t : m; }
    robRob[j] = m;
  }
  for (i = 0; i < nRob; ++i) { int t = does_interact(&rob[i], &ball[0], TOTAL_RADIUS * 2.0, 1); robBall = t > robBall ? t : robBall; }
  for (j = 0; j < 4; ++j) { int m = 0; for (i = 0; i < nRob; ++i) { int t = does_interact(&rob[i], &goal[j], TOTAL_RADIUS * 2.0, 1); m = t > m ? t : m; } robPost[j] = m; }
  for (j = 0; j < 3; ++j) { int m = 0; for (i = 0; i < nRob; ++i) { int t = does_interact(&rob[i], &cross[j], TOTAL_RADIUS * 2.0, 1); m = t > m ? t : m; } robCross[j] = m; }
  for (j = 0; j < 16; ++j) { int m = 0; for (i = 0; i < nRob; ++i) { int t = does_interact(&rob[i], &fcr[j], TOTAL_RADIUS * 2.0, 1); m = t > m ? t : m; } robFc[j] = m; }
  for (j = 0; j < 4; ++j) { int t = does_interact(&ball[0], &goal[j], BALL_RADIUS * 8.0, 0); ballPost = t > ballPost ? t : ballPost; }
  for (j = 0; j < 3; ++j) ballCross[j] = does_interact(&ball[0], &cross[j], BALL_RADIUS * 4.0, 0);
  /* ---- noise :1245-1259 */
  add_noise(vc, 0, 0, &ball[0], noiseType, robBall > ballPost ? robBall : ballPost, magn, randBase, MAXVIS0, 1, 0);
  for (i = 0; i < nRob; ++i) add_noise(vc, 1, i, &rob[i], noiseType, robRob[i], magn, randBase, MAXVIS1, 0, 0);
  for (i = 0; i < 4; ++i) add_noise(vc, 2, i, &goal[i], noiseType, robPost[i], magn, randBase, MAXVIS1, 0, 0);
  for (i = 0; i < 3; ++i) add_noise(vc, 3, i, &cross[i], noiseType, robCross[i] > ballCross[i] ? robCross[i] : ballCross[i], magn, randBase, MAXVIS0, 1, 0);
  for (i = 0; i < 16; ++i) add_noise(vc, 4, i, &fcr[i], noiseType, robFc[i], magn, randBase, MAXVIS0, 0, 1);
  for (i = 0; i < 11; ++i) add_noise_line(vc, i, &line[i], noiseType, magn, randBase, MAXVIS1);
  for (i = 0; i < nRob; ++i) robotsSeen[i] = rob[i].seen != S_NONE;                                   /* :1261 */
  ballsSeen = ball[0].seen != S_NONE && ball[0].seen != S_MISCLASS;                                   /* :1262 */
  /* ---- misclassification swaps :1264-1270 */
  if (ball[0].seen == S_MISCLASS) {
    dm_u32x4 u = rng_(vc, 6, 0, 0);
    Det c = ball[0];
    c.seen = S_NORMAL; c.e3 = (double)dm_randint(u.v[0], -1, 1); c.e4 = (double)dm_randint(u.v[1], -1, 1);
    cross[nCross++] = c;
  }
  for (i = 0; i < nCross; ++i) if (cross[i].seen == S_MISCLASS) { Det b = cross[i]; b.seen = S_NORMAL; b.e3 = 0.0; ball[nBall++] = b; }
  /* ---- filters :1272-1282 */
#define FILTER(arr, n, cond) do { int w_ = 0, k_; for (k_ = 0; k_ < (n); ++k_) if (cond(arr[k_].seen)) arr[w_++] = arr[k_]; (n) = w_; } while (0)
#define KEEP_SEEN(s) ((s) != S_NONE)
#define KEEP_CLASS(s) ((s) != S_NONE && (s) != S_MISCLASS)
  FILTER(ball, nBall, KEEP_CLASS); FILTER(rob, nRob, KEEP_SEEN); FILTER(goal, nGoal, KEEP_SEEN); FILTER(cross, nCross, KEEP_CLASS);
  FILTER(fcr, nFc, KEEP_CLASS); FILTER(line, nLine, KEEP_SEEN);
#undef FILTER
  numLandMarks = nFc + nLine + nCross + nGoal;                                                        /* :1284 */
  /* ---- random false positives :1286-1316 */
  for (i = 0; i < 10; ++i) {
    dm_u32x4 u = rng_(vc, 7, i, 0), u1 = rng_(vc, 7, i, 1);
    if (unit_(u.v[0]) < randBase) {
      const int c = dm_randint(u.v[1], 0, 5);
      const double d = unit_(u.v[2]) * dm_sqrt(MAXVIS1);
      const double a = unit_(u.v[3]) * 2.0 * FOV - FOV;
      Det f;
      memset(&f, 0, sizeof(f));
      f.seen = S_NORMAL; f.has = 1; f.p = vrotated_(v_(d, 0.0), a);
      if (c == 0) { f.size = BALL_RADIUS * 2.0 * (1.0 - 0.4 * (unit_(u1.v[0]) - 0.5)); f.e3 = 0.0; if (nBall < RCP_CAP_BALL) ball[nBall++] = f; else overflow = 1; }
      else if (c == 1) {
        f.size = TOTAL_RADIUS * (1.0 - 0.4 * (unit_(u1.v[0]) - 0.5));
        f.e3 = (unit_(u1.v[1]) - 0.5) * 2.0 * DM_PI; f.e4 = (unit_(u1.v[2]) > 0.5) ? -1.0 : 1.0; f.e5 = (unit_(u1.v[3]) > 0.9) ? 1.0 : 0.0;
        if (nRob < RCP_CAP_ROB) rob[nRob++] = f; else overflow = 1;
      } else if (c == 2) {
        f.size = GOALPOST_RADIUS * (1.0 - 0.4 * (unit_(u1.v[0]) - 0.5)); f.e3 = (double)dm_randint(u1.v[1], -1, 1); f.e4 = (double)dm_randint(u1.v[2], -1, 1);
        if (nGoal < RCP_CAP_GOAL) goal[nGoal++] = f; else overflow = 1;
      } else if (c == 3) {
        f.size = PENALTY_RADIUS * (1.0 - 0.4 * (unit_(u1.v[0]) - 0.5)); f.e3 = (double)dm_randint(u1.v[1], -1, 1); f.e4 = (double)dm_randint(u1.v[2], -1, 1);
        if (nCross < RCP_CAP_CROSS) cross[nCross++] = f; else overflow = 1;
      } else if (c == 4) {  /* fieldCrossDets.insert(len(crossDets), ...): position = current length of the CROSS list */
        int at = nCross < nFc ? nCross : nFc, k;
        f.size = PENALTY_RADIUS * (1.0 - 0.4 * (unit_(u1.v[0]) - 0.5)); f.e3 = (double)dm_randint(u1.v[1], -1, 1); f.e4 = (double)dm_randint(u1.v[2], -1, 1);
        f.e5 = unit_(u1.v[3]) * DM_PI * 2.0;
        if (nFc < RCP_CAP_FCROSS) { for (k = nFc; k > at; --k) fcr[k] = fcr[k - 1]; fcr[at] = f; ++nFc; } else overflow = 1;
      }
    }
  }
  /* ---- false-positive balls near robots :1318-1327 (REALISTIC only) */
  if (noiseType == 1) {
    for (i = 0; i < nRob; ++i) {
      dm_u32x4 u, u1;
      if (rob[i].seen != S_NORMAL) continue;
      u = rng_(vc, 8, i, 0); u1 = rng_(vc, 8, i, 1);
      if (unit_(u.v[0]) < randBase * 10.0 && vlen_(rob[i].p) < 250.0) {
        Det f;
        memset(&f, 0, sizeof(f));
        if (unit_(u.v[1]) < randBase * 8.0) rob[i].seen = S_NONE; /* stays in the list: the conversion ignores [0] */
        f.seen = S_NORMAL; f.has = 1;
        f.p = vadd_(rob[i].p, vmul_(v_(2.0 * unit_(u.v[2]) - 1.0, 2.0 * unit_(u.v[3]) - 1.0), TOTAL_RADIUS));
        f.size = BALL_RADIUS * 2.0 * (1.0 - 0.4 * (unit_(u1.v[0]) - 0.5)); f.e3 = 0.0;
        if (nBall < RCP_CAP_BALL) ball[nBall++] = f; else overflow = 1;
      }
    }
  }
  /* ---- conversion :1536-1561 */
  {
    const int closest = (ag->id == e->closestID[0] || ag->id == e->closestID[1]) ? 1 : 0;
    float* o = out + RCP_OFF_BALL;
    for (i = 0; i < nBall; ++i, o += 5) {
      o[0] = (float)normalize_(ball[i].p.x, STD_NORM); o[1] = (float)normalize_(ball[i].p.y, STD_NORM);
      o[2] = (float)norm_after_scale_(ball[i].size, SIZE_NORM, BALL_RADIUS * 2.0); o[3] = (float)ball[i].e3; o[4] = (float)closest;
    }
    o = out + RCP_OFF_ROB;
    for (i = 0; i < nRob; ++i, o += 7) {
      double s, c;
      dm_sincos_f(rob[i].e3, &s, &c);
      o[0] = (float)normalize_(rob[i].p.x, STD_NORM); o[1] = (float)normalize_(rob[i].p.y, STD_NORM);
      o[2] = (float)norm_after_scale_(rob[i].size, SIZE_NORM, TOTAL_RADIUS); o[3] = (float)c; o[4] = (float)s;
      o[5] = (float)rob[i].e4; o[6] = (float)rob[i].e5;
    }
    o = out + RCP_OFF_GOAL;
    for (i = 0; i < nGoal; ++i, o += 6) polar_(&goal[i], GOALPOST_RADIUS, team, o);
    o = out + RCP_OFF_CROSS;
    for (i = 0; i < nCross; ++i, o += 6) polar_(&cross[i], PENALTY_RADIUS, team, o);
    o = out + RCP_OFF_FCROSS;
    for (i = 0; i < nFc; ++i, o += 8) {
      double s, c;
      polar_(&fcr[i], PENALTY_RADIUS, team, o);
      dm_sincos_f(fcr[i].e5, &s, &c);
      o[6] = (float)c; o[7] = (float)(-s);
    }
    o = out + RCP_OFF_LINE;
    for (i = 0; i < nLine; ++i, o += 5) { /* normalizeLine :333-349 */
      cpv diff = vsub_(line[i].p2, line[i].p1);
      double dist = fabs(line[i].p2.x * line[i].p1.y - line[i].p2.y * line[i].p1.x) / (vlen_(diff) + 1e-7);
      double ang = dm_atan2(diff.y, diff.x), s, c;
      dm_sincos_f(ang, &s, &c);
      o[0] = (float)scale_(dist, STD_NORM); o[1] = (float)c; o[2] = (float)s; o[3] = (float)line[i].tx; o[4] = (float)line[i].ty;
    }
    o = out + RCP_OFF_TAIL;
    o[0] = (float)nBall; o[1] = (float)nRob; o[2] = (float)nGoal; o[3] = (float)nCross; o[4] = (float)nFc; o[5] = (float)nLine;
    o[6] = (float)numLandMarks; o[7] = (float)ballsSeen;
    for (i = 0; i < R - 1; ++i) o[8 + i] = (float)robotsSeen[i];
  }
  return overflow;
}

int rc_write_partial_obs(RoboCupEnv* e, uint32_t tkey, float* out) {
  int a, ov = 0;
  for (a = 0; a < e->nRobots; ++a) ov |= rc_agent_vision(e, a, e->noiseType, e->noiseMagnitude, tkey, out + (size_t)a * RCP_DIM);
  if (ov) e->obsOverflow = 1;
  return ov;
}
