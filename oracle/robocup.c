/* robocup.c — ORACLE (test infrastructure; never on the product path).
 * CPU restatement of DynEnv/RoboCupEnvironment.py step() with Full observations (+ Robot.py, Ball.py, Goalpost.py,
 * cutils.py friction_robot / friction_ball / normalizeAfterScale).  Each function cites the reference lines.
 * RNG: the serial CPython `random` stream is replaced by Philox blocks keyed (seed, env, episode, purpose, entity, t):
 *   processAction fall dice  -> entity = robot id,            words 0/1/2 = move/turn/kick      (:557,565,577)
 *   getup dice (tick)        -> entity = robot id | 1<<8,     word 0                             (:932)
 *   robotCollision dice      -> entity = pair key | 2<<16,    words 0/1                          (:1064,1068)
 *   goalpostCollision dice   -> entity = pair key | 3<<16,    word 0                             (:1121)
 * Physics underneath is the unpinned Chipmunk restatement (cp_lite); game logic pinned by tests/golden/robocup_*. */
#include "robocup.h"
#include "robocup_partial.h"

#include <math.h>
#include <string.h>

#include "dynenv_math.h"
#include "driving.h" /* apply_friction, CT_* */

#define ROBOT_LENGTH 10.0      /* Robot.py:13-20 */
#define ROBOT_RADIUS 7.5
#define ROBOT_TOTAL_RADIUS 17.5
#define ROBOT_VELOCITY 50.0
#define ROBOT_ANG_VELOCITY 20.0
#define ROBOT_MASS 4000.0
#define ROBOT_HEAD_MAX (2.0 * DM_PI / 3.0)
#define TIME_STEP_MS 10.0      /* 1000 / timeStep */
#define PENALTY_LENGTH 60.0
#define PENALTY_WIDTH 110.0
#define LINE_WIDTH 5.0
#define GOAL_WIDTH 80.0
#define BALL_RADIUS 5.0
#define FIELD_W 900.0
#define FIELD_H 600.0
#define CENTER_CIRCLE_RADIUS 75.0
#define KICK_DISCOUNT 0.5

static inline double vlen(cpv v) { return dm_sqrt(v.x * v.x + v.y * v.y); }
static inline cpv vrotated(cpv v, double a) {
  double s, c;
  dm_sincos(a, &s, &c);
  return cpv_(v.x * c - v.y * s, v.x * s + v.y * c);
}

static void friction_robot(cpBody* b, cpv g, double d, double dt) { apply_friction(b, g, d, dt, 1e-3, 1e-2, 0.0); }   /* cutils.py:93-94 */
static void friction_ball(cpBody* b, cpv g, double d, double dt) { apply_friction(b, g, d, dt, 2.8e-2, 1e-3, 5e-2); } /* cutils.py:98-99 */

cpv rc_robot_pos(const Robot* r) { /* Robot.getPos :90-91 */
  cpv s = cpvadd(r->leftBody.p, r->rightBody.p);
  return cpv_(s.x / 2.0, s.y / 2.0);
}
static double robot_angle(const Robot* r) { return (r->leftBody.a + r->rightBody.a) / 2.0; } /* Robot.getAngle :94-100 */

static dm_u32x4 rc_rng(const RoboCupEnv* e, uint32_t entity) {
  return dm_env_rng(e->seed, e->genv, e->episode, DM_RNG_ROBO_STEP, entity, (uint32_t)e->elapsed);
}
static uint32_t pair_entity(const cpArbiter* arb, uint32_t kind) {
  int sa = arb->a->slot, sb = arb->b->slot;
  uint32_t key = sa < sb ? (uint32_t)(sa * 32 + sb) : (uint32_t)(sb * 32 + sa);
  return key | (kind << 16);
}
static int in_last_kicked(const RoboCupEnv* e, int id) {
  int i;
  for (i = 0; i < e->nLastKicked; ++i) if (e->lastKicked[i] == id) return 1;
  return 0;
}
static void push_last_kicked(RoboCupEnv* e, int id) { /* [id] + lastKicked, truncated to 4 */
  int i, n = e->nLastKicked < 4 ? e->nLastKicked : 3;
  for (i = n; i > 0; --i) e->lastKicked[i] = e->lastKicked[i - 1];
  e->lastKicked[0] = id;
  e->nLastKicked = n + 1;
}

/* ------------------------------------------------------------------ getFreePenaltySpot :792-821, penalize :824-859 */
static void free_penalty_spot(const RoboCupEnv* e, const Robot* robot, cpv* spot, double* angle) {
  double y = e->ballBody.p.y;
  int lower = !(y > RC_H / 2.0); /* availableSpots[:7] if y > H/2 else [7:] */
  int k, j, sel = 0;
  *angle = (y < RC_H / 2.0) ? -DM_PI / 2.0 : DM_PI / 2.0;
  for (k = 0; k < 7; ++k) {
    double sx = (robot->team > 0) ? RC_SIDE + (double)(k + 1) * ROBOT_TOTAL_RADIUS * 3.0
                                  : RC_W - RC_SIDE - (double)(k + 1) * ROBOT_TOTAL_RADIUS * 3.0;
    double sy = lower ? RC_H - RC_SIDE : RC_SIDE;
    int available = 1;
    for (j = 0; j < e->nRobots; ++j) {
      const Robot* rob = &e->robots[j];
      cpv p;
      if (rob == robot) continue;
      p = rc_robot_pos(rob);
      if (vlen(cpv_(sx - p.x, sy - p.y)) < ROBOT_TOTAL_RADIUS * 3.0) { available = 0; break; }
    }
    if (available) { sel = k; break; }
  }
  spot->x = (robot->team > 0) ? RC_SIDE + (double)(sel + 1) * ROBOT_TOTAL_RADIUS * 3.0
                              : RC_W - RC_SIDE - (double)(sel + 1) * ROBOT_TOTAL_RADIUS * 3.0;
  spot->y = lower ? RC_H - RC_SIDE : RC_SIDE;
}

void rc_penalize(RoboCupEnv* e, Robot* robot) {
  int teamIdx = robot->team > 0 ? 0 : 1;
  cpv pos; double angle;
  robot->penalized = 1;
  robot->penalTime = e->penalTimes[teamIdx];
  e->robotRewards[robot->id] -= e->penalTimes[teamIdx] / 2000.0;
  e->penalTimes[teamIdx] += 10000.0;
  free_penalty_spot(e, robot, &pos, &angle);
  robot->leftBody.p = pos; cpBodySetAngle(&robot->leftBody, angle);
  robot->rightBody.p = pos; cpBodySetAngle(&robot->rightBody, angle);
  robot->leftBody.v = cpv_(0.0, 0.0); robot->leftBody.w = 0.0;
  robot->rightBody.v = cpv_(0.0, 0.0); robot->rightBody.w = 0.0;
  if (robot->kicking && robot->jointRemoved) {
    robot->kicking = 0;
    cpSpaceAddConstraint(&e->space, &robot->joint);
    robot->jointRemoved = 0;
  }
}

/* ------------------------------------------------------------------ fall :735-791 */
void rc_fall(RoboCupEnv* e, Robot* robot, int punish) {
  cpv pos = rc_robot_pos(robot);
  cpShape* shapes[CP_MAX_SHAPES];
  int n = cpSpacePointQuery(&e->space, pos, 40.0, shapes, CP_MAX_SHAPES), k;
  if (punish) e->robotRewards[robot->id] -= 2.0;
  for (k = 0; k < n; ++k) { /* canonical slot order (pymunk's query order is unspecified) */
    cpShape* sh = shapes[k];
    double force, len;
    cpv dp;
    if (sh == &robot->leftFoot || sh == &robot->rightFoot) continue;
    if (sh->body->type == CP_BODY_STATIC) continue; /* goalposts: infinite mass, forces never integrated */
    force = ROBOT_VELOCITY * robot->leftBody.m * sh->body->m / 50.0;
    dp = cpvsub(pos, sh->body->p);
    len = vlen(dp);
    dp = cpv_(-dp.x * force / len, -dp.y * force / len);
    cpBodyApplyForceAtWorldPoint(sh->body, dp, pos);
    if (sh == &e->ballShape) {
      if (e->nLastKicked && !in_last_kicked(e, robot->id)) push_last_kicked(e, robot->id);
      if (e->ballOwned != 0) { e->gracePeriod = 0.0; e->ballFreeCntr = 0.0; e->ballOwned = 0; }
    }
  }
  robot->fallen = 1;
  robot->fallCntr += 1;
  robot->fallTime = 4000.0;
  if (robot->fallCntr > 2) rc_penalize(e, robot);
}

/* ------------------------------------------------------------------ Robot.step/turn/kick/turnHead :103-138 */
static void robot_step(Robot* r, int dir) {
  cpv velocity; int has = 1;
  if (r->kicking || r->penalized || r->fallen) return;
  r->moveTime = 500.0;
  if (dir == 0) velocity = cpv_(0.0, 2.0 * ROBOT_VELOCITY);
  else if (dir == 1) velocity = cpv_(0.0, -2.0 * ROBOT_VELOCITY);
  else if (dir == 2) velocity = cpv_(2.5 * ROBOT_VELOCITY, 0.0);
  else if (dir == 3) velocity = cpv_(-2.0 * ROBOT_VELOCITY, 0.0);
  else { has = 0; velocity = cpv_(0.0, 0.0); }
  if (has) r->leftBody.v = vrotated(velocity, r->leftBody.a);
}
static void robot_turn(Robot* r, int dir) {
  if (r->kicking || r->penalized || r->fallen) return;
  r->moveTime = 500.0;
  r->leftBody.w += dir ? ROBOT_ANG_VELOCITY : -ROBOT_ANG_VELOCITY;
}
static void robot_kick(Robot* r, int foot) {
  if (r->kicking || r->penalized || r->fallen) return;
  r->foot = foot;
  r->initPos = foot ? r->rightBody.p : r->leftBody.p;
  r->kicking = 1;
  r->moveTime = 1000.0;
}

/* ------------------------------------------------------------------ processAction :527-581 */
void rc_process_action(RoboCupEnv* e, Robot* robot, const int32_t* action) {
  int move = action[0], turn = action[1], kick = action[2];
  double head = (double)action[3]; /* a float with allowHeadTurn (Box(-3, 3)), an int otherwise: exact either way */
  int canMove;
  dm_u32x4 u = rc_rng(e, (uint32_t)robot->id);
  if (e->allowHeadTurn && e->headActions) head = e->headActions[robot->id];
  if (e->deterministicTurn) head = (double)(-3 * robot->team); /* :529-530 */
  if (!e->allowHeadTurn) head -= 3.0;
  /* the reference raises on a malformed action (:543-550); the batched step cannot: the robot keeps still (the HIP kernel also
   * raises its environment's error flag) */
  if (move < 0 || move > 4 || turn < 0 || turn > 2 || kick < 0 || kick > 2 || !(dm_abs(head) <= 6.0)) return;
  canMove = !(robot->penalized || robot->kicking || robot->fallen);
  if (move > 0 && canMove) {
    double r = e->canFall ? dm_unit(u.v[0]) : 0.0;
    if (r > 0.999) { rc_fall(e, robot, 0); return; }
    robot_step(robot, move - 1);
  }
  if (turn > 0 && canMove) {
    double r = e->canFall ? dm_unit(u.v[1]) : 0.0;
    if (r > 0.999) { rc_fall(e, robot, 0); return; }
    robot_turn(robot, turn - 1);
  }
  if (head != 0.0) { robot->headMoving = head * DM_PI / 720.0; robot->moveTime = 500.0; } /* turnHead :136-138 */
  if (kick > 0 && move == 0 && turn == 0 && canMove) {
    double r = e->canFall ? dm_unit(u.v[2]) : 0.0;
    if (r > 0.99) { rc_fall(e, robot, 0); return; }
    robot_kick(robot, kick - 1);
  }
}

/* ------------------------------------------------------------------ tick :862-1007 */
static int defenders_find(const RoboCupEnv* e, int t, int id) {
  int i;
  for (i = 0; i < e->nDefenders[t]; ++i) if (e->defenders[t][i] == id) return i;
  return -1;
}

void rc_tick(RoboCupEnv* e, Robot* robot) {
  const double time = TIME_STEP_MS;
  cpv pos;
  if (robot->moveTime > 0.0) {
    robot->moveTime -= time;
    if (robot->headMoving != 0.0) {
      robot->headAngle += robot->headMoving;
      robot->headAngle = dm_max(-ROBOT_HEAD_MAX, dm_min(ROBOT_HEAD_MAX, robot->headAngle));
    }
    if (robot->kicking) {
      cpBody* foot = robot->foot ? &robot->rightBody : &robot->leftBody;
      if (robot->moveTime + time > 500.0 && robot->moveTime <= 500.0) {
        if (!robot->jointRemoved) { cpSpaceRemoveConstraint(&e->space, &robot->joint); robot->jointRemoved = 1; }
        foot->v = vrotated(cpv_(ROBOT_VELOCITY * 3.0, 0.0), foot->a);
      }
      if (robot->moveTime + time > 400.0 && robot->moveTime <= 400.0) {
        foot->v = cpvneg(vrotated(cpv_(ROBOT_VELOCITY * 2.5, 0.0), foot->a));
      } else if (robot->moveTime <= 300.0) {
        foot->v = cpv_(0.0, 0.0);
        robot->kicking = 0;
        foot->p = robot->initPos;
        if (robot->jointRemoved) { cpSpaceAddConstraint(&e->space, &robot->joint); robot->jointRemoved = 0; }
      }
    }
    if (robot->moveTime <= 0.0) {
      robot->moveTime = 0.0;
      robot->headMoving = 0.0;
      robot->leftBody.v = cpv_(0.0, 0.0); robot->leftBody.w = 0.0;
      robot->rightBody.v = cpv_(0.0, 0.0); robot->rightBody.w = 0.0;
    }
  }
  if (robot->fallen) {
    robot->fallTime -= time;
    if (robot->fallTime < 0.0) {
      dm_u32x4 u = rc_rng(e, (uint32_t)robot->id | (1u << 8));
      double r = dm_unit(u.v[0]);
      if (r > 0.9 && !robot->penalized && e->canFall) { rc_fall(e, robot, 0); return; }
      robot->fallen = 0;
      robot->fallCntr = 0;
    }
  }
  if (robot->penalized) {
    robot->penalTime -= time;
    if (robot->penalTime <= 0.0) {
      cpv p; double angle;
      robot->penalTime = 0.0; robot->penalized = 0; robot->fallCntr = 0; robot->fallen = 0;
      free_penalty_spot(e, robot, &p, &angle);
      robot->leftBody.p = p; cpBodySetAngle(&robot->leftBody, angle);
      robot->rightBody.p = p; cpBodySetAngle(&robot->rightBody, angle);
    }
  } else {
    int teamIdx = robot->team > 0 ? 0 : 1;
    cpv p = rc_robot_pos(robot);
    double robX = teamIdx ? RC_W - p.x : p.x;
    double penX = RC_SIDE + PENALTY_LENGTH + LINE_WIDTH / 2.0;
    int idx = defenders_find(e, teamIdx, robot->id);
    if (robX < penX && p.y > (RC_H / 2.0 - PENALTY_WIDTH) && p.y < (RC_H / 2.0 + PENALTY_WIDTH)) {
      if (idx < 0) {
        if (e->nDefenders[teamIdx] >= 2) rc_penalize(e, robot);
        else e->defenders[teamIdx][e->nDefenders[teamIdx]++] = robot->id;
      }
    } else if (idx >= 0) {
      int i;
      for (i = idx; i + 1 < e->nDefenders[teamIdx]; ++i) e->defenders[teamIdx][i] = e->defenders[teamIdx][i + 1];
      e->nDefenders[teamIdx]--;
    }
  }
  pos = rc_robot_pos(robot);
  if (pos.y < 0.0 || pos.x < 0.0 || pos.y > RC_H || pos.x > RC_W) rc_penalize(e, robot);
  /* quirk: `pos` was read BEFORE penalize teleported the robot (reference keeps the stale value, :988-1006) */
  if (pos.x != robot->prevPos.x || pos.y != robot->prevPos.y) {
    if ((robot->id == e->closestID[0] || robot->id == e->closestID[1]) && !robot->penalized) {
      cpv ballPos = e->ballBody.p;
      double diff = vlen(cpvsub(pos, ballPos)) - vlen(cpvsub(robot->prevPos, ballPos));
      e->robotRewards[robot->id] -= diff * 0.05;
      e->robotPosRewards[robot->id] += dm_max(0.0, -diff * 0.05);
    }
    robot->prevPos = pos;
  }
}

/* ------------------------------------------------------------------ ballFreeKickProcess :600-619 */
static void ball_free_kick_process(RoboCupEnv* e, int team) {
  if (team == 0) {
    const double time = TIME_STEP_MS;
    if (e->gracePeriod > 0.0) {
      e->gracePeriod -= time;
      if (e->gracePeriod < 0.0) { e->gracePeriod = 0.0; e->ballFreeCntr = 9999.0; }
    } else if (e->ballFreeCntr > 0.0) {
      e->ballFreeCntr -= time;
      if (e->ballFreeCntr < 0.0) { e->ballFreeCntr = 0.0; e->ballOwned = 0; }
    }
  } else {
    e->ballOwned = team;
    e->gracePeriod = 14999.0;
    e->ballFreeCntr = 0.0;
  }
}

/* ------------------------------------------------------------------ isBallOutOfField :622-732 */
int rc_is_ball_out_of_field(RoboCupEnv* e) {
  int finished = 0, team = 0, i, n = e->nPlayers;
  cpv pos = e->ballBody.p;
  double currReward[2] = {0.0, 0.0};
  const double outMin = RC_SIDE - BALL_RADIUS, outMaxX = RC_W - RC_SIDE + BALL_RADIUS, outMaxY = RC_H - RC_SIDE + BALL_RADIUS;
  if (pos.y < outMin || pos.x < outMin || pos.y > outMaxY || pos.x > outMaxX) {
    double x = RC_W / 2.0, y = RC_H / 2.0;
    team = e->nLastKicked ? e->robots[e->lastKicked[0]].team : 1;
    if (pos.y < outMin || pos.y > outMaxY) {
      x = team < 0 ? pos.x + 50.0 : pos.x - 50.0;
      y = pos.y < outMin ? outMin + BALL_RADIUS : outMaxY - BALL_RADIUS;
    } else {
      if (pos.y < RC_H / 2.0 + GOAL_WIDTH && pos.y > RC_H / 2.0 - GOAL_WIDTH) {
        finished = 1;
        if (pos.x < outMin) { currReward[0] += -25.0; currReward[1] += 25.0; e->goals[1] += 1; }
        else { currReward[0] += 25.0; currReward[1] += -25.0; e->goals[0] += 1; }
      } else {
        if (pos.x < outMin) {
          if (team < 0) x = RC_SIDE + PENALTY_LENGTH;
          else { x = RC_SIDE; y = pos.y < RC_H / 2.0 ? RC_SIDE : RC_H - RC_SIDE; }
        } else {
          if (team > 0) x = RC_W - (RC_SIDE + PENALTY_LENGTH);
          else { x = RC_W - RC_SIDE; y = pos.y < RC_H / 2.0 ? RC_SIDE : RC_H - RC_SIDE; }
        }
      }
    }
    e->ballBody.p = cpv_(x, y);
    e->ballBody.v = cpv_(0.0, 0.0);
    e->ballBody.w = 0.0;
  }
  ball_free_kick_process(e, -team);
  if (!finished) {
    currReward[0] += (e->ballBody.p.x - e->ballPrevPos.x) / 20.0;
    currReward[1] -= (e->ballBody.p.x - e->ballPrevPos.x) / 20.0;
  }
  e->ballPrevPos = e->ballBody.p;
  {
    double disc = 1.0; /* kickDiscount ** i */
    for (i = 0; i < e->nLastKicked; ++i) {
      int id = e->lastKicked[i];
      double rew = (id < n ? currReward[0] : currReward[1]) * disc;
      e->robotRewards[id] += rew;
      e->robotPosRewards[id] += dm_max(0.0, rew);
      disc *= KICK_DISCOUNT;
    }
  }
  for (i = 0; i < e->nRobots; ++i) {
    Robot* robot = &e->robots[i];
    int cond1 = (robot->id == e->closestID[0] || robot->id == e->closestID[1]);
    int cond2 = vlen(cpvsub(rc_robot_pos(robot), pos)) < 150.0;
    if (cond1 || cond2) {
      if (!in_last_kicked(e, robot->id))
        e->robotRewards[robot->id] += dm_min(0.0, (robot->id < n ? currReward[0] : currReward[1]) * KICK_DISCOUNT);
    }
  }
  e->teamRewards[0] += currReward[0] * 0.1;
  e->teamRewards[1] += currReward[1] * 0.1;
  {
    cpv bp = e->ballBody.p;
    int best0 = 0, best1 = 0;
    double d0 = INFINITY, d1 = INFINITY;
    for (i = 0; i < n; ++i) {
      cpv d = cpvsub(bp, rc_robot_pos(&e->robots[i]));
      double q = d.x * d.x + d.y * d.y; /* get_length_sqrd: x**2 + y**2 */
      if (q < d0) { d0 = q; best0 = i; }
    }
    for (i = 0; i < n; ++i) {
      cpv d = cpvsub(bp, rc_robot_pos(&e->robots[n + i]));
      double q = d.x * d.x + d.y * d.y;
      if (q < d1) { d1 = q; best1 = i; }
    }
    e->closestID[0] = best0;
    e->closestID[1] = n + best1;
  }
  return finished;
}

/* ------------------------------------------------------------------ collision callbacks :1010-1146 */
static Robot* robot_of_shape(cpShape* s) { return (Robot*)s->user; }

static int robot_pushing_det(cpArbiter* arb, cpSpace* sp, void* data) { /* begin, Robot-Robot */
  cpShape *s1, *s2; Robot *r1, *r2; cpv v1, v2, dp; double adp;
  (void)sp; (void)data;
  cpArbiterGetShapes(arb, &s1, &s2);
  r1 = robot_of_shape(s1); r2 = robot_of_shape(s2);
  v1 = s1->body->v; v2 = s2->body->v;
  dp = cpvsub(rc_robot_pos(r1), rc_robot_pos(r2));
  adp = dm_atan2(dp.y, dp.x);
  r1->mightPush = vlen(v1) > 1.0 && dm_cos(adp - dm_atan2(v1.y, v1.x)) < -0.4;
  r2->mightPush = vlen(v2) > 1.0 && dm_cos(adp - dm_atan2(v2.y, v2.x)) > 0.4;
  r1->touching = 1; r2->touching = 1;
  r1->touchCntr = 0; r2->touchCntr = 0;
  return 1;
}

static double ipow(double b, int n) { return dm_powi(b, n); } /* thresh ** touchCntr, deterministic (dynenv_math.h) */

static void robot_collision(cpArbiter* arb, cpSpace* sp, void* data) { /* post_solve, Robot-Robot */
  RoboCupEnv* e = (RoboCupEnv*)data;
  cpShape *s1, *s2; Robot *r1, *r2;
  dm_u32x4 u;
  double r;
  (void)sp;
  if (!e->canFall) return;
  cpArbiterGetShapes(arb, &s1, &s2);
  r1 = robot_of_shape(s1); r2 = robot_of_shape(s2);
  if (r1 == r2) return;
  if (!(r1->fallen || r1->penalized)) r1->touchCntr += 1;
  if (!(r2->fallen || r2->penalized)) r2->touchCntr += 1;
  u = rc_rng(e, pair_entity(arb, 2));
  r = dm_unit(u.v[0]);
  if (r > ipow(r1->mightPush ? 0.99995 : 0.9999, r1->touchCntr) && !r1->fallen) { rc_fall(e, r1, r1->mightPush); r1->touchCntr = 0; }
  r = dm_unit(u.v[1]);
  if (r > ipow(r2->mightPush ? 0.99995 : 0.9999, r2->touchCntr) && !r2->fallen) { rc_fall(e, r2, r2->mightPush); r2->touchCntr = 0; }
  if (r1->mightPush && !r2->mightPush && r2->fallen && r1->team != r2->team) { rc_penalize(e, r1); r1->touchCntr = 0; }
  else if (r2->mightPush && !r1->mightPush && r1->fallen && r1->team != r2->team) { rc_penalize(e, r2); r2->touchCntr = 0; }
}

static void separate_cb(cpArbiter* arb, cpSpace* sp, void* data) { /* separate :1091-1103 */
  cpShape* sh[2]; int k;
  (void)sp; (void)data;
  sh[0] = arb->a; sh[1] = arb->b;
  for (k = 0; k < 2; ++k) {
    if (sh[k]->collision_type == CT_Robot) {
      Robot* r = robot_of_shape(sh[k]);
      r->touching = 0; r->mightPush = 0; r->touchCntr = 0;
    }
  }
}

static void goalpost_collision(cpArbiter* arb, cpSpace* sp, void* data) { /* post_solve, Robot-Goalpost :1106-1125 */
  RoboCupEnv* e = (RoboCupEnv*)data;
  cpShape *s1, *s2; Robot* robot; dm_u32x4 u;
  (void)sp;
  if (!e->canFall) return;
  cpArbiterGetShapes(arb, &s1, &s2);
  robot = robot_of_shape(s1);
  if (robot->fallen) { robot->touchCntr = 0; return; }
  if (!robot->touching) { robot->touching = 1; robot->touchCntr = 0; }
  robot->touchCntr += 1;
  u = rc_rng(e, pair_entity(arb, 3));
  if (dm_unit(u.v[0]) > ipow(0.9998, robot->touchCntr)) rc_fall(e, robot, 1);
}

static int ball_collision(cpArbiter* arb, cpSpace* sp, void* data) { /* begin, Robot-Ball :1128-1146 */
  RoboCupEnv* e = (RoboCupEnv*)data;
  cpShape *s1, *s2; Robot* robot;
  (void)sp;
  cpArbiterGetShapes(arb, &s1, &s2);
  robot = robot_of_shape(s1);
  if (e->ballOwned != 0) {
    if (robot->team != e->ballOwned && !robot->penalized && e->canFall) rc_penalize(e, robot);
    else { e->ballOwned = 0; e->gracePeriod = 0.0; e->ballFreeCntr = 0.0; }
  }
  push_last_kicked(e, robot->id);
  return 1;
}

/* ------------------------------------------------------------------ scene */
int rc_obs_dim(int nPlayers) { return 4 + 8 + (2 * nPlayers - 1) * 6; }

static void setup_robot(RoboCupEnv* e, int id, cpv pos, int team) { /* Robot.__init__ :22-88 */
  Robot* r = &e->robots[id];
  double angle = team > 0 ? 0.0 : DM_PI;
  cpv a = cpv_(-ROBOT_LENGTH, ROBOT_LENGTH), b = cpv_(ROBOT_LENGTH, ROBOT_LENGTH);
  cpv c = cpv_(-ROBOT_LENGTH, -ROBOT_LENGTH), d = cpv_(ROBOT_LENGTH, -ROBOT_LENGTH);
  memset(r, 0, sizeof(*r));
  cpBodyInit(&r->leftBody, ROBOT_MASS, cpMomentForSegment(ROBOT_MASS, a, b, ROBOT_RADIUS), CP_BODY_DYNAMIC);
  r->leftBody.p = pos; cpBodySetAngle(&r->leftBody, angle); r->leftBody.velocity_func = friction_robot;
  cpSegmentInit(&r->leftFoot, &r->leftBody, a, b, ROBOT_RADIUS, 2 * id);
  r->leftFoot.e = 0.3; r->leftFoot.u = 2.5; r->leftFoot.collision_type = CT_Robot; r->leftFoot.user = r;
  cpBodyInit(&r->rightBody, ROBOT_MASS, cpMomentForSegment(ROBOT_MASS, c, d, ROBOT_RADIUS), CP_BODY_DYNAMIC);
  r->rightBody.p = pos; cpBodySetAngle(&r->rightBody, angle); r->rightBody.velocity_func = friction_robot;
  cpSegmentInit(&r->rightFoot, &r->rightBody, c, d, ROBOT_RADIUS, 2 * id + 1);
  r->rightFoot.e = 0.3; r->rightFoot.u = 2.5; r->rightFoot.collision_type = CT_Robot; r->rightFoot.user = r;
  cpPivotJointInit(&r->joint, &r->leftBody, &r->rightBody, pos);
  r->joint.errorBias = 0.1;
  cpRotaryLimitJointInit(&r->rotJoint, &r->leftBody, &r->rightBody, 0.0, 0.0);
  r->team = team; r->id = id;
  r->prevPos = rc_robot_pos(r);
}

static void build_space(RoboCupEnv* e) {
  cpHandler* h;
  int i;
  cpSpaceInit(&e->space);
  /* _handle_collisions :229-237 */
  h = cpSpaceAddHandler(&e->space, CT_Robot, CT_Goalpost); h->post_solve = goalpost_collision; h->separate = separate_cb; h->data = e;
  h = cpSpaceAddHandler(&e->space, CT_Robot, CT_Robot); h->begin = robot_pushing_det; h->post_solve = robot_collision; h->separate = separate_cb; h->data = e;
  h = cpSpaceAddHandler(&e->space, CT_Robot, CT_Ball); h->begin = ball_collision; h->data = e;
  for (i = 0; i < e->nRobots; ++i) {
    Robot* r = &e->robots[i];
    cpSpaceAddBody(&e->space, &r->leftBody); cpSpaceAddShape(&e->space, &r->leftFoot);
    cpSpaceAddBody(&e->space, &r->rightBody); cpSpaceAddShape(&e->space, &r->rightFoot);
    if (!r->jointRemoved) cpSpaceAddConstraint(&e->space, &r->joint);
    cpSpaceAddConstraint(&e->space, &r->rotJoint);
  }
  cpSpaceAddBody(&e->space, &e->ballBody); cpSpaceAddShape(&e->space, &e->ballShape);
  for (i = 0; i < 4; ++i) cpSpaceAddShape(&e->space, &e->postShape[i]);
}

static void setup_ball_and_posts(RoboCupEnv* e, cpv ballPos) {
  static const double PX[4] = {RC_SIDE, RC_SIDE, RC_W - RC_SIDE, RC_W - RC_SIDE};
  static const double PY[4] = {RC_H / 2.0 + GOAL_WIDTH, RC_H / 2.0 - GOAL_WIDTH, RC_H / 2.0 + GOAL_WIDTH, RC_H / 2.0 - GOAL_WIDTH};
  int i;
  cpBodyInit(&e->ballBody, 10.0, cpMomentForCircle(10.0, 0.0, BALL_RADIUS * 2.0), CP_BODY_DYNAMIC); /* Ball.py:5-20 */
  e->ballBody.p = ballPos; e->ballBody.velocity_func = friction_ball;
  cpCircleInit(&e->ballShape, &e->ballBody, BALL_RADIUS * 2.0, RC_SLOT_BALL);
  e->ballShape.e = 0.98; e->ballShape.u = 3.0; e->ballShape.collision_type = CT_Ball;
  for (i = 0; i < 4; ++i) { /* Goalpost.py:4-14, RoboCupEnvironment.py:295-302 */
    cpBodyInit(&e->postBody[i], 0.0, 0.0, CP_BODY_STATIC);
    e->postBody[i].p = cpv_(PX[i], PY[i]);
    cpCircleInit(&e->postShape[i], &e->postBody[i], 5.0 * 2.0, RC_SLOT_POST + i);
    e->postShape[i].e = 0.95; e->postShape[i].collision_type = CT_Goalpost;
  }
}

void rc_init(RoboCupEnv* e, int nPlayers, uint64_t seed, uint32_t genv, int flags) {
  memset(e, 0, sizeof(*e));
  e->nPlayers = nPlayers > RC_MAX_PLAYERS ? RC_MAX_PLAYERS : nPlayers;
  e->nRobots = 2 * e->nPlayers;
  e->seed = seed; e->genv = genv; e->episode = 0;
  e->canFall = (flags & DYNENV_FLAG_CAN_FALL) != 0;
  e->allowHeadTurn = (flags & DYNENV_FLAG_ALLOW_HEAD_TURN) != 0;
  e->useObsRewards = (flags & DYNENV_FLAG_USE_OBS_REWARDS) != 0;
  e->randomInit = (flags & DYNENV_FLAG_RANDOM_INIT) != 0;
  e->deterministicTurn = (flags & DYNENV_FLAG_DETERMINISTIC_TURN) != 0;
}

/* _create_robot_spots :275-293 (randomInit = False), the 18 random.random() draws consumed in source order */
void rc_spots(const double* rnd, cpv spots[2][5]) {
  const double centX = RC_W / 2.0;
  spots[0][0] = cpv_(centX - (BALL_RADIUS * 2.0 + ROBOT_TOTAL_RADIUS) - rnd[0] * 50.0, RC_H / 2.0 + (rnd[1] - 0.5) * 25.0);
  spots[0][1] = cpv_(centX - (ROBOT_TOTAL_RADIUS + LINE_WIDTH * 2.0) - rnd[2] * 50.0, RC_SIDE + FIELD_H / 4.0 + (rnd[3] - 0.5) * 50.0);
  spots[0][2] = cpv_(centX - (ROBOT_TOTAL_RADIUS + LINE_WIDTH * 2.0) - rnd[4] * 50.0, RC_SIDE + 3.0 * FIELD_H / 4.0 + (rnd[5] - 0.5) * 50.0);
  spots[0][3] = cpv_(centX - (FIELD_W / 4.0) - (rnd[6] - 0.5) * 50.0, RC_SIDE + FIELD_H / 2.0 + (rnd[7] - 0.5) * 50.0);
  spots[0][4] = cpv_(RC_SIDE + 20.0, RC_H / 2.0 + (rnd[8] - 0.5) * 50.0);
  spots[1][0] = cpv_(centX + (CENTER_CIRCLE_RADIUS * 2.0 + ROBOT_TOTAL_RADIUS + LINE_WIDTH / 2.0) + rnd[9] * 50.0, RC_H / 2.0 + (rnd[10] - 0.5) * 50.0);
  spots[1][1] = cpv_(centX + (ROBOT_TOTAL_RADIUS + LINE_WIDTH / 2.0 + CENTER_CIRCLE_RADIUS) + rnd[11] * 50.0, RC_SIDE + FIELD_H / 4.0 + (rnd[12] - 0.5) * 50.0);
  spots[1][2] = cpv_(centX + (ROBOT_TOTAL_RADIUS + LINE_WIDTH / 2.0 + CENTER_CIRCLE_RADIUS) + rnd[13] * 50.0, RC_SIDE + 3.0 * FIELD_H / 4.0 + (rnd[14] - 0.5) * 50.0);
  spots[1][3] = cpv_(centX + (RC_SIDE + FIELD_W / 4.0) + rnd[15] * 50.0, RC_SIDE + FIELD_H / 2.0 + (rnd[16] - 0.5) * 50.0);
  spots[1][4] = cpv_(RC_W - (RC_SIDE + 20.0), RC_H / 2.0 + (rnd[17] - 0.5) * 50.0);
}

/* _create_robot_spots :241-272 (randomInit = True): one random spot in each of 10 field cells (20 random.random() draws in
 * source order), the goal-side cells go to the two teams, the 8 middle ones are dealt by np.random.permutation(8) */
void rc_spots_random(const double* rnd, const int* perm8, cpv spots[2][5]) {
  static const double xL[7] = {RC_SIDE + 10.0, RC_SIDE + 50.0, RC_SIDE + 250.0, RC_SIDE + 450.0, RC_SIDE + 650.0, RC_SIDE + 850.0, RC_SIDE + 890.0};
  static const double yL[3] = {RC_SIDE + 20.0, RC_SIDE + 300.0, RC_SIDE + 580.0};
  cpv rs[10];
  int i, j, k = 0, d = 0, cnt[2] = {1, 1};
  for (i = 0; i < 6; ++i) {
    const int edge = (i == 0 || i == 5);
    for (j = 0; j < (edge ? 1 : 2); ++j) {
      const double yBeg = edge ? yL[0] : yL[j], yEnd = edge ? yL[2] : yL[j + 1];
      const double x = xL[i] + rnd[d] * (xL[i + 1] - xL[i]);
      const double y = yBeg + rnd[d + 1] * (yEnd - yBeg);
      d += 2;
      rs[k++] = cpv_(x, y);
    }
  }
  spots[0][0] = rs[0]; spots[1][0] = rs[9];
  for (i = 0; i < 8; ++i) { const int t = i < 4 ? 0 : 1; spots[t][cnt[t]++] = rs[perm8[i] + 1]; }
}

void rc_reset(RoboCupEnv* e) { /* __init__ :23-64 + _setup_scene :73-99 */
  cpv spots[2][5], ballPos = cpv_(520.0, 370.0); /* W // 2, H // 2 */
  int perm[2][5], t, i, owned = 1;
  uint32_t ep = e->episode;
  double rnd[24];
  for (i = 0; i < 24; ++i) {
    dm_u32x4 u = dm_env_rng(e->seed, e->genv, ep, DM_RNG_ROBO_RESET, (uint32_t)i, 0);
    rnd[i] = dm_unit(u.v[0]);
  }
  if (e->randomInit) {
    int perm8[8];
    for (i = 0; i < 8; ++i) perm8[i] = i;
    for (i = 0; i < 7; ++i) { /* np.random.permutation(8) -> Fisher-Yates */
      dm_u32x4 u = dm_env_rng(e->seed, e->genv, ep, DM_RNG_ROBO_RESET, (uint32_t)(48 + i), 0);
      int j = i + dm_randint(u.v[0], 0, 7 - i);
      int tmp = perm8[i]; perm8[i] = perm8[j]; perm8[j] = tmp;
    }
    rc_spots_random(rnd, perm8, spots);
    /* _create_ball :325-331: position anywhere on the field, random ownership (the 4th draw only if owned) */
    ballPos = cpv_(rnd[20] * FIELD_W + RC_SIDE, rnd[21] * FIELD_H + RC_SIDE);
    owned = rnd[22] > 0.4 ? 1 : 0;
    if (owned != 0 && rnd[23] > 0.5) owned *= -1;
  } else {
    rc_spots(rnd, spots);
  }
  /* np.random.permutation(5) x2 -> Fisher-Yates */
  for (t = 0; t < 2; ++t) {
    for (i = 0; i < 5; ++i) perm[t][i] = i;
    for (i = 0; i < 4; ++i) {
      dm_u32x4 u = dm_env_rng(e->seed, e->genv, ep, DM_RNG_ROBO_RESET, (uint32_t)(32 + t * 8 + i), 0);
      int j = i + dm_randint(u.v[0], 0, 4 - i);
      int tmp = perm[t][i]; perm[t][i] = perm[t][j]; perm[t][j] = tmp;
    }
  }
  for (i = 0; i < e->nPlayers; ++i) setup_robot(e, i, spots[0][perm[0][i]], 1);
  for (i = 0; i < e->nPlayers; ++i) setup_robot(e, e->nPlayers + i, spots[1][perm[1][i]], -1);
  if (e->deterministicTurn) /* :317-319 */
    for (i = 0; i < e->nRobots; ++i) e->robots[i].headAngle = (double)e->robots[i].team * ROBOT_HEAD_MAX;
  setup_ball_and_posts(e, ballPos);
  e->ballPrevPos = e->ballBody.p;
  e->nLastKicked = 0;
  e->elapsed = 0;
  e->ballOwned = owned; e->ballFreeCntr = 9999.0; e->gracePeriod = 0.0;
  e->goals[0] = e->goals[1] = 0; e->closestID[0] = e->closestID[1] = 0;
  e->nDefenders[0] = e->nDefenders[1] = 0;
  e->penalTimes[0] = e->penalTimes[1] = 20000.0;
  e->teamRewards[0] = e->teamRewards[1] = 0.0;
  for (i = 0; i < RC_MAX_ROBOTS; ++i) { e->robotRewards[i] = e->robotPosRewards[i] = e->episodeRewards[i] = e->episodePosRewards[i] = e->episodeObsRewards[i] = 0.0; }
  build_space(e);
  e->episode++;
}

/* ------------------------------------------------------------------ observations :1149-1189, :440-443 */
static inline double norm_after_scale(double pt, double nf, double mean, double team) { return (pt - mean) * nf * team; } /* cutils.py:326-331 */
#define RC_STD_NORM (2.0 / RC_W)
#define RC_MEAN_X (RC_W / 2.0)
#define RC_MEAN_Y (RC_H / 2.0)

void rc_write_full_obs(const RoboCupEnv* e, float* out) {
  int R = e->nRobots, dim = rc_obs_dim(e->nPlayers), a, k;
  for (a = 0; a < R; ++a) {
    const Robot* ag = &e->robots[a];
    float* o = out + (size_t)a * dim;
    double team = (double)ag->team, s, c;
    cpv p = rc_robot_pos(ag);
    double ang = robot_angle(ag);
    float* q;
    o[0] = (float)norm_after_scale(e->ballBody.p.x, RC_STD_NORM, RC_MEAN_X, team);
    o[1] = (float)norm_after_scale(e->ballBody.p.y, RC_STD_NORM, RC_MEAN_Y, team);
    o[2] = (float)(e->ballOwned * ag->team);
    o[3] = (float)(ag->id == e->closestID[0] || ag->id == e->closestID[1]);
    o[4] = (float)norm_after_scale(p.x, RC_STD_NORM, RC_MEAN_X, team);
    o[5] = (float)norm_after_scale(p.y, RC_STD_NORM, RC_MEAN_Y, team);
    dm_sincos(ang + ag->headAngle, &s, &c);
    o[6] = (float)c; o[7] = (float)s;
    dm_sincos(ag->headAngle, &s, &c);
    o[8] = (float)c; o[9] = (float)s;
    o[10] = (float)ag->team;
    o[11] = (float)(ag->fallen || ag->penalized);
    q = o + 12;
    for (k = 0; k < R; ++k) {
      const Robot* rb = &e->robots[k];
      cpv rp;
      if (k == a) continue;
      rp = rc_robot_pos(rb);
      q[0] = (float)norm_after_scale(rp.x, RC_STD_NORM, RC_MEAN_X, team);
      q[1] = (float)norm_after_scale(rp.y, RC_STD_NORM, RC_MEAN_Y, team);
      dm_sincos(robot_angle(rb), &s, &c);
      q[2] = (float)c; q[3] = (float)s;
      q[4] = (float)(rb->team * ag->team);
      q[5] = (float)(rb->fallen || rb->penalized);
      q += 6;
    }
  }
}

/* getFullState(agent=None) :1149-1161 - what step() puts into info['Full State'] (:511): robots [R][6] = (normalize(x,
 * standardNorm, 0), normalize(y, ...), cos a, sin a, team, fallen | penalized) in field coordinates (no team flip), then the
 * ball [3] = (normalize(bx, ...), normalize(by, ...), ballOwned); normalize = cutils.py:318-323 */
static inline double norm_plain(double pt, double nf) { return ((pt * nf) - 0.0) * 2.0 * 1.0; }
void rc_write_global_state(const RoboCupEnv* e, float* out) {
  int R = e->nRobots, a;
  for (a = 0; a < R; ++a) {
    const Robot* rb = &e->robots[a];
    cpv p = rc_robot_pos(rb);
    double s, c;
    float* o = out + (size_t)a * 6;
    o[0] = (float)norm_plain(p.x, RC_STD_NORM);
    o[1] = (float)norm_plain(p.y, RC_STD_NORM);
    dm_sincos(robot_angle(rb), &s, &c);
    o[2] = (float)c; o[3] = (float)s;
    o[4] = (float)rb->team;
    o[5] = (float)(rb->fallen || rb->penalized);
  }
  out[R * 6 + 0] = (float)norm_plain(e->ballBody.p.x, RC_STD_NORM);
  out[R * 6 + 1] = (float)norm_plain(e->ballBody.p.y, RC_STD_NORM);
  out[R * 6 + 2] = (float)e->ballOwned;
}

/* ------------------------------------------------------------------ step :446-524 */
/* processSeens (RoboCupEnvironment.py, `def processSeens`) for one robot: sums over the step's 5 snapshots of numLandMarks,
 * of each other robot's seen flag, and of ballsSeen -> the observation reward */
double rc_process_seens(double lSum, const double* rSum, int nOthers, double bSum) {
  const double factor = 0.0025, bfactor = 0.01;
  double lSeens = lSum / 5.0, rSeens = 0.0, bSeens = bSum;
  int k;
  lSeens = lSeens < 0.0 ? 0.0 : (lSeens > 3.0 ? 3.0 : lSeens);
  for (k = 0; k < nOthers; ++k) rSeens += rSum[k] < 0.0 ? 0.0 : (rSum[k] > 2.0 ? 2.0 : rSum[k]);
  bSeens = bSeens < 0.0 ? 0.0 : (bSeens > 3.0 ? 3.0 : bSeens);
  return factor * (rSeens + lSeens) + bfactor * bSeens;
}

int rc_step(RoboCupEnv* e, const int32_t* actions, float* obs, double* rewards) {
  int R = e->nRobots, n = e->nPlayers, i, a, k, t = 0;
  const int partial = e->obsType == DYNENV_OBS_PARTIAL;
  const int dim = partial ? RCP_DIM : rc_obs_dim(n);
  double obsRewards[RC_MAX_ROBOTS];
  /* processSeens inputs (:1563-1575): per robot the 5 snapshots' numLandMarks, robotsSeen arrays, ballsSeen */
  double lSum[RC_MAX_ROBOTS], rSum[RC_MAX_ROBOTS][RC_MAX_ROBOTS], bSum[RC_MAX_ROBOTS];
  float row[RCP_DIM];
  e->teamRewards[0] = e->teamRewards[1] = 0.0;
  for (a = 0; a < RC_MAX_ROBOTS; ++a) {
    e->robotRewards[a] = e->robotPosRewards[a] = 0.0;
    obsRewards[a] = 0.0; lSum[a] = bSum[a] = 0.0;
    for (k = 0; k < RC_MAX_ROBOTS; ++k) rSum[a][k] = 0.0;
  }
  for (i = 0; i < RC_STEP_ITER; ++i) {
    for (a = 0; a < R; ++a) {
      if (i == 0) rc_process_action(e, &e->robots[a], actions + 4 * a);
      rc_tick(e, &e->robots[a]);
    }
    rc_is_ball_out_of_field(e);
    cpSpaceStep(&e->space, 1.0 / 100.0);
    e->elapsed += 1;
    if (i % 10 == 9) {
      if (partial) {
        for (a = 0; a < R; ++a) {
          float* o = obs ? obs + ((size_t)t * R + a) * dim : row;
          if (rc_agent_vision(e, a, e->noiseType, e->noiseMagnitude, (uint32_t)e->elapsed, o)) e->obsOverflow = 1;
          lSum[a] += (double)o[RCP_OFF_TAIL + 6];
          bSum[a] += (double)o[RCP_OFF_TAIL + 7];
          for (k = 0; k < R - 1; ++k) rSum[a][k] += (double)o[RCP_OFF_TAIL + 8 + k];
        }
      } else if (obs) rc_write_full_obs(e, obs + (size_t)t * R * dim);
      ++t;
    }
  }
  if (partial && e->useObsRewards)
    for (a = 0; a < R; ++a) obsRewards[a] += rc_process_seens(lSum[a], rSum[a], R - 1, bSum[a]);
  for (a = 0; a < R; ++a) {
    double tr = a < n ? e->teamRewards[0] : e->teamRewards[1];
    e->robotRewards[a] += tr;
    e->robotRewards[a] += obsRewards[a]; /* zero for Full observations (processSeens returns early, :1565-1566) */
    e->episodeRewards[a] += e->robotRewards[a];
    e->robotPosRewards[a] += dm_max(0.0, tr);
    e->robotPosRewards[a] += dm_max(obsRewards[a], 0.0); /* np.clip(obsRewards, a_min=0) :503 */
    e->episodePosRewards[a] += e->robotPosRewards[a];
    e->episodeObsRewards[a] += obsRewards[a];
    rewards[a] = e->robotRewards[a];
  }
  return e->elapsed >= RC_MAX_TIME;
}

/* ------------------------------------------------------------------ state blob */
void rc_get_state(const RoboCupEnv* e, dynenv_robocup_state_t* st) {
  int i, t;
  memset(st, 0, sizeof(*st));
  st->elapsed = e->elapsed; st->n_robots = e->nRobots; st->ball_owned = e->ballOwned; st->n_last_kicked = e->nLastKicked;
  for (i = 0; i < 4; ++i) st->last_kicked[i] = i < e->nLastKicked ? e->lastKicked[i] : 0;
  for (t = 0; t < 2; ++t) {
    st->goals[t] = e->goals[t]; st->closest[t] = e->closestID[t]; st->n_def[t] = e->nDefenders[t];
    for (i = 0; i < e->nDefenders[t]; ++i) st->defenders[t][i] = e->defenders[t][i];
    st->penal_times[t] = e->penalTimes[t];
  }
  st->episode = (int32_t)e->episode;
  st->ball_free_cntr = e->ballFreeCntr; st->grace_period = e->gracePeriod;
  st->bpx = e->ballBody.p.x; st->bpy = e->ballBody.p.y; st->bvx = e->ballBody.v.x; st->bvy = e->ballBody.v.y; st->bw = e->ballBody.w;
  st->bprevx = e->ballPrevPos.x; st->bprevy = e->ballPrevPos.y;
  for (i = 0; i < RC_MAX_ROBOTS; ++i) { st->episode_r[i] = e->episodeRewards[i]; st->episode_pos_r[i] = e->episodePosRewards[i]; }
  for (i = 0; i < e->nRobots; ++i) {
    const Robot* r = &e->robots[i]; dynenv_robot_state_t* s = &st->robots[i];
    s->lpx = r->leftBody.p.x; s->lpy = r->leftBody.p.y; s->lvx = r->leftBody.v.x; s->lvy = r->leftBody.v.y; s->la = r->leftBody.a; s->lw = r->leftBody.w;
    s->rpx = r->rightBody.p.x; s->rpy = r->rightBody.p.y; s->rvx = r->rightBody.v.x; s->rvy = r->rightBody.v.y; s->ra = r->rightBody.a; s->rw = r->rightBody.w;
    s->head_angle = r->headAngle; s->head_moving = r->headMoving; s->prevx = r->prevPos.x; s->prevy = r->prevPos.y;
    s->initx = r->initPos.x; s->inity = r->initPos.y; s->penal_time = r->penalTime; s->fall_time = r->fallTime; s->move_time = r->moveTime;
    s->team = r->team; s->penalized = r->penalized; s->touching = r->touching; s->touch_cntr = r->touchCntr; s->might_push = r->mightPush;
    s->fallen = r->fallen; s->fall_cntr = r->fallCntr; s->kicking = r->kicking; s->foot = r->foot; s->joint_removed = r->jointRemoved;
  }
}

void rc_set_state(RoboCupEnv* e, const dynenv_robocup_state_t* st) {
  int i, t;
  e->elapsed = st->elapsed; e->ballOwned = st->ball_owned; e->nLastKicked = st->n_last_kicked;
  for (i = 0; i < 4; ++i) e->lastKicked[i] = st->last_kicked[i];
  for (t = 0; t < 2; ++t) {
    e->goals[t] = st->goals[t]; e->closestID[t] = st->closest[t]; e->nDefenders[t] = st->n_def[t];
    for (i = 0; i < st->n_def[t]; ++i) e->defenders[t][i] = st->defenders[t][i];
    e->penalTimes[t] = st->penal_times[t];
  }
  e->episode = (uint32_t)st->episode;
  e->ballFreeCntr = st->ball_free_cntr; e->gracePeriod = st->grace_period;
  for (i = 0; i < RC_MAX_ROBOTS; ++i) { e->episodeRewards[i] = st->episode_r[i]; e->episodePosRewards[i] = st->episode_pos_r[i]; }
  for (i = 0; i < e->nRobots; ++i) {
    const dynenv_robot_state_t* s = &st->robots[i];
    Robot* r;
    setup_robot(e, i, cpv_(s->lpx, s->lpy), s->team);
    r = &e->robots[i];
    r->leftBody.v = cpv_(s->lvx, s->lvy); cpBodySetAngle(&r->leftBody, s->la); r->leftBody.w = s->lw;
    r->rightBody.p = cpv_(s->rpx, s->rpy); r->rightBody.v = cpv_(s->rvx, s->rvy); cpBodySetAngle(&r->rightBody, s->ra); r->rightBody.w = s->rw;
    r->headAngle = s->head_angle; r->headMoving = s->head_moving; r->prevPos = cpv_(s->prevx, s->prevy); r->initPos = cpv_(s->initx, s->inity);
    r->penalTime = s->penal_time; r->fallTime = s->fall_time; r->moveTime = s->move_time;
    r->penalized = s->penalized; r->touching = s->touching; r->touchCntr = s->touch_cntr; r->mightPush = s->might_push;
    r->fallen = s->fallen; r->fallCntr = s->fall_cntr; r->kicking = s->kicking; r->foot = s->foot; r->jointRemoved = s->joint_removed;
  }
  setup_ball_and_posts(e, cpv_(st->bpx, st->bpy));
  e->ballBody.v = cpv_(st->bvx, st->bvy); e->ballBody.w = st->bw;
  e->ballPrevPos = cpv_(st->bprevx, st->bprevy);
  build_space(e);
}

/* ------------------------------------------------------------------ test hooks (golden tests) */
int rc_test_begin(RoboCupEnv* e, int slotA, int slotB) { return cpSpaceTestCallback(&e->space, slotA, slotB, 0); }
int rc_test_callback(RoboCupEnv* e, int slotA, int slotB, int which) { return cpSpaceTestCallback(&e->space, slotA, slotB, which); }
void rc_test_zero_rewards(RoboCupEnv* e) {
  int i;
  e->teamRewards[0] = e->teamRewards[1] = 0.0;
  for (i = 0; i < RC_MAX_ROBOTS; ++i) e->robotRewards[i] = e->robotPosRewards[i] = 0.0;
}
void rc_test_get_rewards(const RoboCupEnv* e, double* out22) {
  int i;
  for (i = 0; i < RC_MAX_ROBOTS; ++i) { out22[i] = e->robotRewards[i]; out22[10 + i] = e->robotPosRewards[i]; }
  out22[20] = e->teamRewards[0]; out22[21] = e->teamRewards[1];
}
