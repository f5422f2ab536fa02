/* oracle_api.c — ORACLE (test infrastructure): host-pointer C API mirroring include/dynenv.h so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the CPU restatement exactly like the HIP
 * library.  Nothing in dynenv_amd/ may load this. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "driving.h"
#include "robocup.h"
#include "robocup_partial.h"
#include "dynenv.h"
#include "dynenv_math.h"

typedef struct oracle {
  dynenv_cfg_t cfg;
  int n_agents, obs_dim, n_time_steps, action_dim;
  DrivingEnv* drv;
  RoboCupEnv* rc;
  int threads;
} oracle_t;

int oracle_create(const dynenv_cfg_t* cfg, oracle_t** out) {
  oracle_t* o;
  int i;
  if (!cfg || !out || cfg->num_envs <= 0) return DYNENV_ERR_ARG;
  o = (oracle_t*)calloc(1, sizeof(*o));
  o->cfg = *cfg;
  o->threads = 1;
  if (cfg->env_type == DYNENV_DRIVE) {
    if (cfg->obs_type != DYNENV_OBS_FULL && cfg->obs_type != DYNENV_OBS_PARTIAL) { free(o); return DYNENV_ERR_UNSUPPORTED; }
    o->n_agents = cfg->n_players > DYNENV_MAX_CARS ? DYNENV_MAX_CARS : cfg->n_players;
    o->obs_dim = cfg->obs_type == DYNENV_OBS_PARTIAL ? drv_partial_obs_dim() : drv_obs_dim(o->n_agents);
    o->n_time_steps = 1;
    o->action_dim = 2;
    o->drv = (DrivingEnv*)calloc((size_t)cfg->num_envs, sizeof(DrivingEnv));
    for (i = 0; i < cfg->num_envs; ++i) {
      drv_init(&o->drv[i], cfg->n_players, cfg->seed, (uint32_t)(cfg->env_id_offset + i));
      o->drv[i].obsType = cfg->obs_type; o->drv[i].noiseType = cfg->noise_type; o->drv[i].noiseMagnitude = cfg->noise_magnitude;
    }
  } else if (cfg->env_type == DYNENV_ROBO_CUP) {
    if (cfg->obs_type != DYNENV_OBS_FULL && cfg->obs_type != DYNENV_OBS_PARTIAL) { free(o); return DYNENV_ERR_UNSUPPORTED; }
    o->n_agents = 2 * (cfg->n_players > RC_MAX_PLAYERS ? RC_MAX_PLAYERS : cfg->n_players);
    o->obs_dim = cfg->obs_type == DYNENV_OBS_PARTIAL ? rc_partial_obs_dim() : rc_obs_dim(o->n_agents / 2);
    o->n_time_steps = 5;
    o->action_dim = 4;
    o->rc = (RoboCupEnv*)calloc((size_t)cfg->num_envs, sizeof(RoboCupEnv));
    for (i = 0; i < cfg->num_envs; ++i) {
      rc_init(&o->rc[i], cfg->n_players, cfg->seed, (uint32_t)(cfg->env_id_offset + i), cfg->flags);
      o->rc[i].obsType = cfg->obs_type; o->rc[i].noiseType = cfg->noise_type; o->rc[i].noiseMagnitude = cfg->noise_magnitude;
    }
  } else {
    free(o);
    return DYNENV_ERR_UNSUPPORTED;
  }
  *out = o;
  return DYNENV_OK;
}

void oracle_destroy(oracle_t* o) {
  if (!o) return;
  free(o->drv);
  free(o->rc);
  free(o);
}

void oracle_set_threads(oracle_t* o, int n) { o->threads = n > 0 ? n : 1; }

int oracle_layout(const oracle_t* o, dynenv_layout_t* L) {
  int A = o->n_agents;
  memset(L, 0, sizeof(*L));
  L->num_envs = o->cfg.num_envs; L->n_agents = A; L->n_time_steps = o->n_time_steps; L->obs_dim = o->obs_dim;
  L->action_dim = o->action_dim;
  if (o->cfg.env_type == DYNENV_DRIVE && o->cfg.obs_type == DYNENV_OBS_PARTIAL) {
    /* ((cars, obstacles, pedestrians), (self, lanes)) rows of getAgentVision + 4 row counts (DrivingEnvironment.py:977) */
    L->n_blocks = 6;
    L->block_offset[0] = 0; L->block_rows[0] = 1; L->block_feat[0] = 9;
    L->block_offset[1] = 9; L->block_rows[1] = 24; L->block_feat[1] = 7;
    L->block_offset[2] = 9 + 24 * 7; L->block_rows[2] = 32; L->block_feat[2] = 6;
    L->block_offset[3] = L->block_offset[2] + 32 * 6; L->block_rows[3] = 40; L->block_feat[3] = 2;
    L->block_offset[4] = L->block_offset[3] + 40 * 2; L->block_rows[4] = 16; L->block_feat[4] = 4;
    L->block_offset[5] = L->block_offset[4] + 16 * 4; L->block_rows[5] = 1; L->block_feat[5] = 4;
    L->steps_per_episode = DRV_MAX_TIME / DRV_STEP_ITER;
  } else if (o->cfg.env_type == DYNENV_DRIVE) {
    L->n_blocks = 5;
    L->block_offset[0] = 0; L->block_rows[0] = 1; L->block_feat[0] = 9;
    L->block_offset[1] = 9; L->block_rows[1] = A - 1; L->block_feat[1] = 7;
    L->block_offset[2] = 9 + (A - 1) * 7; L->block_rows[2] = DYNENV_MAX_OBST; L->block_feat[2] = 4;
    L->block_offset[3] = L->block_offset[2] + DYNENV_MAX_OBST * 4; L->block_rows[3] = DYNENV_MAX_PEDS; L->block_feat[3] = 2;
    L->block_offset[4] = L->block_offset[3] + DYNENV_MAX_PEDS * 2; L->block_rows[4] = DYNENV_DRIVE_LANES; L->block_feat[4] = 5;
    L->steps_per_episode = DRV_MAX_TIME / DRV_STEP_ITER;
  } else if (o->cfg.obs_type == DYNENV_OBS_PARTIAL) {
    /* ((balls, robots), (goals, crosses, line crosses, lines), (numLandMarks, robotsSeen, ballsSeen)) of getAgentVision; block 6 =
     * the tail: 6 list lengths, numLandMarks, ballsSeen, robotsSeen[9] (oracle/robocup_partial.h) */
    const int off[7] = {RCP_OFF_BALL, RCP_OFF_ROB, RCP_OFF_GOAL, RCP_OFF_CROSS, RCP_OFF_FCROSS, RCP_OFF_LINE, RCP_OFF_TAIL};
    const int rows[7] = {RCP_CAP_BALL, RCP_CAP_ROB, RCP_CAP_GOAL, RCP_CAP_CROSS, RCP_CAP_FCROSS, RCP_CAP_LINE, 1};
    const int feat[7] = {5, 7, 6, 6, 8, 5, 17};
    int i;
    L->n_blocks = 7;
    for (i = 0; i < 7; ++i) { L->block_offset[i] = off[i]; L->block_rows[i] = rows[i]; L->block_feat[i] = feat[i]; }
    L->steps_per_episode = RC_MAX_TIME / RC_STEP_ITER;
  } else {
    L->n_blocks = 3; /* ((ball, robots), (self,)) of RoboCupEnvironment.py:440-443 */
    L->block_offset[0] = 0; L->block_rows[0] = 1; L->block_feat[0] = 4;
    L->block_offset[1] = 4; L->block_rows[1] = 1; L->block_feat[1] = 8;
    L->block_offset[2] = 12; L->block_rows[2] = A - 1; L->block_feat[2] = 6;
    L->steps_per_episode = RC_MAX_TIME / RC_STEP_ITER;
  }
  return DYNENV_OK;
}

int oracle_reset(oracle_t* o, float* obs) {
  int e, E = o->cfg.num_envs;
  size_t stride = (size_t)o->n_time_steps * o->n_agents * o->obs_dim;
#pragma omp parallel for schedule(static) num_threads(o->threads)
  for (e = 0; e < E; ++e) {
    if (o->drv) {
      drv_reset(&o->drv[e]);
      if (obs) drv_write_obs(&o->drv[e], obs + e * stride);
    } else {
      int t;
      rc_reset(&o->rc[e]);
      /* environment_base.py:217-222: nTimeSteps identical copies of the initial observation */
      if (obs) for (t = 0; t < o->n_time_steps; ++t) {
        float* dst = obs + e * stride + (size_t)t * o->n_agents * o->obs_dim;
        /* Partial: nTimeSteps separate getAgentVision calls, each with fresh noise (draw keys: time word = t) */
        if (o->rc[e].obsType == DYNENV_OBS_PARTIAL) rc_write_partial_obs(&o->rc[e], (uint32_t)t, dst);
        else rc_write_full_obs(&o->rc[e], dst);
      }
    }
  }
  return DYNENV_OK;
}

/* head: the continuous head channel [E][A] of RoboCup with allowHeadTurn (RoboCupEnvironment.py:339-342), or NULL */
int oracle_step_head(oracle_t* o, const int32_t* actions, const double* head, float* obs, double* rewards, uint8_t* dones);
int oracle_step(oracle_t* o, const int32_t* actions, float* obs, double* rewards, uint8_t* dones) {
  return oracle_step_head(o, actions, 0, obs, rewards, dones);
}
int oracle_step_head(oracle_t* o, const int32_t* actions, const double* head, float* obs, double* rewards, uint8_t* dones) {
  int e, E = o->cfg.num_envs, A = o->n_agents;
  size_t stride = (size_t)o->n_time_steps * A * o->obs_dim;
#pragma omp parallel for schedule(static) num_threads(o->threads)
  for (e = 0; e < E; ++e) {
    if (o->rc) o->rc[e].headActions = head ? head + (size_t)e * A : 0;
    int d = o->drv ? drv_step(&o->drv[e], actions + (size_t)e * A * 2, obs ? obs + e * stride : 0, rewards + (size_t)e * A)
                   : rc_step(&o->rc[e], actions + (size_t)e * A * 4, obs ? obs + e * stride : 0, rewards + (size_t)e * A);
    dones[e] = (uint8_t)d;
  }
  return DYNENV_OK;
}

int oracle_counts(oracle_t* o, int32_t* counts) {
  int e;
  for (e = 0; e < o->cfg.num_envs; ++e) {
    counts[2 * e] = o->drv ? o->drv[e].nObst : 0;
    counts[2 * e + 1] = o->drv ? o->drv[e].nPeds : 0;
  }
  return DYNENV_OK;
}

int oracle_episode_stats(oracle_t* o, double* ep_r, double* ep_pos_r, double* ep_obs_r, int32_t* goals) {
  int e, a, A = o->n_agents;
  if (o->rc) {
    for (e = 0; e < o->cfg.num_envs; ++e) {
      const RoboCupEnv* r = &o->rc[e];
      for (a = 0; a < A; ++a) {
        if (ep_r) ep_r[e * A + a] = r->episodeRewards[a];
        if (ep_pos_r) ep_pos_r[e * A + a] = r->episodePosRewards[a];
        if (ep_obs_r) ep_obs_r[e * A + a] = r->episodeObsRewards[a];
      }
      if (goals) { goals[2 * e] = r->goals[0]; goals[2 * e + 1] = r->goals[1]; }
    }
    return DYNENV_OK;
  }
  for (e = 0; e < o->cfg.num_envs; ++e) {
    const DrivingEnv* d = &o->drv[e];
    int fin = 0, crashed = 0;
    for (a = 0; a < A; ++a) {
      if (ep_r) ep_r[e * A + a] = d->episodeRewards[a];
      if (ep_pos_r) ep_pos_r[e * A + a] = d->episodePosRewards[a];
      if (ep_obs_r) ep_obs_r[e * A + a] = 0.0; /* DrivingEnvironment.py:314 */
      fin += d->cars[a].finished && !d->cars[a].crashed;
      crashed += d->cars[a].crashed;
    }
    if (goals) { goals[2 * e] = fin; goals[2 * e + 1] = crashed; } /* :315-316 */
  }
  return DYNENV_OK;
}

size_t oracle_state_size(const oracle_t* o) { return o->drv ? sizeof(dynenv_driving_state_t) : sizeof(dynenv_robocup_state_t); }
int oracle_get_state(oracle_t* o, int32_t env, void* blob, size_t n) {
  if (env < 0 || env >= o->cfg.num_envs || n < oracle_state_size(o)) return DYNENV_ERR_ARG;
  if (o->drv) drv_get_state(&o->drv[env], (dynenv_driving_state_t*)blob);
  else rc_get_state(&o->rc[env], (dynenv_robocup_state_t*)blob);
  return DYNENV_OK;
}
int oracle_set_state(oracle_t* o, int32_t env, const void* blob, size_t n) {
  if (env < 0 || env >= o->cfg.num_envs || n < oracle_state_size(o)) return DYNENV_ERR_ARG;
  if (o->drv) drv_set_state(&o->drv[env], (const dynenv_driving_state_t*)blob);
  else rc_set_state(&o->rc[env], (const dynenv_robocup_state_t*)blob);
  return DYNENV_OK;
}
int oracle_overflow(oracle_t* o) {
  int e, f = 0;
  for (e = 0; e < o->cfg.num_envs; ++e) f |= o->drv ? o->drv[e].space.overflow : o->rc[e].space.overflow;
  return f;
}
int oracle_degenerate(oracle_t* o) { /* 16 (error bit 4) if an environment's narrowphase took a fallback normal since its last reset / set_state */
  int e, f = 0;
  for (e = 0; e < o->cfg.num_envs; ++e) f |= o->drv ? o->drv[e].space.degenerate : o->rc[e].space.degenerate;
  return f ? 16 : 0;
}
int oracle_degenerate_env(oracle_t* o, int32_t env) { return (o->drv ? o->drv[env].space.degenerate : o->rc[env].space.degenerate) ? 16 : 0; }
int oracle_obs_overflow(oracle_t* o) {
  int e, f = 0;
  if (o->drv) for (e = 0; e < o->cfg.num_envs; ++e) f |= o->drv[e].obsOverflow;
  return f;
}
int oracle_obs_noise_draws_max(oracle_t* o) { /* most rows that drew noise in one Driving Partial vision pass so far */
  int e, m = 0;
  if (o->drv) for (e = 0; e < o->cfg.num_envs; ++e) if (o->drv[e].obsNoiseDrawsMax > m) m = o->drv[e].obsNoiseDrawsMax;
  return m;
}
void oracle_drv_vision(oracle_t* o, int env, int agent, float* out) {
  DrivingEnv* d = &o->drv[env];
  d->obsOverflow |= drv_agent_vision(d, agent, d->noiseType, d->noiseMagnitude, out) & 0xFF;
}
int oracle_active_contacts(oracle_t* o, int32_t env) { return o->drv ? o->drv[env].space.n_active : o->rc[env].space.n_active; }

/* ---- unit entry points for the golden tests (tests/test_oracle_golden.py) ---- */
void oracle_math(const double* x, const double* y, int n, double* out) {
  int i;
  for (i = 0; i < n; ++i) {
    double s, c;
    dm_sincos(x[i], &s, &c);
    out[5 * i + 0] = s; out[5 * i + 1] = c; out[5 * i + 2] = dm_atan2(y[i], x[i]);
    out[5 * i + 3] = dm_sqrt(dm_abs(x[i])); out[5 * i + 4] = (y[i] != 0.0) ? x[i] / y[i] : 0.0;
  }
}
void oracle_math_f(const double* x, int n, double* out) { /* dm_sincos_f: out[2 i] = sin, out[2 i + 1] = cos */
  int i;
  for (i = 0; i < n; ++i) dm_sincos_f(x[i], &out[2 * i], &out[2 * i + 1]);
}
void oracle_philox(uint32_t k0, uint32_t k1, const uint32_t* ctr, uint32_t* out) {
  dm_u32x4 r = dm_philox(k0, k1, ctr[0], ctr[1], ctr[2], ctr[3]);
  memcpy(out, r.v, 16);
}
void oracle_apply_friction(double m, double* vxyw, double friction, double rotFriction, double spin) {
  cpBody b;
  cpBodyInit(&b, m, 1.0, CP_BODY_DYNAMIC);
  b.v = cpv_(vxyw[0], vxyw[1]); b.w = vxyw[2];
  apply_friction(&b, cpv_(0.0, 0.0), 1.0, 0.01, friction, rotFriction, spin);
  vxyw[0] = b.v.x; vxyw[1] = b.v.y; vxyw[2] = b.w;
}
int oracle_road_is_point_on_road(int road, double x, double y, double angle) {
  DrivingEnv e;
  drv_init(&e, 2, 0, 0);
  return road_is_point_on_road(&e.roads[road], cpv_(x, y), angle);
}
void oracle_road_geometry(int road, double* out /* dir2 normal2 length dirAngle lanes[5][4] walk[2][4] */) {
  DrivingEnv e; const Road* r; int i, k = 0;
  drv_init(&e, 2, 0, 0);
  r = &e.roads[road];
  out[k++] = r->dir.x; out[k++] = r->dir.y; out[k++] = r->normal.x; out[k++] = r->normal.y;
  out[k++] = r->length; out[k++] = r->dirAngle;
  for (i = 0; i < 5; ++i) { out[k++] = r->lanes[i][0].x; out[k++] = r->lanes[i][0].y; out[k++] = r->lanes[i][1].x; out[k++] = r->lanes[i][1].y; }
  for (i = 0; i < 2; ++i) { out[k++] = r->walk[i][0].x; out[k++] = r->walk[i][0].y; out[k++] = r->walk[i][1].x; out[k++] = r->walk[i][1].y; }
}
void oracle_road_get_spot(int road, int lane, int spot, double* out3) {
  DrivingEnv e; cpv p; double a;
  drv_init(&e, 2, 0, 0);
  road_get_spot(&e.roads[road], lane, spot, &p, &a);
  out3[0] = p.x; out3[1] = p.y; out3[2] = a;
}
void oracle_road_get_walk_spot(int road, int side, double length, double width, double* out2) {
  DrivingEnv e; cpv p;
  drv_init(&e, 2, 0, 0);
  p = road_get_walk_spot(&e.roads[road], side, length, width);
  out2[0] = p.x; out2[1] = p.y;
}
/* run single game-logic functions on env `env` (state injected through oracle_set_state) */
void oracle_drv_process_action(oracle_t* o, int env, int car, const int32_t* action) { drv_process_action(&o->drv[env], car, action); }
double oracle_drv_tick(oracle_t* o, int env, int car) {
  DrivingEnv* e = &o->drv[env];
  e->carRewards[car] = 0.0; e->carPosRewards[car] = 0.0;
  drv_tick(e, car);
  return e->carRewards[car];
}
double oracle_drv_pos_reward(oracle_t* o, int env, int car) { return o->drv[env].carPosRewards[car]; }
void oracle_drv_move(oracle_t* o, int env, int ped) { drv_move(&o->drv[env], ped); }
void oracle_drv_lane_rows(oracle_t* o, float* out40) { memcpy(out40, o->drv[0].laneRows, sizeof(o->drv[0].laneRows)); }
double oracle_moment_for_box(double m, double hx, double hy) { return cpMomentForBox(m, hx, hy); }
double oracle_moment_for_circle(double m, double r1, double r2) { return cpMomentForCircle(m, r1, r2); }
double oracle_moment_for_segment(double m, double ax, double ay, double bx, double by, double r) {
  return cpMomentForSegment(m, cpv_(ax, ay), cpv_(bx, by), r);
}

/* ---- two-body physics sandbox for closed-form KATs (tests/test_oracle_physics.py) ----
 * shape kind: 0 circle(radius=sx), 1 capsule (half-length sx, radius sy, along local x), 2 box (half extents sx,sy)
 * in[b] = {kind, sx, sy, mass(<=0: static), px, py, vx, vy, angle, w, e, u}; out[b] = {px,py,vx,vy,angle,w}; returns
 * the number of substeps in which the pair had an active arbiter. */
int oracle_sandbox_two_body(const double* in, int steps, double* out) {
  cpSpace sp; cpBody body[2]; cpShape shape[2];
  int b, s, touched = 0;
  cpSpaceInit(&sp);
  for (b = 0; b < 2; ++b) {
    const double* p = in + 12 * b;
    int kind = (int)p[0];
    double m = p[3], I;
    if (kind == 0) I = cpMomentForCircle(m, 0.0, p[1]);
    else if (kind == 1) I = cpMomentForSegment(m, cpv_(-p[1], 0.0), cpv_(p[1], 0.0), p[2]);
    else I = cpMomentForBox(m, p[1], p[2]);
    cpBodyInit(&body[b], m, I, m > 0.0 ? CP_BODY_DYNAMIC : CP_BODY_STATIC);
    body[b].p = cpv_(p[4], p[5]); body[b].v = cpv_(p[6], p[7]); body[b].w = p[9];
    cpBodySetAngle(&body[b], p[8]);
    if (kind == 0) cpCircleInit(&shape[b], &body[b], p[1], b);
    else if (kind == 1) cpSegmentInit(&shape[b], &body[b], cpv_(-p[1], 0.0), cpv_(p[1], 0.0), p[2], b);
    else cpBoxInit(&shape[b], &body[b], p[1], p[2], b);
    shape[b].e = p[10]; shape[b].u = p[11]; shape[b].collision_type = b;
    cpSpaceAddBody(&sp, &body[b]);
    cpSpaceAddShape(&sp, &shape[b]);
  }
  for (s = 0; s < steps; ++s) { cpSpaceStep(&sp, 0.01); touched += sp.n_active > 0; }
  for (b = 0; b < 2; ++b) {
    out[6 * b + 0] = body[b].p.x; out[6 * b + 1] = body[b].p.y; out[6 * b + 2] = body[b].v.x;
    out[6 * b + 3] = body[b].v.y; out[6 * b + 4] = body[b].a; out[6 * b + 5] = body[b].w;
  }
  return touched;
}

/* two capsule bodies joined like Robot.py:58-60 (PivotJoint error_bias=0.1 + RotaryLimitJoint(0,0)); the left one
 * gets velocity (vx,vy) and angular velocity w0; returns state of both after `steps` substeps. */
void oracle_sandbox_robot_joint(double vx, double vy, double w0, int steps, double* out) {
  cpSpace sp; cpBody body[2]; cpShape shape[2]; cpConstraint pivot, rot;
  int b, s;
  cpSpaceInit(&sp);
  for (b = 0; b < 2; ++b) {
    double y = b ? -10.0 : 10.0;
    double I = cpMomentForSegment(4000.0, cpv_(-10.0, y), cpv_(10.0, y), 7.5);
    cpBodyInit(&body[b], 4000.0, I, CP_BODY_DYNAMIC);
    body[b].p = cpv_(100.0, 100.0);
    cpSegmentInit(&shape[b], &body[b], cpv_(-10.0, y), cpv_(10.0, y), 7.5, b);
    shape[b].e = 0.3; shape[b].u = 2.5;
    cpSpaceAddBody(&sp, &body[b]); cpSpaceAddShape(&sp, &shape[b]);
  }
  cpPivotJointInit(&pivot, &body[0], &body[1], cpv_(100.0, 100.0));
  pivot.errorBias = 0.1;
  cpRotaryLimitJointInit(&rot, &body[0], &body[1], 0.0, 0.0);
  cpSpaceAddConstraint(&sp, &pivot); cpSpaceAddConstraint(&sp, &rot);
  body[0].v = cpv_(vx, vy); body[0].w = w0;
  for (s = 0; s < steps; ++s) cpSpaceStep(&sp, 0.01);
  for (b = 0; b < 2; ++b) {
    out[6 * b + 0] = body[b].p.x; out[6 * b + 1] = body[b].p.y; out[6 * b + 2] = body[b].v.x;
    out[6 * b + 3] = body[b].v.y; out[6 * b + 4] = body[b].a; out[6 * b + 5] = body[b].w;
  }
}
double oracle_bias_coef(int which) {
  cpSpace sp; cpSpaceInit(&sp);
  if (which == 0) return 1.0 - pow(sp.collisionBias, 0.01);
  return 1.0 - pow(0.1, 0.01);
}

/* ---- RoboCup unit entry points for the golden tests ---- */
int rc_test_begin(RoboCupEnv* e, int slotA, int slotB);
void rc_test_zero_rewards(RoboCupEnv* e);
void rc_test_get_rewards(const RoboCupEnv* e, double* out22);
void oracle_rc_process_action(oracle_t* o, int env, int robot, const int32_t* action, double* rew22) {
  rc_test_zero_rewards(&o->rc[env]);
  rc_process_action(&o->rc[env], &o->rc[env].robots[robot], action);
  rc_test_get_rewards(&o->rc[env], rew22);
}
void oracle_rc_tick(oracle_t* o, int env, int robot, double* rew22) {
  rc_test_zero_rewards(&o->rc[env]);
  rc_tick(&o->rc[env], &o->rc[env].robots[robot]);
  rc_test_get_rewards(&o->rc[env], rew22);
}
int oracle_rc_ball(oracle_t* o, int env, double* rew22) {
  int f;
  rc_test_zero_rewards(&o->rc[env]);
  f = rc_is_ball_out_of_field(&o->rc[env]);
  rc_test_get_rewards(&o->rc[env], rew22);
  return f;
}
void oracle_rc_penalize(oracle_t* o, int env, int robot, double* rew22) {
  rc_test_zero_rewards(&o->rc[env]);
  rc_penalize(&o->rc[env], &o->rc[env].robots[robot]);
  rc_test_get_rewards(&o->rc[env], rew22);
}
int oracle_rc_begin(oracle_t* o, int env, int slotA, int slotB, double* rew22) {
  int r;
  rc_test_zero_rewards(&o->rc[env]);
  r = rc_test_begin(&o->rc[env], slotA, slotB);
  rc_test_get_rewards(&o->rc[env], rew22);
  return r;
}
int oracle_rc_joint_count(oracle_t* o, int env) { return o->rc[env].space.n_constraints; }
void oracle_rc_obs(oracle_t* o, int env, float* out) { rc_write_full_obs(&o->rc[env], out); }
void oracle_rc_global_state(oracle_t* o, int env, float* out) { rc_write_global_state(&o->rc[env], out); }
void oracle_rc_spots(const double* rnd18, double* out20) {
  cpv spots[2][5]; int t, i;
  rc_spots(rnd18, spots);
  for (t = 0; t < 2; ++t) for (i = 0; i < 5; ++i) { out20[(t * 5 + i) * 2] = spots[t][i].x; out20[(t * 5 + i) * 2 + 1] = spots[t][i].y; }
}

/* test hook: one snapshot of Partial observations of every robot of env, draw keys with time word `tkey` */
int oracle_rc_partial_obs(oracle_t* o, int env, uint32_t tkey, float* out) { return rc_write_partial_obs(&o->rc[env], tkey, out); }
double oracle_rc_process_seens(double lSum, const double* rSum, int nOthers, double bSum) { return rc_process_seens(lSum, rSum, nOthers, bSum); }

/* ---- golden-test hooks added in round 2 (tests/test_oracle_golden_callbacks.py, *_composition.py) ---- */
int rc_test_callback(RoboCupEnv* e, int slotA, int slotB, int which);
/* Driving begin callbacks carCrash / pedHit / carHit (DrivingEnvironment.py:591-683) on the shape pair (slots) */
int oracle_drv_begin(oracle_t* o, int env, int slotA, int slotB, double* rew10) {
  DrivingEnv* e = &o->drv[env];
  int i, r;
  for (i = 0; i < DYNENV_MAX_CARS; ++i) e->carRewards[i] = 0.0;
  r = cpSpaceTestCallback(&e->space, slotA, slotB, 0);
  for (i = 0; i < DYNENV_MAX_CARS; ++i) rew10[i] = e->carRewards[i];
  return r;
}
/* RoboCup handler callbacks: which = 0 begin, 1 post_solve (robotCollision / goalpostCollision), 2 separate */
int oracle_rc_callback(oracle_t* o, int env, int slotA, int slotB, int which, double* rew22) {
  int i, r;
  rc_test_zero_rewards(&o->rc[env]);
  for (i = 0; i < o->rc[env].space.n_shapes; ++i) cpShapeCacheBB(o->rc[env].space.shapes[i]); /* fall() queries the cache */
  r = rc_test_callback(&o->rc[env], slotA, slotB, which);
  rc_test_get_rewards(&o->rc[env], rew22);
  return r;
}
void oracle_rc_fall(oracle_t* o, int env, int robot, int punish, double* rew22) {
  int i;
  rc_test_zero_rewards(&o->rc[env]);
  for (i = 0; i < o->rc[env].space.n_shapes; ++i) cpShapeCacheBB(o->rc[env].space.shapes[i]);
  rc_fall(&o->rc[env], &o->rc[env].robots[robot], punish);
  rc_test_get_rewards(&o->rc[env], rew22);
}
/* every dynamic body's velocity function once (consumes the forces fall() applied), as Space.step would next */
void oracle_velocity_update(oracle_t* o, int env) {
  cpSpace* s = o->drv ? &o->drv[env].space : &o->rc[env].space;
  int i;
  for (i = 0; i < s->n_bodies; ++i) {
    cpBody* b = s->bodies[i];
    if (b->velocity_func) b->velocity_func(b, s->gravity, 1.0, 0.01);
    else cpBodyUpdateVelocity(b, s->gravity, 1.0, 0.01);
  }
}
/* composition fixtures: Space.step = position update + velocity functions only (the generators' Space stand-in) */
void oracle_set_free_flight(oracle_t* o, int on) {
  int e;
  for (e = 0; e < o->cfg.num_envs; ++e) {
    if (o->drv) o->drv[e].space.test_free_flight = on;
    else o->rc[e].space.test_free_flight = on;
  }
}

/* ---- narrowphase of single shape pairs for the differential fuzz against tests/kat_general.py ----
 * desc[9] = {kind (0 circle, 1 capsule, 2 box), px, py, angle, g0..g4}: circle g0 = r; capsule g0..g3 = local ends a, b, g4 = r;
 * box g0, g1 = half extents.  Shape a gets slot 0, shape b slot 1.  out[15] per pair: cpTestCollide's layout. */
static void fuzz_shape(cpBody* body, cpShape* sh, const double* d, int slot) {
  int kind = (int)d[0];
  cpBodyInit(body, 1.0, 1.0, CP_BODY_DYNAMIC);
  body->p = cpv_(d[1], d[2]);
  cpBodySetAngle(body, d[3]);
  if (kind == 0) cpCircleInit(sh, body, d[4], slot);
  else if (kind == 1) cpSegmentInit(sh, body, cpv_(d[4], d[5]), cpv_(d[6], d[7]), d[8], slot);
  else cpBoxInit(sh, body, d[4], d[5], slot);
}
void oracle_cp_collide_batch(int n, const double* descA, const double* descB, double* out) {
  int i;
  for (i = 0; i < n; ++i) {
    cpBody ba, bb; cpShape sa, sb;
    fuzz_shape(&ba, &sa, descA + 9 * i, 0);
    fuzz_shape(&bb, &sb, descB + 9 * i, 1);
    cpTestCollide(&sa, &sb, out + 15 * i);
  }
}

/* differential tests only: n calls of cpSpaceStep(0.01) on one environment's space with NO game logic around them (for locating
 * the substep in which tests/kat_general.py and cp_lite part ways; in the fuzz scenes the game logic is a no-op by construction) */
void oracle_space_step(oracle_t* o, int env, int n) {
  int i;
  for (i = 0; i < n; ++i) cpSpaceStep(o->drv ? &o->drv[env].space : &o->rc[env].space, 0.01);
}
/* the active arbiters after the last cpSpaceStep: per arbiter slotA slotB count n.x n.y, then per contact r1 r2 jnAcc jtAcc (2 x 6) */
int oracle_space_arbiters(oracle_t* o, int env, double* out, int cap) {
  cpSpace* s = o->drv ? &o->drv[env].space : &o->rc[env].space;
  int i, k;
  for (i = 0; i < s->n_active && i < cap; ++i) {
    const cpArbiter* a = s->active[i];
    double* p = out + 17 * i;
    p[0] = a->a->slot; p[1] = a->b->slot; p[2] = a->count; p[3] = a->n.x; p[4] = a->n.y;
    for (k = 0; k < 2; ++k) {
      const cpContact* c = &a->contacts[k];
      p[5 + 6 * k + 0] = c->r1.x; p[5 + 6 * k + 1] = c->r1.y; p[5 + 6 * k + 2] = c->r2.x; p[5 + 6 * k + 3] = c->r2.y;
      p[5 + 6 * k + 4] = c->jnAcc; p[5 + 6 * k + 5] = c->jtAcc;
    }
  }
  return s->n_active;
}

/* how often a capsule pair's cores touched or crossed since the library was loaded (see cp_lite.c) */
extern long cp_lite_cores_cross;
long oracle_cp_cores_cross(void) { return cp_lite_cores_cross; }

/* ---- round 6: test modes of the mini-Chipmunk (oracle/cp_lite.c) and the per-substep trace (tools/pair_order_cost.py, tools/pymunk_crosscheck.py) ---- */
extern int cp_lite_test_reverse_order;
extern double cp_lite_test_nudge;
void oracle_test_modes(int reverse_order, double nudge) { cp_lite_test_reverse_order = reverse_order; cp_lite_test_nudge = nudge; }

#define ORACLE_TRACE_BODIES 32
#define ORACLE_TRACE_ARBS 40
typedef struct { double* states; int32_t* arbs; int cap, n; } oracle_trace_t;
static oracle_trace_t g_trace; /* one traced environment at a time (a single-threaded tool) */
static void trace_step(cpSpace* s, void* data) {
  oracle_trace_t* t = (oracle_trace_t*)data;
  int i;
  if (t->n < t->cap) {
    double* d = t->states + (size_t)t->n * ORACLE_TRACE_BODIES * 6;
    int32_t* a = t->arbs + (size_t)t->n * ORACLE_TRACE_ARBS;
    for (i = 0; i < s->n_bodies && i < ORACLE_TRACE_BODIES; ++i) {
      const cpBody* b = s->bodies[i];
      d[6 * i + 0] = b->p.x; d[6 * i + 1] = b->p.y; d[6 * i + 2] = b->a; d[6 * i + 3] = b->v.x; d[6 * i + 4] = b->v.y; d[6 * i + 5] = b->w;
    }
    a[0] = s->n_active;
    for (i = 0; i < s->n_active && i + 1 < ORACLE_TRACE_ARBS; ++i) a[i + 1] = (s->active[i]->a->slot << 8) | s->active[i]->b->slot;
  }
  t->n += 1;
}
/* After every physics substep of environment `env` from now on (until its space is rebuilt: reset / set_state), record the dynamic
 * bodies' (px, py, angle, vx, vy, w) in the space's body order into states[cap][32][6] and the active arbiters, in solver order, as
 * (slotA << 8 | slotB) of the narrowphase's shape order into arbs[cap][40] ([0] = count).  Returns the number of bodies. */
int oracle_trace_begin(oracle_t* o, int32_t env, double* states, int32_t* arbs, int cap) {
  cpSpace* s = o->drv ? &o->drv[env].space : &o->rc[env].space;
  g_trace.states = states; g_trace.arbs = arbs; g_trace.cap = cap; g_trace.n = 0;
  s->trace_fn = trace_step; s->trace_data = &g_trace;
  return s->n_bodies;
}
int oracle_trace_count(void) { return g_trace.n; }
void oracle_trace_rewind(void) { g_trace.n = 0; }
