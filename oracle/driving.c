/* driving.c — ORACLE (test infrastructure; never on the product path).
 * CPU restatement of DynEnv/DrivingEnvironment.py (+ Car.py, Pedestrian.py, Obstacle.py, Road.py, cutils.py
 * friction/normalisers).  Every function cites the reference lines it follows.  RNG: the reference's serial
 * CPython/NumPy streams are replaced by Philox4x32-10 keyed per (seed, env, episode, purpose, entity, time) —
 * see include/dynenv_math.h and DESIGN.md "RNG".  Game logic is pinned by tests/golden/driving_*.json
 * (generated from the reference's own Python through tests/golden/gen_golden*.py - incl. whole step() trajectories WITH collisions, gen_golden_contacts.py);
 * the physics underneath (cp_lite: Chipmunk's Space.step) is PARITY UNPINNED against pymunk itself - see cp_lite.h for what stands behind it. */
#include "driving.h"

#include <math.h>
#include <string.h>

#include "dynenv_math.h"

static const double CAR_MASSES[4] = {1200, 1800, 3500, 5000};  /* Car.py:9 */
static const double CAR_WIDTHS[4] = {5, 6, 7, 8};               /* Car.py:10 */
static const double CAR_LENGTHS[4] = {10, 15, 20, 25};          /* Car.py:11 */
static const double CAR_POWERS[4] = {3, 4, 3, 4};               /* Car.py:12 */
#define CAR_ANGLE_DIFF (DM_PI / 180.0)                          /* Car.py:13 */

static inline double vlen(cpv v) { return dm_sqrt(v.x * v.x + v.y * v.y); } /* Vec2d.length */
static inline cpv vrot(cpv v, double a) {                                     /* Vec2d.rotate / rotated */
  double s, c;
  dm_sincos(a, &s, &c);
  return cpv_(v.x * c - v.y * s, v.x * s + v.y * c);
}

/* ------------------------------------------------------------------ cutils.py:78-140 friction */
void apply_friction(cpBody* body, cpv gravity, double damping, double dt, double friction, double rotFriction,
                    double spin) {
  double m, factor, rotFactor, x, y, length, theta, a0, a1;
  cpBodyUpdateVelocity(body, gravity, damping, dt); /* cutils.py:104 */
  m = body->m;
  factor = friction * m;
  rotFactor = rotFriction * m;
  x = body->v.x;
  y = body->v.y;
  length = 1.0 / (dm_abs(x) + dm_abs(y) + 1e-5);
  theta = body->w;
  a0 = x * factor * length;
  a1 = y * factor * length;
  a0 += a1 * spin * theta; /* quirk C6: a1 update uses the already-updated a0 */
  a1 -= a0 * spin * theta;
  if (dm_abs(x) < factor) x = 0.0; else x -= a0;
  if (dm_abs(y) < factor) y = 0.0; else y -= a1;
  if (dm_abs(theta) < rotFactor) theta = 0.0; else theta -= (theta > 0.0 ? rotFactor : -rotFactor);
  body->v = cpv_(x, y);
  body->w = theta;
}
static void friction_car(cpBody* b, cpv g, double d, double dt) { apply_friction(b, g, d, dt, 5e-5, 1e-5, 0.0); }
static void friction_car_crashed(cpBody* b, cpv g, double d, double dt) { apply_friction(b, g, d, dt, 5e-4, 2e-5, 0.0); }
static void friction_pedestrian_dead(cpBody* b, cpv g, double d, double dt) { apply_friction(b, g, d, dt, 5e-2, 2e-4, 0.0); }

/* ------------------------------------------------------------------ Road.py */
void road_init(Road* r, int nLanes, double width, cpv p0, cpv p1) { /* Road.py:11-33 */
  cpv d = cpvsub(p1, p0);
  int i;
  double k;
  r->nLanes = nLanes; r->width = width; r->p0 = p0; r->p1 = p1; r->followDist = 90.0;
  r->length = vlen(d);
  r->dir = cpv_(d.x / r->length, d.y / r->length);
  r->normal = vrot(r->dir, DM_PI / 2.0);
  r->dirAngle = dm_atan2(r->dir.y, r->dir.x);
  for (i = -nLanes; i <= nLanes; ++i) {
    double s = (double)i * width;
    r->lanes[i + nLanes][0] = cpvadd(p0, cpvmult(r->normal, s));
    r->lanes[i + nLanes][1] = cpvadd(p1, cpvmult(r->normal, s));
  }
  k = (double)(nLanes + 1) * width;
  r->walk[0][0] = cpvadd(p0, cpvmult(r->normal, k)); r->walk[0][1] = cpvadd(p1, cpvmult(r->normal, k));
  r->walk[1][0] = cpvsub(p0, cpvmult(r->normal, k)); r->walk[1][1] = cpvsub(p1, cpvmult(r->normal, k));
}

int road_is_point_on_road(const Road* r, cpv point, double angle) { /* Road.py:74-97 */
  cpv pt = cpvsub(point, r->p0);
  double dist = cpvcross(r->dir, pt);
  int pos;
  double dirDist;
  if (dm_abs(dist) >= (double)r->nLanes * r->width + 5.0) return LP_OffRoad;
  pos = LP_OverRoad;
  dirDist = cpvdot(r->dir, pt);
  if (dirDist >= -10.0 && dirDist <= r->length + 10.0) {
    double relAngle = dm_cos(r->dirAngle - angle) * dist;
    pos = relAngle < 0.0 ? LP_InRightLane : LP_InOpposingLane;
  }
  return pos;
}

void road_get_spot(const Road* r, int lane, int spot, cpv* pos, double* angle) { /* Road.py:100-114 */
  int end = lane >= r->nLanes ? 1 : 0;
  cpv p = end ? r->p1 : r->p0;
  cpv spotDir = cpvmult(end ? cpvneg(r->dir) : r->dir, r->followDist);
  cpv laneDir = cpvmult(end ? r->normal : cpvneg(r->normal), r->width);
  double l = (double)(end ? lane - r->nLanes : lane) + 0.5;
  *pos = cpvadd(cpvadd(p, cpvmult(laneDir, l)), cpvmult(spotDir, (double)spot));
  *angle = dm_atan2(spotDir.y, spotDir.x);
}

cpv road_get_walk_spot(const Road* r, int side, double length, double width) { /* Road.py:117-123 */
  cpv w0 = r->walk[side][0], w1 = r->walk[side][1];
  cpv center = cpvadd(w0, cpvmult(cpvsub(w1, w0), length));
  double f = width * r->width;
  cpv off = cpvmult(cpvmult(r->normal, f), side ? 1.0 : -1.0);
  return cpvadd(center, off);
}

/* ------------------------------------------------------------------ Car.py */
void car_accelerate(Car* c, int dir) { /* Car.py:55-94, categorical branch */
  double power, moveDir, s, co;
  cpv velocity;
  if (c->finished) return;
  power = (double)dir;
  moveDir = cpvdot(c->body.v, c->direction);
  if (dir < 0) power = (double)dir * 0.75;
  if (dir == 0) power = (moveDir == 0.0) ? 0.0 : (moveDir > 0.0 ? -2.0 : 2.0);
  else if (dir < 0 && moveDir > 0.0) return;
  else if (dir > 0 && moveDir < 0.0) return;
  dm_sincos(c->body.a, &s, &co);
  velocity = cpv_(CAR_POWERS[c->type] * power * co, CAR_POWERS[c->type] * power * s); /* Vec2d(p,0).rotate(angle) */
  c->body.v = cpvadd(c->body.v, velocity);
  if (dir == 0 && cpvdot(c->body.v, c->direction) * moveDir < 0.0) c->body.v = cpv_(0.0, 0.0);
}

void car_turn(Car* c, int dir) { /* Car.py:97-108 */
  double rot;
  if (c->finished) return;
  rot = (double)dir * CAR_ANGLE_DIFF;
  cpBodySetAngle(&c->body, c->body.a + rot);
  c->direction = vrot(c->direction, rot);
  c->body.v = vrot(c->body.v, rot);
}

static void car_crash(Car* c) { /* Car.py:111-117 */
  c->finished = 1; c->crashed = 1; c->fric = 1;
  c->body.velocity_func = friction_car_crashed;
}

/* ------------------------------------------------------------------ DrivingEnvironment.py game logic */
void drv_process_action(DrivingEnv* e, int index, const int32_t* action) { /* :357-373 */
  int acc = action[0] - 1;
  int steer = (action[1] - 1) * 2;
  /* action_space is MultiDiscrete([3, 3]); the reference raises on a malformed action (:365-368), the batched step lets the
   * car coast instead (the HIP kernel also raises its environment's error flag) */
  if (action[0] < 0 || action[0] > 2 || action[1] < 0 || action[1] > 2) { acc = 0; steer = 0; }
  car_accelerate(&e->cars[index], acc);
  if (steer != 0) car_turn(&e->cars[index], steer);
}

void drv_tick(DrivingEnv* e, int index) { /* :376-426 */
  Car* car = &e->cars[index];
  cpv pos = car->body.p;
  double diff;
  int r;
  car->position = LP_OffRoad;
  for (r = 0; r < 2; ++r) {
    int rp = road_is_point_on_road(&e->roads[r], pos, car->body.a);
    if (rp < car->position) car->position = rp;
  }
  diff = vlen(cpvsub(car->prevPos, car->goal)) - vlen(cpvsub(pos, car->goal));
  if (!car->finished) {
    e->carRewards[index] += diff / 50.0;
    e->carPosRewards[index] += dm_max(0.0, diff / 50.0);
  }
  car->prevPos = pos;
  if (car->position >= LP_OverRoad) {
    if (!car->finished) {
      if (car->position == LP_OverRoad && vlen(cpvsub(pos, car->goal)) < DRV_DIST_THRESHOLD) {
        car->position = LP_AtGoal;
        car->finished = 1;
        e->carRewards[index] += (double)(DRV_MAX_TIME - e->elapsed) / 100.0;
        e->carPosRewards[index] += (double)(DRV_MAX_TIME - e->elapsed) / 100.0;
        car->body.velocity_func = friction_car_crashed;
        car->fric = 1;
      } else {
        car_crash(car);
        e->carRewards[index] -= vlen(car->body.v) / 5.0;
      }
    }
  } else if (car->position == LP_InOpposingLane) {
    if (!car->finished) e->carRewards[index] -= vlen(car->body.v) / 10000.0;
  }
  /* :414-426 — quirk C2: the position clamp writes to a temporary Vec2d; only the velocity is zeroed */
  if (car->prevPos.x >= DRV_W + 50.0) car->body.v = cpv_(0.0, 0.0);
  if (car->prevPos.x <= -50.0) car->body.v = cpv_(0.0, 0.0);
  if (car->prevPos.y >= DRV_H + 50.0) car->body.v = cpv_(0.0, 0.0);
  if (car->prevPos.y <= -50.0) car->body.v = cpv_(0.0, 0.0);
}

static int drv_is_off_road(const DrivingEnv* e, cpv point) { /* :509-520 */
  int position = LP_OffRoad, r;
  for (r = 0; r < 2; ++r) {
    int rp = road_is_point_on_road(&e->roads[r], point, 0.0);
    if (rp < position) position = rp;
  }
  return position >= LP_OverRoad;
}
static int drv_is_out(cpv pos) { return pos.x <= 0.0 || pos.y <= 0.0 || pos.x >= DRV_W || pos.y >= DRV_H; } /* :523-524 */

void drv_move(DrivingEnv* e, int k) { /* :429-506 */
  Ped* ped = &e->peds[k];
  int isOffRoad, isOut;
  if (ped->dead) return;
  isOffRoad = drv_is_off_road(e, ped->body.p);
  isOut = drv_is_out(ped->body.p);
  if (ped->moving > 0) {
    ped->moving = ped->moving - DRV_TIME_DIFF > 0 ? ped->moving - DRV_TIME_DIFF : 0;
    if (ped->crossing) {
      if (!ped->beginCrossing && isOffRoad) {
        ped->moving = 0; ped->crossing = 0; ped->body.v = cpv_(0.0, 0.0);
      } else if (ped->beginCrossing && !isOffRoad) {
        ped->beginCrossing = 0;
      }
    }
    if (isOut) { ped->moving = 0; ped->body.v = cpv_(0.0, 0.0); }
  } else {
    if (!ped->crossing) {
      /* :468-499 random.randint(5000,30000), randint(-2,2), random()<0.05, randint(1,2) -> one Philox block */
      dm_u32x4 u = dm_env_rng(e->seed, e->genv, e->episode, DM_RNG_PED_MOVE, (uint32_t)k, (uint32_t)e->elapsed);
      int speed;
      cpv dir = ped->direction;
      ped->moving = dm_randint(u.v[0], 5000, 30000);
      speed = dm_randint(u.v[1], -2, 2);
      if (!isOffRoad) {
        ped->crossing = 1; ped->beginCrossing = 0;
        if (speed == 0) speed = 2;
      } else if (isOut) {
        dir = drv_is_out(cpvadd(ped->body.p, ped->direction)) ? cpvneg(ped->direction) : ped->direction;
      } else if (dm_unit(u.v[2]) < 0.05) {
        ped->crossing = 1; ped->beginCrossing = 1;
        dir = ped->side ? ped->normal : cpvneg(ped->normal);
        ped->side = ped->side ? 0 : 1;
        speed = dm_randint(u.v[3], 1, 2);
      }
      ped->body.v = cpvmult(cpvmult(dir, (double)ped->speed), (double)speed); /* speed * dir * speed */
    } else if (isOffRoad) {
      ped->crossing = 0; ped->beginCrossing = 0;
    }
  }
}

/* ------------------------------------------------------------------ collision callbacks :587-683 */
static int ignore_collision(cpArbiter* arb, cpSpace* s, void* d) { (void)arb; (void)s; (void)d; return 0; }

static int car_crash_cb(cpArbiter* arb, cpSpace* s, void* data) { /* carCrash :591-637 */
  DrivingEnv* e = (DrivingEnv*)data;
  cpShape *s1, *s2;
  Car *car1, *car2;
  int index1, index2;
  double v1l, v2l;
  (void)s;
  cpArbiterGetShapes(arb, &s1, &s2);
  car1 = (Car*)s1->user; car2 = (Car*)s2->user;
  index1 = (int)(car1 - e->cars); index2 = (int)(car2 - e->cars);
  v1l = vlen(car1->body.v) / 5.0;
  v2l = vlen(car2->body.v) / 5.0;
  if (!car1->crashed) e->carRewards[index1] -= v1l;
  if (!car2->crashed) e->carRewards[index2] -= v2l;
  if (car1->position != LP_InRightLane && !car1->crashed) e->carRewards[index1] -= v1l;
  if (car2->position != LP_InRightLane && !car2->crashed) e->carRewards[index2] -= v2l;
  if (car1->position == LP_InRightLane && car2->position == LP_InRightLane) {
    cpv v1 = car1->body.v, v2 = car2->body.v;
    cpv dp = cpvsub(car1->body.p, car2->body.p);
    if (vlen(v1) > 1.0 && dm_cos(dm_atan2(dp.y, dp.x) - dm_atan2(v1.y, v1.x)) < -0.4 && !car1->crashed)
      e->carRewards[index1] -= v1l;
    if (vlen(v2) > 1.0 && dm_cos(dm_atan2(dp.y, dp.x) - dm_atan2(v2.y, v2.x)) > 0.4 && !car2->crashed)
      e->carRewards[index2] -= v2l;
  }
  car_crash(car1);
  car_crash(car2);
  return 1;
}

static void ped_die(Ped* p) { /* Pedestrian.py:40-47 */
  p->moving = 0; p->body.v = cpv_(0.0, 0.0); p->dead = 1;
  p->body.velocity_func = friction_pedestrian_dead;
}

static int ped_hit_cb(cpArbiter* arb, cpSpace* s, void* data) { /* pedHit :640-667 */
  DrivingEnv* e = (DrivingEnv*)data;
  cpShape *s1, *s2;
  Car* car; Ped* ped;
  cpv v1; double v1l;
  (void)s;
  cpArbiterGetShapes(arb, &s1, &s2);
  car = (Car*)s1->user; ped = (Ped*)s2->user;
  v1 = car->body.v;
  v1l = vlen(v1);
  if (v1l > 1.0) {
    cpv dp;
    ped_die(ped);
    dp = cpvsub(car->body.p, ped->body.p);
    if (dm_cos(dm_atan2(dp.y, dp.x) - dm_atan2(v1.y, v1.x)) < -0.4 && !car->finished) {
      int index = (int)(car - e->cars);
      car_crash(car);
      e->carRewards[index] -= v1l / 5.0;
    }
  } else {
    return 0;
  }
  return 1;
}

static int car_hit_cb(cpArbiter* arb, cpSpace* s, void* data) { /* carHit :670-683 */
  DrivingEnv* e = (DrivingEnv*)data;
  cpShape *s1, *s2;
  Car* car; int index;
  (void)s;
  cpArbiterGetShapes(arb, &s1, &s2);
  car = (Car*)s1->user;
  index = (int)(car - e->cars);
  if (!car->finished) e->carRewards[index] -= vlen(car->body.v) / 5.0;
  car_crash(car);
  return 1;
}

/* ------------------------------------------------------------------ scene construction */
static inline double normalize(double pt, double normFactor, double mean) { /* cutils.py:318-323, team=1 */
  return ((pt * normFactor) - mean) * 2.0 * 1.0;
}
#define STD_NORM_X (0.5 / (DRV_W + 100.0))
#define STD_NORM_Y (0.5 / (DRV_H + 100.0))
#define STD_NORM_W (1.0 / 15.0)
#define STD_NORM_H (1.0 / 25.0)

int drv_obs_dim(int nPlayers) { return 9 + (nPlayers - 1) * 7 + DYNENV_MAX_OBST * 4 + DYNENV_MAX_PEDS * 2 + DYNENV_DRIVE_LANES * 5; }

static void setup_car(DrivingEnv* e, int i, cpv center, double angle, int type, int team, cpv goal) { /* Car.py:15-52 */
  Car* c = &e->cars[i];
  double mass = CAR_MASSES[type];
  memset(c, 0, sizeof(*c));
  c->width = CAR_WIDTHS[type]; c->height = CAR_LENGTHS[type];
  c->type = type; c->team = team; c->goal = goal;
  c->direction = vrot(cpv_(1.0, 0.0), angle);
  c->position = LP_OffRoad;
  c->prevPos = center;
  cpBodyInit(&c->body, mass, cpMomentForBox(mass, c->height, c->width), CP_BODY_DYNAMIC);
  c->body.p = center;
  cpBodySetAngle(&c->body, angle);
  c->body.velocity_func = friction_car;
  cpBoxInit(&c->shape, &c->body, c->height, c->width, DRV_SLOT_CAR + i);
  c->shape.e = 0.05; c->shape.collision_type = CT_Car; c->shape.user = c;
}

static void setup_ped(DrivingEnv* e, int k, cpv center, int road, int side, int speed) { /* Pedestrian.py:8-38 */
  Ped* p = &e->peds[k];
  memset(p, 0, sizeof(*p));
  cpBodyInit(&p->body, 90.0, cpMomentForCircle(90.0, 0.0, 2.5 * 2.0), CP_BODY_DYNAMIC);
  p->body.p = center;
  cpCircleInit(&p->shape, &p->body, 2.5 * 2.0, DRV_SLOT_PED + k);
  p->shape.e = 0.05; p->shape.collision_type = CT_Pedestrian; p->shape.user = p;
  p->road = road; p->side = side; p->speed = speed;
  p->direction = e->roads[road].dir; p->normal = e->roads[road].normal;
}

static void setup_static_box(Obst* o, cpv center, double w, double h, int slot) { /* Obstacle.py:8-19 */
  memset(o, 0, sizeof(*o));
  cpBodyInit(&o->body, 0.0, 0.0, CP_BODY_STATIC);
  o->body.p = center;
  o->w = w; o->h = h;
  cpBoxInit(&o->shape, &o->body, w, h, slot);
  o->shape.e = 0.05; o->shape.collision_type = CT_Obstacle; o->shape.user = o;
}

static void build_space(DrivingEnv* e) {
  cpHandler* h;
  int i;
  cpSpaceInit(&e->space); /* environment_base.py:126-128 */
  /* DrivingEnvironment.py:65-74 */
  h = cpSpaceAddHandler(&e->space, CT_Pedestrian, CT_Pedestrian); h->begin = ignore_collision; h->data = e;
  h = cpSpaceAddHandler(&e->space, CT_Pedestrian, CT_Obstacle); h->begin = ignore_collision; h->data = e;
  h = cpSpaceAddHandler(&e->space, CT_Car, CT_Car); h->begin = car_crash_cb; h->data = e;
  h = cpSpaceAddHandler(&e->space, CT_Car, CT_Pedestrian); h->begin = ped_hit_cb; h->data = e;
  h = cpSpaceAddHandler(&e->space, CT_Car, CT_Obstacle); h->begin = car_hit_cb; h->data = e;
  for (i = 0; i < 4; ++i) cpSpaceAddShape(&e->space, &e->buildings[i].shape);
  for (i = 0; i < e->nPlayers; ++i) { cpSpaceAddBody(&e->space, &e->cars[i].body); cpSpaceAddShape(&e->space, &e->cars[i].shape); }
  for (i = 0; i < e->nPeds; ++i) { cpSpaceAddBody(&e->space, &e->peds[i].body); cpSpaceAddShape(&e->space, &e->peds[i].shape); }
  for (i = 0; i < e->nObst; ++i) cpSpaceAddShape(&e->space, &e->obst[i].shape);
}

static void build_lane_rows(DrivingEnv* e) { /* getFullState lanes :689-695 (negative-index quirk kept) */
  int row = 0, r, i;
  for (r = 0; r < 2; ++r) {
    const Road* l = &e->roads[r];
    int n = l->nLanes, cnt = 2 * n + 1;
    for (i = -n; i <= n; ++i) {
      int idx = ((i - n) % cnt + cnt) % cnt; /* python Lanes[i - nLanes] */
      e->laneRows[row][0] = (float)normalize(l->lanes[idx][0].x, STD_NORM_X, 0.0);
      e->laneRows[row][1] = (float)normalize(l->lanes[idx][0].y, STD_NORM_Y, 0.0);
      e->laneRows[row][2] = (float)normalize(l->lanes[idx][1].x, STD_NORM_X, 0.0);
      e->laneRows[row][3] = (float)normalize(l->lanes[idx][1].y, STD_NORM_Y, 0.0);
      e->laneRows[row][4] = (float)((i == n || i == -n) ? 1 : (i == 0 ? -1 : 0));
      ++row;
    }
  }
}

void drv_init(DrivingEnv* e, int nPlayers, uint64_t seed, uint32_t genv) {
  memset(e, 0, sizeof(*e));
  e->nPlayers = nPlayers > DYNENV_MAX_CARS ? DYNENV_MAX_CARS : nPlayers; /* environment_base.py:57 */
  e->seed = seed; e->genv = genv; e->episode = 0;
  /* _create_roads :110-115 */
  road_init(&e->roads[0], 2, 35.0, cpv_(875.0, 0.0), cpv_(875.0, 1000.0));
  road_init(&e->roads[1], 1, 35.0, cpv_(0.0, 500.0), cpv_(1750.0, 500.0));
  build_lane_rows(e);
}

void drv_reset(DrivingEnv* e) { /* environment_base.py:205-211 -> DrivingEnvironment.__init__ :20-56, _setup_scene :58-63 */
  static const cpv BUILDINGS[4] = {{365.0, 200.0}, {365.0, 800.0}, {1385.0, 200.0}, {1385.0, 800.0}};
  int i, A = e->nPlayers;
  int spots[30];
  uint32_t ep = e->episode;
  e->elapsed = 0; e->allFinished = 0; e->teamReward = 0.0;
  for (i = 0; i < DYNENV_MAX_CARS; ++i) {
    e->carRewards[i] = e->carPosRewards[i] = 0.0;
    e->episodeRewards[i] = e->episodePosRewards[i] = 0.0;
  }
  /* _create_buildings :100-108 */
  for (i = 0; i < 4; ++i) setup_static_box(&e->buildings[i], BUILDINGS[i], 400.0, 225.0, DRV_SLOT_BUILDING + i);
  /* _create_agents :88-98 + getUniqueSpots :527-551 (np.random.permutation(30)[:A] -> partial Fisher-Yates) */
  for (i = 0; i < 30; ++i) spots[i] = i;
  for (i = 0; i < A; ++i) {
    dm_u32x4 u = dm_env_rng(e->seed, e->genv, ep, DM_RNG_RESET_PERM, (uint32_t)i, 0);
    int j = i + dm_randint(u.v[0], 0, 29 - i);
    int t = spots[i]; spots[i] = spots[j]; spots[j] = t;
  }
  for (i = 0; i < A; ++i) {
    dm_u32x4 u = dm_env_rng(e->seed, e->genv, ep, DM_RNG_RESET_AGENT, (uint32_t)i, 0);
    int roadSel = dm_randint(u.v[0], 0, 1), endSel = dm_randint(u.v[1], 0, 1);
    int team = dm_randint(u.v[2], 0, 2), type = dm_randint(u.v[3], 0, 3);
    cpv goal = endSel ? e->roads[roadSel].p1 : e->roads[roadSel].p0;
    int spotID = spots[i];
    int roadID = spotID < 20 ? 0 : 1;
    int laneID, spot;
    cpv pos; double angle;
    spotID -= roadID ? 20 : 0;
    laneID = spotID / 5; spot = spotID % 5;
    road_get_spot(&e->roads[roadID], laneID, spot, &pos, &angle);
    setup_car(e, i, pos, angle, type, team, goal);
  }
  {
    dm_u32x4 u = dm_env_rng(e->seed, e->genv, ep, DM_RNG_RESET_COUNTS, 0, 0);
    int nPed = dm_randint(u.v[0], 10, 20), nObstRaw = dm_randint(u.v[1], 10, 20);
    /* createRandomPedestrians :554-566 */
    e->nPeds = nPed;
    for (i = 0; i < nPed; ++i) {
      dm_u32x4 a = dm_env_rng(e->seed, e->genv, ep, DM_RNG_RESET_PED, (uint32_t)i, 0);
      dm_u32x4 b = dm_env_rng(e->seed, e->genv, ep, DM_RNG_RESET_PED, (uint32_t)i, 1);
      int road = dm_randint(a.v[0], 0, 1), side = dm_randint(a.v[1], 0, 1);
      double len = dm_unit(a.v[2]), wid = dm_unit(a.v[3]) / 2.0 + 0.25;
      setup_ped(e, i, road_get_walk_spot(&e->roads[road], side, len, wid), road, side, dm_randint(b.v[0], 3, 6));
    }
    /* createRandomObstacles :569-584 (those on a road are dropped) */
    e->nObst = 0;
    for (i = 0; i < nObstRaw; ++i) {
      dm_u32x4 a = dm_env_rng(e->seed, e->genv, ep, DM_RNG_RESET_OBST, (uint32_t)i, 0);
      int road = dm_randint(a.v[0], 0, 1), side = dm_randint(a.v[1], 0, 1);
      double len = dm_unit(a.v[2]), wid = dm_unit(a.v[3]) / 2.0 + 0.25;
      cpv c = road_get_walk_spot(&e->roads[road], side, len, wid);
      if (drv_is_off_road(e, c)) { setup_static_box(&e->obst[e->nObst], c, 10.0, 10.0, DRV_SLOT_OBST + e->nObst); e->nObst++; }
    }
  }
  build_space(e);
  e->episode++;
}

/* ------------------------------------------------------------------ observations :686-747, :121-124 */
static void car_row(const Car* c, float* o) {
  double s, co;
  dm_sincos(c->body.a, &s, &co);
  o[0] = (float)normalize(c->body.p.x, STD_NORM_X, 0.0);
  o[1] = (float)normalize(c->body.p.y, STD_NORM_Y, 0.0);
  o[2] = (float)co;
  o[3] = (float)s;
  o[4] = (float)normalize(c->width, STD_NORM_W, 0.5);
  o[5] = (float)normalize(c->height, STD_NORM_H, 0.5);
}

void drv_write_full_obs(const DrivingEnv* e, float* out) {
  int A = e->nPlayers, dim = drv_obs_dim(A), a, c, k;
  for (a = 0; a < A; ++a) {
    float* o = out + (size_t)a * dim;
    float* p;
    memset(o, 0, sizeof(float) * dim);
    car_row(&e->cars[a], o);
    o[6] = (float)normalize(e->cars[a].goal.x, STD_NORM_X, 0.0);
    o[7] = (float)normalize(e->cars[a].goal.y, STD_NORM_Y, 0.0);
    o[8] = (float)e->cars[a].finished;
    p = o + 9;
    for (c = 0; c < A; ++c) {
      if (c == a) continue;
      car_row(&e->cars[c], p);
      p[6] = (float)e->cars[c].finished;
      p += 7;
    }
    p = o + 9 + (A - 1) * 7;
    for (k = 0; k < e->nObst; ++k) {
      p[4 * k + 0] = (float)normalize(e->obst[k].body.p.x, STD_NORM_X, 0.0);
      p[4 * k + 1] = (float)normalize(e->obst[k].body.p.y, STD_NORM_Y, 0.0);
      p[4 * k + 2] = (float)normalize(e->obst[k].w, STD_NORM_W, 0.5);
      p[4 * k + 3] = (float)normalize(e->obst[k].h, STD_NORM_H, 0.5);
    }
    p += DYNENV_MAX_OBST * 4;
    for (k = 0; k < e->nPeds; ++k) {
      p[2 * k + 0] = (float)normalize(e->peds[k].body.p.x, STD_NORM_X, 0.0);
      p[2 * k + 1] = (float)normalize(e->peds[k].body.p.y, STD_NORM_Y, 0.0);
    }
    p += DYNENV_MAX_PEDS * 2;
    memcpy(p, e->laneRows, sizeof(e->laneRows));
  }
}

int drv_env_obs_dim(const DrivingEnv* e) { return e->obsType == DYNENV_OBS_PARTIAL ? drv_partial_obs_dim() : drv_obs_dim(e->nPlayers); }

void drv_write_obs(DrivingEnv* e, float* out) { /* :290-294 and environment_base.py:217-222 */
  if (e->obsType == DYNENV_OBS_PARTIAL) {
    int a, dim = drv_partial_obs_dim();
    for (a = 0; a < e->nPlayers; ++a) {
      const int r = drv_agent_vision(e, a, e->noiseType, e->noiseMagnitude, out + (size_t)a * dim);
      e->obsOverflow |= r & 0xFF;
      if ((r >> 8) > e->obsNoiseDrawsMax) e->obsNoiseDrawsMax = r >> 8;
    }
  } else {
    drv_write_full_obs(e, out);
  }
}

/* ------------------------------------------------------------------ step :248-322 */
int drv_step(DrivingEnv* e, const int32_t* actions, float* obs, double* rewards) {
  int A = e->nPlayers, i, a, k;
  e->teamReward = 0.0;
  for (a = 0; a < DYNENV_MAX_CARS; ++a) e->carRewards[a] = e->carPosRewards[a] = 0.0;
  for (i = 0; i < DRV_STEP_ITER; ++i) {
    int allFin = 1;
    for (a = 0; a < A; ++a) {
      if (i % 10 == 0) drv_process_action(e, a, actions + 2 * a);
      drv_tick(e, a);
    }
    for (k = 0; k < e->nPeds; ++k) drv_move(e, k);
    cpSpaceStep(&e->space, 1.0 / 100.0);
    e->elapsed += 1;
    for (a = 0; a < A; ++a) allFin = allFin && (e->cars[a].finished && !e->cars[a].crashed);
    if (!e->allFinished && allFin) {
      e->allFinished = 1;
      e->teamReward += (double)(DRV_MAX_TIME - e->elapsed) / 100.0;
    }
    if (i % 10 == 9 && obs) drv_write_obs(e, obs);
  }
  for (a = 0; a < A; ++a) {
    e->carRewards[a] += e->teamReward;
    e->episodeRewards[a] += e->carRewards[a];
    e->carPosRewards[a] += dm_max(0.0, e->teamReward);
    e->episodePosRewards[a] += e->carPosRewards[a];
    rewards[a] = e->carRewards[a];
  }
  return e->elapsed >= DRV_MAX_TIME;
}

/* ------------------------------------------------------------------ state blob */
void drv_get_state(const DrivingEnv* e, dynenv_driving_state_t* st) {
  int i;
  memset(st, 0, sizeof(*st));
  st->elapsed = e->elapsed; st->all_finished = e->allFinished; st->n_cars = e->nPlayers;
  st->n_peds = e->nPeds; st->n_obst = e->nObst; st->episode = (int32_t)e->episode;
  for (i = 0; i < DYNENV_MAX_CARS; ++i) { st->episode_r[i] = e->episodeRewards[i]; st->episode_pos_r[i] = e->episodePosRewards[i]; }
  for (i = 0; i < e->nPlayers; ++i) {
    const Car* c = &e->cars[i]; dynenv_car_state_t* s = &st->cars[i];
    s->px = c->body.p.x; s->py = c->body.p.y; s->vx = c->body.v.x; s->vy = c->body.v.y; s->angle = c->body.a; s->w = c->body.w;
    s->dirx = c->direction.x; s->diry = c->direction.y; s->prevx = c->prevPos.x; s->prevy = c->prevPos.y;
    s->goalx = c->goal.x; s->goaly = c->goal.y;
    s->type = c->type; s->team = c->team; s->finished = c->finished; s->crashed = c->crashed;
    s->lane_pos = c->position; s->fric = c->fric;
  }
  for (i = 0; i < e->nPeds; ++i) {
    const Ped* p = &e->peds[i]; dynenv_ped_state_t* s = &st->peds[i];
    s->px = p->body.p.x; s->py = p->body.p.y; s->vx = p->body.v.x; s->vy = p->body.v.y;
    s->road = p->road; s->side = p->side; s->dead = p->dead; s->moving = p->moving; s->speed = p->speed;
    s->crossing = p->crossing; s->begin_crossing = p->beginCrossing;
  }
  for (i = 0; i < e->nObst; ++i) { st->obst_x[i] = e->obst[i].body.p.x; st->obst_y[i] = e->obst[i].body.p.y; }
}

void drv_set_state(DrivingEnv* e, const dynenv_driving_state_t* st) {
  static const cpv BUILDINGS[4] = {{365.0, 200.0}, {365.0, 800.0}, {1385.0, 200.0}, {1385.0, 800.0}};
  int i;
  e->elapsed = st->elapsed; e->allFinished = st->all_finished; e->nPlayers = st->n_cars;
  e->nPeds = st->n_peds; e->nObst = st->n_obst; e->episode = (uint32_t)st->episode;
  for (i = 0; i < DYNENV_MAX_CARS; ++i) { e->episodeRewards[i] = st->episode_r[i]; e->episodePosRewards[i] = st->episode_pos_r[i]; }
  for (i = 0; i < 4; ++i) setup_static_box(&e->buildings[i], BUILDINGS[i], 400.0, 225.0, DRV_SLOT_BUILDING + i);
  for (i = 0; i < e->nPlayers; ++i) {
    const dynenv_car_state_t* s = &st->cars[i];
    Car* c;
    setup_car(e, i, cpv_(s->px, s->py), s->angle, s->type, s->team, cpv_(s->goalx, s->goaly));
    c = &e->cars[i];
    c->body.v = cpv_(s->vx, s->vy); c->body.w = s->w;
    c->direction = cpv_(s->dirx, s->diry); c->prevPos = cpv_(s->prevx, s->prevy);
    c->finished = s->finished; c->crashed = s->crashed; c->position = s->lane_pos; c->fric = s->fric;
    c->body.velocity_func = s->fric ? friction_car_crashed : friction_car;
  }
  for (i = 0; i < e->nPeds; ++i) {
    const dynenv_ped_state_t* s = &st->peds[i];
    Ped* p;
    setup_ped(e, i, cpv_(s->px, s->py), s->road, s->side, s->speed);
    p = &e->peds[i];
    p->body.v = cpv_(s->vx, s->vy);
    p->dead = s->dead; p->moving = s->moving; p->crossing = s->crossing; p->beginCrossing = s->begin_crossing;
    if (p->dead) p->body.velocity_func = friction_pedestrian_dead;
  }
  for (i = 0; i < e->nObst; ++i) setup_static_box(&e->obst[i], cpv_(st->obst_x[i], st->obst_y[i]), 10.0, 10.0, DRV_SLOT_OBST + i);
  build_space(e);
}
