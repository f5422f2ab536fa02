/* driving.h — ORACLE (test infrastructure): CPU restatement of DynEnv/DrivingEnvironment.py + Car/Pedestrian/
 * Obstacle/Road/cutils game logic on top of cp_lite.  One environment at a time, scalar fp64. */
#ifndef ORACLE_DRIVING_H
#define ORACLE_DRIVING_H

#include "cp_lite.h"
#include "dynenv.h"

/* LanePosition (cutils.py:143-148) */
enum { LP_AtGoal = 0, LP_InRightLane = 1, LP_InOpposingLane = 2, LP_OverRoad = 3, LP_OffRoad = 4 };
/* CollisionType (cutils.py:68-74) */
enum { CT_Ball = 0, CT_Goalpost = 1, CT_Robot = 2, CT_Car = 3, CT_Pedestrian = 4, CT_Obstacle = 5 };

typedef struct {
  int nLanes;
  double width, length, dirAngle, followDist;
  cpv p0, p1, dir, normal;
  cpv lanes[5][2];
  cpv walk[2][2];
} Road;

typedef struct {
  cpBody body;
  cpShape shape;
  int type, team, finished, crashed, position, fric;
  double width, height;
  cpv direction, prevPos, goal;
} Car;

typedef struct {
  cpBody body;
  cpShape shape;
  int road, side, dead, moving, speed, crossing, beginCrossing;
  cpv direction, normal;
} Ped;

typedef struct {
  cpBody body;
  cpShape shape;
  double w, h;
} Obst;

typedef struct DrivingEnv {
  cpSpace space;
  Road roads[2];
  Car cars[DYNENV_MAX_CARS];
  Ped peds[DYNENV_MAX_PEDS];
  Obst obst[DYNENV_MAX_OBST];
  Obst buildings[4];
  int nPlayers, nPeds, nObst;
  int elapsed, allFinished;
  double teamReward;
  double carRewards[DYNENV_MAX_CARS], carPosRewards[DYNENV_MAX_CARS];
  double episodeRewards[DYNENV_MAX_CARS], episodePosRewards[DYNENV_MAX_CARS];
  uint64_t seed;
  uint32_t genv, episode;
  float laneRows[DYNENV_DRIVE_LANES][5];
  int obsType, noiseType; /* ObservationType / NoiseType (cutils.py:29-51) */
  double noiseMagnitude;
  int obsOverflow;
  int obsNoiseDrawsMax; /* most rows that drew noise in one agent's vision pass so far (test instrumentation) */
} DrivingEnv;

#define DRV_W 1700.0
#define DRV_H 1000.0
#define DRV_MAX_TIME 6000
#define DRV_STEP_ITER 10
#define DRV_TIME_DIFF 10
#define DRV_DIST_THRESHOLD 100.0

/* shape slots = canonical collision order: cars 0..9, pedestrians 10..29, obstacles 30..49, buildings 50..53 */
#define DRV_SLOT_CAR 0
#define DRV_SLOT_PED 10
#define DRV_SLOT_OBST 30
#define DRV_SLOT_BUILDING 50

int drv_obs_dim(int nPlayers);
void drv_init(DrivingEnv* e, int nPlayers, uint64_t seed, uint32_t genv);
void drv_reset(DrivingEnv* e); /* new episode: scene re-randomisation (environment_base.py:205-211) */
void drv_write_full_obs(const DrivingEnv* e, float* out /* [A][obs_dim] */);
int drv_partial_obs_dim(void);
int drv_agent_vision(const DrivingEnv* e, int agentIdx, int noiseType, double magn, float* out);
void drv_write_obs(DrivingEnv* e, float* out); /* Full or Partial according to e->obsType */
int drv_env_obs_dim(const DrivingEnv* e);
/* returns done flag; rewards[A] */
int drv_step(DrivingEnv* e, const int32_t* actions /* [A][2] */, float* obs, double* rewards);
void drv_get_state(const DrivingEnv* e, dynenv_driving_state_t* st);
void drv_set_state(DrivingEnv* e, const dynenv_driving_state_t* st);

/* pieces exported for golden tests */
void road_init(Road* r, int nLanes, double width, cpv p0, cpv p1);
int road_is_point_on_road(const Road* r, cpv point, double angle);
void road_get_spot(const Road* r, int lane, int spot, cpv* pos, double* angle);
cpv road_get_walk_spot(const Road* r, int side, double length, double width);
void apply_friction(cpBody* body, cpv gravity, double damping, double dt, double friction, double rotFriction,
                    double spin);
void car_accelerate(Car* c, int dir);
void car_turn(Car* c, int dir);
void drv_tick(DrivingEnv* e, int index);
void drv_move(DrivingEnv* e, int k);
void drv_process_action(DrivingEnv* e, int index, const int32_t* action);

#endif
