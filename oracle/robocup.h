/* robocup.h — ORACLE (test infrastructure): CPU restatement of DynEnv/RoboCupEnvironment.py (+ Robot.py, Ball.py,
 * Goalpost.py) on top of cp_lite.  Full observation (configs[2]) and Partial observation (robocup_partial.c). */
#ifndef ORACLE_ROBOCUP_H
#define ORACLE_ROBOCUP_H

#include "cp_lite.h"
#include "dynenv.h"

#define RC_MAX_PLAYERS 5
#define RC_MAX_ROBOTS 10
#define RC_W 1040.0
#define RC_H 740.0
#define RC_MAX_TIME 12000
#define RC_STEP_ITER 50
#define RC_SIDE 70.0
#define RC_OBS_DIM 66 /* ball 4 + self 8 + 9 others x 6 (for nPlayers = 5) */

/* canonical shape slots: feet 2*id (left) / 2*id+1 (right), ball 20, goalposts 21..24 */
#define RC_SLOT_BALL 20
#define RC_SLOT_POST 21

typedef struct {
  cpBody leftBody, rightBody;
  cpShape leftFoot, rightFoot;
  cpConstraint joint, rotJoint;
  int team, id;
  double headAngle, headMoving;
  cpv prevPos, initPos;
  int penalized, touching, touchCntr, mightPush, fallen, fallCntr, kicking, foot, jointRemoved;
  double penalTime, fallTime, moveTime;
} Robot;

typedef struct RoboCupEnv {
  cpSpace space;
  Robot robots[RC_MAX_ROBOTS];
  cpBody ballBody;
  cpShape ballShape;
  cpv ballPrevPos;
  int lastKicked[4], nLastKicked;
  cpBody postBody[4];
  cpShape postShape[4];
  int nPlayers, nRobots;
  int elapsed;
  int ballOwned;
  double ballFreeCntr, gracePeriod;
  int goals[2], closestID[2];
  int defenders[2][RC_MAX_ROBOTS], nDefenders[2];
  double penalTimes[2];
  double teamRewards[2], robotRewards[RC_MAX_ROBOTS], robotPosRewards[RC_MAX_ROBOTS];
  double episodeRewards[RC_MAX_ROBOTS], episodePosRewards[RC_MAX_ROBOTS];
  int canFall, allowHeadTurn, useObsRewards, randomInit, deterministicTurn; /* class switches RoboCupEnvironment.py:18-21, :24 */
  const double* headActions; /* continuous head channel of this step ([2n], allowHeadTurn: Box(-3,3) :339-342) or NULL */
  int obsType, noiseType;      /* ObservationType / NoiseType (cutils.py:29-51) */
  double noiseMagnitude;
  double episodeObsRewards[RC_MAX_ROBOTS];
  int obsOverflow;
  uint64_t seed;
  uint32_t genv, episode;
} RoboCupEnv;

int rc_obs_dim(int nPlayers);
void rc_init(RoboCupEnv* e, int nPlayers, uint64_t seed, uint32_t genv, int flags);
void rc_reset(RoboCupEnv* e);
void rc_write_full_obs(const RoboCupEnv* e, float* out /* [2n][obs_dim] */);
void rc_write_global_state(const RoboCupEnv* e, float* out /* [2n * 6 + 3] */);
int rc_step(RoboCupEnv* e, const int32_t* actions /* [2n][4] */, float* obs /* [5][2n][obs_dim] or NULL */, double* rewards);

double rc_process_seens(double lSum, const double* rSum, int nOthers, double bSum);
void rc_process_action(RoboCupEnv* e, Robot* r, const int32_t* action);
void rc_tick(RoboCupEnv* e, Robot* r);
int rc_is_ball_out_of_field(RoboCupEnv* e);
void rc_penalize(RoboCupEnv* e, Robot* r);
void rc_fall(RoboCupEnv* e, Robot* r, int punish);
cpv rc_robot_pos(const Robot* r);
void rc_spots(const double* rnd18, cpv spots[2][5]);
void rc_spots_random(const double* rnd20, const int* perm8, cpv spots[2][5]);
void rc_get_state(const RoboCupEnv* e, dynenv_robocup_state_t* st);
void rc_set_state(RoboCupEnv* e, const dynenv_robocup_state_t* st);

#endif
