// robocup_dev.h — device-side layout of the batched RoboCup environment (HBM SoA + LDS tile).
//
// One wavefront per environment, like Driving.  Lane l doubles as
//   * owner of dynamic body l: feet 2*id (left) / 2*id+1 (right) for robot id 0..9, ball = 20
//   * owner of robot l (l < 10) for the joint solver: PivotJoint + RotaryLimitJoint of Robot.py:58-60
//   * detector of collision pairs {l, 64+l, ...} of the 294 canonical pairs (5 rounds)
//   * owner of contact-cache slot l (l < RC_NS)
// The order-dependent game logic (processAction / tick / isBallOutOfField and the collision callbacks with their
// fall()/penalize() side effects) runs as a scalar program on lane 0 — it is strictly sequential in the reference.
#pragma once
#include "dev_common.h"

#define RC_MAXR 10
#define RC_NB 32
#define RC_NS 16
#define RC_NPAIR_ROUNDS 5 /* ceil(294 / 64) */
#define RC_BALL 20
#define RC_POST 21
#define RC_W 1040.0
#define RC_H 740.0
#define RC_SIDE 70.0
#define RC_MAX_TIME 12000

// body f64 fields in HBM
enum { RB_PX = 0, RB_PY, RB_VX, RB_VY, RB_ANG, RB_W, RB_VBX, RB_VBY, RB_WB, RB_FX, RB_FY, RB_TQ, RB_COUNT };
// robot f64 fields
enum { RR_HEAD = 0, RR_HEADMOV, RR_PREVX, RR_PREVY, RR_INITX, RR_INITY, RR_PENALT, RR_FALLT, RR_MOVET, RR_JX, RR_JY, RR_JROT,
       RR_COUNT };
// robot int fields
enum { RI_FLAGS = 0, RI_TOUCHC, RI_FALLC, RI_COUNT };
// env ints
enum { RE_ELAPSED = 0, RE_OWNED, RE_GOAL0, RE_GOAL1, RE_CLOSE0, RE_CLOSE1, RE_DEF0, RE_DEF1, RE_NLK, RE_LK0, RE_LK1, RE_LK2,
       RE_LK3, RE_NCON, RE_EPISODE, RE_OCC, RE_ERR, RE_CORDER /* 20 entries */, RE_PIVFIRST = RE_CORDER + 20 /* bit r: robot r's pivot joint precedes its rotary limit in the constraint list */,
       RE_COUNT = RE_CORDER + 20 + 3 };
// env doubles
enum { RD_FREECNT = 0, RD_GRACE, RD_PT0, RD_PT1, RD_BPREVX, RD_BPREVY, RD_COUNT = 8 };

enum { ARB_FIRST_ = 0, ARB_NORMAL_ = 1, ARB_IGNORE_ = 2, ARB_CACHED_ = 3 };  // cpArbiterState

// robot flag word: team(+1 -> bit0 = 1, -1 -> 0) penalized touching mightPush fallen kicking foot jointRemoved
#define RF_TEAMPOS 1
#define RF_PENAL 2
#define RF_TOUCH 4
#define RF_PUSH 8
#define RF_FALLEN 16
#define RF_KICK 32
#define RF_FOOT 64
#define RF_JREM 128

struct RcConst {
  double footInertia, ballInertia;
  /* per-robot joint constants (both feet have the same mass and inertia; anchors at the body origins), computed on the
     host with the expression sequence of cpPivotJoint/cpRotaryLimitJoint preStep: identical bits, no per-substep divisions */
  double ballIinv, footMinv, footIinv, jkk0, jkk1, jkk2, jkk3, jiSum;
  uint16_t pairs[RC_NPAIR_ROUNDS * 64];
  /* Partial observation (robocup_partial.hip): the scene by vision lane - 10..13 goalposts, 14..16 penalty crosses,
     17..32 line crosses, 33..43 lines (P = first end point, Q = second); T0/T1 = the two tags of each entry
     (RoboCupEnvironment._create_football_field / _create_goalposts) */
  double visPx[48], visPy[48], visQx[48], visQy[48], visT0[48], visT1[48];
};

struct RvSnap;
struct RcState {
  int E, n, R, obs_dim; /* n players per team, R = 2n robots */
  uint64_t seed;
  int env_id_offset, flags;
  double* body;  /* [RB_COUNT][E][32] */
  double* rob;   /* [RR_COUNT][E][16] */
  int* robi;     /* [RI_COUNT][E][16] */
  int* envi;     /* [E][RE_COUNT] */
  double* envd;  /* [E][RD_COUNT] */
  double* epr;   /* [2][E][16] */
  double* epo;   /* [E][16] episode sum of the observation rewards (processSeens) */
  struct RvSnap* snap; /* [E][5] what getAgentVision reads, exported at the step's five snapshots (Partial observation) */
  double* prew0; /* [E][16] positive part of the step's robot + team reward, before the observation reward (Partial) */
  int* seenPart; /* [E][5][120] per-snapshot seen counts of the environments whose vision runs in the deferred launch */
  int* deferList; /* [E + 1 + 8]: [0] = number of environments deferred in this step (zeroed before every step launch), then
                     id | first deferred pass << 20 (pass = snapshot * R + agent; the passes before it ran in the step launch);
                     [E + 1] = the forecast of the step's slowest environment, in cycles = the previous step's maximum of the
                     environments' own times (0: none above RC_SCHED_MIN), [E + 2] = this step's maximum so far (every environment
                     runs its vision passes until the forecast end, rc_step_body).  Per-step scratch, allocated outside the
                     checkpointed arrays like seenPart. */
  int obs_type, noise_type;
  double noise_magn;
  int* s_pair;   /* [E][NS] */
  int* s_meta;
  uint32_t* s_hash; /* [2][E][NS] */
  double* s_imp;    /* [4][E][NS] */
  const uint64_t* pairTab; /* [64][2] per lane: {pair codes of rounds 0..3 (16 bits each), pair code of round 4 | feet-pair bits << 32},
                              filtered for this handle's R (a pair with an absent foot reads 0xFFFF); read-only after create */
};
