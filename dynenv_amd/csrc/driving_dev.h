// driving_dev.h — device-side data layout of the batched Driving environment (HBM SoA + LDS tile).
//
// Mapping: ONE WAVEFRONT (64 lanes) PER ENVIRONMENT.  Lane l doubles as
//   * object l of the scene: cars 0..9, pedestrians 10..29 (dynamic bodies, state in the LDS tile), obstacles 30..49,
//     buildings 50..53 - for integration and for the broadphase (bit i of lane j's mask = canonical pair (car i, j))
//   * one quarter of a candidate pair in the narrowphase (four lanes per pair)
//   * the owner of contact-cache slot l    (l < DRV_NS persistent arbiters)
// HBM layout is field-major with the 32 body slots of one env contiguous (`field[e][32]`), so a wave's load of one
// field is a single 256-byte coalesced segment.
#pragma once
#include "dev_common.h"

#define DRV_MAXA 10
#define DRV_MAXP 20
#define DRV_MAXO 20
#define DRV_NB 32          /* dynamic body slots per env (30 used) */
#define DRV_NS 24          /* persistent arbiter (contact cache) slots per env (LDS budget: 16 envs per CU) */
#define DRV_SLOT_PED 10
#define DRV_SLOT_OBST 30
#define DRV_SLOT_BLD 50
#define DRV_W 1700.0
#define DRV_H 1000.0
#define DRV_MAX_TIME 6000
#define DRV_TIME_DIFF 10
#define DRV_LANE_ROWS 8

// body f64 fields
enum { BF_PX = 0, BF_PY, BF_VX, BF_VY, BF_ANG, BF_W, BF_VBX, BF_VBY, BF_WB, BF_COUNT };
// car f64 fields
enum { CF_DIRX = 0, CF_DIRY, CF_PREVX, CF_PREVY, CF_GOALX, CF_GOALY, CF_COUNT };
// env int fields
enum { EI_ELAPSED = 0, EI_ALLFIN, EI_NPED, EI_NOBST, EI_EPISODE, EI_OCC, EI_ERR, EI_PAD,
       EI_N_FAST, EI_N_QUIET, EI_N_CONTACT, EI_N_SLOTS, /* diagnostics: substeps per path, sum of live slots */
       EI_N_WHY_CAND, EI_N_WHY_MOVING, EI_N_WHY_INERT, EI_N_STEADY, EI_N_LIGHT, /* why the quiescent shortcut was not taken */
       EI_N_SPLIT, /* contact-path substeps whose sweeps ran in drv_solve_general_split (general multi-level solves) */
       EI_DEFER_OBS, /* Partial: first agent whose observation is left to drv_partial_obs_deferred_kernel (A = none) */
       EI_COUNT };
// arbiter states (Chipmunk cpArbiterState)
enum { ARB_FIRST = 0, ARB_NORMAL = 1, ARB_IGNORE = 2, ARB_CACHED = 3 };
// LanePosition (cutils.py:143-148)
enum { LP_AtGoal = 0, LP_InRightLane = 1, LP_InOpposingLane = 2, LP_OverRoad = 3, LP_OffRoad = 4 };

// car flag word:  type[0:2) team[2:4) finished[4] crashed[5] fric[6] lanepos[8:11)
// ped flag word:  road[0] side[1] dead[2] crossing[3] begin[4] speed[8:12)
#define CARF_TYPE(f) ((f)&3)
#define CARF_TEAM(f) (((f) >> 2) & 3)
#define CARF_PACK(type, team, fin, crashed, fric, lp) \
  ((type) | ((team) << 2) | ((fin) << 4) | ((crashed) << 5) | ((fric) << 6) | ((lp) << 8))
#define PEDF_PACK(road, side, dead, crossing, begin, speed) \
  ((road) | ((side) << 1) | ((dead) << 2) | ((crossing) << 3) | ((begin) << 4) | ((speed) << 8))

struct DrvRoad {
  int nLanes;
  double width, length, dirAngle, followDist;
  double cosDir0; /* dm_cos(dirAngle - 0.0): the pedestrian / obstacle road test always passes angle 0 */
  V2 p0, p1, dir, normal;
  V2 walk[2][2];
};

struct DrvConst {
  DrvRoad roads[2];
  float laneRows[DRV_LANE_ROWS * 5];
  double carMass[4], carInertia[4], carHx[4], carHy[4], carPower[4];
  double pedMass, pedInertia;
  double turnCos[2], turnSin[2]; /* cos/sin(-/+ 2*pi/180) */
};

// per-lane inputs of the Partial observation of one environment (lane = body slot / obstacle / car index)
struct PvIn {
  double px, py, ang; /* body slot `lane` (< 32), 0 for unused slots */
  double ox, oy;      /* obstacle `lane` (< nObst) */
  double gx, gy;      /* goal of car `lane` (< A) */
  int flags;          /* flag word of body slot `lane` */
};

struct DrvState {
  int E, A, obs_dim, pad0;
  uint64_t seed;
  int env_id_offset, pad1;
  double* body;   /* [BF_COUNT][E][32] */
  double* carx;   /* [CF_COUNT][E][16] */
  int* flags;     /* [E][32] */
  int* aux;       /* [E][32]  ped `moving` (ms) */
  double* obst;   /* [2][E][20] */
  int* envi;      /* [E][EI_COUNT] */
  double* epr;    /* [2][E][16] episode_r, episode_pos_r */
  int* s_pair;    /* [E][NS]  pair id (i<<8|j) */
  int* s_meta;    /* [E][NS]  state | count<<8 | age<<16 */
  uint32_t* s_hash; /* [2][E][NS] */
  double* s_imp;  /* [4][E][NS] jn0 jt0 jn1 jt1 */
  int* lastcand;  /* [E][64] per-lane candidate mask of the last substep (-1 = unknown); envi[EI_PAD] = inert | steady<<1 | vbValid<<2 */
  /* SIMD isolation of the slow environments (scheduling only - which block steps which environment; see drv_iso_assign):
     iso[0..2] = list lengths, iso[3..5] = slowest environment's cycles, iso[7] = placeholders that gave up waiting,
     iso[8..10] = "the block -> SIMD placement isolation relies on was observed" (written by the launch before the one that
     reads it), iso[11] = launches in which it was not, iso[12] = cool-down counter of the validation, iso[13] / iso[14] = tick / pv_par on the device (tick_src), iso[DRV_ISO_HDR + buf * DRV_ISO_LIST + k] = ids; three buffers used in
     rotation (step t reads buffer t % 3, fills (t + 1) % 3, clears (t + 2) % 3); iso_done[e] = tick of e's last finished step;
     iso_hw[t & 1][b] = hardware id (XCC | SE | SH | CU | SIMD) block b of step t ran on.  None of this is simulation state:
     it is allocated outside the checkpointed arrays and starts over at dynenv_checkpoint_load. */
  int* iso;
  int* iso_done;
  unsigned* iso_hw;
  /* Partial observation: the environments whose agent passes the step launch left to the deferred one (scheduling scratch as
     well): pvq[16 p] = length, pvq[32 + p E + k] = ids of the list of parity p = pv_par (the host flips it every step; the deferred
     launch of a step clears the other parity's length for the next one). */
  int* pvq;
  int pv_par;
  int tick, iso_on;  /* iso_on: 0 off, 1 isolation (E = one residency round), 2 slow environments first (E larger) */
  int tick_src;      /* 0: `tick` / `pv_par` above, advanced by the host per launch; 1: the device words iso[13] / iso[14], advanced by
                        drv_tick_advance_kernel in front of every step - set for good once a step has been captured into a hipGraph */
};
#ifndef DRV_ISO_MAX
#define DRV_ISO_MAX 192  /* at most this many environments get a SIMD of their own (iso_on = 1).  The list fills in FINISHING order: a cap
                            below the number of listed environments drops the slowest of them (round 6: 64 -> 192 with the threshold at
                            75 % instead of 80 %: -0.6 %; 40: +2.2 %; 256 at 65 %: +2 %) */
#endif
#ifndef DRV_ISO_MIN_E
#define DRV_ISO_MIN_E 3072 /* isolation (mode 1) for DRV_ISO_MIN_E < E <= 4096: measured -6.5 % at 4095, -6 % at 4000, -4 % at 3584, -0.8 % at 3072, +0.8 % at 2048 (HISTORY "Round 5") */
#endif
#define DRV_ISO_LIST 256 /* capacity of a list; iso_on = 2 (more environments than fit at once) starts up to this many slow ones first */
#define DRV_ISO_HDR 16    /* header words of DrvState.iso in front of the three lists */
#define DRV_ISO_WORDS (DRV_ISO_HDR + 3 * DRV_ISO_LIST)
#define DRV_ISO_COOLDOWN 64 /* validated launches isolation stays off for after one that did not validate */
#define DRV_ISO_G0 256     /* first SIMD group isolation uses: groups 256..511 are the ones whose four blocks always share a SIMD */
#define DRV_ISO_GSPAN 256  /* ... and how many of them the device-side validation checks (DRV_ISO_MAX <= GSPAN) */
#define DRV_ISO_GROUPS 1024 /* SIMDs of an MI355X: blocks b, b + 1024, b + 2048, b + 3072 of a launch share one (measured) */
