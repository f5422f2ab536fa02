/* dynenv.h — C ABI of the MI355X-native batched DynEnv step() (libdynenv_hip.so).
 *
 * Drop-in boundary for the reference's hot path.  Each entry point names the reference interface it replaces
 * (paths relative to the reference repo root):
 *
 *   dynenv_create        <- DynEnv/__init__.py:6-25 make_dyn_env(...) building num_envs DrivingEnvironment /
 *                           RoboCupEnvironment objects (DrivingEnvironment.py:20-56, RoboCupEnvironment.py:23-71)
 *   dynenv_reset         <- DynEnv/utils/subproc_vec_env.py:118-122 SubprocVecEnv.reset ->
 *                           DynEnv/environment_base.py:205-224 EnvironmentBase.reset
 *   dynenv_step          <- DynEnv/utils/subproc_vec_env.py:102-111 step_async/step_wait ->
 *                           DynEnv/DrivingEnvironment.py:248-322 / DynEnv/RoboCupEnvironment.py:446-524 step
 *   dynenv_episode_stats <- info['episode_r'|'episode_p_r'|'episode_o_r'|'episode_g'] (DrivingEnvironment.py:310-316,
 *                           RoboCupEnvironment.py:516-521)
 *   dynenv_seed          <- DynEnv/environment_base.py:190-193 set_random_seed
 *   dynenv_get/set_state <- (no reference equivalent; SURVEY §5 "checkpoint/resume" + parity tests)
 *   dynenv_destroy       <- SubprocVecEnv.close (subproc_vec_env.py:124-133)
 *
 * All functions return 0 on success, <0 on error (dynenv_last_error() gives the text).  Pointers named *_dev are
 * DEVICE pointers (HBM) owned by the caller; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * No torch types cross this boundary.  A handle is single-threaded; distinct handles are independent (dynenv_last_error() is
 * per calling thread).
 * There is NO CPU fallback: every entry fails with DYNENV_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef DYNENV_H
#define DYNENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYNENV_ABI_VERSION 3 /* 2: dynenv_debug_counters writes 16 words; checkpoints hold simulation state only.  3: dynenv_error_flags - Partial-observation rows dropped moved from bit 1 to its own bit 3; dynenv_step may be stream-captured */

/* DynEnvType / ObservationType / NoiseType values are the reference's (cutils.py:10-51) */
#define DYNENV_ROBO_CUP 0
#define DYNENV_DRIVE 1
#define DYNENV_OBS_FULL 0
#define DYNENV_OBS_PARTIAL 1
#define DYNENV_NOISE_RANDOM 0
#define DYNENV_NOISE_REALISTIC 1

#define DYNENV_OK 0
#define DYNENV_ERR_ARG (-1)
#define DYNENV_ERR_NO_DEVICE (-2)
#define DYNENV_ERR_HIP (-3)
#define DYNENV_ERR_UNSUPPORTED (-4)
#define DYNENV_ERR_ACTION (-5) /* reference raises on malformed actions, DrivingEnvironment.py:262-263,365-368 */

/* RoboCup class-level switches (RoboCupEnvironment.py:18-21): randomInit (:241, :326 random robot spots, ball position and
 * ownership), deterministicTurn (:317 head starts at team * headMaxAngle, :529 the head action is forced to -3 * team),
 * canFall, useObsRewards; and the constructor's allowHeadTurn (:24, :339, :539).  All five are implemented. */
#define DYNENV_FLAG_RANDOM_INIT 1
#define DYNENV_FLAG_DETERMINISTIC_TURN 2
#define DYNENV_FLAG_CAN_FALL 4
#define DYNENV_FLAG_USE_OBS_REWARDS 8
#define DYNENV_FLAG_ALLOW_HEAD_TURN 16

#define DYNENV_MAX_CARS 10
#define DYNENV_MAX_PEDS 20
#define DYNENV_MAX_OBST 20
#define DYNENV_DRIVE_LANES 8

typedef struct dynenv_cfg {
  int32_t abi_version;    /* DYNENV_ABI_VERSION */
  int32_t env_type;       /* DYNENV_DRIVE | DYNENV_ROBO_CUP */
  int32_t num_envs;       /* environments owned by THIS handle (this GPU's shard) */
  int32_t n_players;      /* nPlayers: cars (Driving, <=10) or robots per team (RoboCup, <=5) */
  int32_t obs_type;       /* DYNENV_OBS_* */
  int32_t noise_type;     /* DYNENV_NOISE_* */
  double noise_magnitude; /* 0..5 */
  uint64_t seed;
  int32_t env_id_offset;  /* global id of local env 0: RNG streams are keyed by global id => shard-count invariant */
  int32_t flags;          /* DYNENV_FLAG_* */
  int32_t device_id;      /* HIP device ordinal */
  int32_t reserved;
} dynenv_cfg_t;

/* dense, padded observation layout of one (env, time step, agent) row; all float32 */
typedef struct dynenv_layout {
  int32_t num_envs, n_agents, n_time_steps, obs_dim;
  int32_t action_dim;       /* 2 Driving, 4 RoboCup */
  int32_t n_blocks;         /* number of row blocks below */
  int32_t block_offset[8];  /* float offset inside the obs row */
  int32_t block_rows[8];    /* row capacity */
  int32_t block_feat[8];    /* features per row */
  int32_t steps_per_episode;
  int32_t reserved;
} dynenv_layout_t;

/* Driving layout block ids: mirrors ((cars, obstacles, pedestrians), (self, lanes)) of DrivingEnvironment.py:121-124 */
#define DYNENV_DRV_BLOCK_SELF 0
#define DYNENV_DRV_BLOCK_CARS 1
#define DYNENV_DRV_BLOCK_OBST 2
#define DYNENV_DRV_BLOCK_PEDS 3
#define DYNENV_DRV_BLOCK_LANES 4

/* ---- canonical per-env state blob (checkpoint / parity tests).  set_state clears the contact cache. ---- */
typedef struct dynenv_car_state {
  double px, py, vx, vy, angle, w;
  double dirx, diry, prevx, prevy, goalx, goaly;
  int32_t type, team, finished, crashed, lane_pos, fric; /* fric: 0 friction_car, 1 friction_car_crashed */
  int32_t pad[2];
} dynenv_car_state_t;

typedef struct dynenv_ped_state {
  double px, py, vx, vy;
  int32_t road, side, dead, moving, speed, crossing, begin_crossing, pad;
} dynenv_ped_state_t;

typedef struct dynenv_driving_state {
  int32_t elapsed, all_finished, n_cars, n_peds, n_obst, episode;
  int32_t pad[2];
  double episode_r[DYNENV_MAX_CARS], episode_pos_r[DYNENV_MAX_CARS];
  dynenv_car_state_t cars[DYNENV_MAX_CARS];
  dynenv_ped_state_t peds[DYNENV_MAX_PEDS];
  double obst_x[DYNENV_MAX_OBST], obst_y[DYNENV_MAX_OBST];
} dynenv_driving_state_t;

/* RoboCup blob.  set_state clears the contact cache, the joints' accumulated impulses and pending fall forces. */
#define DYNENV_MAX_ROBOTS 10
typedef struct dynenv_robot_state {
  double lpx, lpy, lvx, lvy, la, lw; /* left foot body */
  double rpx, rpy, rvx, rvy, ra, rw; /* right foot body */
  double head_angle, head_moving, prevx, prevy, initx, inity, penal_time, fall_time, move_time;
  int32_t team, penalized, touching, touch_cntr, might_push, fallen, fall_cntr, kicking, foot, joint_removed;
  int32_t pad[2];
} dynenv_robot_state_t;

typedef struct dynenv_robocup_state {
  int32_t elapsed, n_robots, ball_owned, n_last_kicked;
  int32_t last_kicked[4], goals[2], closest[2], n_def[2], defenders[2][DYNENV_MAX_ROBOTS];
  int32_t episode, pad;
  double ball_free_cntr, grace_period, penal_times[2];
  double bpx, bpy, bvx, bvy, bw, bprevx, bprevy;
  double episode_r[DYNENV_MAX_ROBOTS], episode_pos_r[DYNENV_MAX_ROBOTS];
  dynenv_robot_state_t robots[DYNENV_MAX_ROBOTS];
} dynenv_robocup_state_t;

typedef struct dynenv dynenv_t;

int dynenv_abi_version(void);
const char* dynenv_last_error(void);

int dynenv_create(const dynenv_cfg_t* cfg, dynenv_t** out);
void dynenv_destroy(dynenv_t* h);
int dynenv_layout(const dynenv_t* h, dynenv_layout_t* out);
int dynenv_seed(dynenv_t* h, uint64_t seed);

/* (Re)build every environment's scene and write the first observation(s): obs_dev float32 [E, T, A, obs_dim]. */
int dynenv_reset(dynenv_t* h, float* obs_dev, void* stream);

/* One environment step for all E environments (10 / 50 physics substeps fused in one launch).
 *   actions_dev  int32 [E, A, action_dim]      obs_dev float32 [E, T, A, obs_dim]
 *   rewards_dev  float64 [E, A]                dones_dev uint8 [E]
 * Does NOT auto-reset: the host mirror calls dynenv_reset after a terminal step (SubprocVecEnv semantics).
 * obs_dev may be NULL (no observation written) except for RoboCup with Partial observations, whose rewards contain the
 * observation reward of processSeens (RoboCupEnvironment.py:497-524) and come out of the same pass: DYNENV_ERR_ARG then.
 * May be captured into a hipGraph (stream capture) and replayed: per-step scheduling counters move to the device on the first
 * captured call, for the rest of the handle's life (INTEGRATION.md "Scheduling, checkpoints and graphs"). */
int dynenv_step(dynenv_t* h, const int32_t* actions_dev, float* obs_dev, double* rewards_dev, uint8_t* dones_dev,
                void* stream);

/* The noise-free Full observation of the CURRENT state, whatever the handle's observation type: full_dev float32
 * [E, A, dynenv_full_obs_dim(h)] in the Full layout (Driving: self 9 | cars (A-1) x 7 | obstacles 20 x 4 | pedestrians 20 x 2 |
 * lanes 8 x 5; RoboCup: ball 4 | self 8 | robots (A-1) x 6).  This is what the reference puts into info['Full State'] /
 * info['Recon States'] after every step in every observation mode (getFullState, DrivingEnvironment.py:306-307,
 * RoboCupEnvironment.py:511-512) and what a caller needs after dynenv_set_state / dynenv_checkpoint_load. */
int dynenv_full_obs_dim(const dynenv_t* h);
int dynenv_full_obs(dynenv_t* h, float* full_dev, void* stream);

/* RoboCup's info['Full State'] = getFullState(agent=None) of the CURRENT state (RoboCupEnvironment.py:511, :1149-1161), which is
 * NOT a per-agent row of dynenv_full_obs: robots float32 [A][6] = (normalize(x, standardNorm, 0), normalize(y, standardNorm, 0),
 * cos a, sin a, team, fallen | penalized) in field coordinates (no team flip; normalize = cutils.py:318-323), then the ball
 * [3] = (normalize(bx), normalize(by), ballOwned): state_dev float32 [E, dynenv_global_state_dim(h)] with dim = 6 A + 3.
 * The reference's trainer reads the robots' last column from it (models/train.py:272).  Driving: dim 0 and
 * DYNENV_ERR_UNSUPPORTED - its getFullState(None) (DrivingEnvironment.py:697-711) is columns {0..5, 8} of every car's own self
 * block in dynenv_full_obs plus the shared obstacle / pedestrian / lane blocks, bit for bit. */
int dynenv_global_state_dim(const dynenv_t* h);
int dynenv_global_state(dynenv_t* h, float* state_dev, void* stream);

/* dynenv_step with the reference's continuous head action: RoboCup built with allowHeadTurn (make_dyn_env's
 * use_continuous_actions, DynEnv/__init__.py:11) takes Tuple((MultiDiscrete([5, 3, 3]), Box(-3, 3, (1,))))
 * (RoboCupEnvironment.py:339-342); head_dev float64 [E, A] carries the Box channel (turnHead(head), Robot.py:136-138),
 * actions_dev[..., 3] is ignored then.  head_dev == NULL: the discrete head action of actions_dev[..., 3] (what dynenv_step
 * does).  DYNENV_ERR_ARG unless the handle is RoboCup with DYNENV_FLAG_ALLOW_HEAD_TURN. */
int dynenv_step_head(dynenv_t* h, const int32_t* actions_dev, const double* head_dev, float* obs_dev, double* rewards_dev,
                     uint8_t* dones_dev, void* stream);

/* Measurement hook (bench.py's roofline leg): three caller-owned hipEvent_t (as void*; NULL = none, all NULL = off) that every
 * following dynenv_step / dynenv_step_head records on ITS launch stream - before the step's dominant kernel (drv_step_kernel,
 * rc_step_kernel, drv_step_partial_kernel, rc_step_partial_kernel), right after it, and after the step's last kernel (the
 * deferred-observation / finalize launches of the Partial paths).  hipEventElapsedTime(begin, main_done) is that kernel's own
 * launch duration.  The events are overwritten by the next step: synchronise on ev_end before stepping again. */
int dynenv_set_step_events(dynenv_t* h, void* ev_begin, void* ev_main_done, void* ev_end);

/* Per-env object counts of the current episode: int32 [E, 2] = (n_obstacles, n_pedestrians) (Driving). */
int dynenv_counts(dynenv_t* h, int32_t* counts_dev, void* stream);

/* Episode accumulators: ep_r, ep_pos_r, ep_obs_r float64 [E, A]; goals int32 [E, 2]. Any pointer may be NULL. */
int dynenv_episode_stats(dynenv_t* h, double* ep_r_dev, double* ep_pos_r_dev, double* ep_obs_r_dev,
                         int32_t* goals_dev, void* stream);

/* Copy one environment's canonical state blob to / from HOST memory (synchronous; test + checkpoint path). */
size_t dynenv_state_size(const dynenv_t* h);
int dynenv_get_state(dynenv_t* h, int32_t env_idx, void* host_blob, size_t nbytes);
int dynenv_set_state(dynenv_t* h, int32_t env_idx, const void* host_blob, size_t nbytes);

int dynenv_sync(dynenv_t* h, void* stream);

/* OR over all environments of the kernels' error flags (bit 0: contact cache overflow, a pair was dropped; bit 1: an action
 * outside the action space was seen - the reference raises there, DrivingEnvironment.py:365-368 / RoboCupEnvironment.py:543-550;
 * here that agent's action is ignored for the step and the flag stays up until the next reset; bit 3: a Partial observation
 * list had more rows than its capacity in the layout and the rows beyond it were DROPPED - the reference's lists have no cap
 * (DrivingEnvironment.py:816-890); SURVEY Appendix E's worst case is above the Driving capacities (24 cars / 32 obstacles / 40
 * pedestrians / 16 lanes), astronomically unlikely, and no longer silent: the host mirror's step() raises on it; bit 2: Driving Partial, the list of environments left to the deferred
 * observation launch is full - it is cleared between steps by launches that alternate a parity; a guard that no supported use
 * reaches: a captured dynenv_step keeps that parity on the device); bit 4 (value 16): RoboCup, the CORES of two
 * capsules (feet, Robot.py:38-52) were exactly collinear or exactly touching.  Cores that CROSS - the two feet of one robot that
 * has been knocked about get there in play - take the normal Chipmunk's EPA gives (the minimum-translation axis of the two cores;
 * DESIGN.md 2b); in this measure-zero case even that normal's SIGN is a convention, and from that substep on the environment may
 * not be what pymunk would have computed.  A dynenv_set_state blob can put two feet on one line.  Sticky until the next reset /
 * set_state; the host mirror's step() raises on it; bit 5 (value 32):
 * RoboCup, a foot velocity or a joint impulse left the finite range during a solve (never seen): the joints' arithmetic drops
 * products that are zeros for finite operands only, so that substep is not the reference's - reported like bit 4.
 * Synchronises the device. */
int dynenv_error_flags(dynenv_t* h, int32_t* out);

/* Diagnostics (Driving), summed over environments since the last reset: out16 = {substeps on the no-contact fast path,
 * on the quiescent shortcut, on the full contact path, sum of live contact-cache slots, contact-path substeps caused by
 * a changed candidate set / a moving body / a non-inert arbiter, substeps served by a steady replay, by the light mode,
 * contact-path substeps whose sweeps ran as split-lane general multi-level solves, the number of environments the next step
 * gives a SIMD of their own (-1: isolation off for this handle), isolation placeholders that gave up waiting (stays 0),
 * the handle's scheduling mode (0 none, 1 SIMD isolation, 2 slow environments first, 3 timing only: Partial observations), whether the block -> SIMD placement
 * isolation relies on validated for the next step (mode 1; it is re-checked on the device every launch and isolation holds
 * off while it does not validate), launches whose placement did not validate (since isolation was last started), the number
 * of times the host paused isolation - plain launches for 2048 steps - because more than half of the launches of a 64-step
 * window did not validate: a device shared with another handle or kernel}.
 * Synchronises the device. */
int dynenv_debug_counters(dynenv_t* h, int64_t* out16);
/* Diagnostics: where the regular blocks of the last Driving step ran - out[b] = XCC id << 16 | HW_ID bits (SE 15:13, SH 12,
 * CU 11:8, SIMD 5:4) of block b; recorded only by handles in scheduling mode 1 (it is what the self-validation of the SIMD
 * isolation reads).  Returns the number of words written (0: not recorded), < 0 on error.  Synchronises the device. */
int dynenv_debug_placement(dynenv_t* h, uint32_t* out, int32_t n);

/* ---- de-duplicated transport format for the multi-GPU all-gather.  Where every agent row of an (env, time) ends in the
 * same tail (Driving Full: obstacles, pedestrians, lane rows = 160 of 232 floats), the packed form holds the A prefixes
 * of `split` floats followed by the tail once: A*split + (D - split) floats instead of A*D.  n_env_time = E*T rows of
 * [A][D].  unpack(pack(x)) == x bit for bit when the tails are identical (they are written from one LDS copy). ---- */
int dynenv_obs_pack(const float* obs_dev, int64_t n_env_time, int32_t A, int32_t D, int32_t split, float* packed_dev, void* stream);
int dynenv_obs_unpack(const float* packed_dev, int64_t n_env_time, int32_t A, int32_t D, int32_t split, float* obs_dev, void* stream);
/* the gathered buffer of an all-gather: n_ranks packed blocks `src_stride_floats` apart -> obs_dev [n_ranks][n_env_time][A][D], one launch */
int dynenv_obs_unpack_ranks(const float* packed_dev, int64_t src_stride_floats, int32_t n_ranks, int64_t n_env_time, int32_t A,
                            int32_t D, int32_t split, float* obs_dev, void* stream);
/* Peer-compacted transport format of a Driving Full observation (reference: DrivingEnvironment.getFullState,
 * DrivingEnvironment.py:686-747).  Agent a's row is [self 9 | the other A-1 cars x 7 | tail]; the 7 floats of car c are
 * columns {0..5, 8} of c's own self block, so per (env, time) the A self blocks plus the tail once carry everything:
 * 9A + (D - 9 - 7(A-1)) floats instead of A*D.  unpack_peers_ranks expands n_ranks gathered blocks (src_stride_floats
 * apart) into the dense [n_ranks][n_env_time][A][D] tensor, bit for bit. */
int dynenv_obs_pack_peers(const float* obs_dev, int64_t n_env_time, int32_t A, int32_t D, float* packed_dev, void* stream);
int dynenv_obs_unpack_peers_ranks(const float* packed_dev, int64_t src_stride_floats, int32_t n_ranks, int64_t n_env_time,
                                  int32_t A, int32_t D, float* obs_dev, void* stream);

/* ---- exact checkpoint (the reference has none).  Unlike the canonical per-env blob of
 * dynenv_get_state (which drops the contact cache), a checkpoint is every device array of the handle bit for bit -
 * bodies, contact cache, shortcut state, episode counters, seed - so that load + the same actions reproduces the run
 * exactly from the middle of an episode.  Host buffers; the handle must have the configuration the checkpoint was taken
 * with (env type, sizes, observation/noise type, flags, env_id_offset). ---- */
size_t dynenv_checkpoint_size(const dynenv_t* h);
int dynenv_checkpoint_save(dynenv_t* h, void* buf_host, size_t nbytes);
int dynenv_checkpoint_load(dynenv_t* h, const void* buf_host, size_t nbytes);

/* ---- ragged -> padded arranger.  Replaces InOutArranger.rearrange_inputs / rearrange_outputs of the
 * reference's input layer (DynEnv/models/models.py:219-250, :252-274) on the dense observation tensor this library
 * writes.  Pure functions of device buffers on the current HIP device (no handle).  A "group" is the list of object
 * types the reference passes as one element of obs[..., 0] / obs[..., 1], e.g. Driving movable = (cars, obstacles,
 * pedestrians).  P = E*A players, player index p = e*A + a (models.py:222-223).                                   ---- */
#define DYNENV_ARR_MAX_TYPES 4
#define DYNENV_ARR_COUNT_CONST 0 /* every (env, time, agent) row holds `count_value` objects of the type */
#define DYNENV_ARR_COUNT_ENV 1   /* count = count_env[env * count_stride + count_index]   (dynenv_counts() layout) */
#define DYNENV_ARR_COUNT_ROW 2   /* count = (int) obs_row[count_index]   (Driving Partial keeps list lengths in the row tail) */
typedef struct dynenv_arr_type {
  int32_t offset, feat, cap; /* block inside an observation row: float offset, features per object, row capacity */
  int32_t count_mode, count_value, count_index, count_stride, reserved;
} dynenv_arr_type_t;
typedef struct dynenv_arr_plan {
  int32_t n_types, n_time, n_players, max_count; /* max_count = max over (time, player) of the summed counts (:233) */
  int64_t total[DYNENV_ARR_MAX_TYPES];           /* N_i = number of objects of type i over all (time, player) */
} dynenv_arr_plan_t;

/* ints of scratch dynenv_arrange_plan needs for E*T*A rows and n_types types */
int64_t dynenv_arrange_scratch_ints(int32_t E, int32_t T, int32_t A, int32_t n_types);
/* Phase 1 (rearrange_inputs :226-236): counts [n_types][T][P], obj_counts [T][P], base [n_types][T][P] = exclusive prefix
 * sums in (time, player) order (where the objects of (t, p) start inside inputs[i]).  Synchronises the stream and fills
 * *plan_host (the reference's `.item()` on maxCount is the same host round trip). */
int dynenv_arrange_plan(const float* obs_dev, int32_t E, int32_t T, int32_t A, int32_t D, const dynenv_arr_type_t* types,
                        int32_t n_types, const int32_t* count_env_dev, int32_t* counts_dev, int32_t* obj_counts_dev,
                        int32_t* base_dev, int32_t* scratch_dev, dynenv_arr_plan_t* plan_host, void* stream);
/* Phase 2 (rearrange_inputs :238-245 + the index arithmetic of rearrange_outputs :259-268): inputs_dev[i] float32
 * [N_i][feat_i] in (time, player, object) order; slot_dev[i] int32 [N_i] = destination row of each object in the padded
 * tensor [T][max_count][P][F] viewed as [T*max_count*P][F]; mask_dev uint8 [T][P][max_count], 1 = padding (createMask).
 * Any of the pointers may be NULL to skip that output. */
int dynenv_arrange_gather(const float* obs_dev, int32_t E, int32_t T, int32_t A, int32_t D, const dynenv_arr_type_t* types,
                          int32_t n_types, const int32_t* counts_dev, const int32_t* base_dev, int32_t max_count,
                          float* const* inputs_dev, int32_t* const* slot_dev, uint8_t* mask_dev, void* stream);
/* Phase 3 (rearrange_outputs :257-268, catAndPad): padded[slot[n]][:] = emb[n][:] for the N embeddings of one type;
 * padded_dev float32 [T][max_count][P][F] must have been zeroed by the caller (the padding). */
int dynenv_arrange_scatter(const float* emb_dev, const int32_t* slot_dev, int64_t N, int32_t F, float* padded_dev, void* stream);
/* Phase 3, fused (preferred): writes EVERY element of padded_dev float32 [T][max_count][P][F] in one pass - the
 * embedding of the object that owns the slot, or zero - so no pre-zeroing and no second pass per type.  emb_dev[i] is
 * float32 [N_i][F] (NULL for a type without objects); F must be a multiple of 4 and the buffers 16-byte aligned. */
int dynenv_arrange_pad(const float* const* emb_dev, const int32_t* counts_dev, const int32_t* base_dev, int32_t n_types,
                       int32_t T, int32_t P, int32_t max_count, int32_t F, float* padded_dev, void* stream);


/* Device self-test of the deterministic math header: evaluates sincos/atan2/sqrt/div on n host-provided doubles and
 * returns the raw results so tests can compare them bit-for-bit with the host evaluation. out: [n,5]. */
int dynenv_math_selftest(const double* x_host, const double* y_host, int32_t n, double* out_host, int32_t device_id);

#ifdef __cplusplus
}
#endif
#endif /* DYNENV_H */
