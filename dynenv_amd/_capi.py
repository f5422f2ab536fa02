"""ctypes binding of include/dynenv.h (libdynenv_hip.so).  No fallback: a missing library or GPU raises."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYNENV_HIP_LIB", os.path.join(PKG, "libdynenv_hip.so"))  # override: A/B builds only

DYNENV_ABI_VERSION = 3
ERR_NO_DEVICE = -2


class DynEnvError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("env_type", C.c_int32), ("num_envs", C.c_int32),
                ("n_players", C.c_int32), ("obs_type", C.c_int32), ("noise_type", C.c_int32),
                ("noise_magnitude", C.c_double), ("seed", C.c_uint64), ("env_id_offset", C.c_int32),
                ("flags", C.c_int32), ("device_id", C.c_int32), ("reserved", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("n_agents", C.c_int32), ("n_time_steps", C.c_int32),
                ("obs_dim", C.c_int32), ("action_dim", C.c_int32), ("n_blocks", C.c_int32),
                ("block_offset", C.c_int32 * 8), ("block_rows", C.c_int32 * 8), ("block_feat", C.c_int32 * 8),
                ("steps_per_episode", C.c_int32), ("reserved", C.c_int32)]


CAR_F = ("px", "py", "vx", "vy", "angle", "w", "dirx", "diry", "prevx", "prevy", "goalx", "goaly")
CAR_I = ("type", "team", "finished", "crashed", "lane_pos", "fric")
PED_F = ("px", "py", "vx", "vy")
PED_I = ("road", "side", "dead", "moving", "speed", "crossing", "begin_crossing")


class CarState(C.Structure):
    _fields_ = [(n, C.c_double) for n in CAR_F] + [(n, C.c_int32) for n in CAR_I] + [("pad", C.c_int32 * 2)]


class PedState(C.Structure):
    _fields_ = [(n, C.c_double) for n in PED_F] + [(n, C.c_int32) for n in PED_I] + [("pad", C.c_int32)]


class DrivingState(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("elapsed", "all_finished", "n_cars", "n_peds", "n_obst", "episode")] + \
               [("pad", C.c_int32 * 2), ("episode_r", C.c_double * 10), ("episode_pos_r", C.c_double * 10),
                ("cars", CarState * 10), ("peds", PedState * 20), ("obst_x", C.c_double * 20),
                ("obst_y", C.c_double * 20)]


ROBOT_F = ("lpx", "lpy", "lvx", "lvy", "la", "lw", "rpx", "rpy", "rvx", "rvy", "ra", "rw", "head_angle", "head_moving",
           "prevx", "prevy", "initx", "inity", "penal_time", "fall_time", "move_time")
ROBOT_I = ("team", "penalized", "touching", "touch_cntr", "might_push", "fallen", "fall_cntr", "kicking", "foot",
           "joint_removed")


class RobotState(C.Structure):
    _fields_ = [(n, C.c_double) for n in ROBOT_F] + [(n, C.c_int32) for n in ROBOT_I] + [("pad", C.c_int32 * 2)]


class RoboCupState(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("elapsed", "n_robots", "ball_owned", "n_last_kicked")] + \
               [("last_kicked", C.c_int32 * 4), ("goals", C.c_int32 * 2), ("closest", C.c_int32 * 2),
                ("n_def", C.c_int32 * 2), ("defenders", (C.c_int32 * 10) * 2), ("episode", C.c_int32),
                ("pad", C.c_int32), ("ball_free_cntr", C.c_double), ("grace_period", C.c_double),
                ("penal_times", C.c_double * 2)] + \
               [(n, C.c_double) for n in ("bpx", "bpy", "bvx", "bvy", "bw", "bprevx", "bprevy")] + \
               [("episode_r", C.c_double * 10), ("episode_pos_r", C.c_double * 10), ("robots", RobotState * 10)]


FLAG_RANDOM_INIT, FLAG_DETERMINISTIC_TURN, FLAG_CAN_FALL, FLAG_USE_OBS_REWARDS, FLAG_ALLOW_HEAD_TURN = 1, 2, 4, 8, 16

EXPORTS = ["dynenv_abi_version", "dynenv_last_error", "dynenv_create", "dynenv_destroy", "dynenv_layout",
           "dynenv_seed", "dynenv_reset", "dynenv_step", "dynenv_counts", "dynenv_episode_stats",
           "dynenv_state_size", "dynenv_get_state", "dynenv_set_state", "dynenv_sync", "dynenv_math_selftest",
           "dynenv_error_flags", "dynenv_debug_counters", "dynenv_debug_placement", "dynenv_arrange_scratch_ints", "dynenv_arrange_plan",
           "dynenv_arrange_gather", "dynenv_arrange_scatter", "dynenv_arrange_pad", "dynenv_checkpoint_size",
           "dynenv_checkpoint_save", "dynenv_checkpoint_load", "dynenv_obs_pack", "dynenv_obs_unpack", "dynenv_obs_unpack_ranks",
           "dynenv_obs_pack_peers", "dynenv_obs_unpack_peers_ranks", "dynenv_step_head", "dynenv_full_obs", "dynenv_full_obs_dim",
           "dynenv_global_state", "dynenv_global_state_dim", "dynenv_set_step_events"]

ARR_MAX_TYPES = 4
ARR_COUNT_CONST, ARR_COUNT_ENV, ARR_COUNT_ROW = 0, 1, 2


class ArrType(C.Structure):  # dynenv_arr_type_t
    _fields_ = [("offset", C.c_int32), ("feat", C.c_int32), ("cap", C.c_int32), ("count_mode", C.c_int32),
                ("count_value", C.c_int32), ("count_index", C.c_int32), ("count_stride", C.c_int32),
                ("reserved", C.c_int32)]


class ArrPlan(C.Structure):  # dynenv_arr_plan_t
    _fields_ = [("n_types", C.c_int32), ("n_time", C.c_int32), ("n_players", C.c_int32), ("max_count", C.c_int32),
                ("total", C.c_int64 * ARR_MAX_TYPES)]


_lib = None


def load():
    """Load the HIP library; raises if it has not been built (there is no Python/CPU substitute)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DynEnvError("libdynenv_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(dynenv_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise DynEnvError("libdynenv_hip.so does not export " + name)
    lib.dynenv_last_error.restype = C.c_char_p
    lib.dynenv_state_size.restype = C.c_size_t
    vp = C.c_void_p
    lib.dynenv_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
    lib.dynenv_destroy.argtypes = [vp]
    lib.dynenv_destroy.restype = None
    lib.dynenv_layout.argtypes = [vp, C.POINTER(Layout)]
    lib.dynenv_seed.argtypes = [vp, C.c_uint64]
    lib.dynenv_reset.argtypes = [vp, vp, vp]
    lib.dynenv_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dynenv_step_head.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.dynenv_full_obs.argtypes = [vp, vp, vp]
    lib.dynenv_full_obs_dim.argtypes = [vp]
    lib.dynenv_global_state.argtypes = [vp, vp, vp]
    lib.dynenv_global_state_dim.argtypes = [vp]
    lib.dynenv_set_step_events.argtypes = [vp, vp, vp, vp]
    i32 = C.c_int32
    lib.dynenv_arrange_scratch_ints.argtypes = [i32, i32, i32, i32]
    lib.dynenv_arrange_scratch_ints.restype = C.c_int64
    lib.dynenv_arrange_plan.argtypes = [vp, i32, i32, i32, i32, C.POINTER(ArrType), i32, vp, vp, vp, vp, vp,
                                        C.POINTER(ArrPlan), vp]
    lib.dynenv_arrange_gather.argtypes = [vp, i32, i32, i32, i32, C.POINTER(ArrType), i32, vp, vp, i32,
                                          C.POINTER(vp), C.POINTER(vp), vp, vp]
    lib.dynenv_arrange_scatter.argtypes = [vp, vp, C.c_int64, i32, vp, vp]
    lib.dynenv_obs_pack.argtypes = [vp, C.c_int64, i32, i32, i32, vp, vp]
    lib.dynenv_obs_unpack.argtypes = [vp, C.c_int64, i32, i32, i32, vp, vp]
    lib.dynenv_obs_unpack_ranks.argtypes = [vp, C.c_int64, i32, C.c_int64, i32, i32, i32, vp, vp]
    lib.dynenv_obs_pack_peers.argtypes = [vp, C.c_int64, i32, i32, vp, vp]
    lib.dynenv_obs_unpack_peers_ranks.argtypes = [vp, C.c_int64, i32, C.c_int64, i32, i32, vp, vp]
    lib.dynenv_checkpoint_size.argtypes = [vp]
    lib.dynenv_checkpoint_size.restype = C.c_size_t
    lib.dynenv_checkpoint_save.argtypes = [vp, vp, C.c_size_t]
    lib.dynenv_checkpoint_load.argtypes = [vp, vp, C.c_size_t]
    lib.dynenv_arrange_pad.argtypes = [C.POINTER(vp), vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.dynenv_counts.argtypes = [vp, vp, vp]
    lib.dynenv_episode_stats.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dynenv_state_size.argtypes = [vp]
    lib.dynenv_get_state.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.dynenv_set_state.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    lib.dynenv_sync.argtypes = [vp, vp]
    lib.dynenv_math_selftest.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
    lib.dynenv_error_flags.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.dynenv_debug_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.dynenv_debug_placement.argtypes = [vp, vp, C.c_int32]
    if lib.dynenv_abi_version() != DYNENV_ABI_VERSION:
        raise DynEnvError("ABI version mismatch between dynenv_amd and libdynenv_hip.so")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dynenv_last_error().decode("utf-8", "replace")
        raise DynEnvError("%s failed (%d): %s" % (what, rc, msg))
