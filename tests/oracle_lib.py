"""ctypes wrapper around oracle/liboracle.so — the CPU restatement used as the parity checker.

Test infrastructure only: nothing under dynenv_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("ORACLE_LIB", os.path.join(ORACLE_DIR, "liboracle.so"))  # override: the sanitizer build (make -C oracle asan)


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
ROBOCUP_DEFAULT_FLAGS = FLAG_CAN_FALL | FLAG_USE_OBS_REWARDS  # class switches, RoboCupEnvironment.py:18-21


def build():
    if "ORACLE_LIB" not in os.environ:
        subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_state_size.restype = C.c_size_t
        _lib.oracle_drv_tick.restype = C.c_double
        _lib.oracle_drv_pos_reward.restype = C.c_double
        nargs = {"oracle_moment_for_box": 3, "oracle_moment_for_circle": 3, "oracle_moment_for_segment": 6}
        for f, n in nargs.items():
            getattr(_lib, f).restype = C.c_double
            getattr(_lib, f).argtypes = [C.c_double] * n
        _lib.oracle_road_is_point_on_road.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.oracle_road_get_walk_spot.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        _lib.oracle_apply_friction.argtypes = [C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """Batched view over the scalar oracle with the same call shape as dynenv_amd's C ABI."""

    def __init__(self, env_type=1, num_envs=1, n_players=10, obs_type=0, noise_type=0, noise_magnitude=0.0,
                 seed=42, env_id_offset=0, flags=0, threads=1):
        self.l = lib()
        self.cfg = Cfg(1, env_type, num_envs, n_players, obs_type, noise_type, noise_magnitude, seed,
                       env_id_offset, flags, 0, 0)
        self.h = C.c_void_p()
        rc = self.l.oracle_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("oracle_create failed: %d" % rc)
        self.layout = Layout()
        self.l.oracle_layout(self.h, C.byref(self.layout))
        L = self.layout
        self.E, self.A, self.T, self.D, self.K = L.num_envs, L.n_agents, L.n_time_steps, L.obs_dim, L.action_dim
        self.l.oracle_set_threads(self.h, threads)
        self.obs = np.zeros((self.E, self.T, self.A, self.D), np.float32)
        self.rewards = np.zeros((self.E, self.A), np.float64)
        self.dones = np.zeros((self.E,), np.uint8)

    def close(self):
        if self.h:
            self.l.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.l.oracle_reset(self.h, _p(self.obs))
        return self.obs

    def step(self, actions, head=None):
        """head: the continuous head channel [E, A] float64 of RoboCup with allowHeadTurn (Box(-3, 3)), else None"""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.E, self.A, self.K), a.shape
        if head is None:
            self.l.oracle_step(self.h, _p(a), _p(self.obs), _p(self.rewards), _p(self.dones))
        else:
            hd = np.ascontiguousarray(head, dtype=np.float64)
            assert hd.shape == (self.E, self.A), hd.shape
            self.l.oracle_step_head(self.h, _p(a), _p(hd), _p(self.obs), _p(self.rewards), _p(self.dones))
        return self.obs, self.rewards, self.dones

    def step_noobs(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        self.l.oracle_step(self.h, _p(a), None, _p(self.rewards), _p(self.dones))
        return self.rewards, self.dones

    def counts(self):
        c = np.zeros((self.E, 2), np.int32)
        self.l.oracle_counts(self.h, _p(c))
        return c

    def episode_stats(self):
        r = np.zeros((self.E, self.A)); p = np.zeros((self.E, self.A)); o = np.zeros((self.E, self.A))
        g = np.zeros((self.E, 2), np.int32)
        self.l.oracle_episode_stats(self.h, _p(r), _p(p), _p(o), _p(g))
        return r, p, o, g

    def get_state(self, env=0):
        st = DrivingState() if self.cfg.env_type == 1 else RoboCupState()
        rc = self.l.oracle_get_state(self.h, env, C.byref(st), C.sizeof(st))
        assert rc == 0
        return st

    def set_state(self, env, st):
        rc = self.l.oracle_set_state(self.h, env, C.byref(st), C.sizeof(st))
        assert rc == 0

    def global_state(self):
        """RoboCup getFullState(agent=None) of every environment: float32 [E, 6 A + 3]"""
        out = np.zeros((self.E, 6 * self.A + 3), np.float32)
        for e in range(self.E):
            self.l.oracle_rc_global_state(self.h, e, _p(out[e]))
        return out

    def overflow(self):
        return self.l.oracle_overflow(self.h)

    def degenerate(self):
        """16 (error bit 4 of dynenv_error_flags) if a capsule pair's narrowphase took the fallback normal since the last reset / set_state"""
        return self.l.oracle_degenerate(self.h)

    def degenerate_env(self, env):
        return self.l.oracle_degenerate_env(self.h, env)

    def active_contacts(self, env=0):
        return self.l.oracle_active_contacts(self.h, env)


def state_to_dict(st):
    """Flatten a DrivingState into numpy arrays for comparisons."""
    nc, npd, no = st.n_cars, st.n_peds, st.n_obst
    cars_f = np.array([[getattr(st.cars[i], n) for n in CAR_F] for i in range(nc)]).reshape(nc, len(CAR_F))
    cars_i = np.array([[getattr(st.cars[i], n) for n in CAR_I] for i in range(nc)], dtype=np.int64)
    peds_f = np.array([[getattr(st.peds[i], n) for n in PED_F] for i in range(npd)]).reshape(npd, 4)
    peds_i = np.array([[getattr(st.peds[i], n) for n in PED_I] for i in range(npd)], dtype=np.int64).reshape(npd, 7)
    return dict(scalars=np.array([st.elapsed, st.all_finished, nc, npd, no, st.episode]),
                episode_r=np.array(st.episode_r[:nc]), episode_pos_r=np.array(st.episode_pos_r[:nc]),
                cars_f=cars_f, cars_i=cars_i, peds_f=peds_f, peds_i=peds_i,
                obst=np.array([list(st.obst_x[:no]), list(st.obst_y[:no])]))


def rc_state_to_dict(st):
    n = st.n_robots
    rf = np.array([[getattr(st.robots[i], k) for k in ROBOT_F] for i in range(n)])
    ri = np.array([[getattr(st.robots[i], k) for k in ROBOT_I] for i in range(n)], dtype=np.int64)
    sc = np.array([st.elapsed, n, st.ball_owned, st.n_last_kicked] + list(st.last_kicked)[:st.n_last_kicked] +
                  list(st.goals) + list(st.closest) + list(st.n_def) + list(st.defenders[0])[:st.n_def[0]] +
                  list(st.defenders[1])[:st.n_def[1]] + [st.episode])
    fl = np.array([st.ball_free_cntr, st.grace_period, st.penal_times[0], st.penal_times[1], st.bpx, st.bpy, st.bvx,
                   st.bvy, st.bw, st.bprevx, st.bprevy])
    return dict(robots_f=rf, robots_i=ri, scalars=sc, floats=fl, episode_r=np.array(st.episode_r[:n]),
                episode_pos_r=np.array(st.episode_pos_r[:n]))
