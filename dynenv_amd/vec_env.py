"""Host-side mirror of the reference's vectorised environment for the batched HIP step().

`BatchedDynEnv` exposes the surface callers of `make_dyn_env` rely on (DynEnv/utils/base_vec_env.py:57-233 VecEnv,
DynEnv/utils/subproc_vec_env.py:75-172 SubprocVecEnv): `reset`, `step_async`, `step_wait`, `step`, `close`,
`get_attr`, `set_attr`, `env_method`, `seed`, `num_envs`, `observation_space`, `action_space` — but all `num_envs`
environments live on one MI355X and one call into libdynenv_hip.so steps them together.

Two ways to consume a step:
  * fast path  `step_flat(actions)` -> (obs[E,T,A,D] f32, rewards[E,A] f64, dones[E] u8) torch tensors in HBM;
  * compat path `step(actions)` -> the reference's ragged object array obs[E,T,A,3], np rewards/dones, info dicts
    (materialised on the host from the dense tensors, only when a legacy consumer asks).

PyTorch is used for device memory and streams only.  There is no CPU fallback: constructing an env without a GPU,
or without the built library, raises.
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Tuple as TTuple

import numpy as np

from . import _capi
from . import spaces as sp
from .enums import DynEnvType, NoiseType, ObservationType


@dataclass
class StateSpaceDescriptor:  # environment_base.py:18-21
    numItemsPerGridCell: int
    space: object


@dataclass
class PredictionDescriptor:  # environment_base.py:24-31
    numContinuous: int = None
    numBinary: int = 0
    contIdx: List[int] = None
    binaryIdx: List[int] = None
    posIdx: List[int] = (0, 1)
    categoricIdx: int = None


@dataclass
class RecoDescriptor:  # environment_base.py:34-38
    featureGridSize: TTuple[int, int]
    fullStateSpace: List[StateSpaceDescriptor]
    targetDefs: List[PredictionDescriptor]


def _driving_spaces(obs_type):
    """DrivingEnvironment._create_observation_space / _setup_action_space / _setup_reconstruction_info
    (DrivingEnvironment.py:129-232); mean = 5.0 always (quirk C1, :235)."""
    mean = 5.0
    pos_xy = sp.Box(-mean * 2, +mean * 2, shape=(2,))
    width_height = sp.Box(-10, 10, shape=(2,))
    orientation = sp.Box(-1, 1, shape=(2,))
    typ = sp.Box(-1, 1, shape=(1,))
    self_space = sp.Dict([("position", pos_xy), ("orientation", orientation), ("width_height", width_height),
                          ("goal_position", pos_xy), ("finished", sp.MultiBinary(1))])
    car_space = sp.Dict([("position", pos_xy), ("orientation", orientation), ("width_height", width_height),
                         ("finished", sp.MultiBinary(1))])
    pedestrian_space = sp.Dict([("position", pos_xy)])
    if obs_type == ObservationType.FULL:
        lane_space = sp.Dict([("points", sp.Box(-mean * 2, mean * 2, shape=(4,))), ("type", typ)])
        obstacle_space = sp.Dict([("position", pos_xy), ("width_height", width_height)])
    else:
        lane_space = sp.Dict([("signed_distance", sp.Box(-mean * 2, mean * 2, shape=(1,))),
                              ("orientation", orientation), ("type", typ)])
        obstacle_space = sp.Dict([("position", pos_xy), ("orientation", orientation), ("width_height", width_height)])
    observation_space = sp.Tuple([sp.Tuple([car_space, obstacle_space, pedestrian_space]),
                                  sp.Tuple([self_space, lane_space])])
    action_space = sp.Tuple((sp.MultiDiscrete([3, 3]),))
    size = sp.Box(-10, 10, shape=(2,))
    conf = sp.MultiBinary(1)
    self_state = StateSpaceDescriptor(1, sp.Dict([("position", pos_xy), ("orientation", orientation), ("size", size),
                                                  ("confidence", conf)]))
    car_state = StateSpaceDescriptor(4, sp.Dict([("position", pos_xy), ("orientation", orientation), ("size", size),
                                                 ("confidence", conf)]))
    obstacle_state = StateSpaceDescriptor(4, sp.Dict([("position", pos_xy), ("size", size), ("confidence", conf)]))
    ped_state = StateSpaceDescriptor(6, sp.Dict([("position", pos_xy), ("confidence", conf)]))
    reco = RecoDescriptor(featureGridSize=(10, 17),
                          fullStateSpace=[self_state, car_state, obstacle_state, ped_state],
                          targetDefs=[PredictionDescriptor(numContinuous=4, contIdx=[2, 3, 4, 5]),
                                      PredictionDescriptor(numContinuous=4, contIdx=[2, 3, 4, 5]),
                                      PredictionDescriptor(numContinuous=2, contIdx=[2, 3]),
                                      PredictionDescriptor(numContinuous=0)])
    return observation_space, action_space, reco


def _robocup_spaces(obs_type, allow_head_turn):
    """RoboCupEnvironment._create_observation_space / _setup_action_space / _setup_reconstruction_info
    (RoboCupEnvironment.py:101-132,338-430), Full observation; mean = 2.0 always (quirk C1, :67)."""
    mean = 2.0
    pos_xy = sp.Box(-mean * 2, +mean * 2, shape=(2,))
    team = sp.Box(-1, 1, shape=(1,))
    self_space = sp.Dict([("position", pos_xy), ("orientation", sp.Box(-1, 1, shape=(4,))), ("team", team),
                          ("penalized or penalized", sp.MultiBinary(1))])
    ball_space = sp.Dict([("position", pos_xy), ("team", team), ("closest", sp.MultiBinary(1))])
    robot_space = sp.Dict([("position", pos_xy), ("orientation", sp.Box(-1, 1, shape=(2,))), ("team", team),
                           ("penalized or penalized", sp.MultiBinary(1))])
    observation_space = sp.Tuple([sp.Tuple([ball_space, robot_space]), sp.Tuple([self_space, ])])
    if obs_type == ObservationType.PARTIAL:  # RoboCupEnvironment.py:346-430, the `else` branch of _create_observation_space
        rad = sp.Box(-mean * 2, +mean * 2, shape=(1,))
        pos_radial = sp.Box(-1, +1, shape=(3,))
        typ = sp.Box(-1, +1, shape=(2,))
        line_space = sp.Dict([("position", pos_radial), ("type", typ)])
        cross_space = goalpost_space = sp.Dict([("position", pos_radial), ("radius", rad), ("type", typ)])
        field_cross_space = sp.Dict([("position", pos_radial), ("radius", rad), ("type", typ), ("angle", sp.Box(-1, +1, shape=(2,)))])
        p_robot = sp.Dict([("position", pos_xy), ("radius", rad), ("orientation", sp.Box(-1, 1, shape=(2,))), ("team", team),
                           ("penalized or penalized", sp.MultiBinary(1))])
        p_ball = sp.Dict([("position", pos_xy), ("radius", rad), ("team", team), ("closest", sp.MultiBinary(1))])
        observation_space = sp.Tuple([sp.Tuple([p_ball, p_robot]),
                                      sp.Tuple([goalpost_space, cross_space, field_cross_space, line_space])])
    if allow_head_turn:
        action_space = sp.Tuple((sp.MultiDiscrete([5, 3, 3]), sp.Box(low=-3, high=3, shape=(1,))))
    else:
        action_space = sp.Tuple((sp.MultiDiscrete([5, 3, 3, 7]),))
    position_xy = sp.Box(-2, +2, shape=(2,))
    conf = sp.MultiBinary(1)
    ball_state = StateSpaceDescriptor(1, sp.Dict([("position", position_xy), ("team", sp.Box(-1, 1, shape=(1,))),
                                                  ("confidence", conf)]))
    robot_state = StateSpaceDescriptor(4, sp.Dict([("position", position_xy),
                                                   ("orientation", sp.Box(-1.0, +1.0, shape=(2,))),
                                                   ("team", sp.Box(-1, 1, shape=(1,))), ("active", sp.MultiBinary(1)),
                                                   ("confidence", conf)]))
    reco = RecoDescriptor(featureGridSize=(1, 1), fullStateSpace=[ball_state, robot_state],
                          targetDefs=[PredictionDescriptor(numContinuous=1, contIdx=[2, ]),
                                      PredictionDescriptor(numContinuous=3, numBinary=1, contIdx=[2, 3, 4], binaryIdx=[5, ])])
    return observation_space, action_space, reco


class LazyInfo(dict):
    """The per-environment `info` dict of the reference (subproc_vec_env.py:17-23, DrivingEnvironment.py:306-316) whose two
    expensive entries - 'Full State' and 'Recon States', lists of per-agent arrays - are built from the step's single
    host copy of the observations only when somebody reads them.  Everything else of the dict protocol
    behaves as if they had been there all along."""
    LAZY = ("Full State", "Recon States")

    def __init__(self, make, eager=None):
        dict.__init__(self, eager or {})
        self._make = make

    def _materialise(self):
        if self._make is not None:
            full, recon = self._make()
            self._make = None
            dict.__setitem__(self, "Full State", full)
            dict.__setitem__(self, "Recon States", recon)

    def __missing__(self, key):
        if key in self.LAZY and self._make is not None:
            self._materialise()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        if key in self.LAZY:
            self._materialise()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key in self.LAZY or dict.__contains__(self, key)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        self._materialise()
        return dict.__len__(self)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __repr__(self):
        self._materialise()
        return dict.__repr__(self)


class BatchedDynEnv(object):
    """All `num_envs` environments of one GPU shard behind the reference's VecEnv surface."""

    metadata = {"render.modes": []}

    def __init__(self, env_type, num_envs, num_players, observationType=ObservationType.FULL,
                 noiseType=NoiseType.REALISTIC, noiseMagnitude=0, use_continuous_actions=False, seed=42,
                 device=None, env_id_offset=0, flags=0, out_buffers=None):
        import torch  # device memory + streams only
        if not torch.cuda.is_available():
            raise _capi.DynEnvError("dynenv_amd needs an MI355X (HIP device); there is no CPU fallback")
        self._torch = torch
        self._lib = _capi.load()
        env_type = DynEnvType(env_type)
        if env_type == DynEnvType.DRIVE and use_continuous_actions:
            # reference quirk C4: the continuous-action branch of DrivingEnvironment.processAction is broken
            raise NotImplementedError("continuous actions are broken in the reference Driving env (acc/steer unbound)")
        if observationType == ObservationType.IMAGE:
            raise NotImplementedError("Image observations are out of scope (SURVEY.md §2)")
        if env_type == DynEnvType.ROBO_CUP and flags == 0:
            # class-level switches of the reference (RoboCupEnvironment.py:18-21): canFall=True, useObsRewards=True;
            # make_dyn_env passes allowHeadTurn=use_continuous_actions (DynEnv/__init__.py:9-11)
            flags = _capi.FLAG_CAN_FALL | _capi.FLAG_USE_OBS_REWARDS
            if use_continuous_actions:
                flags |= _capi.FLAG_ALLOW_HEAD_TURN
        self.env_type = env_type
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.cfg = _capi.Cfg(_capi.DYNENV_ABI_VERSION, int(env_type), int(num_envs), int(num_players),
                             int(observationType), int(noiseType), float(noiseMagnitude), int(seed),
                             int(env_id_offset), int(flags), self.device.index or 0, 0)
        self._h = C.c_void_p()
        _capi.check(self._lib.dynenv_create(C.byref(self.cfg), C.byref(self._h)), "dynenv_create")
        self.layout = _capi.Layout()
        _capi.check(self._lib.dynenv_layout(self._h, C.byref(self.layout)), "dynenv_layout")
        L = self.layout
        self.num_envs, self.n_agents, self.n_time_steps = L.num_envs, L.n_agents, L.n_time_steps
        self.obs_dim, self.action_dim, self.steps_per_episode = L.obs_dim, L.action_dim, L.steps_per_episode
        self.observationType, self.noiseType, self.noiseMagnitude = observationType, noiseType, noiseMagnitude
        if env_type == DynEnvType.DRIVE:
            self.observation_space, self.action_space, self.recoDescriptor = _driving_spaces(observationType)
            self.stepNum = 6000 / 10.0  # DrivingEnvironment.py:49
        else:
            self.observation_space, self.action_space, self.recoDescriptor = _robocup_spaces(
                observationType, bool(flags & _capi.FLAG_ALLOW_HEAD_TURN))
            self.stepNum = 12000 / 10.0 / 5  # RoboCupEnvironment.py:61
        E, T, A, D = self.num_envs, self.n_time_steps, self.n_agents, self.obs_dim
        if out_buffers is None:
            self.obs = torch.zeros((E, T, A, D), dtype=torch.float32, device=self.device)
            self.rewards = torch.zeros((E, A), dtype=torch.float64, device=self.device)
            self.dones = torch.zeros((E,), dtype=torch.uint8, device=self.device)
        else:  # e.g. views into a packed all-gather slab (dynenv_amd.distributed)
            self.obs, self.rewards, self.dones = out_buffers
        self._actions = torch.zeros((E, A, self.action_dim), dtype=torch.int32, device=self.device)
        self.terminal_obs = None
        self._episode_step = 0
        self._needs_reset = True
        self._pending = None
        self.closed = False

    def use_buffers(self, obs, rewards, dones):
        """Switch the output buffers of the next reset/step (ping-pong slabs of dynenv_amd.distributed.StepGather)."""
        assert tuple(obs.shape) == tuple(self.obs.shape) and obs.dtype == self.obs.dtype and obs.is_contiguous()
        self.obs, self.rewards, self.dones = obs, rewards, dones

    # ------------------------------------------------------------------ fast path
    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def reset_flat(self):
        _capi.check(self._lib.dynenv_reset(self._h, C.c_void_p(self.obs.data_ptr()), self._stream()), "dynenv_reset")
        self._episode_step = 0
        self._needs_reset = False
        return self.obs

    def _stage_actions(self, actions):
        torch = self._torch
        if isinstance(actions, torch.Tensor):
            a = actions
            if a.device != self.device or a.dtype != torch.int32 or not a.is_contiguous():
                a = a.to(device=self.device, dtype=torch.int32).contiguous()
        else:
            a = torch.as_tensor(np.ascontiguousarray(actions, dtype=np.int32)).to(self.device)
        if tuple(a.shape) != (self.num_envs, self.n_agents, self.action_dim):
            # DrivingEnvironment.py:262-263
            raise Exception("Error: There must be %d actions for every %s" %
                            (self.action_dim, "car" if self.env_type == DynEnvType.DRIVE else "robot"))
        return a

    def step_flat(self, actions, auto_reset=True, validate=False):
        """One env step of every environment: ONE kernel launch on torch's current stream."""
        if self._needs_reset:
            raise _capi.DynEnvError("call reset() before step()")
        torch = self._torch
        a = self._stage_actions(actions)
        if validate and self.env_type == DynEnvType.DRIVE:
            if bool(((a < 0) | (a > 2)).any()):  # DrivingEnvironment.py:365-368
                raise Exception("Error: Acceleration must be between +/-3")
        if validate and self.env_type == DynEnvType.ROBO_CUP:  # RoboCupEnvironment.py:543-550
            hi = torch.tensor([4, 2, 2, 6], device=a.device, dtype=a.dtype)
            if bool(((a < 0) | (a > hi)).any()):
                raise Exception("Error: Robot movement must be categorical in the range [0-4]")
        _capi.check(self._lib.dynenv_step(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                          C.c_void_p(self.rewards.data_ptr()), C.c_void_p(self.dones.data_ptr()),
                                          self._stream()), "dynenv_step")
        self._episode_step += 1
        self.last_done = self._episode_step >= self.steps_per_episode  # fixed-length episodes
        if self.last_done and auto_reset:
            # SubprocVecEnv worker semantics (subproc_vec_env.py:19-22): keep the terminal observation, return the reset one
            self.terminal_obs = self.obs.clone()
            self._episode_stats = self.episode_stats()
            self.reset_flat()
        return self.obs, self.rewards, self.dones

    def episode_stats(self):
        torch = self._torch
        E, A = self.num_envs, self.n_agents
        r = torch.empty((E, A), dtype=torch.float64, device=self.device)
        p = torch.empty_like(r)
        o = torch.empty_like(r)
        g = torch.empty((E, 2), dtype=torch.int32, device=self.device)
        _capi.check(self._lib.dynenv_episode_stats(self._h, C.c_void_p(r.data_ptr()), C.c_void_p(p.data_ptr()),
                                                   C.c_void_p(o.data_ptr()), C.c_void_p(g.data_ptr()), self._stream()),
                    "dynenv_episode_stats")
        return r, p, o, g

    def counts(self):
        c = self._torch.empty((self.num_envs, 2), dtype=self._torch.int32, device=self.device)
        _capi.check(self._lib.dynenv_counts(self._h, C.c_void_p(c.data_ptr()), self._stream()), "dynenv_counts")
        return c

    def error_flags(self):
        f = C.c_int32(0)
        _capi.check(self._lib.dynenv_error_flags(self._h, C.byref(f)), "dynenv_error_flags")
        return f.value

    def debug_counters(self):
        out = (C.c_int64 * 12)()
        _capi.check(self._lib.dynenv_debug_counters(self._h, out), "dynenv_debug_counters")
        return dict(fast=out[0], quiescent=out[1], contact=out[2], slot_sum=out[3], why_cand=out[4], why_moving=out[5],
                    why_inert=out[6], steady=out[7], light=out[8])

    def get_state(self, env=0):
        st = _capi.DrivingState() if self.env_type == DynEnvType.DRIVE else _capi.RoboCupState()
        _capi.check(self._lib.dynenv_get_state(self._h, env, C.byref(st), C.sizeof(st)), "dynenv_get_state")
        return st

    def set_state(self, env, st):
        _capi.check(self._lib.dynenv_set_state(self._h, env, C.byref(st), C.sizeof(st)), "dynenv_set_state")
        self._needs_reset = False

    # ------------------------------------------------------------------ exact checkpoint
    def checkpoint(self):
        """Every device array of the handle, bit for bit, as a numpy uint8 array (contact cache, shortcut state, episode
        counters and seed included): restore() + the same actions reproduces the run exactly, mid-episode too."""
        import numpy as np
        n = int(self._lib.dynenv_checkpoint_size(self._h))
        buf = np.empty((n,), np.uint8)
        _capi.check(self._lib.dynenv_checkpoint_save(self._h, C.c_void_p(buf.ctypes.data), n), "dynenv_checkpoint_save")
        return buf

    def restore(self, buf):
        import numpy as np
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        _capi.check(self._lib.dynenv_checkpoint_load(self._h, C.c_void_p(buf.ctypes.data), buf.size), "dynenv_checkpoint_load")
        self._needs_reset = False

    def refresh_obs(self):
        """Re-emit the observation of the current state (after set_state)."""
        raise NotImplementedError

    # ------------------------------------------------------------------ reference-compatible (legacy) path
    def _compat_obs(self, obs_t, counts):
        """dense [E,T,A,D] -> object ndarray [E,T,A,3] of ((cars, obstacles, peds), (self, lanes), (1,1,1))"""
        o = obs_t if isinstance(obs_t, np.ndarray) else obs_t.detach().cpu().numpy()
        E, T, A, D = o.shape
        L = self.layout
        off, rows, feat = list(L.block_offset), list(L.block_rows), list(L.block_feat)
        out = np.empty((E, T, A, 3), dtype=object)
        if self.env_type == DynEnvType.ROBO_CUP and self.observationType == ObservationType.PARTIAL:
            # getAgentVision: ((balls, robots), (goals, crosses, line crosses, lines), (numLandMarks, robotsSeen, ballsSeen))
            tail = off[6]
            for e in range(E):
                for t in range(T):
                    for a in range(A):
                        r = o[e, t, a]
                        n = [int(x) for x in r[tail:tail + 6]]
                        lists = [r[off[k]:off[k] + n[k] * feat[k]].reshape(n[k], feat[k]) for k in range(6)]
                        out[e, t, a, 0] = [lists[0], lists[1]]
                        out[e, t, a, 1] = [lists[2], lists[3], lists[4], lists[5]]
                        out[e, t, a, 2] = (int(r[tail + 6]), r[tail + 8:tail + 8 + (A - 1)].astype("uint8"), bool(r[tail + 7]))
            return out
        # Views into `o` (a fresh host copy per step), blocks reshaped once for the whole batch: the per-(env, time, agent)
        # Python work is only the assembly of the reference's nested lists.
        ones = (1, 1, 1)
        if self.env_type == DynEnvType.ROBO_CUP:  # ((ball, robots), (self,), (1,1,1)) RoboCupEnvironment.py:440-443
            ball = o[..., 0:4].reshape(E, T, A, 1, 4)
            selfr = o[..., 4:12].reshape(E, T, A, 1, 8)
            robs = o[..., 12:12 + (A - 1) * 6].reshape(E, T, A, A - 1, 6)
            for e in range(E):
                for t in range(T):
                    for a in range(A):
                        out[e, t, a, 0] = [ball[e, t, a], robs[e, t, a]]
                        out[e, t, a, 1] = [selfr[e, t, a], ]
                        out[e, t, a, 2] = ones
            return out
        if self.observationType == ObservationType.PARTIAL:  # ragged rows of getAgentVision, lengths in the last 4 floats
            cars = o[..., off[1]:off[1] + rows[1] * 7].reshape(E, T, A, rows[1], 7)
            obst = o[..., off[2]:off[2] + rows[2] * 6].reshape(E, T, A, rows[2], 6)
            peds = o[..., off[3]:off[3] + rows[3] * 2].reshape(E, T, A, rows[3], 2)
            lanes = o[..., off[4]:off[4] + rows[4] * 4].reshape(E, T, A, rows[4], 4)
            selfr = o[..., 0:9].reshape(E, T, A, 1, 9)
            n = o[..., D - 4:].astype(np.int64)
            for e in range(E):
                for t in range(T):
                    for a in range(A):
                        nc, no, npd, nl = n[e, t, a]
                        out[e, t, a, 0] = [cars[e, t, a, :nc], obst[e, t, a, :no], peds[e, t, a, :npd]]
                        out[e, t, a, 1] = [selfr[e, t, a], lanes[e, t, a, :nl]]
                        out[e, t, a, 2] = ones
            return out
        selfr = o[..., off[0]:off[0] + 9].reshape(E, T, A, 1, 9)
        cars = o[..., off[1]:off[1] + rows[1] * 7].reshape(E, T, A, rows[1], 7)
        obst = o[..., off[2]:off[2] + rows[2] * 4].reshape(E, T, A, rows[2], 4)
        peds = o[..., off[3]:off[3] + rows[3] * 2].reshape(E, T, A, rows[3], 2)
        lanes = o[..., off[4]:off[4] + rows[4] * 5].reshape(E, T, A, rows[4], 5)
        for e in range(E):
            n_obst, n_ped = int(counts[e, 0]), int(counts[e, 1])
            for t in range(T):
                for a in range(A):
                    out[e, t, a, 0] = [cars[e, t, a], obst[e, t, a, :n_obst], peds[e, t, a, :n_ped]]
                    out[e, t, a, 1] = [selfr[e, t, a], lanes[e, t, a]]
                    out[e, t, a, 2] = ones
        return out

    def _full_states(self, obs_np, counts, e):
        """info['Full State'] / info['Recon States'] (DrivingEnvironment.py:306-307) from agent rows of env e."""
        L = self.layout
        off, rows = list(L.block_offset), list(L.block_rows)
        A = self.n_agents
        if self.env_type == DynEnvType.ROBO_CUP and self.observationType == ObservationType.PARTIAL:
            return None, None  # the Partial launch does not emit the noise-free full state (see the Driving note below)
        if self.env_type == DynEnvType.ROBO_CUP:  # getFullState(agent) = [ball, self, robots] (:1164-1188)
            recon = []
            for a in range(A):
                r = obs_np[e, -1, a]
                recon.append([r[0:4].reshape(1, 4).copy(), r[4:12].reshape(1, 8).copy(),
                              r[12:12 + (A - 1) * 6].reshape(A - 1, 6).copy()])
            return recon[0], recon  # (the reference's agent=None variant is un-normalised; not reproduced)
        n_obst, n_ped = int(counts[e, 0]), int(counts[e, 1])
        if self.observationType == ObservationType.PARTIAL:
            # info['Full State'] needs the noise-free full state, which the Partial launch does not emit (only the
            # out-of-scope reconstruction losses of the reference's trainer consume it)
            return None, None
        recon = []
        for a in range(A):
            r = obs_np[e, -1, a]
            recon.append([r[0:9].reshape(1, 9).copy(), r[off[1]:off[1] + rows[1] * 7].reshape(rows[1], 7).copy(),
                          r[off[2]:off[2] + n_obst * 4].reshape(n_obst, 4).copy(),
                          r[off[3]:off[3] + n_ped * 2].reshape(n_ped, 2).copy(),
                          r[off[4]:off[4] + rows[4] * 5].reshape(rows[4], 5).copy()])
        # complete state: every car row = own [x,y,cos,sin,w,h] + finished
        cars = np.stack([np.concatenate([recon[a][0][0, :6], recon[a][0][0, 8:9]]) for a in range(A)]).astype(np.float32)
        full = [cars, recon[0][2], recon[0][3], recon[0][4]]
        return full, recon

    def reset(self):
        self.reset_flat()
        counts = self.counts().cpu().numpy()
        return self._compat_obs(self.obs, counts)

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        actions, self._pending = self._pending, None
        a = np.asarray(actions)
        counts_before = self.counts().cpu().numpy()
        self.step_flat(a, auto_reset=False, validate=True)
        rewards = self.rewards.cpu().numpy().copy()
        done = bool(self.last_done)
        dones = np.full((self.num_envs,), done, dtype=bool)
        obs_np = self.obs.cpu().numpy()  # the step's ONE device->host copy; obs and infos are views of / built from it
        infos = []
        stats = None
        if done:
            stats = [x.cpu().numpy() for x in self.episode_stats()]
            term = self._compat_obs(obs_np, counts_before)
        for e in range(self.num_envs):
            eager = {}
            if done:
                eager["episode_r"] = stats[0][e].copy()
                eager["episode_p_r"] = stats[1][e].copy()
                eager["episode_o_r"] = stats[2][e].copy() if self.env_type == DynEnvType.ROBO_CUP else [0, ] * self.n_agents
                eager["episode_g"] = [int(stats[3][e, 0]), int(stats[3][e, 1])]
                eager["terminal_observation"] = [list(term[e, t]) for t in range(self.n_time_steps)]
            infos.append(LazyInfo(lambda e=e: self._full_states(obs_np, counts_before, e), eager))
        if done:
            self.terminal_obs = self.obs.clone()
            self.reset_flat()
            counts = self.counts().cpu().numpy()
            obs = self._compat_obs(self.obs, counts)
        else:
            obs = self._compat_obs(obs_np, counts_before)
        return obs, rewards, dones, tuple(infos)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        """Working replacement for SubprocVecEnv.seed (which calls a non-existent env.seed, quirk C5)."""
        _capi.check(self._lib.dynenv_seed(self._h, int(seed if seed is not None else 0)), "dynenv_seed")
        return [seed] * self.num_envs

    def _get_indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def get_attr(self, attr_name, indices=None):
        idx = self._get_indices(indices)
        if not hasattr(self, attr_name):
            raise AttributeError(attr_name)
        return [getattr(self, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        idx = self._get_indices(indices)
        if method_name == "get_agent_locs":
            o = self.obs.cpu().numpy()
            if self.env_type == DynEnvType.ROBO_CUP:  # RoboCupEnvironment.py:434-435: self rows [:, 0:6]
                return [[o[e, -1, a, 4:10].reshape(1, 6).copy() for a in range(self.n_agents)] for e in idx]
            # DrivingEnvironment.py:126-127: self rows [x, y, cos, sin] per agent
            return [[o[e, -1, a, 0:4].reshape(1, 4).copy() for a in range(self.n_agents)] for e in idx]
        if method_name == "set_random_seed":
            return self.seed(*method_args)
        raise NotImplementedError(method_name)

    def render(self, *a, **k):
        return None

    def close(self):
        if not self.closed and self._h:
            self._lib.dynenv_destroy(self._h)
            self._h = None
            self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_dyn_env(env, num_envs, num_players, render, observationType, noiseType, noiseMagnitude,
                 use_continuous_actions, **kwargs):
    """Drop-in for DynEnv.make_dyn_env (DynEnv/__init__.py:6-25): returns (vec_env, name)."""
    if render:
        raise NotImplementedError("rendering (pygame/cv2) is out of scope; use render=False")
    if env is DynEnvType.ROBO_CUP or env == DynEnvType.ROBO_CUP:
        name = "RoboCup"
    elif env is DynEnvType.DRIVE or env == DynEnvType.DRIVE:
        name = "Driving"
    else:
        raise ValueError
    venv = BatchedDynEnv(DynEnvType(env), num_envs, num_players, observationType, noiseType, noiseMagnitude,
                         use_continuous_actions, **kwargs)
    return venv, name
