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


def _to_host(t):
    """Device tensor -> numpy through a PINNED staging tensor: torch's caching host allocator hands the block of the previous
    step back (no 38 MB of fresh page faults per step, ~20 ms at 4096 envs) and the copy runs at PCIe speed.  The numpy array
    keeps the tensor alive; every step gets its own block, as the reference returns fresh arrays."""
    import torch
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    return h.numpy()


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


class LazyObsArray(object):
    """The reference's observation `np.ndarray(dtype=object)` of shape [E, T, A, 3] (subproc_vec_env.py:201 over
    DrivingEnvironment.py:123 / RoboCupEnvironment.py:442) without the E*T*A Python objects: it keeps the step's dense host
    copy [E, T, A, D] and builds an element - `(movable-object arrays, static/self arrays, seen info)` - when it is indexed.
    Indexing follows numpy's basic rules (ints, slices, Ellipsis); a sub-array is again lazy (`obs[..., :-1]`,
    `obs[..., -1]` as models/train.py:67-68 does); `np.asarray(obs)` / `obs.materialize()` gives the reference's eager object array."""
    dtype = np.dtype(object)

    def __init__(self, owner, dense, counts, sel=None, box=None):
        # `dense`: the step's observations [E, T, A, D] - a numpy array, or a device tensor (a snapshot the step made in HBM)
        # that is copied to the host the first time an element is looked at; `box` shares that copy among sub-arrays
        self._owner, self._counts = owner, counts
        self._box = box if box is not None else [dense]
        E, T, A, _ = dense.shape
        self._sel = sel if sel is not None else (range(E), range(T), range(A), range(3))  # per axis: range (kept) or int (dropped)

    @property
    def _dense(self):
        d = self._box[0]
        if not isinstance(d, np.ndarray):
            d = self._box[0] = _to_host(d)
        return d

    @property
    def shape(self):
        return tuple(len(s) for s in self._sel if not isinstance(s, int))

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        sh = self.shape
        if not sh:
            raise TypeError("len() of unsized object")
        return sh[0]

    def _element(self, e, t, a, k):
        return self._owner._compat_element(self._dense, self._counts, e, t, a)[k]

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        kept = [i for i, s in enumerate(self._sel) if not isinstance(s, int)]
        if any(i is Ellipsis for i in idx):
            p = [i for i, x in enumerate(idx) if x is Ellipsis][0]
            idx = idx[:p] + (slice(None),) * (len(kept) - (len(idx) - 1)) + idx[p + 1:]
        if len(idx) > len(kept):
            raise IndexError("too many indices for array")
        idx = idx + (slice(None),) * (len(kept) - len(idx))
        sel = list(self._sel)
        for ax, i in zip(kept, idx):
            r = sel[ax]
            if isinstance(i, slice):
                sel[ax] = r[i]
            else:
                sel[ax] = r[int(i)]   # IndexError like numpy when out of range
        if all(isinstance(x, int) for x in sel):
            return self._element(*sel)
        return LazyObsArray(self._owner, self._box[0], self._counts, tuple(sel), self._box)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def materialize(self):
        """The eager object ndarray of this (sub-)array, element for element what the reference returns."""
        E, T, A, _ = self._dense.shape
        if all(isinstance(x, range) and x == range(n) for x, n in zip(self._sel, (E, T, A, 3))) and hasattr(self._owner, "_compat_obs_bulk"):
            return self._owner._compat_obs_bulk(self._dense, self._counts)  # the whole array: the bulk builder
        axes = [([s] if isinstance(s, int) else list(s)) for s in self._sel]
        out = np.empty(tuple(len(a) for a in axes), dtype=object)
        for ie, e in enumerate(axes[0]):
            for it, t in enumerate(axes[1]):
                for ia, a in enumerate(axes[2]):
                    el = self._owner._compat_element(self._dense, self._counts, e, t, a)
                    for ik, k in enumerate(axes[3]):
                        out[ie, it, ia, ik] = el[k]
        return out.reshape(self.shape)

    def __array__(self, dtype=None, copy=None):
        return self.materialize()

    def tolist(self):
        return self.materialize().tolist()

    def __repr__(self):
        return "LazyObsArray(shape=%r, dtype=object)" % (self.shape,)


class LazyInfos(object):
    """The per-environment `info` dicts of one step as a read-only sequence (the reference returns a tuple of dicts,
    subproc_vec_env.py:109-111): a dict is built when it is asked for and kept, so a consumer that never looks at
    `info` pays nothing for 4096 of them."""

    def __init__(self, n, make):
        self._n, self._make, self._cache = n, make, {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[k] for k in range(*i.indices(self._n)))
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        d = self._cache.get(i)
        if d is None:
            d = self._cache[i] = self._make(i)
        return d

    def __iter__(self):
        return (self[i] for i in range(self._n))


class BatchedDynEnv(object):
    """All `num_envs` environments of one GPU shard behind the reference's VecEnv surface."""

    metadata = {"render.modes": []}

    def __init__(self, env_type, num_envs, num_players, observationType=ObservationType.FULL,
                 noiseType=NoiseType.REALISTIC, noiseMagnitude=0, use_continuous_actions=False, seed=42,
                 device=None, env_id_offset=0, flags=None, out_buffers=None, eager_compat=False):
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
        if flags is None:  # the reference's defaults; an explicit value (0 included: canFall = useObsRewards = False) is taken as is
            flags = 0
            if env_type == DynEnvType.ROBO_CUP:
                # class-level switches of the reference (RoboCupEnvironment.py:18-21): canFall=True, useObsRewards=True,
                # randomInit=False, deterministicTurn=False; make_dyn_env passes allowHeadTurn=use_continuous_actions
                # (DynEnv/__init__.py:9-11)
                flags = _capi.FLAG_CAN_FALL | _capi.FLAG_USE_OBS_REWARDS
                if use_continuous_actions:
                    flags |= _capi.FLAG_ALLOW_HEAD_TURN
        self.flags = int(flags)
        self.eager_compat = bool(eager_compat)  # step()/reset() return the eager object ndarray / tuple of dicts instead of the lazy forms
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
        self.allow_head_turn = env_type == DynEnvType.ROBO_CUP and bool(self.flags & _capi.FLAG_ALLOW_HEAD_TURN)
        self.full_obs_dim = int(self._lib.dynenv_full_obs_dim(self._h))
        self._counts_np = None
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
        self._counts_np = None  # the scene (obstacle / pedestrian counts) changed
        return self.obs

    def _stage_actions(self, actions):
        """-> (int32 [E, A, action_dim] on the device, float64 [E, A] head channel or None).  With allowHeadTurn the 4th action is
        the reference's continuous head turn (Box(-3, 3), RoboCupEnvironment.py:339-342): any float in the 4th column (or a
        (discrete [E, A, 3], head [E, A]) pair) travels as float64 and is not truncated."""
        torch = self._torch
        head = None
        if self.allow_head_turn:
            if isinstance(actions, (tuple, list)) and len(actions) == 2:
                disc, hd = actions
                disc = torch.as_tensor(np.asarray(disc) if not isinstance(disc, torch.Tensor) else disc)
                hd = torch.as_tensor(np.asarray(hd, dtype=np.float64) if not isinstance(hd, torch.Tensor) else hd)
                actions = torch.cat([disc.to(torch.float64), hd.to(torch.float64).reshape(disc.shape[0], disc.shape[1], 1)], -1)
            a_f = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(actions, dtype=np.float64))
            if a_f.dim() == 3 and a_f.shape[-1] == 4:
                head = a_f[..., 3].to(device=self.device, dtype=torch.float64).contiguous()
                a = a_f.to(device=self.device).to(torch.int32).contiguous()
                a[..., 3] = 0
            else:
                a = a_f.to(device=self.device).to(torch.int32).contiguous()
        elif isinstance(actions, torch.Tensor):
            a = actions
            if a.device != self.device or a.dtype != torch.int32 or not a.is_contiguous():
                a = a.to(device=self.device, dtype=torch.int32).contiguous()
        else:
            a = torch.as_tensor(np.ascontiguousarray(actions, dtype=np.int32)).to(self.device)
        if tuple(a.shape) != (self.num_envs, self.n_agents, self.action_dim):
            # DrivingEnvironment.py:262-263
            raise Exception("Error: There must be %d actions for every %s" %
                            (self.action_dim, "car" if self.env_type == DynEnvType.DRIVE else "robot"))
        return a, head

    def step_flat(self, actions, auto_reset=True, validate=False):
        """One env step of every environment: ONE kernel launch on torch's current stream.  Nothing is waited for and no flag is
        read here: a consumer of Partial observations that wants to know whether a list ever outgrew the layout's capacity (rows
        dropped; astronomically unlikely, include/dynenv.h error bit 3) calls error_flags() at its own pace; step() does, and raises."""
        if self._needs_reset:
            raise _capi.DynEnvError("call reset() before step()")
        torch = self._torch
        if auto_reset and torch.cuda.is_current_stream_capturing():
            # the host's position in the episode does not advance at replay: a reset decided now would be frozen into the graph (or never come)
            raise _capi.DynEnvError("step_flat(auto_reset=True) inside a stream capture: capture with auto_reset=False and reset between replays")
        a, head = self._stage_actions(actions)
        if validate and self.env_type == DynEnvType.DRIVE:
            if bool(((a < 0) | (a > 2)).any()):  # DrivingEnvironment.py:365-368
                raise Exception("Error: Acceleration must be between +/-3")
        if validate and self.env_type == DynEnvType.ROBO_CUP:  # RoboCupEnvironment.py:543-550
            hi = torch.tensor([4, 2, 2], device=a.device, dtype=a.dtype)
            if bool(((a[..., :3] < 0) | (a[..., :3] > hi)).any()):
                raise Exception("Error: Robot movement must be categorical in the range [0-4]")
            bad_head = (head.abs() > 6).any() if head is not None else ((a[..., 3] < 0) | (a[..., 3] > 6)).any()
            if bool(bad_head):
                raise Exception("Error: Head turn must be between +/-6")
        if head is not None:
            _capi.check(self._lib.dynenv_step_head(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(head.data_ptr()),
                                                   C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.rewards.data_ptr()),
                                                   C.c_void_p(self.dones.data_ptr()), self._stream()), "dynenv_step_head")
        else:
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
        out = (C.c_int64 * 16)()
        _capi.check(self._lib.dynenv_debug_counters(self._h, out), "dynenv_debug_counters")
        return dict(fast=out[0], quiescent=out[1], contact=out[2], slot_sum=out[3], why_cand=out[4], why_moving=out[5],
                    why_inert=out[6], steady=out[7], light=out[8], split=out[9], isolated_next=out[10], isolation_timeouts=out[11],
                    isolation_mode=out[12], placement_validated=out[13], placement_invalid_launches=out[14], isolation_pauses=out[15])

    def debug_placement(self):
        """uint32 [4096]: XCC << 16 | SE, SH, CU, SIMD bits of HW_ID of every regular block of the last step (mode 1 handles), else empty"""
        import numpy as np
        out = np.zeros((4096,), np.uint32)
        n = self._lib.dynenv_debug_placement(self._h, C.c_void_p(out.ctypes.data), out.size)
        if n < 0:
            _capi.check(n, "dynenv_debug_placement")
        return out[:n]

    def get_state(self, env=0):
        st = _capi.DrivingState() if self.env_type == DynEnvType.DRIVE else _capi.RoboCupState()
        _capi.check(self._lib.dynenv_get_state(self._h, env, C.byref(st), C.sizeof(st)), "dynenv_get_state")
        return st

    def _substeps(self):
        return 50 if self.env_type == DynEnvType.ROBO_CUP else 10

    def set_state(self, env, st):
        """Episodes are lock-step: `dones` and the auto-reset follow ONE host-side position in the episode, and that
        position follows the blob's `elapsed` - like restore() - so a blob with an edited `elapsed` moves host and device together.
        (Blobs that put different environments at different times make `dones` meaningless; the device keeps stepping them.)"""
        _capi.check(self._lib.dynenv_set_state(self._h, env, C.byref(st), C.sizeof(st)), "dynenv_set_state")
        step = int(st.elapsed) // self._substeps()
        if not self._needs_reset and step != self._episode_step and self.num_envs > 1 and not getattr(self, "_set_state_moved", False):
            # ONE blob with another `elapsed` than the batch's moves the host's episode position - and with it `dones` and the auto-reset of
            # ALL environments; the device keeps per-environment times.  Legitimate when every environment is being set (the tests do);
            # said once so that a scene written into one environment does not silently shift the others' episode end.
            import warnings
            warnings.warn("set_state: the blob's elapsed (%d) puts the batch's episode position at step %d (was %d): dones / auto-reset of every "
                          "environment follow it" % (int(st.elapsed), step, self._episode_step), stacklevel=2)
            self._set_state_moved = True
        self._needs_reset = False
        self._episode_step = step

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
        self._counts_np = None
        # the host's position in the (lock-step, fixed-length) episode follows the restored device state: auto-reset and
        # `dones` are driven by it
        self._episode_step = int(self.get_state(0).elapsed) // self._substeps()

    def full_state_obs(self):
        """The noise-free Full observation of the current state, [E, A, full_obs_dim] float32 on the device, whatever the
        observation type (getFullState: what info['Full State'] / info['Recon States'] are made of)."""
        out = self._torch.empty((self.num_envs, self.n_agents, self.full_obs_dim), dtype=self._torch.float32, device=self.device)
        _capi.check(self._lib.dynenv_full_obs(self._h, C.c_void_p(out.data_ptr()), self._stream()), "dynenv_full_obs")
        return out

    def global_state(self):
        """RoboCup: getFullState(agent=None) of the current state for every environment, float32 [E, 6 A + 3] on the device -
        robots [A, 6] = (x, y, cos, sin, team, fallen | penalized) in field coordinates, then the ball (x, y, ballOwned) - the
        content of info['Full State'] (RoboCupEnvironment.py:511, :1149-1161)."""
        if self.env_type != DynEnvType.ROBO_CUP:
            raise NotImplementedError("Driving's getFullState(None) is made of columns of full_state_obs()")
        out = self._torch.empty((self.num_envs, 6 * self.n_agents + 3), dtype=self._torch.float32, device=self.device)
        _capi.check(self._lib.dynenv_global_state(self._h, C.c_void_p(out.data_ptr()), self._stream()), "dynenv_global_state")
        return out

    def refresh_obs(self):
        """Re-emit the observation of the current state into `self.obs` (after set_state / restore).  Full observations only:
        a Partial observation is a noisy draw keyed by the step that produced it."""
        if self.observationType != ObservationType.FULL:
            raise NotImplementedError("refresh_obs re-emits Full observations; use full_state_obs() for the true state")
        full = self.full_state_obs()
        self.obs.copy_(full.unsqueeze(1).expand(-1, self.n_time_steps, -1, -1))
        return self.obs

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

    def _compat_obs_bulk(self, obs_t, counts):
        """`_compat_obs` without a Python-level loop over (env, time, agent): the per-agent arrays of a type are made by iterating
        the type's block once in C (`np.fromiter(iter(block), object)`: one view per row), ragged blocks grouped by their row
        count, and the reference's nested lists by `zip`.  Element for element what `_compat_obs` builds (the test compares
        them); ~1 us per agent instead of 2.6 (five ndarray objects and two lists per agent remain - the reference's format)."""
        import gc
        import itertools
        o = obs_t if isinstance(obs_t, np.ndarray) else _to_host(obs_t.detach())
        E, T, A, D = o.shape
        N = E * T * A
        gc_was_on = gc.isenabled()
        gc.disable()  # ~8 N container objects are about to be allocated: every 700th would trigger a collection pass over them
        try:
            return self._compat_obs_bulk_build(o, counts, E, T, A, D, N)
        finally:
            if gc_was_on:
                gc.enable()

    def _compat_obs_bulk_build(self, o, counts, E, T, A, D, N):
        import itertools
        L = self.layout
        off, rows, feat = list(L.block_offset), list(L.block_rows), list(L.block_feat)
        flat = o.reshape(N, D)

        def views(lo, cap, f, lens=None, dtype=None):
            block = flat[:, lo:lo + cap * f].reshape(N, cap, f)
            if dtype is not None:
                block = block.astype(dtype)
            if lens is None:
                return np.fromiter(iter(block), dtype=object, count=N)
            res = np.empty(N, dtype=object)
            for n in np.unique(lens):
                pos = np.nonzero(lens == n)[0]
                res[pos] = np.fromiter(iter(block[pos, :int(n)]), dtype=object, count=len(pos))
            return res

        def lists(*cols):
            return np.fromiter(map(list, zip(*cols)), dtype=object, count=N)
        out = np.empty((N, 3), dtype=object)
        ones = np.fromiter(itertools.repeat((1, 1, 1), N), dtype=object, count=N)
        if self.env_type == DynEnvType.ROBO_CUP and self.observationType == ObservationType.PARTIAL:
            tail = off[6]
            n = flat[:, tail:tail + 6].astype(np.int64)
            v = [views(off[k], rows[k], feat[k], n[:, k]) for k in range(6)]
            seen = np.fromiter(iter(flat[:, tail + 8:tail + 8 + (A - 1)].astype("uint8")), dtype=object, count=N)
            out[:, 0] = lists(v[0], v[1])
            out[:, 1] = lists(v[2], v[3], v[4], v[5])
            out[:, 2] = np.fromiter(zip(flat[:, tail + 6].astype(np.int64).tolist(), seen, (flat[:, tail + 7] != 0).tolist()),
                                    dtype=object, count=N)
        elif self.env_type == DynEnvType.ROBO_CUP:
            out[:, 0] = lists(views(0, 1, 4), views(12, A - 1, 6))
            out[:, 1] = lists(views(4, 1, 8))
            out[:, 2] = ones
        elif self.observationType == ObservationType.PARTIAL:
            n = flat[:, D - 4:].astype(np.int64)
            out[:, 0] = lists(views(off[1], rows[1], 7, n[:, 0]), views(off[2], rows[2], 6, n[:, 1]), views(off[3], rows[3], 2, n[:, 2]))
            out[:, 1] = lists(views(0, 1, 9), views(off[4], rows[4], 4, n[:, 3]))
            out[:, 2] = ones
        else:
            c = np.asarray(counts).astype(np.int64)
            n_obst, n_ped = np.repeat(c[:, 0], T * A), np.repeat(c[:, 1], T * A)
            out[:, 0] = lists(views(off[1], rows[1], 7), views(off[2], rows[2], 4, n_obst), views(off[3], rows[3], 2, n_ped))
            out[:, 1] = lists(views(off[0], 1, 9), views(off[4], rows[4], 5))
            out[:, 2] = ones
        return out.reshape(E, T, A, 3)

    def _compat_element(self, o, counts, e, t, a):
        """One element [e, t, a] of the reference's observation array: [movable-object arrays, static / self arrays, seen info]
        (the same views / values `_compat_obs` assembles for the whole batch)."""
        L = self.layout
        off, rows, feat = list(L.block_offset), list(L.block_rows), list(L.block_feat)
        A, D = self.n_agents, self.obs_dim
        r = o[e, t, a]
        if self.env_type == DynEnvType.ROBO_CUP and self.observationType == ObservationType.PARTIAL:
            tail = off[6]
            n = [int(x) for x in r[tail:tail + 6]]
            lists = [r[off[k]:off[k] + n[k] * feat[k]].reshape(n[k], feat[k]) for k in range(6)]
            return [[lists[0], lists[1]], [lists[2], lists[3], lists[4], lists[5]],
                    (int(r[tail + 6]), r[tail + 8:tail + 8 + (A - 1)].astype("uint8"), bool(r[tail + 7]))]
        if self.env_type == DynEnvType.ROBO_CUP:
            return [[r[0:4].reshape(1, 4), r[12:12 + (A - 1) * 6].reshape(A - 1, 6)], [r[4:12].reshape(1, 8), ], (1, 1, 1)]
        if self.observationType == ObservationType.PARTIAL:
            nc, no, npd, nl = [int(x) for x in r[D - 4:]]
            return [[r[off[1]:off[1] + rows[1] * 7].reshape(rows[1], 7)[:nc], r[off[2]:off[2] + rows[2] * 6].reshape(rows[2], 6)[:no],
                     r[off[3]:off[3] + rows[3] * 2].reshape(rows[3], 2)[:npd]],
                    [r[0:9].reshape(1, 9), r[off[4]:off[4] + rows[4] * 4].reshape(rows[4], 4)[:nl]], (1, 1, 1)]
        n_obst, n_ped = int(counts[e, 0]), int(counts[e, 1])
        return [[r[off[1]:off[1] + rows[1] * 7].reshape(rows[1], 7), r[off[2]:off[2] + rows[2] * 4].reshape(rows[2], 4)[:n_obst],
                 r[off[3]:off[3] + rows[3] * 2].reshape(rows[3], 2)[:n_ped]],
                [r[off[0]:off[0] + 9].reshape(1, 9), r[off[4]:off[4] + rows[4] * 5].reshape(rows[4], 5)], (1, 1, 1)]

    def _host_counts(self):
        """(n_obstacles, n_pedestrians) per environment on the host: they change at a reset only, so one copy per episode."""
        if self._counts_np is None:
            self._counts_np = self.counts().cpu().numpy()
        return self._counts_np

    def _wrap_obs(self, obs_np, counts):
        return self._compat_obs_bulk(obs_np, counts) if self.eager_compat else LazyObsArray(self, obs_np, counts)

    def _full_states(self, full_np, counts, e, glob_np=None):
        """info['Full State'] / info['Recon States'] (DrivingEnvironment.py:306-307, RoboCupEnvironment.py:511-512) of env e from
        the noise-free Full rows [E, A, full_obs_dim] of the state after the step (whatever the observation type) and, for
        RoboCup, the rows of dynenv_global_state [E, 6 A + 3]."""
        A = self.n_agents
        if self.env_type == DynEnvType.ROBO_CUP:
            recon = []  # getFullState(agent) = [ball, self, robots] (:1164-1188)
            for a in range(A):
                r = full_np[e, a]
                recon.append([r[0:4].reshape(1, 4).copy(), r[4:12].reshape(1, 8).copy(),
                              r[12:12 + (A - 1) * 6].reshape(A - 1, 6).copy()])
            g = glob_np[e]  # getFullState(None) = [robots [A, 6], ball [3]] (:1149-1161)
            return [g[:6 * A].reshape(A, 6).copy(), g[6 * A:6 * A + 3].copy()], recon
        n_obst, n_ped = int(counts[e, 0]), int(counts[e, 1])
        o1 = 9 + (A - 1) * 7
        o2, o3 = o1 + 80, o1 + 120
        recon = []
        for a in range(A):
            r = full_np[e, a]
            recon.append([r[0:9].reshape(1, 9).copy(), r[9:o1].reshape(A - 1, 7).copy(),
                          r[o1:o1 + n_obst * 4].reshape(n_obst, 4).copy(),
                          r[o2:o2 + n_ped * 2].reshape(n_ped, 2).copy(),
                          r[o3:o3 + 40].reshape(8, 5).copy()])
        # complete state: every car row = own [x,y,cos,sin,w,h] + finished
        cars = np.stack([np.concatenate([recon[a][0][0, :6], recon[a][0][0, 8:9]]) for a in range(A)]).astype(np.float32)
        full = [cars, recon[0][2], recon[0][3], recon[0][4]]
        return full, recon

    def reset(self):
        self.reset_flat()
        return self._wrap_obs(self.obs.clone() if not self.eager_compat else self.obs.cpu().numpy(), self._host_counts())

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        actions, self._pending = self._pending, None
        counts = self._host_counts()
        self.step_flat(actions if self.allow_head_turn else np.asarray(actions), auto_reset=False, validate=True)
        # the state behind info['Full State'] / info['Recon States']: with Full observations it is the step's last snapshot; with
        # Partial ones a noise-free Full row is emitted now (one short launch) and copied to the host only if somebody asks
        full_dev = self.full_state_obs() if self.observationType != ObservationType.FULL else None
        glob_dev = self.global_state() if self.env_type == DynEnvType.ROBO_CUP else None  # 252 B per environment, one short launch
        rewards = self.rewards.cpu().numpy().copy()
        if self.observationType == ObservationType.PARTIAL and self.error_flags() & 8:
            # the reference's observation lists have no cap (DrivingEnvironment.py:816-890); the dense layout has, and rows beyond it were
            # dropped: not the reference's observation any more - never silently (include/dynenv.h, error bit 3)
            raise _capi.DynEnvError("Partial observation: a list had more rows than the dense layout's capacity; rows were dropped (error bit 3)")
        if self.env_type == DynEnvType.ROBO_CUP and self.error_flags() & 32:
            raise _capi.DynEnvError("RoboCup: a velocity or joint impulse left the finite range (error bit 5): the state is not the reference's any more")
        if self.env_type == DynEnvType.ROBO_CUP and self.error_flags() & 16:
            # two capsule cores exactly collinear / exactly touching: the sign of the contact normal is a convention there
            # (Robot.py:38-52; include/dynenv.h, error bit 4) - possibly not pymunk's trajectory from here on, never silently
            raise _capi.DynEnvError("RoboCup: the cores of two feet are exactly collinear / touching: the contact normal's sign is a convention "
                                    "(error bit 4); reset() or set_state() clears it")
        done = bool(self.last_done)
        dones = np.full((self.num_envs,), done, dtype=bool)
        # The step's observations stay in HBM (a snapshot: self.obs is rewritten by the next step) until somebody looks at them:
        # ONE device->host copy then serves obs and infos alike.  (A fresh 38 MB host buffer per step costs ~20 ms of page
        # faults at 4096 envs - more than the step.)
        lazy_obs = LazyObsArray(self, self.obs.clone() if not self.eager_compat else self.obs.cpu().numpy(), counts)
        cache = {}

        def full_np():
            if "full" not in cache:
                cache["full"] = full_dev.cpu().numpy() if full_dev is not None else lazy_obs._dense[:, -1]
            return cache["full"]

        def glob_np():
            if glob_dev is None:
                return None
            if "glob" not in cache:
                cache["glob"] = glob_dev.cpu().numpy()
            return cache["glob"]
        stats = term = None
        if done:
            stats = [x.cpu().numpy() for x in self.episode_stats()]
            term = self._compat_obs(lazy_obs._dense, counts)

        def make_info(e):
            eager = {}
            if done:
                eager["episode_r"] = stats[0][e].copy()
                eager["episode_p_r"] = stats[1][e].copy()
                eager["episode_o_r"] = stats[2][e].copy() if self.env_type == DynEnvType.ROBO_CUP else [0, ] * self.n_agents
                eager["episode_g"] = [int(stats[3][e, 0]), int(stats[3][e, 1])]
                eager["terminal_observation"] = [list(term[e, t]) for t in range(self.n_time_steps)]
            return LazyInfo(lambda: self._full_states(full_np(), counts, e, glob_np()), eager)
        infos = LazyInfos(self.num_envs, make_info)
        if self.eager_compat:
            infos = tuple(infos)
        if done:
            self.terminal_obs = self.obs.clone()
            self.reset_flat()
            obs = self._wrap_obs(self.obs.clone() if not self.eager_compat else self.obs.cpu().numpy(), self._host_counts())
        else:
            obs = self._compat_obs_bulk(lazy_obs._dense, counts) if self.eager_compat else lazy_obs
        return obs, rewards, dones, infos

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
            o = self.full_state_obs().cpu().numpy()  # getFullState(agent): the true state, whatever the observation type
            if self.env_type == DynEnvType.ROBO_CUP:  # RoboCupEnvironment.py:434-435: self rows [:, 0:6]
                return [[o[e, a, 4:10].reshape(1, 6).copy() for a in range(self.n_agents)] for e in idx]
            # DrivingEnvironment.py:126-127: self rows [x, y, cos, sin] per agent
            return [[o[e, a, 0:4].reshape(1, 4).copy() for a in range(self.n_agents)] for e in idx]
        if method_name == "set_random_seed":
            return self.seed(*method_args)
        raise NotImplementedError(method_name)

    def render(self, *a, **k):
        return None

    def get_images(self, *a, **k):
        """base_vec_env.py:174-178: the rendered frames of the sub-environments.  Rendering (pygame) is outside the step path."""
        raise NotImplementedError("rendering is not part of the batched step path (SURVEY.md section 2: out of scope)")

    @property
    def unwrapped(self):  # base_vec_env.py:203-208
        return self

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
