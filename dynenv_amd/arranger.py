"""GPU-side ragged -> padded arranger: the host mirror of the reference's `InOutArranger`
(DynEnv/models/models.py:208-274) over the dense observation tensor `BatchedDynEnv.step_flat()` returns.

The reference's input layer does, per forward pass, in triple-nested Python over object arrays:
    inputs, counts = arranger.rearrange_inputs(x)                  # x = obs[..., g], ragged lists
    outs = [block(torch.tensor(obj)) for block, obj in zip(blocks, inputs)]
    outs, masks = arranger.rearrange_outputs(outs, counts, device)
Here the same three calls exist with the same results, but `x` is the dense device tensor and all index work runs in the
HIP kernels of `csrc/arranger_kernels.hip` through the C ABI (`dynenv_arrange_*`, include/dynenv.h).  No CPU fallback.
"""
import ctypes as C
import os

from . import _capi
from .enums import DynEnvType, ObservationType


def groups_for(env):
    """The object-type groups of an environment's observation, as the reference splits them into obs[..., 0] (movable
    objects) and obs[..., 1] (self / static rows): DrivingEnvironment.py:121-124, :977; RoboCupEnvironment.py:440-443.
    Returns {"movable": [ArrType...], "static": [ArrType...]}."""
    L = env.layout
    off, rows, feat = list(L.block_offset), list(L.block_rows), list(L.block_feat)
    A, D = env.n_agents, env.obs_dim

    def ty(offset, f, cap, mode=_capi.ARR_COUNT_CONST, value=0, index=0, stride=0):
        return _capi.ArrType(offset, f, cap, mode, value, index, stride, 0)

    if env.env_type == DynEnvType.ROBO_CUP and env.observationType == ObservationType.PARTIAL:
        t = off[6]  # tail: list lengths of balls, robots, goals, crosses, line crosses, lines
        mk = lambda k: ty(off[k], feat[k], rows[k], _capi.ARR_COUNT_ROW, index=t + k)
        return {"movable": [mk(0), mk(1)], "static": [mk(2), mk(3), mk(4), mk(5)]}
    if env.env_type == DynEnvType.ROBO_CUP:  # row = [ball 4 | self 8 | robots (A-1) x 6], see vec_env._compat_obs
        return {"movable": [ty(0, 4, 1, value=1), ty(12, 6, A - 1, value=A - 1)], "static": [ty(4, 8, 1, value=1)]}
    if env.observationType == ObservationType.PARTIAL:  # list lengths live in the last four floats of the row
        return {"movable": [ty(off[1], 7, rows[1], _capi.ARR_COUNT_ROW, index=D - 4),
                            ty(off[2], 6, rows[2], _capi.ARR_COUNT_ROW, index=D - 3),
                            ty(off[3], 2, rows[3], _capi.ARR_COUNT_ROW, index=D - 2)],
                "static": [ty(0, 9, 1, value=1), ty(off[4], 4, rows[4], _capi.ARR_COUNT_ROW, index=D - 1)]}
    return {"movable": [ty(off[1], feat[1], rows[1], value=rows[1]),                       # the other A-1 cars
                        ty(off[2], feat[2], rows[2], _capi.ARR_COUNT_ENV, index=0, stride=2),  # obstacles of the env
                        ty(off[3], feat[3], rows[3], _capi.ARR_COUNT_ENV, index=1, stride=2)],  # pedestrians
            "static": [ty(off[0], 9, 1, value=1), ty(off[4], feat[4], rows[4], value=rows[4])]}


class GpuInOutArranger(object):
    """Same role and call sequence as the reference's InOutArranger(nObjectTypes, nPlayers, nTime) (models.py:208-216);
    `nPlayers` is E*A as in InputLayer (:283).  `types` describes where each object type lives in an observation row."""

    def __init__(self, types, nEnvs, nAgents, nTime, obs_dim, device=None):
        import torch
        if not torch.cuda.is_available():
            raise _capi.DynEnvError("dynenv_amd needs an MI355X (HIP device); there is no CPU fallback")
        self._torch = torch
        self._lib = _capi.load()
        self.nObjectTypes = len(types)
        if not 1 <= self.nObjectTypes <= _capi.ARR_MAX_TYPES:
            raise ValueError("1..%d object types per group" % _capi.ARR_MAX_TYPES)
        self.types = (_capi.ArrType * self.nObjectTypes)(*types)
        self.E, self.A, self.nTime, self.D = int(nEnvs), int(nAgents), int(nTime), int(obs_dim)
        self.nPlayers = self.E * self.A
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        TP = self.nTime * self.nPlayers
        i32 = dict(dtype=torch.int32, device=self.device)
        self._counts = torch.empty((self.nObjectTypes, self.nTime, self.nPlayers), **i32)
        self._obj_counts = torch.empty((self.nTime, self.nPlayers), **i32)
        self._base = torch.empty((self.nObjectTypes, self.nTime, self.nPlayers), **i32)
        n = int(self._lib.dynenv_arrange_scratch_ints(self.E, self.nTime, self.A, self.nObjectTypes))
        self._scratch = torch.empty((n + 2,), **i32)
        self.feats = [int(t.feat) for t in types]

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def rearrange_inputs(self, obs, count_env=None):
        """obs: dense float32 device tensor [E, T, A, D]; count_env: int32 [E, 2] from BatchedDynEnv.counts() when a type
        counts per environment.  Returns (inputs, (counts, maxCount, objCounts, slots)) like models.py:219-250:
        inputs[i] float32 [N_i, feat_i] in (time, player, object) order; counts int32 [type, T, P]; objCounts [T, P]."""
        torch = self._torch
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous()
        assert tuple(obs.shape) == (self.E, self.nTime, self.A, self.D), (tuple(obs.shape), (self.E, self.nTime, self.A, self.D))
        plan = _capi.ArrPlan()
        _capi.check(self._lib.dynenv_arrange_plan(self._p(obs), self.E, self.nTime, self.A, self.D, self.types,
                                                  self.nObjectTypes, self._p(count_env), self._p(self._counts),
                                                  self._p(self._obj_counts), self._p(self._base), self._p(self._scratch),
                                                  C.byref(plan), self._stream()), "dynenv_arrange_plan")
        max_count = int(plan.max_count)
        inputs = [torch.empty((int(plan.total[i]), self.feats[i]), dtype=torch.float32, device=self.device)
                  for i in range(self.nObjectTypes)]
        slots = [torch.empty((int(plan.total[i]),), dtype=torch.int32, device=self.device) for i in range(self.nObjectTypes)]
        mask = torch.empty((self.nTime, self.nPlayers, max_count), dtype=torch.uint8, device=self.device)
        vp = C.c_void_p
        in_ptrs = (vp * self.nObjectTypes)(*[t.data_ptr() if t.numel() else None for t in inputs])
        sl_ptrs = (vp * self.nObjectTypes)(*[t.data_ptr() if t.numel() else None for t in slots])
        _capi.check(self._lib.dynenv_arrange_gather(self._p(obs), self.E, self.nTime, self.A, self.D, self.types,
                                                    self.nObjectTypes, self._p(self._counts), self._p(self._base), max_count,
                                                    in_ptrs, sl_ptrs, self._p(mask) if max_count else vp(0), self._stream()),
                    "dynenv_arrange_gather")
        return inputs, (self._counts, max_count, self._obj_counts, slots, mask)

    def rearrange_outputs(self, outs, countArr, device=None):
        """outs[i]: float32 [N_i, F] embeddings of inputs[i] (or None).  Returns (padded [T, maxCount, P, F],
        masks = list over time of bool [P, maxCount]) like models.py:252-274."""
        torch = self._torch
        _, max_count, _, slots, mask = countArr
        F = next(int(o.shape[1]) for o in outs if o is not None)
        # a type without any object has out = None in the reference (:262); its counts are all zero, so it shifts nothing
        ok = F % 4 == 0 and all(o is None or (o.is_cuda and o.dtype == torch.float32 and o.is_contiguous() and
                                              o.shape[1] == F and o.data_ptr() % 16 == 0) for o in outs)
        n_obj = sum(int(o.shape[0]) for o in outs if o is not None)
        # One pass over the padded tensor (5.4-5.7 TB/s on MI355X) unless it is almost all padding: below ~6 % occupancy a memset
        # (7 TB/s) + a scatter of the few rows moves less (driving Partial at 19 %: 0.150 ms in one pass against 0.196 ms).
        dense = n_obj >= float(os.environ.get('DYNENV_ARR_DENSE_MIN', '0.06')) * self.nTime * max_count * self.nPlayers
        if ok and dense:  # one pass over the padded tensor, padding zeros included
            padded = torch.empty((self.nTime, max_count, self.nPlayers, F), dtype=torch.float32, device=self.device)
            vp = C.c_void_p
            ptrs = (vp * self.nObjectTypes)(*[o.data_ptr() if (o is not None and o.shape[0]) else None for o in outs])
            _capi.check(self._lib.dynenv_arrange_pad(ptrs, self._p(self._counts), self._p(self._base), self.nObjectTypes,
                                                     self.nTime, self.nPlayers, max_count, F, self._p(padded),
                                                     self._stream()), "dynenv_arrange_pad")
        else:  # odd widths: zero fill + one scatter per type
            padded = torch.zeros((self.nTime, max_count, self.nPlayers, F), dtype=torch.float32, device=self.device)
            for i, out in enumerate(outs):
                if out is None or out.shape[0] == 0:
                    continue
                out = out.contiguous().float()
                _capi.check(self._lib.dynenv_arrange_scatter(self._p(out), self._p(slots[i]), int(out.shape[0]), F,
                                                             self._p(padded), self._stream()), "dynenv_arrange_scatter")
        masks = [mask[t].bool() for t in range(self.nTime)]
        return padded, masks
