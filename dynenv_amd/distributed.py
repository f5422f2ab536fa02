"""Multi-GPU sharding: environments are independent, so each rank (one process per GPU) owns a contiguous range of
global env ids and steps it locally; the only exchange is ONE all-gather per env step of a packed slab
[ obs f32 | rewards f64 | dones u8 ] (RCCL over xGMI through torch.distributed backend "nccl"; "gloo" in CPU tests).

RNG streams are keyed by GLOBAL env id (dynenv_cfg.env_id_offset), so per-env results are invariant to the number of
shards.  If the consumer (policy) is itself data-parallel, pass gather=False and skip the collective entirely.
"""
import numpy as np


def shard_range(total_envs, rank, world_size):
    """Contiguous env ranges: rank g owns [g*E/G, (g+1)*E/G)."""
    if total_envs % world_size:
        raise ValueError("total_envs must be divisible by world_size")
    per = total_envs // world_size
    return rank * per, per


class PackedSlab(object):
    """One flat byte buffer per rank holding obs|rewards|dones so a single collective moves a whole step.

    split=None: the obs region is the dense tensor itself (the step kernel writes into it, nothing is copied).
    split=k (GPU only): every agent row of an (env, time) ends in the same D-k floats (Driving Full: obstacles,
    pedestrians, lanes); the slab then carries the de-duplicated form (dynenv_obs_pack, 2.6x fewer bytes for A=10) -
    the step kernel writes a separate dense tensor, pack() fills the slab, unpack() restores dense views after the gather."""

    def __init__(self, torch, device, E, T, A, D, split=None):
        self.torch = torch
        self.E, self.T, self.A, self.D, self.split = E, T, A, D, split
        self.row = A * D if split is None else A * split + (D - split)  # floats per (env, time)
        self.obs_bytes = E * T * self.row * 4
        self.rew_off = (self.obs_bytes + 255) // 256 * 256
        self.rew_bytes = E * A * 8
        self.done_off = (self.rew_off + self.rew_bytes + 255) // 256 * 256
        self.nbytes = (self.done_off + E + 255) // 256 * 256
        self.buf = torch.zeros((self.nbytes,), dtype=torch.uint8, device=device)
        obs, self.rewards, self.dones = self.views(self.buf)
        if split is None:
            self.obs = obs
        else:
            self._packed = obs
            self.obs = torch.zeros((E, T, A, D), dtype=torch.float32, device=device)
            self._lib = None

    def views(self, buf):
        t = self.torch
        obs = buf[:self.obs_bytes].view(t.float32)
        obs = obs.view(self.E, self.T, self.A, self.D) if self.split is None else obs.view(self.E, self.T, self.row)
        rew = buf[self.rew_off:self.rew_off + self.rew_bytes].view(t.float64).view(self.E, self.A)
        dones = buf[self.done_off:self.done_off + self.E]
        return obs, rew, dones

    def _capi(self):
        if self._lib is None:
            from . import _capi
            self._lib = (_capi, _capi.load())
        return self._lib

    def pack(self):
        """dense self.obs -> the slab's de-duplicated obs region (a no-op without split); on the current stream"""
        if self.split is None:
            return
        import ctypes as C
        capi, lib = self._capi()
        st = C.c_void_p(self.torch.cuda.current_stream(self.buf.device).cuda_stream)
        capi.check(lib.dynenv_obs_pack(C.c_void_p(self.obs.data_ptr()), self.E * self.T, self.A, self.D, self.split,
                                       C.c_void_p(self._packed.data_ptr()), st), "dynenv_obs_pack")

    def gathered_views(self, gbuf, world_size, dense_out=None):
        """[G, E_loc, ...] views into the gathered buffer (zero-copy; with split the obs are unpacked into dense_out)."""
        t = self.torch
        g = gbuf.view(world_size, self.nbytes)
        rew = g[:, self.rew_off:self.rew_off + self.rew_bytes].view(t.float64).view(world_size, self.E, self.A)
        dones = g[:, self.done_off:self.done_off + self.E]
        if self.split is None:
            obs = g[:, :self.obs_bytes].view(t.float32).view(world_size, self.E, self.T, self.A, self.D)
            return obs, rew, dones
        import ctypes as C
        capi, lib = self._capi()
        st = C.c_void_p(t.cuda.current_stream(self.buf.device).cuda_stream)
        # the packed obs regions of the ranks sit `nbytes` apart in the gathered buffer: one launch unpacks them all
        capi.check(lib.dynenv_obs_unpack_ranks(C.c_void_p(g.data_ptr()), self.nbytes // 4, world_size, self.E * self.T, self.A,
                                               self.D, self.split, C.c_void_p(dense_out.data_ptr()), st), "dynenv_obs_unpack_ranks")
        return dense_out, rew, dones


class StepGather(object):
    """All-gather of the packed step outputs.  Works on any backend (nccl=RCCL on MI355X, gloo on CPU).

    Blocking use: `views = gather()` after the step wrote `slab`.
    Pipelined use (two slabs): `h = gather.start(k)` after step k wrote `slabs[k % 2]`; the collective runs on the
    process group's own stream while the kernel of step k+1 writes the other slab; `h.wait()` makes the current stream wait
    for it and returns the gathered views.  A slab is reused only after the gather that read it has been waited for."""

    def __init__(self, torch, dist, slab, group=None, slab2=None):
        self.torch, self.dist, self.group = torch, dist, group
        self.slabs = [slab] + ([slab2] if slab2 is not None else [])
        self.slab = slab
        self.world_size = dist.get_world_size(group)
        self.gbufs = [torch.zeros((self.world_size * s.nbytes,), dtype=torch.uint8, device=s.buf.device) for s in self.slabs]
        self.gbuf = self.gbufs[0]
        self._pending = [None] * len(self.slabs)
        self.dense = [torch.zeros((self.world_size, s.E, s.T, s.A, s.D), dtype=torch.float32, device=s.buf.device)
                      if s.split is not None else None for s in self.slabs]

    def __call__(self):
        self.slab.pack()
        self.dist.all_gather_into_tensor(self.gbuf, self.slab.buf, group=self.group)
        return self.slab.gathered_views(self.gbuf, self.world_size, self.dense[0])

    class _Handle(object):
        def __init__(self, owner, idx, work):
            self.owner, self.idx, self.work = owner, idx, work

        def wait(self):
            if self.work is not None:
                self.work.wait()
                self.work = None
            o = self.owner
            return o.slabs[self.idx].gathered_views(o.gbufs[self.idx], o.world_size, o.dense[self.idx])

    def start(self, k):
        """Begin the all-gather of slabs[k % n]; returns a handle whose wait() yields the gathered views."""
        i = k % len(self.slabs)
        self.slabs[i].pack()
        work = self.dist.all_gather_into_tensor(self.gbufs[i], self.slabs[i].buf, group=self.group, async_op=True)
        h = StepGather._Handle(self, i, work)
        self._pending[i] = h
        return h

    def release(self, k):
        """Before slabs[k % n] is overwritten: wait for the gather that still reads it (no-op if none is pending)."""
        i = k % len(self.slabs)
        if self._pending[i] is not None:
            self._pending[i].wait()
            self._pending[i] = None

    def drain(self):
        for i in range(len(self.slabs)):
            self.release(i)


def shared_tail_split(env):
    """Float offset from which every agent row of an (env, time) is identical, or None.  Driving Full: the obstacle,
    pedestrian and lane blocks (DrivingEnvironment.getFullState gives every agent the same lists)."""
    from .enums import DynEnvType, ObservationType
    if env.env_type == DynEnvType.DRIVE and env.observationType == ObservationType.FULL:
        return int(env.layout.block_offset[2])
    return None


class ShardedDynEnv(object):
    """One process per GPU: local BatchedDynEnv over this rank's env range + optional end-of-step all-gather."""

    def __init__(self, env_type, total_envs, num_players, gather=True, seed=42, **kw):
        import torch
        import torch.distributed as dist
        from .vec_env import BatchedDynEnv
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        off, per = shard_range(total_envs, self.rank, self.world_size)
        device = kw.pop("device", "cuda:%d" % torch.cuda.current_device())
        probe = BatchedDynEnv(env_type, 1, num_players, seed=seed, device=device, **kw)
        T, A, D = probe.n_time_steps, probe.n_agents, probe.obs_dim
        split = shared_tail_split(probe)
        probe.close()
        self.slab = PackedSlab(torch, torch.device(device), per, T, A, D, split=split if gather else None)
        self.slab2 = PackedSlab(torch, torch.device(device), per, T, A, D, split=split) if gather else None
        self.env = BatchedDynEnv(env_type, per, num_players, seed=seed, device=device, env_id_offset=off,
                                 out_buffers=(self.slab.obs, self.slab.rewards, self.slab.dones), **kw)
        self.gather = StepGather(torch, dist, self.slab, slab2=self.slab2) if gather else None
        self._k = 0

    def reset(self):
        if self.gather is not None:  # back to slab 0, with no gather still reading it
            self.gather.drain()
            self.env.use_buffers(self.slab.obs, self.slab.rewards, self.slab.dones)
            self._k = 1  # the next step writes slab 1 while nothing reads slab 0 asynchronously
        self.env.reset_flat()
        return self.gather() if self.gather else (self.env.obs, self.env.rewards, self.env.dones)

    def step(self, local_actions, wait=True):
        """wait=True: the reference's lock-step semantics (the global view of this step is returned).
        wait=False: returns a handle; the all-gather overlaps the next step's kernel (`handle.wait()` -> views)."""
        if self.gather is None:
            self.env.step_flat(local_actions)
            return self.env.obs, self.env.rewards, self.env.dones
        k = self._k
        self._k += 1
        self.gather.release(k)
        sl = self.gather.slabs[k % 2]
        self.env.use_buffers(sl.obs, sl.rewards, sl.dones)
        self.env.step_flat(local_actions)
        h = self.gather.start(k)
        return h.wait() if wait else h
