"""Multi-GPU sharding: environments are independent, so each rank (one process per GPU) owns a contiguous range of
global env ids and steps it locally; the only exchange is ONE all-gather per env step of a packed slab
[ obs f32 | rewards f64 | dones u8 ] (RCCL over xGMI through torch.distributed backend "nccl"; "gloo" in CPU tests).
Driving Full observations travel in an exact compacted form (9.3x fewer bytes for 10 agents) and the transport of step k
runs on a side stream beside the kernel of step k+1.

RNG streams are keyed by GLOBAL env id (dynenv_cfg.env_id_offset), so per-env results are invariant to the number of
shards.  If the consumer (policy) is itself data-parallel, pass gather=False and skip the collective entirely.
"""
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the only mode this pool's driver supports); effective if HIP has not started yet
import numpy as np


def shard_range(total_envs, rank, world_size):
    """Contiguous env ranges: rank g owns [g*E/G, (g+1)*E/G)."""
    if total_envs % world_size:
        raise ValueError("total_envs must be divisible by world_size")
    per = total_envs // world_size
    return rank * per, per


# ---- the compacted transport formats, stated as index maps in numpy.  The HIP kernels (csrc/dynenv_capi.hip dynenv_obs_pack*,
# dynenv_obs_unpack*) implement exactly these; PackedSlab uses the numpy form for CPU tensors (gloo tests), and
# tests/test_gpu_sharded.py holds the kernels to it.
PEER_SELF, PEER_COLS = 9, 7
PEER_COL_MAP = (0, 1, 2, 3, 4, 5, 8)  # car c's row in another agent's list = these columns of c's own self block
                                      # (x, y, cos, sin, w, h, finished: DrivingEnvironment.getFullState :703-721)


def pack_tail_np(obs, split):
    """[N, A, D] -> [N, A * split + (D - split)]: the A row prefixes, then the tail every row shares (taken from agent 0)."""
    n, a, d = obs.shape
    return np.concatenate([obs[:, :, :split].reshape(n, a * split), obs[:, 0, split:]], axis=1)


def unpack_tail_np(packed, a, d, split):
    n = packed.shape[0]
    out = np.empty((n, a, d), packed.dtype)
    out[:, :, :split] = packed[:, :a * split].reshape(n, a, split)
    out[:, :, split:] = packed[:, None, a * split:]
    return out


def pack_peers_np(obs):
    """Driving Full [N, A, D] -> [N, A * 9 + tail]: the A self blocks, then the obstacle / pedestrian / lane tail once."""
    n, a, d = obs.shape
    cars_end = PEER_SELF + (a - 1) * PEER_COLS
    return np.concatenate([obs[:, :, :PEER_SELF].reshape(n, a * PEER_SELF), obs[:, 0, cars_end:]], axis=1)


def unpack_peers_np(packed, a, d):
    """Row of agent i = [self_i | for c != i ascending: self_c[PEER_COL_MAP] | tail]"""
    n = packed.shape[0]
    cars_end = PEER_SELF + (a - 1) * PEER_COLS
    selfb = packed[:, :a * PEER_SELF].reshape(n, a, PEER_SELF)
    out = np.empty((n, a, d), packed.dtype)
    for i in range(a):
        out[:, i, :PEER_SELF] = selfb[:, i]
        for q, c in enumerate([c for c in range(a) if c != i]):
            out[:, i, PEER_SELF + q * PEER_COLS:PEER_SELF + (q + 1) * PEER_COLS] = selfb[:, c][:, list(PEER_COL_MAP)]
        out[:, i, cars_end:] = packed[:, a * PEER_SELF:]
    return out


class PackedSlab(object):
    """One flat byte buffer per rank holding obs|rewards|dones so a single collective moves a whole step.

    split=None, peers=False: the obs region is the dense tensor itself (the step kernel writes into it, nothing is copied).
    split=k (GPU only): every agent row of an (env, time) ends in the same D-k floats; the slab carries the A prefixes and the
    tail once (dynenv_obs_pack).
    peers=True (GPU only, Driving Full): a row's "other cars" block repeats columns of those cars' own self blocks, so the slab
    carries the A self blocks and the tail once (dynenv_obs_pack_peers: 250 instead of 2320 floats per env for A=10).
    With split / peers the step kernel writes a separate dense tensor (`obs`), pack() fills the slab and gathered_views()
    expands the gathered slabs of all ranks into a dense [G, E, T, A, D] tensor again, bit for bit."""

    PEER_SELF, PEER_COLS = PEER_SELF, PEER_COLS

    def __init__(self, torch, device, E, T, A, D, split=None, peers=False):
        self.torch = torch
        self.E, self.T, self.A, self.D, self.split, self.peers = E, T, A, D, split, bool(peers)
        if peers:
            self.row = A * self.PEER_SELF + (D - self.PEER_SELF - (A - 1) * self.PEER_COLS)
        else:
            self.row = A * D if split is None else A * split + (D - split)  # floats per (env, time)
        self.packed = peers or split is not None
        self.obs_bytes = E * T * self.row * 4
        self.rew_off = (self.obs_bytes + 255) // 256 * 256
        self.rew_bytes = E * A * 8
        self.done_off = (self.rew_off + self.rew_bytes + 255) // 256 * 256
        self.nbytes = (self.done_off + E + 255) // 256 * 256
        self.buf = torch.zeros((self.nbytes,), dtype=torch.uint8, device=device)
        obs, self.rewards, self.dones = self.views(self.buf)
        self._lib = None
        if not self.packed:
            self.obs = obs
        else:
            self._packed = obs
            self.obs = torch.zeros((E, T, A, D), dtype=torch.float32, device=device)

    def views(self, buf):
        t = self.torch
        obs = buf[:self.obs_bytes].view(t.float32)
        obs = obs.view(self.E, self.T, self.row) if self.packed else obs.view(self.E, self.T, self.A, self.D)
        rew = buf[self.rew_off:self.rew_off + self.rew_bytes].view(t.float64).view(self.E, self.A)
        dones = buf[self.done_off:self.done_off + self.E]
        return obs, rew, dones

    def _capi(self):
        if self._lib is None:
            from . import _capi
            self._lib = (_capi, _capi.load())
        return self._lib

    def pack(self):
        """dense self.obs -> the slab's compacted obs region (a no-op for the dense layout); on the current stream"""
        if not self.packed:
            return
        if self.buf.device.type == "cpu":  # the index maps in numpy (gloo tests; a CPU tensor never meets the HIP library)
            o = self.obs.numpy().reshape(self.E * self.T, self.A, self.D)
            p = pack_peers_np(o) if self.peers else pack_tail_np(o, self.split)
            self._packed.copy_(self.torch.from_numpy(p).view(self.E, self.T, self.row))
            return
        import ctypes as C
        capi, lib = self._capi()
        st = C.c_void_p(self.torch.cuda.current_stream(self.buf.device).cuda_stream)
        src, dst = C.c_void_p(self.obs.data_ptr()), C.c_void_p(self._packed.data_ptr())
        if self.peers:
            capi.check(lib.dynenv_obs_pack_peers(src, self.E * self.T, self.A, self.D, dst, st), "dynenv_obs_pack_peers")
        else:
            capi.check(lib.dynenv_obs_pack(src, self.E * self.T, self.A, self.D, self.split, dst, st), "dynenv_obs_pack")

    def gathered_views(self, gbuf, world_size, dense_out=None):
        """[G, E_loc, ...] views into the gathered buffer (zero-copy; a compacted obs region is expanded into dense_out)."""
        t = self.torch
        g = gbuf.view(world_size, self.nbytes)
        rew = g[:, self.rew_off:self.rew_off + self.rew_bytes].view(t.float64).view(world_size, self.E, self.A)
        dones = g[:, self.done_off:self.done_off + self.E]
        if not self.packed:
            obs = g[:, :self.obs_bytes].view(t.float32).view(world_size, self.E, self.T, self.A, self.D)
            return obs, rew, dones
        if gbuf.device.type == "cpu":
            p = g[:, :self.obs_bytes].contiguous().view(t.float32).numpy().reshape(world_size * self.E * self.T, self.row)
            o = unpack_peers_np(p, self.A, self.D) if self.peers else unpack_tail_np(p, self.A, self.D, self.split)
            dense_out.copy_(t.from_numpy(o).view(world_size, self.E, self.T, self.A, self.D))
            return dense_out, rew, dones
        import ctypes as C
        capi, lib = self._capi()
        st = C.c_void_p(t.cuda.current_stream(self.buf.device).cuda_stream)
        src, dst = C.c_void_p(g.data_ptr()), C.c_void_p(dense_out.data_ptr())
        # the compacted obs regions of the ranks sit `nbytes` apart in the gathered buffer: one launch expands them all
        if self.peers:
            capi.check(lib.dynenv_obs_unpack_peers_ranks(src, self.nbytes // 4, world_size, self.E * self.T, self.A, self.D, dst, st),
                       "dynenv_obs_unpack_peers_ranks")
        else:
            capi.check(lib.dynenv_obs_unpack_ranks(src, self.nbytes // 4, world_size, self.E * self.T, self.A, self.D, self.split,
                                                   dst, st), "dynenv_obs_unpack_ranks")
        return dense_out, rew, dones


class StepGather(object):
    """All-gather of the packed step outputs.  Works on any backend (nccl=RCCL on MI355X, gloo on CPU).

    Blocking use: `views = gather()` after the step wrote `slab`.
    Pipelined use (a ring of n >= 2 slabs): `h = gather.start(k)` after step k wrote `slabs[k % n]`; `h.wait()` returns the
    gathered views.  On a GPU the whole transport of step k - compaction, the collective, expansion into the dense global
    tensor - is queued on a side stream behind an event of the step kernel, so it runs beside the kernels of the next steps
    (which are latency bound and leave HBM idle).  The launch stream is never made to wait for the side stream when a slab is
    reused: a cross-queue wait costs ~35 us of idle GPU per step on this stack even when it is already satisfied
    (tools/overlap_probe.py).  Instead release(k) throttles the HOST until the transport that last read slabs[k % n] has
    finished, i.e. the host runs at most n steps ahead of the transports.  The views of step k stay valid until step k+n
    is started."""

    def __init__(self, torch, dist, slab, group=None, slab2=None, more=()):
        self.torch, self.dist, self.group = torch, dist, group
        self.slabs = [slab] + ([slab2] if slab2 is not None else []) + list(more)
        self.slab = slab
        self.world_size = dist.get_world_size(group)
        self.gbufs = [torch.zeros((self.world_size * s.nbytes,), dtype=torch.uint8, device=s.buf.device) for s in self.slabs]
        self.gbuf = self.gbufs[0]
        self._pending = [None] * len(self.slabs)
        self.dense = [torch.zeros((self.world_size, s.E, s.T, s.A, s.D), dtype=torch.float32, device=s.buf.device)
                      if s.packed else None for s in self.slabs]
        self.device = slab.buf.device
        # high priority: its own hardware queue (a default-priority stream created after RCCL's can end up sharing the launch
        # stream's queue, which serialises the two), and the small transport kernels are dispatched ahead of the big step grid
        self.comm = torch.cuda.Stream(self.device, priority=-1) if self.device.type == "cuda" else None
        # the expansion of step k (one kernel that writes the dense [G, E, T, A, D] tensor: 8x the bytes the collective moved at
        # 8 ranks) runs on a second side stream, so that it overlaps the collective of step k + 1 instead of delaying it: the
        # transport's throughput is max(collective, expansion) per step, not their sum
        self.expand = torch.cuda.Stream(self.device, priority=-1) if self.device.type == "cuda" and any(s.packed for s in self.slabs) else None
        if self.comm is not None:
            self._ready = [torch.cuda.Event() for _ in self.slabs]
            self._gathered = [torch.cuda.Event() for _ in self.slabs]
            self._done = [torch.cuda.Event() for _ in self.slabs]

    def __call__(self):
        self.slab.pack()
        self.dist.all_gather_into_tensor(self.gbuf, self.slab.buf, group=self.group)
        return self.slab.gathered_views(self.gbuf, self.world_size, self.dense[0])

    class _Handle(object):
        def __init__(self, owner, idx, work=None, views=None, done=None):
            self.owner, self.idx, self.work, self.views, self.done = owner, idx, work, views, done
            self._waited = False

        def wait(self):
            """The gathered views of this step, usable on the current stream."""
            o = self.owner
            if self.done is not None:  # GPU: everything is already queued on the side stream
                if not self._waited:
                    o.torch.cuda.current_stream(o.device).wait_event(self.done)
                    self._waited = True
                return self.views
            if self.work is not None:
                self.work.wait()
                self.work = None
                self.views = o.slabs[self.idx].gathered_views(o.gbufs[self.idx], o.world_size, o.dense[self.idx])
            return self.views

        def finish(self):
            """Host-side: block until the transport has completed (before its slab is rewritten)."""
            if self.done is not None:
                self.done.synchronize()
            else:
                self.wait()

    def start(self, k):
        """Begin the all-gather of slabs[k % n]; returns a handle whose wait() yields the gathered views."""
        i = k % len(self.slabs)
        sl = self.slabs[i]
        if self.comm is None:
            sl.pack()
            work = self.dist.all_gather_into_tensor(self.gbufs[i], sl.buf, group=self.group, async_op=True)
            h = StepGather._Handle(self, i, work=work)
        else:
            t = self.torch
            self._ready[i].record(t.cuda.current_stream(self.device))
            with t.cuda.stream(self.comm):
                self.comm.wait_event(self._ready[i])
                sl.pack()
                # a blocking-semantics collective is stream-ordered on the current (= side) stream; the host does not wait
                self.dist.all_gather_into_tensor(self.gbufs[i], sl.buf, group=self.group)
                if self.expand is None:
                    views = sl.gathered_views(self.gbufs[i], self.world_size, self.dense[i])
                    self._done[i].record(self.comm)
                else:
                    self._gathered[i].record(self.comm)
            if self.expand is not None:
                with t.cuda.stream(self.expand):
                    self.expand.wait_event(self._gathered[i])
                    views = sl.gathered_views(self.gbufs[i], self.world_size, self.dense[i])
                    self._done[i].record(self.expand)
            h = StepGather._Handle(self, i, views=views, done=self._done[i])
        self._pending[i] = h
        return h

    def release(self, k):
        """Before slabs[k % n] is overwritten: the transport that still reads it must be over (no-op if none is pending)."""
        i = k % len(self.slabs)
        if self._pending[i] is not None:
            self._pending[i].finish()
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


def transport_layout(env):
    """PackedSlab keyword arguments of the most compact exact transport format of this environment's observation."""
    from .enums import DynEnvType, ObservationType
    if env.env_type == DynEnvType.DRIVE and env.observationType == ObservationType.FULL:
        assert int(env.layout.block_offset[2]) == PackedSlab.PEER_SELF + (env.n_agents - 1) * PackedSlab.PEER_COLS
        return dict(peers=True)
    return dict()


class ShardedDynEnv(object):
    """One process per GPU: local BatchedDynEnv over this rank's env range + optional end-of-step all-gather."""

    def __init__(self, env_type, total_envs, num_players, gather=True, seed=42, ring=4, **kw):
        import torch
        import torch.distributed as dist
        from .vec_env import BatchedDynEnv
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        off, per = shard_range(total_envs, self.rank, self.world_size)
        device = kw.pop("device", "cuda:%d" % torch.cuda.current_device())
        probe = BatchedDynEnv(env_type, 1, num_players, seed=seed, device=device, **kw)
        T, A, D = probe.n_time_steps, probe.n_agents, probe.obs_dim
        layout = transport_layout(probe) if gather and torch.device(device).type == "cuda" else {}
        probe.close()
        self.slab = PackedSlab(torch, torch.device(device), per, T, A, D, **layout)
        self.ring = max(2, int(ring)) if gather else 1
        more = [PackedSlab(torch, torch.device(device), per, T, A, D, **layout) for _ in range(self.ring - 1)]
        self.env = BatchedDynEnv(env_type, per, num_players, seed=seed, device=device, env_id_offset=off,
                                 out_buffers=(self.slab.obs, self.slab.rewards, self.slab.dones), **kw)
        self.gather = StepGather(torch, dist, self.slab, more=more) if gather else None
        self._k = 0

    def reset(self):
        if self.gather is not None:  # back to slab 0, with no gather still reading it
            self.gather.drain()
            self.env.use_buffers(self.slab.obs, self.slab.rewards, self.slab.dones)
            self._k = 1  # the next step writes slab 1; the views returned here (slab 0) stay valid for ring - 1 steps
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
        sl = self.gather.slabs[k % self.ring]
        self.env.use_buffers(sl.obs, sl.rewards, sl.dones)
        self.env.step_flat(local_actions)
        h = self.gather.start(k)
        return h.wait() if wait else h
