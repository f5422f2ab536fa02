"""Host-side cost (microseconds per call, no device sync inside the timed region) of the calls one sharded step makes.
Usage (GPU box): python tools/host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from dynenv_amd import BatchedDynEnv, DynEnvType
from dynenv_amd.distributed import PackedSlab, StepGather, transport_layout
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
E, A = 4096, 10
probe = BatchedDynEnv(DynEnvType.DRIVE, 1, A, device=dev); T, D = probe.n_time_steps, probe.obs_dim; lay = transport_layout(probe); probe.close()
slabs = [PackedSlab(torch, dev, E, T, A, D, **lay) for _ in range(2)]
gather = StepGather(torch, dist, slabs[0], slab2=slabs[1])
env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=1, device=dev, out_buffers=(slabs[0].obs, slabs[0].rewards, slabs[0].dones))
env.reset_flat()
acts = torch.randint(0, 3, (E, A, 2), device=dev, dtype=torch.int32)
N = 150


def timeit(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        fn(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host %7.1f us/call   (+%.1f ms to drain the queue)" % (name, (t1 - t0) / N * 1e6, (t2 - t1) * 1e3))


timeit("env.step_flat", lambda k: env.step_flat(acts))
timeit("env.use_buffers", lambda k: env.use_buffers(slabs[k % 2].obs, slabs[k % 2].rewards, slabs[k % 2].dones))
timeit("slab.pack (ctypes launch)", lambda k: slabs[0].pack())
ev = torch.cuda.Event()
timeit("event.record + stream.wait_event", lambda k: (ev.record(), gather.comm.wait_event(ev)))
def ctx(k):
    with torch.cuda.stream(gather.comm):
        pass
timeit("with torch.cuda.stream(side)", ctx)
timeit("all_gather_into_tensor (sync op)", lambda k: dist.all_gather_into_tensor(gather.gbufs[0], slabs[0].buf))
def ag_async(k):
    w = dist.all_gather_into_tensor(gather.gbufs[0], slabs[0].buf, async_op=True); w.wait()
timeit("all_gather_into_tensor (async + wait)", ag_async)
timeit("gathered_views (unpack launch + views)", lambda k: slabs[0].gathered_views(gather.gbufs[0], 1, gather.dense[0]))
def full(k):
    gather.release(k); sl = gather.slabs[k % 2]; env.use_buffers(sl.obs, sl.rewards, sl.dones); env.step_flat(acts); gather.start(k)
timeit("whole pipelined step", full)
gather.drain()
dist.destroy_process_group()
