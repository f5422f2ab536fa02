"""Rehearsal of the N = 8 transport pipeline on ONE GPU, in the window the driver's scaling runs time (steps 5..24 after a reset,
where a step kernel takes ~0.13 ms): the product's StepGather with buffers for 8 ranks, the real pack / unpack kernels (the
expansion writes the dense [8, E, T, A, D] tensor), and the collective replaced by the world-1 RCCL call plus a spin of the
duration an 8-rank all-gather of the compacted slabs is expected to take over xGMI (argument, default 120 us).
Compares one side stream (pack, collective, expansion in sequence) with two (the expansion of step k beside the collective of
step k + 1).  Usage (GPU box): python tools/transport_pipeline_probe.py [collective_us]"""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from dynenv_amd import BatchedDynEnv, DynEnvType
from dynenv_amd.distributed import PackedSlab, StepGather, transport_layout

coll_us = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
E, A, G, RING = 4096, 10, 8, 4
probe = BatchedDynEnv(DynEnvType.DRIVE, 1, A, device=dev); T, D = probe.n_time_steps, probe.obs_dim; lay = transport_layout(probe); probe.close()
# spin cycles per microsecond of torch.cuda._sleep (calibrated)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10000000 / (e0.elapsed_time(e1) * 1e3)


class FakeDist(object):
    """world-1 RCCL all-gather into the first rank's block + a spin standing in for the other seven ranks' traffic"""
    def get_world_size(self, group=None): return G
    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        dist.all_gather_into_tensor(out[:inp.numel()], inp)
        torch.cuda._sleep(int(coll_us * cyc_per_us))


def run(two_streams, steps=20, warmup=5, reps=5):
    slabs = [PackedSlab(torch, dev, E, T, A, D, **lay) for _ in range(RING)]
    gather = StepGather(torch, FakeDist(), slabs[0], more=slabs[1:])
    if not two_streams:
        gather.expand = None
    env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=1, device=dev, out_buffers=(slabs[0].obs, slabs[0].rewards, slabs[0].dones))
    g = torch.Generator(device=dev).manual_seed(5)
    pool = [torch.randint(0, 3, (E, A, 2), generator=g, device=dev, dtype=torch.int32) for _ in range(64)]
    best = 1e9
    for _ in range(reps):
        gather.drain(); env.use_buffers(slabs[0].obs, slabs[0].rewards, slabs[0].dones)
        env.reset_flat()
        k = 0
        def one():
            nonlocal k
            gather.release(k); sl = gather.slabs[k % RING]
            env.use_buffers(sl.obs, sl.rewards, sl.dones); env.step_flat(pool[k & 63], auto_reset=False); gather.start(k); k += 1
        for _ in range(warmup): one()
        gather.drain(); torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(steps): one()
        gather.drain(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    env.close()
    return best


print("collective stand-in %.0f us; kernel alone in this window ~0.13 ms/step" % coll_us)
print("one side stream  (pack -> collective -> expansion): %.4f ms/step" % run(False))
print("two side streams (expansion of k beside the collective of k+1): %.4f ms/step" % run(True))
dist.destroy_process_group()
