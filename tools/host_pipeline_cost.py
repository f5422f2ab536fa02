"""Host-side cost of one sharded step of the pipelined protocol: the time the Python / HIP-runtime / RCCL enqueue path needs per
step when nothing makes the host wait (small batch, ring longer than the run).
Usage (GPU box): python tools/host_pipeline_cost.py"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from dynenv_amd import BatchedDynEnv, DynEnvType
from dynenv_amd.distributed import PackedSlab, StepGather, transport_layout
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
E, A, RING = 64, 10, 96  # a ring longer than the timed run: release() never blocks, the loop time is pure enqueue cost
probe = BatchedDynEnv(DynEnvType.DRIVE, 1, A, device=dev); T, D = probe.n_time_steps, probe.obs_dim; lay = transport_layout(probe); probe.close()
acts = torch.randint(0, 3, (E, A, 2), device=dev, dtype=torch.int32)


def run(name, make):
    slabs = [PackedSlab(torch, dev, E, T, A, D, **lay) for _ in range(RING)]
    gather = make(slabs)
    env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=1, device=dev, out_buffers=(slabs[0].obs, slabs[0].rewards, slabs[0].dones))
    env.reset_flat()
    k = 0
    def one():
        nonlocal k
        if gather is None:
            env.step_flat(acts, auto_reset=False)
        else:
            gather.release(k); sl = gather.slabs[k % RING]
            env.use_buffers(sl.obs, sl.rewards, sl.dones); env.step_flat(acts, auto_reset=False); gather.start(k)
        k += 1
    for _ in range(8): one()
    if gather is not None: gather.drain()
    torch.cuda.synchronize()
    k = 0
    t0 = time.perf_counter()
    for _ in range(80): one()
    t1 = time.perf_counter()
    if gather is not None: gather.drain()
    torch.cuda.synchronize()
    print("%-60s host %.1f us/step to enqueue   (%.1f us/step until the GPU has finished)" % (name, (t1 - t0) / 80 * 1e6, (time.perf_counter() - t0) / 80 * 1e6))
    env.close()


def one_stream(slabs):
    g = StepGather(torch, dist, slabs[0], more=slabs[1:]); g.expand = None; return g

run("step_flat alone", lambda slabs: None)
run("pipelined transport, one side stream", one_stream)
run("pipelined transport, two side streams", lambda slabs: StepGather(torch, dist, slabs[0], more=slabs[1:]))
dist.destroy_process_group()
