"""What does running the transport of step k beside the kernel of step k+1 cost?  GPU ms/step of several launch schemes
(no collective involved: isolates event / side-stream effects).  Usage (GPU box): python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynenv_amd import BatchedDynEnv, DynEnvType
from dynenv_amd.distributed import PackedSlab, transport_layout
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
E, A = 4096, 10
probe = BatchedDynEnv(DynEnvType.DRIVE, 1, A, device=dev); T, D = probe.n_time_steps, probe.obs_dim; lay = transport_layout(probe); probe.close()
NS = 8
slabs = [PackedSlab(torch, dev, E, T, A, D, **lay) for _ in range(NS)]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gb = [torch.zeros((G * s.nbytes,), dtype=torch.uint8, device=dev) for s in slabs]
dense = [torch.zeros((G, E, T, A, D), device=dev) for s in slabs]
env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=1, device=dev, out_buffers=(slabs[0].obs, slabs[0].rewards, slabs[0].dones))
g = torch.Generator(device=dev).manual_seed(5)
pool = [torch.randint(0, 3, (E, A, 2), generator=g, device=dev, dtype=torch.int32) for _ in range(64)]
side = torch.cuda.Stream(dev)
ready = [torch.cuda.Event() for _ in range(NS)]
done = [torch.cuda.Event() for _ in range(NS)]
N, W = 600, 600


def run(name, body):
    env.reset_flat()
    for k in range(W):
        env.step_flat(pool[k & 63])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(N):
        body(k)
    e1.record()
    torch.cuda.synchronize()
    print("%-70s %.4f ms/step" % (name, e0.elapsed_time(e1) / N), flush=True)


RCCL = os.environ.get("PROBE_RCCL", "0")
if RCCL != "0":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    G = 1


def transport(k, i):
    slabs[i].pack()
    if RCCL == "1":
        dist.all_gather_into_tensor(gb[i], slabs[i].buf)
    elif RCCL == "2":
        dist.all_gather_into_tensor(gb[i], slabs[i].buf, async_op=True).wait()
    else:
        for r in range(G):
            gb[i][r * slabs[i].nbytes:(r + 1) * slabs[i].nbytes].copy_(slabs[i].buf, non_blocking=True)
    slabs[i].gathered_views(gb[i], G, dense[i])


def s1(k):
    env.step_flat(pool[k & 63])
def s2(k):
    env.step_flat(pool[k & 63]); ready[k & 1].record()
def s3(k):
    env.step_flat(pool[k & 63]); transport(k, 0)
def s4(k):
    i = k & 1
    torch.cuda.current_stream().wait_event(done[i])
    env.use_buffers(slabs[i].obs, slabs[i].rewards, slabs[i].dones)
    env.step_flat(pool[k & 63]); ready[i].record()
    with torch.cuda.stream(side):
        side.wait_event(ready[i]); transport(k, i); done[i].record(side)
def s5(k):
    i = k & 1
    env.use_buffers(slabs[i].obs, slabs[i].rewards, slabs[i].dones)
    env.step_flat(pool[k & 63]); ready[i].record()
    with torch.cuda.stream(side):
        side.wait_event(ready[i]); transport(k, i)
def s6(k):
    i = k & 1
    env.use_buffers(slabs[i].obs, slabs[i].rewards, slabs[i].dones)
    env.step_flat(pool[k & 63]); ready[i].record()
    with torch.cuda.stream(side):
        side.wait_event(ready[i]); slabs[i].pack()


def deep(depth):
    def f(k):
        i = k % depth
        torch.cuda.current_stream().wait_event(done[i])
        env.use_buffers(slabs[i].obs, slabs[i].rewards, slabs[i].dones)
        env.step_flat(pool[k & 63]); ready[i].record()
        with torch.cuda.stream(side):
            side.wait_event(ready[i]); transport(k, i); done[i].record(side)
    return f


def deep_host(depth):
    def f(k):
        i = k % depth
        done[i].synchronize()  # HOST waits (throttle); the launch stream gets no barrier packet
        env.use_buffers(slabs[i].obs, slabs[i].rewards, slabs[i].dones)
        env.step_flat(pool[k & 63]); ready[i].record()
        with torch.cuda.stream(side):
            side.wait_event(ready[i]); transport(k, i); done[i].record(side)
    return f


print("world size emulated for the expansion: %d, RCCL mode %s" % (G, RCCL))
run("kernel only", s1)
run("kernel + event record on the launch stream", s2)
run("kernel + pack + copy + expand on the launch stream", s3)
run("side-stream transport, launch stream waits for transport k-2", s4)
run("side-stream transport, launch stream never waits (unsafe, probe only)", s5)
run("side stream runs only the pack kernel, launch stream never waits", s6)
run("side-stream transport, ring of 3 slabs (launch stream waits for transport k-3)", deep(3))
run("side-stream transport, ring of 4 slabs", deep(4))
for d in (2, 3, 4, 8):
    run("side-stream transport, ring of %d slabs, host-side throttle" % d, deep_host(d))
if RCCL != "0":
    from dynenv_amd.distributed import StepGather
    sg = StepGather(torch, dist, slabs[0], more=slabs[1:4])
    def prod(k):
        sg.release(k); sl = sg.slabs[k % 4]; env.use_buffers(sl.obs, sl.rewards, sl.dones); env.step_flat(pool[k & 63]); sg.start(k)
    run("product StepGather, ring of 4", prod)
    sg.drain()
    t0 = time.perf_counter()
    for k in range(N):
        prod(k)
    sg.drain(); torch.cuda.synchronize()
    print("product StepGather wall clock incl. drain: %.4f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
    dist.barrier(); torch.cuda.synchronize()
    run("product StepGather, ring of 4, after a dist.barrier()", prod)
    sg.drain()
