"""Device time of the transport's own kernels for the 8-rank case of BASELINE.json configs[4] (4096 environments per rank, Driving
Full): compaction of one rank's slab, expansion of the 8 gathered compacted slabs into the dense [8, E, T, A, D] tensor, beside
a memset and a copy of the same size.  Usage (GPU box): python tools/transport_kernel_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynenv_amd import BatchedDynEnv, DynEnvType
from dynenv_amd.distributed import PackedSlab, transport_layout
dev = torch.device("cuda", 0)
E, A, G = 4096, 10, 8
probe = BatchedDynEnv(DynEnvType.DRIVE, 1, A, device=dev); T, D = probe.n_time_steps, probe.obs_dim; lay = transport_layout(probe); probe.close()
sl = PackedSlab(torch, dev, E, T, A, D, **lay)
gb = torch.zeros((G * sl.nbytes,), dtype=torch.uint8, device=dev)
dense = torch.zeros((G, E, T, A, D), device=dev)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mb = dense.numel() * 4 / 1e6
tp = t(lambda: sl.pack()); tu = t(lambda: sl.gathered_views(gb, G, dense))
x = torch.empty_like(dense)
tm = t(lambda: x.zero_()); tc = t(lambda: x.copy_(dense))
print("compaction of one slab       %6.1f us" % tp)
print("expansion, 8 ranks           %6.1f us  (%.0f MB written: %.2f TB/s)" % (tu, mb, mb / tu))
print("memset of the same size      %6.1f us  (%.2f TB/s)" % (tm, mb / tm))
print("copy of the same size        %6.1f us" % tc)
