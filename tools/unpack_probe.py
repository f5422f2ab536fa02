#!/usr/bin/env python3
"""How fast does a rank expand the all-gathered, peer-compacted Driving Full observations of G ranks x E environments into the dense
tensor (dynenv_obs_unpack_peers_ranks: the HBM-bound end of the multi-GPU transport; 304 MB written for 8 x 4096)?  Beside a memset of
the same size.   Usage (GPU box):  python tools/unpack_probe.py [ranks] [envs]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynenv_amd import _capi  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
A, D = 10, 232
P = A * 9 + (D - (9 + (A - 1) * 7))
lib = _capi.load()
packed = torch.randn((G, E, P), device="cuda")
dense = torch.empty((G, E, A, D), device="cuda")
vp = C.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)


def run():
    _capi.check(lib.dynenv_obs_unpack_peers_ranks(vp(packed.data_ptr()), E * P, G, E, A, D, vp(dense.data_ptr()), st), "unpack")


for name, fn in (("expand", run), ("memset", lambda: dense.zero_())):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("%s: %d ranks x %d envs, %.0f MB written: %.1f us = %.2f TB/s" % (name, G, E, dense.numel() * 4 / 1e6, us, dense.numel() * 4 / us / 1e6))
# exactness of the expansion against the numpy definition of the format
from dynenv_amd.distributed import unpack_peers_np  # noqa: E402
run()
want = unpack_peers_np(packed[:1, :64].cpu().numpy().reshape(64, P), A, D).reshape(64, A, D)
assert (dense[0, :64].cpu().numpy() == want).all()
print("expansion exact")
