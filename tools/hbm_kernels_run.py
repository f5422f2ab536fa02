#!/usr/bin/env python3
"""The two genuinely HBM-bound kernels of the repository, run a few dozen times at the sizes bench.py / DESIGN.md quote - the program
tools/profile_round.sh wraps in rocprofv3 (kernel stats, FETCH_SIZE, WRITE_SIZE) for the workload "hbm":
  arr_pad_cols_kernel            GpuInOutArranger.rearrange_outputs on Driving Full observations of 4096 environments, F = 128 (bench.py arranger_leg)
  obs_unpack_peers_rows_kernel   the expansion of 8 ranks x 4096 environments of all-gathered, peer-compacted observations (tools/unpack_probe.py)
Prints their algorithmic bytes per launch (what profile_collect.py divides the measured traffic by)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from dynenv_amd import _capi  # noqa: E402

dev = torch.device("cuda", 0)
leg = bench.arranger_leg(torch, dev, 4096, 42, reps=30)
G, E, A, D = 8, 4096, 10, 232
P = A * 9 + (D - (9 + (A - 1) * 7))
lib = _capi.load()
packed = torch.randn((G, E, P), device=dev)
dense = torch.empty((G, E, A, D), device=dev)
vp = C.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
for _ in range(31):
    _capi.check(lib.dynenv_obs_unpack_peers_ranks(vp(packed.data_ptr()), E * P, G, E, A, D, vp(dense.data_ptr()), st), "unpack")
torch.cuda.synchronize()
print(json.dumps({"arr_pad_cols_kernel": {"alg_bytes": leg["roofline"]["alg_bytes"], "event_timed_ms": leg["rearrange_outputs_ms"]},
                  "obs_unpack_peers_rows_kernel": {"alg_bytes": packed.numel() * 4 + dense.numel() * 4}}))
