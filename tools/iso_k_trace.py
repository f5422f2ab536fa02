"""How many environments the SIMD-isolation list holds over an episode (Driving Full, 4096 envs): K per step from dynenv_debug_counters
(isolated_next).  Usage (GPU box): python3 tools/iso_k_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynenv_amd import BatchedDynEnv, DynEnvType
env = BatchedDynEnv(DynEnvType.DRIVE, 4096, 10, seed=42)
env.reset_flat()
g = torch.Generator(device="cuda").manual_seed(4321)
pool = [torch.randint(0, 3, (4096, 10, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(16)]
ks = []
for i in range(600):
    env.step_flat(pool[i & 15], auto_reset=False)
    if i % 10 == 9:
        ks.append(env.debug_counters().get("isolated_next", -1))
print("K (environments isolated in the next step), every 10th step:", ks)
print("max", max(ks), "mean %.1f" % (sum(ks) / len(ks)))
