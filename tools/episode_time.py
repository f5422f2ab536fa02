#!/usr/bin/env python3
"""Whole-episode mean step time of one workload at 4096 envs for the library DYNENV_HIP_LIB selects (A/B of build variants).
Usage (GPU box): DYNENV_HIP_LIB=dynenv_amd/libdynenv_hip_x.so [ET_ENVS=256 ET_STEPS=100] python tools/episode_time.py driving [repeats]
(256 environments = one wave per CU: the time of a lone wave)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "driving"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
robocup, partial = w.startswith("robocup"), w.endswith("partial")
E, A = int(os.environ.get("ET_ENVS", "4096")), 10
kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3) if partial else {}
env = BatchedDynEnv(DynEnvType.ROBO_CUP if robocup else DynEnvType.DRIVE, E, 5 if robocup else 10, seed=int(os.environ.get("ET_SEED", "42")), **kw)
g = torch.Generator(device="cuda").manual_seed(4321)
if robocup:
    hi = torch.tensor([5, 3, 3, 7], device="cuda")
    pool = [(torch.rand((E, A, 4), generator=g, device="cuda") * hi).to(torch.int32) for _ in range(16)]
else:
    pool = [torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(16)]
steps = int(os.environ.get("ET_STEPS", "240" if robocup else "600"))
out = []
for r in range(rep + 1):
    env.reset_flat()
    torch.cuda.synchronize()
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    for i in range(steps):
        env.step_flat(pool[i & 15], auto_reset=False)
    k1.record()
    torch.cuda.synchronize()
    if r:
        out.append(k0.elapsed_time(k1) / steps)
print("%s %s: %s ms/step (episodes 2..%d), error flags %d, digest %.12g" % (
    os.environ.get("DYNENV_HIP_LIB", "default"), w, " ".join("%.4f" % x for x in out), rep + 1, env.error_flags(),
    float(env.rewards.sum().item())))
