#!/usr/bin/env python3
"""Where does a Driving Partial agent pass (driving_partial.hip pv_env) spend its time?  -DDRV_PROFILE build, one whole episode of
configs[3] at 4096 environments; cycles summed over every pass of the run (step launch + deferred launch), divided by the passes.
Each s_memtime stamp costs 100-200 cycles of its own.  Usage (GPU box): python3 tools/vision_profile.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.environ.get("PROFILE_LIB", os.path.join(ROOT, "dynenv_amd", "libdynenv_hip_prof.so"))
os.environ["DYNENV_HIP_LIB"] = PROF
from dynenv_amd import build as _b  # noqa: E402
if not os.path.exists(PROF) or any(os.path.getmtime(d) > os.path.getmtime(PROF) for d in _b.DEPS if os.path.exists(d)):
    _b.build(out=PROF, defines=("DRV_PROFILE",))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
env = BatchedDynEnv(DynEnvType.DRIVE, 4096, 10, seed=42, observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3)
env.reset_flat()
g = torch.Generator(device="cuda").manual_seed(1)
for s in range(STEPS):
    env.step_flat(torch.randint(0, 3, (4096, 10, 2), dtype=torch.int32, device="cuda", generator=g), auto_reset=False)
env.debug_counters()
d = np.loadtxt("gpurun_out/dbgv.txt").sum(0)
n = max(d[8], 1.0)
names = ["detection (transform, sincos)", "lane rows", "blockers (corner angles: pooled atan2)", "buildings + list positions", "pedestrian x blocker pairs",
         "noise (2 Philox blocks, atan2, sincos)", "random false positives", "assembly + row out"]
print("Driving Partial agent pass, %d passes (%d steps x 4096 environments x 10 agents + reset): cycles per pass, share" % (n, STEPS))
tot = d[:8].sum()
for k, nm in enumerate(names):
    print("  %-42s %8.0f  %5.1f %%" % (nm, d[k] / n, 100 * d[k] / tot))
print("  %-42s %8.0f" % ("whole pass", tot / n))
