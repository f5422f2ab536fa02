"""Per-environment cycle breakdown of one RoboCup step (50 substeps) from the -DDRV_PROFILE build.
Usage (GPU box):  python tools/robocup_profile.py [step]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.environ.get("PROFILE_LIB", os.path.join(ROOT, "dynenv_amd", "libdynenv_hip_prof.so"))
os.environ["DYNENV_HIP_LIB"] = PROF
from dynenv_amd import build as _b
if not os.path.exists(PROF) or any(os.path.getmtime(d) > os.path.getmtime(PROF) for d in _b.DEPS if os.path.exists(d)):
    _b.build(out=PROF, defines=("DRV_PROFILE",))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(ROOT)
import torch, numpy as np
from dynenv_amd import BatchedDynEnv, DynEnvType
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NE = int(os.environ.get("PROFILE_ENVS", "4096"))
env = BatchedDynEnv(DynEnvType.ROBO_CUP, NE, 5, seed=42)
env.reset_flat()
g = torch.Generator(device="cuda"); g.manual_seed(1)
hi = torch.tensor([5, 3, 3, 7], device="cuda")  # MultiDiscrete([5, 3, 3, 7]) RoboCupEnvironment.py:342
for s in range(STEP):
    a = (torch.rand((4096, 10, 4), device="cuda", generator=g) * hi).to(torch.int32)[:NE].contiguous()  # (the same actions per environment id whatever NE is)
    env.step_flat(a)
env.debug_counters()
d = np.loadtxt("gpurun_out/rcprof.txt")[:NE]
names = ["sequential-logic substeps", "common part (logic+position+broadphase+quiet joints)", "-", "contacts+prestep", "joint prestep", "velocity", "warm start", "solver", "post-solve", "touched", "levels", "TOTAL"]
print("per-env cycles of one step (50 substeps): mean / p99 / max")
for k, n in enumerate(names):
    c = d[:, k]
    print("  %-54s %10.0f %10.0f %10.0f" % (n, c.mean(), np.percentile(c, 99), c.max()))
top = np.argsort(-d[:, 11])[:12]
print("slowest environments: " + " | ".join(names))
for k in top: print("  ", " ".join("%8d" % v for v in d[k]))
q = np.loadtxt("gpurun_out/rcprof2.txt")[:NE]
print("contacts + prestep of the slowest environments, cycles per step: candidate list | narrowphase passes | slot record | callbacks, expiry | levels | prestep + bias-lane share | rc_physics calls with contact work")
for k in top: print("  ", " ".join("%8d" % v for v in q[k, :7]))
print("solver iterations of the step that changed NO accumulated impulse (of 10 per rc_physics call with an active arbiter), slowest environments:", [int(q[k, 7]) for k in top])
p3 = np.loadtxt("gpurun_out/rcprof3.txt")[:NE]
n3 = ["game logic", "position + shape cache + AABB", "broadphase", "quiet test (feet_far_apart)", "velocity update (quiet)", "joints (quiet)", "quiet substeps", "calls of the common part"]
print("the common part, cycles per step: mean over all environments / mean of the 12 slowest")
for k, n in enumerate(n3): print("  %-34s %10.0f %10.0f" % (n, p3[:, k].mean(), p3[top, k].mean()))
p4 = np.loadtxt("gpurun_out/rcprof4.txt")[:NE]
print("the batched game logic, cycles per step: mean over all environments / mean of the 12 slowest")
for k, n in enumerate(["loads + event test", "tick", "ball", "closest robots", "barrier + lane-0 stores"]): print("  %-34s %10.0f %10.0f" % (n, p4[:, k].mean(), p4[top, k].mean()))
print("the general solve of the slowest environments, cycles per step: level passes | joint phases | level passes run | cycles per level pass | per joint phase (10 per general solve)")
for k in top:
    nl, nj = int(p4[k, 7]) & 0xFFFFFFFF, int(p4[k, 7]) >> 32
    print("  %9d %9d %6d %8.0f %8.0f" % (p4[k, 5], p4[k, 6], nl, p4[k, 5] / max(nl, 1), p4[k, 6] / max(nj, 1)))
if os.environ.get("PROFILE_SAVE"): np.save(os.environ["PROFILE_SAVE"], d[:, 11])
