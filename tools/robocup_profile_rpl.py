"""Per-environment cycle breakdown of one RoboCup step of the register-resident (robot per lane) step kernel from a
-DDRV_PROFILE -DRC_FULL_EPW=2 build.  Usage (GPU box):  python tools/robocup_profile_rpl.py [step]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "dynenv_amd", "libdynenv_hip_rplprof.so")
os.environ["DYNENV_HIP_LIB"] = PROF
from dynenv_amd import build as _b
if not os.path.exists(PROF) or any(os.path.getmtime(d) > os.path.getmtime(PROF) for d in _b.DEPS if os.path.exists(d)):
    _b.build(out=PROF, defines=("DRV_PROFILE", "RC_FULL_EPW=2"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(ROOT)
import torch, numpy as np
from dynenv_amd import BatchedDynEnv, DynEnvType
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 100
env = BatchedDynEnv(DynEnvType.ROBO_CUP, 4096, 5, seed=42)
env.reset_flat()
g = torch.Generator(device="cuda"); g.manual_seed(1)
hi = torch.tensor([5, 3, 3, 7], device="cuda")
for s in range(STEP):
    a = (torch.rand((4096, 10, 4), device="cuda", generator=g) * hi).to(torch.int32)
    env.step_flat(a)
env.debug_counters()
d = np.loadtxt("gpurun_out/rcprof.txt")
nslow, ngen = d[:, 5] // 100, d[:, 5] % 100
tot = d[:, 11]
print("per-env cycles of one step: game logic | position+quiet test | quiet physics | general path (flush+broadphase+physics+reload) | TOTAL")
def row(m, name):
    if m.any(): print("  %-34s n=%4d  %8.0f %8.0f %8.0f %8.0f | %8.0f   slow-logic substeps %.1f general substeps %.1f" % (
        name, m.sum(), d[m, 0].mean(), d[m, 1].mean(), d[m, 2].mean(), d[m, 4].mean(), tot[m].mean(), nslow[m].mean(), ngen[m].mean()))
row(np.ones(len(d), bool), "all")
row(ngen == 0, "no general substep")
row((ngen > 0) & (ngen < 10), "1-9 general substeps")
row((ngen >= 10) & (ngen < 50), "10-49")
row(ngen == 50, "50")
print("TOTAL quantiles 1/50/90/99/100 %:", np.percentile(tot, [1, 50, 90, 99, 100]).astype(int))
print("physics stage sums (contacts+prestep, warm start, solver, post-solve) of the 50-general envs:", d[ngen == 50][:, [3, 6, 7, 8]].mean(0).astype(int) if (ngen == 50).any() else None)
top = np.argsort(-tot)[:8]
print(d[top].astype(int))
