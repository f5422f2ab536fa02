"""Where does the Driving step kernel spend its time?  Builds the -DDRV_PROFILE variant of the library, drives 4096
environments to a late step of an episode (many resting contacts) and prints, per environment, the cycles of one
launch and of the contact path's stages.  The launch lasts as long as its slowest environment, so the top of the list
is what to optimise.  Usage (GPU box):  python tools/contact_profile.py [step]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.environ.get("PROFILE_LIB", os.path.join(ROOT, "dynenv_amd", "libdynenv_hip_prof.so"))
os.environ["DYNENV_HIP_LIB"] = PROF  # read by dynenv_amd._capi at import
from dynenv_amd import build as _b
if not os.path.exists(PROF) or any(os.path.getmtime(d) > os.path.getmtime(PROF) for d in _b.DEPS if os.path.exists(d)):
    _b.build(out=PROF, defines=("DRV_PROFILE",))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(ROOT)
import torch, numpy as np
from dynenv_amd import BatchedDynEnv, DynEnvType
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 560
NE = int(os.environ.get("PROFILE_ENVS", "4096"))
env = BatchedDynEnv(DynEnvType.DRIVE, NE, 10, seed=42)
env.reset_flat()
g = torch.Generator(device="cuda"); g.manual_seed(1)
for s in range(STEP):
    a = torch.randint(0, 3, (NE, 10, 2), dtype=torch.int32, device="cuda", generator=g)
    env.step_flat(a)
print("substep path counters of the whole run:", env.debug_counters())
d = np.loadtxt("gpurun_out/dbgw.txt")[:NE]
c = d[:, 0]
print("cycles: mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f" % (c.mean(), *np.percentile(c, [50, 90, 99]), c.max()))
for k in range(11):
    m = d[:, 1] == k
    if m.any(): print("nContact=%d: n=%d mean cycles %.0f max %.0f, mean occ %.2f" % (k, m.sum(), c[m].mean(), c[m].max(), d[m, 2].mean()))
top = np.argsort(-c)[:15]
print(d[top])
ql = np.loadtxt("gpurun_out/dbgl.txt")[:NE]
print("light substep, cycles per step (10 substeps): loads+processAction | lane classification | tick / pedestrian FSM | position update+rotation+box | broadphase | whole function")
print("   top envs:"); [print("     ", ql[k, :6].astype(int)) for k in top[:8]]
print("   mean over all envs:", ql[:, :6].mean(0).astype(int), " mean over envs without contact substeps:", ql[d[:, 1] == 0, :6].mean(0).astype(int))
qraw = np.loadtxt("gpurun_out/dbgl.txt", dtype=np.uint64)[:NE]
print("narrowphase of the top envs, cycles per step: flag clear + candidate list | the pair tests (SAT, clipping) | slot matching, mailbox, barriers")
for k in top[:10]: print("     ", int(qraw[k, 6] & np.uint64(0xFFFFFFFF)), int(qraw[k, 6] >> np.uint64(32)), int(qraw[k, 7]))
m = d[:, 1] == 10
for o in range(0, 25):
    mm = m & (d[:, 2] == o)
    if mm.any(): print("nContact=10 occ=%d n=%d mean %.0f" % (o, mm.sum(), c[mm].mean()))
praw = np.loadtxt("gpurun_out/dbgp.txt", dtype=np.uint64)[:NE]
p = praw.astype(float)
lv = praw[:, 6]
modes = np.stack([(lv >> np.uint64(12 * k)) & np.uint64(0xFFF) for k in range(1, 5)], 1)  # level passes (x10 iterations) per solver mode
p[:, 6] = (lv & np.uint64(0xFFF)).astype(float)
ltry = (praw[:, 5] >> np.uint64(16)).astype(float)  # contact-path calls that ran to the end after a failed light-mode attempt
p[:, 5] = (praw[:, 5] & np.uint64(0xFFFF)).astype(float)
print("full-path calls that started with a failed light-mode attempt, top envs:", [int(ltry[k]) for k in top[:10]], " all envs: %d of %d calls" % (ltry.sum(), p[:, 5].sum()))
ncand = (praw[:, 7] >> np.uint64(16)).astype(float)  # candidate pairs handed to the narrowphase, summed over the step's calls
p[:, 7] = (praw[:, 7] & np.uint64(0xFFFF)).astype(float)
print("narrowphase candidates per contact-path call, top envs:", [round(ncand[k] / max(p[k, 5], 1), 1) for k in top[:10]], " all contact envs: mean %.1f" % (ncand[p[:, 5] > 0] / p[p[:, 5] > 0, 5]).mean())
print("solver level passes by mode [single bias-only, single general, multi bias-only, multi general] of the top envs:")
for k in top[:10]: print("  ", int(c[k]), modes[k])
print("all envs:", modes.sum(0))
# (the solve function reports its own stages: prestep | verdicts + slot record in d[2], velocity update | warm start + iterations in d[3])
tail = (praw[:, 2] >> np.uint64(32)).astype(float); p[:, 2] = (praw[:, 2] & np.uint64(0xFFFFFFFF)).astype(float)
iters = (praw[:, 3] >> np.uint64(32)).astype(float); p[:, 3] = (praw[:, 3] & np.uint64(0xFFFFFFFF)).astype(float)
print("solve function of the top envs: prestep | velocity update | warm start + iterations | verdicts + slot record | call, arguments, rest")
for k in top[:10]: print("  ", int(p[k, 2]), int(p[k, 3]), int(iters[k]), int(tail[k]), int(p[k, 4] - p[k, 2] - p[k, 3] - iters[k] - tail[k]))
print("top envs: total | load phase1 broad fast contact book store+obs | narrow slots prestep velupd solver | calls levels touched")
for k in top[:10]: print(int(c[k]), d[k, 4:11].astype(int), p[k].astype(int))
m = d[:, 1] == 10
print("mean over nContact=10 envs:", d[m, 4:11].mean(0).astype(int), p[m].mean(0).astype(int), "total", int(c[m].mean()))
m0 = d[:, 1] == 0
print("mean over nContact=0 envs:", d[m0, 4:11].mean(0).astype(int), "total", int(c[m0].mean()))
r = np.loadtxt("gpurun_out/dbgr.txt")
names = ["steady", "untouched", "freed", "contact ids changed", "first contact / not NORMAL", "bodies moving before prestep",
         "accumulated impulses changed", "bodies moving after solve", "(6) with zero stored jn", "(6) car-ped", "(6) car-car", "(6) car-static",
         "active arbiters in multi-level calls that are frozen + steady", "active arbiters (all full-path calls)",
         "... of which frozen + steady", "active arbiters in multi-level calls"]
print("slot outcomes over the whole run (per slot per contact-path call):")
for n, v in zip(names, r): print("  %-34s %d" % (n, v))

# stages of the slot update (between the narrowphase and the prestep) of the last step
sraw = np.loadtxt("gpurun_out/dbgs.txt")[:NE]
print("slot update of the top envs, cycles per step: cpArbiterUpdate on the slot lanes | rank | begin callbacks | expiry + component closure + bias reset | levels")
for k in top[:10]: print("     ", sraw[k, :5].astype(int))
