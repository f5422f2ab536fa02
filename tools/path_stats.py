import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynenv_amd import BatchedDynEnv, DynEnvType
E, A = 4096, 10
env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=42)
env.reset_flat()
g = torch.Generator(device="cuda").manual_seed(0)
prev = dict(fast=0, quiescent=0, contact=0, slot_sum=0, why_cand=0, why_moving=0, why_inert=0, steady=0, light=0, split=0)
for s in range(600):
    a = torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32)
    env.step_flat(a, auto_reset=False)
    if s % 100 == 99:
        c = env.debug_counters()
        d = {k: c[k] - prev[k] for k in prev}
        tot = d["fast"] + d["quiescent"] + d["contact"] + d["steady"]
        print("steps %3d-%3d: fast %.3f quiescent %.3f contact %.3f (cand-changed %.3f moving %.3f not-inert %.3f) steady-replay %.3f (+%.3f of contact via light mode), split-lane general sweeps in %.4f  avg live slots %.2f; isolated next step %d, isolation timeouts %d" % (s - 99, s, d["fast"] / tot, d["quiescent"] / tot, d["contact"] / tot, d["why_cand"] / tot, d["why_moving"] / tot, d["why_inert"] / tot, d["steady"] / tot, d["light"] / tot, d["split"] / tot, d["slot_sum"] / tot, c["isolated_next"], c["isolation_timeouts"]))
        prev = c
