import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dynenv_amd import BatchedDynEnv, DynEnvType
E, A = 4096, 10
env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=42)
g = torch.Generator(device="cuda").manual_seed(4321)
pool = [torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(16)]
for rep in range(2):
    env.reset_flat()
    for blk in range(12):
        torch.cuda.synchronize()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for i in range(50):
            env.step_flat(pool[i & 15], auto_reset=False)
        k1.record(); torch.cuda.synchronize()
        c = env.debug_counters()
        print(rep, blk * 50 + 50, "%.4f ms" % (k0.elapsed_time(k1) / 50), {k: v for k, v in c.items() if "isol" in k or "place" in k})
