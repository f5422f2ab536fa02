import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, numpy as np
from dynenv_amd import BatchedDynEnv, DynEnvType
import oracle_lib as ol
E=4096
for seed in (42,):
    env = BatchedDynEnv(DynEnvType.ROBO_CUP, E, 5, seed=seed); env.reset_flat()
    g = torch.Generator(device="cuda").manual_seed(1234)
    hi = torch.tensor([5,3,3,7], device="cuda")
    pool = [(torch.rand((E,10,4), generator=g, device="cuda")*hi).to(torch.int32) for _ in range(16)]
    first=None
    for i in range(240):
        env.step_flat(pool[i&15], auto_reset=False)
        f = env.error_flags()
        if f and first is None:
            first=(i,f); print("seed",seed,"first error flags",f,"at step",i)
            # find env
            import ctypes as C
            break
    print("final flags", env.error_flags())
    if first:
        # replay on oracle for the same env id to see whether the oracle reports too
        ora = ol.OracleEnv(env_type=0, num_envs=E, n_players=5, seed=seed, flags=ol.ROBOCUP_DEFAULT_FLAGS, threads=16); ora.reset()
        for i in range(first[0]+1):
            ora.step_noobs(pool[i&15].cpu().numpy())
        print("oracle degenerate:", ora.degenerate(), [e for e in range(E) if ora.degenerate_env(e)][:10])
        bad=[e for e in range(E) if ora.degenerate_env(e)]
        for e in bad[:3]:
            st=ora.get_state(e)
            print(e, [(r.lpx,r.lpy,r.la,r.rpx,r.rpy,r.ra,r.fallen,r.penalized) for r in list(st.robots)[:10]])
