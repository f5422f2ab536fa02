"""Two 4096-environment Driving handles on two streams, with and without SIMD isolation, alternating (a, b, a, b, ...):
   python tools/two_handles_time.py [passes]
ms per step pair for every pass; the isolation counters of the first handle at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dynenv_amd

E, A, steps = 4096, 10, 300
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mk = lambda seed: dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, E, A, seed=seed)
g = torch.Generator(device="cuda").manual_seed(9)
acts = [torch.randint(0, 3, (E, A, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(steps)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(pair):
    for h in pair:
        h.reset_flat()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a in acts:
        for h, st in zip(pair, streams):
            with torch.cuda.stream(st):
                h.step_flat(a, auto_reset=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


os.environ["DYNENV_NO_ISOLATION"] = "1"
ref = [mk(11), mk(12)]
del os.environ["DYNENV_NO_ISOLATION"]
iso = [mk(11), mk(12)]
for p in range(passes):
    print("pass %d: without isolation %.4f ms, with %.4f ms per step pair" % (p, run(ref), run(iso)), flush=True)
print(iso[0].debug_counters())
