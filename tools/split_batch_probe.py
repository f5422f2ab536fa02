#!/usr/bin/env python3
"""Throughput against batch size and against K independent sub-batches on K streams.

A launch of the step kernel lasts as long as its slowest environment, and 4096 environments are exactly one residency round
(16 waves per CU), so most SIMDs idle through the tail of every launch.  Two ways to fill that tail without touching the
arithmetic of any environment:
  (i)  K handles of E/K environments (env_id_offset = k E/K: the same global environments, bit for bit) on K streams,
       stepped round-robin - sub-batch k's step i+1 starts when ITS step i is done, beside the tails of the others;
  (ii) one handle with more environments than fit at once (8192 ... 32768): later blocks backfill retired ones.
Whole-episode means (every step of an episode, resets excluded), wall clock around a device synchronise.

  python tools/split_batch_probe.py [--workload driving|robocup] [--envs 4096] [--ks 1,2,4,8] [--sizes 8192,16384,32768]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_split(torch, dev, workload, E, K, steps, seed=42, check=None):
    from dynenv_amd import BatchedDynEnv, DynEnvType
    robocup = workload == "robocup"
    players = 5 if robocup else 10
    A = 10
    et = DynEnvType.ROBO_CUP if robocup else DynEnvType.DRIVE
    probe = BatchedDynEnv(et, 1, players, device=dev)
    T, D = probe.n_time_steps, probe.obs_dim
    probe.close()
    obs = torch.zeros((E, T, A, D), dtype=torch.float32, device=dev)
    rew = torch.zeros((E, A), dtype=torch.float64, device=dev)
    don = torch.zeros((E,), dtype=torch.uint8, device=dev)
    n = E // K
    envs = [BatchedDynEnv(et, n, players, seed=seed, device=dev, env_id_offset=k * n,
                          out_buffers=(obs[k * n:(k + 1) * n], rew[k * n:(k + 1) * n], don[k * n:(k + 1) * n])) for k in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)] if K > 1 else [torch.cuda.current_stream(dev)]
    g = torch.Generator(device=dev).manual_seed(4321)
    if robocup:
        hi = torch.tensor([5, 3, 3, 7], device=dev)
        pool = [(torch.rand((E, A, 4), generator=g, device=dev) * hi).to(torch.int32) for _ in range(16)]
    else:
        pool = [torch.randint(0, 3, (E, A, 2), generator=g, device=dev, dtype=torch.int32) for _ in range(16)]
    sub = [[p[k * n:(k + 1) * n] for k in range(K)] for p in pool]
    torch.cuda.synchronize(dev)

    def episode(timed):
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                envs[k].reset_flat()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    envs[k].step_flat(sub[i & 15][k], auto_reset=False)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, t_host

    episode(False)
    dt, t_host = episode(True)
    digest = (float(rew.sum().item()), int(obs.view(torch.int32).to(torch.int64).sum().item()))
    for e in envs:
        assert e.error_flags() == 0
        e.close()
    return {"workload": workload, "envs": E, "K": K, "ms_per_step": dt / steps * 1e3, "host_enqueue_ms_per_step": t_host / steps * 1e3,
            "agent_steps_per_s": E * A * steps / dt, "digest": digest}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="driving")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--ks", default="1,2,4,8")
    ap.add_argument("--sizes", default="8192,16384,32768")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    steps = a.steps or (240 if a.workload == "robocup" else 600)
    res = []
    for K in [int(x) for x in a.ks.split(",") if x]:
        r = run_split(torch, dev, a.workload, a.envs, K, steps)
        res.append(r)
        print(json.dumps(r), flush=True)
    base = res[0]["digest"] if res else None
    for r in res:
        r["same_results_as_K1"] = r["digest"] == base
    for E in [int(x) for x in a.sizes.split(",") if x]:
        r = run_split(torch, dev, a.workload, E, 1, steps)
        res.append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
