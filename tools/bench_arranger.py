#!/usr/bin/env python3
"""Throughput of the GPU arranger on real observations at BASELINE sizes: one JSON line per config with
the HBM roofline of the gather kernel and the numpy oracle (a restatement of the reference's InOutArranger) timed on a
bounded sample.  Usage (GPU box):  python tools/bench_arranger.py [--envs 4096] [--feat 128]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--feat", type=int, default=128, help="embedding width F of rearrange_outputs")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import numpy as np
    import torch
    import arranger as oa
    from dynenv_amd import BatchedDynEnv, DynEnvType, GpuInOutArranger, NoiseType, ObservationType, groups_for
    E = args.envs
    for cfg in ("driving_full", "robocup", "driving_partial"):
        if cfg == "robocup":
            env = BatchedDynEnv(DynEnvType.ROBO_CUP, E, 5, seed=42); hi = [5, 3, 3, 7]
        elif cfg == "driving_partial":
            env = BatchedDynEnv(DynEnvType.DRIVE, E, 10, observationType=ObservationType.PARTIAL,
                                noiseType=NoiseType.REALISTIC, noiseMagnitude=3, seed=42); hi = [3, 3]
        else:
            env = BatchedDynEnv(DynEnvType.DRIVE, E, 10, seed=42); hi = [3, 3]
        env.reset_flat()
        g = torch.Generator(device="cuda").manual_seed(1)
        hit = torch.tensor(hi, device="cuda")
        for _ in range(20):
            a = (torch.rand((E, env.n_agents, len(hi)), generator=g, device="cuda") * hit).to(torch.int32)
            obs, _, _ = env.step_flat(a, auto_reset=False)
        obs = obs.contiguous()
        count_env = env.counts() if env.env_type == DynEnvType.DRIVE else None
        types = groups_for(env)["movable"]
        arr = GpuInOutArranger(types, E, env.n_agents, env.n_time_steps, env.obs_dim)
        inputs, countArr = arr.rearrange_inputs(obs, count_env)
        outs = [torch.randn((i.shape[0], args.feat), device="cuda") for i in inputs]
        arr.rearrange_outputs(outs, countArr)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(args.reps):
            inputs, countArr = arr.rearrange_inputs(obs, count_env)
        ev[1].record()
        for _ in range(args.reps):
            padded, masks = arr.rearrange_outputs(outs, countArr)
        ev[2].record()
        torch.cuda.synchronize()
        ms_in = ev[0].elapsed_time(ev[1]) / args.reps
        ms_out = ev[1].elapsed_time(ev[2]) / args.reps
        n_obj = [int(i.shape[0]) for i in inputs]
        TP = env.n_time_steps * E * env.n_agents
        # algorithmic bytes: rows read + rows written + slot per object + mask + counts/base per (type, time, player)
        b_in = sum(n * t.feat * 4 * 2 + n * 4 for n, t in zip(n_obj, types)) + TP * countArr[1] + TP * len(types) * 8 + TP * 4
        b_out = sum(n_obj) * args.feat * 4 + padded.numel() * 4  # embeddings read once + every padded element written once
        # CPU: the numpy restatement of the reference arranger on a bounded sample of the same observations
        Es = min(E, 128)
        compat = env._compat_obs(obs[:Es], count_env[:Es].cpu().numpy() if count_env is not None else None)
        x = [[[list(compat[e, t, p, 0]) for p in range(env.n_agents)] for t in range(env.n_time_steps)] for e in range(Es)]
        t0 = time.perf_counter()
        o_in, o_cnt = oa.rearrange_inputs(x, len(types), Es * env.n_agents, env.n_time_steps)
        o_outs = [np.zeros((len(i), args.feat), np.float32) if len(i) else None for i in o_in]
        oa.rearrange_outputs(o_outs, o_cnt)
        cpu_s = time.perf_counter() - t0
        agent_rows = E * env.n_agents * env.n_time_steps
        print(json.dumps({
            "metric": "arranged agent-observations/s", "config": cfg, "envs": E, "players": E * env.n_agents,
            "time_steps": env.n_time_steps, "objects": n_obj, "max_count": countArr[1], "embed_width": args.feat,
            "rearrange_inputs_ms": ms_in, "rearrange_outputs_ms": ms_out,
            "value": agent_rows / ((ms_in + ms_out) * 1e-3), "unit": "agent-observations/s",
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                         "rearrange_inputs": {"alg_bytes": b_in, "achieved": b_in / (ms_in * 1e-3) / 1e9},
                         "rearrange_outputs": {"alg_bytes": b_out, "achieved": b_out / (ms_out * 1e-3) / 1e9}},
            "cpu_baseline": {"kind": "port", "cores": 1, "value": Es * env.n_agents * env.n_time_steps / cpu_s,
                             "unit": "agent-observations/s",
                             "sample": "oracle/arranger.py (numpy restatement of InOutArranger) on the first %d envs, %.2f s" % (Es, cpu_s)},
        }))
        env.close()


if __name__ == "__main__":
    main()
