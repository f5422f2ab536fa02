"""Rate of the reference-compatible step() at BASELINE's batch size: the lazy observation / info containers
(what a drop-in caller gets), the same with the caller touching EVERY element, the eager object array of round 1, and the
PCIe-inclusive floor (step_flat + one device->host copy of the observations).   Usage (GPU box): python tools/compat_rate.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dynenv_amd import DynEnvType, NoiseType, ObservationType, make_dyn_env


def rate(venv, a, n, touch):
    venv.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        obs, rew, dones, infos = venv.step(a)
        if touch:
            np.asarray(obs)
    return (time.perf_counter() - t0) / n


for E in (1024, 4096):
    a = np.ones((E, 10, 2), np.int64)
    venv, _ = make_dyn_env(DynEnvType.DRIVE, E, 10, False, ObservationType.FULL, NoiseType.REALISTIC, 0, False)
    venv.reset()
    lazy = rate(venv, a, 20, False)
    touched = rate(venv, a, 5, True)
    obs = venv.step(a)[0]
    dense, counts = obs._dense, obs._counts
    t2 = time.perf_counter()
    for _ in range(3):
        loop = venv._compat_obs(dense, counts)
    loop_ms = (time.perf_counter() - t2) / 3
    del loop
    venv.close()
    venv, _ = make_dyn_env(DynEnvType.DRIVE, E, 10, False, ObservationType.FULL, NoiseType.REALISTIC, 0, False, eager_compat=True)
    venv.reset()
    eager = rate(venv, a, 3, False)
    ad = torch.ones((E, 10, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        venv.step_flat(ad, auto_reset=False)
        venv.obs.cpu()
    torch.cuda.synchronize()
    floor = (time.perf_counter() - t1) / 20
    venv.close()
    print("E=%d Driving Full: compat step() lazy %.2f ms = %.1f M agent-steps/s | every element materialised (np.asarray(obs): D2H + the bulk "
          "builder) %.1f ms = %.2f M | the plain triple loop on the same host copy %.1f ms | eager_compat=True step() %.1f ms = %.2f M | step_flat + obs D2H %.2f ms = %.1f M"
          % (E, lazy * 1e3, E * 10 / lazy / 1e6, touched * 1e3, E * 10 / touched / 1e6, loop_ms * 1e3, eager * 1e3, E * 10 / eager / 1e6,
             floor * 1e3, E * 10 / floor / 1e6))
