import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynenv_amd import make_dyn_env, DynEnvType, ObservationType, NoiseType
for E in (64, 1024):
    venv, _ = make_dyn_env(DynEnvType.DRIVE, E, 10, False, ObservationType.FULL, NoiseType.REALISTIC, 0, False)
    venv.reset()
    a = np.ones((E, 10, 2), np.int64)
    venv.step(a)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): venv.step(a)
    dt = (time.perf_counter() - t0) / n
    # host copy only (what a C caller with host buffers would pay): obs D2H
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): venv.step_flat(torch.ones((E, 10, 2), dtype=torch.int32, device="cuda")); h = venv.obs.cpu()
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t1) / 20
    print("E=%d compat step(): %.1f ms/step = %.2f M agent-steps/s ; step_flat + obs D2H: %.3f ms/step = %.1f M agent-steps/s" % (E, dt * 1e3, E * 10 / dt / 1e6, dt2 * 1e3, E * 10 / dt2 / 1e6))
    venv.close()
