#!/usr/bin/env python3
"""The HIP path itself against trajectories of the reference's OWN `step()`: tools/reference_step_fuzz.py, run in the build container with
FUZZ_DUMP=<dir>, keeps every trajectory it generates (inputs, actions and what the reference computed: rewards, observations, states);
this tool, on the GPU box, replays each of them through the C ABI (one 1-environment handle per trajectory) under the checks and tolerances
of tests/test_oracle_golden_contacts.py - what tests/test_gpu_golden_contacts.py does for the committed fixtures, on a population.

   FUZZ_DUMP=tests/golden/_fuzz python3 tools/reference_step_fuzz.py 150 100 50 30 40 100 100 100 150     (build container)
   python3 tools/hip_reference_step_fuzz.py tests/golden/_fuzz                                             (GPU box)
"""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402  (state structs only: the oracle is not run)
import test_oracle_golden_contacts as tc  # noqa: E402


def main():
    import torch
    import dynenv_amd as da
    assert torch.cuda.is_available(), "needs the MI355X"
    d = sys.argv[1]
    envs = []

    def driving(n_players, seed, offset):
        env = da.BatchedDynEnv(da.DynEnvType.DRIVE, 1, n_players, seed=seed, env_id_offset=offset)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, dn = env.step_flat(a[None], auto_reset=False)
            return o[0, 0].cpu().numpy(), r[0].cpu().numpy(), int(dn[0])

        def stats():
            r, p, o, g = env.episode_stats()
            return r[0].cpu().numpy(), p[0].cpu().numpy(), o[0].cpu().numpy(), g[0].cpu().numpy()
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0)), stats

    def robocup(n, seed, offset, flags, magn=None):
        kw = {} if magn is None else dict(observationType=da.ObservationType.PARTIAL, noiseType=da.NoiseType.REALISTIC, noiseMagnitude=magn)
        env = da.BatchedDynEnv(da.DynEnvType.ROBO_CUP, 1, n, seed=seed, env_id_offset=offset, flags=flags, **kw)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, dn = env.step_flat(a[None], auto_reset=False)
            return o[0].cpu().numpy(), r[0].cpu().numpy(), int(dn[0])
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))

    def driving_partial(n_players, seed, offset, magn):
        env = da.BatchedDynEnv(da.DynEnvType.DRIVE, 1, n_players, observationType=da.ObservationType.PARTIAL, noiseType=da.NoiseType.REALISTIC,
                               noiseMagnitude=magn, seed=seed, env_id_offset=offset)
        env.reset_flat()
        envs.append(env)

        def step(a):
            o, r, dn = env.step_flat(a[None], auto_reset=False)
            return o[0, 0].cpu().numpy(), r[0].cpu().numpy(), int(dn[0])
        return (lambda st: env.set_state(0, st)), step, (lambda: env.get_state(0))

    kinds = {}
    for f in sorted(glob.glob(os.path.join(d, "*.npz"))):
        kinds.setdefault(os.path.basename(f).rsplit("_", 1)[0], []).append(f)
    bad = 0
    for kind, files in kinds.items():
        t0, fails, steps, flags, excused = time.time(), [], 0, 0, []
        for f in files:
            z = np.load(f)
            del envs[:]
            try:
                if kind.startswith("driving_partial"):
                    tc.check_partial_trajectory(z, "t", driving_partial)
                    steps += len(z["t_actions"])
                elif kind.startswith("driving"):
                    tc.check_trajectory(z, "t", driving)
                    steps += len(z["t_actions"])
                elif kind.startswith("robocup_partial"):
                    steps += tc.check_robocup_trajectory(z, "t", robocup, partial=True, own_line_slack=excused)
                else:
                    steps += tc.check_robocup_trajectory(z, "t", robocup)
            except AssertionError as e:
                fails.append((os.path.basename(f), str(e)[:160]))
            for env in envs:
                flags |= int(env.error_flags()) & ~16   # (bit 4: exactly touching capsule cores, reported by design)
                env.close()
        print("%-18s %4d trajectories of the reference's step(), %6d steps replayed on the HIP path (tolerances of tests/test_oracle_golden_contacts.py): "
              "%d failures, error flags %d%s  (%.0f s)" % (kind, len(files), steps, len(fails), flags,
                 ", %d rows excused (a penalized robot's own side line: decided by the last bit of libm's sin / cos)" % len(excused) if excused else "", time.time() - t0))
        for x in fails:
            print("   FAILURE", x)
        bad += len(fails) + (1 if flags else 0)
        sys.stdout.flush()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
