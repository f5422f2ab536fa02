"""Deterministic replay files (the reference has no checkpointing of environments).

A replay file = the configuration of a BatchedDynEnv + an exact checkpoint + the actions applied afterwards + a checksum
of the rewards each step produced.  `replay()` rebuilds the environment, loads the checkpoint, re-applies the actions and
checks every step's rewards bit for bit — the tool for hunting a parity drift or resuming a 32k-env job after a fault.
"""
import numpy as np

from .enums import DynEnvType, NoiseType, ObservationType
from .vec_env import BatchedDynEnv


def _reward_digest(rewards, dones):
    """digest of one step: xor and sum of the raw bits of its rewards, number of finished environments"""
    bits = rewards.detach().cpu().numpy().view(np.uint64).ravel()
    d = dones.detach().cpu().numpy().astype(np.uint64).ravel()
    return np.array([np.bitwise_xor.reduce(bits), bits.sum(dtype=np.uint64), d.sum(dtype=np.uint64)], np.uint64)


class ReplayRecorder(object):
    """rec = ReplayRecorder(env); obs, rew, done = rec.step(actions) ...; rec.save(path)"""

    def __init__(self, env):
        self.env = env
        c = env.cfg
        self.meta = dict(env_type=int(c.env_type), num_envs=int(c.num_envs), n_players=int(c.n_players), obs_type=int(c.obs_type),
                         noise_type=int(c.noise_type), noise_magnitude=float(c.noise_magnitude), seed=int(c.seed),
                         env_id_offset=int(c.env_id_offset), flags=int(c.flags))
        self.checkpoint = env.checkpoint()
        self.actions, self.heads, self.digests = [], [], []

    def step(self, actions, auto_reset=True):
        # what the kernel is given: the discrete channels as int32 and, with allowHeadTurn, the continuous head action as
        # float64 (RoboCupEnvironment.py:339-342) - recorded bit for bit, not truncated
        a, head = self.env._stage_actions(actions)
        out = self.env.step_flat(actions, auto_reset=auto_reset)
        self.actions.append(a.detach().cpu().numpy().astype(np.int8))
        if head is not None:
            self.heads.append(head.detach().cpu().numpy().astype(np.float64))
        self.digests.append(_reward_digest(out[1], out[2]))
        return out

    def save(self, path):
        extra = {"heads": np.stack(self.heads)} if self.heads else {}
        np.savez_compressed(path, checkpoint=self.checkpoint, actions=np.stack(self.actions), digests=np.stack(self.digests),
                            auto_reset=np.array(1, np.int8), **extra, **{"meta_" + k: np.array(v) for k, v in self.meta.items()})


def replay(path, device=None, on_step=None):
    """Re-run a replay file.  Returns (env, n_steps); raises AssertionError at the first step whose rewards/dones differ."""
    import torch
    z = np.load(path)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    env = BatchedDynEnv(DynEnvType(m["env_type"]), m["num_envs"], m["n_players"], observationType=ObservationType(m["obs_type"]),
                        noiseType=NoiseType(m["noise_type"]), noiseMagnitude=m["noise_magnitude"], seed=m["seed"],
                        device=device, env_id_offset=m["env_id_offset"], flags=m["flags"])
    env.restore(z["checkpoint"])
    acts, digs = z["actions"], z["digests"]
    heads = z["heads"] if "heads" in z.files else None
    for i in range(acts.shape[0]):
        if heads is not None:  # (discrete [E, A, 3], head [E, A] float64): the continuous head channel as recorded
            a = (acts[i][..., :3].astype(np.int32), heads[i])
        else:
            a = torch.tensor(acts[i].astype(np.int32), device=env.device)
        out = env.step_flat(a)
        d = _reward_digest(out[1], out[2])
        assert np.array_equal(d, digs[i]), "replay diverged at step %d" % i
        if on_step is not None:
            on_step(i, out)
    return env, acts.shape[0]
