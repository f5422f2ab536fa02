"""N > 1 path on CPU: world_size-2 `gloo` run of the packed-slab all-gather (dynenv_amd.distributed) with the oracle
standing in for the per-rank environments.  Checks (a) the single collective reassembles obs|rewards|dones exactly and
(b) shard-count invariance: env results depend on the GLOBAL env id only."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_envs, steps, out_dir, pipelined=False, robocup=False, transport="dense"):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    from dynenv_amd.distributed import PackedSlab, StepGather, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, per = shard_range(total_envs, rank, world)
    A = 10
    if robocup:
        ora = ol.OracleEnv(env_type=0, num_envs=per, n_players=5, seed=42, env_id_offset=off, flags=ol.ROBOCUP_DEFAULT_FLAGS)
    else:
        ora = ol.OracleEnv(env_type=1, num_envs=per, n_players=A, seed=42, env_id_offset=off)
    # the transport formats of dynenv_amd.distributed on CPU tensors (their index maps in numpy): dense, shared-tail, peer-compacted
    layout = {} if transport == "dense" else dict(split=9 + (A - 1) * 7) if transport == "tail" else dict(peers=True)
    slab = PackedSlab(torch, torch.device("cpu"), per, ora.T, A, ora.D, **layout)
    slab2 = PackedSlab(torch, torch.device("cpu"), per, ora.T, A, ora.D, **layout) if pipelined else None
    gather = StepGather(torch, dist, slab, slab2=slab2)
    if transport != "dense":
        assert slab.nbytes < PackedSlab(torch, torch.device("cpu"), per, ora.T, A, ora.D).nbytes // 2
    rng = np.random.default_rng(123)
    if robocup:
        all_actions = np.stack([rng.integers(0, k, size=(steps, total_envs, A)) for k in (5, 3, 3, 7)], -1).astype(np.int32)
    else:
        all_actions = rng.integers(0, 3, size=(steps, total_envs, A, 2)).astype(np.int32)

    def publish(obs, rew, done):
        slab.obs.copy_(torch.from_numpy(obs))
        slab.rewards.copy_(torch.from_numpy(rew))
        slab.dones.copy_(torch.from_numpy(done))
        return gather()

    obs = ora.reset()
    g_obs, _, _ = publish(obs, ora.rewards, ora.dones)
    outs = [g_obs.reshape(total_envs, ora.T, A, ora.D).clone().numpy()]

    def snapshot(views):
        g_obs, g_rew, g_done = views
        return (g_obs.reshape(total_envs, ora.T, A, ora.D).clone().numpy(),
                g_rew.reshape(total_envs, A).clone().numpy(), g_done.reshape(total_envs).clone().numpy())

    if pipelined:
        # the bench.py / ShardedDynEnv(wait=False) protocol: step k writes slabs[k % 2], its gather is started and only
        # waited for when the slab is about to be rewritten (or the consumer asks for the views)
        handles = []
        for s in range(steps):
            gather.release(s)
            sl = gather.slabs[s % 2]
            o, r, d = ora.step(all_actions[s, off:off + per])
            sl.obs.copy_(torch.from_numpy(o)); sl.rewards.copy_(torch.from_numpy(r)); sl.dones.copy_(torch.from_numpy(d))
            handles.append(gather.start(s))
            if s >= 1:  # consume step s-1 one step late
                outs.append(snapshot(handles[s - 1].wait()))
        outs.append(snapshot(handles[-1].wait()))
        gather.drain()
    else:
        for s in range(steps):
            o, r, d = ora.step(all_actions[s, off:off + per])
            outs.append(snapshot(publish(o, r, d)))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), obs0=outs[0], obs=np.stack([x[0] for x in outs[1:]]),
                 rew=np.stack([x[1] for x in outs[1:]]), done=np.stack([x[2] for x in outs[1:]]), actions=all_actions)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined,robocup,transport,world", [(False, False, "dense", 2), (True, False, "dense", 2), (True, False, "compact", 2),
                                                               (False, False, "tail", 2), (True, True, "dense", 2),
                                                               (True, False, "compact", 4), (True, False, "compact", 8)])
def test_two_rank_gather_and_shard_invariance(tmp_path, oracle_built, pipelined, robocup, transport, world):
    """world 2 for every transport; world 4 and 8 (the rank counts of the driver's scaling run, rehearsed over gloo) for the one bench.py uses"""
    import torch.multiprocessing as mp
    total, steps = 8, 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, steps, str(tmp_path), pipelined, robocup, transport), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    import oracle_lib as ol
    single = (ol.OracleEnv(env_type=0, num_envs=total, n_players=5, seed=42, env_id_offset=0, flags=ol.ROBOCUP_DEFAULT_FLAGS) if robocup
              else ol.OracleEnv(env_type=1, num_envs=total, n_players=10, seed=42, env_id_offset=0))
    o0 = single.reset()
    np.testing.assert_array_equal(z["obs0"], o0)
    for s in range(steps):
        o, r, d = single.step(z["actions"][s])
        np.testing.assert_array_equal(z["obs"][s], o)
        np.testing.assert_array_equal(z["rew"][s], r)
        np.testing.assert_array_equal(z["done"][s], d)
