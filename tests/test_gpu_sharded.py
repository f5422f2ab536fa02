"""ShardedDynEnv on the GPU with the RCCL backend (world_size 1: one GPU box): the pipelined all-gather protocol
(step k gathered while the kernel of step k+1 runs, ping-pong slabs) returns exactly what a plain BatchedDynEnv computes."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_env_pipelined_gather_rccl_world1():
    import torch
    import torch.distributed as dist
    from dynenv_amd import BatchedDynEnv, DynEnvType
    from dynenv_amd.distributed import ShardedDynEnv
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        E, A, steps = 64, 10, 24
        sh = ShardedDynEnv(DynEnvType.DRIVE, E, A, gather=True, seed=3, device="cuda:0")
        ref = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=3, device="cuda:0")
        g_obs, _, _ = sh.reset()
        assert torch.equal(g_obs[0], ref.reset_flat())
        rng = np.random.default_rng(0)
        acts = [torch.tensor(rng.integers(0, 3, (E, A, 2)).astype(np.int32), device="cuda:0") for _ in range(steps)]
        handles, want = [], []
        for k in range(steps):
            handles.append(sh.step(acts[k], wait=False))
            o, r, d = ref.step_flat(acts[k])
            want.append((o.clone(), r.clone(), d.clone()))
            if k >= 1:  # consume step k-1 one step late, as a pipelined consumer would
                go, gr, gd = handles[k - 1].wait()
                assert torch.equal(go[0], want[k - 1][0]) and torch.equal(gr[0], want[k - 1][1]) and torch.equal(gd[0], want[k - 1][2])
        go, gr, gd = handles[-1].wait()
        assert torch.equal(go[0], want[-1][0]) and torch.equal(gr[0], want[-1][1])
        # lock-step use after pipelined use
        go, gr, gd = sh.step(acts[0])
        o, r, d = ref.step_flat(acts[0])
        assert torch.equal(go[0], o) and torch.equal(gr[0], r)
    finally:
        dist.destroy_process_group()


def test_obs_pack_unpack_roundtrip():
    import ctypes as C
    import torch
    from dynenv_amd import _capi
    lib = _capi.load()
    ET, A, D, split = 37, 10, 232, 72
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((ET, A, D), device="cuda", generator=g)
    x[:, :, split:] = x[:, :1, split:]  # every agent row shares the tail
    P = A * split + (D - split)
    packed = torch.empty((ET, P), device="cuda")
    y = torch.empty_like(x)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _capi.check(lib.dynenv_obs_pack(C.c_void_p(x.data_ptr()), ET, A, D, split, C.c_void_p(packed.data_ptr()), st), "pack")
    _capi.check(lib.dynenv_obs_unpack(C.c_void_p(packed.data_ptr()), ET, A, D, split, C.c_void_p(y.data_ptr()), st), "unpack")
    assert torch.equal(x, y)
    assert torch.equal(packed[:, :A * split].reshape(ET, A, split), x[:, :, :split]) and torch.equal(packed[:, A * split:], x[:, 0, split:])
