"""ShardedDynEnv on the GPU with the RCCL backend (world_size 1: one GPU box): the pipelined all-gather protocol
(the transport of step k runs on a side stream beside the following kernels, ring of slabs) returns exactly what a plain BatchedDynEnv computes."""
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


@pytest.mark.parametrize("A,tail", [(10, 160), (2, 160), (3, 21), (1, 8)])
def test_obs_peer_compaction_roundtrip(A, tail):
    """Driving Full rows rebuilt from the A self blocks + the tail once (dynenv_obs_pack_peers / unpack_peers_ranks)."""
    import ctypes as C
    import torch
    from dynenv_amd import _capi
    lib = _capi.load()
    ET, G, S, K = 29, 3, 9, 7
    carsEnd = S + (A - 1) * K
    D = carsEnd + tail
    P = A * S + tail
    g = torch.Generator(device="cuda").manual_seed(A)
    selfb = torch.randn((G, ET, A, S), device="cuda", generator=g)
    tl = torch.randn((G, ET, tail), device="cuda", generator=g)
    x = torch.empty((G, ET, A, D), device="cuda")
    cols = [0, 1, 2, 3, 4, 5, 8]
    for a in range(A):
        x[:, :, a, :S] = selfb[:, :, a]
        others = [c for c in range(A) if c != a]
        for q, c in enumerate(others):
            x[:, :, a, S + q * K:S + (q + 1) * K] = selfb[:, :, c][..., cols]
        x[:, :, a, carsEnd:] = tl
    stride = P * ET + 64  # the gathered blocks of consecutive ranks are further apart than their obs regions
    packed = torch.zeros((G, stride), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for r in range(G):
        _capi.check(lib.dynenv_obs_pack_peers(C.c_void_p(x[r].data_ptr()), ET, A, D, C.c_void_p(packed[r].data_ptr()), st), "pack")
    want = torch.cat([selfb.reshape(G, ET, A * S), tl], dim=2).reshape(G, ET * P)
    assert torch.equal(packed[:, :ET * P], want)
    y = torch.full_like(x, float("nan"))
    _capi.check(lib.dynenv_obs_unpack_peers_ranks(C.c_void_p(packed.data_ptr()), stride, G, ET, A, D, C.c_void_p(y.data_ptr()), st), "unpack")
    assert torch.equal(x, y)
