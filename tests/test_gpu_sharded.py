"""ShardedDynEnv on the GPU with the RCCL backend (world_size 1: one GPU box): the pipelined all-gather protocol
(the transport of step k runs on a side stream beside the following kernels, ring of slabs) returns exactly what a plain BatchedDynEnv computes."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _process_group_teardown():
    yield
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_sharded_env_pipelined_gather_rccl_world1():
    import torch
    import torch.distributed as dist
    from dynenv_amd import BatchedDynEnv, DynEnvType
    from dynenv_amd.distributed import ShardedDynEnv
    _rccl_world1()
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
        pass  # the process group (RCCL, world_size 1) is shared by this module's tests


def test_pipelined_gather_across_an_episode_end_with_auto_reset():
    """the pipelined protocol (ring of 4 slabs, transport of step k beside the following kernels) for many times the ring length,
    through the lock-step episode end: step 599 returns done and - auto-reset, SubprocVecEnv semantics - the FIRST observation of
    the next episode, written into that step's slab by the reset kernels; every gathered view equals a plain BatchedDynEnv"""
    import torch
    from dynenv_amd import BatchedDynEnv, DynEnvType
    from dynenv_amd.distributed import ShardedDynEnv
    _rccl_world1()
    E, A = 48, 10
    sh = ShardedDynEnv(DynEnvType.DRIVE, E, A, gather=True, seed=5, device="cuda:0", ring=4)
    ref = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=5, device="cuda:0")
    S = ref.steps_per_episode
    g_obs, _, _ = sh.reset()
    assert torch.equal(g_obs[0], ref.reset_flat())
    g = torch.Generator(device="cuda:0").manual_seed(3)
    pool = [torch.randint(0, 3, (E, A, 2), generator=g, device="cuda:0", dtype=torch.int32) for _ in range(32)]
    handles, want, dones = {}, {}, 0
    lag = 2
    for k in range(S + 40):
        handles[k] = sh.step(pool[k & 31], wait=False)
        o, r, d = ref.step_flat(pool[k & 31])          # auto_reset=True
        if k >= S - 20:                                 # keep (and compare) the window around the episode end
            want[k] = (o.clone(), r.clone(), d.clone())
        if k - lag in want:
            go, gr, gd = handles.pop(k - lag).wait()
            w = want.pop(k - lag)
            assert torch.equal(go[0], w[0]) and torch.equal(gr[0], w[1]) and torch.equal(gd[0], w[2]), "step %d" % (k - lag)
            dones += int(w[2].all())
        elif k - lag in handles and k - lag < S - 20:
            handles.pop(k - lag).wait()                 # consumed, not compared: the slabs keep rotating
    for k in sorted(want):
        go, gr, gd = handles.pop(k).wait()
        assert torch.equal(go[0], want[k][0]) and torch.equal(gr[0], want[k][1]), "step %d" % k
    assert dones == 1, "exactly one episode end in the window"
    assert sh.env._episode_step == 40 and ref._episode_step == 40
    sh.gather.drain()
    sh.env.close(); ref.close()


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


def _rccl_world1():
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return torch, dist


def test_pack_kernels_follow_the_numpy_index_maps_on_real_observations():
    """dynenv_obs_pack / dynenv_obs_pack_peers on observations the step kernel wrote == dynenv_amd.distributed.pack_*_np (the
    format's definition, checked on the CPU against the oracle in tests/test_host_and_abi.py), and the unpack kernels invert them"""
    import ctypes as C
    import torch
    from dynenv_amd import BatchedDynEnv, DynEnvType, _capi
    from dynenv_amd.distributed import pack_peers_np, pack_tail_np
    lib = _capi.load()
    E, A = 32, 10
    env = BatchedDynEnv(DynEnvType.DRIVE, E, A, seed=11, device="cuda:0")
    env.reset_flat()
    rng = np.random.default_rng(2)
    for _ in range(30):
        env.step_flat(rng.integers(0, 3, (E, A, 2)).astype(np.int32))
    D, split = env.obs_dim, 9 + (A - 1) * 7
    x = env.obs.reshape(E, A, D).contiguous()
    xn = x.cpu().numpy()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p_tail = torch.empty((E, A * split + D - split), device="cuda")
    p_peer = torch.empty((E, A * 9 + D - split), device="cuda")
    _capi.check(lib.dynenv_obs_pack(C.c_void_p(x.data_ptr()), E, A, D, split, C.c_void_p(p_tail.data_ptr()), st), "pack")
    _capi.check(lib.dynenv_obs_pack_peers(C.c_void_p(x.data_ptr()), E, A, D, C.c_void_p(p_peer.data_ptr()), st), "pack_peers")
    np.testing.assert_array_equal(p_tail.cpu().numpy(), pack_tail_np(xn, split))
    np.testing.assert_array_equal(p_peer.cpu().numpy(), pack_peers_np(xn))
    y = torch.full_like(x, float("nan"))
    _capi.check(lib.dynenv_obs_unpack_peers_ranks(C.c_void_p(p_peer.data_ptr()), p_peer.numel(), 1, E, A, D, C.c_void_p(y.data_ptr()), st), "unpack")
    assert torch.equal(x, y)
    env.close()


@pytest.mark.parametrize("kind,transport", [("DRIVE", "tail"), ("DRIVE", "dense"), ("ROBO_CUP", "auto"), ("DRIVE_PARTIAL", "auto")])
def test_pipelined_gather_over_rccl_other_transports_and_environments(kind, transport):
    """the whole N > 1 path at world_size 1 for the shared-tail and dense transports and for the environments whose
    observations travel dense (RoboCup, Partial observations): gathered views == a plain BatchedDynEnv, bit for bit"""
    torch, dist = _rccl_world1()
    from dynenv_amd import BatchedDynEnv, DynEnvType, NoiseType, ObservationType
    from dynenv_amd.distributed import PackedSlab, StepGather, shared_tail_split, transport_layout
    robocup = kind == "ROBO_CUP"
    kw = dict(observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC, noiseMagnitude=3) if kind == "DRIVE_PARTIAL" else {}
    et = DynEnvType.ROBO_CUP if robocup else DynEnvType.DRIVE
    E, n, steps = 32, 5 if robocup else 10, 10
    probe = BatchedDynEnv(et, 1, n, device="cuda:0", **kw)
    T, A, D = probe.n_time_steps, probe.n_agents, probe.obs_dim
    if transport == "auto":
        layout = transport_layout(probe)
        assert layout == {}, "only Driving Full observations have a compacted form"
    else:
        layout = dict(split=shared_tail_split(probe)) if transport == "tail" else {}
    probe.close()
    slabs = [PackedSlab(torch, torch.device("cuda:0"), E, T, A, D, **layout) for _ in range(3)]
    gather = StepGather(torch, dist, slabs[0], more=slabs[1:])
    env = BatchedDynEnv(et, E, n, seed=6, device="cuda:0", out_buffers=(slabs[0].obs, slabs[0].rewards, slabs[0].dones), **kw)
    ref = BatchedDynEnv(et, E, n, seed=6, device="cuda:0", **kw)
    env.reset_flat(); ref.reset_flat()
    rng = np.random.default_rng(1)
    hi = (5, 3, 3, 7) if robocup else (3, 3)
    handles, want = [], []
    for k in range(steps):
        a = torch.tensor(np.stack([rng.integers(0, h, (E, A)) for h in hi], -1).astype(np.int32), device="cuda:0")
        gather.release(k)
        sl = gather.slabs[k % 3]
        env.use_buffers(sl.obs, sl.rewards, sl.dones)
        env.step_flat(a, auto_reset=False)
        handles.append(gather.start(k))
        o, r, d = ref.step_flat(a, auto_reset=False)
        want.append((o.clone(), r.clone()))
        if k >= 1:
            go, gr, gd = handles[k - 1].wait()
            assert torch.equal(go[0], want[k - 1][0]) and torch.equal(gr[0], want[k - 1][1]), "step %d" % (k - 1)
    go, gr, gd = handles[-1].wait()
    assert torch.equal(go[0], want[-1][0]) and torch.equal(gr[0], want[-1][1])
    gather.drain()
    env.close(); ref.close()


def _two_rank_worker(rank, world, port, kind, out_dir):
    """one of two processes that share the box's single GPU: the process group is gloo (RCCL refuses two ranks on one device),
    everything else - slabs in HBM, pack kernels, side stream, ring, unpack for G = 2 - is the path the 8-GPU run takes"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynenv_amd import BatchedDynEnv, DynEnvType
    from dynenv_amd.distributed import ShardedDynEnv, shard_range
    robocup = kind == "ROBO_CUP"
    et = DynEnvType.ROBO_CUP if robocup else DynEnvType.DRIVE
    total, n, steps = int(os.environ.get("SHARD_TEST_TOTAL", "48")), 5 if robocup else 10, 14
    sh = ShardedDynEnv(et, total, n, gather=True, seed=9, device="cuda:0")
    off, per = shard_range(total, rank, world)
    A = sh.env.n_agents
    rng = np.random.default_rng(5)
    hi = (5, 3, 3, 7) if robocup else (3, 3)
    acts = [np.stack([rng.integers(0, h, (total, A)) for h in hi], -1).astype(np.int32) for _ in range(steps)]
    ref = BatchedDynEnv(et, total, n, seed=9, device="cuda:0") if rank == 0 else None
    g_obs, _, _ = sh.reset()
    ok = True
    if rank == 0:
        ok = ok and torch.equal(g_obs.reshape(ref.obs.shape), ref.reset_flat())
    handles, want = [], []
    for k in range(steps):
        handles.append(sh.step(torch.tensor(acts[k][off:off + per], device="cuda:0"), wait=False))
        if rank == 0:
            o, r, d = ref.step_flat(torch.tensor(acts[k], device="cuda:0"), auto_reset=True)
            want.append((o.clone(), r.clone(), d.clone()))
        if k >= 2:  # consume two steps late
            go, gr, gd = handles[k - 2].wait()
            if rank == 0:
                w = want[k - 2]
                ok = ok and torch.equal(go.reshape(w[0].shape), w[0]) and torch.equal(gr.reshape(w[1].shape), w[1]) and torch.equal(gd.reshape(w[2].shape), w[2])
    for k in (steps - 2, steps - 1):
        go, gr, gd = handles[k].wait()
        if rank == 0:
            w = want[k]
            ok = ok and torch.equal(go.reshape(w[0].shape), w[0]) and torch.equal(gr.reshape(w[1].shape), w[1])
    sh.gather.drain()
    torch.cuda.synchronize()
    if rank == 0:
        with open(os.path.join(out_dir, "ok.txt"), "w") as f:
            f.write("ok" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["DRIVE", "ROBO_CUP"])
def test_two_ranks_sharing_the_gpu_match_one_big_batch(tmp_path, kind):
    """world_size 2 with BOTH ranks on this box's one GPU (gloo process group over device tensors): rank g owns environments
    [24 g, 24 g + 24), the gathered global views of the pipelined protocol equal a single 48-environment BatchedDynEnv bit for
    bit - shard-count invariance and the G = 2 transport (compacted for Driving, dense for RoboCup) on the real kernels"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "ok.txt")).read() == "ok"


def test_four_ranks_sharing_the_gpu_match_one_big_batch(tmp_path, monkeypatch):
    """the same with FOUR ranks of 512 environments each (the n-rank unpack kernel with more than one peer, the ring of
    slabs with three peers' data in flight) against a single 2048-environment handle, bit for bit.  Four processes on the card: inside
    the box's limit of six."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("SHARD_TEST_TOTAL", "2048")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(4, port, "DRIVE", str(tmp_path)), nprocs=4, join=True)
    assert open(os.path.join(str(tmp_path), "ok.txt")).read() == "ok"


def test_bench_rehearsal_with_four_ranks_on_this_gpu():
    """`python3 bench.py --gpus 4 --rehearse-one-gpu --envs 512`: the launcher with more than two children, every rank's preflight line
    (world size, device, environment ids, slab bytes) BEFORE anything is timed, one JSON line whose shard_check is green."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "8", "--warmup", "3", "--envs", "512", "--rehearse-one-gpu"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    sc = d["shard_check"]
    assert d["n_gpus"] == 4 and sc["rccl_world_size"] == 4 and sc["env_id_offset_per_rank"] == [0, 512, 1024, 1536] and sc["envs_per_rank"] == [512] * 4
    assert d["kernel_error_flags"] == 0 and min(sc["value_full_episode_per_rank"]) > 0
    assert len(sc["preflight"]) == 4 and all(p["world_size"] == 4 and p["slab_bytes"] == d["config"]["gather_bytes_per_rank"] for p in sc["preflight"])
    # (the four ranks write to one stderr: lines may run into each other)
    assert err.count("[bench preflight] rank") == 4 and err.count("of world_size 4") == 4, err[-1500:]
    assert abs(d["env_steps_per_s"] - 4 * 512 * 8 / (d["ms_per_step"] * 8e-3)) < 1e-6 * d["env_steps_per_s"]


@pytest.mark.parametrize("how", ["launcher", "as_typed"])
def test_bench_launched_like_the_driver_with_two_ranks_on_this_gpu(how):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` as the driver starts the scaling runs - and plain
    `python3 bench.py --gpus 2 ...` (the script then starts that launcher itself, as a child process) - with
    --rehearse-one-gpu (both ranks on cuda:0, gloo): one JSON line on stdout, n_gpus 2, all ranks' environments counted"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tail = ["bench.py", "--gpus", "2", "--steps", "8", "--warmup", "3", "--envs", "256", "--rehearse-one-gpu"]
    cmd = [sys.executable] + tail if how == "as_typed" else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port)] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 3 and d["scaling"] == "weak" and d["kernel_error_flags"] == 0
    assert d["config"]["rccl_world_size"] == 2 and "REHEARSAL" in d["config"]["parallelism"]
    assert abs(d["env_steps_per_s"] - 2 * 256 * 8 / (d["ms_per_step"] * 8e-3)) < 1e-6 * d["env_steps_per_s"]
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
    # the line proves its own sharding and carries every rank's N = 1-equivalent whole-episode figure
    sc = d["shard_check"]
    assert sc["rccl_world_size"] == 2 and sc["env_id_offset_per_rank"] == [0, 256] and sc["envs_per_rank"] == [256, 256]
    assert len(sc["value_full_episode_per_rank"]) == 2 and min(sc["value_full_episode_per_rank"]) > 0
    assert d["timed_region"]["spread_over_one_episode"] == 8 and d["timed_region"]["whole_episodes"] == 0
