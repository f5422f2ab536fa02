"""A captured `dynenv_step` replays correctly: three steps captured with torch.cuda.graph, replayed 200 times =
one whole Driving episode, bit-identical to a handle stepped eagerly - observations, rewards and dones after every replay, state
blobs at the end - for Driving Full and Partial at 4096 environments (the sizes at which the per-step scheduling counters that a
replay would freeze are in use: SIMD isolation lists, the deferred-vision list's parity) and for RoboCup.  -m gpu."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(oracle_built):
    import torch
    import dynenv_amd
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return dynenv_amd


def _mk(gpu, kind, E, partial):
    kw = {}
    if partial:
        kw = dict(observationType=gpu.ObservationType.PARTIAL, noiseType=gpu.NoiseType.REALISTIC, noiseMagnitude=3)
    if kind == "driving":
        return gpu.BatchedDynEnv(gpu.DynEnvType.DRIVE, E, 10, seed=11, **kw)
    return gpu.BatchedDynEnv(gpu.DynEnvType.ROBO_CUP, E, 5, seed=11, **kw)


@pytest.mark.parametrize("kind,E,partial,replays", [("driving", 4096, False, 200), ("driving", 4096, True, 200), ("driving", 8192, False, 60),
                                                    ("robocup", 1024, False, 40), ("robocup", 1024, True, 40)])
def test_captured_steps_replay_like_eager_steps(gpu, kind, E, partial, replays):
    import torch
    dev = torch.device("cuda", 0)
    eager, graphed = _mk(gpu, kind, E, partial), _mk(gpu, kind, E, partial)
    A, K = eager.n_agents, eager.action_dim
    gen = torch.Generator(device=dev).manual_seed(5)
    hi = torch.tensor([3, 3] if kind == "driving" else [5, 3, 3, 7], device=dev)

    def draw():
        return (torch.rand((3, E, A, K), generator=gen, device=dev) * hi).to(torch.int32)
    eager.reset_flat()
    graphed.reset_flat()
    # a few eager steps on both first: the capture starts mid-episode, with scheduling state already in place
    for a in draw():
        eager.step_flat(a, auto_reset=False)
        graphed.step_flat(a, auto_reset=False)
    static_a = draw()
    outs = [(torch.zeros_like(graphed.obs), torch.zeros_like(graphed.rewards), torch.zeros_like(graphed.dones)) for _ in range(3)]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(3):
            graphed.use_buffers(*outs[k])
            graphed.step_flat(static_a[k], auto_reset=False)
    # (the capture itself executed nothing: the first replay performs steps 4..6)
    for r in range(replays):
        acts = draw()
        static_a.copy_(acts)
        g.replay()
        for k in range(3):
            o, rw, d = eager.step_flat(acts[k], auto_reset=False)
            assert torch.equal(o.view(torch.int32), outs[k][0].view(torch.int32)), (r, k, "observations")
            assert torch.equal(rw.view(torch.int64), outs[k][1].view(torch.int64)), (r, k, "rewards")
            assert torch.equal(d, outs[k][2]), (r, k, "dones")
    assert eager.error_flags() == 0 and graphed.error_flags() == 0
    for e in (0, 1, E // 2, E - 1):
        a, b = eager.get_state(e), graphed.get_state(e)
        assert bytes(a) == bytes(b), e
    # eager steps after the capture keep working on the same handle (the counters stay on the device)
    for a in draw():
        o, rw, d = eager.step_flat(a, auto_reset=False)
        graphed.use_buffers(*outs[0])
        o2, rw2, d2 = graphed.step_flat(a, auto_reset=False)
        assert torch.equal(o.view(torch.int32), o2.view(torch.int32)) and torch.equal(rw.view(torch.int64), rw2.view(torch.int64))
    if kind == "driving" and E == 4096 and not partial:
        dc = graphed.debug_counters()
        assert dc["isolation_mode"] == 1, dc          # isolation stayed on through the replays ...
        assert dc["isolation_timeouts"] == 0, dc      # ... and no placeholder ever waited for a tick that did not come
    eager.close()
    graphed.close()


def test_steps_without_observation_after_a_capture_keep_the_deferred_lists_in_step(gpu):
    """once a step of a Driving Partial handle has been captured the deferred-vision parity lives on the device; a step WITHOUT
    an observation buffer (include/dynenv.h allows it for Driving) must leave it alone, as the eager host does - else the list such a step
    skips is never cleared, fills up with stale entries and, after E of them, drops environments (error bit 2, rows not written)."""
    import torch
    from dynenv_amd import _capi
    dev = torch.device("cuda", 0)
    E = 4096
    eager, graphed = _mk(gpu, "driving", E, True), _mk(gpu, "driving", E, True)
    A, K = eager.n_agents, eager.action_dim
    gen = torch.Generator(device=dev).manual_seed(9)

    def draw():
        return torch.randint(0, 3, (E, A, K), generator=gen, device=dev, dtype=torch.int32)

    def step_no_obs(env, a):
        _capi.check(env._lib.dynenv_step(env._h, C.c_void_p(a.data_ptr()), None, C.c_void_p(env.rewards.data_ptr()),
                                         C.c_void_p(env.dones.data_ptr()), env._stream()), "dynenv_step")
        env._episode_step += 1
    eager.reset_flat()
    graphed.reset_flat()
    static_a = draw()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step_flat(static_a, auto_reset=False)
    g.replay()
    eager.step_flat(static_a, auto_reset=False)
    for it in range(150):
        a = draw()
        step_no_obs(eager, a)
        step_no_obs(graphed, a)
        assert torch.equal(eager.rewards.view(torch.int64), graphed.rewards.view(torch.int64)), it
        a = draw()
        o, rw, _ = eager.step_flat(a, auto_reset=False)
        o2, rw2, _ = graphed.step_flat(a, auto_reset=False)
        assert torch.equal(o.view(torch.int32), o2.view(torch.int32)), (it, "observations")
        assert torch.equal(rw.view(torch.int64), rw2.view(torch.int64)), (it, "rewards")
    assert eager.error_flags() == 0 and graphed.error_flags() == 0
    eager.close()
    graphed.close()


def test_auto_reset_inside_a_capture_is_refused(gpu):
    """The host's episode position is not advanced by a replay: an auto-reset decided at capture time would be frozen into the graph."""
    import torch
    env = _mk(gpu, "driving", 64, False)
    env.reset_flat()
    a = torch.zeros((64, env.n_agents, env.action_dim), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception, match="auto_reset"):
        with torch.cuda.graph(g, stream=s):
            env.step_flat(a, auto_reset=True)
    torch.cuda.synchronize()
    env.close()
