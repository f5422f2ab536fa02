"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/dynenv.h declares,
fails loudly without a GPU (no CPU fallback), and the product package never touches the oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from dynenv_amd import _capi, build
    build.build()
    return _capi


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "dynenv.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|size_t|char\s*\*|const char\s*\*)\s*\*?\s*(dynenv_\w+)\s*\(", txt, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.load()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libdynenv_hip.so does not export %s declared in include/dynenv.h" % n
    assert set(names) <= set(capi.EXPORTS) | {"dynenv_abi_version", "dynenv_last_error"}
    assert lib.dynenv_abi_version() == capi.DYNENV_ABI_VERSION == 3


def test_struct_sizes_match_the_header(capi):
    # sizes the C compiler gives the blob structs (compiled on the fly with gcc)
    src = '#include <stdio.h>\n#include "dynenv.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(dynenv_cfg_t), ' \
          'sizeof(dynenv_layout_t), sizeof(dynenv_car_state_t), sizeof(dynenv_ped_state_t), sizeof(dynenv_driving_state_t), ' \
          'sizeof(dynenv_robocup_state_t));return 0;}'
    exe = "/tmp/dynenv_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I" + os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    sizes = [int(x) for x in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(capi.Cfg), C.sizeof(capi.Layout), C.sizeof(capi.CarState), C.sizeof(capi.PedState),
                     C.sizeof(capi.DrivingState), C.sizeof(capi.RoboCupState)]


def test_no_gpu_means_loud_failure_not_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    lib = capi.load()
    cfg = capi.Cfg(capi.DYNENV_ABI_VERSION, 1, 4, 10, 0, 1, 0.0, 42, 0, 0, 0, 0)
    h = C.c_void_p()
    rc = lib.dynenv_create(C.byref(cfg), C.byref(h))
    assert rc == capi.ERR_NO_DEVICE and not h
    assert b"no CPU fallback" in lib.dynenv_last_error()
    import dynenv_amd
    with pytest.raises(capi.DynEnvError):
        dynenv_amd.BatchedDynEnv(dynenv_amd.DynEnvType.DRIVE, 4, 10)
    with pytest.raises(capi.DynEnvError):
        dynenv_amd.make_dyn_env(dynenv_amd.DynEnvType.DRIVE, 4, 10, False, dynenv_amd.ObservationType.FULL,
                                dynenv_amd.NoiseType.REALISTIC, 0, False)


def test_argument_validation(capi):
    lib = capi.load()
    h = C.c_void_p()
    bad = capi.Cfg(99, 1, 4, 10, 0, 1, 0.0, 42, 0, 0, 0, 0)
    assert lib.dynenv_create(C.byref(bad), C.byref(h)) == -1 and b"ABI" in lib.dynenv_last_error()
    bad = capi.Cfg(capi.DYNENV_ABI_VERSION, 1, 0, 10, 0, 1, 0.0, 42, 0, 0, 0, 0)
    assert lib.dynenv_create(C.byref(bad), C.byref(h)) == -1
    bad = capi.Cfg(capi.DYNENV_ABI_VERSION, 1, 4, 10, 0, 1, 7.0, 42, 0, 0, 0, 0)  # environment_base.py:162-164
    assert lib.dynenv_create(C.byref(bad), C.byref(h)) == -1 and b"noise magnitude" in lib.dynenv_last_error()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "dynenv_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"oracle_lib|liboracle|/oracle/|import oracle|from oracle", txt):
                    offenders.append(os.path.join(d, f))
    assert not offenders, "product files reference the oracle: %s" % offenders


def test_enums_and_spaces_mirror_the_reference():
    from dynenv_amd import DynEnvType, NoiseType, ObservationType
    from dynenv_amd import spaces as sp
    from dynenv_amd.vec_env import _driving_spaces, _robocup_spaces
    assert (int(DynEnvType.ROBO_CUP), int(DynEnvType.DRIVE)) == (0, 1)
    assert (int(ObservationType.FULL), int(ObservationType.PARTIAL), int(ObservationType.IMAGE)) == (0, 1, 2)
    assert DynEnvType.argparse("drive") is DynEnvType.DRIVE and str(NoiseType.REALISTIC) == "realistic"
    obs, act, reco = _driving_spaces(ObservationType.FULL)
    assert act.spaces[0].nvec.tolist() == [3, 3] and reco.featureGridSize == (10, 17)
    assert list(obs.spaces[1].spaces[1].spaces) == ["points", "type"]       # Full lanes: 4 points + type
    assert sp.flatdim(obs.spaces[0].spaces[0]) == 7 and sp.flatdim(obs.spaces[1].spaces[0]) == 9
    obs, act, reco = _robocup_spaces(ObservationType.FULL, False)
    assert act.spaces[0].nvec.tolist() == [5, 3, 3, 7]
    assert sp.flatdim(obs.spaces[0].spaces[0]) == 4 and sp.flatdim(obs.spaces[0].spaces[1]) == 6
    assert sp.flatdim(obs.spaces[1].spaces[0]) == 8


def test_shard_ranges():
    from dynenv_amd.distributed import shard_range
    assert [shard_range(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 4096)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 3)


def test_lazy_info_behaves_like_the_eager_dict():
    """'Full State' / 'Recon States' are built on first access, the dict protocol does not notice"""
    from dynenv_amd.vec_env import LazyInfo
    calls = []

    def make():
        calls.append(1)
        return ["full"], ["recon"]
    d = LazyInfo(make, {"episode_r": 1.5})
    assert "episode_r" in d.keys.__self__ and not calls  # touching the eager part builds nothing
    assert "Full State" in d and "Recon States" in d and not calls
    assert d["episode_r"] == 1.5 and not calls
    assert d["Recon States"] == ["recon"] and calls == [1]
    assert d["Full State"] == ["full"] and calls == [1]
    assert set(d.keys()) == {"episode_r", "Full State", "Recon States"} and len(d) == 3
    d2 = LazyInfo(make)
    assert d2.get("Full State") == ["full"] and d2.get("nope", 7) == 7
    d3 = LazyInfo(make)
    assert sorted(k for k in d3) == ["Full State", "Recon States"]
    with pytest.raises(KeyError):
        d3["episode_r"]


def test_packed_slab_layouts_on_cpu():
    """Transport formats of the multi-GPU slab: dense, shared-tail de-duplicated, peer-compacted (sizes and views only;
    the pack / expand kernels are GPU tests)."""
    import torch
    from dynenv_amd.distributed import PackedSlab
    E, T, A, D = 8, 1, 10, 232
    dense = PackedSlab(torch, torch.device("cpu"), E, T, A, D)
    tail = PackedSlab(torch, torch.device("cpu"), E, T, A, D, split=72)
    peers = PackedSlab(torch, torch.device("cpu"), E, T, A, D, peers=True)
    assert (dense.row, tail.row, peers.row) == (A * D, A * 72 + 160, A * 9 + 160)
    assert dense.obs.shape == (E, T, A, D) and dense.obs.data_ptr() == dense.buf.data_ptr()  # the kernel writes the slab itself
    for s in (tail, peers):
        assert s.packed and s.obs.shape == (E, T, A, D) and s.obs.data_ptr() != s.buf.data_ptr()
    for s in (dense, tail, peers):
        assert s.rew_off % 256 == 0 and s.done_off % 256 == 0 and s.nbytes % 256 == 0
        assert s.rewards.shape == (E, A) and s.rewards.dtype == torch.float64 and s.dones.shape == (E,)
        g = torch.zeros((3 * s.nbytes,), dtype=torch.uint8)
        if not s.packed:
            o, r, d = s.gathered_views(g, 3)
            assert o.shape == (3, E, T, A, D) and r.shape == (3, E, A) and d.shape == (3, E)
    assert peers.nbytes < tail.nbytes < dense.nbytes


def test_lazy_obs_array_indexes_like_the_object_ndarray():
    """LazyObsArray (the compat observation of step() / reset()) follows numpy's basic indexing on [E, T, A, 3]"""
    from dynenv_amd.vec_env import LazyInfos, LazyObsArray

    class Owner(object):
        def _compat_element(self, dense, counts, e, t, a):
            return [("mov", e, t, a), ("stat", e, t, a), (1, 1, 1)]
    E, T, A = 4, 2, 3
    dense = np.zeros((E, T, A, 7), np.float32)
    lazy = LazyObsArray(Owner(), dense, None)
    ref = np.empty((E, T, A, 3), dtype=object)
    for e in range(E):
        for t in range(T):
            for a in range(A):
                for k, v in enumerate(Owner()._compat_element(dense, None, e, t, a)):
                    ref[e, t, a, k] = v
    assert lazy.shape == ref.shape and lazy.ndim == 4 and lazy.dtype == object and len(lazy) == E
    full = np.asarray(lazy)
    assert full.shape == ref.shape and all(full[i] == ref[i] for i in np.ndindex(*ref.shape))
    cases = [(1, 0, 2, 1), (-1, -1, -1, -1), (Ellipsis, -1), (Ellipsis, slice(None, -1)), (slice(1, 3),), (2,), (slice(None), 1),
             (0, Ellipsis, 2), (slice(None, None, 2), slice(None), slice(1, None), 0), (Ellipsis,)]
    for idx in cases:
        got, exp = lazy[idx], ref[idx]
        if isinstance(exp, np.ndarray):
            assert got.shape == exp.shape, idx
            g = np.asarray(got)
            assert all(g[i] == exp[i] for i in np.ndindex(*exp.shape)), idx
        else:
            assert got == exp, idx
    assert lazy[1][0][2][0] == ref[1][0][2][0]
    assert [x.shape for x in lazy] == [(T, A, 3)] * E
    with pytest.raises(IndexError):
        lazy[E]
    with pytest.raises(IndexError):
        lazy[0, 0, 0, 0, 0]
    made = []
    infos = LazyInfos(5, lambda i: made.append(i) or {"i": i})
    assert len(infos) == 5 and not made
    assert infos[3]["i"] == 3 and infos[-2] is infos[3] and made == [3]
    assert [d["i"] for d in infos] == [0, 1, 2, 3, 4] and sorted(made) == [0, 1, 2, 3, 4]
    assert isinstance(infos[1:3], tuple) and len(infos[1:3]) == 2


@pytest.mark.parametrize("A", [10, 4, 2])
def test_compacted_transport_index_maps_on_real_observations(oracle_built, A):
    """The multi-GPU transport formats as index maps (dynenv_amd.distributed.pack_*_np / unpack_*_np, what the pack / unpack
    kernels implement): on real Driving Full observations pack -> unpack gives the dense tensor back bit for bit - i.e. the
    "other cars" block of every agent row IS columns {0..5, 8} of those cars' self blocks and the tail IS shared."""
    import oracle_lib as ol
    from dynenv_amd.distributed import pack_peers_np, pack_tail_np, unpack_peers_np, unpack_tail_np
    E = 6
    ora = ol.OracleEnv(env_type=1, num_envs=E, n_players=A, seed=17)
    obs = ora.reset().copy()
    rng = np.random.default_rng(4)
    for s in range(25):
        obs, _, _ = ora.step(rng.integers(0, 3, (E, A, 2)).astype(np.int32))
    D = ora.D
    o = obs.reshape(E, A, D)
    split = 9 + (A - 1) * 7
    p = pack_peers_np(o)
    assert p.shape == (E, A * 9 + (D - split))
    np.testing.assert_array_equal(unpack_peers_np(p, A, D), o)
    t = pack_tail_np(o, split)
    assert t.shape == (E, A * split + (D - split))
    np.testing.assert_array_equal(unpack_tail_np(t, A, D, split), o)


def test_transport_layout_falls_back_to_dense_where_rows_are_not_redundant():
    """only Driving Full has the redundancy the compacted formats exploit; everything else travels dense"""
    from dynenv_amd import DynEnvType, ObservationType
    from dynenv_amd.distributed import shared_tail_split, transport_layout

    class Probe(object):
        def __init__(self, env_type, obs, A=10):
            self.env_type, self.observationType, self.n_agents = env_type, obs, A

            class L(object):
                block_offset = [0, 9, 9 + (A - 1) * 7, 0, 0, 0, 0, 0]
            self.layout = L()
    assert transport_layout(Probe(DynEnvType.DRIVE, ObservationType.FULL)) == dict(peers=True)
    assert shared_tail_split(Probe(DynEnvType.DRIVE, ObservationType.FULL)) == 72
    for et, ob in ((DynEnvType.DRIVE, ObservationType.PARTIAL), (DynEnvType.ROBO_CUP, ObservationType.FULL),
                   (DynEnvType.ROBO_CUP, ObservationType.PARTIAL)):
        assert transport_layout(Probe(et, ob)) == {} and shared_tail_split(Probe(et, ob)) is None


def _deep_same(a, b):
    if isinstance(a, (list, tuple)):
        assert type(a) is type(b) and len(a) == len(b), (type(a), type(b))
        for x, y in zip(a, b):
            _deep_same(x, y)
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
        np.testing.assert_array_equal(a, b)
    else:
        assert type(a) is type(b) and a == b, (a, b)


@pytest.mark.parametrize("env_type,n,partial", [(1, 10, False), (1, 4, True), (0, 5, False), (0, 3, True)])
def test_bulk_compat_builder_equals_the_loop_builder(oracle_built, env_type, n, partial):
    """the reference's ragged object array [E, T, A, 3] built without a Python loop over agents (vec_env._compat_obs_bulk) is
    element for element - container types, dtypes, shapes, values - what the plain triple loop (_compat_obs) builds, on real
    observations of all four layouts (the oracle's: same dense layout as the library's)"""
    import oracle_lib as ol
    from dynenv_amd.enums import DynEnvType, ObservationType
    from dynenv_amd.vec_env import BatchedDynEnv
    E = 9
    kw = dict(obs_type=1, noise_type=1, noise_magnitude=3.0) if partial else {}
    ora = ol.OracleEnv(env_type=env_type, num_envs=E, n_players=n, seed=3, flags=ol.ROBOCUP_DEFAULT_FLAGS if env_type == 0 else 0, **kw)
    ora.reset()
    rng = np.random.default_rng(0)
    hi = (5, 3, 3, 7) if env_type == 0 else (3, 3)
    for _ in range(6):
        obs, _, _ = ora.step(np.stack([rng.integers(0, k, (E, ora.A)) for k in hi], -1).astype(np.int32))
    host = BatchedDynEnv.__new__(BatchedDynEnv)   # the builders only read the layout (no device, no handle)
    host.layout, host.env_type = ora.layout, DynEnvType(env_type)
    host.observationType = ObservationType.PARTIAL if partial else ObservationType.FULL
    host.n_agents, host.obs_dim, host.closed = ora.A, ora.D, True
    counts = ora.counts()
    loop = host._compat_obs(obs.copy(), counts)
    bulk = host._compat_obs_bulk(obs.copy(), counts)
    assert bulk.shape == loop.shape == (E, ora.T, ora.A, 3) and bulk.dtype == object
    for idx in np.ndindex(*loop.shape):
        _deep_same(bulk[idx], loop[idx])
    if not partial and env_type == 1:
        assert len({bulk[e, 0, 0, 0][1].shape[0] for e in range(E)}) > 1, "ragged obstacle lists must differ between environments"


def test_bench_reports_traffic_only_for_the_kernels_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/pmc_traffic.json (a PMC pass cannot run inside bench.py): the file is stamped with the
    sha256 of the kernel sources it was measured on, and a figure measured on other sources is reported as null with the reason"""
    import importlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    sha = bench.kernel_source_sha()
    assert len(sha) == 16 and int(sha, 16) >= 0
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert "kernel_source_sha16" in committed and "drv_step_kernel_bytes_per_launch" in committed
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: sha)
    (prof / "pmc_traffic.json").write_text(json.dumps({"kernel_source_sha16": sha, "drv_step_kernel_bytes_per_launch": 123.0,
                                                        "drv_step_kernel_detail": {"step_bytes_all_kernels": 456.0}}))
    t, d = bench.measured_traffic("drv_step_kernel")
    assert t == 123.0 and d["traffic_whole_step"] == 456.0 and sha in d["traffic_source"]
    (prof / "pmc_traffic.json").write_text(json.dumps({"kernel_source_sha16": "0" * 16, "drv_step_kernel_bytes_per_launch": 123.0}))
    t, d = bench.measured_traffic("drv_step_kernel")
    assert t is None and "STALE" in d["traffic_source"]
    (prof / "pmc_traffic.json").unlink()
    assert bench.measured_traffic("drv_step_kernel") == (None, {"traffic_source": None})


def test_bench_timed_steps_weigh_an_episode():
    """the K timed steps of bench.py are a stratified sample of an episode (or whole episodes), never its cheap start"""
    sys.path.insert(0, ROOT)
    import bench
    for n, L in ((20, 600), (1, 600), (7, 240), (599, 600), (600, 600)):
        at = bench.spread_positions(n, L)
        assert len(at) == n and all(0 <= p < L for p in at) and at == sorted(at) and len(set(at)) == n
        if 2 * n <= L:  # every one of the n equal slices of the episode holds exactly one timed step
            assert [int(p * n / L) for p in at] == list(range(n))
    assert bench.spread_positions(20, 600)[:3] == [15, 45, 75]
    for n, L in ((20, 600), (3, 600), (8, 240), (1, 600)):
        blocks = bench.spread_blocks(n, L)
        assert sum(k for _, k in blocks) == n and all(0 <= a and a + k <= L for a, k in blocks)
        assert all(blocks[i][0] + blocks[i][1] <= blocks[i + 1][0] for i in range(len(blocks) - 1))
