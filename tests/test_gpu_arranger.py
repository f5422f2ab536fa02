"""GPU arranger (dynenv_arrange_* through the C ABI, dynenv_amd.GpuInOutArranger) against
  (1) the vectors the reference's own InOutArranger produced (tests/golden/arranger.npz), and
  (2) the numpy oracle (oracle/arranger.py) on real observations of the three GPU configs."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import arranger as oa  # noqa: E402
from test_oracle_golden_arranger import CASES, G, ragged_from_golden  # noqa: E402

pytestmark = pytest.mark.gpu


def dense_from_ragged(x, feats, caps):
    """pack ragged x[env][time][player][type] into a dense row: blocks of cap_i * feat_i floats, counts in the row tail"""
    E, T, A, nT = len(x), len(x[0]), len(x[0][0]), len(feats)
    offs = np.concatenate([[0], np.cumsum([c * f for c, f in zip(caps, feats)])]).astype(int)
    D = int(offs[-1]) + nT
    o = np.full((E, T, A, D), 7.5, np.float32)  # junk in the padding: the arranger must not read it
    for e in range(E):
        for t in range(T):
            for a in range(A):
                for i in range(nT):
                    r = np.asarray(x[e][t][a][i], np.float32)
                    o[e, t, a, offs[i]:offs[i] + r.size] = r.reshape(-1)
                    o[e, t, a, D - nT + i] = len(r)
    return o, offs, D


@pytest.mark.parametrize("dense_min", ["0", "2"])   # both forms of rearrange_outputs: one pass over the padded tensor / zero fill + scatter
@pytest.mark.parametrize("name", CASES)
def test_gpu_arranger_reproduces_reference_vectors(name, dense_min, monkeypatch):
    import torch
    monkeypatch.setenv("DYNENV_ARR_DENSE_MIN", dense_min)
    from dynenv_amd import GpuInOutArranger, _capi
    x, (E, T, A, nT) = ragged_from_golden(name)
    feats = [int(f) for f in G[name + "/feats"]]
    caps = [int(G[name + "/x_counts"][..., i].max()) + 1 for i in range(nT)]
    o, offs, D = dense_from_ragged(x, feats, caps)
    types = [_capi.ArrType(int(offs[i]), feats[i], caps[i], _capi.ARR_COUNT_ROW, 0, D - nT + i, 0, 0) for i in range(nT)]
    arr = GpuInOutArranger(types, E, A, T, D)
    obs = torch.tensor(o, device="cuda")
    inputs, countArr = arr.rearrange_inputs(obs)
    counts, maxCount, objCounts = countArr[:3]
    assert maxCount == int(G[name + "/maxCount"])
    assert np.array_equal(counts.cpu().numpy(), G[name + "/counts"])
    assert np.array_equal(objCounts.cpu().numpy(), G[name + "/objCounts"])
    outs = []
    for i in range(nT):
        got = inputs[i].cpu().numpy()
        assert np.array_equal(got, G["%s/inputs%d" % (name, i)]), "inputs of type %d" % i
        # the same fixed linear "embedding" the generator applied (host numpy, so that the comparison stays exact)
        outs.append(torch.tensor(got @ G["%s/W%d" % (name, i)], device="cuda") if got.size else None)
    padded, masks = arr.rearrange_outputs(outs, countArr)
    assert np.array_equal(padded.cpu().numpy(), G[name + "/padded"])
    assert np.array_equal(torch.stack(masks).cpu().numpy(), G[name + "/masks"])


def _ragged_from_compat(compat, group):
    E, T, A = compat.shape[:3]
    return [[[list(compat[e, t, a, group]) for a in range(A)] for t in range(T)] for e in range(E)]


@pytest.mark.parametrize("cfg", ["driving_full", "driving_partial", "robocup", "robocup_partial"])
def test_gpu_arranger_on_real_observations(cfg):
    import torch
    from dynenv_amd import BatchedDynEnv, DynEnvType, GpuInOutArranger, NoiseType, ObservationType, groups_for
    E = 6
    if cfg == "robocup":
        env = BatchedDynEnv(DynEnvType.ROBO_CUP, E, 2, seed=5)
        hi = [5, 3, 3, 7]
    elif cfg == "robocup_partial":
        env = BatchedDynEnv(DynEnvType.ROBO_CUP, E, 3, observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC,
                            noiseMagnitude=5, seed=5)
        hi = [5, 3, 3, 7]
    elif cfg == "driving_partial":
        env = BatchedDynEnv(DynEnvType.DRIVE, E, 5, observationType=ObservationType.PARTIAL, noiseType=NoiseType.REALISTIC,
                            noiseMagnitude=3, seed=5)
        hi = [3, 3]
    else:
        env = BatchedDynEnv(DynEnvType.DRIVE, E, 5, seed=5)
        hi = [3, 3]
    env.reset_flat()
    rng = np.random.default_rng(1)
    for _ in range(7):
        act = np.stack([rng.integers(0, k, (E, env.n_agents)) for k in hi], -1).astype(np.int32)
        obs, _, _ = env.step_flat(torch.tensor(act, device="cuda"), auto_reset=False)
    count_env = env.counts() if env.env_type == DynEnvType.DRIVE else None
    compat = env._compat_obs(obs, count_env.cpu().numpy() if count_env is not None else None)
    for gi, gname in enumerate(("movable", "static")):
        types = groups_for(env)[gname]
        arr = GpuInOutArranger(types, E, env.n_agents, env.n_time_steps, env.obs_dim)
        inputs, countArr = arr.rearrange_inputs(obs.contiguous(), count_env)
        x = _ragged_from_compat(compat, gi)
        o_inputs, (o_counts, o_max, o_obj) = oa.rearrange_inputs(x, len(types), E * env.n_agents, env.n_time_steps)
        assert countArr[1] == o_max
        assert np.array_equal(countArr[0].cpu().numpy(), o_counts)
        assert np.array_equal(countArr[2].cpu().numpy(), o_obj)
        F = max(t.feat for t in types) + (0 if gi == 0 else 3)  # movable: as is (scatter path if odd); static: another width
        F = F if not cfg.startswith("robocup") else ((F + 3) // 4) * 4  # a multiple of 4 takes the fused pad kernel
        outs, o_outs = [], []
        for i in range(len(types)):
            got = inputs[i].cpu().numpy()
            want = np.asarray(o_inputs[i], np.float32).reshape(-1, types[i].feat)
            assert np.array_equal(got, want), (cfg, gname, i)
            emb = np.zeros((got.shape[0], F), np.float32)  # "embedding" = the features themselves, zero-padded to F
            emb[:, :got.shape[1]] = got
            outs.append(torch.tensor(emb, device="cuda") if got.size else None)
            o_outs.append(emb if got.size else None)
        padded, masks = arr.rearrange_outputs(outs, countArr)
        o_padded, o_masks = oa.rearrange_outputs(o_outs, (o_counts, o_max, o_obj))
        assert np.array_equal(padded.cpu().numpy(), o_padded)
        assert np.array_equal(torch.stack(masks).cpu().numpy(), np.stack(o_masks))
    env.close()
