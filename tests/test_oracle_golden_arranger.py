"""The arranger oracle (oracle/arranger.py) against vectors computed by the reference's own InOutArranger
(tests/golden/gen_golden_arranger.py -> arranger.npz): inputs per type, counts, maxCount, objCounts, padded output, masks."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import arranger as oa  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "arranger.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})


def ragged_from_golden(name):
    E, T, A, nT = (int(v) for v in G[name + "/shape"])
    cnt = G[name + "/x_counts"]
    pos = [0] * nT
    x = []
    for e in range(E):
        env = []
        for t in range(T):
            players = []
            for a in range(A):
                s = []
                for i in range(nT):
                    c = int(cnt[e, t, a, i])
                    rows = G["%s/x_rows%d" % (name, i)]
                    s.append(rows[pos[i]:pos[i] + c])
                    pos[i] += c
                players.append(s)
            env.append(players)
        x.append(env)
    return x, (E, T, A, nT)


@pytest.mark.parametrize("name", CASES)
def test_arranger_oracle_matches_reference(name):
    x, (E, T, A, nT) = ragged_from_golden(name)
    inputs, (counts, max_count, obj_counts) = oa.rearrange_inputs(x, nT, E * A, T)
    assert max_count == int(G[name + "/maxCount"])
    assert np.array_equal(counts, G[name + "/counts"])
    assert np.array_equal(obj_counts, G[name + "/objCounts"])
    feats = G[name + "/feats"]
    for i in range(nT):
        got = np.asarray(inputs[i], np.float32).reshape(-1, int(feats[i]))
        assert np.array_equal(got, G["%s/inputs%d" % (name, i)]), "inputs of type %d" % i
    outs = [inp @ G["%s/W%d" % (name, i)] if inp.size else None for i, inp in enumerate(inputs)]
    padded, masks = oa.rearrange_outputs(outs, (counts, max_count, obj_counts))
    assert padded.shape == G[name + "/padded"].shape
    assert np.array_equal(padded, G[name + "/padded"])
    assert np.array_equal(np.stack(masks), G[name + "/masks"])
