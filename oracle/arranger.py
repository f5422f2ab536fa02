"""ORACLE (test infrastructure) — numpy restatement of the reference's ragged->padded arranger.

Follows DynEnv/models/models.py: InOutArranger.rearrange_inputs (:219-250), InOutArranger.rearrange_outputs (:252-274),
ObsMask.catAndPad (:147-163), ObsMask.createMask (:166-180), Indexer.getRange (:188-205).  Pinned against the reference's
own InOutArranger by tests/golden/gen_golden_arranger.py -> tests/golden/arranger.npz (tests/test_oracle_golden_arranger.py).
Only tests/, __graft_entry__.smoke() and bench tools may import this module; the product path (dynenv_amd/arranger.py)
runs HIP kernels through the C ABI.
"""
import numpy as np


def rearrange_inputs(x, n_object_types, n_players, n_time):
    """x[env][time][player] = sequence of per-type arrays [count, feat]  (the reference's obs[..., 0] / obs[..., 1]).
    Returns (inputs, (counts, maxCount, objCounts)) as models.py:219-250:
      inputs[type]   float array [N_type, feat_type]: all objects of the type, ordered (time, env*A+player, object);
                     np.array([]) if the type has no object at all
      counts         int array [type, time, P]     P = n_players = E*A
      objCounts      int array [time, P]           sum over types
      maxCount       int                            max of objCounts"""
    num_t = len(x[0])
    per_time = [[s for env in range(len(x)) for s in x[env][t]] for t in range(num_t)]  # :222-223 env-major chain
    counts = np.array([[[len(s[i]) for s in per_time[t]] for t in range(num_t)] for i in range(n_object_types)],
                      dtype=np.int64).reshape(n_object_types, num_t, -1)
    assert counts.shape[2] == n_players and num_t == n_time
    obj_counts = counts.sum(0)
    max_count = int(obj_counts.max())
    inputs = []
    for i in range(n_object_types):
        rows = [row for t in range(num_t) for s in per_time[t] if len(s[i]) for row in s[i]]  # :238-244
        inputs.append(np.stack(rows) if len(rows) else np.array([]))
    return inputs, (counts, max_count, obj_counts)


def rearrange_outputs(outs, count_arr):
    """outs[type] = per-object embeddings [N_type, F] in the order of rearrange_inputs (or None).
    Returns (padded [T, maxCount, P, F], masks [T][P, maxCount] bool) as models.py:252-274: per (time, player) the
    embeddings of its objects, types in order, zero-padded to maxCount; mask True where slot >= objCounts."""
    counts, max_count, obj_counts = count_arr
    n_types, n_time, n_players = counts.shape
    feat = next(o.shape[1] for o in outs if o is not None)
    dtype = next(o.dtype for o in outs if o is not None)
    padded = np.zeros((n_time, n_players, max_count, feat), dtype=dtype)
    prev = np.zeros(n_types, dtype=np.int64)  # Indexer.prev: running offsets in (time, player) order, :201-205
    for t in range(n_time):
        for p in range(n_players):
            off = 0
            for i in range(n_types):
                c = int(counts[i, t, p])
                if outs[i] is not None:
                    padded[t, p, off:off + c] = outs[i][prev[i]:prev[i] + c]
                    off += c
                prev[i] += c
    padded = padded.transpose(0, 2, 1, 3)  # :268 permute(0, 2, 1, 3)
    masks = [np.arange(max_count)[None, :] >= obj_counts[t][:, None] for t in range(n_time)]
    return padded, masks
