#!/usr/bin/env python3
"""The arranger oracle (oracle/arranger.py) against the reference's OWN InOutArranger (DynEnv/models/models.py:208-274) on random ragged
observations - the population behind tests/golden/arranger.npz (five cases).  Build container only (it imports /root/reference).
   python3 tools/arranger_reference_fuzz.py [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import arranger as oa  # noqa: E402
import gen_golden_arranger as ga  # noqa: E402


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    mm = ga.ref_arranger_module()
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "11")))
    t0, bad, rows, empties = time.time(), 0, 0, 0
    for k in range(n):
        E, T, A, nT = int(rng.integers(1, 5)), int(rng.choice([1, 1, 2, 5])), int(rng.integers(1, 7)), int(rng.integers(1, 5))
        feats = tuple(int(x) for x in rng.integers(1, 10, nT))
        max_counts = tuple(int(x) for x in rng.integers(0, 10, nT))          # (0: a type that never has a row)
        x = ga.make_case(rng, E, T, A, feats, max_counts, allow_empty_type=bool(rng.random() < 0.2) and nT > 1)
        arr = mm.InOutArranger(nT, E * A, T)
        try:
            inputs, (counts, maxCount, objCounts) = arr.rearrange_inputs(x)
        except Exception as e:      # (the reference itself cannot arrange e.g. a batch without any row: not a case)
            empties += 1
            continue
        if not any(np.asarray(inp).size for inp in inputs):   # no row of any type in the whole batch (the reference returns an empty [T, 0, P, 0]): out of the domain
            empties += 1
            continue
        F = 6
        W = [rng.standard_normal((f, F)).astype(np.float32) for f in feats]
        outs = [torch.tensor(np.asarray(inp, np.float32).reshape(-1, f) @ w) if np.asarray(inp).size else None for inp, w, f in zip(inputs, W, feats)]
        padded, masks = arr.rearrange_outputs(outs, (counts, maxCount, objCounts), "cpu")
        oin, (oc, om, oo) = oa.rearrange_inputs(x, nT, E * A, T)
        ok = om == int(maxCount) and np.array_equal(oc, np.asarray(counts)) and np.array_equal(oo, np.asarray(objCounts))
        for i in range(nT):
            ok = ok and np.array_equal(np.asarray(oin[i], np.float32).reshape(-1, feats[i]), np.asarray(inputs[i], np.float32).reshape(-1, feats[i]))
        oouts = [np.asarray(inp, np.float32).reshape(-1, f) @ w if np.asarray(inp).size else None for inp, w, f in zip(oin, W, feats)]
        op, omk = oa.rearrange_outputs(oouts, (oc, om, oo))
        ok = ok and op.shape == tuple(padded.shape) and np.array_equal(op, padded.numpy()) and np.array_equal(np.stack(omk), np.stack([m.numpy() for m in masks]))
        bad += 0 if ok else 1
        rows += int(sum(np.asarray(c).sum() for c in oc)) if ok else 0
        if not ok:
            print("MISMATCH case", k, (E, T, A, nT), feats, max_counts)
    print("arranger: %d random ragged batches (1-4 envs x 1-5 time steps x 1-6 players x 1-4 object types, 0-9 rows each, empty types) through the reference's own "
          "InOutArranger and oracle/arranger.py: inputs per type, counts, maxCount, objCounts, padded tensor and masks identical in %d, %d mismatches, "
          "%d batches without any row (out of the domain)  (%.0f s)" % (n, n - bad - empties, bad, empties, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
