#!/usr/bin/env python3
"""Golden vectors for the ragged->padded arranger, computed by the REFERENCE's own InOutArranger
(/root/reference/DynEnv/models/models.py:208-274), imported here with stand-ins for the third-party modules it pulls in
but does not use on this path (gym, imageio, matplotlib; pymunk & co. via gen_golden.install_standins).  Inputs are small
synthetic ragged observations of the shapes the three environments emit; outputs are what the reference returned.
Run in the build container only (the reference never ships):  python tests/golden/gen_golden_arranger.py
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def ref_arranger_module():
    gg.install_standins()
    for n in ("imageio", "matplotlib", "matplotlib.pyplot", "mpl_toolkits", "mpl_toolkits.axes_grid1",
              "mpl_toolkits.axes_grid1.inset_locator"):
        if n not in sys.modules:
            try:
                importlib.import_module(n)
            except Exception:
                sys.modules[n] = types.ModuleType(n)
    il = sys.modules["mpl_toolkits.axes_grid1.inset_locator"]
    if not hasattr(il, "zoomed_inset_axes"):
        il.zoomed_inset_axes = None
    sys.modules["gym.spaces"].flatdim = lambda s: 0
    for sub in ("models", "utils"):
        m = types.ModuleType("DynEnv." + sub)
        m.__path__ = [os.path.join(gg.REF, "DynEnv", sub)]
        sys.modules["DynEnv." + sub] = m
    return importlib.import_module("DynEnv.models.models")


def make_case(rng, E, T, A, feats, max_counts, allow_empty_type=False):
    """x[env][time][player] = list of per-type arrays [count, feat] (float32), counts ragged per (env, time, player)"""
    x = []
    for _ in range(E):
        env = []
        for _t in range(T):
            players = []
            for _a in range(A):
                s = []
                for i, (f, mc) in enumerate(zip(feats, max_counts)):
                    c = 0 if (allow_empty_type and i == 1) else int(rng.integers(0, mc + 1))
                    s.append(rng.standard_normal((c, f)).astype(np.float32))
                players.append(s)
            env.append(players)
        x.append(env)
    return x


def main():
    import torch
    mm = ref_arranger_module()
    rng = np.random.default_rng(7)
    cases = {
        "driving_full": dict(E=3, T=1, A=4, feats=(7, 4, 2), max_counts=(3, 6, 5)),
        "driving_partial": dict(E=2, T=1, A=5, feats=(7, 6, 2), max_counts=(6, 8, 9)),
        "robocup": dict(E=2, T=5, A=4, feats=(4, 6), max_counts=(1, 3)),
        "static": dict(E=2, T=1, A=3, feats=(9, 5), max_counts=(1, 8)),
        "empty_type": dict(E=2, T=2, A=3, feats=(7, 4, 2), max_counts=(2, 3, 4), allow_empty_type=True),
    }
    out = {}
    F = 8
    for name, kw in cases.items():
        x = make_case(rng, **kw)
        E, T, A = kw["E"], kw["T"], kw["A"]
        nT = len(kw["feats"])
        arr = mm.InOutArranger(nT, E * A, T)
        inputs, (counts, maxCount, objCounts) = arr.rearrange_inputs(x)
        # a fixed linear "embedding" per type so that rearrange_outputs has something type-specific to place
        W = [rng.standard_normal((f, F)).astype(np.float32) for f in kw["feats"]]
        outs = [torch.tensor(inp @ w) if inp.size else None for inp, w in zip(inputs, W)]
        padded, masks = arr.rearrange_outputs(outs, (counts, maxCount, objCounts), "cpu")
        out[name + "/shape"] = np.array([E, T, A, nT], np.int64)
        out[name + "/feats"] = np.array(kw["feats"], np.int64)
        # the ragged input, flattened: per (env, time, player, type) the count, and all rows of a type concatenated in
        # generation order (env, time, player) -- NOT the arranger's order
        cnt = np.array([[[[len(x[e][t][a][i]) for i in range(nT)] for a in range(A)] for t in range(T)] for e in range(E)], np.int64)
        out[name + "/x_counts"] = cnt
        for i in range(nT):
            rows = [x[e][t][a][i] for e in range(E) for t in range(T) for a in range(A) if len(x[e][t][a][i])]
            out["%s/x_rows%d" % (name, i)] = np.concatenate(rows) if rows else np.zeros((0, kw["feats"][i]), np.float32)
            out["%s/W%d" % (name, i)] = W[i]
            out["%s/inputs%d" % (name, i)] = np.asarray(inputs[i], np.float32).reshape(-1, kw["feats"][i])
        out[name + "/counts"] = np.array(counts, np.int64)
        out[name + "/maxCount"] = np.array(maxCount, np.int64)
        out[name + "/objCounts"] = objCounts.numpy()
        out[name + "/padded"] = padded.numpy()
        out[name + "/masks"] = np.stack([m.numpy() for m in masks])
    np.savez_compressed(os.path.join(HERE, "arranger.npz"), **out)
    print("wrote arranger.npz:", {k: v.shape for k, v in out.items() if k.endswith("/padded")})


if __name__ == "__main__":
    main()
