#!/usr/bin/env python3
"""Does the caller's feature-id order matter?  C3-shaped rows as generated (id = frequency rank) against the same
rows with the ids pushed through a random permutation: ms per SGD step and per kernel.

    python3 tools/id_order_time.py [config] [rows]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402


def run(d, k, n1, label, steps=60, batch_rows=250_000):
    ds = DataSet.from_arrays(d, name=label, batch_rows=batch_rows, device=0).cache()
    fm = FMModel(n1 - 1, k, seed=3, device=0, init_on_device=True)
    L = _ffi.load()
    nb = ds.n_batches

    def step(j):
        _ffi.check(L.fmhip_sgd_step(fm.handle, ds.handle, j % nb, 0.02, 0.0, 1e-4, 1e-4, None))
    for j in range(8):
        step(j)
    _ffi.check(L.fmhip_synchronize(fm.handle))
    t = time.perf_counter()
    for j in range(steps):
        step(j)
    _ffi.check(L.fmhip_synchronize(fm.handle))
    ms = (time.perf_counter() - t) / steps * 1e3
    _ffi.check(L.fmhip_profile_begin(fm.handle))
    for j in range(16):
        step(j)
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(fm.handle, C.byref(p)))
    kern = {n: round(v["ms"] / max(v["launches"], 1) * 1e3, 1) for n, v in p.as_dict().items() if isinstance(v, dict) and v["launches"]}
    out = {"ids": label, "ms_per_step": ms, "kernels_us": kern, "hot_ids": len(ds.layout()["hot_ids"])}
    ds.unpersist()
    fm.close()
    return out


if __name__ == "__main__":
    config = sys.argv[1] if len(sys.argv) > 1 else "C3"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
    cfg = synth.CONFIGS[config]
    d = synth.make_config(config, rows=rows)
    n1, k = cfg["features"], cfg["k"]
    print(json.dumps(run(d, k, n1, "as generated (%s)" % ("hashed slots" if cfg.get("criteo") else "id = frequency rank"))), flush=True)
    perm = np.random.Generator(np.random.PCG64(5)).permutation(n1).astype(np.int32)
    d2 = dict(d)
    d2["col"] = perm[d["col"]]
    print(json.dumps(run(d2, k, n1, "randomly permuted")), flush=True)
    # the remedy on the caller's side: relabel by descending frequency (a pure renaming of the features)
    cnt = np.bincount(d2["col"], minlength=n1)
    rank = np.empty(n1, np.int32)
    rank[np.argsort(-cnt, kind="stable")] = np.arange(n1, dtype=np.int32)
    d3 = dict(d2)
    d3["col"] = rank[d2["col"]]
    print(json.dumps(run(d3, k, n1, "permuted, then relabelled by frequency")), flush=True)
