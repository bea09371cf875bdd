#!/usr/bin/env python3
"""Upper bound for a forward-side dense block of the 64 hottest features: the same rows with those features' entries
REMOVED (hot block off), against the shipped layout.  us per launch (HIP events).

    python3 tools/fwd_bound_probe.py [config] [rows]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FeatureOrder, FMModel, _ffi, synth  # noqa: E402


def run(d, k, n1, label, hot_block=None, batch_rows=250_000):
    ds = DataSet(d["row_ptr"], d["col"], d["val"], d["y"], batch_rows=batch_rows, device=0, hot_block=hot_block).cache()
    fm = FMModel(n1 - 1, k, seed=3, device=0, init_on_device=True)
    L = _ffi.load()
    nb = ds.n_batches
    for j in range(8):
        _ffi.check(L.fmhip_sgd_step(fm.handle, ds.handle, j % nb, 0.02, 0.0, 1e-4, 1e-4, None))
    _ffi.check(L.fmhip_profile_begin(fm.handle))
    for j in range(16):
        _ffi.check(L.fmhip_sgd_step(fm.handle, ds.handle, j % nb, 0.02, 0.0, 1e-4, 1e-4, None))
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(fm.handle, C.byref(p)))
    kern = {n: round(v["ms"] / max(v["launches"], 1) * 1e3, 1) for n, v in p.as_dict().items() if isinstance(v, dict) and v["launches"]}
    out = {"rows": label, "nnz": int(d["row_ptr"][-1]), "kernels_us": kern}
    ds.unpersist()
    fm.close()
    return out


if __name__ == "__main__":
    config = sys.argv[1] if len(sys.argv) > 1 else "C3"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
    cfg = synth.CONFIGS[config]
    d = synth.make_config(config, rows=rows)
    n1, k = cfg["features"], cfg["k"]
    d["col"] = FeatureOrder.fit(d["col"], n1).relabel(d["col"])          # id = frequency rank
    print(json.dumps(run(d, k, n1, "as shipped")), flush=True)
    for top in (16, 64):
        keep = d["col"] >= top
        rp = np.zeros(len(d["row_ptr"]), np.int64)
        np.cumsum(np.add.reduceat(keep.astype(np.int64), d["row_ptr"][:-1].clip(max=len(keep) - 1)) * (np.diff(d["row_ptr"]) > 0), out=rp[1:])
        d2 = dict(row_ptr=rp, col=d["col"][keep], val=d["val"][keep], y=d["y"])
        print(json.dumps(run(d2, k, n1, "entries of the %d hottest features removed, no hot block" % top, hot_block=False)), flush=True)
