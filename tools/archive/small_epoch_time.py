#!/usr/bin/env python3
"""How launch-bound is an SGD epoch on a small dataset?  Wall time per epoch (fmhip_sgd_epoch, back to back,
one sync at the end) against the sum of the kernels' own durations (HIP events, fmhip_profile_*).

    python3 tools/small_epoch_time.py
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, HipSGD, _ffi, synth  # noqa: E402


def run(rows, features, k, nnz, batch_rows, epochs=50, regs=(0.0, 1e-4, 1e-4)):
    d = synth.make_zipf(7, rows, features, nnz, nnz, zipf_s=0.0 if features <= 1000 else 1.05)
    d["k"] = k
    ds = DataSet.from_arrays(d, name="small", batch_rows=batch_rows, device=0).cache()
    fm = FMModel(features - 1, k, seed=3, device=0)
    sgd = HipSGD(eta=0.02, reg0=regs[0], regw=regs[1], regv=regs[2])
    L = _ffi.load()
    for _ in range(3):
        sgd.learn(fm, ds)
    _ffi.check(L.fmhip_synchronize(fm.handle))
    t = time.perf_counter()
    for _ in range(epochs):
        sgd.learn(fm, ds)
    _ffi.check(L.fmhip_synchronize(fm.handle))
    wall = (time.perf_counter() - t) / epochs
    _ffi.check(L.fmhip_profile_begin(fm.handle))
    for _ in range(4):
        sgd.learn(fm, ds)
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(fm.handle, C.byref(p)))
    pd = p.as_dict()
    kern = sum(v["ms"] for v in pd.values() if isinstance(v, dict)) / 4
    out = {"rows": rows, "features": features, "k": k, "batch_rows": batch_rows, "steps_per_epoch": ds.n_batches,
           "epoch_wall_us": wall * 1e6, "epoch_kernels_us": kern * 1e3, "mse": sgd.last_stats["sse"] / rows}
    ds.unpersist()
    fm.close()
    return out


if __name__ == "__main__":
    for args in ((10_000, 1000, 8, 10, 1000), (10_000, 1000, 8, 10, 10_000), (100_000, 10_000, 16, 20, 10_000),
                 (100_000, 10_000, 16, 20, 100_000), (1_000_000, 100_000, 32, 40, 250_000)):
        print(json.dumps(run(*args)), flush=True)
