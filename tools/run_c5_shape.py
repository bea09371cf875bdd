#!/usr/bin/env python3
"""Smoke/scale check at the C5 shape (Criteo-like width): 2^25 hashed features, k=64, so V is 8.6 GB
(> 4 GiB: the kernels take their flat-load paths) and the packed gradient 8.9 GB.  Rows are a flag."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    n1, k = 1 << 25, 64
    t = time.time()
    d = synth.make_zipf(20261008, rows, n1, 25, 39, zipf_s=1.2)
    print("generated %d rows, %d nnz in %.1f s" % (rows, d["row_ptr"][-1], time.time() - t), flush=True)
    t = time.time()
    ds = DataSet.from_arrays(d, batch_rows=250000).cache()
    print("dataset on device in %.1f s: %s" % (time.time() - t, ds.info()), flush=True)
    fm = FMModel(n1 - 1, k, seed=1)
    L = _ffi.load()
    t = time.time()
    hm, hd = fm.handle, ds.handle
    print("model on device in %.1f s" % (time.time() - t), flush=True)
    r0 = fm.computeRMSE(ds)
    _ffi.check(L.fmhip_profile_begin(hm))
    st = _ffi.Stats()
    t = time.time()
    for _ in range(3):
        _ffi.check(L.fmhip_sgd_epoch(hm, hd, 0.05, 0.0, 0.0, 0.0, None, C.byref(st)))
    dt = time.time() - t
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
    fm._device_updated()
    r1 = fm.computeRMSE(ds)
    print("3 epochs in %.3f s (%.2f G nnz/s); rmse %.4f -> %.4f; nonfinite %d" %
          (dt, 3 * d["row_ptr"][-1] / dt / 1e9, r0, r1, st.nonfinite))
    print({k2: round(v["ms"] / max(v["launches"], 1) * 1e3, 1) for k2, v in p.as_dict().items()})
    assert r1 < r0 and st.nonfinite == 0


if __name__ == "__main__":
    main()
