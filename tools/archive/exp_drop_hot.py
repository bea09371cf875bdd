#!/usr/bin/env python3
"""Experiment: kernel times on C3 with the H hottest features' entries removed from the sparse
streams (what would remain for the gather kernels if those features were handled as a dense block)."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.make_config("C3")
keep = d["col"] >= H
rows = np.repeat(np.arange(len(d["row_ptr"]) - 1), np.diff(d["row_ptr"]))
cnt = np.bincount(rows[keep], minlength=len(d["row_ptr"]) - 1)
d2 = dict(row_ptr=np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), col=d["col"][keep], val=d["val"][keep], y=d["y"])
print("H=%d: kept %d of %d entries (%.1f%%)" % (H, keep.sum(), len(keep), 100.0 * keep.mean()))
ds = DataSet.from_arrays(d2, batch_rows=250000).cache()
w0, w, v = synth.init_params(1, 100000, 32)
fm = FMModel(99999, 32); fm.w0, fm.w, fm.v = w0, w, v
L = _ffi.load(); hm, hd = fm.handle, ds.handle
res = []
for rnd in range(4):
    _ffi.check(L.fmhip_profile_begin(hm))
    for j in range(ds.n_batches):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j, 0.02, 0.0, 1e-4, 1e-4, None))
    p = _ffi.Profile(); _ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
    if rnd: res.append([p.ms[i] / max(p.launches[i], 1) * 1e3 for i in range(5)])
print("median us fwd/red/bwd/fix/apply:", np.round(np.median(np.array(res), axis=0), 1))
