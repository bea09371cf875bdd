#!/usr/bin/env python3
"""Experiment (r05_experiments.md section 8): what would STORING the rows of a batch in the forward's walk order buy?  The forward
walks every batch's rows longest first through an index (`order`): a wave's eight rows then read their extents (row_ptr), labels
and entries from eight unrelated places and write e and their P rows there.  Here the caller's rows are permuted BEFORE the
dataset is made — every batch of 250k rows sorted by length, longest first — so the library's order is the identity and those
accesses fall on neighbouring addresses: the upper bound of what a "length-sorted slab" layout inside the library could gain
from its per-row side, measured without building it.  C3's shape, kernels per step by HIP events.
    [FMHIP_TUNE=row_order=0] python3 tools/r05_presorted_rows.py [sorted|band|stored]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "stored"
cfg = synth.CONFIGS["C3"]
d = synth.make_config("C3", rows=1_000_000)
B = 250_000
if mode in ("sorted", "band"):
    # sorted: every batch by length; band: every sixteenth of a batch (a row band of the backward's band-affine plan: the bands
    # keep their rows, hence their share of the nonzeros) — run with FMHIP_TUNE=row_order=0: the forward then walks stored order
    rp = d["row_ptr"]
    n = len(rp) - 1
    length = np.diff(rp)
    chunk = B if mode == "sorted" else B // 16
    perm = np.concatenate([b0 + np.argsort(-length[b0:min(b0 + chunk, n)], kind="stable") for b0 in range(0, n, chunk)])
    new_len = length[perm]
    new_rp = np.concatenate([[0], np.cumsum(new_len)]).astype(np.int64)
    src = np.repeat(rp[:-1][perm], new_len) + (np.arange(new_rp[-1]) - np.repeat(new_rp[:-1], new_len))
    d = dict(d, row_ptr=new_rp, col=d["col"][src], val=d["val"][src], y=d["y"][perm])
L = _ffi.load()
ds = DataSet.from_arrays(d, batch_rows=B).cache()
w0, w, v = synth.init_params(cfg["seed"] + 1000, cfg["features"], cfg["k"])
fm = FMModel(cfg["features"] - 1, cfg["k"])
fm.w0, fm.w, fm.v = w0, np.random.default_rng(1).normal(0, 0.01, cfg["features"]), v
hm, hd = fm.handle, ds.handle
for rep in range(2):
    for j in range(600):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j % 4, 0.02, 0.0, 1e-4, 1e-4, None))
    _ffi.check(L.fmhip_synchronize(hm))
    _ffi.check(L.fmhip_profile_begin(hm))
    for j in range(80):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j % 4, 0.02, 0.0, 1e-4, 1e-4, None))
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
    us = {n: p.ms[i] / max(p.steps[i], 1) * 1e3 for i, n in enumerate(_ffi.KERNEL_NAMES) if p.launches[i]}
    st = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))
    print("rows %s (FMHIP_TUNE=%s): kernels per step (us) %s  sum %.1f | last batch mse %.6f" % (mode, os.environ.get("FMHIP_TUNE", ""), {n: round(x, 1) for n, x in us.items()}, sum(us.values()), st.sse / max(st.rows, 1)), flush=True)
