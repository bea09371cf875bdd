#!/usr/bin/env python3
"""What bounds the forward: the C3 rows with their cold feature ids (rank >= 64) folded into fewer and fewer ids, so the
same number of 128-B V-row gathers is served from HBM/Infinity Cache, from L2, or from L1.  Forward and backward
times per launch from the library's HIP events.   python3 tools/fwd_locality_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

cfg = synth.CONFIGS["C3"]
base = synth.make_config("C3", rows=1_000_000)
n1, k = cfg["features"], cfg["k"]
L = _ffi.load()
for fold in (0, 32768, 4096, 256):
    d = dict(base)
    if fold:
        c = base["col"]
        d["col"] = np.where(c < 64, c, 64 + (c - 64) % fold).astype(c.dtype)
    ds = DataSet.from_arrays(d, batch_rows=250_000).cache()
    fm = FMModel(n1 - 1, k, seed=3, init_on_device=True)
    hm, hd = fm.handle, ds.handle
    for j in range(8):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j % 4, 0.02, 0.0, 1e-4, 1e-4, None))
    _ffi.check(L.fmhip_profile_begin(hm))
    for j in range(40):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j % 4, 0.02, 0.0, 1e-4, 1e-4, None))
    p = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
    us = [p.ms[i] / max(p.launches[i], 1) * 1e3 for i in range(4)]
    print("cold ids folded into %6s: forward %.1f us, backward %.1f, then %.1f / %.1f" % (fold or "none", us[0], us[1], us[2], us[3]), flush=True)
    fm.close(discard=True)
    ds.unpersist()
