#!/usr/bin/env python3
"""The forward alone, with and without its P-row stores: the training forward (inside fmhip_sgd_step, HIP events) against the
residual-only forward of fmhip_rmse (no P rows written), same rows, same model.  What the 32 MB of P stores per launch cost
the V gathers (they pass through the same L2s).   python3 tools/fwd_modes_time.py [C3|C2|C5] [batch rows]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C3"
batch_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
cfg = synth.CONFIGS[cfg_name]
rows = 1_000_000
d = synth.make_config(cfg_name, rows=rows)
n1, k = cfg["features"], cfg["k"]
L = _ffi.load()
ds = DataSet.from_arrays(d, batch_rows=batch_rows).cache()
nb = ds.n_batches - (1 if rows % batch_rows else 0)          # full batches only
fm = FMModel(n1 - 1, k, seed=3, init_on_device=True)
hm, hd = fm.handle, ds.handle
for j in range(8):
    _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, 0.02, 0.0, 1e-4, 1e-4, None))
_ffi.check(L.fmhip_profile_begin(hm))
for j in range(40):
    _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, 0.02, 0.0, 1e-4, 1e-4, None))
p = _ffi.Profile()
_ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
train_fwd = p.ms[0] / p.launches[0] * 1e3
r = C.c_double()
_ffi.check(L.fmhip_rmse(hm, hd, C.byref(r), None))
_ffi.check(L.fmhip_synchronize(hm))
t = time.perf_counter()
for _ in range(10):
    _ffi.check(L.fmhip_rmse(hm, hd, C.byref(r), None))
_ffi.check(L.fmhip_synchronize(hm))
rmse_pass = (time.perf_counter() - t) / 10 / 4 * 1e6
print("batch %d rows (%.2f row walks per slot of the persistent grid at Kp = 32): %.3f ns per row;" % (batch_rows, batch_rows / 40960.0, train_fwd * 1e3 / batch_rows), end=" ")
print("%s: training forward %.1f us per launch; residual-only forward (+ its reduce launch, wall clock / 4 batches) %.1f us" % (cfg_name, train_fwd, rmse_pass))
