#!/usr/bin/env python3
"""Kernel times of the training step from the library's HIP events (forward, backward, fixup).
python3 tools/bwd_time.py [C3|C2|C5] [k (default: the config's)] [KEY=VALUE per-model tuning | gKEY=VALUE library-wide ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[cfg_name]
d = synth.make_config(cfg_name, rows=1_000_000)
L = _ffi.load()
for kv in sys.argv[2:]:
    if kv.startswith("g") and "=" in kv:                      # gKEY=VALUE: library-wide, before the dataset is built (hot block, pages)
        _ffi.check(L.fmhip_tune(int(kv[1:].split("=")[0]), int(kv.split("=")[1])))
ds = DataSet.from_arrays(d, batch_rows=250_000).cache()
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else cfg["k"]
fm = FMModel(cfg["features"] - 1, k, seed=3, init_on_device=True)
hm, hd = fm.handle, ds.handle
for kv in sys.argv[2:]:
    if "=" in kv and not kv.startswith("g"):
        _ffi.check(L.fmhip_model_tune(hm, int(kv.split("=")[0]), int(kv.split("=")[1])))
for j in range(8):
    _ffi.check(L.fmhip_sgd_step(hm, hd, j % 4, 0.02, 0.0, 1e-4, 1e-4, None))
_ffi.check(L.fmhip_profile_begin(hm))
for j in range(80):
    _ffi.check(L.fmhip_sgd_step(hm, hd, j % 4, 0.02, 0.0, 1e-4, 1e-4, None))
p = _ffi.Profile()
_ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
us = [p.ms[i] / max(p.launches[i], 1) * 1e3 for i in range(4)]
print("%s k=%d %s: forward %.1f us, backward %.1f, fixup (+ update) %.1f" % (cfg_name, k, " ".join(a for a in sys.argv[2:] if "=" in a), us[0], us[2], us[3]), flush=True)
