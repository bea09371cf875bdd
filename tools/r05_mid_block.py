#!/usr/bin/env python3
"""Experiment (r05_experiments.md section 6): the backward with the mid-frequency columns cut by row band (FMHIP_MID_BLOCK="lo:hi",
read at dataset creation).  Run ON THE GPU BOX, once per setting, C3's shape (1M rows x 100k features, k = 32, 250k-row batches):
    python3 tools/r05_mid_block.py <gradient.npy> [merged]      # env FMHIP_MID_BLOCK unset: writes the reference gradient of batch 1
    FMHIP_MID_BLOCK=1000:10000 python3 tools/r05_mid_block.py <gradient.npy> [merged]   # compares with it, then times the step's kernels
Needs the experiment patch (git apply profiles/r05_mid_block.patch; not part of the library).
`merged`: leave the merged finish on (wrong with pieces: timing only)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import DataSet, FMModel, _ffi, synth  # noqa: E402

path = sys.argv[1]
merged = len(sys.argv) > 2 and sys.argv[2] == "merged"
mid = os.environ.get("FMHIP_MID_BLOCK")
cfg = synth.CONFIGS["C3"]
d = synth.make_config("C3", rows=1_000_000)
L = _ffi.load()
if not merged:
    _ffi.check(L.fmhip_tune(_ffi.TUNE_MERGED_FINISH, 0))
ds = DataSet.from_arrays(d, batch_rows=250_000).cache()
lay = ds.layout()
w0, w, v = synth.init_params(cfg["seed"] + 1000, cfg["features"], cfg["k"])
fm = FMModel(cfg["features"] - 1, cfg["k"])
fm.w0, fm.w, fm.v = w0, np.random.default_rng(1).normal(0, 0.01, cfg["features"]), v
gv, gw, g0, st = fm.batchGradient(ds, 1)
if mid is None:
    np.save(path, np.concatenate([gv.ravel(), gw.ravel(), [g0]]))
    print("reference gradient written: |G_V| %.6g |G_w| %.6g g0 %.6g" % (np.linalg.norm(gv), np.linalg.norm(gw), g0))
else:
    ref = np.load(path)
    got = np.concatenate([gv.ravel(), gw.ravel(), [g0]])
    scale = np.abs(ref).max()
    print("FMHIP_MID_BLOCK=%s: max |dG| / max |G| = %.3g, rel-L2 %.3g  (band plan: %d of %d ranges band-affine)" %
          (mid, np.abs(got - ref).max() / scale, np.linalg.norm(got - ref) / np.linalg.norm(ref), lay["band_affine_ranges"], lay["ranges"]))
hm, hd = fm.handle, ds.handle
for j in range(400):
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
print("mid %s merged %s: kernels per step (us) %s  sum %.1f  | band-affine %d of %d ranges | last batch mse %.6f nonfinite %d" %
      (mid, merged, {n: round(x, 1) for n, x in us.items()}, sum(us.values()), lay["band_affine_ranges"], lay["ranges"], st.sse / max(st.rows, 1), st.nonfinite), flush=True)
