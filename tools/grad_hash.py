#!/usr/bin/env python3
"""Prints hashes of a batch gradient and of the parameters after a few SGD steps on fixed synthetic data: run it with two builds
of the library (FMHIP_LIB=sparkfm_amd/lib/libfmhip_<name>.so, tools/build_variant.sh) to check that a kernel change which
only regroups loads or moves registers left every bit where it was.
    python3 tools/grad_hash.py [k, default 32] [rows, default 300000] [features, default 100000]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkfm_amd as fmhip  # noqa: E402
from sparkfm_amd import synth  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
n1 = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
d = synth.make_zipf(7, rows, n1, 20, 60, zipf_s=1.05)
w0, w, v = synth.init_params(11, n1, k, stdev=0.05)
for hot in (None, False):
    ds = fmhip.DataSet.from_arrays(d, batch_rows=rows // 3, hot_block=hot).cache()
    fm = fmhip.FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = w0, w, v
    gv, gw, g0, st = fm.batchGradient(ds, 1)
    h = hashlib.sha256(np.ascontiguousarray(gv).tobytes() + np.ascontiguousarray(gw).tobytes()).hexdigest()[:16]
    sgd = fmhip.HipSGD(eta=0.05, regw=1e-4, regv=1e-4)
    for _ in range(2):
        sgd.learn(fm, ds)
    p = hashlib.sha256(np.ascontiguousarray(fm.v).tobytes() + np.ascontiguousarray(fm.w).tobytes() + np.float64(fm.w0).tobytes()).hexdigest()[:16]
    print("k=%d rows=%d n1=%d hot=%s: gradient %s g0 %.17g sse %.17g | after 2 epochs %s" % (k, rows, n1, hot, h, g0, st["sse"], p))
    ds.unpersist()
    fm.close()
