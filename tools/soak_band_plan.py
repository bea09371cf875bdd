#!/usr/bin/env python3
"""Soak of the backward at sizes where the band-affine placement is planned (>= 1,024 ranges per batch): Zipf rows of random
skew, 20k-120k rows per batch, 200-20,000 features, k 8-64, hot block on / off, ragged last batch — the gradient of two batches
against the fp64 oracle and against the default placement, run-to-run identity, one SGD epoch.
    python3 tools/soak_band_plan.py [cases, default 10] [first seed, default 1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sparkfm_amd as fmhip  # noqa: E402
from sparkfm_amd import _ffi, synth  # noqa: E402
from test_gpu_parity import check_grad, make  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = _ffi.load()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([8, 16, 32, 64]))
    n1 = int(rng.choice([200, 3000, 20000]))
    br = int(rng.choice([20000, 50000, 120000]))
    n_rows = int(br * rng.choice([2, 2.6, 3]))
    lo = int(rng.integers(4, 20))
    hi = int(lo + rng.integers(1, 30))
    zs = float(rng.choice([0.8, 1.05, 1.3]))
    hot = int(rng.integers(0, 2))
    tag = "case %d k=%d n1=%d rows=%d batch %d nnz/row %d..%d zipf %.2f hot %d" % (seed, k, n1, n_rows, br, lo, hi, zs, hot)
    d = synth.make_zipf(7000 + seed, n_rows, n1, lo, hi, zipf_s=zs)
    a = dict(n1=n1, k=k, row_ptr=d["row_ptr"], col=d["col"], val=d["val"].astype(np.float64), y=d["y"].astype(np.float64),
             w0=0.1, w=rng.normal(0, 0.05, n1), v=rng.normal(0, 0.05, (k, n1)))
    try:
        L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, hot)
        ds, fm = make(fmhip, a, batch_rows=br)
    finally:
        L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 1)
    lay = ds.layout()
    try:
        for b in (0, ds.n_batches - 1):
            r0, r1 = b * br, min(n_rows, (b + 1) * br)
            g = {}
            for mode in (0, 2, 2):
                _ffi.check(L.fmhip_model_tune(fm.handle, _ffi.TUNE_XCD_PLACEMENT, mode))
                g.setdefault(mode, []).append(fm.batchGradient(ds, b))
            (gv0, gw0, g00, st0), = g[0]
            (gv2, gw2, g02, st2), (gv2b, gw2b, _, _) = g[2]
            assert np.array_equal(gv2, gv2b) and np.array_equal(gw2, gw2b), "not deterministic"
            ogv, ogw, og0, osse, _ = oracle.batch_grad(a["w0"], a["w"], a["v"], r0, r1, a["row_ptr"], a["col"], a["val"], a["y"], threads=8)
            check_grad(gv2, gw2, ogv, ogw, np.abs(a["v"]).max())
            check_grad(gv0, gw0, ogv, ogw, np.abs(a["v"]).max())
            assert abs(st2["sse"] - osse) <= 1e-5 * osse, "sse"
        eta, regs = 0.02, (0.0, 1e-3, 1e-3)
        sgd = fmhip.HipSGD(eta=eta, reg0=regs[0], regw=regs[1], regv=regs[2])
        fm2 = sgd.learn(fm, ds)
        w0, w, v, sse = oracle.sgd_epoch(a["w0"], a["w"], a["v"], br, a["row_ptr"], a["col"], a["val"], a["y"], eta, *regs, threads=8)
        if np.isfinite(v).all() and np.abs(v).max() < 1e3:
            ev = float(np.linalg.norm(fm2.v - v) / np.linalg.norm(v))
            assert ev <= 1e-4, ("epoch", ev)
        print("ok", tag, "ranges %d planned %d band-affine %d" % (lay["ranges"], lay["planned_ranges"], lay["band_affine_ranges"]), flush=True)
    except AssertionError as e:
        raise AssertionError("%s: %s" % (tag, e))
    ds.unpersist()
    fm.close()
