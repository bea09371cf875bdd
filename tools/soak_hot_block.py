#!/usr/bin/env python3
"""Soak of the dense hot block: datasets with 1-128 frequent features (some of them at the gradient-side pages' density only,
one occurring twice in a row / stored with explicit zeros), 1-8 pages, models from barely wider than the block to far wider than
a batch touches (rows-only update, lazy decay), buffer-view and flat-address kernels — predictions, every batch's gradient and
two SGD epochs against the fp64 oracle.
    python3 tools/soak_hot_block.py [cases, default 30] [first seed, default 1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sparkfm_amd as fmhip  # noqa: E402
from sparkfm_amd import _ffi  # noqa: E402
from test_gpu_parity import TOL_Y, check_grad, hot_problem, make, term_scale  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = _ffi.load()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([4, 16, 32, 64, 100]))
    n_hot = int(rng.choice([1, 5, 16, 17, 40, 64, 90, 128]))
    n1 = int(n_hot + rng.choice([5, 10, 300, 5000]))
    n_rows = int(rng.choice([200, 1500, 4000]))
    n_low = int(rng.integers(0, n_hot)) if rng.random() < 0.4 else 0
    dup = int(rng.integers(0, n_hot)) if rng.random() < 0.3 else None
    zero = int(rng.integers(0, n_hot)) if rng.random() < 0.3 else None
    pages, flat = int(rng.integers(1, 9)), int(rng.integers(0, 2))
    br = int(rng.choice([64, 300, 700]))
    regs = (0.0, float(rng.choice([0.0, 1e-3])), float(rng.choice([0.0, 1e-3])))
    tag = "case %d k=%d hot=%d (low %d, dup %s, zero %s) n1=%d rows=%d batch %d pages %d flat %d regs %s" % (seed, k, n_hot, n_low, dup, zero, n1, n_rows, br, pages, flat, regs)
    regs = (regs[0], regs[1], float(rng.choice([regs[2], 1e-2])))
    a, hot_ids = hot_problem(9000 + seed, n_rows, n1, k, n_hot, dup, zero, n_low)
    a["val"] = a["val"].astype(np.float32).astype(np.float64)
    pad = int(rng.choice([0, 0, 600, 6000]))                     # features the data never touch: a model far wider than the batch
    if pad:                                                      # (with few or no sparse columns left beside the block: the rows-only update's edge)
        a["n1"] = n1 + pad
        a["w"] = np.concatenate([a["w"], rng.normal(0, 0.1, pad)])
        a["v"] = np.concatenate([a["v"], rng.normal(0, 0.1, (k, pad))], axis=1)
    try:
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, flat), L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 1), L.fmhip_tune(_ffi.TUNE_HOT_PAGES, pages)
        ds, fm = make(fmhip, a, batch_rows=br)
    finally:
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, 0), L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 1), L.fmhip_tune(_ffi.TUNE_HOT_PAGES, 4)
    try:
        yh = fm.predict(ds)
        oy = oracle.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"])
        assert (np.abs(yh - oy) <= TOL_Y * term_scale(a)).all(), "predict"
        for j in range(0, ds.n_batches, max(1, ds.n_batches // 4)):
            lo, hi = j * br, min(n_rows, (j + 1) * br)
            gv, gw, g0, st = fm.batchGradient(ds, j)
            ogv, ogw, og0, osse, _ = oracle.batch_grad(a["w0"], a["w"], a["v"], lo, hi, a["row_ptr"], a["col"], a["val"], a["y"], threads=4)
            check_grad(gv, gw, ogv, ogw, np.abs(a["v"]).max())
            assert abs(st["sse"] - osse) <= 1e-5 * osse + 1e-9, "sse"
        eta = 0.02 if br >= 300 else 0.005
        sgd = fmhip.HipSGD(eta=eta, reg0=regs[0], regw=regs[1], regv=regs[2])
        w0, w, v = a["w0"], a["w"], a["v"]
        for _ in range(2):
            fm = sgd.learn(fm, ds)
            w0, w, v, sse = oracle.sgd_epoch(w0, w, v, br, a["row_ptr"], a["col"], a["val"], a["y"], eta, *regs)
        if np.isfinite(v).all() and np.abs(v).max() < 1e3:
            ev = float(np.linalg.norm(fm.v - v) / np.linalg.norm(v))
            ew = float(np.linalg.norm(fm.w - w) / max(np.linalg.norm(w), 1e-9))
            assert ev <= 1e-4 and ew <= 1e-4, ("epochs", ev, ew)
        print("ok", tag, "dense ids:", len([i for i in ds.layout()["hot_ids_all"] if i >= 0]), flush=True)
    except AssertionError as e:
        raise AssertionError("%s: %s" % (tag, e))
    ds.unpersist()
    fm.close()
