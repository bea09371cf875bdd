#!/usr/bin/env python3
"""Soak of the ALS learner (fmhip_als_epoch, fp64; the reference's own fit: S/fm/lib/ALS.scala:15-75): randomly shaped
datasets on both sides of the LDS sweep's 10,000-row limit, long-column marks that send some / all / no columns through the
chip-wide step (FMHIP_ALS_LONG is read per epoch launch), two epochs each against the fp64 oracle at 1e-8.
    python3 tools/soak_als.py [cases, default 40] [first seed, default 1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sparkfm_amd as fmhip  # noqa: E402
from helpers import random_problem  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    n_rows = int(rng.choice([1, 7, 300, 2500, 9999, 10000, 10001, 14000, 30000]))
    n1 = int(rng.choice([2, 9, 40, 300, 1500]))
    k = int(rng.choice([1, 2, 3, 8]))
    hi = int(rng.integers(1, min(12, n1 - 1) + 1))
    lo = int(rng.integers(0, hi + 1))
    mark = str(rng.choice(["", "1", "50", "1000"]))
    regs = (float(rng.choice([0.0, 0.01])), float(rng.choice([0.0, 0.1])), float(rng.choice([0.5, 5.0, 10.0])))
    if mark:
        os.environ["FMHIP_ALS_LONG"] = mark
    else:
        os.environ.pop("FMHIP_ALS_LONG", None)
    a = random_problem(50000 + seed, n_rows, n1, k, lo, hi, empty_rows=(int(rng.integers(0, n_rows)),) if n_rows > 3 else ())
    ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"]).cache()
    fm = fmhip.FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
    fm.reg0, fm.regw, fm.regv = regs
    w0, w, v = a["w0"], a["w"], a["v"]
    tag = "case %d rows=%d n1=%d k=%d nnz/row %d..%d long-column mark %r regs %s" % (seed, n_rows, n1, k, lo, hi, mark, regs)
    for _ in range(2):
        fmhip.HipALS.run().learn(fm, ds)
        w0, w, v = oracle.als_epoch(w0, w, v, regs[0], regs[1], regs[2], a["row_ptr"], a["col"], a["val"], a["y"])
    ok = (np.allclose(fm.v, v, rtol=1e-8, atol=1e-11) and np.allclose(fm.w, w, rtol=1e-8, atol=1e-11) and abs(fm.w0 - w0) <= 1e-9 * abs(w0) + 1e-12
          and np.array_equal(fm.v[:, n1 - 1], a["v"][:, n1 - 1]))
    assert ok, (tag, float(np.abs(fm.v - v).max()), float(np.abs(fm.w - w).max()), fm.w0, w0)
    print("ok", tag, flush=True)
    ds.unpersist()
    fm.close()
