#!/usr/bin/env python3
"""Replays one case of the random-shapes property test (tests/test_gpu_parity.py: _random_shapes) and prints where the GPU's
SGD epoch leaves the oracle's.   python3 tools/replay_case.py SEED CASE [flat] [k values] [regs:r0,rw,rv] [tune:KEY=VALUE] [gtune:KEY=VALUE]
QUICK=1 in the environment stops after the epoch's comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sparkfm_amd as fmhip  # noqa: E402
from helpers import random_problem  # noqa: E402
from sparkfm_amd import _ffi  # noqa: E402
from test_gpu_parity import make  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
ks = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 5, 8, 13, 16, 32, 40, 64]
L = _ffi.load()
L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
rng = np.random.default_rng(seed)
for case in range(want + 1):
    k = int(rng.choice(ks))
    n_rows = int(rng.integers(1, 1200))
    n1 = int(rng.integers(2, 400)) if case < 24 else int(rng.integers(400, 6000))
    hi = int(rng.integers(1, min(n1, 70) + 1))
    lo = int(rng.integers(0, hi + 1))
    batch_rows = int(rng.choice([0, 1, 7, 64, 300, 5000]))
    empty = tuple(rng.integers(0, n_rows, 2).tolist())
    a = random_problem(1000 + case + (seed - 20261003) * 1000, n_rows, n1, k, lo, hi, empty_rows=empty) if case == want else None
    if case % 3 == 0:
        hot = rng.integers(0, n1, 2)
        if case == want and len(a["col"]):                           # a few dominating features, as the test makes them
            for r in range(n_rows):
                s_ = slice(a["row_ptr"][r], a["row_ptr"][r + 1])
                if s_.stop - s_.start >= 2 and not np.isin(hot, a["col"][s_]).any():
                    a["col"][s_.start] = hot[0]
    # (the remaining draws of a case — the batch index — follow dataset creation; they do not matter for the last case)
    if case < want:
        # replay the draw of b: needs nb
        br = batch_rows if batch_rows > 0 else n_rows
        nb = (n_rows + br - 1) // br
        rng.integers(0, nb)
print("case", want, dict(k=k, n_rows=n_rows, n1=n1, lo=lo, hi=hi, batch_rows=batch_rows, empty=empty), "row lengths", np.diff(a["row_ptr"]))
regs = (0.01, 0.01, 0.01) if want % 4 else (0.0, 0.0, 0.0)
for extra in sys.argv[5:]:                                           # regs:r0,rw,rv   tune:KEY=VALUE (per model, after creation)
    if extra.startswith("regs:"):
        regs = tuple(float(x) for x in extra[5:].split(","))
for extra in sys.argv[5:]:                                           # gtune:KEY=VALUE: process-wide, before the dataset is built
    if extra.startswith("gtune:"):
        _ffi.check(L.fmhip_tune(int(extra[6:].split("=")[0]), int(extra.split("=")[1])))
ds, fm = make(fmhip, a, batch_rows=batch_rows)
for extra in sys.argv[5:]:
    if extra.startswith("tune:"):
        _ffi.check(L.fmhip_model_tune(fm.handle, int(extra[5:].split("=")[0]), int(extra.split("=")[1])))
for b in range(ds.n_batches):
    bi = ds.batch_info(b)
    r0, r1 = bi["row0"], bi["row0"] + bi["rows"]
    gv, gw, g0, st = fm.batchGradient(ds, b)
    ogv, ogw, og0, osse, oe = oracle.batch_grad(a["w0"], a["w"], a["v"], r0, r1, a["row_ptr"], a["col"], a["val"], a["y"])
    p0, p1 = int(a["row_ptr"][r0]), int(a["row_ptr"][r1])
    x = a["val"][p0:p1]
    er = np.repeat(np.asarray(oe), np.diff(a["row_ptr"][r0:r1 + 1]))
    cancel = np.zeros(n1)
    np.add.at(cancel, a["col"][p0:p1], np.abs(er * x * x))                          # sum |e x^2| per feature: what v multiplies
    d = np.abs(gv - ogv)
    f_, i = np.unravel_index(np.argmax(d), d.shape)
    print("batch %d: max |gv - ogv| %.3e at (f=%d, i=%d): gpu %.9g oracle %.9g; max |ogv| %.3e; |v| %.3e x sum|e x^2| %.3e = %.3e; layout %s"
          % (b, d.max(), f_, i, gv[f_, i], ogv[f_, i], np.abs(ogv).max(), abs(a["v"][f_, i]), cancel[i], abs(a["v"][f_, i]) * cancel[i],
             {k_: v_ for k_, v_ in ds.layout().items() if k_ in ("hot_ids", "hot_pages", "nnz_sparse_backward")}))
    rb = np.repeat(np.arange(r1 - r0), np.diff(a["row_ptr"][r0:r1 + 1]))
    cb = a["col"][p0:p1]
    ex = np.abs(er * x)
    qrow = np.zeros((r1 - r0, k)); np.add.at(qrow, rb, (a["v"][:, cb] * x).T)
    terms = np.zeros(n1); np.add.at(terms, cb, ex * (np.abs(qrow).max(axis=1)[rb] + np.abs(x) * np.abs(a["v"][:, cb]).max(axis=0)))
    absw = np.zeros(n1); np.add.at(absw, cb, ex)
    vmax = np.abs(a["v"]).max()
    scale = max(np.abs(ogv).max(), np.abs(ogw).max() * vmax, absw.max() * vmax, 1e-6)
    rowmax = np.maximum(np.abs(ogv).max(axis=0), 1e-3 * scale)
    tol = 1e-4 * rowmax + 2e-6 * terms
    ratio = d / tol[None, :]
    f2, i2 = np.unravel_index(np.argmax(ratio), ratio.shape)
    cnt = int((cb == i2).sum())
    print("   worst ratio %.3f at (f=%d, i=%d): gpu %.9g oracle %.9g |d| %.3e tol %.3e (rowmax %.3e terms %.3e) entries of this feature in the batch: %d"
          % (ratio.max(), f2, i2, gv[f2, i2], ogv[f2, i2], d[f2, i2], tol[i2], rowmax[i2], terms[i2], cnt))
    if ds.n_batches > 4:
        break
br = ds.info()["batch_rows"]
eta = 0.02 if br >= 64 else 0.001
sgd = fmhip.HipSGD(eta=eta, reg0=regs[0], regw=regs[1], regv=regs[2])
sgd.learn(fm, ds)
w0, w, v, sse = oracle.sgd_epoch(a["w0"], a["w"], a["v"], br, a["row_ptr"], a["col"], a["val"], a["y"], eta, *regs)
dv = np.abs(fm.v - v)
print("regs", regs, "eta", eta, "|dv| max", dv.max(), "norm", np.linalg.norm(fm.v - v), "of", np.linalg.norm(v), "w0", fm.w0, w0)
bad = np.argwhere(dv > 1e-6)
print("entries off by > 1e-6:", len(bad), bad[:10].tolist(), "features in the data:", sorted(set(a["col"].tolist()))[:20])
for f_, i in bad[:6]:
    print("  v[%d,%d]: gpu %.9g oracle %.9g initial %.9g" % (f_, i, fm.v[f_, i], v[f_, i], a["v"][f_, i]))
per_feature = np.linalg.norm(fm.v - v, axis=0)
lay = ds.layout()
print("deviation by feature (largest 12):", [(int(i), "%.2e" % per_feature[i], "dense" if i in lay.get("hot_ids_all", []) else "sparse", int((a["col"] == i).sum()))
                                              for i in np.argsort(-per_feature)[:12]])
if os.environ.get("QUICK"):
    sys.exit(0)

# is a deviation from the oracle an error or the amplification of fp32 rounding by the training dynamics?  The same epoch with
# the dense hot block off (another summation order, same arithmetic otherwise): if two GPU runs differ from each other as much
# as each differs from the oracle, it is the dynamics.
gpu_v, gpu_w0 = fm.v.copy(), fm.w0
ds.unpersist(); fm.close()
L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 0)
ds, fm = make(fmhip, a, batch_rows=batch_rows)
L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 1)
fmhip.HipSGD(eta=eta, reg0=regs[0], regw=regs[1], regv=regs[2]).learn(fm, ds)
print("hot block off: |v - oracle| %.3e, |v - v(hot on)| %.3e (hot on vs oracle: %.3e); w0 %.9g vs %.9g (hot on) vs %.9g (oracle)"
      % (np.linalg.norm(fm.v - v), np.linalg.norm(fm.v - gpu_v), np.linalg.norm(gpu_v - v), fm.w0, gpu_w0, w0))
# and the oracle itself in another summation order: rows of each batch reversed (fp64: the difference shows the conditioning)

# where does a hot-block deviation enter: predictions and the first batch's gradient at the INITIAL parameters, block on / off
for hot_on in (1, 0):
    ds.unpersist(); fm.close()
    L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, hot_on)
    ds, fm = make(fmhip, a, batch_rows=batch_rows)
    L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 1)
    yh = fm.predict(ds)
    oy = oracle.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"])
    bi = ds.batch_info(0)
    gv, gw, g0, st = fm.batchGradient(ds, 0)
    ogv, ogw, og0, osse, oe = oracle.batch_grad(a["w0"], a["w"], a["v"], 0, bi["rows"], a["row_ptr"], a["col"], a["val"], a["y"])
    print("hot %d: yhat rel-L2 err %.2e (max abs %.2e of |y|max %.2e); G_V rel-L2 %.2e, G_w rel-L2 %.2e, G_w0 %.9g vs %.9g, sse %.9g vs %.9g; dense ids %d"
          % (hot_on, np.linalg.norm(yh - oy) / np.linalg.norm(oy), np.abs(yh - oy).max(), np.abs(oy).max(), np.linalg.norm(gv - ogv) / np.linalg.norm(ogv),
             np.linalg.norm(gw - ogw) / np.linalg.norm(ogw), g0, og0, st["sse"], osse, len([i for i in ds.layout()["hot_ids_all"] if i >= 0])))

# ... and batch by batch (a ragged last batch, batches the hot features are absent from)
ds.unpersist(); fm.close()
ds, fm = make(fmhip, a, batch_rows=batch_rows)
errs = []
for b in range(ds.n_batches):
    bi = ds.batch_info(b)
    r0, r1 = bi["row0"], bi["row0"] + bi["rows"]
    gv, gw, g0, st = fm.batchGradient(ds, b)
    ogv, ogw, og0, osse, oe = oracle.batch_grad(a["w0"], a["w"], a["v"], r0, r1, a["row_ptr"], a["col"], a["val"], a["y"])
    errs.append((b, bi["rows"], float(np.linalg.norm(gv - ogv) / max(np.linalg.norm(ogv), 1e-30)), float(np.linalg.norm(gw - ogw) / max(np.linalg.norm(ogw), 1e-30)), abs(g0 - og0)))
print("per batch (rows, G_V rel err, G_w rel err, |dG_w0|):", [(b, r, "%.1e" % e1, "%.1e" % e2, "%.1e" % e3) for b, r, e1, e2, e3 in errs])
# one step at a time from the same start: after which step does the model leave the oracle?
ds.unpersist(); fm.close()
ds, fm = make(fmhip, a, batch_rows=batch_rows)
w0o, wo, vo = a["w0"], a["w"], a["v"]
for b in range(ds.n_batches):
    bi = ds.batch_info(b)
    _ffi.check(L.fmhip_sgd_step(fm.handle, ds.handle, b, eta, regs[0], regs[1], regs[2], None))
    fm._device_updated()
    w0o, wo, vo, _ = oracle.sgd_step(w0o, wo, vo, bi["row0"], bi["row0"] + bi["rows"], a["row_ptr"], a["col"], a["val"], a["y"], eta, *regs)
    print("after step %d (%d rows): |v - oracle| %.2e  |w - oracle| %.2e  w0 %.9g vs %.9g" % (b, bi["rows"], np.linalg.norm(fm.v - vo), np.linalg.norm(fm.w - wo), fm.w0, w0o))
