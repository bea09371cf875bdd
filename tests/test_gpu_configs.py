"""Parity at the BASELINE.json configurations the single-GPU suite did not reach in round 1:

  C4  10M rows x 1M features, k=32 (the 8-GPU config): ONE rank's shard — 1.25M rows — against the oracle
  C5  Criteo-shaped (39 fields, hashed, k=64): a shard narrow enough for host RAM (2^22 slots x 200k rows),
      with and without weight decay (lazy rows-only update vs the oracle's eager one), and once more with the
      flat-address kernels that the full 2^25-slot width (V = 8.6 GB > 4 GiB) takes
  plus the scoring-only dataset / fit -> held-out RMSE flow of the reference's demo (S/driver.scala:100-112)
  and the library-side RCCL exchange with one rank (two ranks: gated on a second GPU).

Tolerances as in test_gpu_parity.py (fp32 device vs the fp64 oracle on identical inputs).
"""
import math
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from sparkfm_amd import _ffi  # noqa: F401  (the tuning keys' names)
from test_gpu_parity import TOL_Y, check_grad, term_scale

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def fmhip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sparkfm_amd
    return sparkfm_amd


def sample_rows_check(fm, ds, d, w0, w, v, n_sample, seed):
    """oracle.predict on a random sample of rows (full model) vs the GPU's predictions of those rows."""
    rng = np.random.default_rng(seed)
    row_ptr, col = d["row_ptr"], d["col"]
    rows = np.sort(rng.choice(len(row_ptr) - 1, n_sample, replace=False))
    lens = (row_ptr[rows + 1] - row_ptr[rows]).astype(np.int64)
    sub_ptr = np.concatenate([[0], np.cumsum(lens)])
    idx = np.repeat(row_ptr[rows], lens) + (np.arange(int(sub_ptr[-1])) - np.repeat(sub_ptr[:-1], lens))
    val = d["val"][idx].astype(np.float64)
    oy = oracle.predict(w0, w, v, sub_ptr, col[idx], val)
    yh = fm.predict(ds)
    scale = term_scale(dict(y=oy, row_ptr=sub_ptr, col=col[idx], val=val, w0=w0, w=w, v=v))
    assert (np.abs(yh[rows] - oy) <= TOL_Y * scale).all(), float((np.abs(yh[rows] - oy) / scale).max())
    return yh


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_c4_one_rank_shard(fmhip):
    """BASELINE config 4's per-GPU shard: 1.25M rows x 1M features, k=32, mini-batches of 625k rows (the
    bench's data-parallel default).  Sampled predictions, the FULL gradient of batch 0 element by element,
    and two SGD steps with weight decay (dense update: a batch touches most of the model's hot rows, the
    cold majority only decays) against the oracle."""
    from sparkfm_amd import synth
    cfg = synth.CONFIGS["C4"]
    n_rows, br = 1_250_000, 625_000
    d = synth.make_config("C4", rows=n_rows)
    n1, k = cfg["features"], cfg["k"]
    rng = np.random.default_rng(4)
    w0, w, v = 0.05, rng.normal(0, 0.05, n1), rng.normal(0, 0.05, (k, n1))
    val, y = d["val"].astype(np.float64), d["y"].astype(np.float64)
    ds = fmhip.DataSet.from_arrays(d, batch_rows=br).cache()
    assert ds.info()["n_batches"] == 2 and ds.info()["dimension"] <= n1 - 1
    fm = fmhip.FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = w0, w, v
    sample_rows_check(fm, ds, d, w0, w, v, 3000, 5)
    threads = min(oracle.max_threads(), 16)
    gv, gw, g0, st = fm.batchGradient(ds, 0)
    ogv, ogw, og0, osse, _ = oracle.batch_grad(w0, w, v, 0, br, d["row_ptr"], d["col"], val, y, threads=threads)
    check_grad(gv, gw, ogv, ogw, np.abs(v).max())
    assert g0 == pytest.approx(og0, rel=1e-4, abs=1e-2)
    assert st["sse"] == pytest.approx(osse, rel=1e-5) and st["rows"] == br and st["nnz"] == d["row_ptr"][br]
    del gv, ogv
    sgd = fmhip.HipSGD(eta=0.02, regw=1e-4, regv=1e-4)
    ow0, ow, ov = w0, w, v
    for b in range(2):
        s = sgd.step(fm, ds, b)
        ow0, ow, ov, sse = oracle.sgd_step(ow0, ow, ov, b * br, min(n_rows, (b + 1) * br), d["row_ptr"], d["col"], val, y,
                                           0.02, 0.0, 1e-4, 1e-4, threads=threads)
        assert s["sse"] == pytest.approx(sse, rel=1e-5)
    assert rel(fm.v, ov) <= 1e-5 and rel(fm.w, ow) <= 1e-5 and fm.w0 == pytest.approx(ow0, rel=1e-5, abs=1e-7)
    ds.unpersist()
    fm.close()


C5_ROWS, C5_SLOTS, C5_BATCH = 200_000, 1 << 22, 100_000


@pytest.fixture(scope="module")
def c5():
    from sparkfm_amd import synth
    d = synth.make_config("C5", rows=C5_ROWS, features=C5_SLOTS)
    k = synth.CONFIGS["C5"]["k"]
    rng = np.random.default_rng(55)
    w0 = 0.05
    w = rng.normal(0, 0.05, C5_SLOTS)
    v = (rng.standard_normal((C5_SLOTS, k), dtype=np.float32) * np.float32(0.02)).T.astype(np.float64)
    val, y = d["val"].astype(np.float64), d["y"].astype(np.float64)
    threads = min(oracle.max_threads(), 8)
    ogv, ogw, og0, osse, _ = oracle.batch_grad(w0, w, v, 0, C5_BATCH, d["row_ptr"], d["col"], val, y, threads=threads)
    return dict(d=d, k=k, w0=w0, w=w, v=v, val=val, y=y, threads=threads, grad=(ogv, ogw, og0, osse))


@pytest.mark.parametrize("flat,hot", [(0, 1), (1, 1), (1, 0)])
def test_c5_criteo_shape_gradient(fmhip, c5, flat, hot):
    """C5's row shape (39 hashed fields, k=64, duplicate slots inside a row possible): sampled predictions
    and the full gradient of one batch.  flat=1 forces the flat-address forward and the non-pipelined
    backward — the kernels the real 2^25-slot width selects because V and P pass 4 GiB."""
    from sparkfm_amd import _ffi
    L = _ffi.load()
    d, k, w0, w, v = c5["d"], c5["k"], c5["w0"], c5["w"], c5["v"]
    rp = d["row_ptr"]
    lens = np.diff(rp)
    assert 25 <= lens.min() and lens.max() <= 39 and d["col"].max() < C5_SLOTS and d["val"].max() < 8.0
    try:
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, flat)
        L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, hot)
        ds = fmhip.DataSet.from_arrays(d, batch_rows=C5_BATCH).cache()
        fm = fmhip.FMModel(C5_SLOTS - 1, k)
        fm.w0, fm.w, fm.v = w0, w, v
        sample_rows_check(fm, ds, d, w0, w, v, 2000, 6)
        gv, gw, g0, st = fm.batchGradient(ds, 0)
    finally:
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, 0)
        L.fmhip_tune(_ffi.TUNE_HOT_BLOCK, 1)
    ogv, ogw, og0, osse = c5["grad"]
    check_grad(gv, gw, ogv, ogw, np.abs(v).max())
    assert g0 == pytest.approx(og0, rel=1e-4, abs=1e-2)
    assert st["sse"] == pytest.approx(osse, rel=1e-5) and st["rows"] == C5_BATCH and st["nnz"] == rp[C5_BATCH]
    feat, ptr, trows, tvals = ds.transposeInput(0)                    # index gathers: bit-exact, hot block or not
    cols0 = d["col"][:rp[C5_BATCH]]
    np.testing.assert_array_equal(feat, np.unique(cols0).astype(np.int32))
    order = np.argsort(cols0, kind="stable")
    np.testing.assert_array_equal(trows, np.repeat(np.arange(C5_BATCH), lens[:C5_BATCH])[order].astype(np.int32))
    np.testing.assert_array_equal(tvals, d["val"][:rp[C5_BATCH]][order])
    ds.unpersist()
    fm.close()


@pytest.mark.parametrize("regs,flat,lazy", [((0.0, 1e-3, 1e-3), 0, 1), ((0.0, 0.0, 0.0), 0, 1), ((0.0, 1e-3, 1e-3), 1, 1),
                                            ((0.0, 1e-3, 1e-3), 0, 0)])
def test_c5_criteo_shape_sgd(fmhip, c5, regs, flat, lazy):
    """Two epochs (4 steps) on the C5 shard vs the oracle.  A batch touches ~3 % of the 2^22 rows, so the
    update is rows-only; with weight decay that is the LAZY form (the decay of every row rides in a scale
    of the tables, fm_apply.hip) checked here against the oracle's EAGER update; lazy=0 forces the dense
    update for comparison; flat=1 takes the > 4 GiB kernels."""
    from sparkfm_amd import _ffi
    L = _ffi.load()
    d, k, val, y = c5["d"], c5["k"], c5["val"], c5["y"]
    try:
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, flat)
        L.fmhip_tune(_ffi.TUNE_LAZY_DECAY, lazy)
        ds = fmhip.DataSet.from_arrays(d, batch_rows=C5_BATCH).cache()
        fm = fmhip.FMModel(C5_SLOTS - 1, k)
        fm.w0, fm.w, fm.v = c5["w0"], c5["w"], c5["v"]
        sgd = fmhip.HipSGD(eta=0.05, reg0=regs[0], regw=regs[1], regv=regs[2])
        stats = []
        for _ in range(2):
            sgd.learn(fm, ds)
            stats.append(sgd.last_stats["sse"])
        mid_rmse = fm.computeRMSE(ds)          # scoring a lazily decayed model: the scale must be applied
        gv, v_after = None, fm.v
    finally:
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, 0)
        L.fmhip_tune(_ffi.TUNE_LAZY_DECAY, 1)
    ow0, ow, ov = c5["w0"], c5["w"], c5["v"]
    for e in range(2):
        ow0, ow, ov, sse = oracle.sgd_epoch(ow0, ow, ov, C5_BATCH, d["row_ptr"], d["col"], val, y, 0.05, *regs,
                                            threads=c5["threads"])
        assert stats[e] == pytest.approx(sse, rel=2e-5)
    assert rel(v_after, ov) <= 1e-5 and rel(fm.w, ow) <= 1e-5 and fm.w0 == pytest.approx(ow0, rel=1e-5, abs=1e-7)
    assert mid_rmse == pytest.approx(oracle.rmse(ow0, ow, ov, d["row_ptr"], d["col"], val, y, threads=c5["threads"]), rel=1e-5)
    ds.unpersist()
    fm.close()


@pytest.mark.parametrize("k", [16, 32])
def test_lazy_decay_long_run_and_refold(fmhip, k):
    """Lazy weight decay over many steps: 300 steps with strong decay on a wide model (the scale drops far
    enough to be folded back into the tables at least once) track the oracle's eager update; switching to a
    dense step (another dataset whose batch touches most rows) folds the scale and stays on track.  k = 16: rows with a spare
    slot that carries the linear weight (its own scale); k = 32: rows without one."""
    from helpers import random_problem
    a = random_problem(77, 600, 5000, k, 2, 12)
    ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"], batch_rows=100).cache()
    fm = fmhip.FMModel(a["n1"] - 1, a["k"])
    fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
    eta, regs = 0.5, (0.0, 0.2, 0.2)                  # (1 - 0.1)^300 ~ 2e-14 << 2^-24: several folds
    sgd = fmhip.HipSGD(eta=eta, reg0=regs[0], regw=regs[1], regv=regs[2])
    w0, w, v = a["w0"], a["w"], a["v"]
    for _ in range(50):
        sgd.learn(fm, ds)
        w0, w, v, _ = oracle.sgd_epoch(w0, w, v, 100, a["row_ptr"], a["col"], a["val"], a["y"], eta, *regs)
    scale = max(np.abs(v).max(), 1e-30)
    assert np.abs(fm.v - v).max() <= 2e-4 * scale and np.abs(fm.w - w).max() <= 2e-4 * max(np.abs(w).max(), 1e-30)
    assert fm.w0 == pytest.approx(w0, rel=1e-4, abs=1e-6)
    assert fm.computeRMSE(ds) == pytest.approx(oracle.rmse(w0, w, v, a["row_ptr"], a["col"], a["val"], a["y"]), rel=1e-4)
    ds.unpersist()
    fm.close()


def test_scoring_only_rows_and_the_demo_flow(fmhip, tmp_path):
    """S/driver.scala:100-112 end to end, from a libFM text file: loadLibFMFile -> splitByRandom ->
    FM(train).learnWith(HipSGD) -> computeRMSE(test).  The held-out split lives on the device as a
    scoring-only dataset (fmhip_rows_create: no transposes); single rows go through fmhip_predict_rows."""
    from sparkfm_amd import DataCollection, FMUtils, _ffi, synth
    d = synth.make_zipf(321, 6000, 400, 3, 20, zipf_s=1.05)
    raw = fmhip.DataSet.from_arrays(dict(d, val=d["val"].astype(np.float64), y=d["y"].astype(np.float64)), name="raw")
    path = str(tmp_path / "raw.libfm")
    FMUtils.saveAsLibFMFile(raw, path, index_offset=0)             # 0: a file that round-trips through the loader (quirk Q9)
    loaded = FMUtils.loadLibFMFile(path)
    np.testing.assert_array_equal(loaded.col, raw.col)
    np.testing.assert_allclose(loaded.val, raw.val, atol=5.1e-4)   # the saver's "#.###" format (S/fm/FMUtils.scala:71-74)
    coll = DataCollection.splitByRandom(loaded, 0.8, 0.2, seed=9, batch_rows=1000)
    train, test = coll.trainingSet, coll.testSet
    assert test.scoring and not train.scoring and train.size + test.size == 6000
    trainer = fmhip.FM(train, 8, maxIteration=5, seed=3)
    w0, w, v = synth.init_params(12, coll.dimension + 1, 8, stdev=0.05)
    k1 = train.dimension + 1
    fm = trainer.learnWith(fmhip.HipSGD.run(eta=0.1, regw=1e-4, regv=1e-4), init=(w0, w[:k1], v[:, :k1]))
    ow0, ow, ov = w0, w[:k1].copy(), v[:, :k1].copy()
    for _ in range(5):
        ow0, ow, ov, _ = oracle.sgd_epoch(ow0, ow, ov, 1000, train.row_ptr, train.col, train.val, train.y, 0.1, 0.0, 1e-4, 1e-4)
    assert rel(fm.v, ov) <= 1e-4
    # held-out RMSE: the test split may hold a feature index the training split never saw -> widen like the
    # reference would have to (new FMModel(dimension) is sized by the TRAINING set, S/fm/impl/FactorizationMachines.scala:39)
    keep = test.col <= train.dimension
    if keep.all():
        rmse_gpu = fm.computeRMSE(test)
        assert rmse_gpu == pytest.approx(oracle.rmse(ow0, ow, ov, test.row_ptr, test.col, test.val, test.y), rel=1e-4)
        assert rmse_gpu < math.sqrt(np.mean(test.y ** 2))
        with pytest.raises(_ffi.FmhipError, match="scoring only"):
            fmhip.HipSGD(eta=0.1).learn(fm, test)
    else:
        with pytest.raises(_ffi.FmhipError):
            fm.computeRMSE(test)
    # FMModel.predict(SparseVector) for ad-hoc rows, incl. an empty one (quirk Q6: w0 exactly)
    r = 17
    s = slice(train.row_ptr[r], train.row_ptr[r + 1])
    want = oracle.predict(fm.w0, fm.w, fm.v, np.array([0, s.stop - s.start]), train.col[s], train.val[s])[0]
    assert fm.predict((train.col[s], train.val[s])) == pytest.approx(want, rel=1e-5, abs=1e-6)
    assert fm.predict(([], [])) == np.float32(fm.w0)
    test.unpersist()


def test_get_rows_serves_the_fp64_masters_while_they_are_fresh(fmhip):
    """`fm.w(i)` / `fm.v(::, i)` of a few features (fmhip_model_get_rows): right after the parameters were set the exact
    doubles come back (as fmhip_model_get_params returns them), not their fp32 device copies; once an fp32 step has run,
    the device values."""
    from sparkfm_amd import synth
    d = synth.make_zipf(5, 600, 90, 3, 12, zipf_s=1.05)
    rng = np.random.default_rng(8)
    w, v = rng.normal(0, 0.1, 90) + 1e-11, rng.normal(0, 0.1, (8, 90)) + 1e-11      # not representable in fp32
    ds = fmhip.DataSet.from_arrays(d, batch_rows=300).cache()
    fm = fmhip.FMModel(89, 8)
    fm.w0, fm.w, fm.v = 0.25, w, v
    ids = np.array([3, 0, 89, 3, 41], np.int32)
    fm.handle
    w_r, v_r = fm.rows(ids)
    np.testing.assert_array_equal(w_r, w[ids])
    np.testing.assert_array_equal(v_r, v[:, ids])
    fmhip.HipSGD(eta=0.01).learn(fm, ds)
    w_r, v_r = fm.rows(ids)
    np.testing.assert_array_equal(w_r, fm.w[ids])
    np.testing.assert_array_equal(v_r, fm.v[:, ids])
    assert (w_r.astype(np.float32) == w_r).all() and not np.array_equal(v_r, v[:, ids])
    ds.unpersist()
    fm.close()


def test_device_side_init(fmhip):
    """fmhip_model_init_normal: `new FMModel` drawn on the GPU — N(mean, stdev) moments, w = w0 = 0,
    reproducible per seed, padding untouched (k=20 -> 32 floats per row)."""
    fm = fmhip.FMModel(49_999, 20, init_mean=0.0, init_stdev=0.01, seed=7, init_on_device=True)
    v = fm.v
    assert v.shape == (20, 50_000) and fm.w0 == 0.0 and not fm.w.any()
    assert abs(v.mean()) < 1e-4 and v.std() == pytest.approx(0.01, rel=0.01)
    assert abs(((v / 0.01) ** 4).mean() - 3.0) < 0.1                   # kurtosis of a normal
    fm2 = fmhip.FMModel(49_999, 20, init_stdev=0.01, seed=7, init_on_device=True)
    np.testing.assert_array_equal(fm2.v, v)
    fm3 = fmhip.FMModel(49_999, 20, init_stdev=0.01, seed=8, init_on_device=True)
    assert not np.array_equal(fm3.v, v)
    for m in (fm, fm2, fm3):
        m.close()


def test_close_that_drops_the_parameters_says_so(fmhip):
    """a model drawn on the device and closed unread — or closed with discard=True after training — has lost its
    parameters; reading them afterwards must raise instead of re-drawing the initial values or serving stale host arrays.
    Assigning new parameters brings the model back."""
    from sparkfm_amd import synth
    d = synth.make_zipf(12, 3000, 400, 4, 20, zipf_s=1.05)
    ds = fmhip.DataSet.from_arrays(d, batch_rows=1000).cache()
    sgd = fmhip.HipSGD(eta=0.05)
    fm = fmhip.FMModel(399, 8, seed=3, init_on_device=True)
    sgd.learn(fm, ds)
    fm.close()                                     # never read: no pull, the trained values are gone
    for read in (lambda: fm.v, lambda: fm.w, lambda: fm.w0, lambda: fm.computeRMSE(ds), lambda: fm.handle):
        with pytest.raises(RuntimeError, match="discarded by close"):
            read()
    fm2 = fmhip.FMModel(399, 8, seed=3)
    v0 = fm2.v.copy()
    sgd.learn(fm2, ds)
    fm2.close(discard=True)                        # trained, host arrays stale
    with pytest.raises(RuntimeError, match="discarded by close"):
        _ = fm2.v
    fm2.w0, fm2.w, fm2.v = 0.0, np.zeros(400), v0  # new parameters: the model is usable again
    np.testing.assert_array_equal(fm2.v, v0)
    assert np.isfinite(fm2.computeRMSE(ds))
    fm3 = fmhip.FMModel(399, 8, seed=3)
    sgd.learn(fm3, ds)
    fm3.close()                                    # the default still pulls: trained values survive the close
    assert not np.array_equal(fm3.v, v0)
    ds.unpersist()
    for m in (fm2, fm3):
        m.close()


def _free_port():
    """A rendezvous for torch.distributed that cannot collide: a file store in a fresh directory (a port found by binding to 0
    and closing can be taken by someone else before the ranks bind it again — seen once on a GPU box: EADDRINUSE)."""
    import tempfile
    return "file://" + os.path.join(tempfile.mkdtemp(prefix="fmhip_rdzv_"), "store")


@pytest.mark.parametrize("exchange", ["dense", "sharded"])
@pytest.mark.parametrize("upper", [(0.45,), (), (0.1, 0.3, 0.6)])
def test_library_side_exchange_one_rank(fmhip, upper, exchange):
    """fmhip_comm_create / fmhip_dp_plan / fmhip_dp_epoch with a world of ONE rank: RCCL is loaded, the
    communicator is built, the collectives of the overlapped schedule really run (in place, on the second
    stream: all-reduces, or the sharded update's reduce-scatter + all-gather) — and the result must be
    bit-identical to the plain dense step."""
    from sparkfm_amd import synth
    from sparkfm_amd.distributed import HipDataParallelSGD, RcclComm
    d = synth.make_zipf(91, 5000, 900, 4, 30, zipf_s=1.05)
    w0, w, v = synth.init_params(6, 900, 32, stdev=0.05)
    w = np.random.default_rng(1).normal(0, 0.05, 900)
    out = []
    for mode in ("dp", "plain"):
        ds = fmhip.DataSet.from_arrays(d, batch_rows=1500).cache()
        fm = fmhip.FMModel(899, 32)
        fm.w0, fm.w, fm.v = w0, w, v
        if mode == "dp":
            comm = RcclComm(fm, 0, 1)
            if len(upper) == 3:
                # hold the comm stream for every collective as a 40 GB/s all-reduce would: the schedule must not care
                from sparkfm_amd import _ffi as ffi
                ffi.check(ffi.load().fmhip_comm_emulate(comm.handle, 40.0))
                # ... spent by 8 workgroups streaming the payload through HBM (read, written back unchanged): not a bit may move
                ffi.check(ffi.load().fmhip_comm_emulate_load(comm.handle, 8))
            dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, upper_fractions=upper, exchange=exchange)
            for _ in range(2):
                dp.learn(fm, ds)
            assert len(dp.cuts) == len(upper) and dp.cuts == sorted(dp.cuts, reverse=True)
            assert dp.last_stats["rows"] == 500 and dp.last_stats["steps"] == 4
            comm.close()
        else:
            from sparkfm_amd import _ffi
            L = _ffi.load()
            for _ in range(2):
                for b in range(4):
                    _ffi.check(L.fmhip_step_compute(fm.handle, ds.handle, b))
                    _ffi.check(L.fmhip_step_apply(fm.handle, 0.05, 0.0, 1e-3, 1e-3))
            fm._device_updated()
        out.append((fm.w0, fm.w.copy(), fm.v.copy()))
        ds.unpersist()
        fm.close()
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


def test_touched_rows_exchange_one_rank_over_rccl(fmhip):
    """fmhip_dp_exchange(FMHIP_EXCHANGE_TOUCHED) with a world of one rank over real RCCL (ncclAllGather of the ids, the
    packed all-reduce, both in place): on a model far wider than the data the result must be bit-identical to the plain
    step, which takes the same rows-only update with lazy weight decay; the union is the batch's touched rows."""
    from sparkfm_amd import synth
    from sparkfm_amd.distributed import HipDataParallelSGD, RcclComm
    n1 = 200_000
    d = synth.make_zipf(93, 6000, 900, 4, 30, zipf_s=1.05)
    w0, w, v = synth.init_params(6, n1, 32, stdev=0.05)
    w = np.random.default_rng(1).normal(0, 0.05, n1)
    out = []
    for mode in ("touched", "plain"):
        ds = fmhip.DataSet.from_arrays(d, batch_rows=1500).cache()
        fm = fmhip.FMModel(n1 - 1, 32)
        fm.w0, fm.w, fm.v = w0, w, v
        if mode == "touched":
            comm = RcclComm(fm, 0, 1)
            dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-2, regv=1e-2, exchange="touched")
            for _ in range(3):
                dp.learn(fm, ds)
            info = dp.exchange_info()
            touched = [ds.batch_info(b)["n_columns"] for b in range(ds.n_batches)]
            assert info["mode"] == "touched" and info["id_slots_per_rank"] >= max(touched)
            assert min(touched) <= info["mean_union_rows"] <= max(touched) + 64 + 1       # + unused hot slots' padding entry
            assert dp.last_stats["rows"] == 1500 and dp.last_stats["steps"] == 4
            comm.close()
        else:
            sgd = fmhip.HipSGD(eta=0.05, regw=1e-2, regv=1e-2)
            for _ in range(3):
                sgd.learn(fm, ds)
        out.append((fm.w0, fm.w.copy(), fm.v.copy()))
        ds.unpersist()
        fm.close()
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


def test_library_side_exchange_two_ranks():
    """Two ranks, two GPUs, RCCL inside the library (fmhip_dp_epoch): bit-identical replicas that match the
    oracle over the equivalent global batches; uneven shards (rank 1 runs out of batches first).  Needs a
    second GPU: the round's test box has one, the driver's multi-GPU node runs it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import tempfile
    port, out = str(_free_port()), os.path.join(tempfile.mkdtemp(), "dp")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_rccl_worker.py"), str(r), "2", port, out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    np.testing.assert_array_equal(r0["v"], r1["v"])
    np.testing.assert_array_equal(r0["w"], r1["w"])
    assert float(r0["w0"]) == float(r1["w0"])
    w0, w, v = _oracle_two_rank_epochs()
    assert rel(r0["v"], v) <= 1e-5 and rel(r0["w"], w) <= 1e-5 and float(r0["w0"]) == pytest.approx(w0, rel=1e-5, abs=1e-7)


def _oracle_two_rank_epochs(n1=800):
    """The oracle over the global batches of dist_rccl_worker's two uneven shards (2 epochs of 3 steps)."""
    from sparkfm_amd import synth
    shards = [synth.make_zipf(77, 3000, 800, 4, 24, zipf_s=1.05, row_begin=0),
              synth.make_zipf(77, 1700, 800, 4, 24, zipf_s=1.05, row_begin=3000)]
    w0, w, v = synth.init_params(5, n1, 32, stdev=0.05)
    w = np.random.default_rng(9).normal(0, 0.05, n1)
    for _ in range(2):
        for j in range(3):                       # rank 0: 3 batches of 1000; rank 1: 2 (1000 + 700), then zeros
            rp, cols, vals, ys = [0], [], [], []
            for d in shards:
                n = len(d["y"])
                for r in range(j * 1000, min(n, (j + 1) * 1000)):
                    a, b = d["row_ptr"][r], d["row_ptr"][r + 1]
                    cols.append(d["col"][a:b])
                    vals.append(d["val"][a:b].astype(np.float64))
                    rp.append(rp[-1] + (b - a))
                    ys.append(float(d["y"][r]))
            w0, w, v, _ = oracle.sgd_step(w0, w, v, 0, len(ys), np.array(rp, np.int64), np.concatenate(cols),
                                          np.concatenate(vals), np.array(ys), 0.05, 0.0, 1e-3, 1e-3)
    return w0, w, v


@pytest.mark.parametrize("k,hot_ids_on_top", [(32, False), (16, False), (64, False), (32, True), (16, True), (64, True)])
def test_pipelined_exchange_one_rank_over_rccl(fmhip, k, hot_ids_on_top):
    """FMHIP_EXCHANGE_PIPELINED with one rank over real RCCL (every collective really runs, in place, on the second stream; the
    comm stream held as a 40 GB/s all-reduce would hold it, with the footprint): fmhip_dp_epoch, fmhip_dp_steps over a list of
    positions that wraps around the epoch, and single fmhip_dp_step_at calls give the SAME bits (the overlap changes when things
    run, not what they compute) and match the plain step up to the order of the forward's fp32 sums, and the fp64 oracle.
    hot_ids_on_top: the ids reversed, so the features of the dense hot block — those in a tenth of the rows and
    more — carry the HIGHEST ids, at or above the plan's top cut: their parameter rows are still being exchanged while pass A of the
    next position runs, so the block's prologue must run in pass B (FwdArgs::hot_in_b); read in pass A they would be one update
    behind, which nothing else would notice (the replicas stay identical)."""
    from sparkfm_amd import _ffi, synth
    from sparkfm_amd.distributed import HipDataParallelSGD, RcclComm
    L = _ffi.load()
    d = synth.make_zipf(93, 5000, 900, 4, 30, zipf_s=1.05)
    if hot_ids_on_top:
        d = dict(d, col=(899 - d["col"]).astype(np.int32))
    w0, w, v = synth.init_params(6, 900, k, stdev=0.05)
    w = np.random.default_rng(1).normal(0, 0.05, 900)
    positions = np.array([0, 1, 2, 3, 2, 0, 3, 1, 1], np.int64)
    out = {}
    for mode in ("epoch+steps", "stepwise", "plain"):
        ds = fmhip.DataSet.from_arrays(d, batch_rows=1500).cache()
        fm = fmhip.FMModel(899, k)
        fm.w0, fm.w, fm.v = w0, w, v
        if mode != "plain":
            comm = RcclComm(fm, 0, 1).selftest()
            _ffi.check(L.fmhip_comm_emulate(comm.handle, 40.0))
            _ffi.check(L.fmhip_comm_emulate_load(comm.handle, 8))
            dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, upper_fractions=(0.1, 0.3, 0.6), exchange="pipelined")
            dp.plan(fm, ds)
            assert len(dp.cuts) == 3
            if hot_ids_on_top:       # the case is what it claims: a feature of the forward's hot page sits at or above the top cut
                assert max(ds.layout()["hot_ids"]) >= max(dp.cuts)
            if mode == "epoch+steps":
                dp.learn(fm, ds)
                dp.steps_at(fm, ds, positions)                      # fmhip_dp_steps
            else:
                for p in list(range(4)) + positions.tolist():
                    dp.step_at(fm, ds, int(p))
            fm._device_updated()
            out[mode] = (fm.w0, fm.w.copy(), fm.v.copy())
            comm.close()
        else:
            for p in list(range(4)) + positions.tolist():
                _ffi.check(L.fmhip_step_compute(fm.handle, ds.handle, int(p)))
                _ffi.check(L.fmhip_step_apply(fm.handle, 0.05, 0.0, 1e-3, 1e-3))
            fm._device_updated()
            out[mode] = (fm.w0, fm.w.copy(), fm.v.copy())
        ds.unpersist()
        fm.close()
    a, b, c = out["epoch+steps"], out["stepwise"], out["plain"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.linalg.norm(a[2] - c[2]) <= 1e-5 * np.linalg.norm(c[2]) and np.linalg.norm(a[1] - c[1]) <= 1e-5 * np.linalg.norm(c[1])
    ow0, ow, ov = w0, w.copy(), v.copy()
    for p in list(range(4)) + positions.tolist():
        lo, hi = p * 1500, min(5000, (p + 1) * 1500)
        ow0, ow, ov, _ = oracle.sgd_step(ow0, ow, ov, lo, hi, d["row_ptr"], d["col"], d["val"].astype(np.float64), d["y"].astype(np.float64),
                                         0.05, 0.0, 1e-3, 1e-3)
    assert np.linalg.norm(a[2] - ov) <= 1e-5 * np.linalg.norm(ov) and np.linalg.norm(a[1] - ow) <= 1e-5 * np.linalg.norm(ow)


@pytest.mark.parametrize("exchange", ["dense", "sharded", "pipelined"])
@pytest.mark.parametrize("fractions,world", [("", 2), ("0.3", 2), ("0.05,0.15,0.3,0.55", 2), ("0.12,0.4", 3)])
def test_library_side_exchange_two_ranks_on_one_gpu_host_staged(fractions, world, exchange):
    """The library's data-parallel step with two (three) real ranks on the one GPU of the test box: fmhip_dp_epoch over
    fmhip_comm_create_external, every collective staged through the host and summed by gloo (RCCL refuses two ranks on
    one device).  Everything but the transport is the RCCL path: the plan broadcast from rank 0, the global row count,
    the interval schedule with 0 / 1 / 4 cuts, the rank that runs out of rows and contributes zeros, the step count
    agreed by a max-reduce.  Replicas bit-identical, same sequence of collectives on both ranks, oracle matched."""
    import tempfile
    port, out = str(_free_port()), os.path.join(tempfile.mkdtemp(), "dp")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_rccl_worker.py"), str(r), str(world), port, out, "host", fractions,
                               exchange]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0 = np.load(out + ".0.npz")
    for r in range(1, world):                    # world = 3: the third rank holds no rows at all and still keeps in step
        r1 = np.load(out + ".%d.npz" % r)
        np.testing.assert_array_equal(r0["v"], r1["v"])
        np.testing.assert_array_equal(r0["w"], r1["w"])
        assert float(r0["w0"]) == float(r1["w0"])
        np.testing.assert_array_equal(r0["calls"], r1["calls"])
        np.testing.assert_array_equal(r0["cuts"], r1["cuts"])
    n_cuts = len([x for x in fractions.split(",") if x])
    assert len(r0["cuts"]) == n_cuts and int(r0["steps"]) == 3 and int(r0["rows"]) == 1000      # last global batch: 1000 + 0 rows
    # per epoch: 1 max-reduce (step count + validation); per step: the row count + (1 region, or 3 per interval); per plan: 1 max-reduce + cuts
    sums = r0["calls"][r0["calls"][:, 0] == 0]
    if exchange in ("dense", "pipelined"):         # pipelined: the same collectives, the coldest interval's last
        per_step = 1 + (1 if n_cuts == 0 else 3 * (n_cuts + 1))
    else:
        # sharded: per interval a reduce-scatter of G_V (kind 4), two all-reduces (G_w with or without the scalars, G_b), an all-gather of V (kind 5)
        per_step = 1 + 2 * (n_cuts + 1)
        for kind in (4, 5):
            seg = r0["calls"][r0["calls"][:, 0] == kind][:, 1]
            assert len(seg) == 2 * 3 * (n_cuts + 1)
            # equal shares per rank: the intervals' shares add up to ceil(800 / world) rows of 32 floats
            assert seg.reshape(6, -1).sum(axis=1).tolist() == [-(-800 // world) * 32] * 6
    assert len(sums) == 2 * 3 * per_step
    w0, w, v = _oracle_two_rank_epochs()
    assert rel(r0["v"], v) <= 1e-5 and rel(r0["w"], w) <= 1e-5 and float(r0["w0"]) == pytest.approx(w0, rel=1e-5, abs=1e-7)


def test_bench_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --transport host`: the bench's whole N > 1 flow — self-launch of the ranks (before the
    parent touches the GPU), gloo control plane, the library's communicator, the cut / exchange-mode tuning, the timed
    region with its barriers and max over ranks, the exchange's own profile pass, the legs without exchange, the C3 twin,
    the JSON assembly — with both ranks on the one GPU of the test box and every collective staged through the host
    (RCCL refuses two ranks on one device).  Timings mean nothing here; the flow and the line's keys are what is tested."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--steps", "4", "--warmup", "2",
           "--rows", "200000", "--batch-rows", "100000", "--no-pmc", "--cpu-budget", "3", "--tune-budget", "12"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    # the record is re-written after every leg: every line is a complete JSON object, the LAST one is the record
    lines = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.strip()]
    stages = [ln["record"]["stage"] for ln in lines]
    # N > 1: a first headline with the default plan BEFORE the cut / mode sweep (a sweep that hangs on an unseen node must not cost
    # the record), then the headline of the plan the sweep chose; from there on the headline never changes
    assert stages[0].startswith("headline (default plan") and "headline" in stages[1:] and lines[-1]["record"]["final"] is True, stages
    assert "cut_tuning" not in lines[0]["exchange"] and lines[0]["value"] > 0 and lines[0]["exchange"]["mode"] == "dense"
    after = lines[stages.index("headline"):]
    assert all(ln["value"] == after[0]["value"] and ln["roofline"] is not None for ln in after)
    out = lines[-1]
    assert not out["legs"]["skipped"], out["legs"]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 4 and out["warmup"] == 2
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_rows_per_gpu"]        # one job: the global batch is what is fixed
    assert out["metric"] == "nnz_per_sec_fm_sgd_training" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["config"]["workload"].startswith("C4") and out["config"]["rows_per_gpu"] == 200000
    x = out["exchange"]
    assert x["nranks"] == 2 and x["transport"] == "host" and x["mode"] in ("dense", "sharded", "pipelined")
    # the communicator passed its self-test on every rank, and after the timed steps the replicas hold the same bits
    assert "passed" in x["selftest"] and x["replicas"]["identical"] is True and x["replicas"]["rows_compared"] > 1000
    assert {t["exchange"] for t in x["cut_tuning"]} == {"dense", "sharded", "pipelined"} and all(t["ms_per_step"] > 0 for t in x["cut_tuning"])
    assert x["exposed_comm_ms"] >= 0 and x["comm_busy_ms"] > 0
    assert x["c3_on_every_gpu"]["value"] > 0 and x["c3_on_every_gpu"]["exchange"] in ("dense", "sharded")
    assert x["per_gpu_without_exchange"]["value"] > 0 and x["c4_one_gpu"]["value"] > 0 and x["scaling_vs_c4_one_gpu"] > 0
    assert out["train"]["nonfinite"] == 0 and out["sustained"]["steps"] > 0
    # the N > 1 line carries the CPU baseline timed on rank 0 (the counter-based roofline of an N > 1 line is asserted by the
    # eight-rank bench test, which lets rank 0 run its rocprofv3 passes; --no-pmc here: this batch size has no committed pass)
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert out["roofline"]["frac"] is None and out["roofline"]["requested_only"]["requested_bytes"] > 0       # no counters, no HBM-side claim


@pytest.mark.parametrize("world,n1", [(2, 800), (3, 50_000)])
def test_touched_rows_exchange_on_one_gpu_host_staged(world, n1):
    """fmhip_dp_exchange(FMHIP_EXCHANGE_TOUCHED): the data-parallel step that exchanges only the gradient rows some rank
    touched — the unions of the lock-step schedule formed once, in the plan (all-gather of ids -> sort -> unique per step),
    the backward writing straight into a compact buffer with one row per union feature, one all-reduce of that buffer,
    rows-only update with lazy weight decay — with two and three real ranks on one GPU over the host-staged transport;
    uneven shards, a rank without rows, a model of 50,000 features of which the data touch 800 (the untouched rows must
    decay exactly as in the oracle's dense update).  Replicas bit-identical, same collectives everywhere, the oracle over
    the global batches matched."""
    import tempfile
    port, out = str(_free_port()), os.path.join(tempfile.mkdtemp(), "dp")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_rccl_worker.py"), str(r), str(world), port, out, "host", "",
                               "touched", str(n1)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0 = np.load(out + ".0.npz")
    for r in range(1, world):
        r1 = np.load(out + ".%d.npz" % r)
        np.testing.assert_array_equal(r0["v"], r1["v"])
        np.testing.assert_array_equal(r0["w"], r1["w"])
        assert float(r0["w0"]) == float(r1["w0"])
        np.testing.assert_array_equal(r0["calls"], r1["calls"])
    assert int(r0["steps"]) == 3 and int(r0["rows"]) == 1000
    kinds = r0["calls"][:, 0]
    # the plan (made once for both epochs): one id all-gather per step of the schedule; per step: the row count + the compact buffer
    assert (kinds == 3).sum() == 3 and (kinds == 0).sum() == 12
    assert 0 < float(r0["mean_union"]) <= 802                             # the union of the touched rows (+ one padding entry), not the model
    packed = r0["calls"][(kinds == 0) & (r0["calls"][:, 1] > 1)][:, 1]
    assert len(packed) == 6 and packed.max() <= 1664 + 804 * 32 and ((packed < 0.1 * n1 * 34).all() or n1 == 800)
    w0, w, v = _oracle_two_rank_epochs(n1)
    assert rel(r0["v"], v) <= 1e-5 and rel(r0["w"], w) <= 1e-5 and float(r0["w0"]) == pytest.approx(w0, rel=1e-5, abs=1e-7)


@pytest.mark.parametrize("n_rows,n1,k", [(2000, 25, 3), (10000, 300, 4), (10001, 300, 2)])
def test_als_sweeps_lds_and_fallback(fmhip, n_rows, n1, k):
    """The two ALS sweeps against the oracle (S/fm/lib/ALS.scala:15-75): columns far longer than the 128 entries
    the one-wave walk keeps in registers (25 features over 2000 rows), exactly the 10,000 rows whose e and q
    fill the LDS, and one row more (falls back to the workgroup-wide sweep over global memory)."""
    from helpers import random_problem
    a = random_problem(900 + n_rows + k, n_rows, n1, k, 1, min(12, n1 - 1))
    ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"]).cache()
    fm = fmhip.FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
    fm.reg0, fm.regw, fm.regv = 0.01, 0.1, 5.0
    w0, w, v = a["w0"], a["w"], a["v"]
    for _ in range(2):
        fmhip.HipALS.run().learn(fm, ds)
        w0, w, v = oracle.als_epoch(w0, w, v, 0.01, 0.1, 5.0, a["row_ptr"], a["col"], a["val"], a["y"])
    np.testing.assert_allclose(fm.v, v, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(fm.w, w, rtol=1e-8, atol=1e-11)
    assert fm.w0 == pytest.approx(w0, rel=1e-9, abs=1e-12)
    ds.unpersist()
    fm.close()


@pytest.mark.parametrize("n_rows,n1,k,long_col", [(30000, 9, 3, None), (12000, 40, 2, "1050"), (11000, 60, 2, "1")])
def test_als_long_columns_on_the_whole_chip(fmhip, monkeypatch, n_rows, n1, k, long_col):
    """Datasets past the LDS sweep follow a launch plan made from the column lengths (als_kernels.hip): columns of at
    least kAlsLongColumn = 8192 entries take the chip-wide step (partial sums per workgroup, a fixed tree over the
    partials, the update of e and q slice by slice), runs of shorter ones a single workgroup.  9 features over 30,000
    rows: every column is long by size; FMHIP_ALS_LONG lowers the mark so that long and short columns alternate
    (1050) and so that EVERY column, down to single entries, goes through the two launches (1).  Two epochs against
    the fp64 oracle; quirk Q1 (the last slot is never trained) included."""
    from helpers import random_problem
    if long_col:
        monkeypatch.setenv("FMHIP_ALS_LONG", long_col)
    a = random_problem(7000 + n_rows + k, n_rows, n1, k, 1, min(6, n1 - 1))
    ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"]).cache()
    lens = np.bincount(a["col"], minlength=n1)
    mark = int(long_col) if long_col else 8192
    assert (lens >= mark).any() and (long_col != "1050" or (lens < mark).any())
    fm = fmhip.FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
    fm.reg0, fm.regw, fm.regv = 0.01, 0.1, 5.0
    w0, w, v = a["w0"], a["w"], a["v"]
    for _ in range(2):
        fmhip.HipALS.run().learn(fm, ds)
        w0, w, v = oracle.als_epoch(w0, w, v, 0.01, 0.1, 5.0, a["row_ptr"], a["col"], a["val"], a["y"])
    np.testing.assert_allclose(fm.v, v, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(fm.w, w, rtol=1e-8, atol=1e-11)
    assert fm.w0 == pytest.approx(w0, rel=1e-9, abs=1e-12)
    assert np.array_equal(fm.v[:, n1 - 1], a["v"][:, n1 - 1]) and fm.w[n1 - 1] == a["w"][n1 - 1]      # quirk Q1
    ds.unpersist()
    fm.close()


def _one_hot_fields(seed, n_rows, sizes, extra=0):
    """Rows with one id per field (fields = consecutive id ranges of the given sizes), like the reference's MovieLens
    demo (S/driver.scala:73-113: a user field and an item field); `extra`: that many more random ids from a last, shared range."""
    rng = np.random.default_rng(seed)
    cols, off = [], 0
    for sz in sizes:
        cols.append(off + rng.integers(0, sz, n_rows))
        off += sz
    col = np.stack(cols, axis=1)
    if extra:
        col = np.concatenate([col, off + np.stack([rng.permutation(40)[:extra] for _ in range(n_rows)])], axis=1)
        off += 40
    nnz_r = col.shape[1]
    val = np.where(rng.random(col.shape) < 0.5, 1.0, rng.uniform(0.2, 1.0, col.shape))
    y = rng.normal(3.5, 1.0, n_rows)
    return dict(row_ptr=np.arange(0, (n_rows + 1) * nnz_r, nnz_r, dtype=np.int64), col=col.reshape(-1).astype(np.int32), val=val.reshape(-1).astype(np.float64),
                y=y, n1=off)


@pytest.mark.parametrize("n_rows,sizes,extra,k", [(9000, (300, 200), 0, 4), (9000, (120, 80, 50), 0, 3), (60000, (2000, 1500), 0, 4), (30000, (500, 400), 2, 2)])
def test_als_level_schedule_is_the_sequential_sweep(fmhip, monkeypatch, n_rows, sizes, extra, k):
    """ALS.learn's sweep (S/fm/lib/ALS.scala:36-70) level by level: columns that share no row commute exactly, so the columns
    of a level run side by side and the result must be THE sequential sweep's — bit for bit against the one-wave LDS walk
    (<= 10,000 rows: the same column step, the same summation order), to fp64 rounding against the workgroup walk of larger
    datasets (another summation order inside a column), and 1e-8 against the oracle.  One-hot fields give one level per
    field; a last range of shared ids (extra) adds levels of a few columns behind them."""
    d = _one_hot_fields(7, n_rows, sizes, extra)
    n1 = d["n1"]
    ds = fmhip.DataSet(d["row_ptr"], d["col"], d["val"], d["y"]).cache()
    lv = ds.alsLevels()
    assert lv["columns"] == ds.batch_info(0)["n_columns"] and lv["levels"] >= len(sizes)
    if not extra:
        assert lv["levels"] == len(sizes) and lv["widest_level"] <= max(sizes)       # one level per field
    rng = np.random.default_rng(3)
    w0, w, v = 0.3, rng.normal(0, 0.1, n1), rng.normal(0, 0.1, (k, n1))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FMHIP_ALS_LEVELS", mode)
        fm = fmhip.FMModel(n1 - 1, k)
        fm.w0, fm.w, fm.v = w0, w, v
        als = fmhip.HipALS.run()
        for _ in range(2):
            als.learn(fm, ds)
        out[mode] = (fm.w0, fm.w.copy(), fm.v.copy())
        fm.close()
    monkeypatch.delenv("FMHIP_ALS_LEVELS")
    if n_rows <= 10000:
        assert out["1"][0] == out["0"][0]
        np.testing.assert_array_equal(out["1"][1], out["0"][1])
        np.testing.assert_array_equal(out["1"][2], out["0"][2])
    else:
        assert abs(out["1"][0] - out["0"][0]) <= 1e-12 and np.abs(out["1"][1] - out["0"][1]).max() <= 1e-11 and np.abs(out["1"][2] - out["0"][2]).max() <= 1e-11
    o0, ow, ov = w0, w, v
    for _ in range(2):
        o0, ow, ov = oracle.als_epoch(o0, ow, ov, 0.0, 0.0, 10.0, d["row_ptr"], d["col"], d["val"], d["y"])
    assert abs(out["1"][0] - o0) <= 1e-8 and np.abs(out["1"][1] - ow).max() <= 1e-8 and np.abs(out["1"][2] - ov).max() <= 1e-8
    # the default choice: wide levels -> the level sweep (the same bits as forcing it)
    fm = fmhip.FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = w0, w, v
    als = fmhip.HipALS.run()
    for _ in range(2):
        als.learn(fm, ds)
    if lv["levels"] * 16 <= lv["columns"]:
        np.testing.assert_array_equal(fm.v, out["1"][2])
    fm.close()
    ds.unpersist()


def test_als_refuses_rows_with_a_repeated_feature(fmhip):
    from sparkfm_amd import _ffi
    ds = fmhip.DataSet.from_rows([(1.0, ([0, 2, 2], [1.0, 2.0, 0.5])), (0.0, ([1], [1.0]))]).cache()
    fm = fmhip.FMModel(2, 2)
    with pytest.raises(_ffi.FmhipError, match="twice"):
        fmhip.HipALS.run().learn(fm, ds)
    assert np.isfinite(fm.computeRMSE(ds))          # everything else accepts such rows


@pytest.mark.parametrize("n1", [30000, 700])
@pytest.mark.parametrize("k,regs", [(8, (0.01, 1e-3, 1e-3)), (32, (0.0, 1e-3, 2e-3)), (32, (0.0, 0.0, 0.0)), (64, (0.01, 0.0, 1e-3)),
                                    (100, (0.0, 1e-3, 1e-3))])
def test_where_the_update_runs_does_not_change_a_bit(fmhip, k, regs, n1):
    """Three places for the same update: a launch of its own (keys 10 = 11 = 0), inside the fixup launch beside the
    fixups (key 11, the default; taken when the update is the dense pass: n1 = 700) and inside the column walk's
    flush (key 10; rows-only update: n1 = 30000 is far wider than a batch).  Same operations on the same values ->
    the same bits, for packed rows (k=8), plain rows, wide rows, hot block on, cut columns (hot features spanning
    many ranges), with and without (lazy) weight decay; and all of them track the oracle."""
    from sparkfm_amd import _ffi
    from test_gpu_parity import hot_problem
    L = _ffi.load()
    a, _ = hot_problem(500 + k, 4000, n1, k, 9)
    outs = []
    for fused, merged in ((0, 1), (0, 0), (1, 0)):
        L.fmhip_tune(_ffi.TUNE_FUSED_UPDATE, fused)
        L.fmhip_tune(_ffi.TUNE_MERGED_FINISH, merged)
        try:
            ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"], batch_rows=1100).cache()
            fm = fmhip.FMModel(a["n1"] - 1, k)
            fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
            sgd = fmhip.HipSGD(eta=0.05, reg0=regs[0], regw=regs[1], regv=regs[2])
            for _ in range(3):
                sgd.learn(fm, ds)
            outs.append((fm.w0, fm.w.copy(), fm.v.copy(), sgd.last_stats["sse"]))
        finally:
            L.fmhip_tune(_ffi.TUNE_FUSED_UPDATE, 0)
            L.fmhip_tune(_ffi.TUNE_MERGED_FINISH, 1)
        ds.unpersist()
        fm.close()
    # merged == plain always; fused (rows-only form) == plain when the plain update is rows-only too, or there is no decay
    assert outs[0][0] == outs[1][0] and outs[0][3] == outs[1][3]
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    if n1 == 30000 or regs[1:] == (0.0, 0.0):
        assert outs[2][0] == outs[1][0]
        np.testing.assert_array_equal(outs[2][1], outs[1][1])
        np.testing.assert_array_equal(outs[2][2], outs[1][2])
    w0, w, v = a["w0"], a["w"], a["v"]
    for _ in range(3):
        w0, w, v, _ = oracle.sgd_epoch(w0, w, v, 1100, a["row_ptr"], a["col"], a["val"], a["y"], 0.05, *regs)
    for o in outs:
        assert rel(o[2], v) <= 1e-5 and rel(o[1], w) <= 1e-5 and o[0] == pytest.approx(w0, rel=1e-5, abs=1e-7)


def test_per_dataset_layout_options(fmhip):
    """fmhip_dataset_create_opts: the dense hot block is chosen per dataset, not through process-wide state; both
    layouts report every stored nonzero and give the same gradient."""
    from test_gpu_parity import hot_problem
    a, hot_ids = hot_problem(808, 3000, 500, 32, 12)
    grads = []
    for hb in (True, False):
        ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"], batch_rows=700, hot_block=hb).cache()
        lay = ds.layout()
        assert (len(lay["hot_ids"]) > 0) == hb and (lay["nnz_sparse"] < ds.nnz) == hb
        if hb:
            assert set(lay["hot_ids"]) <= set(int(h) for h in hot_ids)
        fm = fmhip.FMModel(a["n1"] - 1, a["k"])
        fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
        gv, gw, g0, st = fm.batchGradient(ds, 1)
        assert st["nnz"] == a["row_ptr"][1400] - a["row_ptr"][700]
        grads.append((gv, gw))
        ds.unpersist()
        fm.close()
    check_grad(grads[0][0], grads[0][1], grads[1][0], grads[1][1], np.abs(a["v"]).max())


def test_bench_fallback_exchange_over_torch_distributed():
    """`bench.py --exchange torch`: the exchange the bench falls back to when the library's own RCCL communicator cannot be
    created on a node (bench.py: "fell back to torch.distributed") — the split step (fmhip_step_compute / fmhip_step_apply) with
    the packed gradient all-reduced by torch.distributed's nccl backend on the library's stream.  One rank here (a one-rank nccl
    group is what the test box can form): the flow, the line and a finite training are what is tested."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dp", "--exchange", "torch", "--config", "C4",
           "--rows", "200000", "--batch-rows", "100000", "--steps", "4", "--warmup", "2", "--no-extra", "--no-pmc", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.strip()][-1])
    assert out["config"]["exchange"] == "torch" and "torch.distributed" in out["config"]["allreduce"]
    assert out["value"] > 0 and out["train"]["nonfinite"] == 0 and out["config"]["batch_rows_per_gpu"] == 100000


def test_jni_shim_end_to_end_without_a_jvm(fmhip, tmp_path):
    """The JNI shim's model / dataset / training / scoring / communicator natives, driven by tests/jni_harness.c through an
    in-memory JNIEnv (copying arrays, poisoned on release) on the GPU: two SGD epochs, parameters, RMSE, predictions over a
    dataset and over loose rows, a scoring-only dataset, and a one-rank RCCL communicator's plan, epoch and ordered epoch —
    each bit for bit what the same calls through the C ABI give; library errors arrive as RuntimeException with
    fmhip_last_error()'s text."""
    import subprocess
    from helpers import build_jni_harness
    r = subprocess.run([build_jni_harness(tmp_path), "gpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert "jni_harness gpu:" in r.stdout.decode() and "checks ok" in r.stdout.decode()


def test_a_single_batch_dataset_takes_the_hot_block_when_asked(fmhip):
    """Full-batch SGD: a dataset of ONE batch is what the ALS learner walks, so by default its transpose stays whole (no dense
    hot block); asked for by name (fmhip_dataset_opts::hot_block >= 1) the block is built, full-batch SGD steps track the oracle
    on it, and fmhip_als_epoch refuses that dataset — the default one still serves both learners."""
    from test_gpu_parity import hot_problem
    a, hot_ids = hot_problem(909, 2500, 400, 32, 10)
    regs = (0.0, 1e-3, 1e-3)
    w0, w, v = a["w0"], a["w"], a["v"]
    for _ in range(3):
        w0, w, v, _ = oracle.sgd_epoch(w0, w, v, 0, a["row_ptr"], a["col"], a["val"], a["y"], 0.05, *regs)
    for hb in (None, 4):
        ds = fmhip.DataSet(a["row_ptr"], a["col"], a["val"], a["y"], batch_rows=0, hot_block=hb).cache()
        assert ds.n_batches == 1
        lay = ds.layout()
        assert (len(lay["hot_ids"]) > 0) == (hb is not None) and (lay["nnz_sparse"] < ds.nnz) == (hb is not None)
        fm = fmhip.FMModel(a["n1"] - 1, a["k"])
        fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
        sgd = fmhip.HipSGD(eta=0.05, reg0=regs[0], regw=regs[1], regv=regs[2])
        for _ in range(3):
            sgd.learn(fm, ds)
        assert rel(fm.v, v) <= 1e-5 and rel(fm.w, w) <= 1e-5 and fm.w0 == pytest.approx(w0, rel=1e-5, abs=1e-7)
        from sparkfm_amd._ffi import FmhipError
        als = fmhip.HipALS()
        fm.reg0, fm.regw, fm.regv = 0.0, 0.01, 0.01
        if hb is None:
            als.learn(fm, ds)                                           # the whole transpose is there
        else:
            with pytest.raises(FmhipError, match="hot block"):
                als.learn(fm, ds)
        ds.unpersist()
        fm.close()


def test_permutation_of_a_rows_nonzeros(fmhip):
    """SURVEY §4's property: a row is a set of (index, value) pairs — storing them in another order (the reference's
    loader neither sorts nor reorders, S/fm/FMUtils.scala:28-36) changes the prediction and the gradient only by fp32
    summation order, and the transposes not at all (rows ascend inside a column whatever the stored order)."""
    from helpers import random_problem
    a = random_problem(4242, 2500, 300, 32, 0, 40, empty_rows=(3,), sort_idx=True)
    b = {k_: (v.copy() if isinstance(v, np.ndarray) else v) for k_, v in a.items()}
    rng = np.random.default_rng(1)
    for r in range(2500):
        s = slice(a["row_ptr"][r], a["row_ptr"][r + 1])
        p = rng.permutation(s.stop - s.start)
        b["col"][s] = a["col"][s][p]
        b["val"][s] = a["val"][s][p]
    res = []
    for x in (a, b):
        ds = fmhip.DataSet(x["row_ptr"], x["col"], x["val"], x["y"], batch_rows=900).cache()
        fm = fmhip.FMModel(x["n1"] - 1, x["k"])
        fm.w0, fm.w, fm.v = x["w0"], x["w"], x["v"]
        res.append((fm.predict(ds), fm.batchGradient(ds, 1), ds.transposeInput(1)))
        ds.unpersist()
        fm.close()
    sc = term_scale(a)
    assert (np.abs(res[0][0] - res[1][0]) <= 2 * TOL_Y * sc).all()
    check_grad(res[0][1][0], res[0][1][1], res[1][1][0], res[1][1][1], np.abs(a["v"]).max())
    for u, v in zip(res[0][2], res[1][2]):
        np.testing.assert_array_equal(u, v)
    oy = oracle.predict(a["w0"], a["w"], a["v"], a["row_ptr"], a["col"], a["val"])
    assert (np.abs(res[1][0] - oy) <= TOL_Y * sc).all()


def test_c5_real_width_on_the_rows_a_small_dataset_touches(fmhip):
    """BASELINE config 5's REAL width — 2^25 hashed slots x k=64, V = 8.6 GB, so the forward takes its flat-address
    kernels by size (not through the test knob) — with a dataset small enough for the oracle: the model is drawn on
    the device, the parameters of the ~10^5 touched features are fetched (fmhip_model_get_rows), and the oracle runs the
    equivalent compact problem (feature ids remapped to 0..m-1).  Predictions and one epoch of 3 steps with (lazy) weight
    decay must agree on every touched row; untouched rows must have decayed by (1 - eta*lambda)^3 and nothing else."""
    from sparkfm_amd import synth
    n1, k, br = 1 << 25, 64, 10_000
    d = synth.make_config("C5", rows=30_000)
    assert d["col"].max() < n1 and d["col"].max() > (1 << 24)            # really uses the upper half of the id space
    ds = fmhip.DataSet.from_arrays(d, batch_rows=br).cache()
    fm = fmhip.FMModel(n1 - 1, k, init_stdev=0.02, seed=3, init_on_device=True)
    feats = np.unique(d["col"]).astype(np.int32)
    ccol = np.searchsorted(feats, d["col"]).astype(np.int32)
    rp, val, y = d["row_ptr"], d["val"].astype(np.float64), d["y"].astype(np.float64)
    w_t, v_t = fm.rows(feats)
    assert not w_t.any() and abs(v_t.std() - 0.02) < 2e-4
    rng = np.random.default_rng(0)
    others = np.setdiff1d(rng.integers(0, n1, 3000).astype(np.int32), feats)[:1000]
    _, v_u0 = fm.rows(others)
    # forward at full width
    yh = fm.predict(ds)
    oy = oracle.predict(0.0, w_t, v_t, rp, ccol, val, threads=8)
    sc = term_scale(dict(y=oy[:2000], row_ptr=rp[:2001], col=ccol, val=val, w0=0.0, w=w_t, v=v_t))
    assert (np.abs(yh[:2000] - oy[:2000]) <= TOL_Y * sc).all()
    assert np.abs(yh - oy).max() <= 1e-4 * (1 + np.abs(oy).max())
    # one epoch = 3 steps with weight decay: rows-only update + lazy decay on the GPU, eager dense decay in the oracle
    eta, regs = 0.05, (0.0, 1e-3, 1e-3)
    sgd = fmhip.HipSGD(eta=eta, reg0=regs[0], regw=regs[1], regv=regs[2])
    sgd.learn(fm, ds)
    ow0, ow, ov, osse = oracle.sgd_epoch(0.0, w_t, v_t, br, rp, ccol, val, y, eta, *regs, threads=8)
    assert sgd.last_stats["sse"] == pytest.approx(osse, rel=2e-5) and sgd.last_stats["nonfinite"] == 0
    w_g, v_g = fm.rows(feats)
    assert rel(v_g, ov) <= 1e-5 and rel(w_g, ow) <= 1e-5 and fm.w0 == pytest.approx(ow0, rel=1e-5, abs=1e-7)
    _, v_u = fm.rows(others)
    np.testing.assert_allclose(v_u, v_u0 * (1.0 - eta * regs[2]) ** 3, rtol=2e-6, atol=0)
    assert fm.computeRMSE(ds) == pytest.approx(oracle.rmse(ow0, ow, ov, rp, ccol, val, y, threads=8), rel=1e-5)
    ds.unpersist()
    fm.close()


@pytest.mark.parametrize("regs", [(0.0, 0.0, 0.0), (0.0, 0.05, 0.05)])
def test_training_on_relabelled_ids_is_the_same_model(fmhip, regs):
    """FeatureOrder (fmhip_feature_counts / _rank_from_counts / _relabel_columns): ids that arrive in an arbitrary
    order are relabelled by frequency, the model trains in the internal numbering, parameters come back in the
    caller's — a pure renaming, so the oracle run on the CALLER's ids must be matched (prediction, two epochs of SGD
    with and without decay, RMSE), and the frequent features really end up at the small ids (hot block = ids 0..15)."""
    from sparkfm_amd import FeatureOrder, synth
    d = synth.make_zipf(909, 3000, 700, 0, 30, zipf_s=1.05)
    rng = np.random.default_rng(2)
    a = dict(n1=700, k=32, row_ptr=d["row_ptr"], col=d["col"], val=d["val"].astype(np.float64), y=d["y"].astype(np.float64),
             w0=0.05, w=rng.normal(0, 0.1, 700), v=rng.normal(0, 0.1, (32, 700)))
    perm = rng.permutation(a["n1"]).astype(np.int32)
    col = perm[a["col"]]                                           # the caller's (scrambled) ids
    w, v = np.empty_like(a["w"]), np.empty_like(a["v"])
    w[perm], v[:, perm] = a["w"], a["v"]                           # the same model under those ids
    order = FeatureOrder.fit(col, a["n1"])
    ds = fmhip.DataSet(a["row_ptr"], order.relabel(col), a["val"], a["y"], batch_rows=1000).cache()
    hot = ds.layout()["hot_ids"]
    assert len(hot) > 0 and max(hot) < 2 * len(hot)                # the hot block's features are the lowest ids
    fm = fmhip.FMModel(a["n1"] - 1, a["k"])
    order.set_params(fm, a["w0"], w, v)
    yh = fm.predict(ds)
    oy = oracle.predict(a["w0"], w, v, a["row_ptr"], col, a["val"])
    assert (np.abs(yh - oy) <= TOL_Y * term_scale(dict(a, col=col, w=w, v=v))).all()
    sgd = fmhip.HipSGD(eta=0.05, reg0=regs[0], regw=regs[1], regv=regs[2])
    ow0, ow, ov = a["w0"], w, v
    for _ in range(2):
        sgd.learn(fm, ds)
        ow0, ow, ov, _ = oracle.sgd_epoch(ow0, ow, ov, 1000, a["row_ptr"], col, a["val"], a["y"], 0.05, *regs)
    g0, gw, gv = order.get_params(fm)
    assert np.abs(gv - ov).max() <= 2e-4 * np.abs(ov).max() and np.abs(gw - ow).max() <= 2e-4 * max(np.abs(ow).max(), 1e-30)
    assert g0 == pytest.approx(ow0, rel=1e-4, abs=1e-6)
    assert fm.computeRMSE(ds) == pytest.approx(oracle.rmse(ow0, ow, ov, a["row_ptr"], col, a["val"], a["y"]), rel=1e-4)
    ds.unpersist()
    fm.close()


def test_relabelling_on_the_gpu_is_the_host_numbering_bit_for_bit(fmhip):
    """fmhip_feature_counts_gpu / fmhip_rank_from_counts_gpu / fmhip_relabel_columns_gpu against the host arithmetic they
    replace: the same counts, the same order (descending count, ties by ascending id — many ties here, and ids that never
    occur), the same relabelled stream; counts accumulate over partitions; an id outside [0, n1) is refused with its position."""
    from sparkfm_amd import FeatureOrder, _ffi
    rng = np.random.default_rng(17)
    n1 = 300_007
    col = np.minimum((rng.pareto(0.9, 5_000_000) * 40).astype(np.int64), n1 - 1).astype(np.int32)     # heavy head, long tail of ties
    a, b = col[:2_000_000], col[2_000_000:]
    ch = FeatureOrder.counts(b, n1, into=FeatureOrder.counts(a, n1))
    cg = FeatureOrder.counts(b, n1, into=FeatureOrder.counts(a, n1, device=0), device=0)
    np.testing.assert_array_equal(ch, cg)
    assert (ch == 0).sum() > 1000 and np.bincount(ch[ch > 0]).max() > 1000            # absent ids and plenty of ties
    oh, og = FeatureOrder.from_counts(ch), FeatureOrder.from_counts(cg, device=0)
    np.testing.assert_array_equal(oh.rank, og.rank)
    np.testing.assert_array_equal(oh.by_rank, og.by_rank)
    np.testing.assert_array_equal(oh.relabel(col), og.relabel(col))
    bad = col.copy()
    bad[1234567] = n1
    with pytest.raises(_ffi.FmhipError, match="1234567"):
        og.relabel(bad)
    with pytest.raises(_ffi.FmhipError, match="1234567"):
        FeatureOrder.counts(bad, n1, device=0)


def test_bench_killed_inside_an_optional_leg_has_left_its_record():
    """the bench line must not be losable.  `bench.py` writes the whole record as soon as the headline
    exists and re-writes it after every leg; here the HBM-resident leg is made to hang (FMHIP_BENCH_TEST_STALL), the process is
    killed once the line before it has appeared, and the last complete line on stdout is a valid record: the driver's
    keys, `roofline` with HIP-event kernel times and what bounds the step, `cpu_baseline` — everything but the legs that never ran."""
    import json
    import signal
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--rows", "200000", "--batch-rows", "100000",
           "--no-pmc", "--cpu-budget", "2", "--settle", "0.1"]
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")},
               FMHIP_BENCH_TEST_STALL="hbm_resident")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    os.set_blocking(p.stdout.fileno(), False)
    t0, data = time.time(), b""
    while time.time() - t0 < 300 and b'"stage": "extra.scoring"' not in data and p.poll() is None:
        data += p.stdout.read() or b""
        time.sleep(0.2)
    time.sleep(1.0)                       # it is now inside the stalled leg
    assert p.poll() is None, p.stderr.read().decode()[-3000:]
    p.send_signal(signal.SIGKILL)
    p.wait()
    data += p.stdout.read() or b""
    import bench
    out = bench.last_record(data.decode())
    assert out is not None and out["record"]["stage"] == "extra.scoring" and out["record"]["final"] is False
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["steps"] == 12 and out["warmup"] == 3 and out["value"] > 0 and out["config"]["settle_steps_before_warmup"] > 0
    rf = out["roofline"]
    assert rf["avg_launch_ms"] > 0 and rf["kernel"] in ("k_forward", "k_backward") and "HIP events" in rf["avg_launch_ms_from"]
    assert rf["compulsory_hbm_bytes_per_step"] > 0 and 0 < rf["hbm_floor_ms"] < out["ms_per_step"] and 0 < rf["step_ceiling_frac"] <= 1.0
    assert out["cpu_baseline"]["value"] > 0 and out["sustained"]["steps"] > 0 and out["extra"]["scoring"]["value"] > 0
    assert "hbm_resident" not in out["extra"]
    lines = [json.loads(ln) for ln in data.decode().splitlines() if ln.strip().startswith("{") and ln.strip().endswith("}")]
    assert [ln["record"]["stage"] for ln in lines][:2] == ["headline", "cpu_baseline"] and all(ln["value"] == out["value"] for ln in lines)


def test_bench_under_the_drivers_launcher():
    """The driver starts an N > 1 bench as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` (one rank per GPU, env:// rendezvous from RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*).  That launch path — not the bench's own spawn_ranks — with two ranks on the one GPU of the test box
    (`--transport host`: RCCL refuses two ranks on a device): the launcher's environment (OMP_NUM_THREADS=1 and all) reaches the
    ranks, the gloo control plane forms over env://, rank 0's JSON lines are the launcher's stdout and nothing else is, the
    first headline comes BEFORE the cut / mode sweep, the final line is complete."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FMHIP_BENCH_RDZV")}
    for attempt in range(3):
        # the launcher wants a PORT (the bench's own self-launch meets through a file store instead): one found by binding to 0 and
        # closing can be taken by someone else before the launcher binds it — seen once on a GPU box — so that, and only that, is retried
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--steps", "4", "--warmup", "2",
               "--rows", "200000", "--batch-rows", "100000", "--no-pmc", "--cpu-budget", "2", "--tune-budget", "6"]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        if r.returncode == 0 or not any(m in r.stderr.decode().lower() for m in ("eaddrinuse", "address already in use")):
            break
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    out_lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert all(ln.startswith("{") and ln.endswith("}") for ln in out_lines), [ln[:80] for ln in out_lines if not ln.startswith("{")]
    lines = [json.loads(ln) for ln in out_lines]
    stages = [ln["record"]["stage"] for ln in lines]
    assert stages[0].startswith("headline (default plan") and "headline" in stages[1:] and lines[-1]["record"]["final"] is True, stages
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    x = out["exchange"]
    assert x["nranks"] == 2 and "passed" in x["selftest"] and x["replicas"]["identical"] is True
    assert {t["exchange"] for t in x["cut_tuning"]} == {"dense", "sharded", "pipelined"}
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 2          # the launcher's OMP_NUM_THREADS=1 did not pin the baseline to one thread
    assert out["sustained"]["steps"] > 0 and not out["legs"]["skipped"]


def test_cpp_mirror_runs_the_reference_flow_bit_for_bit_with_the_python_mirror(tmp_path):
    """include/sparkfm.hpp on the GPU: tests/cpp_mirror.cpp runs the reference's own flow — FM(dataset, k, maxIteration)
    .learnWith(learner), computeRMSE / predict on a second dataset (S/fm/impl/FactorizationMachines.scala:30-51,
    S/driver.scala:100-112) — with HipSGD over mini-batches and with HipALS (the reference's learner) on a problem made of exact
    rationals; the same flow through the Python mirror makes the same C-ABI calls, so every number must be the SAME BITS: the
    RMSE of every iteration, w0 / w / v after the fit, held-out RMSE, every prediction; and the oracle is matched."""
    import numpy as np
    from helpers import build_cpp_mirror
    from sparkfm_amd import FM, DataSet, HipALS, HipSGD
    path = str(tmp_path / "cpp.bin")
    r = subprocess.run([build_cpp_mirror(tmp_path), "gpu", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"flow ok" in r.stdout, r.stderr.decode()[-2000:]
    got = np.fromfile(path, dtype=np.float64)
    n_rows, n1, k = 3000, 97, 8
    rp, col, val, y = [0], [], [], []
    for rr in range(n_rows):
        for j in range(3 + rr % 5):
            col.append((rr * 7 + j * 31) % n1)
            val.append(0.25 + ((rr + 3 * j) % 8) / 8.0)
        rp.append(len(col))
        y.append(((rr * 37) % 11) / 5.0 - 1.0)
    rp, col, val, y = np.array(rp, np.int64), np.array(col, np.int32), np.array(val), np.array(y)
    dim = int(col.max())
    i = np.arange(dim + 1)
    w0 = 0.125
    w = ((i * 29) % 17 - 8) / 160.0
    v = np.asfortranarray(((np.arange(k)[:, None] * 7 + i[None, :] * 13) % 23 - 11) / 220.0)
    want = []
    # HipSGD
    ds = DataSet(rp, col, val, y, batch_rows=700)
    fit = FM(ds, k, maxIteration=3)
    fm = fit.learnWith(HipSGD.run(eta=0.05, regw=1e-3, regv=1e-3), init=(w0, w, v))
    test = DataSet(rp, col, val, y, batch_rows=0).cache()
    sgd_v = fm.v.copy()
    want += list(fit.rmse_history) + [fm.w0] + list(fm.w) + list(fm.v.ravel(order="F")) + [fm.computeRMSE(test)] + list(fm.predict(test))
    a, b = int(rp[5]), int(rp[6])
    want += [fm.predict((col[a:b], val[a:b])), float(n_rows)]              # (HipSGD.last_stats: the epoch's totals)
    test.unpersist()
    fm.close()
    # HipALS
    ds1 = DataSet(rp, col, val, y, batch_rows=0)
    fit1 = FM(ds1, k, maxIteration=2)
    fm1 = fit1.learnWith(HipALS.run(), init=(w0, w, v))
    ds1.cache()
    want += list(fit1.rmse_history) + [fm1.w0] + list(fm1.w) + list(fm1.v.ravel(order="F")) + [fm1.computeRMSE(ds1)]
    als = (fm1.w0, fm1.w.copy(), fm1.v.copy())
    ds1.unpersist()
    fm1.close()
    want = np.array(want, np.float64)
    assert got.shape == want.shape
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, (bad[:10].tolist(), got[bad[:5]].tolist(), want[bad[:5]].tolist(), "sections: sgd rmse 0-2, w0 3, w 4-%d, v -%d, ..." % (3 + dim + 1, 3 + (dim + 1) * (k + 1)))
    # ... and both are the oracle's: 3 SGD epochs / 2 ALS epochs from the injected parameters
    o0, ow, ov = w0, w.copy(), v.copy()
    for _ in range(3):
        o0, ow, ov, _ = oracle.sgd_epoch(o0, ow, ov, 700, rp, col, val, y, 0.05, 0.0, 1e-3, 1e-3)
    assert np.linalg.norm(sgd_v - ov) <= 1e-5 * np.linalg.norm(ov)
    a0, aw, av = w0, w.copy(), v.copy()
    for _ in range(2):
        a0, aw, av = oracle.als_epoch(a0, aw, av, 0.0, 0.0, 10.0, rp, col, val, y)
    assert abs(als[0] - a0) <= 1e-8 and np.abs(als[1] - aw).max() <= 1e-8 and np.abs(als[2] - av).max() <= 1e-8
