"""BASELINE.json's two multi-GPU configurations at their FULL size on one GPU (288 GB of HBM holds them), as records
rather than anecdotes — size-independent properties where the oracle cannot follow:

  C4  all 10M rows x 1M features, k=32 (400M stored nonzeros), mini-batches of 625k rows
  C5  2^24 Criteo-shaped rows x 2^25 hashed slots, k=64 (V = 8.6 GB; ~590M stored nonzeros), mini-batches of 250k rows

For each: the loss falls over an epoch, no non-finite prediction, ONE step repeated from the same state gives the same
bits, and the predictions of a random sample of rows match the fp64 oracle run on exactly those rows (full model at C4;
at C5 the compact problem over the features the sample touches, as test_c5_real_width_* does).  Marked `slow`: about a
minute for the pair, most of it generating and uploading the rows.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from test_gpu_parity import TOL_Y, term_scale

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def fmhip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sparkfm_amd
    return sparkfm_amd


def _sample(d, n_sample, seed):
    """A random sample of rows as a CSR problem of its own: (row ids, sub_ptr, positions of their entries)."""
    rng = np.random.default_rng(seed)
    rp = d["row_ptr"]
    rows = np.sort(rng.choice(len(rp) - 1, n_sample, replace=False))
    lens = (rp[rows + 1] - rp[rows]).astype(np.int64)
    sub = np.concatenate([[0], np.cumsum(lens)])
    idx = np.repeat(rp[rows], lens) + (np.arange(int(sub[-1])) - np.repeat(sub[:-1], lens))
    return rows, sub, idx


def _epoch(L, ffi, hm, hd, nb, eta, regs):
    st = ffi.Stats()
    ffi.check(L.fmhip_sgd_epoch(hm, hd, eta, *regs, None, C.byref(st)))
    return st.sse / max(st.rows, 1), st.nonfinite, st.rows, st.nnz


def _same_bits_after_one_step(fmhip, L, ffi, make_model, ds, batch, ids, eta, regs):
    """One step of `batch` from the same initial state, twice (two models): the parameters of `ids` agree bit for bit."""
    out = []
    for _ in range(2):
        fm = make_model()
        ffi.check(L.fmhip_sgd_step(fm.handle, ds.handle, batch, eta, *regs, None))
        fm._device_updated()
        out.append(fm.rows(ids))
        fm.close(discard=True)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


def test_c4_all_ten_million_rows(fmhip):
    from sparkfm_amd import _ffi as ffi, synth
    L = ffi.load()
    cfg = synth.CONFIGS["C4"]
    n1, k, br = cfg["features"], cfg["k"], 625_000
    d = synth.make_config("C4")                                  # all 10M rows
    assert len(d["y"]) == 10_000_000
    ds = fmhip.DataSet.from_arrays(d, name="C4", batch_rows=br).cache()
    assert ds.n_batches == 16 and ds.info()["nnz"] == int(d["row_ptr"][-1])
    w0, w, v = synth.init_params(cfg["seed"] + 1000, n1, k)
    w = np.random.default_rng(2).normal(0, 0.01, n1)

    def make_model():
        fm = fmhip.FMModel(n1 - 1, k)
        fm.w0, fm.w, fm.v = w0, w, v
        return fm

    fm = make_model()
    # sampled predictions against the oracle on exactly those rows (the model is small enough for the host: 256 MB)
    rows, sub, idx = _sample(d, 4000, 11)
    val = d["val"][idx].astype(np.float64)
    oy = oracle.predict(w0, w, v, sub, d["col"][idx], val)
    yh = fm.predict(ds)
    assert yh.shape == (10_000_000,) and np.isfinite(yh).all()
    scale = term_scale(dict(y=oy, row_ptr=sub, col=d["col"][idx], val=val, w0=w0, w=w, v=v))
    assert (np.abs(yh[rows] - oy) <= TOL_Y * scale).all()
    # two epochs: the loss falls, nothing non-finite, every row and nonzero is visited
    eta, regs = 0.02, (0.0, 1e-4, 1e-4)
    mse1, bad1, n_rows, nnz = _epoch(L, ffi, fm.handle, ds.handle, 16, eta, regs)
    mse2, bad2, _, _ = _epoch(L, ffi, fm.handle, ds.handle, 16, eta, regs)
    assert n_rows == 10_000_000 and nnz == int(d["row_ptr"][-1]) and bad1 == 0 and bad2 == 0
    assert mse2 < mse1 < float(np.mean(d["y"].astype(np.float64) ** 2)) * 1.01
    fm._device_updated()
    fm.close(discard=True)
    ids = np.unique(np.concatenate([np.arange(0, 4096), np.random.default_rng(5).integers(0, n1, 20_000)])).astype(np.int32)
    _same_bits_after_one_step(fmhip, L, ffi, make_model, ds, 7, ids, eta, regs)
    ds.unpersist()


def test_c5_two_to_the_24_rows_at_the_real_width(fmhip):
    from sparkfm_amd import _ffi as ffi, synth
    L = ffi.load()
    n1, k, br, n_rows = 1 << 25, 64, 250_000, 1 << 24
    d = synth.make_config("C5", rows=n_rows)                    # slots as hashed (the relabelled layout: bench.py's HBM-resident leg)
    nnz = int(d["row_ptr"][-1])
    ds = fmhip.DataSet.from_arrays(d, name="C5", batch_rows=br).cache()
    nb = ds.n_batches
    assert nb == 68 and ds.info()["nnz"] == nnz

    def make_model():
        fm = fmhip.FMModel(n1 - 1, k, seed=7, init_on_device=True)          # the reference's N(0, 0.01) init
        fm.handle
        return fm

    fm = make_model()
    # sampled predictions: the oracle on the compact problem over the features those rows touch
    rows, sub, idx = _sample(d, 3000, 3)
    feats, ccol = np.unique(d["col"][idx], return_inverse=True)
    w_t, v_t = fm.rows(feats)
    val = d["val"][idx].astype(np.float64)
    oy = oracle.predict(0.0, w_t, v_t, sub, ccol.astype(np.int32), val)
    yh = fm.predict(ds)
    assert yh.shape == (n_rows,) and np.isfinite(yh).all()
    scale = term_scale(dict(y=oy, row_ptr=sub, col=ccol.astype(np.int32), val=val, w0=0.0, w=w_t, v=v_t))
    assert (np.abs(yh[rows] - oy) <= TOL_Y * scale).all()
    del yh
    eta, regs = 0.02, (0.0, 1e-4, 1e-4)                          # weight decay on: the lazy rows-only update
    mse1, bad1, seen, seen_nnz = _epoch(L, ffi, fm.handle, ds.handle, nb, eta, regs)
    mse2, bad2, _, _ = _epoch(L, ffi, fm.handle, ds.handle, nb, eta, regs)
    assert seen == n_rows and seen_nnz == nnz and bad1 == 0 and bad2 == 0
    assert mse2 < mse1 < float(np.mean(d["y"].astype(np.float64) ** 2)) * 1.01
    fm._device_updated()
    fm.close(discard=True)
    ids = np.unique(np.concatenate([feats[:20_000], np.random.default_rng(5).integers(0, n1, 20_000)])).astype(np.int32)
    _same_bits_after_one_step(fmhip, L, ffi, make_model, ds, 33, ids, eta, regs)
    ds.unpersist()
