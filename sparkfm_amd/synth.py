"""Seeded synthetic sparse regression data — host-side helper.

Wraps csrc/synth.c (plain C + OpenMP).  Any row shard of a virtual dataset can be generated
independently of the others (one shard per GPU rank), deterministically.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libfmsynth.so")
_lib = None

BASE_SEED = 20261003

# name -> (rows, features, k, nnz_lo, nnz_hi, zipf_s)
CONFIGS = {
    "C1": dict(rows=10_000, features=1_000, k=8, nnz_lo=10, nnz_hi=10, zipf_s=0.0, seed=BASE_SEED + 1),
    "C2": dict(rows=1_000_000, features=100_000, k=16, nnz_lo=20, nnz_hi=60, zipf_s=1.05, seed=BASE_SEED + 2),
    "C3": dict(rows=1_000_000, features=100_000, k=32, nnz_lo=20, nnz_hi=60, zipf_s=1.05, seed=BASE_SEED + 3),
    "C4": dict(rows=10_000_000, features=1_000_000, k=32, nnz_lo=20, nnz_hi=60, zipf_s=1.05, seed=BASE_SEED + 4),
}


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            from . import _build
            _build.build_synth()
        L = C.CDLL(_SO)
        L.fms_zipf_count.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.fms_zipf_count.restype = None
        L.fms_zipf_fill.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_double,
                                    C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fms_zipf_fill.restype = C.c_int
        _lib = L
    return _lib


def make_zipf(seed, n_rows, n_features, nnz_lo, nnz_hi, zipf_s=1.05, k_true=4, noise=0.1, row_begin=0):
    """-> dict(row_ptr int64, col int32, val float32, y float32, n_features)."""
    L = _load()
    row_ptr = np.empty(n_rows + 1, np.int64)
    L.fms_zipf_count(seed, row_begin, n_rows, n_features, nnz_lo, nnz_hi, row_ptr.ctypes.data)
    nnz = int(row_ptr[-1])
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float32)
    y = np.empty(n_rows, np.float32)
    rc = L.fms_zipf_fill(seed, row_begin, n_rows, n_features, nnz_lo, nnz_hi, float(zipf_s), k_true, float(noise),
                         row_ptr.ctypes.data, col.ctypes.data, val.ctypes.data, y.ctypes.data)
    if rc != 0:
        raise MemoryError("synthetic generator ran out of memory")
    return dict(row_ptr=row_ptr, col=col, val=val, y=y, n_features=n_features)


def make_config(name, rows=None, row_begin=0):
    c = CONFIGS[name]
    d = make_zipf(c["seed"], rows if rows is not None else c["rows"], c["features"], c["nnz_lo"], c["nnz_hi"],
                  c["zipf_s"], row_begin=row_begin)
    d["k"] = c["k"]
    return d


def init_params(seed, n1, k, stdev=0.01):
    """w0 = 0, w = 0, V ~ N(0, stdev) — the reference's init in distribution
    (S/fm/FMModel.scala:12-13,17-22; its own draw is unseeded, quirk Q2).  v is (k, n1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return 0.0, np.zeros(n1), rng.normal(0.0, stdev, size=(n1, k)).T.copy()
