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
    # Criteo-shaped: 39 fields (13 numeric + 26 categorical, each missing w.p. 0.1) hashed into 2^25 slots
    "C5": dict(rows=1 << 26, features=1 << 25, k=64, nnz_lo=0, nnz_hi=39, zipf_s=1.2, seed=BASE_SEED + 5, criteo=True),
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
        L.fms_set_threads.argtypes = [C.c_int]
        L.fms_set_threads.restype = None
        L.fms_criteo_count.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        L.fms_criteo_count.restype = None
        L.fms_criteo_fill.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fms_criteo_fill.restype = C.c_int
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP threads of the generator (launchers such as torch.distributed.run export OMP_NUM_THREADS=1)."""
    _load().fms_set_threads(int(n))


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


def make_criteo(seed, n_rows, n_hash=1 << 25, k_true=4, noise=0.1, row_begin=0):
    """Criteo-shaped rows (BASELINE config 5): 39 fields — 13 numeric (one slot each, value
    log1p-like in [0, 8)) + 26 categorical (value 1.0, per-field Zipf(1.1..1.3) popularity over the
    Criteo vocabulary sizes) — each missing w.p. 0.1, hashed into `n_hash` slots."""
    L = _load()
    row_ptr = np.empty(n_rows + 1, np.int64)
    L.fms_criteo_count(seed, row_begin, n_rows, n_hash, row_ptr.ctypes.data)
    nnz = int(row_ptr[-1])
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float32)
    y = np.empty(n_rows, np.float32)
    rc = L.fms_criteo_fill(seed, row_begin, n_rows, n_hash, k_true, float(noise), row_ptr.ctypes.data,
                           col.ctypes.data, val.ctypes.data, y.ctypes.data)
    if rc != 0:
        raise MemoryError("synthetic generator ran out of memory")
    return dict(row_ptr=row_ptr, col=col, val=val, y=y, n_features=n_hash)


def make_config(name, rows=None, row_begin=0, features=None):
    c = CONFIGS[name]
    if c.get("criteo"):
        d = make_criteo(c["seed"], rows if rows is not None else c["rows"], features or c["features"], row_begin=row_begin)
        d["k"] = c["k"]
        return d
    d = make_zipf(c["seed"], rows if rows is not None else c["rows"], c["features"], c["nnz_lo"], c["nnz_hi"],
                  c["zipf_s"], row_begin=row_begin)
    d["k"] = c["k"]
    return d


def init_params(seed, n1, k, stdev=0.01):
    """w0 = 0, w = 0, V ~ N(0, stdev) — the reference's init in distribution
    (S/fm/FMModel.scala:12-13,17-22; its own draw is unseeded, quirk Q2).  v is (k, n1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return 0.0, np.zeros(n1), rng.normal(0.0, stdev, size=(n1, k)).T.copy()
