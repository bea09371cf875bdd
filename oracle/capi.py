"""ctypes binding of oracle/fm_oracle.c (TEST INFRASTRUCTURE ONLY — see fm_oracle.h).

All arrays are numpy; parameters follow the reference's layout: ``v`` has shape
``(k, n+1)`` in Fortran order (breeze column-major, S/fm/FMModel.scala:19), i.e.
flat element ``f + i*k``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfmoracle.so")
_lib = None

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src_newer = (not os.path.exists(_SO)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("fm_oracle.c", "fm_oracle.h")
    )
    if force or src_newer:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.fmo_max_threads.restype = C.c_int
    L.fmo_predict.argtypes = [C.c_int, C.c_double, _f64p, _f64p, C.c_int64, _i64p, _i32p, _f64p, _f64p, C.c_int]
    L.fmo_predict.restype = None
    L.fmo_rmse.argtypes = [C.c_int, C.c_double, _f64p, _f64p, C.c_int64, _i64p, _i32p, _f64p, _f64p, C.c_int]
    L.fmo_rmse.restype = C.c_double
    L.fmo_residual.argtypes = [C.c_int, C.c_double, _f64p, _f64p, C.c_int64, _i64p, _i32p, _f64p, _f64p, _f64p, C.c_int]
    L.fmo_residual.restype = None
    L.fmo_transpose.argtypes = [C.c_int64, C.c_int64, _i64p, _i32p, _f64p, _i64p, _i32p, _f64p]
    L.fmo_transpose.restype = None
    L.fmo_dimension.argtypes = [C.c_int64, _i64p, _i32p]
    L.fmo_dimension.restype = C.c_int32
    L.fmo_term_q.argtypes = [C.c_int, C.c_int, _f64p, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, _f64p]
    L.fmo_term_q.restype = None
    L.fmo_batch_grad.argtypes = [C.c_int, C.c_int64, C.c_double, _f64p, _f64p, C.c_int64, C.c_int64,
                                 _i64p, _i32p, _f64p, _f64p, _f64p, _f64p,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_int]
    L.fmo_batch_grad.restype = None
    L.fmo_sgd_step.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_double), _f64p, _f64p, C.c_int64, C.c_int64,
                               _i64p, _i32p, _f64p, _f64p, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_void_p, C.c_int]
    L.fmo_sgd_step.restype = C.c_double
    L.fmo_sgd_epoch.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_double), _f64p, _f64p, C.c_int64, C.c_int64,
                                C.c_void_p, _i64p, _i32p, _f64p, _f64p,
                                C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
    L.fmo_sgd_epoch.restype = C.c_double
    L.fmo_als_epoch.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_double), _f64p, _f64p,
                                C.c_double, C.c_double, C.c_double, C.c_int64,
                                _i64p, _i32p, _f64p, _f64p, _i64p, _i32p, _f64p, _f64p]
    L.fmo_als_epoch.restype = None
    _lib = L
    return L


def max_threads():
    return int(lib().fmo_max_threads())


def _csr(row_ptr, col, val):
    return (np.ascontiguousarray(row_ptr, np.int64), np.ascontiguousarray(col, np.int32),
            np.ascontiguousarray(val, np.float64))


def _flat_v(v):
    """(k, n1) array -> flat feature-major buffer (copy, element f + i*k)."""
    v = np.asarray(v, np.float64)
    return np.array(v.T, dtype=np.float64, order="C", copy=True).reshape(-1), v.shape[0], v.shape[1]


def predict(w0, w, v, row_ptr, col, val, threads=0):
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    out = np.empty(len(rp) - 1, np.float64)
    lib().fmo_predict(k, float(w0), np.ascontiguousarray(w, np.float64), vf, len(rp) - 1, rp, c, x, out, threads)
    return out


def rmse(w0, w, v, row_ptr, col, val, y, threads=0):
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    return float(lib().fmo_rmse(k, float(w0), np.ascontiguousarray(w, np.float64), vf, len(rp) - 1, rp, c, x,
                                np.ascontiguousarray(y, np.float64), threads))


def residual(w0, w, v, row_ptr, col, val, y, threads=0):
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    e = np.empty(len(rp) - 1, np.float64)
    lib().fmo_residual(k, float(w0), np.ascontiguousarray(w, np.float64), vf, len(rp) - 1, rp, c, x,
                       np.ascontiguousarray(y, np.float64), e, threads)
    return e


def transpose(n1, row_ptr, col, val):
    rp, c, x = _csr(row_ptr, col, val)
    nnz = int(rp[-1])
    col_ptr = np.empty(n1 + 1, np.int64)
    rows = np.empty(nnz, np.int32)
    cval = np.empty(nnz, np.float64)
    lib().fmo_transpose(len(rp) - 1, n1, rp, c, x, col_ptr, rows, cval)
    return col_ptr, rows, cval


def dimension(row_ptr, col):
    rp = np.ascontiguousarray(row_ptr, np.int64)
    return int(lib().fmo_dimension(len(rp) - 1, rp, np.ascontiguousarray(col, np.int32)))


def term_q(v, f, n_rows, col_ptr, rows, cval):
    vf, k, n1 = _flat_v(v)
    q = np.empty(n_rows, np.float64)
    lib().fmo_term_q(k, f, vf, n_rows, n1, np.ascontiguousarray(col_ptr, np.int64),
                     np.ascontiguousarray(rows, np.int32), np.ascontiguousarray(cval, np.float64), q)
    return q


def batch_grad(w0, w, v, r0, r1, row_ptr, col, val, y, threads=1):
    """-> (gv (k,n1), gw (n1,), gw0, sse, e (r1-r0,))"""
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    gv = np.empty(k * n1, np.float64)
    gw = np.empty(n1, np.float64)
    g0, sse = C.c_double(0), C.c_double(0)
    e = np.empty(r1 - r0, np.float64)
    lib().fmo_batch_grad(k, n1, float(w0), np.ascontiguousarray(w, np.float64), vf, r0, r1, rp, c, x,
                         np.ascontiguousarray(y, np.float64), gv, gw, C.byref(g0), C.byref(sse),
                         e.ctypes.data_as(C.c_void_p), threads)
    return gv.reshape(n1, k).T.copy(), gw, g0.value, sse.value, e


def sgd_step(w0, w, v, r0, r1, row_ptr, col, val, y, eta, reg0, regw, regv, threads=1):
    """Returns (w0', w', v', sse) — inputs are not modified."""
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    w = np.array(w, np.float64)
    w0c = C.c_double(float(w0))
    sse = lib().fmo_sgd_step(k, n1, C.byref(w0c), w, vf, r0, r1, rp, c, x, np.ascontiguousarray(y, np.float64),
                             eta, reg0, regw, regv, None, threads)
    return w0c.value, w, vf.reshape(n1, k).T.copy(), float(sse)


def sgd_epoch(w0, w, v, batch_rows, row_ptr, col, val, y, eta, reg0, regw, regv, order=None, threads=1):
    """Returns (w0', w', v', sum of per-batch sse) — inputs are not modified."""
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    w = np.array(w, np.float64)
    w0c = C.c_double(float(w0))
    op = None
    if order is not None:
        order = np.ascontiguousarray(order, np.int64)
        op = order.ctypes.data_as(C.c_void_p)
    sse = lib().fmo_sgd_epoch(k, n1, C.byref(w0c), w, vf, len(rp) - 1, batch_rows, op, rp, c, x,
                              np.ascontiguousarray(y, np.float64), eta, reg0, regw, regv, threads)
    return w0c.value, w, vf.reshape(n1, k).T.copy(), float(sse)


def als_epoch(w0, w, v, reg0, regw, regv, row_ptr, col, val, y):
    """One ALS.learn pass (S/fm/lib/ALS.scala:15-75).  Returns (w0', w', v')."""
    rp, c, x = _csr(row_ptr, col, val)
    vf, k, n1 = _flat_v(v)
    w = np.array(w, np.float64)
    w0c = C.c_double(float(w0))
    col_ptr, rows, cval = transpose(n1, rp, c, x)
    e = np.empty(max(len(rp) - 1, 1), np.float64)
    lib().fmo_als_epoch(k, n1 - 1, C.byref(w0c), w, vf, reg0, regw, regv, len(rp) - 1, rp, c, x,
                        np.ascontiguousarray(y, np.float64), col_ptr, rows, cval, e)
    return w0c.value, w, vf.reshape(n1, k).T.copy()
