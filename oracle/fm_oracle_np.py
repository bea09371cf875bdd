"""numpy twin of oracle/fm_oracle.c — TEST INFRASTRUCTURE ONLY.

A second, independently written fp64 restatement (vectorised with scipy.sparse)
used to cross-check the C oracle on random data.  Same parameter layout as
``oracle.capi``: ``v`` is ``(k, n+1)``.
S/ = /root/reference/src/main/scala/io/edstud/spark/
"""
import numpy as np
import scipy.sparse as sp


def _X(n1, row_ptr, col, val):
    n_rows = len(row_ptr) - 1
    return sp.csr_matrix((np.asarray(val, np.float64), np.asarray(col, np.int32),
                          np.asarray(row_ptr, np.int64)), shape=(n_rows, n1))


def predict(w0, w, v, row_ptr, col, val):
    """S/fm/FMModel.scala:34-63, vectorised: yhat = w0 + Xw + 0.5*sum_f[(XV_f)^2 - X^2 V_f^2]."""
    v = np.asarray(v, np.float64)
    X = _X(v.shape[1], row_ptr, col, val)
    X2 = X.multiply(X).tocsr()
    q = X @ v.T                       # (rows, k)   == precomputeTermQ, S/fm/lib/ALS.scala:146-150
    s = X2 @ (v.T ** 2)
    return w0 + X @ np.asarray(w, np.float64) + 0.5 * (q * q - s).sum(axis=1)


def predict_pairwise(w0, w, v, row_ptr, col, val):
    """The naive definition yhat = w0 + sum w_i x_i + sum_{i<j} <v_i,v_j> x_i x_j (row loop)."""
    v = np.asarray(v, np.float64)
    out = np.empty(len(row_ptr) - 1)
    for r in range(len(out)):
        idx = np.asarray(col[row_ptr[r]:row_ptr[r + 1]])
        x = np.asarray(val[row_ptr[r]:row_ptr[r + 1]], np.float64)
        y = w0 + float(np.dot(np.asarray(w)[idx], x))
        G = (v[:, idx].T @ v[:, idx]) * np.outer(x, x)
        y += float(np.triu(G, 1).sum())
        out[r] = y
    return out


def rmse(w0, w, v, row_ptr, col, val, y):
    """S/Model.scala:13-19."""
    d = np.asarray(y, np.float64) - predict(w0, w, v, row_ptr, col, val)
    return float(np.sqrt((d * d).sum() / len(d)))


def batch_grad(w0, w, v, r0, r1, row_ptr, col, val, y):
    """sum_r e_r h_r(theta), h from S/fm/lib/ALS.scala:56-58,40,21.  -> gv (k,n1), gw, g0, sse, e"""
    v = np.asarray(v, np.float64)
    rp = np.asarray(row_ptr, np.int64)
    sub_ptr = rp[r0:r1 + 1] - rp[r0]
    c = np.asarray(col)[rp[r0]:rp[r1]]
    x = np.asarray(val, np.float64)[rp[r0]:rp[r1]]
    X = _X(v.shape[1], sub_ptr, c, x)
    X2 = X.multiply(X).tocsr()
    q = X @ v.T
    e = predict(w0, w, v, sub_ptr, c, x) - np.asarray(y, np.float64)[r0:r1]
    gv = (X.T @ (q * e[:, None])).T - v * (X2.T @ e)[None, :]
    gw = X.T @ e
    return gv, gw, float(e.sum()), float((e * e).sum()), e


def sgd_step(w0, w, v, r0, r1, row_ptr, col, val, y, eta, reg0, regw, regv):
    gv, gw, g0, sse, _ = batch_grad(w0, w, v, r0, r1, row_ptr, col, val, y)
    B = float(r1 - r0)
    w = np.asarray(w, np.float64)
    v = np.asarray(v, np.float64)
    return (w0 - eta * (g0 / B + reg0 * w0), w - eta * (gw / B + regw * w),
            v - eta * (gv / B + regv * v), sse)
