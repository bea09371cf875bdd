"""DataSet / Features — host mirror of S/DataSet.scala.

The reference wraps ``RDD[(Double, SparseVector[Double])]`` (S/DataSet.scala:42); here the
rows are held flattened as CSR (``row_ptr``, ``col`` = the SparseVector index arrays in
stored order, ``val``, ``y``) and ``cache()`` moves them — together with the per-batch
row->column transposes — into HBM through the C ABI (fmhip_dataset_create).
"""
import ctypes as C

import numpy as np

from . import _ffi


class Features:
    """S/DataSet.scala:9-40 — size, dimension, transpose of the inputs."""

    def __init__(self, row_ptr, col, val):
        self.row_ptr = np.ascontiguousarray(row_ptr, np.int64)
        self.col = np.ascontiguousarray(col, np.int32)
        self.val = np.ascontiguousarray(val)
        if self.val.dtype not in (np.float32, np.float64):
            self.val = self.val.astype(np.float64)
        if self.row_ptr.ndim != 1 or len(self.row_ptr) < 1 or self.row_ptr[0] != 0:
            raise ValueError("row_ptr must be 1-d, start at 0")
        if len(self.col) != self.row_ptr[-1] or len(self.val) != self.row_ptr[-1]:
            raise ValueError("col/val length must equal row_ptr[-1]")

    @property
    def isEmpty(self):  # S/DataSet.scala:19-21
        return self.size == 0

    @property
    def size(self):  # S/DataSet.scala:23-25 — rdd.count
        return len(self.row_ptr) - 1

    @property
    def dimension(self):  # S/DataSet.scala:27-29 — max feature index, 0 when empty
        return int(self.col.max()) if len(self.col) else 0

    @property
    def nnz(self):
        return int(self.row_ptr[-1])


class DataSet(Features):
    """S/DataSet.scala:42-62.  ``rows`` of the reference = (label, SparseVector) pairs."""

    def __init__(self, row_ptr, col, val, y, name="dataset", batch_rows=0, device=0, scoring=False, hot_block=None,
                 row_block_rows=None):
        super().__init__(row_ptr, col, val)
        self.y = np.ascontiguousarray(y, self.val.dtype)
        if len(self.y) != self.size:
            raise ValueError("y must have one label per row")
        self.name = name
        self.batch_rows = int(batch_rows)
        self.device = int(device)
        # scoring: rows + labels only on the device (fmhip_rows_create) — held-out data for predict /
        # computeRMSE (S/driver.scala:100-112); nothing a training step needs is built
        self.scoring = bool(scoring)
        # layout choices of THIS dataset (None = the library's defaults): dense hot block on/off, rows per row block
        self.hot_block, self.row_block_rows = hot_block, row_block_rows
        self._h = None

    # -- constructors -------------------------------------------------------------
    @classmethod
    def apply(cls, name, rows, **kw):
        """DataSet(name, rdd) (S/DataSet.scala:66-68); rows = iterable of (label, (indices, values))."""
        return cls.from_rows(rows, name=name, **kw)

    @classmethod
    def from_rows(cls, rows, name="dataset", **kw):
        ys, cols, vals, ptr = [], [], [], [0]
        for label, (idx, v) in rows:
            idx = np.asarray(idx, np.int32)
            v = np.asarray(v, np.float64)
            if len(idx) != len(v):
                raise ValueError("index/value length mismatch in a row")
            ys.append(float(label))
            cols.append(idx)
            vals.append(v)
            ptr.append(ptr[-1] + len(idx))
        col = np.concatenate(cols) if cols else np.zeros(0, np.int32)
        val = np.concatenate(vals) if vals else np.zeros(0, np.float64)
        return cls(np.asarray(ptr, np.int64), col, val, np.asarray(ys, np.float64), name=name, **kw)

    @classmethod
    def from_arrays(cls, d, **kw):
        return cls(d["row_ptr"], d["col"], d["val"], d["y"], **kw)

    # -- reference surface -----------------------------------------------------------
    @property
    def inputs(self):  # S/DataSet.scala:44
        return Features(self.row_ptr, self.col, self.val)

    @property
    def targets(self):  # S/DataSet.scala:46
        return self.y

    def rows(self):
        """Iterates (label, (indices, values)) like the reference's RDD."""
        for r in range(self.size):
            a, b = self.row_ptr[r], self.row_ptr[r + 1]
            yield float(self.y[r]), (self.col[a:b], self.val[a:b])

    def cache(self):
        """S/DataSet.scala:50-54: rdd.cache() — here: rows + per-batch transposes into HBM."""
        if self._h is None:
            L = _ffi.load()
            h = C.c_void_p()
            f32 = self.val.dtype == np.float32
            if self.scoring:
                fn = L.fmhip_rows_create_f32 if f32 else L.fmhip_rows_create
                _ffi.check(fn(self.device, self.size, _ffi.ptr(self.row_ptr), _ffi.ptr(self.col), _ffi.ptr(self.val),
                              _ffi.ptr(self.y), C.byref(h)))
            elif self.hot_block is not None or self.row_block_rows is not None:
                # hot_block: None = library default, False = off, True = on (all pages), n = on with up to n pages
                hb = -1 if self.hot_block is None else (4 if self.hot_block is True else int(self.hot_block))
                opts = _ffi.DatasetOpts(C.sizeof(_ffi.DatasetOpts), hb,
                                        self.batch_rows, -1 if self.row_block_rows is None else int(self.row_block_rows))
                val64, y64 = self.val.astype(np.float64, copy=False), self.y.astype(np.float64, copy=False)
                _ffi.check(L.fmhip_dataset_create_opts(self.device, self.size, _ffi.ptr(self.row_ptr), _ffi.ptr(self.col),
                                                       _ffi.ptr(val64), _ffi.ptr(y64), C.byref(opts), C.byref(h)))
            else:
                fn = L.fmhip_dataset_create_f32 if f32 else L.fmhip_dataset_create
                _ffi.check(fn(self.device, self.size, _ffi.ptr(self.row_ptr), _ffi.ptr(self.col), _ffi.ptr(self.val),
                              _ffi.ptr(self.y), self.batch_rows, C.byref(h)))
            self._h = h
        return self

    def unpersist(self):  # S/DataSet.scala:56-60
        if self._h is not None:
            _ffi.load().fmhip_dataset_destroy(self._h)
            self._h = None
        return self

    def __del__(self):
        try:
            self.unpersist()
        except Exception:
            pass

    @property
    def handle(self):
        return self.cache()._h

    # -- device-side views -------------------------------------------------------------
    def info(self):
        v = [C.c_int64() for _ in range(5)]
        _ffi.check(_ffi.load().fmhip_dataset_info(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("n_rows", "nnz", "dimension", "batch_rows", "n_batches"), (int(x.value) for x in v)))

    @property
    def n_batches(self):
        return self.info()["n_batches"]

    def batch_info(self, b):
        v = [C.c_int64() for _ in range(4)]
        _ffi.check(_ffi.load().fmhip_dataset_batch_info(self.handle, b, *[C.byref(x) for x in v]))
        return dict(zip(("row0", "rows", "nnz", "n_columns"), (int(x.value) for x in v)))

    def layout(self):
        """How the library laid the rows out: ids held in the dense hot block, nonzeros left in the sparse streams."""
        n, ids, sp = C.c_int32(), np.full(16, -1, np.int32), C.c_int64()
        _ffi.check(_ffi.load().fmhip_dataset_layout(self.handle, C.byref(n), _ffi.ptr(ids), C.byref(sp)))
        pages, n_all, all_ids, spb = C.c_int32(), C.c_int32(), np.full(128, -1, np.int32), C.c_int64()
        _ffi.check(_ffi.load().fmhip_dataset_hot_pages(self.handle, C.byref(pages), C.byref(n_all), _ffi.ptr(all_ids), C.byref(spb)))
        # hot_ids: the two-sided page; hot_ids_all: with the gradient-side pages; nnz_sparse(_backward): entries left in the
        # rows the forward walks / in the transposes the backward walks
        nr, planned, affine = C.c_int64(), C.c_int64(), C.c_int64()
        _ffi.check(_ffi.load().fmhip_dataset_band_plan(self.handle, C.byref(nr), C.byref(planned), C.byref(affine)))
        return dict(hot_ids=ids[:n.value].tolist(), nnz_sparse=int(sp.value), hot_pages=int(pages.value),
                    hot_ids_all=all_ids[:n_all.value].tolist(), nnz_sparse_backward=int(spb.value),
                    ranges=int(nr.value), planned_ranges=int(planned.value), band_affine_ranges=int(affine.value))

    def alsLevels(self):
        """The ALS sweep's level schedule (fmhip_dataset_als_levels): levels, columns, the widest level's columns."""
        v = [C.c_int64() for _ in range(3)]
        _ffi.check(_ffi.load().fmhip_dataset_als_levels(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("levels", "columns", "widest_level"), (int(x.value) for x in v)))

    def transposeInput(self, batch=0):
        """transposeInput (S/DataSet.scala:48, :31-38) of one mini-batch, read back from the GPU:
        (feat, ptr, rows, vals) — ascending present feature ids, offsets, batch-local row ids, values."""
        bi = self.batch_info(batch)
        feat = np.empty(bi["n_columns"], np.int32)
        ptr = np.empty(bi["n_columns"] + 1, np.int32)
        rows = np.empty(bi["nnz"], np.int32)
        vals = np.empty(bi["nnz"], np.float32)
        _ffi.check(_ffi.load().fmhip_dataset_get_transpose(self.handle, batch, _ffi.ptr(feat), _ffi.ptr(ptr),
                                                           _ffi.ptr(rows), _ffi.ptr(vals)))
        return feat, ptr, rows, vals
