"""Model / FMModel — host mirror of S/Model.scala and S/fm/FMModel.scala.

``FMModel`` keeps the reference's public, mutable fields (w0, w, v, reg0, regw, regv) as
fp64 numpy arrays in the reference's layout — ``v`` is ``(num_factor, num_attribute+1)``
Fortran-ordered, i.e. breeze's column-major DenseMatrix (S/fm/FMModel.scala:19) — and a
device-resident fp32 twin behind the C ABI.  Host and device copies are synchronised lazily.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .dataset import DataSet


class Model:
    """S/Model.scala:9-32."""

    def predict(self, features):
        raise NotImplementedError

    def computeRMSE(self, dataset):
        raise NotImplementedError

    # The remaining metrics of S/Model.scala are off the training path and buggy in the reference
    # (quirk Q4): computeMAE there is a SIGNED mean error (no abs, :21-26) and computeAccuracy does
    # an integer division that yields 0 or 1 (:28-30).  Here they are what their names say; the
    # reference's signed quantity is available as computeMeanError.
    def computeMeanError(self, dataset):
        """mean of (target - prediction): what the reference's `computeMAE` actually returns."""
        return float((dataset.y - self.predict(dataset)).mean()) if dataset.size else 0.0

    def computeMAE(self, dataset):
        return float(abs(dataset.y - self.predict(dataset)).mean()) if dataset.size else 0.0

    def computeAccuracy(self, dataset):
        """fraction of rows whose prediction has the target's sign (>= 0 vs < 0), S/Model.scala:29."""
        if not dataset.size:
            return 0.0
        yh = self.predict(dataset)
        return float((((dataset.y >= 0) & (yh >= 0)) | ((dataset.y < 0) & (yh < 0))).mean())


class FMModel(Model):
    def __init__(self, num_attribute, num_factor, init_mean=0.0, init_stdev=0.01, seed=0, device=0, stream=None,
                 init_on_device=False):
        self.num_attribute = int(num_attribute)  # S/fm/FMModel.scala:10
        self.num_factor = int(num_factor)
        self.init_mean, self.init_stdev, self.seed = init_mean, init_stdev, seed
        n1 = self.num_attribute + 1
        # S/fm/FMModel.scala:17-22: w0 = 0, w = 0, v ~ N(mean, stdev).  The reference ignores `seed`
        # (quirk Q2: unseeded breeze Gaussian); here the seed IS honoured so runs are reproducible.
        # init_on_device: the draw happens on the GPU (fmhip_model_init_normal) and no host copy exists until
        # one is asked for — for models too wide to stage on the host (2^25 x 64 doubles = 17 GB)
        self._init_on_device = bool(init_on_device)
        self._w0 = 0.0
        self._w = np.zeros(n1)
        if self._init_on_device:
            self._v = None
        else:
            rng = np.random.Generator(np.random.PCG64(seed))
            self._v = np.asfortranarray(rng.normal(init_mean, init_stdev, size=(n1, self.num_factor)).T)
        self.k0 = True  # S/fm/FMModel.scala:25-26
        self.k1 = True
        self.reg0, self.regw, self.regv = 0.0, 0.0, 10.0  # S/fm/FMModel.scala:29-31 (ALS ridge terms)
        self.device = int(device)
        self._stream = stream
        self._h = None
        self._host_fresh = not self._init_on_device   # host arrays hold the current parameters
        self._dev_fresh = False                       # device holds the current parameters
        self._lost = False                            # close() dropped the only copy of the parameters (see close)

    # -- parameter access (lazy host<->device sync) --------------------------------------
    def _pull(self):
        if self._lost:
            raise RuntimeError("this FMModel's parameters were discarded by close(): the device copy was freed without being read back "
                               "(close(discard=True), or a model drawn on the device that nobody had read) — pull them before closing "
                               "(fm.v, fm.rows(ids)) or assign new ones (fm.w0, fm.w, fm.v = ...)")
        if not self._host_fresh:
            w0 = C.c_double()
            flat = np.empty(self.num_factor * (self.num_attribute + 1))
            _ffi.check(_ffi.load().fmhip_model_get_params(self.handle, C.byref(w0), _ffi.ptr(self._w), _ffi.ptr(flat)))
            self._w0 = w0.value
            self._v = flat.reshape((self.num_factor, self.num_attribute + 1), order="F")
            self._host_fresh = True

    @property
    def w0(self):
        self._pull()
        return self._w0

    def _assign(self):
        """Before a setter replaces one of the three parameters: the other two must be current (or, after a discarding
        close, restart from the state of a fresh model so that assigning all three brings the model back)."""
        if self._lost:
            self._lost = False
            self._host_fresh = True
            self._w0 = 0.0
            self._w = np.zeros(self.num_attribute + 1)
            self._v = np.zeros((self.num_factor, self.num_attribute + 1), order="F")
        self._pull()

    @w0.setter
    def w0(self, x):
        self._assign()
        self._w0 = float(x)
        self._dev_fresh = False

    @property
    def w(self):
        """Mutating the returned array in place requires a following ``touch()``."""
        self._pull()
        return self._w

    @w.setter
    def w(self, x):
        self._assign()
        self._w = np.array(x, np.float64).reshape(self.num_attribute + 1)
        self._dev_fresh = False

    @property
    def v(self):
        self._pull()
        return self._v

    @v.setter
    def v(self, x):
        self._assign()
        x = np.asarray(x, np.float64)
        if x.shape != (self.num_factor, self.num_attribute + 1):
            raise ValueError("v must have shape (num_factor, num_attribute + 1)")
        self._v = np.asfortranarray(x)
        self._dev_fresh = False

    def rows(self, ids):
        """(w[ids], v[:, ids]) straight from the device — `fm.w(i)`, `fm.v(::, i)` of the reference for a few features,
        without copying a model that may not fit the host (fmhip_model_get_rows)."""
        ids = np.ascontiguousarray(ids, np.int32)
        w = np.empty(len(ids))
        v = np.empty(len(ids) * self.num_factor)
        _ffi.check(_ffi.load().fmhip_model_get_rows(self.handle, len(ids), _ffi.ptr(ids), _ffi.ptr(w), _ffi.ptr(v)))
        return w, v.reshape((self.num_factor, len(ids)), order="F")

    def touch(self):
        """Declare that the host arrays were modified in place."""
        self._pull()
        self._dev_fresh = False

    @property
    def handle(self):
        """Device model with the current parameters uploaded."""
        L = _ffi.load()
        if self._lost:
            self._pull()       # raises: there is nothing to upload
        if self._h is None:
            h = C.c_void_p()
            _ffi.check(L.fmhip_model_create(self.device, self.num_attribute, self.num_factor, self._stream, C.byref(h)))
            self._h = h
            if self._init_on_device and not self._host_fresh:
                _ffi.check(L.fmhip_model_init_normal(h, self.seed, self.init_mean, self.init_stdev))
                self._dev_fresh = True
        if not self._dev_fresh:
            flat = np.ascontiguousarray(self._v.reshape(-1, order="F"))
            _ffi.check(L.fmhip_model_set_params(self._h, self._w0, _ffi.ptr(self._w), _ffi.ptr(flat)))
            self._dev_fresh = True
        return self._h

    def _device_updated(self):
        self._host_fresh = False
        self._dev_fresh = True

    def close(self, discard=False):
        """Frees the device model.  The parameters are copied back to the host first so that `fm.w` / `fm.v` keep
        working — except with `discard=True`, or for a model drawn on the device (`init_on_device=True`: it exists so
        that no host copy is made until one is asked for; at 2^25 x 64 the copy is 8.6 GB of fp32 staging plus a 17 GB
        fp64 array) whose parameters nobody has read yet.  After such a close the trained values are GONE and the model
        says so: `fm.w` / `fm.v` / `predict` raise until new parameters are assigned — it does not quietly hand back stale
        host arrays or re-draw the initial ones.  Pull them first (`fm.v`, `fm.rows(ids)`) if they are wanted."""
        if self._h is not None:
            if not discard and not (self._init_on_device and self._v is None):
                self._pull()
            elif not self._host_fresh:
                self._lost = True          # the device held the only current copy
            _ffi.load().fmhip_model_destroy(self._h)
            self._h = None
            self._dev_fresh = False

    def __del__(self):
        try:
            if self._h is not None:
                _ffi.load().fmhip_model_destroy(self._h)
        except Exception:
            pass

    # -- scoring -------------------------------------------------------------------------
    def predict(self, features):
        """FMModel.predict (S/fm/FMModel.scala:34-55).  `features` is one sparse row
        ``(indices, values)`` -> float, or a DataSet -> array (the rdd.mapValues(predict) of
        S/Model.scala:14)."""
        if isinstance(features, DataSet):
            out = np.empty(features.size)
            _ffi.check(_ffi.load().fmhip_predict(self.handle, features.handle, _ffi.ptr(out)))
            return out
        idx, val = features
        idx = np.ascontiguousarray(idx, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        if len(idx) != len(val):
            raise ValueError("index/value length mismatch")
        out = np.empty(1)
        _ffi.check(_ffi.load().fmhip_predict_rows(self.handle, 1, _ffi.ptr(np.array([0, len(idx)], np.int64)), _ffi.ptr(idx),
                                                  _ffi.ptr(val), _ffi.ptr(out)))
        return float(out[0])

    def computeRMSE(self, dataset):
        """Model.computeRMSE (S/Model.scala:13-19)."""
        r = C.c_double()
        _ffi.check(_ffi.load().fmhip_rmse(self.handle, dataset.handle, C.byref(r), None))
        return r.value

    def residual(self, dataset):
        """ALS.precomputeTermE (S/fm/lib/ALS.scala:142-144): e = predict - target."""
        out = np.empty(dataset.size)
        _ffi.check(_ffi.load().fmhip_residual(self.handle, dataset.handle, _ffi.ptr(out)))
        return out

    def termQ(self, dataset):
        """ALS.precomputeTermQ for all factors (S/fm/lib/ALS.scala:146-150): (rows, k)."""
        out = np.empty((dataset.size, self.num_factor))
        _ffi.check(_ffi.load().fmhip_term_q(self.handle, dataset.handle, _ffi.ptr(out)))
        return out

    def batchGradient(self, dataset, batch=0):
        """sum over the batch of e*h (h: S/fm/lib/ALS.scala:56-58,40,21) -> (gv (k,n1), gw, g0, stats)."""
        n1 = self.num_attribute + 1
        gv = np.empty(self.num_factor * n1)
        gw = np.empty(n1)
        g0 = C.c_double()
        st = _ffi.Stats()
        _ffi.check(_ffi.load().fmhip_batch_grad(self.handle, dataset.handle, batch, _ffi.ptr(gv), _ffi.ptr(gw),
                                                C.byref(g0), C.byref(st)))
        return gv.reshape((self.num_factor, n1), order="F"), gw, g0.value, st.as_dict()
