"""FMLearn plug-in point and the HIP mini-batch SGD learner.

``FMLearn.learn(fm, dataset): FMModel`` is the reference's abstract learner
(S/fm/FMLearn.scala:10-12), invoked once per iteration by the fit loop
(S/fm/impl/FactorizationMachines.scala:45).  ``HipSGD`` is the MI355X learner behind it;
``HipSGD.run(...)`` mirrors ``ALS.run()`` (S/fm/lib/ALS.scala:202-208).
"""
import ctypes as C

import numpy as np

from . import _ffi


class FMLearn:
    def learn(self, fm, dataset):
        raise NotImplementedError


class HipSGD(FMLearn):
    """One ``learn`` call = one epoch of mini-batch SGD on the GPU (all batches of `dataset`).

    SparkFM has no SGD learner (its only learner is ALS); the update rule is this build's:
        theta <- theta - eta * (sum_{r in batch} e_r h_r(theta) / |batch| + reg * theta)
    with e, h from S/fm/lib/ALS.scala:142-144 and :56-58/:40/:21.  The regularisers are the
    learner's own (the model's regv = 10 default is an ALS ridge term — quirk Q5).
    """

    def __init__(self, eta=0.05, reg0=0.0, regw=0.0, regv=0.0, shuffle_seed=None):
        self.eta, self.reg0, self.regw, self.regv = float(eta), float(reg0), float(regw), float(regv)
        self.shuffle_seed = shuffle_seed
        self._epoch = 0
        self.last_stats = None

    @classmethod
    def run(cls, **kw):
        return cls(**kw)

    def batch_order(self, n_batches):
        """Visiting order of the (fixed, contiguous) mini-batches for the next epoch."""
        if self.shuffle_seed is None:
            return None
        rng = np.random.Generator(np.random.PCG64([self.shuffle_seed, self._epoch]))
        return rng.permutation(n_batches).astype(np.int64)

    def learn(self, fm, dataset):
        L = _ffi.load()
        order = self.batch_order(dataset.n_batches)
        st = _ffi.Stats()
        _ffi.check(L.fmhip_sgd_epoch(fm.handle, dataset.handle, self.eta, self.reg0, self.regw, self.regv,
                                     _ffi.ptr(order), C.byref(st)))
        fm._device_updated()
        self._epoch += 1
        self.last_stats = st.as_dict()
        return fm

    def step(self, fm, dataset, batch, want_stats=True):
        """A single mini-batch step (fmhip_sgd_step)."""
        st = _ffi.Stats()
        _ffi.check(_ffi.load().fmhip_sgd_step(fm.handle, dataset.handle, batch, self.eta, self.reg0, self.regw,
                                              self.regv, C.byref(st) if want_stats else None))
        fm._device_updated()
        return st.as_dict() if want_stats else None


class HipALS(FMLearn):
    """The reference's own learner on the GPU: one ``learn`` = one ``ALS.learn`` pass
    (S/fm/lib/ALS.scala:15-75) in fp64, using the MODEL's regularisers ``fm.reg0/regw/regv`` as the
    reference does (:21,:40,:56; defaults 0, 0, 10).  ``HipALS.run()`` mirrors ``ALS.run()`` (:202-208).
    Needs a single-batch DataSet (``batch_rows=0``)."""

    @classmethod
    def run(cls):
        return cls()

    def learn(self, fm, dataset):
        _ffi.check(_ffi.load().fmhip_als_epoch(fm.handle, dataset.handle, fm.reg0, fm.regw, fm.regv))
        fm._device_updated()
        return fm
