"""FeatureOrder — relabel feature ids by descending frequency before a dataset goes to the GPU.

A pure renaming of the features (the reference has no counterpart: breeze indexes `w(i)`, `v(f, i)` by whatever
ids the loader produced, S/fm/FMModel.scala:44-52).  The kernels do best when small ids are the frequent ones
(include/fmhip.h, "feature relabelling"); ids that arrive hashed or in dictionary order are relabelled once, the
model is trained in the internal numbering, and parameters are moved back with `to_caller`.

    order = FeatureOrder.fit(col, n1)                    # n1 = dimension + 1 slots of the model
    ds    = DataSet(row_ptr, order.relabel(col), val, y, ...).cache()
    fm    = FMModel(n1 - 1, k); order.set_params(fm, w0, w, v)     # caller's numbering in
    ...train...
    w0, w, v = order.get_params(fm)                                # caller's numbering out

Data-parallel jobs: every rank must hold the SAME order — sum `FeatureOrder.counts` over the ranks
(`fit(..., group=...)` does it over torch.distributed) before ranking.  Not for HipALS: its sweep runs in id order.
"""
import numpy as np

from . import _ffi


class FeatureOrder:
    """`device` (everywhere below): None = the library's host arithmetic; a GPU index = the same three steps on that GPU
    (fmhip_*_gpu: same results bit for bit, for hosts whose cores are the slow part)."""

    def __init__(self, rank, by_rank, device=None):
        self.rank = np.ascontiguousarray(rank, np.int32)          # caller's id -> internal id
        self.by_rank = np.ascontiguousarray(by_rank, np.int32)    # internal id -> caller's id
        self.device = device
        if self.rank.shape != self.by_rank.shape or self.rank.ndim != 1:
            raise ValueError("rank and by_rank must be 1-d and equally long")

    @property
    def n1(self):
        return len(self.rank)

    @staticmethod
    def counts(col, n1, into=None, device=None):
        """Stored nonzeros per feature id (int64[n1]); `into` accumulates over partitions."""
        col = np.ascontiguousarray(col, np.int32)
        out = np.zeros(n1, np.int64) if into is None else into
        if out.dtype != np.int64 or out.shape != (n1,) or not out.flags.c_contiguous:
            raise ValueError("`into` must be a contiguous int64[n1]")
        L = _ffi.load()
        if device is None:
            _ffi.check(L.fmhip_feature_counts(len(col), _ffi.ptr(col), n1, _ffi.ptr(out)))
        else:
            _ffi.check(L.fmhip_feature_counts_gpu(device, len(col), _ffi.ptr(col), n1, _ffi.ptr(out)))
        return out

    @classmethod
    def from_counts(cls, counts, device=None):
        counts = np.ascontiguousarray(counts, np.int64)
        rank = np.empty(len(counts), np.int32)
        by_rank = np.empty(len(counts), np.int32)
        L = _ffi.load()
        if device is None:
            _ffi.check(L.fmhip_rank_from_counts(len(counts), _ffi.ptr(counts), _ffi.ptr(rank), _ffi.ptr(by_rank)))
        else:
            _ffi.check(L.fmhip_rank_from_counts_gpu(device, len(counts), _ffi.ptr(counts), _ffi.ptr(rank), _ffi.ptr(by_rank)))
        return cls(rank, by_rank, device)

    @classmethod
    def fit(cls, col, n1, group=None, distributed=False, device=None):
        """Order of the ids in `col`; `distributed` (or a `group`): counts are summed over the ranks of
        torch.distributed first, so that all of them derive the same order."""
        cnt = cls.counts(col, n1, device=device)
        if distributed or group is not None:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(cnt)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=group)
            cnt = t.cpu().numpy()
        return cls.from_counts(cnt, device)

    @classmethod
    def identity(cls, n1):
        ids = np.arange(n1, dtype=np.int32)
        return cls(ids, ids.copy())

    def relabel(self, col, out=None):
        """rank[col] as int32 (ids outside [0, n1) raise)."""
        col = np.ascontiguousarray(col, np.int32)
        out = np.empty_like(col) if out is None else out
        if self.device is None:
            _ffi.check(_ffi.load().fmhip_relabel_columns(len(col), _ffi.ptr(col), self.n1, _ffi.ptr(self.rank), _ffi.ptr(out)))
        else:
            _ffi.check(_ffi.load().fmhip_relabel_columns_gpu(self.device, len(col), _ffi.ptr(col), self.n1, _ffi.ptr(self.rank), _ffi.ptr(out)))
        return out

    # -- parameters between the two numberings (w: [n1], v: [k, n1] as FMModel holds them) --------
    def to_internal(self, w, v):
        w, v = np.asarray(w), np.asarray(v)
        return w[self.by_rank], v[:, self.by_rank]

    def to_caller(self, w, v):
        w, v = np.asarray(w), np.asarray(v)
        return w[self.rank], v[:, self.rank]

    def set_params(self, fm, w0, w, v):
        wi, vi = self.to_internal(w, v)
        fm.w0, fm.w, fm.v = w0, np.ascontiguousarray(wi), np.ascontiguousarray(vi)

    def get_params(self, fm):
        w, v = self.to_caller(fm.w, fm.v)
        return fm.w0, np.ascontiguousarray(w), np.ascontiguousarray(v)
