"""Data-parallel mini-batch SGD: rows sharded over ranks, one process per GPU.

Every global step each rank runs forward+backward on ITS mini-batch into a packed fp32
gradient buffer (a plain sum over rows), the buffers are summed across ranks with ONE
all-reduce (RCCL over xGMI on MI355X: torch.distributed backend "nccl"), and every rank
applies the identical update, so the parameter replicas stay bit-identical.  This replaces
the reference's driver-side reduce/collect (`RDD.reduce(_+_)`, `collectAsMap`;
S/fm/lib/ALS.scala:153,34,139) — SparkFM itself has no data-parallel update step.

The compute engine is pluggable so the orchestration can be exercised on CPU ranks (gloo)
in tests; the product engine is `HipEngine` (the C ABI).  There is no CPU engine in this
package.
"""
import ctypes as C

from . import _ffi
from .learn import FMLearn


class HipEngine:
    """Packed-gradient step engine over libfmhip (fmhip_step_compute / fmhip_step_apply)."""

    def __init__(self, fm, dataset):
        import torch
        self.torch = torch
        self.fm, self.dataset = fm, dataset
        self.L = _ffi.load()
        n = C.c_int64()
        _ffi.check(self.L.fmhip_grad_floats(fm.handle, C.byref(n)))
        self.grad = torch.zeros(int(n.value), dtype=torch.float32, device="cuda:%d" % fm.device)
        torch.cuda.synchronize(fm.device)   # the zero-fill must land before the library's stream uses the buffer
        _ffi.check(self.L.fmhip_grad_bind(fm.handle, C.c_void_p(self.grad.data_ptr())))
        self.n_batches = dataset.n_batches
        rf, gv = C.c_int64(), C.c_int64()
        _ffi.check(self.L.fmhip_grad_layout(fm.handle, C.byref(rf), C.byref(gv)))
        self.row_floats, self.gv_offset = int(rf.value), int(gv.value)
        self.n1 = fm.num_attribute + 1

    def forward(self, batch):
        _ffi.check(self.L.fmhip_step_forward(self.fm.handle, self.dataset.handle, batch))

    def backward(self, batch, lo, hi, finish):
        """Gradient rows of features lo <= id < hi (intervals in descending order, finish on the last)."""
        _ffi.check(self.L.fmhip_step_backward(self.fm.handle, self.dataset.handle, batch, lo, hi, 1 if finish else 0))

    def gv_slice(self, lo, hi, with_head=False):
        """The part of the packed buffer that holds G_V of features lo <= id < hi; with_head (lo must be
        0): preceded by the head (scalars | G_w | G_b), which lies right in front of feature 0's row."""
        end = self.gv_offset + min(hi, self.n1) * self.row_floats
        if with_head:
            assert lo == 0
            return self.grad[:end]
        return self.grad[self.gv_offset + lo * self.row_floats:end]

    def compute(self, batch):
        _ffi.check(self.L.fmhip_step_compute(self.fm.handle, self.dataset.handle, batch))

    def compute_empty(self):
        """This rank has no rows for the step: contribute a zero gradient (rows = 0)."""
        self.grad.zero_()

    def apply(self, eta, reg0, regw, regv):
        _ffi.check(self.L.fmhip_step_apply(self.fm.handle, eta, reg0, regw, regv))
        self.fm._device_updated()

    def stats(self):
        st = _ffi.Stats()
        _ffi.check(self.L.fmhip_step_stats(self.fm.handle, C.byref(st)))
        return st.as_dict()

    def close(self):
        _ffi.check(self.L.fmhip_grad_bind(self.fm.handle, None))


_streams = {}


def torch_stream_handle(device=0):
    """Makes a dedicated torch stream the CURRENT stream of `device` and returns its hipStream_t.

    Kernels the library launches on it then order naturally with torch.distributed collectives
    (ProcessGroupNCCL waits for, and is waited on by, the current stream) — no host syncs.  A
    dedicated stream is used because the default stream's handle is NULL, which the C ABI reads
    as "create your own stream"."""
    import torch
    if device not in _streams:
        _streams[device] = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(_streams[device])
    h = _streams[device].cuda_stream
    assert h != 0
    return C.c_void_p(h)


class DataParallelSGD(FMLearn):
    """FMLearn whose `learn` runs one data-parallel epoch over this rank's row shard."""

    def __init__(self, eta=0.05, reg0=0.0, regw=0.0, regv=0.0, group=None, engine_factory=HipEngine,
                 always_reduce=False, overlap=True, cuts=None):
        self.eta, self.reg0, self.regw, self.regv = float(eta), float(reg0), float(regw), float(regv)
        self.group = group
        self.engine_factory = engine_factory
        self.always_reduce = always_reduce   # run the collective even in a 1-rank group (self-test)
        # overlap: backward runs per feature interval (cold, high-id features first) and each
        # interval's slice of the gradient is all-reduced while the next interval computes
        self.overlap = overlap
        self.cuts = cuts                     # ascending feature ids [0, c1, ..., n+1]; None = planned from the data
        self._engine = None
        self._key = None

    def engine(self, fm, dataset):
        key = (id(fm), id(dataset))
        if self._engine is None or self._key != key:
            self._engine = self.engine_factory(fm, dataset)
            self._key = key
        return self._engine

    def global_steps(self, eng):
        """Every rank must take the same number of steps: max over ranks of local batches."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return eng.n_batches
        t = torch.tensor([eng.n_batches], dtype=torch.int64, device=eng.grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _collective(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.always_reduce)

    def plan_cuts(self, eng, fractions=(0.45,)):
        """Feature-id cut points shared by all ranks: the interval above the last cut holds about
        `fractions[-1]` of the stored nonzeros (rank 0's shard decides; ids are assumed to be roughly
        frequency-ranked — if they are not, the cuts are still valid, just less useful)."""
        import numpy as np
        import torch
        import torch.distributed as dist
        n1 = eng.n1
        cuts = torch.zeros(len(fractions), dtype=torch.int64, device=eng.grad.device)
        if not self._collective() or dist.get_rank(self.group) == 0:
            col = eng.dataset.col
            cnt = np.bincount(col, minlength=n1)[:n1] if len(col) else np.zeros(n1, np.int64)
            above = np.cumsum(cnt[::-1])[::-1]                      # nonzeros with id >= f
            total = max(int(above[0]) if n1 else 0, 1)
            vals = [int(np.searchsorted(-above, -f * total)) for f in sorted(fractions, reverse=True)]
            cuts = torch.tensor(vals, dtype=torch.int64, device=eng.grad.device)
        if self._collective():
            dist.broadcast(cuts, src=0, group=self.group)
        inner = sorted({int(c) for c in cuts.tolist() if 0 < int(c) < n1})
        self.cuts = [0] + inner + [n1]
        return self.cuts

    def step(self, eng, j):
        import torch.distributed as dist
        live = j < eng.n_batches
        if not (self._collective() and self.overlap and hasattr(eng, "backward")):
            if live:
                eng.compute(j)
            else:
                eng.compute_empty()
            if self._collective():
                dist.all_reduce(eng.grad, op=dist.ReduceOp.SUM, group=self.group)
            eng.apply(self.eta, self.reg0, self.regw, self.regv)
            return
        if self.cuts is None:
            self.plan_cuts(eng)
        works = []
        if live:
            eng.forward(j)
        else:
            eng.compute_empty()
        for i in range(len(self.cuts) - 1, 0, -1):                  # descending: cold features first
            lo, hi = self.cuts[i - 1], self.cuts[i]
            if live:
                eng.backward(j, lo, hi, finish=(i == 1))
            # the last interval (lowest ids) carries the head along: one collective fewer per step
            works.append(dist.all_reduce(eng.gv_slice(lo, hi, with_head=(i == 1)), op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True))
        for w in works:
            w.wait()
        eng.apply(self.eta, self.reg0, self.regw, self.regv)

    def learn(self, fm, dataset):
        eng = self.engine(fm, dataset)
        for j in range(self.global_steps(eng)):
            self.step(eng, j)
        return fm


def shard_rows(n_rows, rank, world):
    """Contiguous row shard [lo, hi) of `rank` (row-count balanced; the synthetic configs
    have i.i.d. row lengths so this is nnz-balanced to within ~0.1 %)."""
    lo = n_rows * rank // world
    hi = n_rows * (rank + 1) // world
    return lo, hi
