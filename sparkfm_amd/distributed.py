"""Data-parallel mini-batch SGD: rows sharded over ranks, one process per GPU.

Every global step each rank runs forward+backward on ITS mini-batch into a packed fp32
gradient buffer (a plain sum over rows), the buffers are summed across ranks (RCCL over xGMI
on MI355X), and every rank applies the identical update, so the parameter replicas stay
bit-identical.  This replaces the reference's driver-side reduce/collect (`RDD.reduce(_+_)`,
`collectAsMap`; S/fm/lib/ALS.scala:153,34,139) — SparkFM itself has no data-parallel update step.

Two learners:

* ``HipDataParallelSGD`` — the product path: a thin caller of the C ABI's ``fmhip_dp_epoch``
  (sparkfm_amd/csrc/fmhip_comm.hip).  The library owns the RCCL communicator, the second
  stream and the overlap of the all-reduce with the feature-chunked backward; Python only ships
  the 128-byte unique id from rank 0 to the others (here over torch.distributed; a Spark driver
  would broadcast it).  The same two symbols are what a JVM ``HipSGD`` calls.
* ``DataParallelSGD`` — the same schedule orchestrated from Python over torch.distributed
  collectives, with a pluggable compute engine so that the sharding, step counts and
  zero-contributions can be exercised on CPU ranks (gloo) in tests; its GPU engine is
  ``HipEngine``.  There is no CPU engine in this package.
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
        # The collectives and grad.zero_() order only against torch's CURRENT stream: the library's
        # kernels must run on that very stream or the all-reduce can read a half-written gradient.
        cur = torch.cuda.current_stream(fm.device).cuda_stream
        mine = getattr(fm._stream, "value", fm._stream)
        if not mine or mine != cur:
            raise RuntimeError("HipEngine needs FMModel(..., stream=torch_stream_handle(device)) and that stream to be "
                               "torch's current stream (model stream %r, current %r)" % (mine, cur))
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
        """Hands the model back its own gradient buffer (the bound torch tensor may then be freed)."""
        if self.grad is not None and self.fm._h is not None:
            self.torch.cuda.synchronize(self.fm.device)
            _ffi.check(self.L.fmhip_grad_bind(self.fm._h, None))
        self.grad = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
            if self._engine is not None and hasattr(self._engine, "close"):
                self._engine.close()     # the old model must not keep pointing at a tensor about to be freed
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


class RcclComm:
    """The library's RCCL communicator of one rank (fmhip_comm_create).  `unique_id`: the 128 bytes
    rank 0 got from `RcclComm.unique_id()`; None = fetch/ship them over torch.distributed."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * _ffi.UNIQUE_ID_BYTES)()
        _ffi.check(_ffi.load().fmhip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, fm, rank, world, unique_id=None, group=None):
        L = _ffi.load()
        if unique_id is None:
            import torch.distributed as dist
            box = [RcclComm.unique_id() if rank == 0 else None]
            if world > 1:
                # `src` is a GLOBAL rank: group rank 0 of a sub-group need not be global rank 0
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast_object_list(box, src=src, group=group)
            unique_id = box[0]
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        _ffi.check(L.fmhip_comm_create(fm.handle, unique_id, rank, world, C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def selftest(self):
        """Collective: known patterns through every collective kind the plan and the step issue (fmhip_comm_selftest).
        Raises on EVERY rank alike if any rank saw wrong elements; `.failed_kinds` then holds the FMHIP_COLL_* bit mask."""
        mask = C.c_int(0)
        rc = _ffi.load().fmhip_comm_selftest(self._h, C.byref(mask))
        self.failed_kinds = mask.value
        _ffi.check(rc)
        return self

    def close(self):
        if self._h:
            _ffi.load().fmhip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostStagedComm(RcclComm):
    """The library's data-parallel step over a transport of the caller's own (fmhip_comm_create_external): here every
    collective is staged through the host and summed by torch.distributed on CPU tensors (gloo).  No overlap with the
    backward and PCIe-bound — it exists for nodes without RCCL and for the two-ranks-on-one-GPU test, which RCCL refuses
    (`ncclCommInitRank: invalid usage`); the schedule, the cuts, the global row count and the update are the library's
    own, exactly as over RCCL."""

    def __init__(self, fm, rank, world, group=None):   # noqa: D107 — does not call RcclComm.__init__ (no unique id)
        import numpy as np
        import torch
        import torch.distributed as dist
        L = _ffi.load()
        self.rank, self.world = rank, world
        self.calls = []                        # (kind, count) of every collective: what the ranks must agree on

        def collective(_ctx, dev, count, kind, stream):
            try:
                self.calls.append((kind, count))
                if kind in (_ffi.COLL_ALLGATHER_I32, _ffi.COLL_ALLGATHER_F32):
                    # world x count elements on the device, this rank's own at rank * count
                    dt, tdt = (np.int32, torch.int32) if kind == _ffi.COLL_ALLGATHER_I32 else (np.float32, torch.float32)
                    host = np.empty(world * count, dt)
                    _ffi.check(L.fmhip_device_read(_ffi.ptr(host), dev, host.nbytes, stream))
                    parts = [torch.empty(count, dtype=tdt) for _ in range(world)]
                    dist.all_gather(parts, torch.from_numpy(host[rank * count:(rank + 1) * count].copy()), group=group)
                    host = np.concatenate([p.numpy() for p in parts])
                    _ffi.check(L.fmhip_device_write(dev, _ffi.ptr(host), host.nbytes, stream))
                    return 0
                if kind == _ffi.COLL_REDUCE_SCATTER_F32:
                    # world x count floats; this rank keeps the sum of ITS segment, the others' are left as they were
                    host = np.empty(world * count, np.float32)
                    _ffi.check(L.fmhip_device_read(_ffi.ptr(host), dev, host.nbytes, stream))
                    t = torch.from_numpy(host)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    mine = np.ascontiguousarray(host[rank * count:(rank + 1) * count])
                    _ffi.check(L.fmhip_device_write(C.c_void_p(dev + rank * count * 4), _ffi.ptr(mine), mine.nbytes, stream))
                    return 0
                dt = np.float32 if kind == _ffi.COLL_SUM_F32 else np.int64
                host = np.empty(count, dt)
                _ffi.check(L.fmhip_device_read(_ffi.ptr(host), dev, host.nbytes, stream))
                t = torch.from_numpy(host)
                if kind == _ffi.COLL_SUM_F32:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                elif kind == _ffi.COLL_MAX_I64:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                else:
                    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                _ffi.check(L.fmhip_device_write(dev, _ffi.ptr(host), host.nbytes, stream))
                return 0
            except Exception:   # noqa: BLE001 — an exception must not cross the C ABI
                import traceback
                traceback.print_exc()
                return 1

        self._fn = _ffi.CollectiveFn(collective)     # kept alive as long as the communicator
        self._h = C.c_void_p()
        _ffi.check(L.fmhip_comm_create_external(fm.handle, rank, world, self._fn, None, C.byref(self._h)))


class ThreadGroup:
    """What N ranks that are THREADS of one process share (ThreadStagedComm): a barrier, one slot per rank and the
    buffer a reduced result is assembled in — plus the little control plane a caller of thread-ranks needs beside the
    library's own collectives (barrier, all-reduce of a few host numbers)."""

    def __init__(self, world, timeout=600.0):
        import threading
        self.world = int(world)
        self.timeout = timeout
        self._barrier = threading.Barrier(self.world)
        self.slots = [None] * self.world
        self.out = None

    def barrier(self):
        self._barrier.wait(self.timeout)

    def abort(self):
        self._barrier.abort()

    def exchange(self, rank, value):
        """-> every rank's `value`, in rank order (a host-side all-gather of Python objects)."""
        self.slots[rank] = value
        self.barrier()
        got = list(self.slots)
        self.barrier()
        return got

    def allreduce(self, rank, values, op="max"):
        """Element-wise max / sum over the ranks of a short list of numbers."""
        got = self.exchange(rank, [float(v) for v in values])
        f = max if op == "max" else sum
        return [f(col) for col in zip(*got)]


class ThreadStagedComm(RcclComm):
    """The library's data-parallel step with every rank a THREAD of this process (fmhip_comm_create_external): each
    collective is staged through the host and reduced between the threads — segment r of a sum by rank r, every element in
    rank order, so all replicas receive the same bits.  This is the `local[*]` shape of the reference (executor tasks are
    threads of one JVM, S/driver.scala:14); with one GPU per thread a real deployment takes RcclComm — this transport
    exists to run a world of 8 on a box with ONE GPU, where RCCL refuses a second rank per device and the test pool admits
    at most 6 processes on the card.  Not a measurement of anything: every byte crosses PCIe twice."""

    def __init__(self, fm, rank, group):   # noqa: D107 — does not call RcclComm.__init__ (no unique id)
        import numpy as np
        L = _ffi.load()
        world = group.world
        self.rank, self.world, self.group = rank, world, group
        self.calls = []                        # (kind, count) of every collective: what the ranks must agree on

        def read(dev, n, dt, stream):
            host = np.empty(n, dt)
            _ffi.check(L.fmhip_device_read(_ffi.ptr(host), dev, host.nbytes, stream))
            return host

        def collective(_ctx, dev, count, kind, stream):
            try:
                self.calls.append((kind, count))
                g = group
                if kind in (_ffi.COLL_ALLGATHER_I32, _ffi.COLL_ALLGATHER_F32):
                    dt = np.int32 if kind == _ffi.COLL_ALLGATHER_I32 else np.float32
                    g.slots[rank] = read(dev + rank * count * 4, count, dt, stream)
                    g.barrier()
                    host = np.concatenate(g.slots)
                    g.barrier()
                    _ffi.check(L.fmhip_device_write(dev, _ffi.ptr(host), host.nbytes, stream))
                    return 0
                if kind == _ffi.COLL_REDUCE_SCATTER_F32:
                    g.slots[rank] = read(dev, world * count, np.float32, stream)
                    g.barrier()
                    mine = g.slots[0][rank * count:(rank + 1) * count].copy()
                    for r in range(1, world):
                        mine += g.slots[r][rank * count:(rank + 1) * count]
                    g.barrier()
                    _ffi.check(L.fmhip_device_write(dev + rank * count * 4, _ffi.ptr(mine), mine.nbytes, stream))
                    return 0
                if kind == _ffi.COLL_SUM_F32:
                    g.slots[rank] = read(dev, count, np.float32, stream)
                    if rank == 0:
                        g.out = np.empty(count, np.float32)
                    g.barrier()
                    lo, hi = count * rank // world, count * (rank + 1) // world      # this rank sums its segment, in rank order
                    seg = g.out[lo:hi]
                    np.copyto(seg, g.slots[0][lo:hi])
                    for r in range(1, world):
                        seg += g.slots[r][lo:hi]
                    g.barrier()
                    _ffi.check(L.fmhip_device_write(dev, _ffi.ptr(g.out), g.out.nbytes, stream))
                    g.barrier()                                                    # nobody reuses `out` before everyone has copied it
                    return 0
                g.slots[rank] = read(dev, count, np.int64, stream)
                g.barrier()
                host = np.maximum.reduce(g.slots) if kind == _ffi.COLL_MAX_I64 else g.slots[0].copy()
                g.barrier()
                _ffi.check(L.fmhip_device_write(dev, _ffi.ptr(host), host.nbytes, stream))
                return 0
            except Exception:   # noqa: BLE001 — an exception must not cross the C ABI
                import traceback
                traceback.print_exc()
                group.abort()                      # the peers' barriers break instead of waiting for this rank
                return 1

        self._fn = _ffi.CollectiveFn(collective)     # kept alive as long as the communicator
        self._h = C.c_void_p()
        _ffi.check(L.fmhip_comm_create_external(fm.handle, rank, world, self._fn, None, C.byref(self._h)))


def run_thread_ranks(world, fn, timeout=600.0):
    """Runs fn(rank, group) on `world` threads of this process (one rank each, a shared ThreadGroup) and returns their
    results in rank order; the first exception of any rank is raised after all threads have ended."""
    import threading
    group = ThreadGroup(world, timeout)
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r, group)
        except BaseException as ex:   # noqa: BLE001
            err[r] = ex
            group.abort()

    threads = [threading.Thread(target=body, args=(r,), name="rank%d" % r) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for ex in err:
        if ex is not None and not isinstance(ex, __import__("threading").BrokenBarrierError):
            raise ex
    for ex in err:
        if ex is not None:
            raise ex
    return out


class HipDataParallelSGD(FMLearn):
    """FMLearn whose `learn` runs one data-parallel epoch over this rank's row shard INSIDE the library:
    forward -> feature-chunked backward overlapped with the RCCL all-reduce -> identical update
    (fmhip_dp_epoch).  `upper_fractions`: ascending shares of the stored nonzeros at or above each cut of
    the backward — (0.05, 0.15, 0.3, 0.55) = five intervals, the first 5 % of the work and over half of the bytes
    (the fastest of the candidates timed against an emulated 8-GPU all-reduce at C4's width);
    () = no overlap: whole backward, one all-reduce."""

    def __init__(self, comm, eta=0.05, reg0=0.0, regw=0.0, regv=0.0, upper_fractions=(0.05, 0.15, 0.3, 0.55), exchange="dense"):
        self.comm = comm
        self.eta, self.reg0, self.regw, self.regv = float(eta), float(reg0), float(regw), float(regv)
        self.upper_fractions = tuple(float(f) for f in upper_fractions)
        # "dense": the whole packed gradient all-reduced in overlapped slices, every rank updates every row; "sharded": the
        # slices reduce-scattered, every rank updates its 1/world share, the updated rows all-gathered; "touched": only the
        # rows some rank's batch touched (fmhip_dp_exchange) — for models far wider than a global batch; "pipelined": the dense
        # exchange with consecutive steps overlapped (the coldest slice travels beside the next position's forward; `learn` and
        # `steps_at` know the next position, a single `step` / `step_at` is the same step without the overlap)
        self.exchange = exchange
        _ffi.check(_ffi.load().fmhip_dp_exchange(comm.handle, _ffi.EXCHANGE_MODES[exchange]))
        self.cuts = None
        self._planned_for = None
        self.last_stats = None

    def set_exchange(self, exchange):
        """Switches what a step exchanges (every rank alike, then `plan` again)."""
        _ffi.check(_ffi.load().fmhip_dp_exchange(self.comm.handle, _ffi.EXCHANGE_MODES[exchange]))
        self.exchange = exchange
        self._planned_for = None

    def plan(self, fm, dataset):
        """Collective: rank 0's data pick the cuts, every rank receives them."""
        import numpy as np
        fr = np.ascontiguousarray(sorted(self.upper_fractions), np.float64)
        cuts = np.zeros(max(len(fr), 1), np.int64)
        _ffi.check(_ffi.load().fmhip_dp_plan(fm.handle, dataset.handle, self.comm.handle, len(fr), _ffi.ptr(fr), _ffi.ptr(cuts)))
        self.cuts = [int(c) for c in cuts[:len(fr)] if c > 0]
        self._planned_for = id(dataset)
        return self.cuts

    def exchange_info(self):
        mode, cap, mean = C.c_int(), C.c_int64(), C.c_double()
        _ffi.check(_ffi.load().fmhip_dp_exchange_info(self.comm.handle, C.byref(mode), C.byref(cap), C.byref(mean)))
        return dict(mode={v: k for k, v in _ffi.EXCHANGE_MODES.items()}[mode.value], id_slots_per_rank=int(cap.value),
                    mean_union_rows=float(mean.value))

    def plan_steps(self):
        """The lock-step steps of an epoch the plan agreed on (the largest batch count of any rank)."""
        n = C.c_int64()
        _ffi.check(_ffi.load().fmhip_dp_plan_info(self.comm.handle, C.byref(n), None))
        return int(n.value)

    def step(self, fm, dataset, batch):
        """One global step, the next of the schedule; batch < 0: this rank contributes zeros."""
        if self._planned_for != id(dataset):
            self.plan(fm, dataset)
        _ffi.check(_ffi.load().fmhip_dp_step(fm.handle, dataset.handle, batch, self.comm.handle, self.eta, self.reg0,
                                             self.regw, self.regv))
        fm._device_updated()

    def step_at(self, fm, dataset, position):
        """One global step at a position of the lock-step schedule that EVERY rank names alike (this rank's batch
        `position`, or zeros if it has fewer): the call of a permuted epoch."""
        if self._planned_for != id(dataset):
            self.plan(fm, dataset)
        _ffi.check(_ffi.load().fmhip_dp_step_at(fm.handle, dataset.handle, position, self.comm.handle, self.eta, self.reg0,
                                                self.regw, self.regv))
        fm._device_updated()

    def steps_at(self, fm, dataset, positions):
        """Several global steps in ONE call, at the named positions of the lock-step schedule (every rank the same list;
        fmhip_dp_steps): what the pipelined exchange needs to overlap each step's last slice with the next position's forward —
        in the other modes the same as step_at per position."""
        import numpy as np
        if self._planned_for != id(dataset):
            self.plan(fm, dataset)
        pos = np.ascontiguousarray(positions, np.int64)
        _ffi.check(_ffi.load().fmhip_dp_steps(fm.handle, dataset.handle, _ffi.ptr(pos), len(pos), self.comm.handle, self.eta, self.reg0,
                                              self.regw, self.regv))
        fm._device_updated()

    def learn(self, fm, dataset, order=None):
        """One data-parallel epoch; `order`: a permutation of range(plan_steps()), the same on every rank
        (e.g. numpy's default_rng(shuffle_seed + epoch).permutation(steps)); None = ascending."""
        import numpy as np
        if self._planned_for != id(dataset):
            self.plan(fm, dataset)
        st = _ffi.Stats()
        L = _ffi.load()
        if order is None:
            _ffi.check(L.fmhip_dp_epoch(fm.handle, dataset.handle, self.comm.handle, self.eta, self.reg0, self.regw, self.regv, C.byref(st)))
        else:
            o = np.ascontiguousarray(order, np.int64)
            _ffi.check(L.fmhip_dp_epoch_order(fm.handle, dataset.handle, self.comm.handle, self.eta, self.reg0, self.regw, self.regv,
                                              _ffi.ptr(o), len(o), C.byref(st)))
        fm._device_updated()
        self.last_stats = st.as_dict()
        return fm


def shard_rows(rows, rank, world):
    """Contiguous row shard [lo, hi) of `rank`.  `rows` = the CSR row_ptr array: balanced by stored
    nonzeros (fmhip_shard_rows) — the right split for skewed real data; `rows` = a row
    count: balanced by rows."""
    import numpy as np
    if np.ndim(rows) == 0:
        n_rows = int(rows)
        return n_rows * rank // world, n_rows * (rank + 1) // world
    rp = np.ascontiguousarray(rows, np.int64)
    lo, hi = C.c_int64(), C.c_int64()
    _ffi.check(_ffi.load().fmhip_shard_rows(len(rp) - 1, _ffi.ptr(rp), world, rank, C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)
