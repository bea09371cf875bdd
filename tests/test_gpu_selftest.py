"""fmhip_comm_selftest: known patterns through every collective kind the plan and the step issue, with the step's own calls.

RCCL with more than one rank has never run on this repo's one-GPU boxes, so the first real execution of the library's
in-place reduce-scatter / all-gather offsets, element counts and grouped all-reduces is somebody's 8-GPU node: the self-test is
what that caller runs once after fmhip_comm_create (bench.py does, and falls back to torch.distributed if it fails).  Here: one
rank over real RCCL, eight thread-ranks over the host-staged transport, and transports that are WRONG in one kind each — the
verdict names that kind and nothing else.  (The reference's reduction is a JVM-side reduce, S/fm/lib/ALS.scala:153: it has no
such failure mode to test.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fmhip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sparkfm_amd
    return sparkfm_amd


def test_selftest_one_rank_over_rccl(fmhip):
    from sparkfm_amd.distributed import RcclComm
    fm = fmhip.FMModel(99, 8)
    comm = RcclComm(fm, 0, 1)
    assert comm.selftest().failed_kinds == 0
    comm.close()
    fm.close()


def test_selftest_eight_thread_ranks(fmhip):
    """All six kinds over 8 ranks: segment offsets r * count for r = 0..7, sums of 8 contributions, rank 0's broadcast."""
    from sparkfm_amd.distributed import ThreadStagedComm, run_thread_ranks

    def rank_main(rank, group):
        fm = fmhip.FMModel(99, 8)
        comm = ThreadStagedComm(fm, rank, group)
        comm.selftest()
        kinds = sorted({k for k, _ in comm.calls})
        counts = [c for k, c in comm.calls if k == 0]
        comm.close()
        fm.close()
        return comm.failed_kinds, kinds, counts

    out = run_thread_ranks(8, rank_main, timeout=120.0)
    assert [o[0] for o in out] == [0] * 8
    assert all(o[1] == [0, 1, 2, 3, 4, 5] for o in out)
    assert all(o[2] == out[0][2] and len(o[2]) == 3 for o in out)          # the three regions of a grouped all-reduce


@pytest.mark.parametrize("broken", [0, 1, 2, 3, 4, 5])
def test_selftest_names_the_kind_a_transport_gets_wrong(fmhip, broken):
    """A one-rank transport is the identity — except for ONE kind, which it answers wrongly (an element off by one, a segment
    that is not the caller's, a maximum that is not one): FMHIP_ERR_COMM and exactly that kind's bit."""
    from sparkfm_amd import _ffi
    L = _ffi.load()
    fm = fmhip.FMModel(99, 8)

    seen = []

    def collective(_ctx, dev, count, kind, stream):
        seen.append(kind)
        if kind != broken or seen.count(kind) > 1:          # (the verdict itself travels as a second int64 maximum: left alone)
            return 0
        width = 8 if kind in (_ffi.COLL_MAX_I64, _ffi.COLL_BCAST0_I64) else 4
        host = np.empty(count * width, np.uint8)
        _ffi.check(L.fmhip_device_read(_ffi.ptr(host), dev, host.nbytes, stream))
        if width == 8:
            host.view(np.int64)[count - 1] += 1
        elif kind == _ffi.COLL_ALLGATHER_I32:
            host.view(np.int32)[count // 2] += 1
        else:
            host.view(np.float32)[count - 1] += 1.0                              # the LAST element: a count short by one shows too
        _ffi.check(L.fmhip_device_write(dev, _ffi.ptr(host), host.nbytes, stream))
        return 0

    fn = _ffi.CollectiveFn(collective)
    h = C.c_void_p()
    _ffi.check(L.fmhip_comm_create_external(fm.handle, 0, 1, fn, None, C.byref(h)))
    mask = C.c_int(-1)
    rc = L.fmhip_comm_selftest(h, C.byref(mask))
    assert rc == -6, rc                                      # FMHIP_ERR_COMM
    assert mask.value == 1 << broken
    assert b"self-test" in L.fmhip_last_error()
    # a sound transport on the same model afterwards: clean
    fn2 = _ffi.CollectiveFn(lambda _c, _d, _n, _k, _s: 0)
    h2 = C.c_void_p()
    _ffi.check(L.fmhip_comm_create_external(fm.handle, 0, 1, fn2, None, C.byref(h2)))
    _ffi.check(L.fmhip_comm_selftest(h2, C.byref(mask)))
    assert mask.value == 0
    L.fmhip_comm_destroy(h)
    L.fmhip_comm_destroy(h2)
    fm.close()
