"""The data-parallel step at the world size BASELINE's target is quoted on — EIGHT ranks — on the one GPU of the test box, and
the threading rule of the C ABI.

RCCL refuses a second rank on a device and the test pool admits at most 6 processes on the card, so the eight ranks are
THREADS of one worker process (sparkfm_amd.distributed.ThreadStagedComm over fmhip_comm_create_external: every collective
staged through the host, segment r of a sum reduced by rank r in rank order) — which is also the shape of the reference's own
deployment, executor tasks as threads of one JVM under `local[*]` (S/driver.scala:14).  Everything but the transport is the
RCCL path: fmhip_dp_plan's agreement over 8 ranks, the global row count, equal shares of n+1 rows over 8 with the slack rows
behind the tables, interval edges at multiples of 8, cap x 8 id slots and the union of 8 ranks' touched rows, a rank without
rows, ranks that run out of batches at different steps.  Checked against the fp64 oracle over the equivalent global batches
(S/fm/lib/ALS.scala:153: the reduction the learner owns)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

ROWS8 = [1700, 700, 0, 1000, 300, 1000, 150, 1000]          # uneven shards; rank 2 holds no rows at all


def case8(**kw):
    cfg = dict(seed=4242, rows=ROWS8, n1_data=800, n1=803, k=32, lo=4, hi=24, batch_rows=500, exchange="dense", fractions=[], epochs=2,
               eta=0.05, regw=1e-3, regv=1e-3, shuffle_seed=None)
    cfg.update(kw)
    return cfg


def run_case(cfg, tmp_path, timeout=600):
    path = os.path.join(str(tmp_path), "case.json")
    with open(path, "w") as fh:
        json.dump(cfg, fh)
    r = subprocess.run([sys.executable, os.path.join(HERE, "dp_case_worker.py"), "threads", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


@pytest.mark.parametrize("exchange", ["dense", "sharded"])
@pytest.mark.parametrize("fractions", [[], [0.3], [0.05, 0.15, 0.3, 0.55]])
def test_eight_ranks_dense_and_sharded(tmp_path, exchange, fractions):
    """803 feature rows over 8 ranks (n+1 is no multiple of 8: the top share reaches into the slack rows), 0 / 1 / 4 cuts, both
    dense modes; replicas bit-identical, same collectives on all 8 ranks, the oracle matched (the worker asserts all three)."""
    s = run_case(case8(exchange=exchange, fractions=fractions), tmp_path)
    assert s["world"] == 8 and s["steps"] == 4 and s["rows"] == 200          # the last global batch: rank 0's 200 rows, seven ranks contribute zeros
    assert s["rel_err_v"] <= 1e-5 and s["rel_err_w"] <= 1e-5
    calls = np.array(s["calls"], np.int64).reshape(-1, 2)
    n_int = len([c for c in s["cuts"] if c > 0]) + 1
    assert len(s["cuts"]) == len(fractions)
    steps = 2 * 4
    if exchange == "dense":
        sums = calls[calls[:, 0] == 0]
        assert len(sums) == steps * (1 + (1 if n_int == 1 else 3 * n_int))
    else:
        assert all(c % 8 == 0 for c in s["cuts"])                               # equal shares: interval edges at multiples of the world
        for kind in (4, 5):
            seg = calls[calls[:, 0] == kind][:, 1]
            assert len(seg) == steps * n_int
            # per step the shares of one rank add up to ceil(803 / 8) = 101 rows of 32 floats: 808 > 803 rows — the slack rows
            assert seg.reshape(steps, -1).sum(axis=1).tolist() == [101 * 32] * steps


@pytest.mark.parametrize("fractions,shuffle,stepwise,k,reverse_ids", [([0.3], None, False, 32, False), ([0.05, 0.15, 0.3, 0.55], 11, False, 32, False),
                                                                   ([0.2, 0.5], None, True, 16, False), ([0.3], 5, False, 64, False), ([], None, False, 32, False),
                                                                   ([0.1, 0.4], None, False, 32, True), ([0.3], 7, False, 16, True)])
def test_eight_ranks_pipelined(tmp_path, fractions, shuffle, stepwise, k, reverse_ids):
    """FMHIP_EXCHANGE_PIPELINED with 8 ranks: intervals from feature 0 up, every step's top slice exchanged beside the next
    position's pass-A forward (fmhip_dp_epoch / _epoch_order), a rank without rows, ranks that run out of batches at different
    steps, a permuted order, step by step through fmhip_dp_step_at (the same step without the overlap), no cut at all (the dense
    step) — replicas bit-identical, the same collectives on all 8 ranks, the fp64 oracle matched (S/fm/lib/ALS.scala:153)."""
    # reverse_ids: the frequent features carry the highest ids, so the dense hot block's features lie at or above the cuts and
    # the two-pass forward must score them in pass B — the oracle over the global batches is still matched
    s = run_case(case8(exchange="pipelined", fractions=fractions, shuffle_seed=shuffle, stepwise=stepwise, k=k, reverse_ids=reverse_ids), tmp_path)
    assert s["world"] == 8 and s["steps"] == 4
    assert s["rel_err_v"] <= 1e-5 and s["rel_err_w"] <= 1e-5
    calls = np.array(s["calls"], np.int64).reshape(-1, 2)
    n_int = len([c for c in s["cuts"] if c > 0]) + 1
    sums = calls[calls[:, 0] == 0]
    assert len(sums) == 2 * 4 * (1 + (1 if n_int == 1 else 3 * n_int))        # per step: |B|, then three regions per interval


@pytest.mark.parametrize("n1", [803, 50_003])
def test_eight_ranks_touched_rows(tmp_path, n1):
    """The touched-rows exchange with 8 ranks: cap x 8 id slots per planned step, the union of eight batches' rows, one compact
    all-reduce per step; a 50,003-row model of which the data touch 800 (the untouched rows decay as in the oracle's dense
    update)."""
    s = run_case(case8(exchange="touched", n1=n1), tmp_path)
    assert s["world"] == 8 and s["steps"] == 4 and s["rows"] == 200
    calls = np.array(s["calls"], np.int64).reshape(-1, 2)
    assert (calls[:, 0] == 3).sum() == 4                                        # the plan: one id all-gather per step of the schedule
    info = s["info"]
    assert info["mode"] == "touched" and 0 < info["mean_union_rows"] <= 802 and info["id_slots_per_rank"] <= 802 + 64
    packed = calls[(calls[:, 0] == 0) & (calls[:, 1] > 1)][:, 1]
    assert len(packed) == 8 and packed.max() <= 1664 + 804 * 32


@pytest.mark.parametrize("exchange", ["dense", "sharded", "touched"])
def test_eight_ranks_permuted_batch_order(tmp_path, exchange):
    """`HipSGD.shuffle_seed` data-parallel: every rank takes the same seeded permutation of the batch positions through
    fmhip_dp_step (a rank without that batch passes -1) — the touched-rows plan is per batch position, so it holds too."""
    s = run_case(case8(exchange=exchange, fractions=[0.3], shuffle_seed=11, n1=2003), tmp_path)
    assert s["world"] == 8 and s["rel_err_v"] <= 1e-5


def test_four_threads_score_through_one_frozen_model():
    """include/fmhip.h's threading rule: the scoring calls hold the model's lock shared and work in a context of their own
    (stream + workspace), so executor threads may score through ONE frozen model at once (S/Model.scala:14 under local[*]).
    4 threads x 6 rounds of predict / rmse / residual / predict_rows over disjoint row sets == the serial results, bit for bit."""
    import sparkfm_amd
    from sparkfm_amd import synth
    d = synth.make_zipf(5, 40_000, 3000, 4, 40, zipf_s=1.05)
    w0, w, v = synth.init_params(3, 3000, 32, stdev=0.05)
    w = np.random.default_rng(4).normal(0, 0.05, 3000)
    fm = sparkfm_amd.FMModel(2999, 32)
    fm.w0, fm.w, fm.v = 0.1, w, v
    _ = fm.handle                                                  # uploaded before any thread starts: the model is frozen from here on
    parts = []
    for t in range(4):
        lo, hi = t * 10_000, (t + 1) * 10_000
        a, b = int(d["row_ptr"][lo]), int(d["row_ptr"][hi])
        sub = dict(row_ptr=d["row_ptr"][lo:hi + 1] - a, col=d["col"][a:b], val=d["val"][a:b], y=d["y"][lo:hi])
        parts.append((sub, sparkfm_amd.DataSet.from_arrays(sub, batch_rows=3000).cache()))

    def score(t):
        sub, ds = parts[t]
        idx, val = sub["col"][:int(sub["row_ptr"][1])], sub["val"][:int(sub["row_ptr"][1])].astype(np.float64)
        return fm.predict(ds), fm.computeRMSE(ds), fm.residual(ds), fm.predict((idx, val))

    serial = [score(t) for t in range(4)]
    assert abs(serial[0][0][0] - serial[0][3]) <= 1e-6 * (1 + abs(serial[0][3]))
    got, errs = [[] for _ in range(4)], []

    def body(t):
        try:
            for _ in range(6):
                got[t].append(score(t))
        except BaseException as ex:   # noqa: BLE001
            errs.append(ex)

    threads = [threading.Thread(target=body, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for t in range(4):
        for yh, rmse, e, one in got[t]:
            np.testing.assert_array_equal(yh, serial[t][0])
            assert rmse == serial[t][1] and one == serial[t][3]
            np.testing.assert_array_equal(e, serial[t][2])
    # ... and against the oracle, so that "equal to the serial result" is not two equal mistakes
    sub = parts[2][0]
    oy = oracle.predict(0.1, w, v, sub["row_ptr"], sub["col"], sub["val"].astype(np.float64))
    assert np.abs(serial[2][0] - oy).max() <= 1e-4
    for _, ds in parts:
        ds.unpersist()
    fm.close()


def test_two_threads_train_two_models_on_one_gpu():
    """Two host threads, each training ITS model on ITS dataset on the same GPU at the same time (handles are independent;
    the process-wide tuning keys are atomic) == the two runs one after the other, bit for bit."""
    import sparkfm_amd
    from sparkfm_amd import synth
    jobs = []
    for t, (k, n1) in enumerate([(32, 900), (16, 5000)]):
        d = synth.make_zipf(60 + t, 20_000, n1, 4, 30, zipf_s=1.05)
        w0, w, v = synth.init_params(8 + t, n1, k, stdev=0.05)
        jobs.append((d, k, n1, w, v))

    def train(t):
        d, k, n1, w, v = jobs[t]
        ds = sparkfm_amd.DataSet.from_arrays(d, batch_rows=4000).cache()
        fm = sparkfm_amd.FMModel(n1 - 1, k)
        fm.w0, fm.w, fm.v = 0.0, w, v
        sgd = sparkfm_amd.HipSGD(eta=0.05, regw=1e-3, regv=1e-3)
        for _ in range(3):
            fm = sgd.learn(fm, ds)
        out = (fm.w0, fm.w.copy(), fm.v.copy(), fm.computeRMSE(ds))
        ds.unpersist()
        fm.close()
        return out

    serial = [train(0), train(1)]
    got, errs = [None, None], []

    def body(t):
        try:
            got[t] = train(t)
        except BaseException as ex:   # noqa: BLE001
            errs.append(ex)

    threads = [threading.Thread(target=body, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for t in range(2):
        assert got[t][0] == serial[t][0] and got[t][3] == serial[t][3]
        np.testing.assert_array_equal(got[t][1], serial[t][1])
        np.testing.assert_array_equal(got[t][2], serial[t][2])


def test_bench_with_eight_thread_ranks():
    """`python bench.py --gpus 8 --transport threads`: the bench's whole N = 8 flow on the one GPU of the test box — eight ranks
    as threads of one process, the library's communicators over the thread transport, plan agreement, cut AND mode tuning over
    8 ranks, the timed region with its barriers and max over ranks, the exchange profile, the legs without exchange, the C3
    twin, rank 0's counter passes (rocprofv3 child processes over tools/pmc_leg.py c4) and CPU baseline while the other ranks
    wait, the JSON assembly.  Timings mean nothing here (every byte crosses PCIe twice); the flow, the record's completeness
    and its wall time are what is tested."""
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--transport", "threads", "--steps", "4", "--warmup", "2",
           "--rows", "100000", "--batch-rows", "50000", "--cpu-budget", "3", "--tune-budget", "12"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    # the record is re-written after every leg: every line is a complete JSON object, the LAST one is the record
    lines = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.strip()]
    stages = [ln["record"]["stage"] for ln in lines]
    # N > 1: a first headline with the default plan BEFORE the cut / mode sweep (a sweep that hangs on an unseen node must not cost
    # the record), then the headline of the plan the sweep chose; from there on the headline never changes
    assert stages[0].startswith("headline (default plan") and "headline" in stages[1:] and lines[-1]["record"]["final"] is True, stages
    assert "cut_tuning" not in lines[0]["exchange"] and lines[0]["value"] > 0 and lines[0]["exchange"]["mode"] == "dense"
    after = lines[stages.index("headline"):]
    assert all(ln["value"] == after[0]["value"] and ln["roofline"] is not None for ln in after)
    out = lines[-1]
    assert not out["legs"]["skipped"], out["legs"]
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["steps"] == 4 and out["warmup"] == 2
    assert out["config"]["global_batch"] == 8 * out["config"]["batch_rows_per_gpu"]        # one job: the global batch is what is fixed
    assert out["metric"] == "nnz_per_sec_fm_sgd_training" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["config"]["workload"].startswith("C4") and out["config"]["rows_per_gpu"] == 100000
    x = out["exchange"]
    assert x["nranks"] == 8 and x["transport"] == "threads" and x["mode"] in ("dense", "sharded", "pipelined")
    # the communicator passed its self-test on every rank, and after the timed steps the replicas hold the same bits
    assert "passed" in x["selftest"] and x["replicas"]["identical"] is True and x["replicas"]["rows_compared"] > 1000
    assert {t["exchange"] for t in x["cut_tuning"]} == {"dense", "sharded", "pipelined"} and all(t["ms_per_step"] > 0 for t in x["cut_tuning"])
    assert any(len(t["cuts"]) >= 2 and all(c % 8 == 0 for c in t["cuts"]) for t in x["cut_tuning"] if t["exchange"] == "sharded")
    assert x["exposed_comm_ms"] >= 0 and x["comm_busy_ms"] > 0
    assert x["c3_on_every_gpu"]["value"] > 0 and x["per_gpu_without_exchange"]["value"] > 0
    assert x["c4_one_gpu"]["value"] > 0 and x["scaling_vs_c4_one_gpu"] > 0
    # the N > 1 line is a complete record: roofline from counters (measured in this run, or the committed C4 pass) + cpu_baseline
    rf = out["roofline"]
    assert rf["basis"] == "counters" and rf["traffic"] > 0 and 0 < rf["frac"] < 1.5 and rf["kernel"] in ("k_forward", "k_backward")
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["kind"] == "port"
    assert out["train"]["nonfinite"] == 0 and out["sustained"]["steps"] > 0
    assert wall < 600, wall


def test_a_rank_that_brings_the_wrong_model_keeps_the_step_alive():
    """A failure only ONE rank can see must not leave its peers inside a collective: in the touched-rows mode a rank
    that calls the step with a model of another width than the plan's contributes zeros — sized by the PLAN's width, its own
    update skipped — and gets the error afterwards; the other rank's step completes and equals the oracle over its rows alone."""
    import ctypes as C
    import dp_cases
    from sparkfm_amd import DataSet, FMModel, _ffi
    from sparkfm_amd.distributed import HipDataParallelSGD, ThreadStagedComm, run_thread_ranks
    L = _ffi.load()
    cfg = case8(rows=[900, 700], exchange="touched", n1=2003, batch_rows=900, epochs=1)
    w0, w, v = dp_cases.init(cfg)

    def body(r, g):
        ds = DataSet.from_arrays(dp_cases.shard(cfg, r), batch_rows=900, device=0).cache()
        fm = FMModel(cfg["n1"] - 1, 32, device=0)
        fm.w0, fm.w, fm.v = w0, w, v
        comm = ThreadStagedComm(fm, r, g)
        dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, exchange="touched", upper_fractions=(0.3,))
        dp.plan(fm, ds)
        if r == 1:
            other = FMModel(cfg["n1"] - 1, 64, device=0)         # rows of 64 floats: not what the plan was made for
            rc = L.fmhip_dp_step_at(other.handle, ds.handle, 0, comm.handle, 0.05, 0.0, 1e-3, 1e-3)
            msg = L.fmhip_last_error().decode()
            other.close(discard=True)
            out = (rc, msg)
        else:
            dp.step_at(fm, ds, 0)
            out = (fm.w0, fm.w.copy(), fm.v.copy())
        g.barrier()
        # ... and the communicator is still usable: a regular step of both ranks afterwards
        dp.step_at(fm, ds, 0)
        res = (out, fm.v.copy())
        g.barrier()
        comm.close()
        ds.unpersist()
        fm.close(discard=True)
        return res

    (r0, v0_after), (r1, v1_after) = run_thread_ranks(2, body, timeout=120.0)
    assert r1[0] == _ffi.load().fmhip_version() * 0 - 1 and "planned for rows of 32 floats" in r1[1] and "contributed zeros" in r1[1]
    d0 = dp_cases.shard(cfg, 0)
    o0, ow, ov, _ = oracle.sgd_step(w0, w, v, 0, 900, d0["row_ptr"], d0["col"], d0["val"].astype(np.float64), d0["y"].astype(np.float64),
                                    0.05, 0.0, 1e-3, 1e-3)
    assert np.linalg.norm(r0[2] - ov) <= 1e-5 * np.linalg.norm(ov) and np.linalg.norm(r0[1] - ow) <= 1e-5 * np.linalg.norm(ow)
    assert np.isfinite(v0_after).all() and np.isfinite(v1_after).all()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["dense", "sharded", "touched"])
def test_a_transport_that_fails_in_mid_step_is_an_error_not_a_crash(exchange):
    """fmhip_comm_create_external: the caller's collective may fail (a peer died, a socket closed).  Whichever collective of a
    step it is — the row count, a slice of the gradient, an all-gather — the step returns FMHIP_ERR_COMM with the callback's code
    in the message, nothing crashes or hangs, the handles can still be destroyed, and a NEW communicator on the same model and
    dataset plans and steps again (the model's values after a failed step are unspecified: the caller restores them)."""
    import ctypes as C
    from helpers import random_problem
    from sparkfm_amd import DataSet, FMModel, _ffi
    L = _ffi.load()
    a = random_problem(31, 1500, 2003, 32, 4, 30)
    ds = DataSet(a["row_ptr"], a["col"], a["val"], a["y"], batch_rows=500, device=0).cache()
    mode = {"dense": 0, "touched": 1, "sharded": 2}[exchange]
    state = {"calls": 0, "fail_at": -1}

    def collective(_ctx, _dev, _count, _kind, _stream):      # a world of one: every collective is the identity
        state["calls"] += 1
        return 7 if state["calls"] == state["fail_at"] else 0

    fn = _ffi.CollectiveFn(collective)

    def fresh():
        fm = FMModel(a["n1"] - 1, 32, device=0)
        fm.w0, fm.w, fm.v = a["w0"], a["w"], a["v"]
        h = C.c_void_p()
        _ffi.check(L.fmhip_comm_create_external(fm.handle, 0, 1, fn, None, C.byref(h)))
        _ffi.check(L.fmhip_dp_exchange(h, mode))
        fr = (C.c_double * 2)(0.12, 0.4)
        state["calls"], state["fail_at"] = 0, -1
        _ffi.check(L.fmhip_dp_plan(fm.handle, ds.handle, h, 2, fr, None))
        return fm, h

    # how many collectives does a clean step make?  Then fail each of them in turn.
    fm, h = fresh()
    before = state["calls"]
    _ffi.check(L.fmhip_dp_step_at(fm.handle, ds.handle, 0, h, 0.05, 0.0, 1e-3, 1e-3))
    _ffi.check(L.fmhip_synchronize(fm.handle))
    per_step = state["calls"] - before
    assert per_step >= 2
    fm._device_updated()
    want_v = fm.v.copy()
    assert np.isfinite(want_v).all() and not np.array_equal(want_v, a["v"])
    _ffi.check(L.fmhip_comm_destroy(h))
    fm.close(discard=True)
    for k in range(1, per_step + 1):
        fm, h = fresh()
        state["fail_at"] = state["calls"] + k
        rc = L.fmhip_dp_step_at(fm.handle, ds.handle, 0, h, 0.05, 0.0, 1e-3, 1e-3)
        msg = L.fmhip_last_error().decode()
        assert rc == -6 and "returned 7" in msg, (k, rc, msg)                   # FMHIP_ERR_COMM, the callback's own code
        _ffi.check(L.fmhip_synchronize(fm.handle))
        _ffi.check(L.fmhip_comm_destroy(h))
        fm.close(discard=True)
    # ... and the model / dataset are still good for a new communicator: the same step, the same result
    fm, h = fresh()
    _ffi.check(L.fmhip_dp_step_at(fm.handle, ds.handle, 0, h, 0.05, 0.0, 1e-3, 1e-3))
    fm._device_updated()
    np.testing.assert_array_equal(fm.v, want_v)
    _ffi.check(L.fmhip_comm_destroy(h))
    fm.close(discard=True)
    ds.unpersist()


@pytest.mark.parametrize("fault", ["different_orders", "not_a_permutation"])
def test_ranks_that_disagree_about_the_epoch_order_fail_together(fault):
    """fmhip_dp_epoch_order: the order array is "the same on every rank by contract" — the mistake the check
    exists to catch is a rank whose copy differs.  The order's validity and a hash of its content travel in the epoch's opening
    max-reduce, so a bad or different array on ONE rank makes EVERY rank return FMHIP_ERR_INVALID before the first step's
    collective; nobody is left waiting in one, and the same communicator then runs a regular epoch."""
    import dp_cases
    from sparkfm_amd import DataSet, FMModel, _ffi
    from sparkfm_amd.distributed import HipDataParallelSGD, ThreadStagedComm, run_thread_ranks
    L = _ffi.load()
    cfg = case8(rows=[1500, 1500, 1000], exchange="dense", batch_rows=500, epochs=1)
    w0, w, v = dp_cases.init(cfg)

    def body(r, g):
        ds = DataSet.from_arrays(dp_cases.shard(cfg, r), batch_rows=500, device=0).cache()
        fm = FMModel(cfg["n1"] - 1, 32, device=0)
        fm.w0, fm.w, fm.v = w0, w, v
        comm = ThreadStagedComm(fm, r, g)
        dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, exchange="dense", upper_fractions=(0.3,))
        dp.plan(fm, ds)
        order = np.array([2, 0, 1], np.int64)
        if r == 1:
            order = np.array([1, 0, 2], np.int64) if fault == "different_orders" else np.array([2, 2, 1], np.int64)
        rc = L.fmhip_dp_epoch_order(fm.handle, ds.handle, comm.handle, 0.05, 0.0, 1e-3, 1e-3, _ffi.ptr(order), 3, None)
        msg = L.fmhip_last_error().decode()
        n_calls = len(getattr(comm, "calls", []))
        g.barrier()
        dp.learn(fm, ds, order=[2, 0, 1])                 # the communicator is as good as before
        out = (rc, msg, n_calls, fm.v.copy())
        g.barrier()
        comm.close()
        ds.unpersist()
        fm.close(discard=True)
        return out

    res = run_thread_ranks(3, body, timeout=120.0)
    for r, (rc, msg, _, _) in enumerate(res):
        assert rc == -1, (r, rc, msg)                      # FMHIP_ERR_INVALID on every rank
        assert ("DIFFERENT orders" in msg) if fault == "different_orders" else ("permutation" in msg), (r, msg)
    assert res[0][2] == res[1][2] == res[2][2]             # ... after the same collectives (the plan's and the epoch's agreement, no step's)
    np.testing.assert_array_equal(res[0][3], res[1][3])
    np.testing.assert_array_equal(res[0][3], res[2][3])


@pytest.mark.parametrize("exchange", ["dense", "sharded"])
def test_a_run_of_steps_survives_a_batch_only_one_rank_rejects(exchange):
    """fmhip_dp_steps in the non-pipelined modes: a position whose batch fails a check only ONE rank can see
    (here: a dataset the plan has not seen, with a larger batch) must not end that rank's run — its peers would be alone in
    the next position's collectives.  The rank contributes zeros to every such position, takes every step, and gets the error
    afterwards; the other rank's run completes and equals the oracle over its own rows."""
    import dp_cases
    from sparkfm_amd import DataSet, FMModel, _ffi
    from sparkfm_amd.distributed import HipDataParallelSGD, ThreadStagedComm, run_thread_ranks
    L = _ffi.load()
    cfg = case8(rows=[1000, 1000], exchange=exchange, batch_rows=500, epochs=1, n1=808)
    w0, w, v = dp_cases.init(cfg)
    positions = np.array([0, 1, 0], np.int64)

    def body(r, g):
        ds = DataSet.from_arrays(dp_cases.shard(cfg, r), batch_rows=500, device=0).cache()
        fm = FMModel(cfg["n1"] - 1, 32, device=0)
        fm.w0, fm.w, fm.v = w0, w, v
        comm = ThreadStagedComm(fm, r, g)
        dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, exchange=exchange, upper_fractions=(0.3,))
        dp.plan(fm, ds)
        run_ds = ds
        if r == 1:      # the same rows cut into ONE batch of 1000: larger than anything the plan agreed on
            run_ds = DataSet.from_arrays(dp_cases.shard(cfg, r), batch_rows=1000, device=0).cache()
        rc = L.fmhip_dp_steps(fm.handle, run_ds.handle, _ffi.ptr(positions), 3, comm.handle, 0.05, 0.0, 1e-3, 1e-3)
        msg = L.fmhip_last_error().decode()
        fm._device_updated()
        out = (rc, msg, fm.w0, fm.w.copy(), fm.v.copy())
        g.barrier()
        comm.close()
        if run_ds is not ds:
            run_ds.unpersist()
        ds.unpersist()
        fm.close(discard=True)
        return out

    r0, r1 = run_thread_ranks(2, body, timeout=120.0)
    assert r0[0] == 0, r0[:2]
    assert r1[0] == -1 and "contributed zeros" in r1[1] and "fmhip_dp_plan" in r1[1], r1[:2]
    # both replicas took the same three steps over rank 0's rows alone (rank 1's position 1 does not exist in its one-batch
    # dataset and position 0 was refused: zeros every time) and hold the same bits
    np.testing.assert_array_equal(r0[4], r1[4])
    d0 = dp_cases.shard(cfg, 0)
    ow0, ow, ov = w0, w.copy(), v.copy()
    for p in positions:
        ow0, ow, ov, _ = oracle.sgd_step(ow0, ow, ov, int(p) * 500, int(p) * 500 + 500, d0["row_ptr"], d0["col"], d0["val"].astype(np.float64),
                                         d0["y"].astype(np.float64), 0.05, 0.0, 1e-3, 1e-3)
    assert np.linalg.norm(r0[4] - ov) <= 1e-5 * np.linalg.norm(ov) and np.linalg.norm(r0[3] - ow) <= 1e-5 * np.linalg.norm(ow)


def test_two_models_on_one_dataset_under_the_pipelined_exchange():
    """the pipelined exchange partitions every row's entries at its plan's top cut.  The partition lives in
    a COPY of the stream (the dataset's own streams never move, so another thread scoring with the dataset is not disturbed),
    is made under the dataset's own lock, and is held while a run walks it: a second model whose plan has ANOTHER top cut gets
    an error that says so (and contributes zeros: no peer hangs) instead of re-partitioning under the first run's launches.
    Here, one rank per model over real one-rank communicators: model A runs pipelined epochs on the shared dataset while a second
    thread scores through a frozen model on the SAME dataset — the scores are the bits a lone caller gets; then the partition is
    re-made for a plan with another cut once nobody holds it."""
    import sparkfm_amd
    from sparkfm_amd import _ffi, synth
    from sparkfm_amd.distributed import HipDataParallelSGD, RcclComm
    L = _ffi.load()
    d = synth.make_zipf(77, 6000, 1200, 4, 30, zipf_s=1.05)
    w0, w, v = synth.init_params(8, 1200, 32, stdev=0.05)
    ds = sparkfm_amd.DataSet.from_arrays(d, batch_rows=1500).cache()
    frozen = sparkfm_amd.FMModel(1199, 32)
    frozen.w0, frozen.w, frozen.v = 0.1, np.random.default_rng(2).normal(0, 0.05, 1200), v
    want = frozen.predict(ds)
    fm = sparkfm_amd.FMModel(1199, 32)
    fm.w0, fm.w, fm.v = w0, w, v
    comm = RcclComm(fm, 0, 1).selftest()
    dp = HipDataParallelSGD(comm, eta=0.05, regw=1e-3, regv=1e-3, upper_fractions=(0.1, 0.4), exchange="pipelined")
    dp.plan(fm, ds)
    got, stop = [], threading.Event()

    def scorer():
        while not stop.is_set():
            got.append(frozen.predict(ds))

    t = threading.Thread(target=scorer)
    t.start()
    for _ in range(6):
        dp.learn(fm, ds)
    _ffi.check(L.fmhip_synchronize(fm.handle))
    stop.set()
    t.join()
    assert len(got) >= 1 and all(np.array_equal(g, want) for g in got)
    cuts_a = list(dp.cuts)
    # another plan, another top cut: once the first run is over the partition is simply re-made
    dp.upper_fractions = (0.3,)
    dp.plan(fm, ds)
    assert max(dp.cuts) != max(cuts_a)
    dp.learn(fm, ds)
    fm._device_updated()
    assert np.isfinite(fm.v).all()
    comm.close()
    ds.unpersist()
    fm.close(discard=True)
    frozen.close(discard=True)
