"""The legs of bench.py beside the headline: the CPU baseline (fp64 oracle), ALS, the HBM-resident Criteo-width model, C4 on one GPU."""
import ctypes as C
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from .counters import committed_pmc, live_pmc
from .roofline import HBM_PEAK, alg_bytes, kernel_table, requested_bytes

DP_GLOBAL_BATCH_ROWS = 5_000_000     # rows per data-parallel step over ALL ranks (C4's 10M rows: two steps per epoch)


def host_cores():
    """Host cores this job may use: the affinity mask, the cgroup CPU quota, and the GPU box's
    per-GPU share (16) — whichever is smallest."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(d, k, n1, batch_rows, eta, regs, w0, w, v, budget_s=15.0):
    """fp64 CPU oracle (kind "port"), all host cores, on a bounded sample: the first m
    mini-batches of the same rows with the same schedule; m sized for ~budget_s of CPU work."""
    from oracle import capi
    L = capi.lib()
    threads = host_cores()
    # launchers such as torch.distributed.run export OMP_NUM_THREADS=1 to every rank: the oracle would then run on ONE thread
    # whatever it is asked for (it clamps to omp_get_max_threads) while this record said 16 — raise the OpenMP limit of this
    # process first, and report the threads that really ran
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        pass
    threads = max(1, min(threads, int(L.fmo_max_threads())))
    n_rows = len(d["row_ptr"]) - 1
    nb = -(-n_rows // batch_rows)
    rp = np.ascontiguousarray(d["row_ptr"], np.int64)

    def run(m, reps=1):
        rows = min(n_rows, m * batch_rows)
        nnz = int(rp[rows])
        col = np.ascontiguousarray(d["col"][:nnz], np.int32)
        val = d["val"][:nnz].astype(np.float64)
        y = d["y"][:rows].astype(np.float64)
        vf = np.array(v.T, dtype=np.float64, order="C", copy=True).reshape(-1)
        ww = np.array(w, np.float64)
        w0c = C.c_double(float(w0))
        sub = rp[:rows + 1].copy()
        t = time.perf_counter()
        for _ in range(reps):
            L.fmo_sgd_epoch(k, n1, C.byref(w0c), ww, vf, rows, batch_rows, None, sub, col, val, y,
                            eta, regs[0], regs[1], regs[2], threads)
        return time.perf_counter() - t, nnz * reps, rows

    run(1)                                   # warm-up (page-faults the per-thread buffers, loads the data)
    t1, nnz1, _ = run(1)
    m = int(max(1, min(nb, budget_s / max(t1, 1e-3))))
    reps = int(max(1, min(200, budget_s / max(t1 * m, 1e-3))))
    tm, nnzm, rows = run(m, reps)
    if tm < 0.7 * budget_s and reps < 200:
        # a one-batch call over-estimates a pass (per-call costs: the per-thread gradient buffers are faulted in anew): size the
        # sample from the passes just timed, so that it really is ~budget_s of CPU work
        reps = int(max(reps + 1, min(200, reps * budget_s / max(tm, 1e-3))))
        tm, nnzm, rows = run(m, reps)
    return {"value": nnzm / tm, "unit": "nnz/s", "cores": threads, "kind": "port",
            "sample": "%d pass(es) over the first %d of %d mini-batches (%d rows) of the same workload = %d nnz, "
                      "fp64 oracle, %d OpenMP threads, %.1f s" % (reps, m, nb, rows, nnzm, threads, tm)}


def scoring_leg(fm, ds, rows, nnz_all):
    """The scoring calls of the path (FMModel.predict / computeRMSE) on the line's own dataset:
    fmhip_rmse = the forward over every batch + the statistics, nothing leaves the device but one double."""
    from sparkfm_amd import _ffi
    L = _ffi.load()
    hm, hd = fm.handle, ds.handle
    r_ = C.c_double(0.0)
    _ffi.check(L.fmhip_rmse(hm, hd, C.byref(r_), None))
    _ffi.check(L.fmhip_synchronize(hm))
    n_pass, t0s = 0, time.perf_counter()
    while n_pass < 5 or time.perf_counter() - t0s < 0.5:
        _ffi.check(L.fmhip_rmse(hm, hd, C.byref(r_), None))
        n_pass += 1
    dts = time.perf_counter() - t0s
    return {"what": "fmhip_rmse over the whole dataset (%d rows, %d nonzeros): forward + statistics per batch, %d passes" % (rows, nnz_all, n_pass),
            "value": nnz_all * n_pass / dts, "unit": "nnz/s", "ms_per_pass": dts / n_pass * 1e3, "rmse": r_.value}


def als_c1(device):
    """BASELINE config 1 (10k rows x 1k features, k=8): one ALS.learn epoch — the reference's own fit
    step (S/fm/lib/ALS.scala:15-75) — on the GPU (fp64, fmhip_als_epoch) beside the CPU oracle's."""
    import oracle
    from sparkfm_amd import DataSet, FMModel, HipALS, synth
    d = synth.make_config("C1")
    ds = DataSet.from_arrays(d, name="C1", device=device).cache()
    fm = FMModel(ds.dimension, d["k"], seed=1, device=device)
    w0, w, v = fm.w0, fm.w.copy(), fm.v.copy()
    als = HipALS.run()
    als.learn(fm, ds)                                   # warm-up (allocations)
    t = time.perf_counter()
    for _ in range(3):
        als.learn(fm, ds)
    _ = fm.w0                                           # pulls the fp64 result: includes the sync
    gpu_s = (time.perf_counter() - t) / 3
    val, y = d["val"].astype(np.float64), d["y"].astype(np.float64)
    t = time.perf_counter()
    for _ in range(3):
        w0, w, v = oracle.als_epoch(w0, w, v, 0.0, 0.0, 10.0, d["row_ptr"], d["col"], val, y)
    cpu_s = (time.perf_counter() - t) / 3
    ds.unpersist()
    fm.close()
    nnz = int(d["row_ptr"][-1])
    return {"workload": "C1: 10000 rows x 1000 features, k=8, one ALS epoch (fp64)", "gpu_s_per_epoch": gpu_s,
            "cpu_oracle_s_per_epoch": cpu_s, "nnz": nnz}


def als_long(device, shapes=((100_000, 1_000, 10), (1_000_000, 1_000, 10), (1_000_000, 100, 10)), k=8):
    """ALS.learn (S/fm/lib/ALS.scala:15-75) where its columns are long — the regime a GPU can win: datasets beyond the LDS
    sweep (more than 10,000 rows), uniform ids, columns of 10^3 (one-workgroup runs), 10^4 and 10^5 entries (the chip-wide
    two-launch step of als_kernels.hip).  One epoch on the GPU (fp64, fmhip_als_epoch) beside the CPU oracle's (one core:
    the sweep is a sequential recurrence), and the largest parameter difference between the two."""
    import oracle
    from sparkfm_amd import DataSet, FMModel, HipALS, synth
    out = []
    for n_rows, n_feat, nnz_r in shapes:
        d = synth.make_zipf(synth.BASE_SEED + 77, n_rows, n_feat, nnz_r, nnz_r, zipf_s=0.0)
        ds = DataSet.from_arrays(d, name="als", device=device).cache()
        fm = FMModel(ds.dimension, k, seed=1, device=device)
        w0, w, v = fm.w0, fm.w.copy(), fm.v.copy()
        als = HipALS.run()
        als.learn(fm, ds)                                   # warm-up (allocations); also the epoch that is compared
        _ = fm.w0
        g = (fm.w0, fm.w.copy(), fm.v.copy())
        t = time.perf_counter()
        als.learn(fm, ds)
        _ = fm.w0                                           # pulls the fp64 result: includes the sync
        gpu_s = time.perf_counter() - t
        val, y = d["val"].astype(np.float64), d["y"].astype(np.float64)
        t = time.perf_counter()
        o = oracle.als_epoch(w0, w, v, 0.0, 0.0, 10.0, d["row_ptr"], d["col"], val, y)
        cpu_s = time.perf_counter() - t
        err = max(abs(g[0] - o[0]), float(np.abs(g[1] - o[1]).max()), float(np.abs(g[2] - o[2]).max()))
        nnz = int(d["row_ptr"][-1])
        out.append({"rows": n_rows, "features": n_feat, "k": k, "nnz": nnz, "column_entries": nnz // n_feat,
                    "gpu_s_per_epoch": gpu_s, "cpu_oracle_s_per_epoch": cpu_s, "cpu_over_gpu": cpu_s / gpu_s,
                    "max_abs_parameter_difference_after_one_epoch": err})
        ds.unpersist()
        fm.close()
    return out


def als_fields(device, n_rows=1_000_000, users=6040, items=3706, k=8):
    """ALS.learn on rows shaped like the reference's own demo (S/driver.scala:73-113: MovieLens — a user field and an item
    field, one id each per row, ML-1M's vocabulary sizes): all columns of a field share no row, so the sweep's level
    schedule (fmhip_dataset_als_levels) has TWO levels and every pass is two launches with thousands of columns side by
    side.  One epoch on the GPU (fp64) beside the CPU oracle's (one core: the reference's sweep is a sequential recurrence),
    the largest parameter difference between the two, and the GPU's own sequential walk for comparison."""
    import oracle
    from sparkfm_amd import DataSet, FMModel, HipALS
    rng = np.random.default_rng(20261004)
    # item popularity ~ 1 / (rank + 30): ML-1M's most rated film has ~3,400 of 1M ratings
    pw = 1.0 / (np.arange(items) + 30.0)
    col = np.stack([rng.integers(0, users, n_rows), users + rng.choice(items, n_rows, p=pw / pw.sum())], axis=1).reshape(-1).astype(np.int32)
    val = np.ones(2 * n_rows, np.float64)
    y = rng.integers(1, 6, n_rows).astype(np.float64)
    row_ptr = np.arange(0, 2 * n_rows + 1, 2, dtype=np.int64)
    ds = DataSet(row_ptr, col, val, y, name="fields", device=device).cache()
    lv = ds.alsLevels()
    out = {"workload": "%d rows x (%d user ids + %d item ids), one id per field and row, k=%d, one ALS epoch (fp64)" % (n_rows, users, items, k),
           "levels": lv["levels"], "columns": lv["columns"], "widest_level": lv["widest_level"]}
    res = {}
    for name, env in (("level_schedule", None), ("sequential_walk", "0")):
        if env is None:
            os.environ.pop("FMHIP_ALS_LEVELS", None)
        else:
            os.environ["FMHIP_ALS_LEVELS"] = env
        fm = FMModel(ds.dimension, k, seed=1, device=device)
        w0, w, v = fm.w0, fm.w.copy(), fm.v.copy()
        als = HipALS.run()
        als.learn(fm, ds)                                   # warm-up (allocations); also the epoch that is compared
        _ = fm.w0
        res[name] = (fm.w0, fm.w.copy(), fm.v.copy())
        t = time.perf_counter()
        als.learn(fm, ds)
        _ = fm.w0                                           # pulls the fp64 result: includes the sync
        out["gpu_s_per_epoch_" + name] = time.perf_counter() - t
        fm.close()
    os.environ.pop("FMHIP_ALS_LEVELS", None)
    t = time.perf_counter()
    o = oracle.als_epoch(w0, w, v, 0.0, 0.0, 10.0, row_ptr, col, val, y)
    out["cpu_oracle_s_per_epoch"] = time.perf_counter() - t
    g = res["level_schedule"]
    out["cpu_over_gpu"] = out["cpu_oracle_s_per_epoch"] / out["gpu_s_per_epoch_level_schedule"]
    out["max_abs_parameter_difference_after_one_epoch"] = max(abs(g[0] - o[0]), float(np.abs(g[1] - o[1]).max()), float(np.abs(g[2] - o[2]).max()))
    q = res["sequential_walk"]
    out["max_abs_difference_level_schedule_vs_sequential_walk"] = max(abs(g[0] - q[0]), float(np.abs(g[1] - q[1]).max()), float(np.abs(g[2] - q[2]).max()))
    ds.unpersist()
    return out


def hbm_resident_leg(device, steps=48, rows=6_000_000, batch_rows=250_000, hashed_too=True, with_pmc=True):
    """A model AND a working set that do not fit the caches: C5's width (2^25 hashed slots, k=64 -> V = 8.6 GB, packed
    gradient 8.9 GB) on one GPU, 6M Criteo-shaped rows = 24 DISTINCT mini-batches of 250k rows, weight decay on (lazy
    rows-only update).  One batch touches ~0.4M parameter rows (~100 MB of V); 24 different ones in a row push well over
    1 GB of V rows, 1.5 GB of P and 1.7 GB of index/value streams through the 256 MiB Infinity Cache between two uses of a
    line (round 2 cycled TWO batches: ~120 MB of V, cache-resident).  Run with the ids relabelled by frequency at load and,
    for comparison, as hashed.  The one place where the counters' bytes are, to a large part, HBM bytes."""
    from sparkfm_amd import DataSet, FeatureOrder, FMModel, _ffi, synth
    L = _ffi.load()
    n1, k = 1 << 25, 64
    regs = (0.0, 1e-4, 1e-4)
    ab = alg_bytes(k)
    t0 = time.time()
    d = synth.make_config("C5", rows=rows)
    t_gen = time.time() - t0
    col_hashed = d["col"]
    nnz_total = int(d["row_ptr"][-1])

    def run(col, n_steps, name):
        d["col"] = col
        ds = DataSet.from_arrays(d, batch_rows=batch_rows, device=device).cache()
        fm = FMModel(n1 - 1, k, seed=5, device=device, init_on_device=True)
        hm, hd, nb = fm.handle, ds.handle, ds.n_batches
        bnnz = [ds.batch_info(b)["nnz"] for b in range(nb)]
        for j in range(min(nb, 8)):
            _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, 0.02, *regs, None))
        _ffi.check(L.fmhip_synchronize(hm))
        _ffi.check(L.fmhip_profile_begin_sampled(hm, 2))
        t1 = time.perf_counter()
        for j in range(n_steps):
            _ffi.check(L.fmhip_sgd_step(hm, hd, (8 + j) % nb, 0.02, *regs, None))
        _ffi.check(L.fmhip_synchronize(hm))
        dt = time.perf_counter() - t1
        prof = _ffi.Profile()
        _ffi.check(L.fmhip_profile_end(hm, C.byref(prof)))
        st = _ffi.Stats()
        _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))
        nnz = sum(bnnz[(8 + j) % nb] for j in range(n_steps))
        lay = ds.layout()
        bi = ds.batch_info(0)
        touched = [ds.batch_info(b)["n_columns"] for b in range(nb)]
        res = dict(name=name, value=nnz / dt, step_ms=dt / n_steps * 1e3, steps=n_steps, batches=nb, prof=prof, lay=lay, bi=bi,
                   mse=st.sse / max(st.rows, 1), nonfinite=st.nonfinite, touched_rows_per_batch=float(np.mean(touched)))
        ds.unpersist()
        fm.close(discard=True)
        return res

    t0 = time.time()
    col_rel = FeatureOrder.fit(col_hashed, n1, device=device).relabel(col_hashed)      # a pure renaming (sparkfm_amd.FeatureOrder, on the GPU), outside any timed region
    t_rel = time.time() - t0
    r = run(col_rel, steps, "relabelled")
    lay, bi, prof = r["lay"], r["bi"], r["prof"]
    share = lay["nnz_sparse"] / max(nnz_total, 1)            # what stayed in the sparse streams (forward)
    share_b = lay["nnz_sparse_backward"] / max(nnz_total, 1)   # ... in the transposes (backward)
    req = requested_bytes(64, bi["rows"], bi["nnz"], int(bi["nnz"] * share), bi["n_columns"], bool(lay["hot_ids"]),
                          bi["n_columns"], False, n1, False, int(bi["nnz"] * share_b), lay["hot_pages"])
    pmc = committed_pmc("C5hbm", k, batch_rows)
    live = None
    if with_pmc:
        # counters of THIS workload shape, measured now (8 distinct batches are enough to defeat the Infinity Cache; the
        # 24-batch run above is the timed one)
        live = live_pmc([os.path.join(ROOT, "tools", "pmc_leg.py"), "c5hbm", "--rows", "2000000", "--batch-rows", str(batch_rows)])
        for kn, e in (live or {}).items():
            pmc.setdefault(kn, {}).update(e, traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation "
                                                            "(tools/pmc_leg.py c5hbm, 8 distinct batches)")
    kern = kernel_table(prof, k, 64, req, pmc, {"forward": n1 * 64 * 4, "backward": bi["rows"] * 64 * 4})
    step_req = sum(e.get("requested_bytes_per_launch", 0) for e in kern.values())
    step_ms = r["step_ms"]
    fabric = pmc.get("step", {}).get("traffic_bytes")
    out = {"workload": "C5 width on one GPU: %d Criteo-shaped rows x 2^25 hashed slots (relabelled by frequency at load), k=64 (V = %.1f GB), "
                       "%d DISTINCT batches of %d rows (%.2f M parameter rows touched per batch), eta 0.02, regw = regv = 1e-4 (lazy rows-only update)" %
                       (rows, n1 * k * 4 / 1e9, r["batches"], batch_rows, r["touched_rows_per_batch"] / 1e6),
           "value": r["value"], "unit": "nnz/s", "ms_per_step": step_ms, "steps": r["steps"], "distinct_batches": r["batches"],
           "touched_V_bytes_per_batch": r["touched_rows_per_batch"] * 64 * 4,
           "hot_block_features": len(lay["hot_ids"]), "hot_block_features_gradient_side": len(lay["hot_ids_all"]),
           "share_of_nonzeros_in_sparse_streams": share, "share_of_nonzeros_in_transposes": share_b,
           "alg_bytes_per_nnz": ab["step"], "alg_GBps": r["value"] * ab["step"] / 1e9,
           "requested_bytes_per_step": step_req, "requested_GBps": step_req / (step_ms * 1e-3) / 1e9,
           "fabric_traffic_bytes_per_step": fabric,
           "fabric_traffic_source": ("live rocprofv3 --pmc passes of this run" if live else "profiles/pmc_traffic.json (committed passes)") if fabric else None,
           "fabric_GBps": fabric / (step_ms * 1e-3) / 1e9 if fabric else None,
           "frac_of_8TBps": fabric / (step_ms * 1e-3) / HBM_PEAK if fabric else None,
           "note": "three byte counts, never to be mixed: ALGORITHMIC (8k+16 B for every stored nonzero: the 13 numeric fields and the "
                   "small vocabularies sit in the dense hot block, popular slots hit the caches, so this exceeds what HBM moves), REQUESTED "
                   "(our own count of the kernels' loads and stores, whatever level serves them) and FABRIC (rocprofv3 FETCH_SIZE x2 + "
                   "WRITE_SIZE: requests that left the L2s; Infinity-Cache hits are still included, no DRAM-side counter separates them on "
                   "this part — profiles/README.md — so an upper bound on HBM bytes).  frac_of_8TBps = fabric bytes / this run's step time / 8 TB/s.",
           "kernels": kern, "last_batch_mse": r["mse"], "nonfinite": r["nonfinite"],
           "setup_s": {"generate": t_gen, "relabel": t_rel}}
    del col_rel
    if hashed_too:
        h = run(col_hashed, max(steps // 2, 8), "hashed")
        pd = h["prof"].as_dict()
        out["ids_as_hashed"] = {"value": h["value"], "unit": "nnz/s", "ms_per_step": h["step_ms"], "steps": h["steps"],
                                "kernel_ms": {n: p["ms"] / p["launches"] for n, p in pd.items() if p["launches"]},
                                "last_batch_mse": h["mse"], "nonfinite": h["nonfinite"],
                                "note": "the same rows with the slots numbered as the hash left them (no frequency relabelling at load)"}
    return out


def c4_one_gpu_leg(device, eta, regs, rows=10_000_000, batch_rows=DP_GLOBAL_BATCH_ROWS, passes=3):
    """BASELINE config 4 — ALL of its 10M rows x 1M features, k=32 — on ONE GPU with the data-parallel runs' GLOBAL batch
    (5M rows: the same job, the same SGD trajectory) and the plain step: the denominator the N > 1 lines (C4 sharded over
    N GPUs) are to be divided by, in the driver-run N = 1 record.  (One GPU's rate hardly depends on the batch: 31.5 / 35.6 /
    32.7 / 31.9 G nnz/s at 625k / 1.25M / 2.5M / 5M rows, tools/c4_batch_sweep.sh.)"""
    from sparkfm_amd import DataSet, FMModel, _ffi, synth
    L = _ffi.load()
    cfg = synth.CONFIGS["C4"]
    t0 = time.time()
    d = synth.make_config("C4", rows=rows)
    t_gen = time.time() - t0
    t0 = time.time()
    ds = DataSet.from_arrays(d, name="C4", batch_rows=batch_rows, device=device).cache()
    t_load = time.time() - t0
    fm = FMModel(cfg["features"] - 1, cfg["k"], seed=cfg["seed"] + 1000, device=device, init_on_device=True)
    hm, hd, nb = fm.handle, ds.handle, ds.n_batches
    nnz = int(d["row_ptr"][-1])
    for j in range(nb):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j, eta, *regs, None))
    _ffi.check(L.fmhip_synchronize(hm))
    st0 = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st0)))
    t0 = time.perf_counter()
    for _ in range(passes):
        for j in range(nb):
            _ffi.check(L.fmhip_sgd_step(hm, hd, j, eta, *regs, None))
    _ffi.check(L.fmhip_synchronize(hm))
    dt = time.perf_counter() - t0
    st = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))
    out = {"workload": "C4 on one GPU: %d rows x %d features, k=%d, batch %d rows (%d batches), the plain step (fmhip_sgd_step), %d passes" %
                       (rows, cfg["features"], cfg["k"], batch_rows, nb, passes),
           "value": nnz * passes / dt, "unit": "nnz/s", "ms_per_step": dt / (passes * nb) * 1e3, "steps": passes * nb, "nnz": nnz,
           "last_batch_mse_after_first_pass": st0.sse / max(st0.rows, 1), "last_batch_mse": st.sse / max(st.rows, 1),
           "nonfinite": st.nonfinite, "setup_s": {"generate": t_gen, "load_transpose_h2d": t_load}}
    ds.unpersist()
    fm.close(discard=True)
    return out

