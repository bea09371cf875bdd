#!/usr/bin/env python3
"""bench.py — nnz/sec of FM mini-batch SGD training on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C4|C5] [--batch-rows B]

A "step" is one mini-batch SGD step (forward + backward + update) over one batch of the
synthetic workload, all inputs resident in HBM before the timed region.

  N = 1   BASELINE config 3 (1M rows x 100k features, k=32): the configuration the metric is quoted on.
  N > 1   BASELINE config 4 (10M rows x 1M features, k=32) sharded by rows: every rank owns a shard of
          the same virtual dataset; ONE job whatever N — a global mini-batch of 5M rows per step (the same
          SGD trajectory on 2, 4 or 8 GPUs), 5M / N rows of it on every rank; the packed gradient
          (136 MB) is all-reduced over RCCL/xGMI every step INSIDE the library (fmhip_dp_step:
          overlapped with the feature-chunked backward).  Total work per step is fixed -> "strong"
          (the one-GPU denominator of the same job is `extra.c4_one_gpu` of the N = 1 record).
          Started either by the driver (python -m torch.distributed.run ... bench.py --gpus N) or by
          `python bench.py --gpus N` alone: with WORLD_SIZE unset the parent spawns the N ranks itself —
          before it has touched the GPU — and forwards rank 0's JSON line.  `--transport threads` runs the N
          ranks as THREADS of this process on GPU 0 (a rehearsal of the whole N-rank flow on a one-GPU box:
          a world of 8 fits neither RCCL, one rank per device, nor the test pool's 6 processes per card).

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events (recorded by the library on
the stream its kernels run on) over the timed steps; `cpu_baseline` times the fp64 CPU oracle (a port —
SparkFM itself needs a JVM, absent here) on a bounded sample of the same workload; `sustained` repeats
the steps back to back for >= 2 s; `extra.hbm_resident` is a Criteo-width model (V = 8.6 GB, k=64: the
only configuration whose tables do not live in L2 / Infinity Cache) with weight decay.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DP_GLOBAL_BATCH_ROWS = 5_000_000     # rows per data-parallel step over ALL ranks (C4's 10M rows: two steps per epoch)
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); measured streaming copy ~6.3e12
# Ceilings for gathers of whole 128-B-multiple rows by where the table lives (MI355X_MICROARCH.md,
# "Indexed rows"): the XCD's own L2, the Infinity Cache, HBM (measured sweep / spec peak)
# (the UPPER end of the guide's measured ranges — 16.8-18.8 TB/s from L2 — so that a "ceiling" is one: a kernel that also
# hits in its CU's L1, which these rates do not price, must not pass it)
CEIL = {"l2_gather": 18.8e12, "mall_gather": 8.6e12, "hbm_gather": 6.1e12, "hbm_stream": 6.3e12}
L2_BYTES_PER_XCD = 4 << 20
MALL_BYTES = 256 << 20


def alg_bytes(k):
    """SURVEY.md §8(d) algorithmic bytes per stored nonzero, split by kernel (fp32/int32):
    forward  = col 4 + val 4 + V-row read 4k + w read 4      = 4k + 12
    backward = V-grad row add 4k + w-grad add 4               = 4k + 4   (SURVEY's figure; the walk itself reads 8 + 4k per entry)
    whole step B_alg(k) = 8k + 16 (plus 16 B/row and 12(n+1)(k+1) B/step for the dense update)."""
    return {"forward": 4 * k + 12, "backward": 4 * k + 4, "step": 8 * k + 16}


def requested_bytes(kp, rows, nnz, nnz_sparse, n_cols, hot, touched_rows, dense_apply, n1p, packed, nnz_sparse_bwd=None, hot_pages=1):
    """Bytes each kernel of one step actually ASKS the memory system for (our own count of its loads and
    stores, whatever level serves them), and the table its gathers hit.  nnz_sparse / nnz_sparse_bwd: the entries
    of the batch in the CSR stream (forward) / in the transposed stream (backward: fewer, the gradient-side pages
    of the dense hot block are not in it); the block product streams P once and 64 B per row and page."""
    row = 4 * kp
    hot_b = 64 * rows if hot else 0
    if nnz_sparse_bwd is None:
        nnz_sparse_bwd = nnz_sparse
    fwd = nnz_sparse * (8 + row + (0 if packed else 4)) + rows * (8 + 4 + row + 4) + hot_b
    # (no separate residual read: with a spare slot e sits in the P row, without one it rides in the row's low mantissa bits)
    bwd = nnz_sparse_bwd * (8 + row) + n_cols * (row + 8) + (rows * row + hot_b * max(hot_pages, 1) if hot else 0)
    apply_rows = n1p if dense_apply else touched_rows
    app = apply_rows * (3 * row + 16)          # V read+write, G read (+ zero store counted with the write)
    return {"forward": fwd, "backward": bwd, "apply": app}


def gather_ceiling(table_bytes, l2_hit=None):
    """Ceiling for a kernel bound by gathers from a table of `table_bytes` that every XCD reads: the
    blend of the L2 and Infinity-Cache gather rates at L2 hit rate h (measured by rocprofv3 where a
    committed profile exists, else the uniform-gather share min(1, 4 MiB / table)); tables beyond the
    Infinity Cache gather at the HBM rate."""
    if table_bytes > MALL_BYTES:
        if l2_hit is None:
            return "hbm_gather", CEIL["hbm_gather"], None
        # skewed gathers from a table in HBM: the measured share hits L2; what misses is served by the Infinity Cache or by
        # HBM in a proportion no counter separates — priced at the faster of the two, so this stays an upper bound
        c = 1.0 / (l2_hit / CEIL["l2_gather"] + (1.0 - l2_hit) / CEIL["mall_gather"])
        return "l2_gather x %.2f + mall_gather x %.2f (L2 misses priced at the Infinity-Cache rate: upper bound)" % (l2_hit, 1.0 - l2_hit), c, l2_hit
    h = l2_hit if l2_hit is not None else min(1.0, L2_BYTES_PER_XCD / max(table_bytes, 1))
    c = 1.0 / (h / CEIL["l2_gather"] + (1.0 - h) / CEIL["mall_gather"])
    return "l2_gather x %.2f + mall_gather x %.2f" % (h, 1.0 - h), c, h


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
    return {"value": nnzm / tm, "unit": "nnz/s", "cores": threads, "kind": "port",
            "sample": "%d pass(es) over the first %d of %d mini-batches (%d rows) of the same workload = %d nnz, "
                      "fp64 oracle, %d OpenMP threads, %.1f s" % (reps, m, nb, rows, nnzm, threads, tm)}


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


def committed_pmc(config, k, batch_rows):
    """Counter-derived figures of the committed rocprofv3 --pmc passes for this configuration
    (profiles/pmc_traffic.json): {kernel: {traffic_bytes, l2_hit}}; empty when no pass exists."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entries = json.load(f)["entries"]
    except (OSError, ValueError, KeyError):
        return {}
    out = {}
    # C5's kernels, width and batch are those of the HBM-resident leg ("C5hbm": the passes wrap tools/run_c5_shape.py)
    for name in (config, config + "hbm"):
        for e in entries:
            if (e.get("config"), e.get("k"), e.get("batch_rows")) == (name, k, batch_rows):
                out.setdefault(e["kernel"], e)
    return out


STEP_KERNELS = ("k_forward", "k_backward", "k_fixup", "k_apply")
PMC_STATE = {"dead": False}      # a counter pass that had to be killed ends the live collection for the run


def pmc_pass(counters, child_argv, skip=4, timeout_s=150, per_step=None):
    """One `rocprofv3 --pmc <counters> -- python3 <child_argv>` run (counter collection only: no trace domain beside it) as
    a CHILD process; -> {kernel: {counter: mean per dispatch after the first `skip` dispatches of that kernel}} for the
    kernels of the SGD step, or None when rocprofv3 is not there / fails (the caller falls back to the committed profile).
    per_step = (warmup_steps, timed_steps) of the child: a kernel the step launches several times (the data-parallel step's
    backward runs once per feature interval) is then summed over a step's launches — the figure is per STEP of that kernel.
    Kernel names are folded as in tools/make_pmc_json.py (k_forward_wt -> k_forward, k_backward_p -> k_backward, ...).
    A pass that runs into its time limit is killed with its whole process group (rocprofv3's grandchild would otherwise keep
    the GPU busy beside the timed legs that follow) and ends the live collection for this run."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out_dir = tempfile.mkdtemp(prefix="fmhip_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc"] + list(counters) + ["-d", out_dir, "-o", "pmc", "--output-format", "csv", "--", sys.executable] + list(child_argv)
        env = dict(os.environ, TMPDIR="/tmp")
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.communicate()
            PMC_STATE["dead"] = True
            sys.stderr.write("[bench] rocprofv3 --pmc %s ran into its %d s limit: process group killed, no further counter passes in this run\n" %
                             (" ".join(counters), timeout_s))
            return None
        if proc.returncode != 0:
            sys.stderr.write("[bench] rocprofv3 --pmc %s failed (rc %d): %s\n" % (" ".join(counters), proc.returncode, err.decode()[-400:]))
            return None
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r"(k_[a-z_0-9]+)", row["Kernel_Name"])
                if not m or not m.group(1).startswith(STEP_KERNELS):
                    continue
                kn = m.group(1).replace("k_forward_wt", "k_forward").replace("k_forward_lds", "k_forward").replace("k_backward_p", "k_backward").replace("k_apply_rows", "k_apply")
                agg[kn][row["Counter_Name"]].append(float(row["Counter_Value"]))
        def mean(v):
            if per_step and len(v) % (per_step[0] + per_step[1]) == 0:
                lps = len(v) // (per_step[0] + per_step[1])              # launches of this kernel per step
                return sum(v[per_step[0] * lps:]) / per_step[1]
            return sum(v[skip:]) / max(len(v[skip:]), 1)
        return {kn: {cn: mean(v) for cn, v in cs.items() if len(v) > skip} for kn, cs in agg.items()} or None
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as ex:
        sys.stderr.write("[bench] rocprofv3 --pmc pass failed: %r\n" % (ex,))
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def live_pmc(child_argv, per_step=None):
    """Fabric-side traffic per launch of the step's kernels, measured NOW: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE
    do not fit one) over tools/pmc_leg.py running the same workload.  Units and the gfx950 correction as
    MI355X_MICROARCH.md prescribes: both counters are KiB; FETCH_SIZE tallies the 128-B requests of wide (16 B per lane)
    reads at 64 B — the row gathers and the dense block's streams are such reads, the 4-B index / value streams are not and
    the counter cannot tell them apart, so the doubled figure is an upper bound.  -> {kernel: {...}} or None."""
    if PMC_STATE["dead"]:
        return None
    fetch = pmc_pass(["FETCH_SIZE"], child_argv, per_step=per_step)
    write = pmc_pass(["WRITE_SIZE"], child_argv, per_step=per_step) if fetch and not PMC_STATE["dead"] else None
    if not fetch or not write:
        return None
    out = {}
    for kn in fetch:
        fr, wr = fetch[kn].get("FETCH_SIZE"), write.get(kn, {}).get("WRITE_SIZE")
        if fr is None or wr is None:
            continue
        out[kn] = {"fetch_raw_bytes": int(fr * 1024), "fetch_corrected_bytes": int(2 * fr * 1024), "write_bytes": int(wr * 1024),
                   "traffic_bytes": int(2 * fr * 1024 + wr * 1024)}
    if out:
        out["step"] = {"traffic_bytes": sum(e["traffic_bytes"] for e in out.values())}
        # the hit rates of THIS run (one more pass: the L2's hits / misses and the L1s' accesses / requests passed on to L2 fit
        # one counter set): what the gather ceilings are blended with, instead of the committed profile's figure
        hits = None if PMC_STATE["dead"] else pmc_pass(["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
                                                       child_argv, per_step=per_step)
        for kn, c in (hits or {}).items():
            if kn in out and c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
                out[kn]["l2_hit"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
                out[kn]["l2_hit_measured_in_this_run"] = True
                if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                    out[kn]["l1_hit_share_of_accesses"] = 1.0 - c.get("TCP_TCC_READ_REQ_sum", 0.0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    return out or None


def roofline_block(kern, dom, ab, pd, pmc, step_ms, live):
    """`roofline` of the JSON line for the dominant kernel `dom`: achieved = bytes the rocprofv3 counters saw leave the L2s per
    launch (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md's units and gfx950 correction) / that kernel's launch duration
    measured by HIP events in THIS run; frac = achieved / 8 TB/s.  The algorithmic figure of SURVEY §8(d) is kept beside it,
    flagged: it prices every stored nonzero at a gathered row and is not an HBM rate."""
    e = kern.get(dom, {})
    avg_ms = e.get("avg_ms")
    traffic = e.get("traffic_bytes")
    basis = "counters"
    if traffic is None:           # no profile of this configuration anywhere: our own count of the kernel's loads and stores
        traffic, basis = e.get("requested_bytes_per_launch"), "requested bytes (no counter pass exists for this configuration)"
    achieved = traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms else None
    requested_only = None
    if basis != "counters":
        # our own count of the kernel's loads and stores is what it ASKS of the memory system, caches included — not an HBM-side
        # figure and no roofline: reported beside the (empty) roofline, never as its `achieved`
        requested_only = {"requested_bytes": traffic, "requested_GBps": achieved, "note": "no counter pass exists for this configuration: no HBM-side figure is claimed"}
        achieved = None
    step_traffic = pmc.get("step", {}).get("traffic_bytes")
    alg = e.get("alg_GBps")
    return {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved * 1e9 / HBM_PEAK if achieved else None, "traffic": traffic if basis == "counters" else None, "basis": basis,
            "requested_only": requested_only,
            "traffic_source": e.get("traffic_source"), "traffic_measured_in_this_run": bool(live),
            "avg_launch_ms": avg_ms, "nnz_per_launch": pd[dom]["nnz"] / max(pd[dom].get("steps") or pd[dom]["launches"], 1),
            "launches_per_step": e.get("launches_per_step", 1),
            "what": "achieved = fabric-side bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE: what left the L2s; Infinity-Cache hits "
                    "are included — the part exposes no DRAM-side or MALL hit counter, profiles/README.md — so an upper bound on HBM bytes) / "
                    "the launch's duration by HIP events in this run; frac = achieved / 8 TB/s",
            "step": {"traffic": step_traffic, "achieved": step_traffic / (step_ms * 1e-3) / 1e9 if step_traffic else None,
                     "frac": step_traffic / (step_ms * 1e-3) / HBM_PEAK if step_traffic else None,
                     "what": "the same for the whole step: the counters' bytes of all its kernels / the measured step time"},
            "algorithmic_bytes_per_nnz": ab[dom], "algorithmic_achieved": alg, "algorithmic_frac": alg * 1e9 / HBM_PEAK if alg else None,
            "algorithmic_note": "SURVEY §8(d)'s figure (4k+4 B for EVERY stored nonzero of the batch / launch time): NOT an HBM rate and may "
                                "pass the peak — V, P and the gradient live in L2 / Infinity Cache at this size and the nonzeros of the dense "
                                "hot block cost one streamed value instead of a gathered row",
            "requested_GBps": e.get("requested_GBps"), "ceiling": e.get("ceiling"), "frac_of_ceiling": e.get("frac_of_ceiling")}


def kernel_table(prof, k, kp, req, pmc, table_bytes):
    """Per-kernel: HIP-event time, algorithmic rate, requested-byte rate, the ceiling
    that binds it and the fraction of THAT ceiling.  A kernel the data-parallel step launches once per feature interval
    (backward, fixup, update) is summed over the launches of one step — the profile counts the steps each kind was timed
    in — so its time, nonzeros and bytes are per STEP, which is what the requested-byte and counter figures beside them are."""
    ab = alg_bytes(k)
    pd = prof.as_dict()
    tot_ms = max(sum(x["ms"] for x in pd.values()), 1e-12)
    kern = {}
    for name, p in pd.items():
        if not p["launches"]:
            continue
        steps = max(p.get("steps") or p["launches"], 1)
        lps = p["launches"] / steps
        avg_ms = p["ms"] / steps
        ent = {"avg_ms": avg_ms, "launches": p["launches"], "share": p["ms"] / tot_ms}
        if lps > 1:
            ent["launches_per_step"] = lps
            ent["avg_ms_is"] = "the sum over the %.3g launches of one step (one per feature interval)" % lps
        if name in ab:
            ent["alg_bytes_per_nnz"] = ab[name]
            ent["alg_GBps"] = (p["nnz"] / steps) * ab[name] / (avg_ms * 1e-3) / 1e9
        if name in req:
            ent["requested_bytes_per_launch"] = req[name]
            ent["requested_GBps"] = req[name] / (avg_ms * 1e-3) / 1e9
            if name in ("forward", "backward"):
                hit = pmc.get("k_" + name, {}).get("l2_hit")
                cname, c, h = gather_ceiling(table_bytes[name], hit)
                if cname == "hbm_gather":
                    # a table beyond the Infinity Cache and no measured hit rate: skewed gathers are served by the caches in a
                    # share nobody measured here, so no rate is a ceiling for them — none is claimed
                    ent["ceiling"] = None
                    ent["frac_of_ceiling"] = None
                    ent["ceiling_note"] = "no ceiling claimed: the table is beyond the Infinity Cache and this configuration has no measured L2 hit rate"
                else:
                    measured = pmc.get("k_" + name, {}).get("l2_hit_measured_in_this_run")
                    ent["ceiling"] = {"name": cname, "GBps": c / 1e9, "table_bytes": table_bytes[name], "l2_hit": h,
                                      "l2_hit_source": (("rocprofv3 --pmc TCC_HIT / TCC_MISS pass of this run" if measured else "profiles/pmc_traffic.json")
                                                        if hit is not None else ("uniform-gather model" if h is not None else None))}
                    l1 = pmc.get("k_" + name, {}).get("l1_hit_share_of_accesses")
                    if l1 is not None:
                        ent["ceiling"]["l1_hit_share_of_accesses"] = l1      # served by the CU's own L1: not priced by the ceiling (it only adds headroom)
            else:
                ent["ceiling"] = {"name": "hbm_stream", "GBps": CEIL["hbm_stream"] / 1e9}
            if ent.get("ceiling"):
                ent["frac_of_ceiling"] = ent["requested_GBps"] / ent["ceiling"]["GBps"]
            if (ent.get("frac_of_ceiling") or 0.0) > 1.0:
                ent["ceiling_exceeded"] = ("the kernel asked for bytes faster than the L2 / Infinity-Cache gather rates allow: the excess was "
                                           "served by the CUs' L1s, which the ceiling does not price")
        pe = pmc.get("k_" + name, {})
        if pe.get("traffic_bytes") is not None:
            ent["traffic_bytes"] = pe["traffic_bytes"]           # fabric-side: FETCH_SIZE x2 + WRITE_SIZE per launch
            ent["traffic_source"] = pe.get("traffic_source", "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this configuration)")
            ent["traffic_GBps"] = pe["traffic_bytes"] / (avg_ms * 1e-3) / 1e9
            ent["traffic_frac_of_8TBps"] = ent["traffic_GBps"] * 1e9 / HBM_PEAK
        kern[name] = ent
    if "apply" in req and "apply" not in kern and "fixup" in kern:
        # merged finish: the dense update ran inside the fixup launch (fmhip_tune key 11)
        ent = kern["fixup"]
        ent["includes"] = "the parameter update (merged finish)"
        ent["requested_bytes_per_launch"] = req["apply"]
        ent["requested_GBps"] = req["apply"] / (ent["avg_ms"] * 1e-3) / 1e9
        ent["ceiling"] = {"name": "hbm_stream", "GBps": CEIL["hbm_stream"] / 1e9}
        ent["frac_of_ceiling"] = ent["requested_GBps"] / ent["ceiling"]["GBps"]
    return kern


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


_PG_GENERATION = [0]


def init_process_group(dist, backend, **kw):
    """torch.distributed's rendezvous: the launcher's env:// (torch.distributed.run sets MASTER_*), or — ranks spawned by this file —
    a file store, one file per process group this run creates (the fallback exchange makes a second one)."""
    rdzv = os.environ.get("FMHIP_BENCH_RDZV")
    if rdzv:
        _PG_GENERATION[0] += 1
        return dist.init_process_group(backend, init_method="%s.%d" % (rdzv, _PG_GENERATION[0]), rank=int(os.environ["RANK"]),
                                       world_size=int(os.environ["WORLD_SIZE"]), **kw)
    return dist.init_process_group(backend, **kw)


class TorchCtl:
    """The bench's control plane over torch.distributed (gloo; nccl when the exchange itself is torch's): barriers and
    reductions of a few timers — never the gradients."""

    def __init__(self, dist, torch, on_gpu):
        self.dist, self.torch, self.on_gpu = dist, torch, on_gpu

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, values, op="max"):
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64)
        if self.on_gpu:
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(x) for x in t.cpu()]

    def sum_counts(self, counts):
        t = self.torch.from_numpy(counts)
        if self.on_gpu:
            t = t.cuda()
        self.dist.all_reduce(t)
        return t.cpu().numpy()


class ThreadCtl:
    """The same over the ranks-as-threads group (--transport threads)."""

    def __init__(self, group, rank):
        self.group, self.rank = group, rank

    def barrier(self):
        self.group.barrier()

    def allreduce(self, values, op="max"):
        return self.group.allreduce(self.rank, values, op)

    def sum_counts(self, counts):
        return sum(self.group.exchange(self.rank, counts))


class NoCtl:
    def barrier(self):
        pass

    def allreduce(self, values, op="max"):
        return [float(v) for v in values]

    def sum_counts(self, counts):
        return counts


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this process
    has not touched the GPU and never will), forward rank 0's JSON line, exit with the worst exit code."""
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    # the ranks meet through a file store in a fresh directory: a port found by binding to 0 and closing it can be taken by
    # someone else before rank 0 binds it again (EADDRINUSE, seen once on a GPU box)
    rdzv = "file://" + os.path.join(tempfile.mkdtemp(prefix="fmhip_bench_rdzv_"), "store")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), FMHIP_BENCH_RDZV=rdzv,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies (no such GPU, out of memory ...) must not leave the others waiting in a collective
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed:
            rc = abs(failed[0].returncode) or 1
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        rc = max(rc, abs(p.returncode or 0))
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=["C1", "C2", "C3", "C4", "C5"],
                    help="default: C3 on one GPU (the metric's configuration), C4 on several (BASELINE's 8-GPU config)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the config's rows / N, capped at 5M — two global batches; 1.25M on one GPU)")
    ap.add_argument("--batch-rows", type=int, default=0, help="mini-batch rows per GPU (default 250000; data-parallel: 5M / N, the global batch is fixed)")
    ap.add_argument("--eta", type=float, default=0.02)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the sustained / HBM-resident / ALS legs")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes that measure the kernels' fabric-side traffic in this run "
                         "(roofline.traffic then comes from the committed profile, profiles/pmc_traffic.json)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"],
                    help="rccl: the library's own communicator (fmhip_dp_step); torch: torch.distributed all-reduce "
                         "orchestrated from Python (sparkfm_amd.distributed.DataParallelSGD)")
    ap.add_argument("--dp-exchange", default="auto", choices=["auto", "dense", "sharded", "touched", "pipelined"],
                    help="what a data-parallel step exchanges (fmhip_dp_exchange): dense = the whole packed gradient all-reduced in "
                         "overlapped slices, every rank updates every row; sharded = the slices reduce-scattered, every rank updates its "
                         "1/N share, the updated rows all-gathered; touched = only the rows some rank touched; auto = touched for C5 "
                         "(an 8.9 GB gradient), otherwise dense and sharded are both timed during warm-up and the faster is kept")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host", "threads"],
                    help="rccl: one rank per GPU, the library's RCCL communicator; host: ALL ranks on GPU 0, the library's step over "
                         "fmhip_comm_create_external with every collective staged through the host and summed by gloo — the same "
                         "schedule, plan and update, for boxes with fewer GPUs than ranks (a rehearsal, not a measurement); threads: the "
                         "same with the ranks as THREADS of this one process (ThreadStagedComm) — a world of 8 on a one-GPU box, where "
                         "RCCL wants one device per rank and the test pool admits 6 processes per card")
    ap.add_argument("--upper-fractions", default="auto",
                    help="cuts of the backward for the overlapped exchange: comma-separated ascending shares of the nonzeros at or "
                         "above each cut (e.g. 0.3 or 0.12,0.4), 'none' = one all-reduce after the whole backward, 'auto' = "
                         "time a few candidates during warm-up and keep the fastest (all ranks agree through a max-reduce)")
    ap.add_argument("--force-dp", action="store_true",
                    help="self-test: take the data-parallel path (RCCL all-reduce included) even with one rank")
    ap.add_argument("--emulate-allreduce", default="",
                    help="RANKS:BUSBW_GBps, one-rank runs only (with --force-dp): hold the comm stream after every collective for "
                         "the time a ring all-reduce over RANKS GPUs at that bus bandwidth would take (fmhip_comm_emulate), so the "
                         "overlap schedule and the cut tuning can be timed on a one-GPU box")
    ap.add_argument("--emulate-load", type=int, default=0,
                    help="with --emulate-allreduce: spend every emulated collective's duration with this many workgroups streaming the "
                         "payload through HBM (fmhip_comm_emulate_load) instead of idling — the CU slots and memory bandwidth a real "
                         "collective takes from the backward beside it")
    ap.add_argument("--no-relabel", action="store_true",
                    help="C5 only: keep the hashed ids as generated instead of relabelling them by frequency at load")
    ap.add_argument("--hot-pages", type=int, default=0,
                    help="A/B: pages of the dense hot block (fmhip_tune key 12; 1 = the two-sided page only, default = library's 3)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B: fmhip_tune(KEY, VALUE) before anything is built (repeatable); see include/fmhip.h")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    args = ap.parse_args()
    if args.transport == "threads":
        if args.exchange != "rccl":
            raise SystemExit("--transport threads runs the library's own step (--exchange rccl)")
        # stdout carries exactly one JSON line (see below)
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch
        from sparkfm_amd import _ffi, synth
        from sparkfm_amd.distributed import run_thread_ranks
        _ffi.load()
        synth.set_threads(max(1, host_cores() // max(1, min(args.gpus, 8))))
        run_thread_ranks(args.gpus, lambda r, g: run_rank(args, r, args.gpus, 0, ThreadCtl(g, r), json_fd, torch, group=g), timeout=1800.0)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args, sys.argv[1:])
    # stdout carries exactly one JSON line: libraries that print banners to fd 1 (RCCL prints its
    # version block there at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    if args.transport == "host":
        local_rank = 0                      # every rank shares the one GPU; the collectives are staged through the host
    torch.cuda.set_device(local_rank)
    use_dp = world > 1 or args.force_dp
    ctl = NoCtl()
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.exchange == "rccl":
            # control plane only (unique id, barriers, the max over ranks): the gradients never pass through it
            init_process_group(dist, "gloo")
        else:
            init_process_group(dist, "nccl", device_id=torch.device("cuda", local_rank))
        ctl = TorchCtl(dist, torch, args.exchange == "torch")
    run_rank(args, rank, world, local_rank, ctl, json_fd, torch)
    if use_dp:
        dist.destroy_process_group()


def run_rank(args, rank, world, local_rank, ctl, json_fd, torch, group=None):
    """One rank of the bench (a process, or a thread under --transport threads)."""
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dp = world > 1 or args.force_dp
    exchange = args.exchange if use_dp else "none"
    from sparkfm_amd import DataSet, FMModel, _ffi, synth
    from sparkfm_amd.distributed import (DataParallelSGD, HipDataParallelSGD, HostStagedComm, RcclComm, ThreadStagedComm,
                                         torch_stream_handle)

    config = args.config or ("C4" if world > 1 else "C3")
    cfg = synth.CONFIGS[config]
    # Data-parallel runs are ONE job whatever N: a GLOBAL mini-batch of DP_GLOBAL_BATCH_ROWS rows per step (the same SGD
    # trajectory on 2, 4 or 8 GPUs), every rank taking 1/N of it.  dp_world: the ranks of that job (an emulated run plays rank
    # 0 of the emulated count).
    dp_world = int(args.emulate_allreduce.split(":")[0]) if (use_dp and args.emulate_allreduce) else world
    rows = args.rows or min(cfg["rows"] // max(dp_world, 1), 5_000_000 if use_dp else 1_250_000)
    k, n1 = cfg["k"], cfg["features"]
    batch_rows = min(args.batch_rows or (max(DP_GLOBAL_BATCH_ROWS // max(dp_world, 1), 1) if use_dp else 250_000), rows)
    regs = (0.0, 1e-4, 1e-4)

    if args.hot_pages:
        _ffi.check(_ffi.load().fmhip_tune(12, args.hot_pages))
    for kv in args.tune:
        key, value = kv.split("=")
        _ffi.check(_ffi.load().fmhip_tune(int(key), int(value)))
    if group is None:       # (thread-ranks share the generator's thread count: set once, before they start)
        synth.set_threads(max(1, host_cores() // max(1, min(world, 8))) if world > 1 else host_cores())
    t0 = time.time()
    d = synth.make_config(config, rows=rows, row_begin=rank * rows)
    relabelled = bool(cfg.get("criteo")) and not args.no_relabel
    if relabelled:
        # hashed slots come in no particular order: relabel by frequency at load (a pure renaming of the features;
        # the counts are summed over the ranks so that every replica uses the same numbering)
        from sparkfm_amd import FeatureOrder
        d["col"] = FeatureOrder.from_counts(ctl.sum_counts(FeatureOrder.counts(d["col"], cfg["features"], device=local_rank)),
                                            device=local_rank).relabel(d["col"])
    t_gen = time.time() - t0
    t0 = time.time()
    ds = DataSet.from_arrays(d, name=config, batch_rows=batch_rows, device=local_rank).cache()
    t_load = time.time() - t0
    wide = n1 * k > (1 << 28)                      # too wide to stage fp64 parameters on the host: draw on the device
    stream = torch_stream_handle(local_rank) if exchange == "torch" else None
    if wide:
        fm = FMModel(n1 - 1, k, seed=cfg["seed"] + 1000, device=local_rank, stream=stream, init_on_device=True)
        w0 = w = v = None
    else:
        w0, w, v = synth.init_params(cfg["seed"] + 1000, n1, k)
        fm = FMModel(n1 - 1, k, device=local_rank, stream=stream)
        fm.w0, fm.w, fm.v = w0, w, v
    L = _ffi.load()
    hm, hd = fm.handle, ds.handle
    nb = ds.n_batches
    binfo = [ds.batch_info(b) for b in range(nb)]
    bnnz = [bi["nnz"] for bi in binfo]
    comm = dp = eng = None
    comm_note = selftest_note = None
    if exchange == "rccl":
        try:
            comm = (ThreadStagedComm(fm, rank, group) if group is not None else
                    (HostStagedComm(fm, rank, world) if args.transport == "host" else RcclComm(fm, rank, world)))
            fixed = None if args.upper_fractions == "auto" else (() if args.upper_fractions == "none" else
                                                                 tuple(float(x) for x in args.upper_fractions.split(",")))
            dp_mode = args.dp_exchange if args.dp_exchange != "auto" else ("touched" if cfg.get("criteo") else "dense")
            dp = HipDataParallelSGD(comm, eta=args.eta, reg0=regs[0], regw=regs[1], regv=regs[2], exchange=dp_mode,
                                    upper_fractions=fixed if fixed is not None else (0.05, 0.15, 0.3, 0.55))
            # before anything is trusted to it: known patterns through every collective kind, with the step's own calls (all ranks
            # get the same verdict; a failure takes the fallback below on every rank alike)
            comm.selftest()
            selftest_note = "fmhip_comm_selftest passed on %d ranks (all six collective kinds)" % world
            dp.plan(fm, ds)
            if args.emulate_allreduce:
                if world != 1:
                    raise SystemExit("--emulate-allreduce is for one-rank runs")
                emu_ranks, emu_busbw = args.emulate_allreduce.split(":")
                emu_ranks, emu_busbw = int(emu_ranks), float(emu_busbw)
                # ring all-reduce: 2(N-1)/N of the payload crosses each rank's links -> payload rate = busbw * N / (2(N-1));
                # the sharded update's reduce-scatter and all-gather are half of that each, and this rank plays rank 0 of N
                _ffi.check(L.fmhip_comm_emulate(comm.handle, emu_busbw * emu_ranks / (2.0 * (emu_ranks - 1))))
                _ffi.check(L.fmhip_comm_emulate_ranks(comm.handle, emu_ranks))
                _ffi.check(L.fmhip_comm_emulate_load(comm.handle, args.emulate_load))
        except Exception as ex:   # noqa: BLE001 — reported in the JSON line, never silent
            if group is not None:
                raise
            # every rank fails or succeeds together (communicator creation is collective); fall back to the
            # Python-orchestrated exchange over torch.distributed
            comm_note = "library-side RCCL exchange unavailable (%r): fell back to torch.distributed" % (ex,)
            sys.stderr.write("[bench] rank %d: %s\n" % (rank, comm_note))
            if comm is not None:          # created, then failed its self-test or the plan: not used again
                comm.close()
                comm = None
            selftest_note = None
            exchange = "torch"
            dist.destroy_process_group()
            init_process_group(dist, "nccl", device_id=torch.device("cuda", local_rank))
            ctl = TorchCtl(dist, torch, True)
            fm.close()
            fm = FMModel(n1 - 1, k, device=local_rank, stream=torch_stream_handle(local_rank), init_on_device=wide,
                         seed=cfg["seed"] + 1000)
            if not wide:
                fm.w0, fm.w, fm.v = w0, w, v
            hm = fm.handle
    if exchange == "torch":
        dp = DataParallelSGD(eta=args.eta, reg0=regs[0], regw=regs[1], regv=regs[2], always_reduce=True,
                             overlap=args.upper_fractions != "none")
        eng = dp.engine(fm, ds)

    def step(j):
        if exchange == "rccl":
            # a POSITION of the lock-step schedule, named by every rank alike (the touched-rows exchange picks that position's
            # planned union; the tuning passes revisit positions out of order)
            _ffi.check(L.fmhip_dp_step_at(hm, hd, j % nb, comm.handle, args.eta, regs[0], regs[1], regs[2]))
        elif exchange == "torch":
            dp.step(eng, j % nb)
        else:
            _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, args.eta, regs[0], regs[1], regs[2], None))

    def steps_run(j0, n):
        """Steps j0 .. j0 + n - 1.  The pipelined exchange takes them in ONE call (fmhip_dp_steps: it overlaps each step's last
        slice with the next position's forward and has to know that position); everything else steps one by one."""
        if exchange == "rccl" and dp.exchange == "pipelined" and n > 0:
            pos = np.ascontiguousarray([(j0 + j) % nb for j in range(n)], np.int64)
            _ffi.check(L.fmhip_dp_steps(hm, hd, _ffi.ptr(pos), n, comm.handle, args.eta, regs[0], regs[1], regs[2]))
        else:
            for j in range(j0, j0 + n):
                step(j)

    def sync():
        _ffi.check(L.fmhip_synchronize(hm))
        torch.cuda.synchronize()

    def barrier():
        ctl.barrier()

    steps_run(0, args.warmup)
    sync()
    barrier()
    tuning = None
    if exchange == "rccl" and args.upper_fractions == "auto" and (world > 1 or args.emulate_allreduce or dp.exchange == "touched"):
        # measure, don't guess: the best cut — and whether the sharded update pays — depends on the collectives' real
        # bandwidth on this node.  Candidates are timed for 8 steps each; the ranks agree through a max-reduce.
        tuning = []
        modes = ("dense", "sharded", "pipelined") if args.dp_exchange == "auto" and dp.exchange != "touched" else (dp.exchange,)
        cands = ((0.3,), (0.2,), (0.12, 0.4), (0.08, 0.25, 0.5), (0.05, 0.15, 0.3, 0.55), (0.04, 0.1, 0.2, 0.35, 0.6), ())
        if dp.exchange == "touched":
            # the compact gradient's slices overlap the backward as the dense one's do; with one rank there is nothing to hide
            # and every cut only costs launches — measured like everything else
            cands = ((), (0.3,), (0.12, 0.4), (0.05, 0.15, 0.3, 0.55))
        if args.transport != "rccl":
            cands = ((0.12, 0.4), ())        # a rehearsal of the flow: every step moves the whole gradient through the host
        for mode in modes:
            dp.set_exchange(mode)
            mode_cands = cands
            if mode == "pipelined" and args.transport == "rccl":
                # the pipelined schedule likes finer cuts (its wire idles only until the first, cheapest interval is walked)
                # ... or few launches with a small top: the top slice should take about as long on the wire as the next pass A,
                # the second interval is the cheap one that starts the wire, and every further cut costs a launch of the walk
                mode_cands = tuple(c_ for c_ in cands if len(c_) >= 2) + ((0.03, 0.07, 0.13, 0.22, 0.35, 0.6), (0.04, 0.1), (0.04, 0.1, 0.3),
                                                                           (0.05, 0.12, 0.35), (0.04, 0.09, 0.2, 0.5))
            for cand in mode_cands:
                dp.upper_fractions = cand
                dp.plan(fm, ds)
                step(0)
                sync()
                barrier()
                t0 = time.perf_counter()
                steps_run(0, 8)          # (8: a pipelined run's first forward pass and last slice are not overlapped with anything)
                sync()
                tt = ctl.allreduce([time.perf_counter() - t0], "max")
                tuning.append({"exchange": mode, "upper_fractions": list(cand), "cuts": list(dp.cuts), "ms_per_step": tt[0] / 8 * 1e3})
        best = min(tuning, key=lambda x: x["ms_per_step"])
        dp.set_exchange(best["exchange"])
        dp.upper_fractions = tuple(best["upper_fractions"])
        dp.plan(fm, ds)
        step(0)
        sync()
        barrier()
    if not os.environ.get("FMHIP_BENCH_NO_EVENTS"):
        # one kernel kind on every 4th step, rotating (a pair of event records costs ~8 us of stream time): with the
        # default 200 steps every kind is timed 12-13 times and the timed region is perturbed by < 1 %
        _ffi.check(L.fmhip_profile_begin_sampled(hm, 4 if args.steps >= 64 else (2 if args.steps >= 16 else 1)))
    sync()
    barrier()
    t0 = time.perf_counter()
    steps_run(args.warmup, args.steps)
    enqueue_s = time.perf_counter() - t0          # what the HOST needed to queue the steps (it must stay ahead of the GPU)
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(hm, C.byref(prof)))
    cprof = None
    if comm is not None:
        # the exchange's own timers (a dozen event records per step) run in a short pass of their own, not in the timed region
        _ffi.check(L.fmhip_comm_profile_begin(comm.handle))
        steps_run(0, 12)
        sync()
        cp = _ffi.CommProfile()
        _ffi.check(L.fmhip_comm_profile_end(comm.handle, C.byref(cp)))
        cprof = cp.as_dict()
        barrier()
    local_nnz = sum(bnnz[j % nb] for j in range(args.warmup, args.warmup + args.steps))
    elapsed = ctl.allreduce([elapsed], "max")[0]
    total_nnz = ctl.allreduce([float(local_nnz)], "sum")[0]
    st = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))

    # ---- sustained: the same steps back to back for >= 2 s (clock / power settled)
    sustained = None
    if not args.no_extra:
        barrier()
        sync()
        n_done, t0 = 0, time.perf_counter()
        while True:
            steps_run(n_done, 4 * nb)
            n_done += 4 * nb
            sync()
            dt = time.perf_counter() - t0
            if ctl.allreduce([1.0 if dt < 2.0 else 0.0], "max")[0] == 0.0:      # every rank takes the same number of steps
                break
        barrier()
        dt = time.perf_counter() - t0
        s_nnz = float(sum(bnnz[j % nb] for j in range(n_done)))
        s_nnz = ctl.allreduce([s_nnz], "sum")[0]
        sustained = {"seconds": dt, "steps": n_done, "value": s_nnz / dt, "unit": "nnz/s", "ms_per_step": dt / n_done * 1e3}

    # ---- do the replicas still agree?  Every rank has taken the same steps up to here (rank 0's legs below are its own): 4,160
    # parameter rows spread over every feature interval, and w0, must be the SAME BITS on all ranks — a collective that moved the
    # wrong elements, or an update that differed, shows here and not as a throughput number from models that have drifted apart
    replicas = None
    if use_dp:
        ids = np.unique(np.concatenate([np.arange(min(64, n1)), np.linspace(0, n1 - 1, 4096).astype(np.int64)])).astype(np.int32)
        rw, rv = fm.rows(ids)
        wts = np.cos(np.arange(rv.size, dtype=np.float64) * 0.7310585786)           # fixed weights: a permutation of rows shows too
        sums = [float(fm.w0), float(rw.sum()), float(rv.sum()), float(np.dot(rv.ravel(order="F"), wts))]
        hi_ = ctl.allreduce(sums, "max")
        lo_ = ctl.allreduce([-x for x in sums], "max")
        finite = all(np.isfinite(x) for x in sums)
        replicas = {"identical": bool(finite and all(a == -b for a, b in zip(hi_, lo_))), "rows_compared": int(len(ids)), "finite": bool(finite),
                    "note": "w0 and %d parameter rows spread over all feature intervals: plain and weighted fp64 sums, max == min over the ranks" % len(ids)}
        if not replicas["identical"]:
            sys.stderr.write("[bench] rank %d: REPLICAS DIFFER after the timed steps: max %r, -min %r\n" % (rank, hi_, lo_))

    # ---- the same shard and batch WITHOUT the exchange (what one GPU of the job does alone)
    no_exchange = one_gpu_plain = None
    if use_dp and rank == 0 and not args.no_extra:
        sync()
        for j in range(4):
            _ffi.check(L.fmhip_step_compute(hm, hd, j % nb))
            _ffi.check(L.fmhip_step_apply(hm, args.eta, *regs))
        sync()
        t0 = time.perf_counter()
        for j in range(args.steps):
            _ffi.check(L.fmhip_step_compute(hm, hd, j % nb))
            _ffi.check(L.fmhip_step_apply(hm, args.eta, *regs))
        sync()
        dt = time.perf_counter() - t0
        no_exchange = {"value": sum(bnnz[j % nb] for j in range(args.steps)) / dt, "unit": "nnz/s", "ms_per_step": dt / args.steps * 1e3,
                       "note": "rank 0 alone, same shard and batch, dense update, no all-reduce"}
        # ... and the plain one-GPU step (fmhip_sgd_step: the update merged into the fixup launch or rows-only, as N = 1 runs
        # it) on the same shard and batch: the denominator for this line's scaling, measured in the same process
        for j in range(4):
            _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, args.eta, *regs, None))
        sync()
        t0 = time.perf_counter()
        for j in range(args.steps):
            _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, args.eta, *regs, None))
        sync()
        dt = time.perf_counter() - t0
        one_gpu_plain = {"value": sum(bnnz[j % nb] for j in range(args.steps)) / dt, "unit": "nnz/s", "ms_per_step": dt / args.steps * 1e3,
                         "note": "rank 0 alone, same shard and batch, the plain one-GPU step (fmhip_sgd_step), while the other ranks wait"}
    barrier()

    # ---- the N = 1 line's own workload under the exchange: C3 on every GPU (weak scaling in the strict sense — the driver's
    # per-N values compare C3 at N = 1 with C4 at N > 1, two different widths; this leg is the like-for-like number)
    twin = None
    if exchange == "rccl" and dp.exchange != "touched" and config != "C3" and not args.no_extra:
        c3 = synth.CONFIGS["C3"]
        rows3 = 1_000_000 if not args.rows else min(1_000_000, max(args.rows, 1000))     # a rehearsal with --rows keeps the twin small too
        d3 = synth.make_config("C3", rows=rows3, row_begin=rank * rows3)
        ds3 = DataSet.from_arrays(d3, name="C3", batch_rows=min(250_000, rows3), device=local_rank).cache()
        fm3 = FMModel(c3["features"] - 1, c3["k"], seed=c3["seed"] + 1000, device=local_rank, init_on_device=True)
        nb3 = ds3.n_batches
        nnz3 = [ds3.batch_info(b)["nnz"] for b in range(nb3)]

        def step3(j):
            _ffi.check(L.fmhip_dp_step_at(fm3.handle, ds3.handle, j % nb3, comm.handle, args.eta, regs[0], regs[1], regs[2]))
        keep, keep_mode = dp.upper_fractions, dp.exchange
        best3 = None
        for mode3 in (("dense", "sharded") if args.dp_exchange == "auto" else (dp.exchange,)):
            dp.set_exchange(mode3)
            for cand in ((), (0.3,), (0.12, 0.4), (0.05, 0.15, 0.3, 0.55)):        # a 13.6 MB gradient wants fewer cuts than C4's 136 MB
                dp.upper_fractions = cand
                dp.plan(fm3, ds3)
                for j in range(3):
                    step3(j)
                _ffi.check(L.fmhip_synchronize(fm3.handle))
                barrier()
                t0 = time.perf_counter()
                for j in range(8):
                    step3(j)
                _ffi.check(L.fmhip_synchronize(fm3.handle))
                tt = ctl.allreduce([time.perf_counter() - t0], "max")
                if best3 is None or tt[0] < best3[0]:
                    best3 = (tt[0], cand, mode3)
        dp.set_exchange(best3[2])
        dp.upper_fractions = best3[1]
        dp.plan(fm3, ds3)
        for j in range(4):
            step3(j)
        _ffi.check(L.fmhip_synchronize(fm3.handle))
        barrier()
        t0 = time.perf_counter()
        for j in range(args.steps):
            step3(j)
        _ffi.check(L.fmhip_synchronize(fm3.handle))
        barrier()
        tm = ctl.allreduce([time.perf_counter() - t0], "max")
        t3 = ctl.allreduce([float(sum(nnz3[j % nb3] for j in range(args.steps)))], "sum")
        twin = {"workload": "C3 on every GPU: %d rows x 100000 features per GPU, k=32, batch %d rows per GPU — the N = 1 line's workload" % (rows3, min(250_000, rows3)),
                "value": t3[0] / tm[0], "unit": "nnz/s", "ms_per_step": tm[0] / args.steps * 1e3,
                "allreduce_bytes_per_step": 4 * (32 + (c3["features"] + 31) // 32 * 32 * 34), "cuts": list(dp.cuts), "exchange": dp.exchange}
        ds3.unpersist()
        fm3.close(discard=True)
        dp.set_exchange(keep_mode)
        dp.upper_fractions = keep
        dp.plan(fm, ds)
        barrier()

    if rank == 0:
        ab = alg_bytes(k)
        kp = 32
        while kp < k:
            kp *= 2
        packed = k < kp
        pmc = committed_pmc(config, k, batch_rows)
        live = None
        if exchange == "rccl" and dp.exchange != "touched" and config == "C4" and not args.no_pmc and not args.tune and not args.hot_pages:
            # N > 1 (or --force-dp): the counters of THIS rank's workload under the step this line timed — the same shard size,
            # batch, cuts and exchange mode through the library's own data-parallel step with a one-rank communicator whose
            # collectives are the identity (tools/pmc_leg.py c4) — measured now, on rank 0's GPU, while the other ranks wait
            fr = ",".join(str(f) for f in dp.upper_fractions) or "none"
            live = live_pmc([os.path.join(ROOT, "tools", "pmc_leg.py"), "c4", "--rows", str(min(rows, 2 * batch_rows, max(batch_rows, 2_500_000))), "--batch-rows", str(batch_rows),
                             "--upper-fractions", fr, "--dp-exchange", dp.exchange, "--steps", "8", "--warmup", "4"], per_step=(4, 8))
            for kn, e in (live or {}).items():
                pmc.setdefault(kn, {}).update(e, traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation on "
                                                                "rank 0's GPU (tools/pmc_leg.py c4: the same shard size, batch, cuts and mode)")
        if world == 1 and not use_dp and not args.no_pmc and config in ("C2", "C3") and not args.tune and not args.hot_pages:
            # the counters of THIS workload, measured now: rocprofv3 child processes over tools/pmc_leg.py (same config,
            # rows and batch; the committed profile is the fallback when rocprofv3 is not available)
            live = live_pmc([os.path.join(ROOT, "tools", "pmc_leg.py"), config.lower(), "--rows", str(rows), "--batch-rows", str(batch_rows)])
            for kn, e in (live or {}).items():
                pmc.setdefault(kn, {}).update(e, traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation "
                                                                "(tools/pmc_leg.py: the same workload)")
        bi = binfo[0]
        lay = ds.layout()
        hot = len(lay["hot_ids"]) > 0
        n_cols = bi["n_columns"]
        nnz0, rows0 = bi["nnz"], bi["rows"]
        nnz0_sparse = int(round(nnz0 * lay["nnz_sparse"] / max(int(d["row_ptr"][-1]), 1)))   # batch 0's share of the sparse streams
        nnz0_sparse_b = int(round(nnz0 * lay["nnz_sparse_backward"] / max(int(d["row_ptr"][-1]), 1)))   # ... of the transposes
        dense_apply = (use_dp and not (exchange == "rccl" and dp.exchange == "touched")) or n_cols * 2 > n1
        req = requested_bytes(kp, rows0, nnz0, nnz0_sparse, n_cols, hot, n_cols, dense_apply, n1, packed, nnz0_sparse_b, lay["hot_pages"])
        table_bytes = {"forward": n1 * kp * 4, "backward": rows0 * kp * 4}
        kern = kernel_table(prof, k, kp, req, pmc, table_bytes)
        # the dominant kernel = the longest launch (not the largest sampled total: kinds are sampled in rotation)
        dom = max(("forward", "backward"), key=lambda n: kern.get(n, {}).get("avg_ms", 0.0))
        pd = prof.as_dict()
        value = total_nnz / elapsed
        step_ms = elapsed / args.steps * 1e3
        # fraction of the step's time that the kernels' own ceilings account for (<= 1 when no kernel beats its ceiling)
        explained_ms = sum(e["requested_bytes_per_launch"] / (e["ceiling"]["GBps"] * 1e9) * 1e3 for e in kern.values() if e.get("ceiling"))
        out = {
            "metric": "nnz_per_sec_fm_sgd_training", "value": value, "unit": "nnz/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d rows x %d features per GPU, k=%d, %s, fp32 mini-batch SGD" %
                                   (config, rows, n1, k, ("39 hashed Criteo-shaped fields" + (", ids relabelled by frequency at load" if relabelled else ""))
                                    if cfg.get("criteo") else
                                    "nnz/row U{%d..%d}, ids Zipf(%.2f)" % (cfg["nnz_lo"], cfg["nnz_hi"], cfg["zipf_s"])),
                       "rows_per_gpu": rows, "features": n1, "k": k, "batch_rows_per_gpu": batch_rows,
                       "global_batch": batch_rows * dp_world,
                       "batches_per_gpu": nb, "nnz_per_gpu": int(d["row_ptr"][-1]), "eta": args.eta, "regs": regs,
                       "dense_hot_block": {"pages": lay["hot_pages"], "features_forward_and_backward": len(lay["hot_ids"]),
                                           "features_backward": len(lay["hot_ids_all"]),
                                           "share_of_nonzeros_left_to_the_forward": lay["nnz_sparse"] / max(int(d["row_ptr"][-1]), 1),
                                           "share_of_nonzeros_left_to_the_backward": lay["nnz_sparse_backward"] / max(int(d["row_ptr"][-1]), 1)},
                       "backward_band_plan": {"ranges": lay["ranges"], "planned": lay["planned_ranges"], "band_affine": lay["band_affine_ranges"],
                                              "share_band_affine": lay["band_affine_ranges"] / max(lay["ranges"], 1),
                                              "note": "ranges of long columns walked on the XCD that owns their row band (fmhip_tune key 4)"},
                       "parallelism": "dp%d" % world, "exchange": exchange,
                       "transport": ("host-staged gloo over fmhip_comm_create_external, all ranks on GPU 0 (a rehearsal of the N-rank flow, "
                                     "not a measurement)" if args.transport == "host" and use_dp else
                                     ("host-staged between the ranks-as-threads of ONE process over fmhip_comm_create_external, all on GPU 0 (a "
                                      "rehearsal of the N-rank flow, not a measurement)" if args.transport == "threads" else ("RCCL" if use_dp else "none"))),
                       "allreduce": ("inside the library, touched rows only" if exchange == "rccl" and dp.exchange == "touched" else
                                     ("inside the library, %s, overlapped with the feature-chunked backward, cuts at features %s" %
                                      ("reduce-scatter -> sharded update -> all-gather" if dp.exchange == "sharded" else "all-reduce, every rank updates every row", dp.cuts))
                                     if exchange == "rccl" and dp.cuts else
                                     ("inside the library, one %s per step" % ("reduce-scatter + all-gather" if dp.exchange == "sharded" else "all-reduce") if exchange == "rccl" else
                                      ("torch.distributed, orchestrated from Python" if exchange == "torch" else "none")))},
            "roofline": roofline_block(kern, dom, ab, pd, pmc, step_ms, live is not None),
            "step_roofline": {"alg_bytes_per_nnz": ab["step"], "alg_GBps": value / world * ab["step"] / 1e9,
                              "requested_bytes_per_step": sum(e.get("requested_bytes_per_launch", 0) for e in kern.values()),
                              "time_at_ceilings_ms": explained_ms, "frac": explained_ms / step_ms,
                              "note": "frac = sum over kernels of (requested bytes / that kernel's ceiling) / measured step time"},
            "kernels": kern,
            "train": {"last_batch_mse": st.sse / max(st.rows, 1), "nonfinite": st.nonfinite},
            "setup_s": {"generate": t_gen, "load_transpose_h2d": t_load},
            "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
        }
        if sustained:
            out["sustained"] = sustained
        if use_dp:
            gf = C.c_int64()
            _ffi.check(L.fmhip_grad_floats(hm, C.byref(gf)))
            payload = int(gf.value) * 4
            xc = {"nranks": world, "allreduce_bytes_per_step": payload, "backend": exchange, "transport": args.transport,
                  "mode": dp.exchange if exchange == "rccl" else "dense"}
            if exchange == "rccl" and dp.exchange == "pipelined":
                xc["mode_note"] = ("pipelined (FMHIP_EXCHANGE_PIPELINED): the dense exchange with consecutive steps overlapped — the coldest feature "
                                   "interval is walked and sent last, and while its slice travels the next position's forward runs over every feature "
                                   "below the top cut (a two-pass forward over rows partitioned at that cut; fmhip_dp_steps hands the library the whole "
                                   "run of positions); same sums and update as the dense mode, the forward's fp32 sums in another order")
            if exchange == "rccl" and dp.exchange == "touched":
                info = dp.exchange_info()
                xc["mode_note"] = ("touched rows (fmhip_dp_exchange): the union of the rows every position's batches touch is planned ONCE "
                                   "(fmhip_dp_plan: all-gather of ids, sort, unique); a step writes its gradient into a compact buffer with one "
                                   "row per union feature, all-reduces it in feature-interval slices under the backward, and applies the "
                                   "rows-only update with lazy weight decay — no id exchange, sort or read-back in the step")
                xc["dense_gradient_bytes"] = payload
                xc["id_slots_per_rank"] = info["id_slots_per_rank"]
                xc["mean_union_rows"] = info["mean_union_rows"]
                xc["allreduce_bytes_per_step"] = int((32 + info["mean_union_rows"] * (kp + 2)) * 4)
                xc["allgather_bytes_per_step"] = int(info["id_slots_per_rank"] * world * 4)
            if cprof and cprof["steps"]:
                xc["exposed_comm_ms"] = cprof["exposed_ms"] / cprof["steps"]
                xc["comm_busy_ms"] = cprof["comm_ms"] / cprof["steps"]
                # ring all-reduce moves 2(N-1)/N of the payload per rank
                busy = max(cprof["comm_ms"] / cprof["steps"], 1e-9)
                xc["alg_GBps"] = payload / busy / 1e6
                xc["bus_GBps"] = payload * (2.0 * (world - 1) / max(world, 1)) / busy / 1e6
            if twin:
                xc["c3_on_every_gpu"] = twin
            if no_exchange:
                xc["per_gpu_without_exchange"] = no_exchange
                xc["efficiency_vs_no_exchange"] = value / (world * no_exchange["value"])
            if one_gpu_plain:
                # the like-for-like scaling of THIS line: the job's throughput over what one GPU does alone on the same workload
                # (C4's shard and batch, the plain step) — the driver's N = 1 line is C3, another width
                xc["%s_one_gpu" % config.lower()] = one_gpu_plain
                xc["scaling_vs_%s_one_gpu" % config.lower()] = value / one_gpu_plain["value"]
            if comm_note:
                xc["note"] = comm_note
            if selftest_note:
                xc["selftest"] = selftest_note
            if replicas:
                xc["replicas"] = replicas
            if tuning:
                xc["cut_tuning"] = tuning
            if args.emulate_allreduce:
                xc["emulated"] = "ring all-reduce over %s GPUs at bus bandwidth %s GB/s, as a delay on the comm stream (one real rank)" % tuple(args.emulate_allreduce.split(":"))
                if args.emulate_load:
                    xc["emulated"] += "; the delay is spent by %d workgroups streaming the payload through HBM (read + write, twice per all-reduce)" % args.emulate_load
            out["exchange"] = xc
        if not args.no_cpu_baseline and not wide:
            # rank 0's host cores, on rank 0's shard of the line's own workload, after every timed region (at N > 1 the other
            # ranks wait at the closing barrier); the ratio at N > 1 is the JOB's throughput over that one-host baseline
            out["cpu_baseline"] = cpu_baseline(d, k, n1, batch_rows, args.eta, regs, w0, w, v, args.cpu_budget)
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_extra and not use_dp:
            scoring = None
            try:
                # the scoring calls of the path (FMModel.predict / computeRMSE) on the line's own
                # dataset: fmhip_rmse = the forward over every batch + the statistics, nothing leaves the device but one double
                r_ = C.c_double(0.0)
                _ffi.check(L.fmhip_rmse(hm, hd, C.byref(r_), None))
                _ffi.check(L.fmhip_synchronize(hm))
                n_pass, t0s = 0, time.perf_counter()
                while n_pass < 5 or time.perf_counter() - t0s < 0.5:
                    _ffi.check(L.fmhip_rmse(hm, hd, C.byref(r_), None))
                    n_pass += 1
                dts = time.perf_counter() - t0s
                nnz_all = int(d["row_ptr"][-1])
                scoring = {"what": "fmhip_rmse over the whole dataset (%d rows, %d nonzeros): forward + statistics per batch, %d passes" % (rows, nnz_all, n_pass),
                           "value": nnz_all * n_pass / dts, "unit": "nnz/s", "ms_per_pass": dts / n_pass * 1e3, "rmse": r_.value}
            except Exception as ex:   # noqa: BLE001
                scoring = {"error": repr(ex)}
            ds.unpersist()
            fm.close(discard=True)
            del d
            extra = {"scoring": scoring}
            try:
                extra["hbm_resident"] = hbm_resident_leg(local_rank, with_pmc=not args.no_pmc)
            except Exception as ex:   # noqa: BLE001
                extra["hbm_resident"] = {"error": repr(ex)}
            try:
                extra["c4_one_gpu"] = c4_one_gpu_leg(local_rank, args.eta, regs)
            except Exception as ex:   # noqa: BLE001
                extra["c4_one_gpu"] = {"error": repr(ex)}
            extra["als_c1"] = als_c1(local_rank)
            try:
                extra["als_long_columns"] = als_long(local_rank)
            except Exception as ex:   # noqa: BLE001
                extra["als_long_columns"] = {"error": repr(ex)}
            try:
                extra["als_fields"] = als_fields(local_rank)
            except Exception as ex:   # noqa: BLE001
                extra["als_fields"] = {"error": repr(ex)}
            out["extra"] = extra
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dp:
        ctl.barrier()
        if comm is not None:
            comm.close()


if __name__ == "__main__":
    main()
