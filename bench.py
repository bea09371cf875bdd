#!/usr/bin/env python3
"""bench.py — nnz/sec of FM mini-batch SGD training on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--batch-rows B]

A "step" is one mini-batch SGD step (forward + backward + update) over one batch of the
synthetic workload, all inputs resident in HBM before the timed region.  With N > 1 it is
launched one process per GPU (torch.distributed.run); every rank owns its own 1M-row shard
of the same virtual dataset (weak scaling) and the packed gradient is all-reduced over
RCCL/xGMI every step.  Rank 0 prints ONE JSON line.

The `roofline` object is measured live with HIP events (recorded by the library on the
stream its kernels run on) over the timed steps; `cpu_baseline` times the fp64 CPU oracle
(a port — SparkFM itself needs a JVM, absent here) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); measured streaming copy ~6.3e12


def alg_bytes(k):
    """SURVEY.md §8(d) algorithmic bytes per stored nonzero, split by kernel (fp32/int32):
    forward  = col 4 + val 4 + V-row read 4k + w read 4      = 4k + 12
    backward = V-grad row add 4k + w-grad add 4               = 4k + 4
    whole step B_alg(k) = 8k + 16 (plus 16 B/row and 12(n+1)(k+1) B/step for the dense update)."""
    return {"forward": 4 * k + 12, "backward": 4 * k + 4, "step": 8 * k + 16}


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
    import oracle
    from oracle import capi
    L = capi.lib()
    threads = host_cores()
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
            "cpu_oracle_s_per_epoch": cpu_s, "nnz": nnz,
            "note": "Gauss-Seidel over features: a fidelity path (one persistent workgroup), not a throughput path"}


def pmc_traffic(config, k, batch_rows, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json), or None when no pass exists for this configuration."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entries = json.load(f)["entries"]
    except (OSError, ValueError, KeyError):
        return None
    for e in entries:
        if (e["config"], e["k"], e["batch_rows"], e["kernel"]) == (config, k, batch_rows, kernel):
            return e["traffic_bytes"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the config's row count, capped at 1.25M)")
    ap.add_argument("--batch-rows", type=int, default=250000, help="mini-batch rows per GPU")
    ap.add_argument("--eta", type=float, default=0.02)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dp", action="store_true",
                    help="self-test: take the data-parallel path (RCCL all-reduce included) even with one rank")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    args = ap.parse_args()
    # stdout carries exactly one JSON line: libraries that print banners to fd 1 (RCCL prints its
    # version block there at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port 29511 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from sparkfm_amd import DataSet, FMModel, _ffi, synth
    from sparkfm_amd.distributed import DataParallelSGD, torch_stream_handle

    cfg = synth.CONFIGS[args.config]
    rows = args.rows or min(cfg["rows"], 1_250_000)
    k, n1 = cfg["k"], cfg["features"]
    batch_rows = min(args.batch_rows, rows)
    regs = (0.0, 1e-4, 1e-4)

    t0 = time.time()
    d = synth.make_config(args.config, rows=rows, row_begin=rank * rows)
    t_gen = time.time() - t0
    w0, w, v = synth.init_params(cfg["seed"] + 1000, n1, k)
    t0 = time.time()
    ds = DataSet.from_arrays(d, name=args.config, batch_rows=batch_rows, device=local_rank).cache()
    t_load = time.time() - t0
    stream = torch_stream_handle(local_rank) if use_dp else None
    fm = FMModel(n1 - 1, k, device=local_rank, stream=stream)
    fm.w0, fm.w, fm.v = w0, w, v
    L = _ffi.load()
    hm, hd = fm.handle, ds.handle
    nb = ds.n_batches
    bnnz = [ds.batch_info(b)["nnz"] for b in range(nb)]
    dp = DataParallelSGD(eta=args.eta, reg0=regs[0], regw=regs[1], regv=regs[2], always_reduce=True) if use_dp else None
    eng = dp.engine(fm, ds) if dp else None

    def step(j):
        if dp:
            dp.step(eng, j % nb)
        else:
            _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, args.eta, regs[0], regs[1], regs[2], None))

    def sync():
        _ffi.check(L.fmhip_synchronize(hm))
        torch.cuda.synchronize()

    overlap_used = bool(dp and dp.overlap)
    try:
        for j in range(args.warmup):
            step(j)
        sync()
    except Exception as ex:   # noqa: BLE001 — only the overlapped exchange is retried, loudly
        if not (dp and dp.overlap):
            raise
        sys.stderr.write("[bench] rank %d: overlapped all-reduce path failed (%r); retrying with the plain "
                         "one-collective-per-step path\n" % (rank, ex))
        dp.overlap = False
        overlap_used = False
        eng.grad.zero_()
        fm.w0, fm.w, fm.v = w0, w, v
        hm = fm.handle
        _ffi.check(L.fmhip_grad_bind(hm, C.c_void_p(eng.grad.data_ptr())))
        for j in range(args.warmup):
            step(j)
        sync()
    if use_dp:
        dist.barrier()
    if not os.environ.get("FMHIP_BENCH_NO_EVENTS"):
        # one kernel kind per step, rotating: the event records barely perturb the timed region
        _ffi.check(L.fmhip_profile_begin_rotating(hm))
    sync()
    t0 = time.perf_counter()
    for j in range(args.warmup, args.warmup + args.steps):
        step(j)
    sync()
    if use_dp:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = _ffi.Profile()
    _ffi.check(L.fmhip_profile_end(hm, C.byref(prof)))
    local_nnz = sum(bnnz[j % nb] for j in range(args.warmup, args.warmup + args.steps))
    if use_dp:
        t = torch.tensor([elapsed, float(local_nnz)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_nnz = float(tmax[0]), float(t[1])
    else:
        total_nnz = float(local_nnz)
    st = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))

    if rank == 0:
        ab = alg_bytes(k)
        pd = prof.as_dict()
        kern = {}
        for name, p in pd.items():
            if p["launches"]:
                avg_ms = p["ms"] / p["launches"]
                ent = {"avg_ms": avg_ms, "launches": p["launches"], "share": p["ms"] / max(sum(x["ms"] for x in pd.values()), 1e-12)}
                if name in ab:
                    ent["alg_bytes_per_nnz"] = ab[name]
                    ent["alg_GBps"] = (p["nnz"] / p["launches"]) * ab[name] / (avg_ms * 1e-3) / 1e9
                kern[name] = ent
        dom = max(("forward", "backward"), key=lambda n: pd[n]["ms"])
        achieved = kern[dom]["alg_GBps"] if dom in kern else float("nan")
        value = total_nnz / elapsed
        out = {
            "metric": "nnz_per_sec_fm_sgd_training", "value": value, "unit": "nnz/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d rows x %d features per GPU, k=%d, nnz/row U{%d..%d}, ids Zipf(%.2f), "
                                   "fp32 mini-batch SGD" % (args.config, rows, n1, k, cfg["nnz_lo"], cfg["nnz_hi"], cfg["zipf_s"]),
                       "rows_per_gpu": rows, "features": n1, "k": k, "batch_rows_per_gpu": batch_rows,
                       "batches_per_gpu": nb, "nnz_per_gpu": int(d["row_ptr"][-1]), "eta": args.eta,
                       "parallelism": "dp%d" % world,
                       "allreduce": ("overlapped with the feature-chunked backward" if overlap_used else
                                     ("one all-reduce per step" if use_dp else "none"))},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved * 1e9 / HBM_PEAK,
                         "traffic": pmc_traffic(args.config, k, batch_rows, "k_" + dom),
                         "alg_bytes_per_nnz": ab[dom], "nnz_per_launch": pd[dom]["nnz"] / max(pd[dom]["launches"], 1),
                         "avg_launch_ms": kern[dom]["avg_ms"] if dom in kern else None},
            "step_roofline": {"alg_bytes_per_nnz": ab["step"], "achieved_GBps": value / world * ab["step"] / 1e9,
                              "frac_of_8TBps": value / world * ab["step"] / HBM_PEAK},
            "kernels": kern,
            "train": {"last_batch_mse": st.sse / max(st.rows, 1), "nonfinite": st.nonfinite},
            "setup_s": {"generate": t_gen, "load_transpose_h2d": t_load},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, k, n1, batch_rows, args.eta, regs, w0, w, v, args.cpu_budget)
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
            out["als_c1"] = als_c1(local_rank)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
