#!/usr/bin/env python3
"""bench.py — nnz/sec of FM mini-batch SGD training on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C4|C5] [--batch-rows B] [--time-budget S]

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
          before it has touched the GPU — and forwards rank 0's JSON lines.  `--transport threads` runs the N
          ranks as THREADS of this process on GPU 0 (a rehearsal of the whole N-rank flow on a one-GPU box:
          a world of 8 fits neither RCCL, one rank per device, nor the test pool's 6 processes per card).

The record cannot be lost (benchkit/emit.py): rank 0 writes the WHOLE JSON line as soon as the headline exists — metric,
value, config, `roofline` (HIP-event kernel times; counter bytes from the committed profile until this run's own passes have
finished), `exchange.*` at N > 1 — and re-writes it, enriched, after every further leg (`cpu_baseline`, the live rocprofv3
counter passes, `sustained`, the twin and one-GPU legs, `extra.*`), each started only if `--time-budget` still covers its
estimate.  THE LAST COMPLETE LINE IS THE RECORD; a run killed inside an optional leg has left a valid one behind.

The timed region carries no event records: clocks are settled first by >= `--settle` seconds of untimed steps, then W
counted warm-up steps, then exactly K steps between barriers; the per-kernel HIP-event times (`roofline.avg_launch_ms`,
`kernels`) come from a short pass of the SAME steps right after it (events recorded by the library on the stream its
kernels run on).  `cpu_baseline` times the fp64 CPU oracle (a port — SparkFM itself needs a JVM, absent here) on a bounded
sample of the same workload; `extra.hbm_resident` is a Criteo-width model (V = 8.6 GB, k=64: the only configuration whose
tables do not live in L2 / Infinity Cache) with weight decay; its counter fraction is copied into `roofline.hbm_resident`.

Parts: this file = the argument parser and one rank's flow (setup, settle, measure, legs); benchkit/roofline.py (byte accounting,
ceilings, the roofline block), benchkit/counters.py (rocprofv3 --pmc child passes), benchkit/record.py (the line's `config` and
`exchange` blocks), benchkit/legs.py (CPU baseline and the N = 1 legs), benchkit/dp_legs.py (the N > 1 cut / mode sweep, one-GPU
denominators, C3 twin), benchkit/ranks.py (multi-rank launch and control plane), benchkit/emit.py (the line and its budget).
"""
import argparse
import ctypes as C
import os
import sys
import time
import types

T_PROCESS_START = time.monotonic()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchkit import dp_legs, record  # noqa: E402
from benchkit.counters import PMC_STATE, STEP_KERNELS, committed_pmc, live_pmc, pmc_pass  # noqa: E402,F401  (re-exported: tools/, tests/)
from benchkit.emit import Budget, Emitter, last_record  # noqa: E402,F401
from benchkit.legs import (DP_GLOBAL_BATCH_ROWS, als_c1, als_fields, als_long, c4_one_gpu_leg, cpu_baseline, hbm_resident_leg,  # noqa: E402,F401
                           host_cores, scoring_leg)
from benchkit.ranks import NoCtl, ThreadCtl, TorchCtl, init_process_group, spawn_ranks  # noqa: E402,F401
from benchkit.roofline import (CEIL, HBM_PEAK, L2_BYTES_PER_XCD, MALL_BYTES, alg_bytes, annotate_roofline, compulsory_hbm_bytes,  # noqa: E402,F401
                               gather_ceiling, kernel_table, requested_bytes, roofline_block)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=["C1", "C2", "C3", "C4", "C5"],
                    help="default: C3 on one GPU (the metric's configuration), C4 on several (BASELINE's 8-GPU config)")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the config's rows / N, capped at 5M — two global batches; 1.25M on one GPU)")
    ap.add_argument("--batch-rows", type=int, default=0, help="mini-batch rows per GPU (default 250000; data-parallel: 5M / N, the global batch is fixed)")
    ap.add_argument("--eta", type=float, default=0.02)
    ap.add_argument("--time-budget", type=float, default=240.0,
                    help="wall-clock seconds for the WHOLE run, counted from process start: the headline line is written as soon as it exists; "
                         "every further leg starts only if the time left covers its estimate (skipped legs are named in `legs.skipped`)")
    ap.add_argument("--settle", type=float, default=0.4,
                    help="seconds of untimed steps BEFORE the counted warm-up (clock / power settle: a 5 ms warm-up does not)")
    ap.add_argument("--tune-budget", type=float, default=45.0,
                    help="N > 1: wall-clock seconds for the cut / exchange-mode sweep (candidates in order of likely merit; all ranks stop together)")
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
                         "1/N share, the updated rows all-gathered; pipelined = dense with consecutive steps overlapped; touched = only "
                         "the rows some rank touched; auto = touched for C5 (an 8.9 GB gradient), otherwise the three dense modes are "
                         "timed during warm-up and the fastest is kept")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host", "threads"],
                    help="rccl: one rank per GPU, the library's RCCL communicator; host: ALL ranks on GPU 0, the library's step over "
                         "fmhip_comm_create_external with every collective staged through the host and summed by gloo — the same "
                         "schedule, plan and update, for boxes with fewer GPUs than ranks (a rehearsal, not a measurement); threads: the "
                         "same with the ranks as THREADS of this one process (ThreadStagedComm) — a world of 8 on a one-GPU box, where "
                         "RCCL wants one device per rank and the test pool admits 6 processes per card")
    ap.add_argument("--upper-fractions", default="auto",
                    help="cuts of the backward for the overlapped exchange: comma-separated ascending shares of the nonzeros at or "
                         "above each cut (e.g. 0.3 or 0.12,0.4), 'none' = one all-reduce after the whole backward, 'auto' = "
                         "time candidates during warm-up and keep the fastest (all ranks agree through a max-reduce)")
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
                    help="A/B: pages of the dense hot block (FMHIP_TUNE_HOT_PAGES; 1 = the two-sided page only, default = library's)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B: fmhip_tune(KEY, VALUE) before anything is built (repeatable); KEY = a number or a name of "
                         "enum fmhip_tune_key without its prefix (include/fmhip_experimental.h), e.g. flat_address=1")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work the cpu_baseline sample is sized for")
    ap.add_argument("--log-dir", default=os.environ.get("FMHIP_BENCH_LOG_DIR", ""),
                    help="N > 1: every rank appends its progress to <log-dir>/rank<r>.progress (and to stderr); ranks spawned by "
                         "this file also have their stderr there")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.transport == "threads":
        if args.exchange != "rccl":
            raise SystemExit("--transport threads runs the library's own step (--exchange rccl)")
        # stdout carries JSON lines only (see below)
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
        spawn_ranks(args, sys.argv[1:], args.log_dir or None)
    # stdout carries JSON lines only: libraries that print banners to fd 1 (RCCL prints its
    # version block there at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    node_env(world)

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


def node_env(world):
    """Environment a multi-process GPU run of ONE node needs, set before anything initialises the GPU (a launcher may not have):
    dmabuf IPC (the pool's host driver supports nothing else: without it RCCL fails in hipIpcGetMemHandle), and — when this host's
    name does not resolve, which gloo's default device needs — the loopback interface for the control plane."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost") and "GLOO_SOCKET_IFNAME" not in os.environ:
        import socket
        try:
            socket.gethostbyname(socket.gethostname())
        except OSError:
            os.environ["GLOO_SOCKET_IFNAME"] = "lo"


def tune_key(name):
    """--tune's KEY: a number, or a name of enum fmhip_tune_key (without the FMHIP_TUNE_ prefix, any case)."""
    from sparkfm_amd import _ffi
    return int(name) if name.lstrip("-").isdigit() else _ffi.TUNE[name.upper()]


def test_stall(leg):
    """Test hook (tests/: a run killed inside an optional leg must have left a valid line): FMHIP_BENCH_TEST_STALL=<leg> makes
    the named leg hang instead of running."""
    if os.environ.get("FMHIP_BENCH_TEST_STALL") == leg:
        sys.stderr.write("[bench] FMHIP_BENCH_TEST_STALL: stalling in leg %r\n" % leg)
        sys.stderr.flush()
        time.sleep(3600)


def run_rank(args, rank, world, local_rank, ctl, json_fd, torch, group=None):
    """One rank of the bench (a process, or a thread under --transport threads)."""
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dp = world > 1 or args.force_dp
    exchange = args.exchange if use_dp else "none"
    from sparkfm_amd import DataSet, FMModel, _ffi, synth
    from sparkfm_amd.distributed import (DataParallelSGD, HipDataParallelSGD, HostStagedComm, RcclComm, ThreadStagedComm,
                                         torch_stream_handle)

    budget = Budget(args.time_budget, T_PROCESS_START)
    emitter = Emitter(json_fd, budget) if rank == 0 else None
    progress_file = None
    if args.log_dir and use_dp:
        try:
            os.makedirs(args.log_dir, exist_ok=True)
            progress_file = open(os.path.join(args.log_dir, "rank%d.progress" % rank), "a", buffering=1)
        except OSError:
            progress_file = None

    def log(msg):
        line = "[bench %7.1fs] rank %d: %s\n" % (budget.elapsed(), rank, msg)
        if use_dp or rank == 0:
            sys.stderr.write(line)
            sys.stderr.flush()
        if progress_file:
            progress_file.write(line)

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
    L = _ffi.load()

    if args.hot_pages:
        _ffi.check(L.fmhip_tune(_ffi.TUNE["HOT_PAGES"], args.hot_pages))
    for kv in args.tune:
        key, value = kv.split("=")
        _ffi.check(L.fmhip_tune(tune_key(key), int(value)))
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
    log("dataset ready: %s, %d rows, batch %d (generate %.1f s, load %.1f s)" % (config, rows, batch_rows, t_gen, t_load))
    wide = n1 * k > (1 << 28)                      # too wide to stage fp64 parameters on the host: draw on the device
    stream = torch_stream_handle(local_rank) if exchange == "torch" else None
    if wide:
        fm = FMModel(n1 - 1, k, seed=cfg["seed"] + 1000, device=local_rank, stream=stream, init_on_device=True)
        w0 = w = v = None
    else:
        w0, w, v = synth.init_params(cfg["seed"] + 1000, n1, k)
        fm = FMModel(n1 - 1, k, device=local_rank, stream=stream)
        fm.w0, fm.w, fm.v = w0, w, v
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
            log("communicator ready (%s), self-test passed, plan made" % args.transport)
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
            log(comm_note)
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

    def agreed(flag):
        """True on every rank iff `flag` is true on ALL of them (rank 0's budget decisions reach the others this way)."""
        return ctl.allreduce([0.0 if flag else 1.0], "max")[0] == 0.0

    def settle(seconds):
        """Untimed steps for >= `seconds` (every rank the same count: the ranks agree after every chunk)."""
        if seconds <= 0:
            return 0
        n_done, t_s = 0, time.perf_counter()
        chunk = max(nb, 4)
        while True:
            steps_run(n_done, chunk)
            n_done += chunk
            sync()
            if ctl.allreduce([1.0 if time.perf_counter() - t_s < seconds else 0.0], "max")[0] == 0.0:
                return n_done
            chunk = min(chunk * 2, 256)

    # ---- settle (clocks, allocations, first-launch costs), then the counted warm-up
    settled = settle(args.settle)
    steps_run(0, args.warmup)
    sync()
    barrier()

    elapsed = total_nnz = enqueue_s = value = step_ms = 0.0
    st = prof = cprof = replicas = None
    kernel_pass_steps = 0

    # ---- the timed region: exactly K steps between barriers, no event records inside; then the per-kernel pass, the
    # exchange's own timers and the replica check.  A function: at N > 1 it runs TWICE — once with the default plan BEFORE the
    # cut / mode sweep, so that a headline line exists whatever the sweep then does on a node nobody has seen (a mode that hangs
    # over real RCCL must not cost the record), and once with the plan the sweep chose.
    def measure():
        nonlocal elapsed, total_nnz, enqueue_s, st, prof, kernel_pass_steps, cprof, replicas, value, step_ms
        sync()
        barrier()
        t0 = time.perf_counter()
        steps_run(args.warmup, args.steps)
        enqueue_s = time.perf_counter() - t0          # what the HOST needed to queue the steps (it must stay ahead of the GPU)
        sync()
        barrier()
        elapsed = time.perf_counter() - t0
        local_nnz = sum(bnnz[j % nb] for j in range(args.warmup, args.warmup + args.steps))
        elapsed = ctl.allreduce([elapsed], "max")[0]
        total_nnz = ctl.allreduce([float(local_nnz)], "sum")[0]
        st = _ffi.Stats()
        _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))
        log("timed region done: %d steps, %.4f ms/step" % (args.steps, elapsed / max(args.steps, 1) * 1e3))

        # ---- per-kernel times: a pass of the same steps right behind the timed region, every kernel of every step between a pair
        # of HIP events on the library's own stream (the records cost ~4 us each between the kernels, nothing inside them)
        prof = _ffi.Profile()
        kernel_pass_steps = 0
        if not os.environ.get("FMHIP_BENCH_NO_EVENTS"):
            kernel_pass_steps = int(min(max(args.steps, 12), 48))
            _ffi.check(L.fmhip_profile_begin(hm))
            steps_run(args.warmup, kernel_pass_steps)
            sync()
            _ffi.check(L.fmhip_profile_end(hm, C.byref(prof)))
        cprof = None
        if comm is not None:
            # the exchange's own timers (a dozen event records per step) run in a short pass of their own too
            _ffi.check(L.fmhip_comm_profile_begin(comm.handle))
            steps_run(0, 12)
            sync()
            cp = _ffi.CommProfile()
            _ffi.check(L.fmhip_comm_profile_end(comm.handle, C.byref(cp)))
            cprof = cp.as_dict()
            barrier()

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
                log("REPLICAS DIFFER after the timed steps: max %r, -min %r" % (hi_, lo_))

        value = total_nnz / elapsed
        step_ms = elapsed / args.steps * 1e3

    # ---- the record (rank 0 builds and re-writes it; the other ranks only take part in the collective legs)
    kp = 32
    while kp < k:
        kp *= 2
    out, state = {}, {"pmc": {}, "live": None, "hbm_resident": None}
    lay = ds.layout() if rank == 0 else None
    nnz_all = int(d["row_ptr"][-1])
    # the run's context for benchkit.record / benchkit.dp_legs (everything below this line only reads these)
    x = types.SimpleNamespace(args=args, rank=rank, world=world, local_rank=local_rank, ctl=ctl, L=L, ffi=_ffi, fm=fm, ds=ds, hm=hm, hd=hd, nb=nb, bnnz=bnnz,
                              dp=dp, comm=comm, regs=regs, exchange=exchange, config=config, cfg=cfg, rows=rows, k=k, kp=kp, n1=n1, batch_rows=batch_rows,
                              dp_world=dp_world, relabelled=relabelled, lay=lay, nnz_all=nnz_all, settled=settled, step=step, steps_run=steps_run, sync=sync,
                              barrier=barrier, log=log, comm_note=comm_note, selftest_note=selftest_note)

    def build_roofline():
        """(Re)builds `roofline`, `step_roofline` and `kernels` from the kernel pass and whatever counter figures exist by now."""
        ab = alg_bytes(k)
        packed = k < kp
        pmc = state["pmc"]
        bi = binfo[0]
        hot = len(lay["hot_ids"]) > 0
        n_cols, nnz0, rows0 = bi["n_columns"], bi["nnz"], bi["rows"]
        nnz0_sparse = int(round(nnz0 * lay["nnz_sparse"] / max(nnz_all, 1)))             # batch 0's share of the sparse streams
        nnz0_sparse_b = int(round(nnz0 * lay["nnz_sparse_backward"] / max(nnz_all, 1)))  # ... of the transposes
        dense_apply = (use_dp and not (exchange == "rccl" and dp.exchange == "touched")) or n_cols * 2 > n1
        req = requested_bytes(kp, rows0, nnz0, nnz0_sparse, n_cols, hot, n_cols, dense_apply, n1, packed, nnz0_sparse_b, lay["hot_pages"])
        table_bytes = {"forward": n1 * kp * 4, "backward": rows0 * kp * 4}
        kern = kernel_table(prof, k, kp, req, pmc, table_bytes)
        out["kernels"] = kern
        if not kern:
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": None, "traffic": None,
                               "note": "no kernel pass ran (FMHIP_BENCH_NO_EVENTS)"}
            return
        # the dominant kernel = the longest launch
        dom = max(("forward", "backward"), key=lambda n: kern.get(n, {}).get("avg_ms", 0.0))
        # fraction of the step's time that the kernels' own ceilings account for (<= 1 when no kernel beats its ceiling)
        explained_ms = sum(e["requested_bytes_per_launch"] / (e["ceiling"]["GBps"] * 1e9) * 1e3 for e in kern.values() if e.get("ceiling"))
        out["step_roofline"] = {"alg_bytes_per_nnz": ab["step"], "alg_GBps": value / world * ab["step"] / 1e9,
                                "requested_bytes_per_step": sum(e.get("requested_bytes_per_launch", 0) for e in kern.values()),
                                "time_at_ceilings_ms": explained_ms, "frac": explained_ms / step_ms,
                                "note": "frac = sum over kernels of (requested bytes / that kernel's ceiling) / measured step time"}
        roof = roofline_block(kern, dom, ab, prof.as_dict(), pmc, step_ms, state["live"] is not None)
        roof["avg_launch_ms_from"] = ("a pass of %d of the same steps right behind the timed region, every kernel between a pair of HIP events on the "
                                      "library's stream (the timed region itself carries no event records)" % kernel_pass_steps)
        comp = compulsory_hbm_bytes(kp, rows0, nnz0_sparse, nnz0_sparse_b, n_cols, (n1 + 3) // 4 * 4, lay["hot_pages"])
        annotate_roofline(roof, comp, step_ms, out["step_roofline"]["frac"])
        if state["hbm_resident"]:
            roof["hbm_resident"] = state["hbm_resident"]
        out["roofline"] = roof

    legs_dp = {"twin": None, "no_exchange": None, "one_gpu_plain": None}

    def exchange_block():
        out["exchange"] = record.exchange_block(x, value, cprof, replicas, tuning, tuning_note, **legs_dp)

    def headline(stage):
        """Rank 0 (re)builds the record from the latest measure() and writes it."""
        if rank != 0:
            return
        state["pmc"] = committed_pmc(config, k, batch_rows)
        out.update({
            "metric": "nnz_per_sec_fm_sgd_training", "value": value, "unit": "nnz/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": record.config_block(x),
            "roofline": None,
            "cpu_baseline": None,
            "train": {"last_batch_mse": st.sse / max(st.rows, 1), "nonfinite": st.nonfinite},
            "setup_s": {"generate": t_gen, "load_transpose_h2d": t_load},
            "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
        })
        build_roofline()
        if use_dp:
            exchange_block()
        emitter.emit(out, stage)
        log("%s line written: %.2f G nnz/s" % (stage, value / 1e9))

    tuning = tuning_note = None
    sweeping = exchange == "rccl" and args.upper_fractions == "auto" and (world > 1 or args.emulate_allreduce or dp.exchange == "touched")
    if sweeping and world > 1:
        measure()
        headline("headline (default plan, before the cut / mode sweep)")
    if sweeping:
        tuning, tuning_note = dp_legs.tune_sweep(x)

    measure()
    headline("headline")

    # ---- rank 0 alone, the other ranks wait at the next barrier: the CPU baseline, then this run's own counter passes
    if rank == 0:
        if not args.no_cpu_baseline and not wide:
            test_stall("cpu_baseline")
            # rank 0's host cores, on rank 0's shard of the line's own workload (at N > 1 the ratio is the JOB's throughput
            # over that one-host baseline)
            cb = budget.run("cpu_baseline", args.cpu_budget * 1.6 + 6.0, cpu_baseline, d, k, n1, batch_rows, args.eta, regs, w0, w, v, args.cpu_budget)
            if cb:
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu"] = value / cb["value"]
                emitter.emit(out, "cpu_baseline")
        pmc_args = None
        if exchange == "rccl" and dp.exchange != "touched" and config == "C4" and not args.no_pmc and not args.tune and not args.hot_pages:
            # N > 1 (or --force-dp): the counters of THIS rank's workload under the step this line timed — the same shard size,
            # batch, cuts and exchange mode through the library's own data-parallel step with a one-rank communicator whose
            # collectives are the identity (tools/pmc_leg.py c4) — measured now, on rank 0's GPU, while the other ranks wait
            fr = ",".join(str(f) for f in dp.upper_fractions) or "none"
            pmc_args = ([os.path.join(ROOT, "tools", "pmc_leg.py"), "c4", "--rows", str(min(rows, 2 * batch_rows, max(batch_rows, 2_500_000))),
                         "--batch-rows", str(batch_rows), "--upper-fractions", fr, "--dp-exchange", dp.exchange, "--steps", "8", "--warmup", "4"], (4, 8),
                        "rank 0's GPU (tools/pmc_leg.py c4: the same shard size, batch, cuts and mode)")
        elif world == 1 and not use_dp and not args.no_pmc and config in ("C2", "C3") and not args.tune and not args.hot_pages:
            # the counters of THIS workload, measured now: rocprofv3 child processes over tools/pmc_leg.py (same config,
            # rows and batch; the committed profile is the fallback when rocprofv3 is not available)
            pmc_args = ([os.path.join(ROOT, "tools", "pmc_leg.py"), config.lower(), "--rows", str(rows), "--batch-rows", str(batch_rows)], None,
                        "(tools/pmc_leg.py: the same workload)")
        if pmc_args:
            test_stall("live_pmc")
            live = budget.run("live_pmc", 75.0, live_pmc, pmc_args[0], per_step=pmc_args[1])
            if live:
                state["live"] = live
                for kn, e in live.items():
                    state["pmc"].setdefault(kn, {}).update(e, traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation " + pmc_args[2])
                build_roofline()
                emitter.emit(out, "live_pmc")
                log("live counter passes done")
    barrier()

    # ---- sustained: the same steps back to back for >= 2 s (clock / power settled)
    sustained = None
    if not args.no_extra and agreed(rank != 0 or budget.allows("sustained", 4.0)):
        test_stall("sustained")
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
        if rank == 0:
            out["sustained"] = sustained
            emitter.emit(out, "sustained")

    # ---- the same shard and batch WITHOUT the exchange (what one GPU of the job does alone): rank 0, the others wait
    if use_dp and rank == 0 and not args.no_extra and budget.allows("one_gpu_legs", 10.0):
        legs_dp["no_exchange"], legs_dp["one_gpu_plain"] = dp_legs.one_gpu_legs(x)
        exchange_block()
        emitter.emit(out, "one_gpu_legs")
    barrier()

    # ---- the N = 1 line's own workload under the exchange: C3 on every GPU (collective: every rank takes part)
    if exchange == "rccl" and dp.exchange != "touched" and config != "C3" and not args.no_extra and agreed(rank != 0 or budget.allows("c3_twin", 25.0)):
        test_stall("c3_twin")
        t_leg = time.monotonic()
        legs_dp["twin"] = dp_legs.c3_twin_leg(x)
        if rank == 0:
            budget.spent["c3_twin"] = time.monotonic() - t_leg
            exchange_block()
            emitter.emit(out, "c3_twin")

    # ---- N = 1: the legs beside the headline (scoring, the HBM-resident model, C4 on one GPU, ALS), each under the budget
    if rank == 0 and world == 1 and not args.no_extra and not use_dp:
        extra = {}
        out["extra"] = extra

        def leg(name, estimate_s, fn, *a, **kw):
            test_stall(name)
            try:
                r_ = budget.run(name, estimate_s, fn, *a, **kw)
            except Exception as ex:   # noqa: BLE001 — an optional leg never takes the record down
                r_ = {"error": repr(ex)}
            if r_ is not None:
                extra[name] = r_
                emitter.emit(out, "extra." + name)
                log("leg %s done (%.1f s)" % (name, budget.spent.get(name, 0.0)))
            return r_

        leg("scoring", 3.0, scoring_leg, fm, ds, rows, nnz_all)
        ds.unpersist()
        fm.close(discard=True)
        del d
        hb = leg("hbm_resident", 45.0 if not args.no_pmc else 20.0, hbm_resident_leg, local_rank, with_pmc=not args.no_pmc)
        if hb and hb.get("frac_of_8TBps") is not None and out.get("roofline"):
            # the one workload whose counter bytes are, to a large part, HBM bytes: its whole-step fraction, at the top level
            state["hbm_resident"] = {"frac": hb["frac_of_8TBps"], "achieved": hb["fabric_GBps"], "unit": "GB/s", "value_nnz_per_s": hb["value"],
                                     "ms_per_step": hb["ms_per_step"], "traffic_source": hb["fabric_traffic_source"],
                                     "workload": hb["workload"],
                                     "what": "whole step of the HBM-resident leg (extra.hbm_resident): counter bytes per step / step time / 8 TB/s — V = 8.6 GB, "
                                             "24 distinct batches: the tables do NOT live in L2 / Infinity Cache here"}
            out["roofline"]["hbm_resident"] = state["hbm_resident"]
            emitter.emit(out, "roofline.hbm_resident")
        leg("c4_one_gpu", 22.0, c4_one_gpu_leg, local_rank, args.eta, regs)
        leg("als_c1", 6.0, als_c1, local_rank)
        leg("als_fields", 12.0, als_fields, local_rank)
        leg("als_long_columns", 25.0, als_long, local_rank)
    if rank == 0:
        emitter.emit(out, "final", final=True)
        log("final line written (%d lines in all)" % emitter.lines)
    if use_dp:
        ctl.barrier()
        if comm is not None:
            comm.close()
    if progress_file:
        progress_file.close()


if __name__ == "__main__":
    main()
