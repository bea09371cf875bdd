"""The data-parallel parts of bench.py beside the timed region: the cut / exchange-mode sweep (bounded by wall-clock time), the
one-GPU denominators measured in the same process, and the C3 twin (the N = 1 line's workload under the exchange).  `x` is the
run's context (bench.py: run_rank): args, rank / world, ctl (barriers and timer reductions), the library handles, the learner."""
import time

# the pipelined schedule likes a small top slice (it should take about as long on the wire as the next position's pass A), a
# cheap second interval that starts the wire, and few launches; the dense / sharded schedules like a deeper pipeline.
# Ordered by what won on emulated 8 x 200-450 GB/s collectives (profiles/r04_emulated_dp_c4_*_modes3.json): under a time
# budget the sweep covers the likely winners first.
TUNE_ORDER = [("pipelined", (0.04, 0.1, 0.3)), ("dense", (0.05, 0.15, 0.3, 0.55)), ("sharded", (0.05, 0.15, 0.3, 0.55)),
              ("pipelined", (0.05, 0.12, 0.35)), ("dense", (0.12, 0.4)), ("pipelined", (0.04, 0.1)), ("sharded", (0.12, 0.4)),
              ("dense", (0.08, 0.25, 0.5)), ("pipelined", (0.04, 0.09, 0.2, 0.5)), ("dense", ()), ("pipelined", (0.05, 0.15, 0.3, 0.55)),
              ("dense", (0.3,)), ("sharded", (0.08, 0.25, 0.5)), ("pipelined", (0.08, 0.25, 0.5)), ("dense", (0.04, 0.1, 0.2, 0.35, 0.6)),
              ("sharded", ()), ("pipelined", (0.12, 0.4)), ("dense", (0.2,)), ("pipelined", (0.03, 0.07, 0.13, 0.22, 0.35, 0.6)),
              ("sharded", (0.3,)), ("pipelined", (0.04, 0.1, 0.2, 0.35, 0.6)), ("sharded", (0.2,)), ("sharded", (0.04, 0.1, 0.2, 0.35, 0.6))]
TOUCHED_CANDS = ((), (0.3,), (0.12, 0.4), (0.05, 0.15, 0.3, 0.55))


def tune_sweep(x):
    """Measure, don't guess: the best cut, and which exchange mode pays, depend on the collectives' real bandwidth on this node.
    Candidates in order of likely merit, 8 steps each, under a WALL-CLOCK budget (--tune-budget): every rank learns every
    candidate's agreed (max over ranks) cost, so all of them stop after the same candidate.  Leaves the learner planned with the
    fastest candidate; -> (candidates timed, note)."""
    args, dp = x.args, x.dp
    tuning = []
    if dp.exchange == "touched":
        order = [("touched", c_) for c_ in TOUCHED_CANDS]
    elif args.dp_exchange == "auto":
        order = list(TUNE_ORDER)
    else:
        order = [(m_, c_) for m_, c_ in TUNE_ORDER if m_ == dp.exchange]
        order += [(dp.exchange, c_) for c_ in sorted({c_ for _, c_ in TUNE_ORDER}) if (dp.exchange, c_) not in order]
    spent, seen_modes = 0.0, set()
    for mode, cand in order:
        # every mode gets its first candidate whatever the budget says (a record without one of the modes cannot say which is best)
        if spent > args.tune_budget and mode in seen_modes:
            continue
        t_c = time.perf_counter()
        dp.set_exchange(mode)
        dp.upper_fractions = cand
        dp.plan(x.fm, x.ds)
        x.step(0)
        x.sync()
        x.barrier()
        t0 = time.perf_counter()
        x.steps_run(0, 8)          # (8: a pipelined run's first forward pass and last slice are not overlapped with anything)
        x.sync()
        tt = x.ctl.allreduce([time.perf_counter() - t0, time.perf_counter() - t_c], "max")
        tuning.append({"exchange": mode, "upper_fractions": list(cand), "cuts": list(dp.cuts), "ms_per_step": tt[0] / 8 * 1e3})
        spent += tt[1]
        seen_modes.add(mode)
    note = "%d of %d candidates timed in %.1f s (--tune-budget %.0f s; every mode at least once)" % (len(tuning), len(order), spent, args.tune_budget)
    x.log("cut / mode sweep: " + note)
    best = min(tuning, key=lambda t: t["ms_per_step"])
    dp.set_exchange(best["exchange"])
    dp.upper_fractions = tuple(best["upper_fractions"])
    dp.plan(x.fm, x.ds)
    x.steps_run(0, 4)
    x.sync()
    x.barrier()
    return tuning, note


def one_gpu_legs(x):
    """Rank 0 alone (the other ranks wait at the caller's barrier): the same shard and batch WITHOUT the exchange — the split step
    with the dense update, and the plain one-GPU step (fmhip_sgd_step: the update merged into the fixup launch or rows-only, as
    N = 1 runs it), the denominator of this line's scaling, measured in the same process.  -> (no_exchange, one_gpu_plain)"""
    args, L, ffi, hm, hd, nb, bnnz, regs = x.args, x.L, x.ffi, x.hm, x.hd, x.nb, x.bnnz, x.regs
    n_leg = int(min(max(args.steps, 8), 64))
    out = []
    for plain in (False, True):
        def one(j):
            if plain:
                ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, args.eta, *regs, None))
            else:
                ffi.check(L.fmhip_step_compute(hm, hd, j % nb))
                ffi.check(L.fmhip_step_apply(hm, args.eta, *regs))
        x.sync()
        for j in range(4):
            one(j)
        x.sync()
        t0 = time.perf_counter()
        for j in range(n_leg):
            one(j)
        x.sync()
        dt = time.perf_counter() - t0
        out.append({"value": sum(bnnz[j % nb] for j in range(n_leg)) / dt, "unit": "nnz/s", "ms_per_step": dt / n_leg * 1e3,
                    "note": ("rank 0 alone, same shard and batch, the plain one-GPU step (fmhip_sgd_step), while the other ranks wait" if plain else
                             "rank 0 alone, same shard and batch, dense update, no all-reduce")})
    return out[0], out[1]


def c3_twin_leg(x):
    """The N = 1 line's own workload under the exchange: C3 on every GPU (weak scaling in the strict sense — the driver's per-N
    values compare C3 at N = 1 with C4 at N > 1, two different widths; this leg is the like-for-like number).  Collective: every
    rank calls it.  Leaves the learner planned for the run's own dataset again."""
    from sparkfm_amd import DataSet, FMModel, synth
    args, L, ffi, dp, regs = x.args, x.L, x.ffi, x.dp, x.regs
    c3 = synth.CONFIGS["C3"]
    rows3 = 1_000_000 if not args.rows else min(1_000_000, max(args.rows, 1000))     # a rehearsal with --rows keeps the twin small too
    d3 = synth.make_config("C3", rows=rows3, row_begin=x.rank * rows3)
    ds3 = DataSet.from_arrays(d3, name="C3", batch_rows=min(250_000, rows3), device=x.local_rank).cache()
    fm3 = FMModel(c3["features"] - 1, c3["k"], seed=c3["seed"] + 1000, device=x.local_rank, init_on_device=True)
    nb3 = ds3.n_batches
    nnz3 = [ds3.batch_info(b)["nnz"] for b in range(nb3)]
    n_leg = int(min(max(args.steps, 8), 64))

    def step3(j):
        ffi.check(L.fmhip_dp_step_at(fm3.handle, ds3.handle, j % nb3, x.comm.handle, args.eta, regs[0], regs[1], regs[2]))
    keep, keep_mode = dp.upper_fractions, dp.exchange
    best3 = None
    for mode3 in (("dense", "sharded") if args.dp_exchange == "auto" else (dp.exchange,)):
        dp.set_exchange(mode3)
        for cand in ((), (0.3,), (0.12, 0.4), (0.05, 0.15, 0.3, 0.55)):        # a 13.6 MB gradient wants fewer cuts than C4's 136 MB
            dp.upper_fractions = cand
            dp.plan(fm3, ds3)
            for j in range(3):
                step3(j)
            ffi.check(L.fmhip_synchronize(fm3.handle))
            x.barrier()
            t0 = time.perf_counter()
            for j in range(8):
                step3(j)
            ffi.check(L.fmhip_synchronize(fm3.handle))
            tt = x.ctl.allreduce([time.perf_counter() - t0], "max")
            if best3 is None or tt[0] < best3[0]:
                best3 = (tt[0], cand, mode3)
    dp.set_exchange(best3[2])
    dp.upper_fractions = best3[1]
    dp.plan(fm3, ds3)
    for j in range(4):
        step3(j)
    ffi.check(L.fmhip_synchronize(fm3.handle))
    x.barrier()
    t0 = time.perf_counter()
    for j in range(n_leg):
        step3(j)
    ffi.check(L.fmhip_synchronize(fm3.handle))
    x.barrier()
    tm = x.ctl.allreduce([time.perf_counter() - t0], "max")
    t3 = x.ctl.allreduce([float(sum(nnz3[j % nb3] for j in range(n_leg)))], "sum")
    twin = {"workload": "C3 on every GPU: %d rows x 100000 features per GPU, k=32, batch %d rows per GPU — the N = 1 line's workload" % (rows3, min(250_000, rows3)),
            "value": t3[0] / tm[0], "unit": "nnz/s", "ms_per_step": tm[0] / n_leg * 1e3,
            "allreduce_bytes_per_step": 4 * (32 + (c3["features"] + 31) // 32 * 32 * 34), "cuts": list(dp.cuts), "exchange": dp.exchange}
    ds3.unpersist()
    fm3.close(discard=True)
    dp.set_exchange(keep_mode)
    dp.upper_fractions = keep
    dp.plan(x.fm, x.ds)
    x.barrier()
    return twin
