"""The parts of bench.py's JSON line that describe the RUN (its `config` block) and, at N > 1, the EXCHANGE (`exchange` block).
`x` is the run's context (bench.py: run_rank builds it): args, rank / world, the model / dataset / learner handles, the workload's
sizes.  Pure assembly: nothing here launches work except fmhip_grad_floats (a size query)."""
import ctypes as C


def config_block(x):
    """`config` of the line: the workload (BASELINE.json's configuration by name, with its sizes), how it was laid out and exchanged."""
    args, cfg, lay, dp, exchange = x.args, x.cfg, x.lay, x.dp, x.exchange
    use_dp = x.world > 1 or args.force_dp
    return {"workload": "%s: %d rows x %d features per GPU, k=%d, %s, fp32 mini-batch SGD" %
                        (x.config, x.rows, x.n1, x.k, ("39 hashed Criteo-shaped fields" + (", ids relabelled by frequency at load" if x.relabelled else ""))
                         if cfg.get("criteo") else
                         "nnz/row U{%d..%d}, ids Zipf(%.2f)" % (cfg["nnz_lo"], cfg["nnz_hi"], cfg["zipf_s"])),
            "rows_per_gpu": x.rows, "features": x.n1, "k": x.k, "batch_rows_per_gpu": x.batch_rows,
            "global_batch": x.batch_rows * x.dp_world,
            "batches_per_gpu": x.nb, "nnz_per_gpu": x.nnz_all, "eta": args.eta, "regs": x.regs,
            "settle_steps_before_warmup": x.settled,
            "dense_hot_block": {"pages": lay["hot_pages"], "features_forward_and_backward": len(lay["hot_ids"]),
                                "features_backward": len(lay["hot_ids_all"]),
                                "share_of_nonzeros_left_to_the_forward": lay["nnz_sparse"] / max(x.nnz_all, 1),
                                "share_of_nonzeros_left_to_the_backward": lay["nnz_sparse_backward"] / max(x.nnz_all, 1)},
            "backward_band_plan": {"ranges": lay["ranges"], "planned": lay["planned_ranges"], "band_affine": lay["band_affine_ranges"],
                                   "share_band_affine": lay["band_affine_ranges"] / max(lay["ranges"], 1),
                                   "note": "ranges of long columns walked on the XCD that owns their row band (FMHIP_TUNE_XCD_PLACEMENT)"},
            "parallelism": "dp%d" % x.world, "exchange": exchange,
            "transport": ("host-staged gloo over fmhip_comm_create_external, all ranks on GPU 0 (a rehearsal of the N-rank flow, "
                          "not a measurement)" if args.transport == "host" and use_dp else
                          ("host-staged between the ranks-as-threads of ONE process over fmhip_comm_create_external, all on GPU 0 (a "
                           "rehearsal of the N-rank flow, not a measurement)" if args.transport == "threads" else ("RCCL" if use_dp else "none"))),
            "allreduce": ("inside the library, touched rows only" if exchange == "rccl" and dp.exchange == "touched" else
                          ("inside the library, %s, overlapped with the feature-chunked backward, cuts at features %s" %
                           ("reduce-scatter -> sharded update -> all-gather" if dp.exchange == "sharded" else "all-reduce, every rank updates every row", dp.cuts))
                          if exchange == "rccl" and dp.cuts else
                          ("inside the library, one %s per step" % ("reduce-scatter + all-gather" if dp.exchange == "sharded" else "all-reduce") if exchange == "rccl" else
                           ("torch.distributed, orchestrated from Python" if exchange == "torch" else "none")))}


def exchange_block(x, value, cprof, replicas, tuning=None, tuning_note=None, twin=None, no_exchange=None, one_gpu_plain=None):
    """`exchange` of an N > 1 line: what travelled, how long the wire was busy / exposed, the self-test and replica verdicts, the
    sweep's candidates, and the legs that give this line its denominators."""
    args, dp, exchange, world, kp = x.args, x.dp, x.exchange, x.world, x.kp
    gf = C.c_int64()
    x.ffi.check(x.L.fmhip_grad_floats(x.hm, C.byref(gf)))
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
        xc["%s_one_gpu" % x.config.lower()] = one_gpu_plain
        xc["scaling_vs_%s_one_gpu" % x.config.lower()] = value / one_gpu_plain["value"]
    if x.comm_note:
        xc["note"] = x.comm_note
    if x.selftest_note:
        xc["selftest"] = x.selftest_note
    if replicas:
        xc["replicas"] = replicas
    if tuning:
        xc["cut_tuning"] = tuning
        xc["cut_tuning_note"] = tuning_note
    if args.emulate_allreduce:
        xc["emulated"] = "ring all-reduce over %s GPUs at bus bandwidth %s GB/s, as a delay on the comm stream (one real rank)" % tuple(args.emulate_allreduce.split(":"))
        if args.emulate_load:
            xc["emulated"] += "; the delay is spent by %d workgroups streaming the payload through HBM (read + write, twice per all-reduce)" % args.emulate_load
    return xc
