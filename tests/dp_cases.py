"""Data-parallel cases shared by the GPU tests, tests/dp_case_worker.py and tools/soak_dp.py: a case is a dict
(shards, model, batch size, cuts, exchange mode ...); `oracle_epochs` runs the fp64 oracle over the equivalent GLOBAL
batches (lock-step: global batch j = every rank's j-th mini-batch, or its order[j]-th when the case shuffles), and
`run_threads` runs the library's own step (fmhip_dp_epoch / fmhip_dp_step) with every rank a thread of this process."""
import numpy as np


def shard(cfg, rank):
    from sparkfm_amd import synth
    rows = cfg["rows"][rank]
    if rows == 0:
        return dict(row_ptr=np.zeros(1, np.int64), col=np.zeros(0, np.int32), val=np.zeros(0, np.float32), y=np.zeros(0, np.float32))
    d = synth.make_zipf(cfg["seed"], rows, cfg["n1_data"], cfg["lo"], cfg["hi"], zipf_s=cfg.get("zipf_s", 1.05),
                        row_begin=int(sum(cfg["rows"][:rank])))
    if cfg.get("reverse_ids"):
        # ids NOT ranked by frequency: the most frequent features carry the highest ids (hashed / field-ordered ids do that to
        # some of them) — the dense hot block's features then sit at or above a data-parallel plan's cuts
        d = dict(d, col=(cfg["n1_data"] - 1 - d["col"]).astype(np.int32))
    return d


def init(cfg):
    from sparkfm_amd import synth
    w0, w, v = synth.init_params(cfg["seed"] + 1, cfg["n1"], cfg["k"], stdev=0.05)
    w = np.random.default_rng(cfg["seed"] + 2).normal(0, 0.05, cfg["n1"])
    return 0.05, w, v


def step_orders(cfg, steps):
    """The batch position every rank takes at each step of each epoch: ascending, or a seeded permutation per epoch
    (`shuffle_seed`: the same on every rank — a rank without that batch contributes zeros)."""
    out = []
    for ep in range(cfg["epochs"]):
        if cfg.get("shuffle_seed") is None:
            out.append(list(range(steps)))
        else:
            out.append([int(x) for x in np.random.default_rng(cfg["shuffle_seed"] + ep).permutation(steps)])
    return out


def n_steps(cfg):
    br = cfg["batch_rows"]
    return max((r + br - 1) // br for r in cfg["rows"])


def oracle_epochs(cfg):
    import oracle
    shards = [shard(cfg, r) for r in range(len(cfg["rows"]))]
    w0, w, v = init(cfg)
    br = cfg["batch_rows"]
    for order in step_orders(cfg, n_steps(cfg)):
        for j in order:
            rp, cols, vals, ys = [0], [], [], []
            for d in shards:
                n = len(d["y"])
                lo, hi = min(n, j * br), min(n, (j + 1) * br)
                if hi > lo:
                    a, b = int(d["row_ptr"][lo]), int(d["row_ptr"][hi])
                    cols.append(d["col"][a:b])
                    vals.append(d["val"][a:b].astype(np.float64))
                    rp.extend((d["row_ptr"][lo + 1:hi + 1] - a + rp[-1]).tolist())
                    ys.append(d["y"][lo:hi].astype(np.float64))
            if not ys:
                continue
            y = np.concatenate(ys)
            w0, w, v, _ = oracle.sgd_step(w0, w, v, 0, len(y), np.array(rp, np.int64), np.concatenate(cols) if cols else np.zeros(0, np.int32),
                                          np.concatenate(vals) if vals else np.zeros(0), y, cfg["eta"], 0.0, cfg["regw"], cfg["regv"])
    return w0, w, v


def run_rank(cfg, rank, make_comm, barrier):
    """One rank of a case through the library; -> dict of what the checks compare."""
    from sparkfm_amd import DataSet, FMModel, _ffi
    from sparkfm_amd.distributed import HipDataParallelSGD
    ds = DataSet.from_arrays(shard(cfg, rank), batch_rows=cfg["batch_rows"], device=0).cache()
    w0, w, v = init(cfg)
    fm = FMModel(cfg["n1"] - 1, cfg["k"], device=0)
    fm.w0, fm.w, fm.v = w0, w, v
    comm = make_comm(fm)
    dp = HipDataParallelSGD(comm, eta=cfg["eta"], regw=cfg["regw"], regv=cfg["regv"], exchange=cfg["exchange"],
                            upper_fractions=tuple(cfg["fractions"]))
    if not cfg.get("stepwise"):
        # fmhip_dp_epoch, or fmhip_dp_epoch_order with the case's seeded permutation (the same array on every rank)
        dp.plan(fm, ds)
        assert dp.plan_steps() == n_steps(cfg)
        for order in step_orders(cfg, n_steps(cfg)):
            dp.learn(fm, ds, order=None if cfg.get("shuffle_seed") is None else order)
        stats = dp.last_stats
    else:
        # the host walks the positions itself through fmhip_dp_step_at (every rank names the position, with or without rows)
        import ctypes
        dp.plan(fm, ds)
        for order in step_orders(cfg, n_steps(cfg)):
            for j in order:
                dp.step_at(fm, ds, j)
        st = _ffi.Stats()
        _ffi.check(_ffi.load().fmhip_step_stats(fm.handle, ctypes.byref(st)))
        stats = st.as_dict()
        stats["steps"] = n_steps(cfg)
    out = dict(w0=fm.w0, w=fm.w.copy(), v=fm.v.copy(), calls=np.array(getattr(comm, "calls", []), np.int64).reshape(-1, 2),
               cuts=np.array(dp.cuts or [], np.int64), rows=stats["rows"], steps=stats["steps"], info=dp.exchange_info())
    barrier()
    comm.close()
    ds.unpersist()
    fm.close(discard=True)
    return out


def run_threads(cfg):
    """Every rank of the case as a thread of THIS process over ThreadStagedComm; -> per-rank result dicts."""
    from sparkfm_amd import _ffi
    from sparkfm_amd.distributed import ThreadStagedComm, run_thread_ranks
    _ffi.load()
    world = len(cfg["rows"])
    return run_thread_ranks(world, lambda r, g: run_rank(cfg, r, lambda fm: ThreadStagedComm(fm, r, g), g.barrier))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-9))


def check(cfg, res, tol=1e-5):
    """Replicas bit-identical, same collectives on every rank, the oracle over the global batches matched."""
    r0 = res[0]
    tag = {k_: cfg[k_] for k_ in ("rows", "n1_data", "n1", "k", "lo", "hi", "batch_rows", "exchange", "fractions", "epochs", "regw", "regv")}
    for r in range(1, len(res)):
        r1 = res[r]
        assert np.array_equal(r0["v"], r1["v"]) and np.array_equal(r0["w"], r1["w"]) and float(r0["w0"]) == float(r1["w0"]), (tag, "replicas differ", r)
        assert np.array_equal(r0["calls"], r1["calls"]), (tag, "the ranks issued different collectives", r)
        assert np.array_equal(r0["cuts"], r1["cuts"]), (tag, "cuts differ", r)
    w0, w, v = oracle_epochs(cfg)
    ev, ew = rel(r0["v"], v), rel(r0["w"], w)
    assert ev <= tol and ew <= tol and abs(float(r0["w0"]) - w0) <= tol * abs(w0) + 1e-6, (tag, ev, ew, float(r0["w0"]), w0)
    return ev, ew
