#!/usr/bin/env python3
"""One workload of bench.py on its own, sized for a rocprofv3 pass (which wraps THIS program: `rocprofv3 --pmc X --
python3 tools/pmc_leg.py ...`; it spawns nothing and never re-executes itself):

    python3 tools/pmc_leg.py c3    [--steps 12] [--warmup 4]            the headline workload: C3, batch 250,000 rows
    python3 tools/pmc_leg.py c5hbm [--rows 2000000] [--hashed]          the HBM-resident leg: 2^25 slots x k=64 (V = 8.6 GB),
                                                                         Criteo-shaped rows, ids relabelled by frequency (or as hashed)
    python3 tools/pmc_leg.py c4 [--rows 1250000] [--batch-rows 625000] [--upper-fractions 0.05,0.15,0.3,0.55] [--dp-exchange dense]
                                                                         one rank's share of the N > 1 line: C4's width through the
                                                                         library's DATA-PARALLEL step (fmhip_dp_step: forward, one
                                                                         backward + fixup per feature interval, per-interval update)
                                                                         over a one-rank communicator whose collectives are the identity
Prints one JSON line describing what ran (rows, batches, nonzeros per launch).
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["c3", "c2", "c5hbm", "c4"])
    ap.add_argument("--upper-fractions", default="0.05,0.15,0.3,0.55", help="c4: the cuts of the backward ('none' = one interval)")
    ap.add_argument("--dp-exchange", default="dense", choices=["dense", "sharded", "pipelined"])
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--batch-rows", type=int, default=250_000)
    ap.add_argument("--hashed", action="store_true", help="c5hbm: keep the hashed ids as generated (no relabelling by frequency)")
    args = ap.parse_args()
    from sparkfm_amd import DataSet, FeatureOrder, FMModel, _ffi, synth
    L = _ffi.load()
    regs = (0.0, 1e-4, 1e-4)
    if args.workload == "c5hbm":
        n1, k = 1 << 25, 64
        rows = args.rows or 2_000_000
        d = synth.make_config("C5", rows=rows)
        if not args.hashed:
            d["col"] = FeatureOrder.fit(d["col"], n1).relabel(d["col"])
        fm = FMModel(n1 - 1, k, seed=5, device=0, init_on_device=True)
    else:
        cfg = synth.CONFIGS[args.workload.upper()]
        n1, k = cfg["features"], cfg["k"]
        rows = args.rows or (1_250_000 if args.workload == "c4" else 1_000_000)
        d = synth.make_config(args.workload.upper(), rows=rows)
        fm = FMModel(n1 - 1, k, seed=cfg["seed"] + 1000, device=0, init_on_device=True)
    ds = DataSet.from_arrays(d, batch_rows=min(args.batch_rows, rows), device=0).cache()
    hm, hd, nb = fm.handle, ds.handle, ds.n_batches
    if args.workload == "c4":
        from sparkfm_amd.distributed import HipDataParallelSGD, RcclComm

        class IdentityComm(RcclComm):
            """A world of one over fmhip_comm_create_external: every collective leaves the buffer as it is."""

            def __init__(self, fm):   # noqa: D107
                self.rank, self.world = 0, 1
                self._fn = _ffi.CollectiveFn(lambda ctx, dev, count, kind, stream: 0)
                self._h = C.c_void_p()
                _ffi.check(L.fmhip_comm_create_external(fm.handle, 0, 1, self._fn, None, C.byref(self._h)))

        comm = IdentityComm(fm)
        fr = () if args.upper_fractions == "none" else tuple(float(x) for x in args.upper_fractions.split(","))
        dp = HipDataParallelSGD(comm, eta=0.02, reg0=regs[0], regw=regs[1], regv=regs[2], upper_fractions=fr, exchange=args.dp_exchange)
        dp.plan(fm, ds)
        for j in range(args.warmup + args.steps):
            _ffi.check(L.fmhip_dp_step(hm, hd, j % nb, comm.handle, 0.02, *regs))
    else:
        for j in range(args.warmup + args.steps):
            _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, 0.02, *regs, None))
    _ffi.check(L.fmhip_synchronize(hm))
    st = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))
    print(json.dumps({"workload": args.workload, "rows": rows, "batches": nb, "steps": args.steps, "warmup": args.warmup,
                      "nnz_per_batch": [ds.batch_info(b)["nnz"] for b in range(min(nb, 4))], "nonfinite": st.nonfinite,
                      "relabelled": args.workload == "c5hbm" and not args.hashed}))
    if args.workload == "c4":
        comm.close()
    ds.unpersist()
    fm.close(discard=True)


if __name__ == "__main__":
    main()
