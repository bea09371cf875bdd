#!/usr/bin/env python3
"""One workload of bench.py on its own, sized for a rocprofv3 pass (which wraps THIS program: `rocprofv3 --pmc X --
python3 tools/pmc_leg.py ...`; it spawns nothing and never re-executes itself):

    python3 tools/pmc_leg.py c3    [--steps 12] [--warmup 4]            the headline workload: C3, batch 250,000 rows
    python3 tools/pmc_leg.py c5hbm [--rows 2000000] [--hashed]          the HBM-resident leg: 2^25 slots x k=64 (V = 8.6 GB),
                                                                         Criteo-shaped rows, ids relabelled by frequency (or as hashed)
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
    ap.add_argument("workload", choices=["c3", "c2", "c5hbm"])
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
        rows = args.rows or 1_000_000
        d = synth.make_config(args.workload.upper(), rows=rows)
        fm = FMModel(n1 - 1, k, seed=cfg["seed"] + 1000, device=0, init_on_device=True)
    ds = DataSet.from_arrays(d, batch_rows=min(args.batch_rows, rows), device=0).cache()
    hm, hd, nb = fm.handle, ds.handle, ds.n_batches
    for j in range(args.warmup + args.steps):
        _ffi.check(L.fmhip_sgd_step(hm, hd, j % nb, 0.02, *regs, None))
    _ffi.check(L.fmhip_synchronize(hm))
    st = _ffi.Stats()
    _ffi.check(L.fmhip_step_stats(hm, C.byref(st)))
    print(json.dumps({"workload": args.workload, "rows": rows, "batches": nb, "steps": args.steps, "warmup": args.warmup,
                      "nnz_per_batch": [ds.batch_info(b)["nnz"] for b in range(min(nb, 4))], "nonfinite": st.nonfinite,
                      "relabelled": args.workload == "c5hbm" and not args.hashed}))
    ds.unpersist()
    fm.close(discard=True)


if __name__ == "__main__":
    main()
