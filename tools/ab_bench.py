#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (cdna guide §5.4 rule 24).

    python tools/ab_bench.py --fwd 0,1,2 --bwd 0,1 [--config C3] [--rounds 5]

Prints the median per-kernel device time (HIP events) for every (fwd, bwd) variant pair.
"""
import argparse
import ctypes as C
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fwd", default="0")
    ap.add_argument("--bwd", default="1")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--batch-rows", type=int, default=131072)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--row-block", type=int, default=0, help="rows per row block of the transposes (fmhip_tune key 3)")
    ap.add_argument("--xcd", type=int, default=0, help="XCD-aware workgroup placement in the backward (fmhip_tune key 4)")
    ap.add_argument("--fwd-occ", type=int, default=0, help="cap on the forward's resident workgroups per CU (fmhip_tune key 6; 0 = all that fit)")
    ap.add_argument("--row-order", type=int, default=1, help="forward walks rows longest-first (fmhip_tune key 7)")
    ap.add_argument("--hot", type=int, default=0, help="dense hot block for the most frequent features (fmhip_tune key 5)")
    ap.add_argument("--tile", type=int, default=0, help="LDS V-tile rows for forward variant 20 (0 = auto: 128 KiB)")
    args = ap.parse_args()
    from sparkfm_amd import DataSet, FMModel, _ffi, synth
    cfg = synth.CONFIGS[args.config]
    rows = args.rows or min(cfg["rows"], 1_000_000)
    k = args.k or cfg["k"]
    d = synth.make_config(args.config, rows=rows)
    n1 = cfg["features"]
    w0, w, v = synth.init_params(1, n1, k)
    if args.row_block:
        _ffi.load().fmhip_tune(_ffi.TUNE_ROW_BLOCK, args.row_block)
    _ffi.load().fmhip_tune(_ffi.TUNE_HOT_BLOCK, args.hot)
    _ffi.load().fmhip_tune(_ffi.TUNE_FORWARD_OCCUPANCY, args.fwd_occ)
    _ffi.load().fmhip_tune(_ffi.TUNE_ROW_ORDER, args.row_order)
    ds = DataSet.from_arrays(d, batch_rows=min(args.batch_rows, rows)).cache()
    fm = FMModel(n1 - 1, k)
    fm.w0, fm.w, fm.v = w0, w, v
    L = _ffi.load()
    hm, hd, nb = fm.handle, ds.handle, ds.n_batches
    if args.tile:
        L.fmhip_tune(_ffi.TUNE_TILE_ROWS, args.tile)
    if args.xcd:
        L.fmhip_tune(_ffi.TUNE_XCD_PLACEMENT, args.xcd)
    variants = list(itertools.product([int(x) for x in args.fwd.split(",")], [int(x) for x in args.bwd.split(",")]))
    res = {vv: [] for vv in variants}
    for rnd in range(args.rounds + 1):
        for vv in variants:
            L.fmhip_tune(_ffi.TUNE_FORWARD_KERNEL, vv[0])
            L.fmhip_tune(_ffi.TUNE_BACKWARD_KERNEL, vv[1])
            _ffi.check(L.fmhip_profile_begin(hm))
            for j in range(nb):
                _ffi.check(L.fmhip_sgd_step(hm, hd, j, 0.02, 0.0, 1e-4, 1e-4, None))
            p = _ffi.Profile()
            _ffi.check(L.fmhip_profile_end(hm, C.byref(p)))
            if rnd:
                res[vv].append([p.ms[i] / max(p.launches[i], 1) * 1e3 for i in range(5)])
    print("%-10s %9s %9s %9s %9s %9s %9s" % ("fwd,bwd", "forward", "reduce", "backward", "fixup", "apply", "sum_us"))
    for vv in variants:
        m = np.median(np.array(res[vv]), axis=0)
        print("%-10s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f" % ("%d,%d" % vv, *m, m.sum()))
    st = _ffi.Stats()
    L.fmhip_step_stats(hm, C.byref(st))
    print("last batch mse %.6f nonfinite %d" % (st.sse / max(st.rows, 1), st.nonfinite))


if __name__ == "__main__":
    main()
