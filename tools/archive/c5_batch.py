#!/usr/bin/env python3
"""The HBM-resident leg (C5's width on one GPU) at several mini-batch sizes.

    python3 tools/c5_batch.py [rows]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for br in (125_000, 250_000, 500_000):
        r = bench.hbm_resident_leg(0, steps=24, rows=rows, batch_rows=br, hashed_too=False, with_pmc=False)
        print(br, round(r["value"] / 1e9, 2), "G nnz/s", round(r["ms_per_step"], 4), "ms/step",
              {k: round(v["avg_ms"] * 1e3, 1) for k, v in r["kernels"].items()}, flush=True)
