#!/usr/bin/env python3
"""The HBM-resident leg of bench.py on its own (so that rocprofv3 can wrap exactly this workload):
C5's width — 2^25 hashed slots, k=64, V = 8.6 GB — on one GPU, Criteo-shaped rows, weight decay on.

    python3 tools/run_c5_shape.py [steps] [rows]        (never starts the rocprofv3 child passes itself: it is what they wrap)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
    print(json.dumps(bench.hbm_resident_leg(0, steps=steps, rows=rows, hashed_too='--hashed-too' in sys.argv, with_pmc=False)))
