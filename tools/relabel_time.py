#!/usr/bin/env python3
"""Frequency relabelling of Criteo-shaped ids (2^25 slots) on the host and on the GPU: python3 tools/relabel_time.py [rows, default 6000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import FeatureOrder, synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
n1 = 1 << 25
d = synth.make_config("C5", rows=rows)
col = d["col"]
out = {}
for name, dev in (("gpu", 0), ("gpu again", 0), ("host", None)):
    t0 = time.time()
    cnt = FeatureOrder.counts(col, n1, device=dev)
    t1 = time.time()
    order = FeatureOrder.from_counts(cnt, device=dev)
    t2 = time.time()
    rel = order.relabel(col)
    t3 = time.time()
    out[name] = rel
    print("%-10s %d rows, %d ids: counts %.3f s, rank %.3f s, relabel %.3f s, total %.3f s" % (name, rows, len(col), t1 - t0, t2 - t1, t3 - t2, t3 - t0), flush=True)
print("same numbering:", bool(np.array_equal(out["gpu"], out["host"])))
