#!/usr/bin/env python3
"""What `DataSet.cache()` costs and where: FMHIP_BUILD_TIMING=1 python3 tools/time_dataset_build.py [config] [rows] [batch_rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FMHIP_BUILD_TIMING", "1")
from sparkfm_amd import DataSet, synth  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "C3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else min(synth.CONFIGS[config]["rows"], 1_250_000)
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 250_000
d = synth.make_config(config, rows=rows)
for rep in range(2):
    t = time.time()
    ds = DataSet.from_arrays(d, batch_rows=batch).cache()
    print("%s: %d rows, %d nnz: cache() %.3f s" % (config, rows, d["row_ptr"][-1], time.time() - t), file=sys.stderr)
    ds.unpersist()
