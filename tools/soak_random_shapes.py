#!/usr/bin/env python3
"""The random-shapes property test of tests/test_gpu_parity.py over MORE seeds than the suite runs (a one-off soak after a
kernel change): GPU gradient and one SGD epoch against the oracle, buffer-view and flat-address kernels.
    python3 tools/soak_random_shapes.py [seeds, default 6] [cases per seed, default 40] [first seed offset, default 1] [k values, comma-separated]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sparkfm_amd  # noqa: E402
import test_gpu_parity as t  # noqa: E402
from sparkfm_amd import _ffi  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
first = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # offset of the first seed from the suite's own
ks = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1, 2, 5, 8, 13, 16, 32, 40, 64)   # e.g. 100,128,200,256: the wide-row kernels
L = _ffi.load()
for s in range(n_seeds):
    for flat in (0, 1):
        L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, flat)
        try:
            t._random_shapes(sparkfm_amd, L, seed=20261003 + first + s, cases=cases, skip_diverged=True, ks=ks)
        finally:
            L.fmhip_tune(_ffi.TUNE_FLAT_ADDRESS, 0)
        print("seed %d flat %d: %d cases ok" % (20261003 + first + s, flat, cases), flush=True)
