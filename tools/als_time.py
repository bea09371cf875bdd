#!/usr/bin/env python3
"""BASELINE config 1's ALS epoch on the GPU beside the CPU oracle (bench.py's als_c1 leg on its own, for rocprofv3)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.als_c1(0)))
