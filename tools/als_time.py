#!/usr/bin/env python3
"""bench.py's ALS legs on their own (for rocprofv3): BASELINE config 1's epoch, and the long-column shapes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps({"als_c1": bench.als_c1(0), "als_long_columns": bench.als_long(0)}))
