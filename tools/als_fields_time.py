#!/usr/bin/env python3
"""The bench's `extra.als_fields` leg on its own: one ALS epoch on MovieLens-shaped rows (a user field and an item field), the
level schedule beside the GPU's sequential walk and the one-core oracle.   python3 tools/als_fields_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.als_fields(0), indent=1))
