#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV of the data-parallel step into a timeline of ONE steady-state step: which kernels ran
on which queue, when, and where the compute stream sat idle.   python3 tools/r05_trace_timeline.py <trace dir>"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+|rocclr_[a-zA-Z]+|ncclDevKernel[A-Za-z_0-9]*)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:24]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# steady state: the last 12 forward-pass-A launches delimit 11 steps; take the middle ones
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_forward_wt")]
if len(starts) < 14:
    print("too few steps in the trace:", len(starts)); sys.exit(1)
a, b = starts[-9], starts[-8]
t0 = rows[a][0]
print("one steady-state step (us from its pass A's start); queue = HSA queue (compute stream / comm stream)")
step = [r for r in rows if rows[a][0] <= r[0] < rows[b][0]]
qs = sorted({r[3] for r in step})
for r in step:
    print("  q%-3s %8.1f -> %8.1f  (%6.1f)  %s" % (r[3], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2]))
print("step length %.1f us" % ((rows[b][0] - t0) / 1e3))
for q in qs:
    ks = [r for r in step if r[3] == q]
    busy = sum(r[1] - r[0] for r in ks) / 1e3
    gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
    print("queue %s: %d kernels, busy %.1f us, gaps between its kernels: %s" % (q, len(ks), busy, " ".join("%.1f" % g for g in gaps)))
# averages over the last 8 steps
tot = []
for i in range(len(starts) - 9, len(starts) - 1):
    tot.append((rows[starts[i + 1]][0] - rows[starts[i]][0]) / 1e3)
print("last 8 steps (us):", " ".join("%.1f" % t for t in tot))
