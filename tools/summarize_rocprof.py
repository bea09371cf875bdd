#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary.

    python tools/summarize_rocprof.py --stats <*_kernel_stats.csv> [--pmc <*_counter_collection.csv> ...] \
        [--skip-first N] > profiles/rNN_summary.txt

PMC values are averaged per kernel over the dispatches after the first N of that kernel
(warm-up).  Unit handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced
reads, so the read side is reported both raw and doubled (the doubled figure is the upper
bound used as `traffic`).
"""
import argparse
import collections
import csv
import re


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--skip-first", type=int, default=8)
    args = ap.parse_args()
    if args.stats:
        print("# rocprofv3 --kernel-trace --stats (durations in us)")
        print("%-28s %6s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in csv.DictReader(open(args.stats)):
            print("%-28s %6s %10.2f %10.2f %10.2f %7s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in args.pmc:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg:
        print("\n# rocprofv3 --pmc (mean per dispatch after the first %d dispatches of each kernel)" % args.skip_first)
        for kn in sorted(agg):
            if not kn.startswith("k_"):
                continue
            c = {cn: (sum(v[args.skip_first:]) / max(len(v[args.skip_first:]), 1)) for cn, v in agg[kn].items()}
            parts = ["%s=%.1f" % (cn, val) for cn, val in sorted(c.items())]
            extra = []
            if "FETCH_SIZE" in c:
                extra.append("read_MB raw=%.1f x2=%.1f" % (c["FETCH_SIZE"] * 1024 / 1e6, 2 * c["FETCH_SIZE"] * 1024 / 1e6))
            if "WRITE_SIZE" in c:
                extra.append("write_MB=%.1f" % (c["WRITE_SIZE"] * 1024 / 1e6))
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
                extra.append("L2_hit_rate=%.3f" % (c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)))
            print("%-24s %s | %s" % (kn, " ".join(parts), "; ".join(extra)))


if __name__ == "__main__":
    main()
