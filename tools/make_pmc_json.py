#!/usr/bin/env python3
"""Condenses the rocprofv3 runs of tools/profile_round.sh into the two committed records:

    python tools/make_pmc_json.py <tag> <config> <k> <batch_rows> [command]  ->  profiles/<tag>_summary.txt
                                                                     profiles/pmc_traffic.json (entries of this config replaced)

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are KiB of
fabric-side requests (Infinity-Cache hits included); on gfx950 FETCH_SIZE counts a 128-B request of a wide
(16 B per lane) coalesced read as 64 B.  The row gathers and the hot block's streams are 16-B-per-lane reads, the
4-B index/value streams are not, and the counter cannot tell them apart: both the raw figure and the doubled one
are recorded, `traffic_bytes` = doubled reads + writes is the UPPER bound used as `traffic`.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTRA_PASSES = ("ta", "ta2", "ta3", "tcp", "tcp2", "ea", "ea2")      # tools/profile_round.sh


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1), (m.group(1) + (m.group(2) or ""))) if m else (None, name[:40])


PER_STEP = None      # (warmup steps, timed steps) of the profiled command: kernels launched several times per step are summed per step


def load_pmc(tag, name, skip):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    full = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            base, fn = short(r["Kernel_Name"])
            if base:
                agg[base][r["Counter_Name"]].append(float(r["Counter_Value"]))
                full[base] = fn
    def mean(v):
        if PER_STEP and len(v) % (PER_STEP[0] + PER_STEP[1]) == 0:
            lps = len(v) // (PER_STEP[0] + PER_STEP[1])
            return sum(v[PER_STEP[0] * lps:]) / PER_STEP[1]
        return sum(v[skip:]) / max(len(v[skip:]), 1)

    out = {}
    for kn, cs in agg.items():
        out[kn] = {cn: mean(v) for cn, v in cs.items()}
    return out, full


def main():
    tag, config, k, batch_rows = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    command = sys.argv[5] if len(sys.argv) > 5 else "python3 bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 4"
    skip = 4
    per_step_note = ""
    if len(sys.argv) > 7:      # <warmup steps> <timed steps>: per-STEP figures (the data-parallel step launches its backward once per interval)
        global PER_STEP
        PER_STEP = (int(sys.argv[6]), int(sys.argv[7]))
        per_step_note = " — kernels launched several times per step: SUM over one step's launches"
    lines = []
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    durations = {}
    if stats:
        lines.append("# rocprofv3 --kernel-trace --stats -- %s   (durations in us)" % command)
        lines.append("%-44s %6s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in csv.DictReader(open(stats[0])):
            base, fn = short(r["Name"])
            lines.append("%-44s %6s %10.2f %10.2f %10.2f %7s" % (fn, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                              float(r["MaxNs"]) / 1e3, r["Percentage"]))
            if base:
                durations[base] = float(r["AverageNs"]) / 1e3
    fetch, full = load_pmc(tag, "fetch", skip)
    write, _ = load_pmc(tag, "write", skip)
    l2, _ = load_pmc(tag, "l2", skip)
    sq, _ = load_pmc(tag, "sq", skip)
    extra = {name: load_pmc(tag, name, skip)[0] for name in EXTRA_PASSES}
    lines.append("")
    lines.append("# rocprofv3 --pmc, one pass per group (mean per dispatch after the first %d dispatches of each kernel%s)" % (skip, per_step_note))
    entries = []
    step_kernels = ("k_forward", "k_backward", "k_fixup", "k_apply")      # the SGD step; the dataset build's kernels are setup
    for kn in sorted(set(fetch) | set(write) | set(l2)):
        if not kn.startswith(step_kernels):
            continue
        fr = fetch.get(kn, {}).get("FETCH_SIZE")
        wr = write.get(kn, {}).get("WRITE_SIZE")
        hit = miss = None
        if kn in l2:
            hit, miss = l2[kn].get("TCC_HIT_sum"), l2[kn].get("TCC_MISS_sum")
        h = hit / max(hit + miss, 1.0) if hit is not None and miss is not None else None
        parts = []
        if fr is not None:
            parts.append("FETCH_SIZE=%.0f KiB (read %.1f MB raw, %.1f MB x2)" % (fr, fr * 1024 / 1e6, 2 * fr * 1024 / 1e6))
        if wr is not None:
            parts.append("WRITE_SIZE=%.0f KiB (%.1f MB)" % (wr, wr * 1024 / 1e6))
        if h is not None:
            parts.append("TCC_HIT=%.0f TCC_MISS=%.0f L2_hit=%.3f" % (hit, miss, h))
        if kn in sq:
            parts.append(" ".join("%s=%.3g" % (c, v) for c, v in sorted(sq[kn].items())))
        for name in EXTRA_PASSES:
            if kn in extra[name]:
                parts.append(" ".join("%s=%.4g" % (c, v) for c, v in sorted(extra[name][kn].items())))
        ea = dict(extra["ea"].get(kn, {}), **extra["ea2"].get(kn, {}))
        if "TCC_EA0_RDREQ_sum" in ea:
            # the request-size breakdown behind FETCH_SIZE: 128-B requests (TCC_BUBBLE), 32-B ones, the rest 64 B
            b128, b32, tot = ea.get("TCC_BUBBLE_sum", 0.0), ea.get("TCC_EA0_RDREQ_32B_sum", 0.0), ea["TCC_EA0_RDREQ_sum"]
            parts.append("read requests: %.3g x128B %.3g x32B %.3g x64B = %.1f MB by request size; to DRAM (MC) %.3g of %.3g" %
                         (b128, b32, tot - b128 - b32, (b128 * 128 + b32 * 32 + (tot - b128 - b32) * 64) / 1e6, ea.get("TCC_EA0_RDREQ_DRAM_sum", float("nan")), tot))
        lines.append("%-44s %s" % (full.get(kn, kn), " | ".join(parts)))
        if fr is not None and wr is not None:
            ent = {"config": config, "k": k, "batch_rows": batch_rows, "kernel": kn.replace("k_forward_wt", "k_forward").replace("k_backward_p", "k_backward").replace("k_apply_rows", "k_apply"),
                   "kernel_instance": full.get(kn, kn), "fetch_raw_bytes": int(fr * 1024), "fetch_corrected_bytes": int(2 * fr * 1024),
                   "write_bytes": int(wr * 1024), "traffic_bytes": int(2 * fr * 1024 + wr * 1024), "l2_hit": h,
                   "avg_us_kernel_trace": durations.get(kn), "per": "step" if PER_STEP else "launch",
                   "correction": "x2 applied to ALL fetches (upper bound: the 4-B index/value streams may not need it)",
                   "source": "profiles/%s_summary.txt" % tag}
            entries.append(ent)
    out_txt = os.path.join(ROOT, "profiles", "%s_summary.txt" % tag)
    open(out_txt, "w").write("\n".join(lines) + "\n")
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        old = json.load(open(pj))
    except (OSError, ValueError):
        old = {"entries": []}
    keep = [e for e in old.get("entries", []) if (e.get("config"), e.get("k"), e.get("batch_rows")) != (config, k, batch_rows)]
    step = {"config": config, "k": k, "batch_rows": batch_rows, "kernel": "step",
            "traffic_bytes": sum(e["traffic_bytes"] for e in entries), "source": "profiles/%s_summary.txt" % tag}
    doc = {"_comment": "Fabric-side traffic per launch from rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE and TCC_HIT/MISS in separate passes; "
                       "tools/profile_round.sh + tools/make_pmc_json.py). FETCH_SIZE is in KiB and doubled per the gfx950 wide-read "
                       "correction of MI355X_MICROARCH.md (upper bound); Infinity-Cache hits are counted too, so this is NOT an HBM "
                       "byte count for cache-resident tables. bench.py copies traffic_bytes into roofline.traffic and l2_hit into the "
                       "kernels' ceilings when no live pass ran; bench.py ALSO runs these passes itself (rocprofv3 --pmc child processes over "
                       "tools/pmc_leg.py, roofline.traffic_measured_in_this_run) and then this file is only the labelled fallback. Entries "
                       "with per = step (C4: the data-parallel step) are summed over the feature-interval launches of one step.",
           "entries": keep + entries + [step]}
    json.dump(doc, open(pj, "w"), indent=1)
    print(open(out_txt).read())


if __name__ == "__main__":
    main()
