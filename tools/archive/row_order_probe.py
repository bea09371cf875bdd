#!/usr/bin/env python3
"""Host-side probe (no GPU): would PERMUTING a mini-batch's rows give the backward's column walk more L2 locality?

The gradient of a batch is a sum over its rows, so rows may be renumbered freely.  The walk cuts a
batch's feature-sorted transpose into ranges of 64 entries; a range whose rows fall into one row band (1/16 of the batch:
2 MB of P at C3) can be placed on the XCD whose L2 holds that band.  For one batch of a BASELINE config this measures,
for several row orders, (a) the share of ranges that lie wholly inside one band — what the band-affine plan can place —
and (b) the mean share of a range's entries that sit in its BEST band (what any range-to-XCD assignment could hit at most).
Orders: as stored; rows sorted by their rarest feature; by a min-hash of their cold feature ids; by the (rarest, 2nd rarest) pair.
    python3 tools/row_order_probe.py [C3|C4|C5] [batch_rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkfm_amd import synth  # noqa: E402


def measure(col, rowid, n_rows, hot, bands=16, range_len=64):
    keep = ~np.isin(col, hot)
    c, r = col[keep], rowid[keep]
    order = np.lexsort((r, c))                    # feature-major, rows ascending inside a column: the transposed stream
    c, r = c[order], r[order]
    n = len(c) // range_len * range_len
    band = (r[:n].astype(np.int64) * bands // n_rows).reshape(-1, range_len)
    same = (band == band[:, :1]).all(axis=1)
    best = np.zeros(len(band))
    for b in range(bands):
        best = np.maximum(best, (band == b).mean(axis=1))
    # ranges inside ONE column (the ones the current plan considers) vs ranges holding several columns
    cc = c[:n].reshape(-1, range_len)
    one_col = cc[:, 0] == cc[:, -1]
    return dict(ranges=len(band), wholly_in_one_band=float(same.mean()), mean_best_band_share=float(best.mean()),
                one_column_ranges=float(one_col.mean()), best_share_multi_column_ranges=float(best[~one_col].mean()) if (~one_col).any() else 0.0)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
    d = synth.make_config(cfg, rows=rows)
    if synth.CONFIGS[cfg].get("criteo"):
        from sparkfm_amd import FeatureOrder
        d["col"] = FeatureOrder.fit(d["col"], synth.CONFIGS[cfg]["features"]).relabel(d["col"])
    col, rp = d["col"], d["row_ptr"]
    lens = np.diff(rp)
    rowid = np.repeat(np.arange(rows, dtype=np.int32), lens)
    cnt = np.bincount(col, minlength=int(col.max()) + 1)
    hot = np.argsort(-cnt)[:64]                      # the dense hot block's features leave the transposed stream
    cold_mark = np.sort(cnt)[::-1][min(len(cnt) - 1, 1000)]      # "cold" = outside the 1000 most frequent features
    res = {}
    res["as stored"] = measure(col, rowid, rows, hot)
    # rarest feature of each row = its largest id when ids are frequency-ranked; here by count
    rare_key = np.full(rows, -1, np.int64)
    c_cnt = cnt[col]
    o = np.lexsort((c_cnt, rowid))                   # per row, entries by ascending count
    first = rp[:-1][lens > 0]
    rarest = np.full(rows, -1, np.int64)
    rarest[lens > 0] = col[o][first]
    perm = np.argsort(rarest, kind="stable")
    new_id = np.empty(rows, np.int32)
    new_id[perm] = np.arange(rows, dtype=np.int32)
    res["rows sorted by their rarest feature"] = measure(col, new_id[rowid], rows, hot)
    second = np.full(rows, -1, np.int64)
    has2 = lens > 1
    second[has2] = col[o][rp[:-1][has2] + 1]
    perm = np.lexsort((second, rarest))
    new_id[perm] = np.arange(rows, dtype=np.int32)
    res["rows sorted by (rarest, 2nd rarest)"] = measure(col, new_id[rowid], rows, hot)
    # min-hash over the cold ids of a row
    h = (col.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)
    h[c_cnt >= cold_mark] = np.iinfo(np.uint64).max
    mh = np.full(rows, np.iinfo(np.uint64).max, np.uint64)
    np.minimum.at(mh, rowid, h)
    perm = np.argsort(mh, kind="stable")
    new_id[perm] = np.arange(rows, dtype=np.int32)
    res["rows sorted by a min-hash of their cold ids"] = measure(col, new_id[rowid], rows, hot)
    for name, m in res.items():
        print("%-46s ranges %d  wholly in one of 16 bands %.3f  mean best-band share %.3f  (one-column ranges %.3f; multi-column ranges' best share %.3f)" %
              (name, m["ranges"], m["wholly_in_one_band"], m["mean_best_band_share"], m["one_column_ranges"], m["best_share_multi_column_ranges"]))


if __name__ == "__main__":
    main()
