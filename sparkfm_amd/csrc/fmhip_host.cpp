// fmhip_host.cpp — the library's pure host arithmetic (fmhip_host.h): no HIP, no GPU.  Part of libfmhip.so, and compiled on
// its own with g++ -fsanitize=address,undefined by the CPU suite (tests/host_arith_harness.cpp).
#include "fmhip_host.h"

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <numeric>

namespace fmhip {
namespace host {

// Host-side passes of the dataset build (validation, dense-hot-block split, forward row order, fp32
// re-pack) run over row chunks on all host cores: they are what `DataSet.cache()` costs before the
// device takes over (single-threaded they took 3.9 s for C4's 10 M rows).
int host_threads(int64_t work_items) {
    unsigned hc = std::thread::hardware_concurrency();
    int64_t t = hc ? (int64_t)hc : 4;
    if (const char *e = getenv("FMHIP_HOST_THREADS")) t = atoi(e);
    t = std::min<int64_t>({t, 32, work_items / 65536 + 1});
    return (int)std::max<int64_t>(t, 1);
}

void shard_bounds(int64_t n_rows, const int64_t *row_ptr, int world, int rank, int64_t *lo, int64_t *hi) {
    const int64_t nnz = row_ptr[n_rows];
    // boundary of rank i: the row offset NEAREST to i/world of the stored nonzeros (so one giant row does
    // not drag every row before it into the same shard); datasets without nonzeros fall back to row counts
    auto bound = [&](int i) -> int64_t {
        if (i <= 0) return 0;
        if (i >= world) return n_rows;
        if (nnz == 0) return n_rows * i / world;
        const int64_t target = (int64_t)((__int128)nnz * i / world);
        int64_t r = std::lower_bound(row_ptr, row_ptr + n_rows + 1, target) - row_ptr;
        if (r > 0 && (r > n_rows || target - row_ptr[r - 1] < row_ptr[r] - target)) --r;
        return r;
    };
    *lo = std::min(bound(rank), n_rows);
    *hi = std::min(std::max(bound(rank + 1), *lo), n_rows);
}

int64_t feature_counts(int64_t nnz, const int32_t *col, int64_t n1, int64_t *counts) {
    const int T = host_threads(nnz);
    std::atomic<int64_t> bad{-1};
    // a private table per thread while that stays small (<= 64 MiB each), one shared table with atomic adds beyond
    const bool private_tables = T > 1 && n1 <= (int64_t)1 << 23;
    std::vector<std::vector<int64_t>> part(private_tables ? (size_t)T : 0);
    parallel_chunks(nnz, T, [&](int t, int64_t lo, int64_t hi) {
        int64_t *dst = counts;
        if (private_tables) {
            part[(size_t)t].assign((size_t)n1, 0);
            dst = part[(size_t)t].data();
        }
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = col[i];
            if (c < 0 || c >= n1) { bad.store(i); return; }
            if (private_tables || T == 1) ++dst[c];
            else __atomic_fetch_add(&dst[c], (int64_t)1, __ATOMIC_RELAXED);
        }
    });
    if (bad.load() >= 0) return bad.load();
    if (private_tables)
        parallel_chunks(n1, T, [&](int, int64_t lo, int64_t hi) {
            for (int t = 0; t < T; ++t) {
                if (part[(size_t)t].empty()) continue;          // (a thread whose chunk was empty never made its table)
                const int64_t *src = part[(size_t)t].data();
                for (int64_t f = lo; f < hi; ++f) counts[f] += src[f];
            }
        });
    return -1;
}

void rank_from_counts(int64_t n1, const int64_t *counts, int32_t *rank, int32_t *by_rank) {
    std::vector<int32_t> order((size_t)n1);
    std::iota(order.begin(), order.end(), 0);
    // descending count, ties by ascending id: every rank of a job derives the same order from the same counts
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return counts[a] > counts[b]; });
    for (int64_t r = 0; r < n1; ++r) {
        rank[order[(size_t)r]] = (int32_t)r;
        if (by_rank) by_rank[r] = order[(size_t)r];
    }
}

int64_t relabel_columns(int64_t nnz, const int32_t *col, int64_t n1, const int32_t *rank, int32_t *out) {
    std::atomic<int64_t> bad{-1};
    parallel_chunks(nnz, host_threads(nnz), [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = col[i];
            if (c < 0 || c >= n1) { bad.store(i); return; }
            out[i] = rank[c];
        }
    });
    return bad.load();
}

// Host-side metadata of one batch from its column offsets (the transposed stream itself is built
// on the device, csc_build.hip): the column open at the start of every 64-entry range and the
// columns whose sum is assembled by k_fixup.
void finish_batch_meta(HostBatch &hb, int32_t nnz, std::vector<int32_t> &cnt, std::vector<int32_t> &base) {
    const size_t nc = hb.cfeat.size();
    // destinations of the column pieces: a feature with one piece stores straight to its G row; a
    // feature with several (row-blocked stream) gets consecutive piece rows, in stream (= row block)
    // order, that k_fixup2 sums.  `cnt` / `base` are zeroed scratch arrays of dimension + 1 entries.
    {
        std::vector<int32_t> multi;
        hb.n_feats = 0;
        for (size_t s = 0; s < nc; ++s) {
            const int32_t c = ++cnt[hb.cfeat[s]];
            if (c == 1) ++hb.n_feats;
            if (c == 2) multi.push_back(hb.cfeat[s]);
        }
        std::sort(multi.begin(), multi.end());
        hb.mp_feat = multi;
        hb.mp_ptr.assign(multi.size() + 1, 0);
        for (size_t m = 0; m < multi.size(); ++m) {
            base[multi[m]] = hb.mp_ptr[m];
            hb.mp_ptr[m + 1] = hb.mp_ptr[m] + cnt[multi[m]];
        }
        hb.n_pieces = multi.empty() ? 0 : hb.mp_ptr[multi.size()];
        hb.cdst.resize(nc);
        for (size_t s = 0; s < nc; ++s) {
            const int32_t f = hb.cfeat[s];
            hb.cdst[s] = cnt[f] > 1 ? -1 - (base[f]++) : f;
        }
        for (size_t s = 0; s < nc; ++s) { cnt[hb.cfeat[s]] = 0; base[hb.cfeat[s]] = 0; }
    }
    const int32_t n_ranges = (int32_t)((nnz + kRangeLen - 1) / kRangeLen);
    hb.range_seg.assign((size_t)n_ranges, 0);
    size_t s = 0;
    for (int32_t rho = 0; rho < n_ranges; ++rho) {
        const int32_t pos = rho * kRangeLen;
        while (s + 1 < nc && hb.cptr[s + 1] <= pos) ++s;
        hb.range_seg[(size_t)rho] = (int32_t)s;
    }
    hb.split_seg.clear();
    hb.split_short.clear();
    // the same predicates k_backward applies: a column spanning two ranges whose remainder in the
    // second is <= kExtend is finished by the first range's slot and needs no fixup; the others are
    // summed by k_fixup, a slot each when they span <= 8 ranges, else a wave each
    for (size_t c = 0; c < nc; ++c) {
        const int32_t ra = hb.cptr[c] / kRangeLen, rb = (hb.cptr[c + 1] - 1) / kRangeLen;
        if (rb > ra && !(rb == ra + 1 && hb.cptr[c + 1] - rb * kRangeLen <= kExtend))
            (rb - ra + 1 <= 8 ? hb.split_short : hb.split_seg).push_back((int32_t)c);
    }
}

// Band-affine placement of one batch's ranges (BwdArgs::xlist).  first/last: the rows of the first and last entry of every
// range.  A range that lies inside ONE column and spans at most a band and a half of rows is "affine" to the band of its
// middle row; XCD x owns a run of consecutive bands (two at 250k-row batches) and its list starts with their ranges, band by band,
// so that one band's slice of P (rows / 16 x 4 Kp bytes: 2 MB at 250k rows of Kp = 32) is what that XCD's L2 holds while
// they are walked; every other range is "free" and fills the lists up to equal length.  Returns the affine count.
int32_t plan_bands(const HostBatch &hb, int32_t cnnz, int64_t rows, const std::vector<int32_t> &first, const std::vector<int32_t> &last,
                   std::vector<int32_t> (&lists)[kXcds], int32_t (&seg)[kXcds][kXSegs + 1]) {
    const int32_t n_ranges = (int32_t)hb.range_seg.size();
    // bands of about 16k rows (2 MB of P at Kp = 32, 4 MB at Kp = 64: C3 and C5's width measured the same with 16 and 32
    // bands of 250k rows), a multiple of the XCD count, at most 8 per XCD
    int n_bands = (int)std::min<int64_t>(((rows + 16383) / 16384 + kXcds - 1) / kXcds * kXcds, (kXSegs - 1) * kXcds);
    n_bands = std::max(n_bands, kRowBands);
    if (const char *ev = getenv("FMHIP_ROW_BANDS")) {           // measurement knob: a multiple of kXcds, at most 8 per XCD
        const int v = atoi(ev);
        if (v >= kXcds && v <= (kXSegs - 1) * kXcds && v % kXcds == 0) n_bands = v;
    }
    const int per_xcd = n_bands / kXcds;
    const int64_t band_rows = std::max<int64_t>((rows + n_bands - 1) / n_bands, 1);
    std::vector<std::vector<int32_t>> by_band((size_t)n_bands);
    std::vector<int32_t> free_ranges;
    for (int32_t rho = 0; rho < n_ranges; ++rho) {
        const int32_t beg = rho * kRangeLen, end = std::min(beg + kRangeLen, cnnz);
        const int32_t seg = hb.range_seg[(size_t)rho];
        const bool one_column = hb.cptr[(size_t)seg] <= beg && hb.cptr[(size_t)seg + 1] >= end;
        const int64_t span = (int64_t)last[(size_t)rho] - first[(size_t)rho];
        if (one_column && end - beg == kRangeLen && span >= 0 && span * 2 <= band_rows * 3) {
            const int64_t band = std::min<int64_t>(((int64_t)first[(size_t)rho] + last[(size_t)rho]) / 2 / band_rows, n_bands - 1);
            by_band[(size_t)band].push_back(rho);
        } else {
            free_ranges.push_back(rho);
        }
    }
    int32_t affine = 0;
    for (int x = 0; x < kXcds; ++x) {
        lists[x].clear();
        for (int b = 0; b < kXSegs - 1; ++b) {                     // one run per band (runs of bands the XCD does not have: empty)
            seg[x][b] = (int32_t)lists[x].size();
            if (b >= per_xcd) continue;
            const auto &v = by_band[(size_t)(x * per_xcd + b)];
            lists[x].insert(lists[x].end(), v.begin(), v.end());
            affine += (int32_t)v.size();
        }
        seg[x][kXSegs - 1] = (int32_t)lists[x].size();            // the last run: this XCD's share of the other ranges
    }
    // The free ranges follow in blocks of 32 consecutive ranges, each block to the list that is shortest so far: close to the
    // round-robin of the default placement — every XCD gets hot (few columns per range) and cold (a flush per entry)
    // stretches of the stream alike; handing each XCD one contiguous eighth instead left the XCD with the coldest
    // features far behind the others (C4: backward 203 -> 268 us) — and the lists end within a block of each other.
    constexpr size_t kBlockRanges = 32;
    for (size_t next = 0; next < free_ranges.size(); next += kBlockRanges) {
        int best = 0;
        for (int x = 1; x < kXcds; ++x)
            if (lists[x].size() < lists[best].size()) best = x;
        const size_t hi = std::min(next + kBlockRanges, free_ranges.size());
        lists[best].insert(lists[best].end(), free_ranges.begin() + (std::ptrdiff_t)next, free_ranges.begin() + (std::ptrdiff_t)hi);
    }
    for (int x = 0; x < kXcds; ++x) seg[x][kXSegs] = (int32_t)lists[x].size();
    return affine;
}

// ALS level schedule (S/fm/lib/ALS.scala:36-70 walks the features in id order; two columns without a common row touch
// disjoint residuals and q entries, so their closed-form steps commute EXACTLY): one pass over the transpose in id order,
// level(c) = 1 + max over c's rows of the level of the last column that touched the row
int32_t als_levels(const std::vector<int32_t> &cptr, const uint32_t *crow, int64_t rows, std::vector<int32_t> &lev_ptr, std::vector<int32_t> &cols) {
    const size_t nc = cptr.empty() ? 0 : cptr.size() - 1;
    std::vector<int32_t> row_level((size_t)std::max<int64_t>(rows, 0), 0), level(nc, 0);
    int32_t n_levels = 0;
    for (size_t c = 0; c < nc; ++c) {
        int32_t lv = 0;
        for (int32_t p = cptr[c]; p < cptr[c + 1]; ++p) lv = std::max(lv, row_level[crow[(size_t)p] & 0x7fffffffu]);
        ++lv;
        level[c] = lv;
        n_levels = std::max(n_levels, lv);
        for (int32_t p = cptr[c]; p < cptr[c + 1]; ++p) row_level[crow[(size_t)p] & 0x7fffffffu] = lv;
    }
    lev_ptr.assign((size_t)n_levels + 1, 0);
    for (size_t c = 0; c < nc; ++c) ++lev_ptr[(size_t)level[c]];
    for (int32_t l = 1; l <= n_levels; ++l) lev_ptr[(size_t)l] += lev_ptr[(size_t)l - 1];
    cols.assign(nc, 0);
    std::vector<int32_t> at(lev_ptr.begin(), lev_ptr.end() - 1);
    for (size_t c = 0; c < nc; ++c) cols[(size_t)at[(size_t)level[c] - 1]++] = (int32_t)c;   // ascending id inside a level
    return n_levels;
}

void choose_cuts(const int32_t *cnt, int64_t n1, int n_fractions, const double *fractions, int64_t *cuts) {
    int64_t total = 0;
    for (int64_t f = 0; f < n1; ++f) total += cnt[f];
    int64_t above = 0, f = n1 - 1;
    for (int i = 0; i < n_fractions; ++i) {
        const double want = std::min(std::max(fractions[i], 0.0), 1.0) * (double)total;
        while (f > 0 && (double)above < want) above += cnt[(size_t)f--];
        cuts[i] = f + 1 < n1 ? f + 1 : 0;
    }
}

std::vector<int64_t> interval_edges(const std::vector<int64_t> &cuts, int64_t n1, int W) {
    std::vector<int64_t> edge{0};
    for (int64_t x : cuts) {
        const int64_t xr = W > 0 ? x / W * W : x;           // the plan rounds already; a plan made for another world may not have
        if (xr > edge.back() && xr < n1) edge.push_back(xr);
    }
    edge.push_back(n1);
    return edge;
}

Share shard_share(int64_t lo, int64_t hi, bool top_interval, int64_t n1, int W, int R) {
    Share s;
    s.hi_r = top_interval ? shard_top(n1, W) : hi;           // the top interval reaches into the slack rows
    s.chunk = (s.hi_r - lo) / W;
    s.vlo = lo + (int64_t)R * s.chunk;
    s.vhi = s.vlo + s.chunk;
    return s;
}

}  // namespace host
}  // namespace fmhip
