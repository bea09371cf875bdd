// fmhip_host.h — the library's PURE HOST ARITHMETIC: everything libfmhip.so computes on the host that needs no GPU and no
// HIP header — row shards, feature relabelling, a batch's range / fixup metadata, the band-affine plan of the backward's ranges,
// the ALS sweep's level schedule, the data-parallel plan's cuts, interval edges and equal shares.  fmhip_dataset.hip /
// fmhip_comm.hip call these; tests/host_arith_harness.cpp compiles fmhip_host.cpp with g++ -fsanitize=address,undefined and
// drives them over random shapes (the sanitizers cannot run on the GPU pool, the index arithmetic can run
// here).  Not installed, not part of the ABI.
#pragma once
#include <stdint.h>

#include <thread>
#include <vector>

#include "fm_constants.h"

namespace fmhip {
namespace host {

// ---- threads -------------------------------------------------------------------------------------------------
// host cores a pass over `work_items` items may use (FMHIP_HOST_THREADS caps it; at most 32, one per 65536 items)
int host_threads(int64_t work_items);

// f(tid, lo, hi) over [0, n) cut into one contiguous chunk per thread
template <class F>
void parallel_chunks(int64_t n, int threads, F f) {
    if (threads <= 1 || n <= 0) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads);
    for (int t = 0; t < threads; ++t) {
        const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
        pool.emplace_back([=]() { f(t, lo, hi); });
    }
    for (auto &th : pool) th.join();
}

// ---- row shards (fmhip_shard_rows) ------------------------------------------------------------------------------
// contiguous shard [lo, hi) of `rank` of `world`, balanced by stored nonzeros (row_ptr[n_rows + 1]); arguments validated by the caller
void shard_bounds(int64_t n_rows, const int64_t *row_ptr, int world, int rank, int64_t *lo, int64_t *hi);

// ---- feature relabelling by frequency (fmhip_feature_counts / _rank_from_counts / _relabel_columns) ----------
// counts[c] += occurrences of c; -> -1, or the index of the first id found outside [0, n1) (counts are then unreliable)
int64_t feature_counts(int64_t nnz, const int32_t *col, int64_t n1, int64_t *counts);
// rank[n1] (and by_rank[n1] unless NULL): position in descending count order, ties by ascending id
void rank_from_counts(int64_t n1, const int64_t *counts, int32_t *rank, int32_t *by_rank);
// out[i] = rank[col[i]]; -> -1, or the index of an id outside [0, n1)
int64_t relabel_columns(int64_t nnz, const int32_t *col, int64_t n1, const int32_t *rank, int32_t *out);

// ---- one batch's metadata from its column offsets --------------------------------------------------------------
struct HostBatch {
    std::vector<int32_t> cfeat, cptr, range_seg, split_seg, split_short, cdst, mp_feat, mp_ptr;
    int32_t n_feats = 0, n_pieces = 0;
};
// cfeat / cptr given: the column open at the start of every 64-entry range, the columns whose sum k_fixup assembles (short: up
// to 8 ranges, else long), the destination of every column piece.  cnt / base: zeroed scratch of dimension + 1 entries (left zeroed)
void finish_batch_meta(HostBatch &hb, int32_t nnz, std::vector<int32_t> &cnt, std::vector<int32_t> &base);

// Band-affine placement of one batch's ranges (BwdArgs::xlist).  first/last: the rows of the first and last entry of every
// range.  -> the number of ranges placed by their band; lists[x] = XCD x's range ids, seg[x] = the runs of lists[x]
int32_t plan_bands(const HostBatch &hb, int32_t cnnz, int64_t rows, const std::vector<int32_t> &first, const std::vector<int32_t> &last,
                   std::vector<int32_t> (&lists)[kXcds], int32_t (&seg)[kXcds][kXSegs + 1]);

// ---- ALS level schedule (S/fm/lib/ALS.scala:36-70 walks the features in id order) ------------------------------
// level(c) = 1 + the largest level of an earlier column sharing a row with c.  crow: the transpose's row ids (bit 31 = a flag,
// masked off), cptr[nc + 1].  -> number of levels; lev_ptr[levels + 1], cols[nc] = columns sorted by (level, id)
int32_t als_levels(const std::vector<int32_t> &cptr, const uint32_t *crow, int64_t rows, std::vector<int32_t> &lev_ptr, std::vector<int32_t> &cols);

// ---- the data-parallel plan (fmhip_dp_plan, fmhip_comm.hip) ------------------------------------------------------
// cnt[n1]: stored nonzeros per feature.  cuts[i] = the id at or above which the share fractions[i] (ascending) of them lies, 0 = collapsed
void choose_cuts(const int32_t *cnt, int64_t n1, int n_fractions, const double *fractions, int64_t *cuts);
// interval edges {0, cuts inside (0, n1) ascending and distinct ..., n1}; W > 0: every cut rounded DOWN to a multiple of W first
std::vector<int64_t> interval_edges(const std::vector<int64_t> &cuts, int64_t n1, int W);
// the sharded update's equal shares: top = n1 rounded up to a multiple of W; rank R's share of [lo, hi_r) (hi_r = top for the top interval)
inline int64_t shard_top(int64_t n1, int W) { return (n1 + W - 1) / W * W; }
struct Share { int64_t hi_r, chunk, vlo, vhi; };
Share shard_share(int64_t lo, int64_t hi, bool top_interval, int64_t n1, int W, int R);

}  // namespace host
}  // namespace fmhip
