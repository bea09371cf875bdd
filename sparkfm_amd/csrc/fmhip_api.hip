// fmhip_api.hip — host side of libfmhip.so: the C ABI declared in include/fmhip.h.
//
// Owns device memory, the per-batch transposes and the launch sequence of one mini-batch
// SGD step:   k_forward -> k_backward -> k_fixup (+ statistics) -> [host all-reduce] -> k_apply
// Everything here is plumbing; the arithmetic lives in fm_kernels.hip.
#include "fmhip_internal.h"
#include "als_kernels.h"
#include "csc_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

using namespace fmhip;

#ifndef FMHIP_FIN_BLOCKS
#define FMHIP_FIN_BLOCKS 2048     // cap on the merged finish's update workgroups (beside ~2k fixup workgroups at C3)
#endif

static_assert(FMHIP_HOT_PAGES == kHotPages && kHotPages * kHotT <= 128, "header and kernels disagree on the hot pages (128-bit slot masks)");
typedef unsigned __int128 slotmask_t;      // one bit per slot of the dense hot block
static_assert(FMHIP_RANGE_LEN == kRangeLen, "header and kernels disagree on the CSC range length");

namespace fmhip {
namespace host {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace host
}  // namespace fmhip

using namespace fmhip::host;

namespace fmhip {
namespace host {

struct ProfScope {
    fmhip_model *m;
    ProfRec r{};
    bool on;
    ProfScope(fmhip_model *m_, int kind, int64_t nnz, int64_t rows) : m(m_), on(m_->profiling) {
        if (on && m->prof_rotate) {
            static const int live[4] = {FMHIP_K_FORWARD, FMHIP_K_BACKWARD, FMHIP_K_FIXUP, FMHIP_K_APPLY};
            const int64_t period = m->prof_period > 0 ? m->prof_period : 1;
            if (m->prof_step % period != 0 || live[(m->prof_step / period) % 4] != kind) on = false;
        }
        if (!on) return;
        r.kind = kind;
        r.nnz = nnz;
        r.rows = rows;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.a, m->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, m->stream);
        m->prof.push_back(r);
    }
};

int set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return FMHIP_OK;
}

// ---- dataset construction -------------------------------------------------------

struct HostBatch {
    std::vector<int32_t> cfeat, cptr, range_seg, split_seg, split_short, cdst, mp_feat, mp_ptr;
    int32_t n_feats = 0, n_pieces = 0;
};

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

template <typename T>
int upload(DevBuf<T> &dst, const T *src, size_t n) {
    TRY(dst.alloc(n));
    if (n) HIP_TRY(hipMemcpy(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return FMHIP_OK;
}

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

// FMHIP_BUILD_TIMING=1: the phases of fmhip_dataset_create on stderr (what `DataSet.cache()` costs, and where)
struct PhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseTimer() : on(getenv("FMHIP_BUILD_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fmhip build] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

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

// scoring = true: rows + labels only (FMModel.predict / Model.computeRMSE on held-out data,
// S/driver.scala:100-112) — no transposes, no hot block, nothing a training step needs
// hot_opt: -1 = the process-wide defaults (fmhip_tune keys 5, 12), 0 = no hot block, n >= 1 = up to n pages of it;
// rb_opt: -1 = the default (key 3)
template <typename FT>
int dataset_create_impl(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const FT *val,
                        const FT *y, int64_t batch_rows, bool scoring, fmhip_dataset_t *out, int hot_opt = -1,
                        int64_t rb_opt = -1) {
    const bool want_hot = hot_opt < 0 ? g_tune[kTuneHot] > 0 : hot_opt > 0;
    const int max_hot_pages = std::max(1, std::min(kHotPages, hot_opt > 0 ? hot_opt : g_tune[kTuneHotPages]));
    const int64_t want_rb = rb_opt < 0 ? (g_tune[kTuneRowBlock] > 0 ? g_tune[kTuneRowBlock] : 0) : rb_opt;
    if (!out) return fail(FMHIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_rows < 0) return fail(FMHIP_ERR_INVALID, "n_rows < 0");
    if (!row_ptr) return fail(FMHIP_ERR_INVALID, "row_ptr is NULL");
    if (row_ptr[0] != 0) return fail(FMHIP_ERR_INVALID, "row_ptr[0] must be 0");
    const int T = host_threads(n_rows);
    PhaseTimer pt;
    {
        std::vector<int64_t> bad((size_t)T, -1);
        parallel_chunks(n_rows, T, [&](int t, int64_t lo, int64_t hi) {
            for (int64_t r = lo; r < hi; ++r)
                if (row_ptr[r + 1] < row_ptr[r]) { bad[(size_t)t] = r; break; }
        });
        for (int64_t r : bad)
            if (r >= 0) return fail(FMHIP_ERR_INVALID, "row_ptr decreases at row %lld", (long long)r);
    }
    const int64_t nnz = row_ptr[n_rows];
    if (nnz > 0 && (!col || !val)) return fail(FMHIP_ERR_INVALID, "col/val is NULL");
    if (n_rows > 0 && !y && !scoring) return fail(FMHIP_ERR_INVALID, "y is NULL");
    int32_t dim = 0;
    {
        const int Tn = host_threads(nnz);
        std::vector<int64_t> bad((size_t)Tn, -1);
        std::vector<int32_t> mx((size_t)Tn, 0);
        parallel_chunks(nnz, Tn, [&](int t, int64_t lo, int64_t hi) {
            int32_t m = 0;
            for (int64_t p = lo; p < hi; ++p) {
                if (col[p] < 0) { bad[(size_t)t] = p; break; }
                m = std::max(m, col[p]);
            }
            mx[(size_t)t] = m;
        });
        for (int64_t p : bad)
            if (p >= 0) return fail(FMHIP_ERR_INVALID, "negative feature index at entry %lld", (long long)p);
        for (int32_t m : mx) dim = std::max(dim, m);
    }
    pt.lap("validate");
    TRY(set_device(device));
    fmhip_dataset *d = new (std::nothrow) fmhip_dataset();
    if (!d) return fail(FMHIP_ERR_NOMEM, "out of host memory");
    d->device = device;
    d->n_rows = n_rows;
    d->nnz = nnz;
    d->dimension = dim;  // S/DataSet.scala:27-29
    d->scoring_only = scoring;
    if (scoring) batch_rows = 262144;   // bounds the forward's workspace; invisible to the caller
    if (batch_rows <= 0 || batch_rows > n_rows) batch_rows = std::max<int64_t>(n_rows, 1);
    d->batch_rows = batch_rows;
    const int64_t nb = n_rows > 0 ? (n_rows + batch_rows - 1) / batch_rows : 0;
    // ---- dense hot block (fmhip_tune keys 5, 12): features present in >= 10 % of the rows, the most frequent first, fill
    // up to `max_pages` pages of kHotT slots; x_rh sits in xhot[page][r][slot].  Page 0's entries leave the sparse
    // streams altogether; the entries of pages 1.. stay in the CSR stream (the forward walks them like any other entry)
    // and leave only the transposes (fm_kernels.h, kHotPages).  A feature that occurs twice in a row, or is stored with
    // an explicit zero, keeps the sparse path.  Single-batch datasets (the ALS learner walks their whole transpose) and
    // row-blocked ones (gradient-side pages) are never split.
    const int64_t *orig_row_ptr = row_ptr;
    std::vector<int64_t> sp_ptr;
    std::unique_ptr<int32_t[]> sp_col_buf;
    std::unique_ptr<float[]> sp_val_buf, xhot_buf;
    std::vector<slotmask_t> hot_masks;
    std::vector<int64_t> bwd_out;          // per batch: entries of the gradient-side pages (in the CSR, not in the CSC)
    std::vector<uint32_t> drop_bits;       // bitmap over feature ids: the gradient-side pages' features
    bool split = false;
    if (want_hot && nb > 1 && nnz > 0 && !scoring) {
        // Frequencies: exact for datasets of up to 8 M nonzeros; beyond that from every s-th row (the
        // choice of hot features is a layout decision — any set that passes the checks below is valid —
        // and a feature in >= 10 % of the rows cannot hide from a sample of millions of entries).
        const int64_t stride = nnz > ((int64_t)8 << 20) ? std::max<int64_t>(1, nnz / ((int64_t)4 << 20)) : 1;
        const int64_t sampled_rows = (n_rows + stride - 1) / stride;
        std::vector<int32_t> cnt((size_t)dim + 1, 0);
        {
            // per-thread histograms while they stay small (<= 256 MB in all), merged in thread order
            const int Ts = ((int64_t)(dim + 1) * T * 4 <= ((int64_t)256 << 20)) ? std::min<int>(T, (int)std::max<int64_t>(sampled_rows / 4096, 1)) : 1;
            std::vector<std::vector<int32_t>> part((size_t)(Ts > 1 ? Ts : 0));
            parallel_chunks(sampled_rows, Ts, [&](int t, int64_t lo, int64_t hi) {
                int32_t *c = cnt.data();
                if (Ts > 1) { part[(size_t)t].assign((size_t)dim + 1, 0); c = part[(size_t)t].data(); }
                for (int64_t i = lo; i < hi; ++i) {
                    const int64_t r = i * stride;
                    for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) ++c[(size_t)col[p]];
                }
            });
            if (Ts > 1)
                parallel_chunks((int64_t)dim + 1, Ts, [&](int, int64_t lo, int64_t hi) {
                    for (const auto &pc : part)
                        for (int64_t f = lo; f < hi; ++f) cnt[(size_t)f] += pc[(size_t)f];
                });
        }
        // candidates in descending order of frequency (ties: ascending id).  Page 0 is dense for the forward too, where a
        // slot costs every row a multiply-add chain: it takes features present in >= 10 % of the rows.  A gradient-side slot
        // costs a row 4 streamed bytes and saves, per entry, an 8-byte stream read, a P-row gather and an e gather (~2 line
        // requests of the texture path, which is what bounds the column walk): those pages take features down to 2.5 %.
        std::vector<int32_t> cand;
        for (int32_t f = 0; f <= dim; ++f)
            if ((int64_t)cnt[(size_t)f] * 40 >= sampled_rows) cand.push_back(f);
        std::sort(cand.begin(), cand.end(), [&](int32_t x, int32_t y) { return cnt[(size_t)x] != cnt[(size_t)y] ? cnt[(size_t)x] > cnt[(size_t)y] : x < y; });
        const size_t max_rest = (size_t)kHotT * (size_t)((want_rb > 0 ? 1 : max_hot_pages) - 1);
        std::vector<int8_t> slot((size_t)dim + 1, -1);
        sp_ptr.assign((size_t)n_rows + 1, 0);
        // pass 1 (one sweep): the CSR length of every row if page 0 leaves the streams, and which candidates may not be
        // dense — one that occurs twice in a row, or is stored with an explicit zero (its G row must have exactly one
        // writer); if any is refused the sweep runs again without it (the ranking moves up)
        size_t used = 0, p0 = 0;   // candidates being tried: the first p0 in page 0 (slots 0..), the next ones in slots kHotT..
        auto slot_of = [&](size_t j) { return (int8_t)(j < p0 ? j : kHotT + (j - p0)); };
        for (;;) {
            p0 = 0;
            while (p0 < cand.size() && p0 < (size_t)kHotT && (int64_t)cnt[(size_t)cand[p0]] * 10 >= sampled_rows) ++p0;
            used = p0 < 2 ? 0 : p0 + std::min(cand.size() - p0, max_rest);
            if (!used) break;
            for (size_t j = 0; j < used; ++j) slot[(size_t)cand[j]] = slot_of(j);
            std::vector<slotmask_t> badv((size_t)T, 0u);
            parallel_chunks(n_rows, T, [&](int t, int64_t lo, int64_t hi) {
                slotmask_t bad = 0;
                for (int64_t r = lo; r < hi; ++r) {
                    slotmask_t seen = 0;
                    int64_t keep = 0;
                    for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
                        const int8_t h = slot[(size_t)col[p]];
                        if (h < 0 || h >= kHotT) ++keep;
                        if (h < 0) continue;
                        if ((seen >> h & 1u) || (float)val[p] == 0.f) bad |= (slotmask_t)1 << h;
                        seen |= (slotmask_t)1 << h;
                    }
                    sp_ptr[(size_t)r + 1] = keep;
                }
                badv[(size_t)t] = bad;
            });
            slotmask_t bad = 0;
            for (slotmask_t x : badv) bad |= x;
            if (!bad) break;
            std::vector<int32_t> ok;
            for (size_t j = 0; j < cand.size(); ++j) {
                if (j < used) slot[(size_t)cand[j]] = -1;
                if (j >= used || !(bad >> slot_of(j) & 1u)) ok.push_back(cand[j]);
            }
            cand.swap(ok);
        }
        std::vector<int32_t>().swap(cnt);
        cand.resize(used);
        if (used) {
            // slots in ascending feature order inside every page (the sweep above does not depend on the numbering)
            const int pages = 1 + (int)((used - p0 + kHotT - 1) / kHotT);
            d->hot_ids.assign((size_t)(pages * kHotT), -1);
            std::sort(cand.begin(), cand.begin() + (std::ptrdiff_t)p0);
            for (size_t j = 0; j < p0; ++j) { d->hot_ids[j] = cand[j]; slot[(size_t)cand[j]] = (int8_t)j; }
            for (size_t lo = p0; lo < used; lo += kHotT) {
                const size_t hi = std::min(used, lo + kHotT);
                std::sort(cand.begin() + (std::ptrdiff_t)lo, cand.begin() + (std::ptrdiff_t)hi);
                for (size_t j = lo; j < hi; ++j) {
                    const size_t h = kHotT + (j - p0);
                    d->hot_ids[h] = cand[j];
                    slot[(size_t)cand[j]] = (int8_t)h;
                }
            }
            if (pages > 1) {
                drop_bits.assign((size_t)(dim + 1) / 32 + 2, 0u);
                for (size_t j = p0; j < used; ++j) drop_bits[(size_t)cand[j] >> 5] |= 1u << (cand[j] & 31);
            }
            hot_masks.assign((size_t)nb, 0u);
            bwd_out.assign((size_t)nb, 0);
            for (int64_t r = 0; r < n_rows; ++r) sp_ptr[(size_t)r + 1] += sp_ptr[(size_t)r];
            // pass 2: fill (buffers left uninitialised: every element is written exactly once)
            const size_t page_floats = (size_t)std::max<int64_t>(n_rows, 1) * kHotT;
            sp_col_buf.reset(new int32_t[(size_t)std::max<int64_t>(sp_ptr[(size_t)n_rows], 1)]);
            sp_val_buf.reset(new float[(size_t)std::max<int64_t>(sp_ptr[(size_t)n_rows], 1)]);
            xhot_buf.reset(new float[page_floats * (size_t)pages]);
            int32_t *sp_col = sp_col_buf.get();
            float *sp_val = sp_val_buf.get(), *xhot = xhot_buf.get();
            std::vector<std::vector<slotmask_t>> tmask((size_t)T, std::vector<slotmask_t>((size_t)nb, 0u));
            std::vector<std::vector<int64_t>> tout((size_t)T, std::vector<int64_t>((size_t)nb, 0));
            parallel_chunks(n_rows, T, [&](int t, int64_t lo, int64_t hi) {
                for (int64_t r = lo; r < hi; ++r) {
                    slotmask_t seen = 0;
                    int64_t o = sp_ptr[(size_t)r], outb = 0;
                    for (int pg = 0; pg < pages; ++pg) {
                        float *xr = xhot + (size_t)pg * page_floats + (size_t)r * kHotT;
                        for (int h = 0; h < kHotT; ++h) xr[h] = 0.f;
                    }
                    for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
                        const int8_t h = slot[(size_t)col[p]];
                        if (h >= 0) {
                            seen |= (slotmask_t)1 << h;
                            xhot[(size_t)(h / kHotT) * page_floats + (size_t)r * kHotT + (h % kHotT)] = (float)val[p];
                        }
                        if (h < 0 || h >= kHotT) {
                            sp_col[(size_t)o] = col[p];
                            sp_val[(size_t)o] = (float)val[p];
                            ++o;
                            if (h >= 0) ++outb;
                        }
                    }
                    tmask[(size_t)t][(size_t)(r / batch_rows)] |= seen;
                    tout[(size_t)t][(size_t)(r / batch_rows)] += outb;
                }
            });
            for (int t = 0; t < T; ++t)
                for (int64_t b = 0; b < nb; ++b) {
                    hot_masks[(size_t)b] |= tmask[(size_t)t][(size_t)b];
                    bwd_out[(size_t)b] += tout[(size_t)t][(size_t)b];
                }
            split = true;
            d->hot_T = kHotT;
            d->hot_pages = pages;
            for (int32_t f : d->hot_ids) d->hot_max_id = std::max(d->hot_max_id, (int64_t)f);
        }
    }
    pt.lap("hot block: choose + split");
    if (split) {
        row_ptr = sp_ptr.data();
        col = sp_col_buf.get();
    }
    const int64_t nnz_s = split ? sp_ptr[(size_t)n_rows] : nnz;
    d->nnz_sparse = nnz_s;
    d->batches.resize((size_t)nb);
    for (int64_t b = 0; b < nb; ++b) {
        BatchMeta &bm = d->batches[(size_t)b];
        bm.row0 = b * batch_rows;
        bm.rows = std::min(batch_rows, n_rows - bm.row0);
        bm.nnz0 = row_ptr[bm.row0];
        bm.nnz_total = orig_row_ptr[bm.row0 + bm.rows] - orig_row_ptr[bm.row0];
        bm.hot_mask = split ? hot_masks[(size_t)b] : 0u;
        const int64_t bn = row_ptr[bm.row0 + bm.rows] - bm.nnz0;
        d->nnz_sparse_bwd += bn - (split ? bwd_out[(size_t)b] : 0);
        if (bn > (int64_t)0x7fffffff - 2 * kRangeLen || bm.rows > 0x7fffffff) {
            delete d;
            return fail(FMHIP_ERR_UNSUPPORTED, "batch %lld holds %lld nonzeros; the per-batch limit is 2^31", (long long)b,
                        (long long)bn);
        }
        bm.nnz = (int32_t)bn;
        bm.cnnz = (int32_t)(bn - (split ? bwd_out[(size_t)b] : 0));
        d->max_rows = std::max(d->max_rows, bm.rows);
    }
    // forward walk order of each batch: rows by (sparse) length, longest first, ties in row order
    {
        std::vector<int32_t> order((size_t)n_rows);
        parallel_chunks(nb, std::min<int>(T, (int)std::max<int64_t>(nb, 1)), [&](int, int64_t blo, int64_t bhi) {
            std::vector<int64_t> start;
            for (int64_t b = blo; b < bhi; ++b) {
                const BatchMeta &bm = d->batches[(size_t)b];
                int64_t maxlen = 0;
                for (int64_t r = 0; r < bm.rows; ++r) maxlen = std::max(maxlen, row_ptr[bm.row0 + r + 1] - row_ptr[bm.row0 + r]);
                start.assign((size_t)maxlen + 2, 0);
                for (int64_t r = 0; r < bm.rows; ++r) ++start[(size_t)(maxlen - (row_ptr[bm.row0 + r + 1] - row_ptr[bm.row0 + r])) + 1];
                for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
                for (int64_t r = 0; r < bm.rows; ++r) {
                    const size_t key = (size_t)(maxlen - (row_ptr[bm.row0 + r + 1] - row_ptr[bm.row0 + r]));
                    order[(size_t)(bm.row0 + start[key]++)] = (int32_t)r;
                }
            }
        });
        const int rc0 = upload(d->row_order, order.data(), order.size());
        if (rc0) {
            delete d;
            return rc0;
        }
    }
    pt.lap("row order + upload");
    // fp32 copies of the streams (device arithmetic is fp32)
    std::vector<float> valf, yf((size_t)n_rows, 0.f);
    const float *val_up = nullptr;
    if (split) {
        val_up = sp_val_buf.get();
    } else if (std::is_same<FT, float>::value) {
        val_up = reinterpret_cast<const float *>(val);
    } else {
        valf.resize((size_t)nnz_s);
        parallel_chunks(nnz_s, host_threads(nnz_s), [&](int, int64_t lo, int64_t hi) {
            for (int64_t p = lo; p < hi; ++p) valf[(size_t)p] = (float)val[p];
        });
        val_up = valf.data();
    }
    if (y)
        for (int64_t r = 0; r < n_rows; ++r) yf[(size_t)r] = (float)y[r];
    const bool keep64 = !split && !scoring && nb == 1 && nnz <= kAlsMaxNnz;
    int rc = FMHIP_OK;
    if ((rc = upload(d->row_ptr, row_ptr, (size_t)n_rows + 1)) || (rc = upload(d->col, col, (size_t)nnz_s)) ||
        (rc = upload(d->val, val_up, (size_t)nnz_s)) || (rc = upload(d->y, yf.data(), (size_t)n_rows)) ||
        (!scoring && ((rc = d->crow.alloc((size_t)nnz_s)) || (rc = d->cval.alloc((size_t)nnz_s)))) ||
        (split && ((rc = upload(d->xhot, xhot_buf.get(), (size_t)std::max<int64_t>(n_rows, 1) * kHotT * (size_t)d->hot_pages)) ||
                   (rc = upload(d->d_hot_ids, d->hot_ids.data(), d->hot_ids.size()))))) {
        delete d;
        return rc;
    }
    std::vector<float>().swap(valf);
    xhot_buf.reset();
    sp_val_buf.reset();
    pt.lap("fp32 re-pack + H2D");
    if (scoring) {
        *out = d;
        return FMHIP_OK;
    }
    if (keep64) {
        std::vector<double> val64((size_t)nnz), y64((size_t)n_rows);
        for (int64_t p = 0; p < nnz; ++p) val64[(size_t)p] = (double)val[p];
        for (int64_t r = 0; r < n_rows; ++r) y64[(size_t)r] = (double)y[r];
        // feature-sorted copy of every row (stable: equal ids keep their stored order)
        std::vector<int32_t> scol((size_t)nnz);
        std::vector<double> sval((size_t)nnz);
        std::vector<char> dupv((size_t)T, 0);      // one byte per thread (vector<bool> packs bits: concurrent writes would race)
        parallel_chunks(n_rows, T, [&](int t, int64_t lo, int64_t hi) {
            std::vector<int32_t> idx;
            bool dup = false;
            for (int64_t r = lo; r < hi; ++r) {
                const int64_t p0 = row_ptr[r], len = row_ptr[r + 1] - p0;
                idx.resize((size_t)len);
                for (int64_t j = 0; j < len; ++j) idx[(size_t)j] = (int32_t)j;
                std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y2) { return col[p0 + x] < col[p0 + y2]; });
                for (int64_t j = 0; j < len; ++j) {
                    scol[(size_t)(p0 + j)] = col[p0 + idx[(size_t)j]];
                    sval[(size_t)(p0 + j)] = val64[(size_t)(p0 + idx[(size_t)j])];
                    if (j && scol[(size_t)(p0 + j)] == scol[(size_t)(p0 + j - 1)]) dup = true;
                }
            }
            dupv[(size_t)t] = dup ? 1 : 0;
        });
        for (char b : dupv) d->als_dup = d->als_dup || b != 0;
        if ((rc = upload(d->val64, val64.data(), val64.size())) || (rc = upload(d->y64, y64.data(), y64.size())) ||
            (rc = d->cval64.alloc((size_t)nnz)) || (rc = upload(d->scol, scol.data(), scol.size())) ||
            (rc = upload(d->sval64, sval.data(), sval.size()))) {
            delete d;
            return rc;
        }
    }
    // per-batch row -> column transposes, built on the device (csc_build.hip); only the small
    // column index (offsets, feature ids) comes back to the host
    std::vector<HostBatch> hbs((size_t)nb);
    {
        int32_t max_nnz = 0;
        for (const BatchMeta &bm : d->batches) max_nnz = std::max(max_nnz, bm.nnz);
        const size_t max_cols = (size_t)max_nnz;   // row-blocked streams repeat a feature once per block
        // gradient-side hot pages: their entries are keyed dim + 1 and sort behind every real column
        const bool drop = !drop_bits.empty();
        const int32_t drop_key = dim + 1;
        DevBuf<uint32_t> d_drop;
        if (drop && (rc = upload(d_drop, drop_bits.data(), drop_bits.size()))) {
            delete d;
            return rc;
        }
        int key_bits = 1;
        while (key_bits < 31 && ((int64_t)1 << key_bits) <= (int64_t)dim + (drop ? 1 : 0)) ++key_bits;
        // optional row blocking of the transposes (fmhip_tune key 3): entries sorted by (row block,
        // feature) so that a block's slice of P stays L2-resident while its columns are walked
        int64_t rb_rows = want_rb;
        int rb_bits = 0;
        if (rb_rows > 0) {
            const int64_t blocks = (d->max_rows + rb_rows - 1) / rb_rows;
            while (((int64_t)1 << rb_bits) < blocks) ++rb_bits;
            if (key_bits + rb_bits > 31) { rb_rows = 0; rb_bits = 0; }
        }
        d->rb_rows = rb_bits > 0 ? rb_rows : 0;
        const int32_t rb_div = d->rb_rows > 0 ? (int32_t)std::min<int64_t>(d->rb_rows, INT32_MAX) : INT32_MAX;
        std::vector<int32_t> cnt((size_t)dim + 2, 0), base((size_t)dim + 2, 0);
        DevBuf<int32_t> keys_a, keys_b, rowid, starts, feats, count;
        DevBuf<uint32_t> idx_a, idx_b;
        DevBuf<uint8_t> flags, tmp;
        CscScratch sc;
        size_t tmp_bytes = 0;
        hipError_t he = max_nnz ? csc_scratch_bytes((size_t)max_nnz, key_bits, &tmp_bytes) : hipSuccess;
        if (he != hipSuccess) {
            delete d;
            return fail(FMHIP_ERR_HIP, "rocPRIM scratch query failed: %s", hipGetErrorString(he));
        }
        if ((rc = keys_a.alloc((size_t)max_nnz)) || (rc = keys_b.alloc((size_t)max_nnz)) || (rc = idx_a.alloc((size_t)max_nnz)) ||
            (rc = idx_b.alloc((size_t)max_nnz)) || (rc = rowid.alloc((size_t)max_nnz)) || (rc = flags.alloc((size_t)max_nnz)) ||
            (rc = starts.alloc(max_cols + 1)) || (rc = feats.alloc(max_cols + 1)) || (rc = count.alloc(1)) ||
            (rc = tmp.alloc(tmp_bytes + 16))) {
            delete d;
            return rc;
        }
        sc.keys_a = keys_a.p; sc.keys_b = keys_b.p; sc.idx_a = idx_a.p; sc.idx_b = idx_b.p; sc.rowid = rowid.p;
        sc.flags = flags.p; sc.starts = starts.p; sc.feats = feats.p; sc.count = count.p; sc.tmp = tmp.p; sc.tmp_bytes = tmp_bytes;
        for (int64_t b = 0; b < nb; ++b) {
            const BatchMeta &bm = d->batches[(size_t)b];
            HostBatch &hb = hbs[(size_t)b];
            he = csc_build_batch(nullptr, sc, d->row_ptr.p, d->col.p, d->val.p, keep64 ? d->val64.p : nullptr, bm.row0, bm.rows,
                                 bm.nnz0, bm.nnz, key_bits, rb_div, rb_bits, d->crow.p, d->cval.p, keep64 ? d->cval64.p : nullptr,
                                 drop ? d_drop.p : nullptr, drop_key);
            int32_t nc = 0;
            if (he == hipSuccess) he = hipMemcpy(&nc, sc.count, sizeof nc, hipMemcpyDeviceToHost);
            if (he == hipSuccess) {
                hb.cfeat.resize((size_t)nc);
                hb.cptr.resize((size_t)nc + 1);
                if (nc) {
                    he = hipMemcpy(hb.cfeat.data(), sc.feats, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToHost);
                    if (he == hipSuccess) he = hipMemcpy(hb.cptr.data(), sc.starts, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToHost);
                }
                hb.cptr[(size_t)nc] = bm.nnz;
                if (he == hipSuccess && drop && nc > 0 && hb.cfeat[(size_t)nc - 1] == drop_key) {
                    // the pseudo-column of the dropped entries: the stream ends where it starts
                    hb.cfeat.pop_back();
                    hb.cptr.pop_back();
                }
                if (he == hipSuccess && hb.cptr.back() != bm.cnnz) {
                    delete d;
                    return fail(FMHIP_ERR_HIP, "batch %lld: the transpose holds %d entries, the host counted %d", (long long)b,
                                hb.cptr.back(), bm.cnnz);
                }
            }
            if (he != hipSuccess) {
                delete d;
                return fail(FMHIP_ERR_HIP, "device transpose of batch %lld failed: %s", (long long)b, hipGetErrorString(he));
            }
            finish_batch_meta(hb, bm.cnnz, cnt, base);
        }
    }
    pt.lap("device transposes + metadata");
    // bitmap of the features whose gradient rows the fixup launch assembles (cut columns + hot block), per batch:
    // the merged finish skips them in its dense pass.  Kept for models of up to 2^24 features (2 MiB per batch).
    std::vector<uint32_t> own;
    if (d->rb_rows == 0 && dim < (1 << 24) && nb * ((int64_t)dim / 32 + 1) <= ((int64_t)1 << 26)) {
        d->own_words = (int64_t)dim / 32 + 1;
        own.assign((size_t)(nb * d->own_words), 0u);
        for (int64_t b = 0; b < nb; ++b) {
            uint32_t *bits = own.data() + (size_t)(b * d->own_words);
            const HostBatch &hb = hbs[(size_t)b];
            for (const std::vector<int32_t> *lst : {&hb.split_seg, &hb.split_short})
                for (int32_t c : *lst) { const int32_t f = hb.cfeat[(size_t)c]; bits[f >> 5] |= 1u << (f & 31); }
            for (int32_t f : d->hot_ids)
                if (f >= 0) bits[f >> 5] |= 1u << (f & 31);
            d->batches[(size_t)b].own_off = b * d->own_words;
        }
    }
    std::vector<int32_t> cfeat, cptr, range_seg, split_seg, split_short, cdst, mp_feat, mp_ptr;
    for (int64_t b = 0; b < nb; ++b) {
        BatchMeta &bm = d->batches[(size_t)b];
        HostBatch &hb = hbs[(size_t)b];
        bm.n_feats = hb.n_feats;
        bm.n_mp = (int32_t)hb.mp_feat.size();
        bm.mp_off = (int64_t)mp_feat.size();
        bm.n_pieces = hb.n_pieces;
        d->max_pieces = std::max(d->max_pieces, bm.n_pieces);
        cdst.insert(cdst.end(), hb.cdst.begin(), hb.cdst.end());
        mp_feat.insert(mp_feat.end(), hb.mp_feat.begin(), hb.mp_feat.end());
        mp_ptr.insert(mp_ptr.end(), hb.mp_ptr.begin(), hb.mp_ptr.end());
        bm.n_cols = (int32_t)hb.cfeat.size();
        bm.col_off = (int64_t)cfeat.size();
        bm.n_ranges = (int32_t)hb.range_seg.size();
        bm.range_off = (int64_t)range_seg.size();
        bm.n_split = (int32_t)hb.split_seg.size();
        bm.split_off = (int64_t)split_seg.size();
        bm.n_split_short = (int32_t)hb.split_short.size();
        bm.split_short_off = (int64_t)split_short.size();
        split_short.insert(split_short.end(), hb.split_short.begin(), hb.split_short.end());
        d->max_ranges = std::max(d->max_ranges, bm.n_ranges);
        cfeat.insert(cfeat.end(), hb.cfeat.begin(), hb.cfeat.end());
        cptr.insert(cptr.end(), hb.cptr.begin(), hb.cptr.end());
        range_seg.insert(range_seg.end(), hb.range_seg.begin(), hb.range_seg.end());
        split_seg.insert(split_seg.end(), hb.split_seg.begin(), hb.split_seg.end());
        HostBatch().cfeat.swap(hb.cfeat);
    }
    d->h_cfeat = cfeat;
    d->h_cptr = cptr;
    d->h_split = split_seg;
    d->h_split_short = split_short;
    if ((rc = upload(d->cfeat, cfeat.data(), cfeat.size())) || (rc = upload(d->cptr, cptr.data(), cptr.size())) ||
        (rc = upload(d->range_seg, range_seg.data(), range_seg.size())) ||
        (rc = upload(d->split_seg, split_seg.data(), split_seg.size())) ||
        (rc = upload(d->split_short, split_short.data(), split_short.size())) || (rc = upload(d->cdst, cdst.data(), cdst.size())) ||
        (rc = upload(d->mp_feat, mp_feat.data(), mp_feat.size())) || (rc = upload(d->mp_ptr, mp_ptr.data(), mp_ptr.size())) ||
        (rc = upload(d->own_bits, own.data(), own.size()))) {
        delete d;
        return rc;
    }
    pt.lap("pack + upload column index");
    *out = d;
    return FMHIP_OK;
}

// ---- model helpers ----------------------------------------------------------------

int check_pair(fmhip_model_t m, fmhip_dataset_t d) {
    if (!m || !d) return fail(FMHIP_ERR_INVALID, "model or dataset is NULL");
    if (m->device != d->device) return fail(FMHIP_ERR_INVALID, "model on device %d, dataset on device %d", m->device, d->device);
    if (d->dimension > m->n)
        return fail(FMHIP_ERR_SHAPE, "dataset has feature index %lld but the model has num_attribute = %lld",
                    (long long)d->dimension, (long long)m->n);
    return set_device(m->device);
}

// training calls need the transposes a scoring-only dataset does not have
int check_train(fmhip_model_t m, fmhip_dataset_t d) {
    TRY(check_pair(m, d));
    if (d->scoring_only)
        return fail(FMHIP_ERR_UNSUPPORTED, "dataset was created with fmhip_rows_create (scoring only): it has no transposes to train on");
    return FMHIP_OK;
}

int check_batch(fmhip_dataset_t d, int64_t batch) {
    if (batch < 0 || batch >= (int64_t)d->batches.size())
        return fail(FMHIP_ERR_INVALID, "batch %lld out of range [0, %zu)", (long long)batch, d->batches.size());
    return FMHIP_OK;
}

int ensure_workspace(fmhip_model_t m, fmhip_dataset_t d) {
    TRY(m->P.ensure((size_t)std::max<int64_t>(d->max_rows, 1) * m->Kp));
    TRY(m->e.ensure((size_t)std::max<int64_t>(d->max_rows, 1)));
    TRY(m->part.ensure((size_t)std::max<int32_t>(d->max_ranges, 1) * 2 * (m->Kp + kPartPad)));
    TRY(m->pieces.ensure((size_t)std::max<int32_t>(d->max_pieces, 1) * (m->Kp + kPartPad)));
    TRY(m->bsum.ensure((size_t)kMaxFwdBlocks * 4));
    if (d->hot_T) TRY(m->hot_part.ensure((size_t)hot_blocks(m->Kp, d->max_rows) * d->hot_pages * kHotT * (m->Kp + kPartPad)));
    return FMHIP_OK;
}

FwdArgs fwd_args(fmhip_model_t m, fmhip_dataset_t d, const BatchMeta &bm) {
    FwdArgs a{};
    a.row_ptr = d->row_ptr.p;
    a.col = d->col.p;
    a.val = d->val.p;
    a.y = d->y.p;
    a.V = m->V.p;
    {
        // tables of 4 GiB and more do not fit a 32-bit buffer view and take the flat-address kernels
        // (fmhip_tune key 8 forces those for any size, so that tests reach them on small inputs)
        const uint64_t vb = (uint64_t)m->n1p * m->Kp * sizeof(float);
        a.v_bytes = (vb < 0xffffffffull && !m->tv(kTuneFlat)) ? (uint32_t)vb : 0u;
    }
    a.sv = (float)m->sv;
    a.sw = (float)m->sw;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.row0 = bm.row0;
    // longest-first row order: pays for wide rows only (k=64: -8 %); at Kp = 32 it changes nothing but the
    // locality of the per-row streams (forward FETCH_SIZE 184 -> 269 MB), so narrow models walk in stored order
    a.order = (m->tv(kTuneRowOrder) && m->Kp >= 64) ? d->row_order.p + bm.row0 : nullptr;
    a.n_rows = (int32_t)bm.rows;
    a.P = m->P.p;
    a.e = m->e.p;
    a.yhat = nullptr;
    a.pack_k = m->pack_k();
    a.hot_T = d->hot_T;
    a.xhot = d->hot_T ? d->xhot.p + (size_t)bm.row0 * kHotT : nullptr;
    a.hot_ids = d->d_hot_ids.p;
    a.bsum = m->bsum.p;
    {
        // LDS V-tile size: as many hot rows as fit 128 KiB (+ their w), capped by the model
        int64_t t = (128 * 1024) / ((int64_t)m->Kp * 4);
        if (m->tv(kTuneTile) > 0) t = m->tv(kTuneTile);
        a.tile_rows = (int32_t)std::min<int64_t>(t, m->n1);
        a.wt_rows = (int32_t)std::min<int64_t>(m->tv(kTuneTile) > 0 ? m->tv(kTuneTile) : 6144, m->n1);   // 24 KiB
        a.variant = m->tv(kTuneFwd);
        a.occ_cap = m->tv(kTuneFwdOcc);
    }
    return a;
}

BwdArgs bwd_args(fmhip_model_t m, fmhip_dataset_t d, int64_t b) {
    const BatchMeta &bm = d->batches[(size_t)b];
    BwdArgs a{};
    a.crow = d->crow.p + bm.nnz0;
    a.cval = d->cval.p + bm.nnz0;
    a.range_seg = d->range_seg.p + bm.range_off;
    a.cfeat = d->cfeat.p + bm.col_off;
    a.cdst = d->cdst.p + bm.col_off;
    a.pieces = m->pieces.p;
    a.mp_feat = d->mp_feat.p + bm.mp_off;
    a.mp_ptr = d->mp_ptr.p + bm.mp_off + b;
    a.n_mp = bm.n_mp;
    a.pack_k = m->pack_k();
    a.cptr = d->cptr.p + bm.col_off + b;
    a.split_seg = d->split_seg.p + bm.split_off;
    a.split_short = d->split_short.p + bm.split_short_off;
    a.n_split_short = bm.n_split_short;
    a.nnz = bm.cnnz;
    a.n_ranges = bm.n_ranges;
    a.rho_lo = 0;
    a.rho_hi = bm.n_ranges;
    a.xcd_chunk = m->tv(kTuneXcd) > 0 ? 1 : 0;
    a.pipelined = m->tv(kTuneBwd);
    a.n_split = bm.n_split;
    a.P = m->P.p;
    {
        const uint64_t pb = (uint64_t)bm.rows * m->Kp * sizeof(float);
        a.p_bytes = (pb < 0xffffffffull && !m->tv(kTuneFlat)) ? (uint32_t)pb : 0u;
    }
    a.e = m->e.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    if (m->view) {     // the rows go to a compact buffer (touched-rows exchange): column s -> row view->cdst[s]
        a.GV = m->view->GV;
        a.Gw = m->view->Gw;
        a.Gb = m->view->Gb;
        a.cdst = m->view->cdst;
    }
    a.part = m->part.p;
    return a;
}

// forward of one batch: P = e*q, e, per-block statistics partials
int step_forward(fmhip_model_t m, fmhip_dataset_t d, int64_t b) {
    const BatchMeta &bm = d->batches[(size_t)b];
    TRY(ensure_workspace(m, d));
    if (m->grad_dirty) {
        HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
        m->grad_dirty = false;
    }
    {
        ProfScope ps(m, FMHIP_K_FORWARD, bm.nnz_total, bm.rows);
        HIP_TRY(launch_forward(m->Kp, kFwdTrain, fwd_args(m, d, bm), m->stream, &m->fwd_parts));
    }
    m->grad_dirty = true;
    m->last_nnz = bm.nnz_total;
    m->last_rows = bm.rows;
    m->bw_next_hi = INT64_MAX;
    m->hot_pending = d->hot_T > 0;
    return FMHIP_OK;
}

// gradient rows of the dense hot block (whole batch; they do not depend on the feature interval, so
// the first backward call of a step forms them and every later interval finds them complete): the
// work rides in that call's backward and fixup launches
void hot_attach(fmhip_model_t m, fmhip_dataset_t d, const BatchMeta &bm, BwdArgs &ba) {
    if (!m->hot_pending) return;
    HotArgs &h = ba.hot;
    h.P = m->P.p;
    h.e = m->e.p;
    h.xhot = d->xhot.p + (size_t)bm.row0 * kHotT;
    h.page_stride = std::max<int64_t>(d->n_rows, 1) * kHotT;
    h.pages = d->hot_pages;
    h.hot_ids = m->view ? m->view->hot_pos : d->d_hot_ids.p;
    h.part = m->hot_part.p;
    h.GV = ba.GV;
    h.Gw = ba.Gw;
    h.Gb = ba.Gb;
    h.n_rows = (int32_t)bm.rows;
    h.pack_k = m->pack_k();
    h.nblk = hot_blocks(m->Kp, bm.rows);
    h.upd = ba.upd;
    ba.hot_blocks = h.nblk;
    m->hot_pending = false;
}

// backward + fixup of the columns whose feature id lies in [feat_lo, feat_hi) into the packed
// gradient.  The CSC stream is sorted by feature, so the interval is a contiguous run of entries;
// the range holding its first entry is walked by THIS call in full (the entries of lower features
// in it produce G rows / head partials that the call covering them consumes later), the range
// holding the first entry of feat_hi is left to the call that covers feat_hi.  `finish` adds the
// residual-statistics reduction (once per step, with the last interval).
int step_backward(fmhip_model_t m, fmhip_dataset_t d, int64_t b, int64_t feat_lo, int64_t feat_hi, bool finish,
                  double *acc, const FusedPlan *fused) {
    const BatchMeta &bm = d->batches[(size_t)b];
    BwdArgs ba = bwd_args(m, d, b);
    if (fused) {
        if (fused->mode == 1) ba.upd = fused->upd;
        if (finish) { ba.red_w0 = m->w0.p; ba.red_eta = (float)fused->eta; ba.red_reg0 = (float)fused->reg0; }
    }
    const bool whole = feat_lo <= 0 && feat_hi >= m->n1;
    // the block product rides in the first call whose interval reaches down to the highest hot id (callers that cut the
    // backward go from the top down: every hot row is complete before the interval holding it is exchanged, and the
    // cold intervals in front — whose exchange the rest of the backward hides — are not held up by it)
    if (whole || finish || d->hot_max_id >= feat_lo) hot_attach(m, d, bm, ba);
    if (d->rb_rows > 0 && !whole)
        return fail(FMHIP_ERR_UNSUPPORTED, "feature-interval backward is not available on a row-blocked dataset");
    if (whole) {   // the common case needs no host-side searches
        if (finish) {
            ba.red_bsum = m->bsum.p;
            ba.red_nblocks = m->fwd_parts;
            ba.red_rows = (int32_t)bm.rows;
            ba.red_scal = m->view ? m->view->scal : m->scal();
            ba.red_acc = acc;
        }
        {
            ProfScope ps(m, FMHIP_K_BACKWARD, bm.nnz_total, bm.rows);
            HIP_TRY(launch_backward(m->Kp, ba, m->stream));
        }
        if (fused && fused->mode == 2) {
            // merged finish: the fixup launch also updates the parameters (its own rows from registers, the rest in
            // extra workgroups beside it); the column walk above stored its gradient rows as usual
            ApplyArgs &f = ba.fin;
            f.V = m->V.p;
            f.w = m->w.p;
            f.w0 = m->w0.p;
            f.GV = m->GV();
            f.Gw = m->Gw();
            f.Gb = m->Gb();
            f.scal = m->scal();
            f.rows = m->scal() + 2;
            f.n1 = m->n1;
            f.row_lo = 0;
            f.row_hi = m->n1;
            f.do_w0 = 0;                                   // the statistics block steps w0 (red_w0)
            f.pack_k = m->pack_k();
            f.eta = (float)fused->eta;
            f.reg0 = (float)fused->reg0;
            f.regw = (float)fused->regw;
            f.regv = (float)fused->regv;
            f.sv_in = (float)m->sv;
            f.sw_in = (float)m->sw;
            f.eta_v = f.eta_w = f.eta;
            f.invb_val = fused->upd.invb;
            f.use_invb_val = 1;
            int64_t blocks = (m->n1 * (m->Kp / 4) + 255) / 256;
            ba.fin_blocks = (int32_t)std::min<int64_t>(std::max<int64_t>(blocks, 1), FMHIP_FIN_BLOCKS);
            ba.fin_own = d->own_bits.p + bm.own_off;
            ba.fin_own_bits = (int32_t)std::min<int64_t>(d->own_words * 32, INT32_MAX);
        }
        {
            ProfScope ps(m, FMHIP_K_FIXUP, bm.nnz_total, bm.rows);
            HIP_TRY(launch_fixup(m->Kp, ba, m->stream));
            HIP_TRY(launch_fixup2(m->Kp, ba, m->stream));
        }
        return FMHIP_OK;
    }
    const int32_t *hf = d->h_cfeat.data() + bm.col_off, *hp = d->h_cptr.data() + bm.col_off + b;
    const int32_t *hs = d->h_split.data() + bm.split_off;
    const int32_t s_lo = (int32_t)(std::lower_bound(hf, hf + bm.n_cols, (int32_t)std::min<int64_t>(feat_lo, INT32_MAX)) - hf);
    const int32_t s_hi = (int32_t)(std::lower_bound(hf, hf + bm.n_cols, (int32_t)std::min<int64_t>(feat_hi, INT32_MAX)) - hf);
    const int32_t e_lo = hp[s_lo], e_hi = hp[s_hi];            // entry interval of the columns
    ba.rho_lo = e_lo / kRangeLen;
    ba.rho_hi = s_hi >= bm.n_cols ? bm.n_ranges : e_hi / kRangeLen;   // the straddling range goes to the next interval
    const int32_t sp_lo = (int32_t)(std::lower_bound(hs, hs + bm.n_split, s_lo) - hs);
    const int32_t sp_hi = (int32_t)(std::lower_bound(hs, hs + bm.n_split, s_hi) - hs);
    ba.split_seg += sp_lo;
    ba.n_split = sp_hi - sp_lo;
    {
        const int32_t *hss = d->h_split_short.data() + bm.split_short_off;
        const int32_t q_lo = (int32_t)(std::lower_bound(hss, hss + bm.n_split_short, s_lo) - hss);
        const int32_t q_hi = (int32_t)(std::lower_bound(hss, hss + bm.n_split_short, s_hi) - hss);
        ba.split_short += q_lo;
        ba.n_split_short = q_hi - q_lo;
    }
    if (finish) {
        ba.red_bsum = m->bsum.p;
        ba.red_nblocks = m->fwd_parts;
        ba.red_rows = (int32_t)bm.rows;
        ba.red_scal = m->view ? m->view->scal : m->scal();
        ba.red_acc = acc;
    }
    const int64_t nnz_part = (int64_t)e_hi - e_lo;
    {
        ProfScope ps(m, FMHIP_K_BACKWARD, nnz_part, bm.rows);
        HIP_TRY(launch_backward(m->Kp, ba, m->stream));
    }
    {
        ProfScope ps(m, FMHIP_K_FIXUP, nnz_part, bm.rows);
        HIP_TRY(launch_fixup(m->Kp, ba, m->stream));
    }
    return FMHIP_OK;
}

// forward + backward + fixup of one batch into the packed gradient (fused: straight into the parameters)
int step_compute(fmhip_model_t m, fmhip_dataset_t d, int64_t b, double *acc, const FusedPlan *fused = nullptr) {
    TRY(step_forward(m, d, b));
    return step_backward(m, d, b, 0, INT64_MAX, true, acc, fused);
}

// Can this step apply its gradient rows inside the backward (no exchange, no separate update launch)?  It is the
// rows-only update, so weight decay must be expressible through the tables' scale (lazy decay, fm_apply.hip).
bool plan_fused(fmhip_model_t m, fmhip_dataset_t d, int64_t b, double eta, double reg0, double regw, double regv, FusedPlan *p) {
    const double dv = 1.0 - eta * regv, dw = 1.0 - eta * regw;
    const bool decay = regw != 0.0 || regv != 0.0;
    const bool lazy_ok = !decay || (m->tv(kTuneLazy) && dv >= 0.5 && dw >= 0.5 && dv <= 1.0 && dw <= 1.0);
    const BatchMeta &bm0 = d->batches[(size_t)b];
    p->eta = eta;
    p->reg0 = reg0;
    p->regw = regw;
    p->regv = regv;
    {
        const float rows = (float)bm0.rows;
        p->upd.invb = rows > 0.f ? 1.0f / rows : 0.f;
    }
    // merged finish (key 11): when the step's update is the DENSE pass (the batch touches most of the model, or decay
    // cannot ride in the scale) it runs inside the fixup launch, beside the fixups, instead of as a launch of its own
    const int64_t touched = (int64_t)bm0.n_cols + d->hot_pages * kHotT;
    const bool rows_only = lazy_ok && touched * 2 <= m->n1;
    if (m->tv(kTuneMerged) && !m->tv(kTuneFused) && d->rb_rows == 0 && !rows_only && bm0.own_off >= 0 && d->dimension <= m->n) {
        p->mode = 2;
        p->sv_out = p->sw_out = 1.0;      // the dense pass folds the scale
        return true;
    }
    if (!m->tv(kTuneFused) || d->rb_rows != 0) return false;
    if (!lazy_ok) return false;
    p->mode = 1;
    p->sv_out = m->sv * dv;
    p->sw_out = m->sw * dw;
    p->upd.V = m->V.p;
    p->upd.w = m->w.p;
    p->upd.sv = (float)m->sv;
    p->upd.eta_v = (float)(eta / p->sv_out);
    p->upd.eta_w = (float)(eta / p->sw_out);
    return true;
}

// brings lazily decayed tables back to scale 1 (dense pass)
int fold_scales(fmhip_model_t m) {
    if (m->sv == 1.0 && m->sw == 1.0) return FMHIP_OK;
    HIP_TRY(launch_rescale(m->Kp, m->V.p, m->w.p, m->n1, m->pack_k(), (float)m->sv, (float)m->sw, m->stream));
    m->sv = m->sw = 1.0;
    return FMHIP_OK;
}

// what step_apply leaves behind, for a step whose update already happened inside the backward
int finish_fused(fmhip_model_t m, const FusedPlan &p) {
    m->sv = p.sv_out;
    m->sw = p.sw_out;
    if (m->sv < 0x1p-24 || m->sw < 0x1p-24) TRY(fold_scales(m));
    m->grad_dirty = false;        // nothing but the statistics head was written
    m->host64_fresh = false;
    ++m->prof_step;
    return FMHIP_OK;
}

// `d`/`b` given: the gradient in the buffer is exactly batch b's (no exchange happened), so the update
// may be restricted to the rows that batch touched — their decay, and everyone else's, rides in the
// tables' scale (lazy weight decay, fm_apply.hip).  Otherwise the dense pass, which also folds a pending
// scale back to 1.
int step_apply(fmhip_model_t m, double eta, double reg0, double regw, double regv, fmhip_dataset_t d, int64_t b) {
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    double sv_out = 1.0, sw_out = 1.0;
    const double dv = 1.0 - eta * regv, dw = 1.0 - eta * regw;
    const bool decay = regw != 0.0 || regv != 0.0;
    if (d && b >= 0 && d->rb_rows == 0 && (!decay || (m->tv(kTuneLazy) && dv >= 0.5 && dw >= 0.5 && dv <= 1.0 && dw <= 1.0))) {
        const BatchMeta &bm = d->batches[(size_t)b];
        const int64_t touched = (int64_t)bm.n_cols + d->hot_pages * kHotT;
        if (touched * 2 <= m->n1) {     // otherwise the dense, perfectly coalesced pass is as cheap
            a.feat = d->cfeat.p + bm.col_off;
            a.n_feat = bm.n_cols;
            a.hot_ids = d->d_hot_ids.p;
            a.n_hot = d->hot_pages * kHotT;
            sv_out = m->sv * dv;
            sw_out = m->sw * dw;
        }
    }
    a.eta_v = (float)(eta / sv_out);
    a.eta_w = (float)(eta / sw_out);
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    a.scal = m->scal();
    a.rows = m->scal() + 2;
    a.n1 = m->n1;
    a.row_lo = 0;
    a.row_hi = m->n1;
    a.do_w0 = 1;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    {
        ProfScope ps(m, FMHIP_K_APPLY, m->last_nnz, m->last_rows);
        HIP_TRY(launch_apply(m->Kp, a, m->stream));
    }
    m->sv = sv_out;
    m->sw = sw_out;
    // fp32 tables lose nothing to a small scale until their values approach the denormal range; fold long before
    if (m->sv < 0x1p-24 || m->sw < 0x1p-24) TRY(fold_scales(m));
    m->grad_dirty = false;
    m->host64_fresh = false;
    ++m->prof_step;
    return FMHIP_OK;
}

// The dense update of the feature rows [lo, hi) only — the data-parallel step applies an interval as soon as its
// slice of the gradient has been exchanged (fmhip_comm.hip).  `rows`: device float holding the global row count;
// `last`: the final interval of the step (also steps w0 from the head's scalars and closes the step's bookkeeping).
int step_apply_interval(fmhip_model_t m, double eta, double reg0, double regw, double regv, int64_t lo, int64_t hi,
                        const float *rows, bool last) {
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    a.eta_v = a.eta_w = (float)eta;
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    a.scal = m->scal();
    a.rows = rows;
    a.n1 = m->n1;
    a.row_lo = lo;
    a.row_hi = hi;
    a.do_w0 = last ? 1 : 0;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    if (hi > lo || last) {
        ProfScope ps(m, FMHIP_K_APPLY, m->last_nnz, m->last_rows);
        HIP_TRY(launch_apply(m->Kp, a, m->stream));
    }
    if (last) {
        m->sv = m->sw = 1.0;      // every interval folded the pending scale
        m->grad_dirty = false;
        m->host64_fresh = false;
        ++m->prof_step;
    }
    return FMHIP_OK;
}

int step_apply_shard(fmhip_model_t m, double eta, double reg0, double regw, double regv, int64_t lo, int64_t hi, int64_t hi_r,
                     int64_t vlo, int64_t vhi, const float *rows, bool last, hipStream_t s) {
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    a.eta_v = a.eta_w = (float)eta;
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    a.scal = m->scal();
    a.rows = rows;
    a.n1 = m->n1;
    hi = std::min(hi, m->n1);
    a.row_lo = std::min(std::max(vlo, lo), hi);
    a.row_hi = std::min(std::max(vhi, a.row_lo), hi);
    a.w_lo = lo;
    a.w_hi = hi;
    a.z_hi = std::max(hi_r, hi);
    a.do_w0 = last ? 1 : 0;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    HIP_TRY(launch_apply_shard(m->Kp, a, s));
    if (last) {
        m->sv = m->sw = 1.0;      // every share folded the pending scale; the all-gather spreads the folded rows
        m->host64_fresh = false;
        ++m->prof_step;
    }
    return FMHIP_OK;
}

// can weight decay ride in the tables' scale for this (eta, reg)?  (no decay at all: trivially)
bool lazy_decay_ok(fmhip_model_t m, double eta, double regw, double regv) {
    const double dv = 1.0 - eta * regv, dw = 1.0 - eta * regw;
    if (regw == 0.0 && regv == 0.0) return true;
    return m->tv(kTuneLazy) && dv >= 0.5 && dw >= 0.5 && dv <= 1.0 && dw <= 1.0;
}

int step_apply_rows(fmhip_model_t m, double eta, double reg0, double regw, double regv, const int32_t *feat, int32_t n_feat,
                    const float *rows, const GradView *view) {
    if (!lazy_decay_ok(m, eta, regw, regv))
        return fail(FMHIP_ERR_UNSUPPORTED, "a rows-only update needs weight decay that fits the tables' scale (0.5 <= 1 - eta*reg <= 1)");
    const double sv_out = m->sv * (1.0 - eta * regv), sw_out = m->sw * (1.0 - eta * regw);
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    a.feat = feat;
    a.n_feat = n_feat;
    a.hot_ids = nullptr;
    a.n_hot = 0;
    a.eta_v = (float)(eta / sv_out);
    a.eta_w = (float)(eta / sw_out);
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = view ? view->GV : m->GV();
    a.Gw = view ? view->Gw : m->Gw();
    a.Gb = view ? view->Gb : m->Gb();
    a.scal = view ? view->scal : m->scal();
    a.g_compact = view ? 1 : 0;
    a.rows = rows;
    a.n1 = m->n1;
    a.row_lo = 0;
    a.row_hi = m->n1;
    a.do_w0 = 1;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    {
        ProfScope ps(m, FMHIP_K_APPLY, m->last_nnz, m->last_rows);
        HIP_TRY(launch_apply(m->Kp, a, m->stream));
    }
    m->sv = sv_out;
    m->sw = sw_out;
    if (m->sv < 0x1p-24 || m->sw < 0x1p-24) TRY(fold_scales(m));
    m->grad_dirty = false;
    m->host64_fresh = false;
    ++m->prof_step;
    return FMHIP_OK;
}

int read_scal(fmhip_model_t m, fmhip_stats *st) {
    float h[4];
    HIP_TRY(hipMemcpyAsync(h, m->scal(), sizeof h, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    st->sum_e = h[0];
    st->sse = h[1];
    st->rows = (int64_t)llround(h[2]);
    st->nonfinite = (int64_t)llround(h[3]);
    return FMHIP_OK;
}

int read_acc(fmhip_model_t m, fmhip_stats *st) {
    double h[4];
    HIP_TRY(hipMemcpyAsync(h, m->acc.p, sizeof h, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    st->sum_e = h[0];
    st->sse = h[1];
    st->rows = (int64_t)llround(h[2]);
    st->nonfinite = (int64_t)llround(h[3]);
    return FMHIP_OK;
}

template <typename FT>
int set_params_impl(fmhip_model_t m, FT w0, const FT *w, const FT *v) {
    if (!m || !w || !v) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(m->device));
    std::vector<float> hV((size_t)m->n1p * m->Kp, 0.f), hw((size_t)m->n1p, 0.f);
    for (int64_t i = 0; i < m->n1; ++i) {
        hw[(size_t)i] = (float)w[i];
        for (int f = 0; f < m->k; ++f) hV[(size_t)i * m->Kp + f] = (float)v[f + i * (int64_t)m->k];
        if (m->pack_k() >= 0) hV[(size_t)i * m->Kp + m->k] = (float)w[i];   // packed rows: w_i rides in slot k
    }
    const float hw0 = (float)w0;
    m->h_w0 = (double)w0;
    m->h_w.assign((size_t)m->n1, 0.0);
    m->h_v.assign((size_t)m->n1 * m->k, 0.0);
    for (int64_t i = 0; i < m->n1; ++i) m->h_w[(size_t)i] = (double)w[i];
    for (int64_t j = 0; j < m->n1 * m->k; ++j) m->h_v[(size_t)j] = (double)v[j];
    m->host64_fresh = true;
    m->sv = m->sw = 1.0;
    HIP_TRY(hipMemcpyAsync(m->V.p, hV.data(), hV.size() * sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->w.p, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->w0.p, &hw0, sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return FMHIP_OK;
}

template <typename FT>
int get_params_impl(fmhip_model_t m, FT *w0, FT *w, FT *v) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    if (m->host64_fresh) {   // nothing has trained in fp32 since the masters were written: return them exactly
        if (w0) *w0 = (FT)m->h_w0;
        if (w) for (int64_t i = 0; i < m->n1; ++i) w[i] = (FT)m->h_w[(size_t)i];
        if (v) for (int64_t j = 0; j < m->n1 * m->k; ++j) v[j] = (FT)m->h_v[(size_t)j];
        return FMHIP_OK;
    }
    std::vector<float> hV((size_t)m->n1p * m->Kp), hw((size_t)m->n1p);
    float hw0 = 0.f;
    HIP_TRY(hipMemcpyAsync(hV.data(), m->V.p, hV.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hw.data(), m->w.p, hw.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(&hw0, m->w0.p, sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (w0) *w0 = (FT)hw0;
    // a lazily decayed model stores U with V = sv*U (fm_apply.hip); with sv = sw = 1 the products are exact
    for (int64_t i = 0; i < m->n1; ++i) {
        if (w) w[i] = (FT)((double)(m->pack_k() >= 0 ? hV[(size_t)i * m->Kp + m->k] : hw[(size_t)i]) * m->sw);
        if (v)
            for (int f = 0; f < m->k; ++f) v[f + i * (int64_t)m->k] = (FT)((double)hV[(size_t)i * m->Kp + f] * m->sv);
    }
    return FMHIP_OK;
}

}  // namespace host
}  // namespace fmhip

// =================================================================== C ABI

extern "C" {

int fmhip_version(void) { return FMHIP_VERSION; }

const char *fmhip_last_error(void) { return g_err.c_str(); }

int fmhip_tune(int key, int value) {
    if (key < 0 || key >= kTuneCount) return fail(FMHIP_ERR_INVALID, "unknown tuning key %d", key);
    g_tune[key] = value;
    return FMHIP_OK;
}

int fmhip_model_tune(fmhip_model_t m, int key, int value) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (key < 0 || key >= kTuneCount) return fail(FMHIP_ERR_INVALID, "unknown tuning key %d", key);
    if (key == kTuneRowBlock || key == kTuneHot || key == kTuneHotPages)
        return fail(FMHIP_ERR_INVALID, "tuning key %d decides a DATASET's layout: state it in fmhip_dataset_opts (or the process default, fmhip_tune)", key);
    m->tune[key] = value < 0 ? -1 : value;
    return FMHIP_OK;
}

int fmhip_device_count(int *count) {
    if (!count) return fail(FMHIP_ERR_INVALID, "count is NULL");
    *count = 0;
    HIP_TRY(hipGetDeviceCount(count));
    return FMHIP_OK;
}

int fmhip_model_create(int device, int64_t num_attribute, int32_t num_factor, void *stream, fmhip_model_t *out) {
    if (!out) return fail(FMHIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_attribute < 0 || num_attribute >= ((int64_t)1 << 31) - 8)
        return fail(FMHIP_ERR_INVALID, "num_attribute %lld out of range", (long long)num_attribute);
    if (num_factor < 1) return fail(FMHIP_ERR_INVALID, "num_factor must be >= 1");
    if (num_factor > FMHIP_MAX_FACTORS)
        return fail(FMHIP_ERR_UNSUPPORTED, "num_factor %d > FMHIP_MAX_FACTORS (%d)", num_factor, FMHIP_MAX_FACTORS);
    TRY(set_device(device));
    fmhip_model *m = new (std::nothrow) fmhip_model();
    if (!m) return fail(FMHIP_ERR_NOMEM, "out of host memory");
    m->device = device;
    for (int &t : m->tune) t = -1;          // every key follows the process-wide default until fmhip_model_tune says otherwise
    m->n = num_attribute;
    m->n1 = num_attribute + 1;
    m->n1p = (m->n1 + 3) & ~(int64_t)3;
    m->k = num_factor;
    m->Kp = padded_factors(num_factor);
    if (stream) {
        m->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete m;
            return fail(FMHIP_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        m->own_stream = true;
    }
    int rc;
    const size_t slack = (size_t)fmhip_model::kSlackRows * m->Kp;       // zero rows behind the tables (sharded exchange)
    if ((rc = m->V.alloc((size_t)m->n1p * m->Kp + slack)) || (rc = m->w.alloc((size_t)m->n1p)) || (rc = m->w0.alloc(1)) ||
        (rc = m->grad_own.alloc(m->grad_floats() + slack)) || (rc = m->acc.alloc(4))) {
        fmhip_model_destroy(m);
        return rc;
    }
    m->grad = m->grad_own.p;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMemsetAsync(m->V.p, 0, m->V.n * sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->w.p, 0, m->w.n * sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->w0.p, 0, sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->grad, 0, m->grad_own.n * sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->acc.p, 0, 4 * sizeof(double), m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
        fmhip_model_destroy(m);
        return fail(FMHIP_ERR_HIP, "zero-initialisation failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return FMHIP_OK;
}

int fmhip_model_destroy(fmhip_model_t m) {
    if (!m) return FMHIP_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (auto &r : m->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return FMHIP_OK;
}

int fmhip_model_info(fmhip_model_t m, int64_t *num_attribute, int32_t *num_factor, int32_t *padded) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (num_attribute) *num_attribute = m->n;
    if (num_factor) *num_factor = m->k;
    if (padded) *padded = m->Kp;
    return FMHIP_OK;
}

int fmhip_model_init_normal(fmhip_model_t m, uint64_t seed, double mean, double stdev) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    HIP_TRY(launch_init_normal(m->Kp, m->V.p, m->w.p, m->w0.p, m->n1, m->n1p, m->k, seed, (float)mean, (float)stdev, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->sv = m->sw = 1.0;
    m->host64_fresh = false;     // the device holds the parameters; the fp64 masters are refreshed on demand
    std::vector<double>().swap(m->h_w);
    std::vector<double>().swap(m->h_v);
    return FMHIP_OK;
}

int fmhip_model_get_rows(fmhip_model_t m, int64_t n, const int32_t *ids, double *w, double *v) {
    if (!m || n < 0 || (n > 0 && !ids)) return fail(FMHIP_ERR_INVALID, "NULL argument or negative count");
    for (int64_t j = 0; j < n; ++j)
        if (ids[j] < 0 || ids[j] > m->n) return fail(FMHIP_ERR_SHAPE, "feature id %d outside [0, %lld]", ids[j], (long long)m->n);
    if (n == 0) return FMHIP_OK;
    TRY(set_device(m->device));
    DevBuf<int32_t> dids;
    DevBuf<float> dv, dw;
    TRY(dids.alloc((size_t)n));
    TRY(dv.alloc((size_t)n * m->Kp));
    TRY(dw.alloc((size_t)n));
    HIP_TRY(hipMemcpyAsync(dids.p, ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(launch_gather_rows(m->Kp, m->V.p, m->w.p, dids.p, n, dv.p, dw.p, m->stream));
    std::vector<float> hv((size_t)n * m->Kp), hw((size_t)n);
    HIP_TRY(hipMemcpyAsync(hv.data(), dv.p, hv.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hw.data(), dw.p, hw.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (int64_t j = 0; j < n; ++j) {       // scales of a lazily decayed model, packed rows: as get_params
        if (w) w[j] = (double)(m->pack_k() >= 0 ? hv[(size_t)j * m->Kp + m->k] : hw[(size_t)j]) * m->sw;
        if (v)
            for (int f = 0; f < m->k; ++f) v[f + j * (int64_t)m->k] = (double)hv[(size_t)j * m->Kp + f] * m->sv;
    }
    return FMHIP_OK;
}

int fmhip_model_set_params(fmhip_model_t m, double w0, const double *w, const double *v) { return set_params_impl<double>(m, w0, w, v); }
int fmhip_model_get_params(fmhip_model_t m, double *w0, double *w, double *v) { return get_params_impl<double>(m, w0, w, v); }
int fmhip_model_set_params_f32(fmhip_model_t m, float w0, const float *w, const float *v) { return set_params_impl<float>(m, w0, w, v); }
int fmhip_model_get_params_f32(fmhip_model_t m, float *w0, float *w, float *v) { return get_params_impl<float>(m, w0, w, v); }

int fmhip_synchronize(fmhip_model_t m) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return FMHIP_OK;
}

int fmhip_dataset_create(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                         const double *y, int64_t batch_rows, fmhip_dataset_t *out) {
    return dataset_create_impl<double>(device, n_rows, row_ptr, col, val, y, batch_rows, false, out);
}

int fmhip_dataset_create_opts(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                              const double *y, const fmhip_dataset_opts *opts, fmhip_dataset_t *out) {
    if (!opts || opts->struct_size != (int32_t)sizeof(fmhip_dataset_opts))
        return fail(FMHIP_ERR_INVALID, "opts is NULL or its struct_size is not sizeof(fmhip_dataset_opts)");
    return dataset_create_impl<double>(device, n_rows, row_ptr, col, val, y, opts->batch_rows, false, out, opts->hot_block,
                                       opts->row_block_rows);
}

int fmhip_rows_create(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                      const double *y, fmhip_dataset_t *out) {
    return dataset_create_impl<double>(device, n_rows, row_ptr, col, val, y, 0, true, out);
}

int fmhip_rows_create_f32(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const float *val,
                          const float *y, fmhip_dataset_t *out) {
    return dataset_create_impl<float>(device, n_rows, row_ptr, col, val, y, 0, true, out);
}

int fmhip_dataset_create_f32(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const float *val,
                             const float *y, int64_t batch_rows, fmhip_dataset_t *out) {
    return dataset_create_impl<float>(device, n_rows, row_ptr, col, val, y, batch_rows, false, out);
}

int fmhip_dataset_destroy(fmhip_dataset_t d) {
    if (!d) return FMHIP_OK;
    (void)hipSetDevice(d->device);
    delete d;
    return FMHIP_OK;
}

int fmhip_dataset_info(fmhip_dataset_t d, int64_t *n_rows, int64_t *nnz, int64_t *dimension, int64_t *batch_rows,
                       int64_t *n_batches) {
    if (!d) return fail(FMHIP_ERR_INVALID, "dataset is NULL");
    if (n_rows) *n_rows = d->n_rows;
    if (nnz) *nnz = d->nnz;
    if (dimension) *dimension = d->dimension;
    if (batch_rows) *batch_rows = d->batch_rows;
    if (n_batches) *n_batches = (int64_t)d->batches.size();
    return FMHIP_OK;
}

int fmhip_dataset_batch_info(fmhip_dataset_t d, int64_t batch, int64_t *row0, int64_t *rows, int64_t *nnz,
                             int64_t *n_columns) {
    if (!d) return fail(FMHIP_ERR_INVALID, "dataset is NULL");
    TRY(check_batch(d, batch));
    const BatchMeta &bm = d->batches[(size_t)batch];
    if (row0) *row0 = bm.row0;
    if (rows) *rows = bm.rows;
    if (nnz) *nnz = bm.nnz_total;
    if (n_columns) *n_columns = bm.n_feats + __builtin_popcountll((uint64_t)bm.hot_mask) + __builtin_popcountll((uint64_t)(bm.hot_mask >> 64));
    return FMHIP_OK;
}

int fmhip_dataset_get_transpose(fmhip_dataset_t d, int64_t batch, int32_t *feat, int32_t *ptr, int32_t *rows,
                                float *vals) {
    if (!d) return fail(FMHIP_ERR_INVALID, "dataset is NULL");
    TRY(check_batch(d, batch));
    if (d->scoring_only) return fail(FMHIP_ERR_UNSUPPORTED, "a scoring-only dataset has no transposes");
    TRY(set_device(d->device));
    const BatchMeta &bm = d->batches[(size_t)batch];
    // read the stream back and merge the pieces of a feature (one per row block, in row-block = row
    // order) so the caller sees one column per feature whatever the device layout
    std::vector<int32_t> hrow((size_t)bm.cnnz);
    std::vector<float> hval((size_t)bm.cnnz);
    if (bm.cnnz) {
        HIP_TRY(hipMemcpy(hrow.data(), d->crow.p + bm.nnz0, (size_t)bm.cnnz * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hval.data(), d->cval.p + bm.nnz0, (size_t)bm.cnnz * sizeof(float), hipMemcpyDeviceToHost));
    }
    const int32_t *hf = d->h_cfeat.data() + bm.col_off, *hp = d->h_cptr.data() + bm.col_off + batch;
    std::vector<int32_t> order((size_t)bm.n_cols);
    for (int32_t s = 0; s < bm.n_cols; ++s) order[(size_t)s] = s;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return hf[x] < hf[y]; });
    // the dense hot block's columns (never present in the transposed stream) are merged in by feature id
    const int n_slots = d->hot_pages * kHotT;
    const size_t page_floats = (size_t)std::max<int64_t>(d->n_rows, 1) * kHotT;
    std::vector<float> hx;
    std::vector<int> hslots;               // the batch's live slots, by ascending feature id
    if (bm.hot_mask) {
        hx.resize((size_t)bm.rows * n_slots);
        for (int pg = 0; pg < d->hot_pages; ++pg)
            HIP_TRY(hipMemcpy(hx.data() + (size_t)pg * bm.rows * kHotT, d->xhot.p + (size_t)pg * page_floats + (size_t)bm.row0 * kHotT,
                              (size_t)bm.rows * kHotT * sizeof(float), hipMemcpyDeviceToHost));
        for (int h = 0; h < n_slots; ++h)
            if (d->hot_ids[(size_t)h] >= 0 && (bm.hot_mask >> h & 1u)) hslots.push_back(h);
        std::sort(hslots.begin(), hslots.end(), [&](int x, int y) { return d->hot_ids[(size_t)x] < d->hot_ids[(size_t)y]; });
    }
    int32_t nf = 0, pos = 0;
    size_t hnext = 0;
    auto emit_hot_below = [&](int64_t bound) {
        for (; hnext < hslots.size(); ++hnext) {
            const int h = hslots[hnext];
            const int32_t id = d->hot_ids[(size_t)h];
            if ((int64_t)id >= bound) break;
            if (feat) feat[nf] = id;
            if (ptr) ptr[nf] = pos;
            ++nf;
            const float *xp = hx.data() + (size_t)(h / kHotT) * bm.rows * kHotT + (h % kHotT);
            for (int64_t r = 0; r < bm.rows; ++r) {
                const float x = xp[(size_t)r * kHotT];
                if (x != 0.f) {
                    if (rows) rows[pos] = (int32_t)r;
                    if (vals) vals[pos] = x;
                    ++pos;
                }
            }
        }
    };
    for (int32_t i = 0; i < bm.n_cols; ++i) {
        const int32_t s = order[(size_t)i];
        if (i == 0 || hf[s] != hf[order[(size_t)i - 1]]) {
            emit_hot_below(hf[s]);
            if (feat) feat[nf] = hf[s];
            if (ptr) ptr[nf] = pos;
            ++nf;
        }
        for (int32_t p = hp[s]; p < hp[s + 1]; ++p, ++pos) {
            if (rows) rows[pos] = hrow[(size_t)p] & 0x7fffffff;
            if (vals) vals[pos] = hval[(size_t)p];
        }
    }
    emit_hot_below(INT64_MAX);
    if (ptr) ptr[nf] = pos;
    return FMHIP_OK;
}

// ---- scoring

static int score_pass(fmhip_model_t m, fmhip_dataset_t d, double *yhat, double *e_out, double *q_out, fmhip_stats *st) {
    TRY(check_pair(m, d));
    TRY(ensure_workspace(m, d));
    HIP_TRY(hipMemsetAsync(m->acc.p, 0, 4 * sizeof(double), m->stream));
    DevBuf<float> dy;
    if (yhat) TRY(dy.alloc((size_t)std::max<int64_t>(d->max_rows, 1)));
    std::vector<float> hbuf;
    for (size_t b = 0; b < d->batches.size(); ++b) {
        const BatchMeta &bm = d->batches[b];
        FwdArgs a = fwd_args(m, d, bm);
        a.yhat = dy.p;
        int parts = 0;
        HIP_TRY(launch_forward(m->Kp, q_out ? kFwdQ : kFwdResidual, a, m->stream, &parts));
        HIP_TRY(launch_reduce_blocks(m->bsum.p, parts, (int32_t)bm.rows, nullptr, m->acc.p, m->stream));
        if (yhat || e_out) {
            hbuf.resize((size_t)bm.rows);
            if (yhat) {
                HIP_TRY(hipMemcpyAsync(hbuf.data(), dy.p, (size_t)bm.rows * sizeof(float), hipMemcpyDeviceToHost, m->stream));
                HIP_TRY(hipStreamSynchronize(m->stream));
                for (int64_t r = 0; r < bm.rows; ++r) yhat[bm.row0 + r] = hbuf[(size_t)r];
            }
            if (e_out) {
                HIP_TRY(hipMemcpyAsync(hbuf.data(), m->e.p, (size_t)bm.rows * sizeof(float), hipMemcpyDeviceToHost, m->stream));
                HIP_TRY(hipStreamSynchronize(m->stream));
                for (int64_t r = 0; r < bm.rows; ++r) e_out[bm.row0 + r] = hbuf[(size_t)r];
            }
        }
        if (q_out) {
            hbuf.resize((size_t)bm.rows * m->Kp);
            HIP_TRY(hipMemcpyAsync(hbuf.data(), m->P.p, hbuf.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            for (int64_t r = 0; r < bm.rows; ++r)
                for (int f = 0; f < m->k; ++f) q_out[(bm.row0 + r) * m->k + f] = hbuf[(size_t)r * m->Kp + f];
        }
    }
    if (st) {
        memset(st, 0, sizeof *st);
        TRY(read_acc(m, st));
        st->nnz = d->nnz;
    }
    return FMHIP_OK;
}

int fmhip_predict(fmhip_model_t m, fmhip_dataset_t d, double *yhat) {
    if (!yhat) return fail(FMHIP_ERR_INVALID, "yhat is NULL");
    return score_pass(m, d, yhat, nullptr, nullptr, nullptr);
}

int fmhip_predict_rows(fmhip_model_t m, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                       double *yhat) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (n_rows > 0 && !yhat) return fail(FMHIP_ERR_INVALID, "yhat is NULL");
    fmhip_dataset_t d = nullptr;
    TRY(dataset_create_impl<double>(m->device, n_rows, row_ptr, col, val, nullptr, 0, true, &d));
    const int rc = n_rows > 0 ? score_pass(m, d, yhat, nullptr, nullptr, nullptr) : FMHIP_OK;
    fmhip_dataset_destroy(d);
    return rc;
}

int fmhip_residual(fmhip_model_t m, fmhip_dataset_t d, double *e) {
    if (!e) return fail(FMHIP_ERR_INVALID, "e is NULL");
    return score_pass(m, d, nullptr, e, nullptr, nullptr);
}

int fmhip_term_q(fmhip_model_t m, fmhip_dataset_t d, double *q) {
    if (!q) return fail(FMHIP_ERR_INVALID, "q is NULL");
    return score_pass(m, d, nullptr, nullptr, q, nullptr);
}

int fmhip_rmse(fmhip_model_t m, fmhip_dataset_t d, double *rmse, fmhip_stats *stats) {
    if (!rmse) return fail(FMHIP_ERR_INVALID, "rmse is NULL");
    fmhip_stats st;
    TRY(score_pass(m, d, nullptr, nullptr, nullptr, &st));
    // S/Model.scala:13-19: sqrt(sum (y - yhat)^2 / size); (y - yhat)^2 == e^2
    *rmse = st.rows > 0 ? std::sqrt(st.sse / (double)st.rows) : 0.0;
    if (stats) *stats = st;
    return FMHIP_OK;
}

// ---- training

int fmhip_sgd_step(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, double eta, double reg0, double regw,
                   double regv, fmhip_stats *stats) {
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    FusedPlan fp{};
    const bool fused = plan_fused(m, d, batch, eta, reg0, regw, regv, &fp);
    TRY(step_compute(m, d, batch, nullptr, fused ? &fp : nullptr));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        TRY(read_scal(m, stats));
        stats->nnz = d->batches[(size_t)batch].nnz_total;
        stats->steps = 1;
    }
    return fused ? finish_fused(m, fp) : step_apply(m, eta, reg0, regw, regv, d, batch);
}

int fmhip_sgd_epoch(fmhip_model_t m, fmhip_dataset_t d, double eta, double reg0, double regw, double regv,
                    const int64_t *order, fmhip_stats *stats) {
    TRY(check_train(m, d));
    const int64_t nb = (int64_t)d->batches.size();
    if (order)
        for (int64_t j = 0; j < nb; ++j) TRY(check_batch(d, order[j]));
    HIP_TRY(hipMemsetAsync(m->acc.p, 0, 4 * sizeof(double), m->stream));
    for (int64_t j = 0; j < nb; ++j) {
        const int64_t b = order ? order[j] : j;
        FusedPlan fp{};
        const bool fused = plan_fused(m, d, b, eta, reg0, regw, regv, &fp);
        TRY(step_compute(m, d, b, m->acc.p, fused ? &fp : nullptr));
        TRY(fused ? finish_fused(m, fp) : step_apply(m, eta, reg0, regw, regv, d, b));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        TRY(read_acc(m, stats));
        stats->nnz = d->nnz;
        stats->steps = nb;
    }
    return FMHIP_OK;
}

int fmhip_batch_grad(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, double *gv, double *gw, double *gw0,
                     fmhip_stats *stats) {
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    TRY(step_compute(m, d, batch, nullptr));
    std::vector<float> hG(m->grad_floats()), hV((size_t)m->n1p * m->Kp);
    HIP_TRY(hipMemcpyAsync(hG.data(), m->grad, hG.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hV.data(), m->V.p, hV.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->grad_dirty = false;
    const float *sc = hG.data(), *Gw = sc + kGradHead, *Gb = Gw + m->n1p, *GV = sc + m->head_floats();
    for (int64_t i = 0; i < m->n1; ++i) {
        if (gw) gw[i] = m->pack_k() >= 0 ? GV[(size_t)i * m->Kp + m->k] : Gw[i];
        if (gv)
            for (int f = 0; f < m->k; ++f)
                gv[f + i * (int64_t)m->k] = (double)GV[(size_t)i * m->Kp + f] - (double)hV[(size_t)i * m->Kp + f] * m->sv * (double)Gb[i];
    }
    if (gw0) *gw0 = sc[0];
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->sum_e = sc[0];
        stats->sse = sc[1];
        stats->rows = (int64_t)llround(sc[2]);
        stats->nonfinite = (int64_t)llround(sc[3]);
        stats->nnz = d->batches[(size_t)batch].nnz_total;
    }
    return FMHIP_OK;
}

// ---- ALS (the reference's own learner), fp64

int fmhip_als_epoch(fmhip_model_t m, fmhip_dataset_t d, double reg0, double regw, double regv) {
    TRY(check_train(m, d));
    if (d->batches.size() > 1 || (d->nnz > 0 && !d->val64.p))
        return fail(FMHIP_ERR_UNSUPPORTED, "ALS walks the whole-dataset transpose: create the dataset with batch_rows <= 0 "
                                           "(single batch, at most 2^27 stored nonzeros)");
    if (d->rb_rows > 0) return fail(FMHIP_ERR_UNSUPPORTED, "ALS needs a dataset without row blocks (fmhip_tune key 3 = 0)");
    if (d->als_dup)
        return fail(FMHIP_ERR_UNSUPPORTED, "ALS: a row stores the same feature index twice; the column walk updates every row of a "
                                           "column at once and needs the (row, feature) pairs to be distinct");
    if (!m->host64_fresh) {   // parameters last changed by fp32 SGD: start from their fp64 widening
        std::vector<double> w((size_t)m->n1), v((size_t)m->n1 * m->k);
        double w0 = 0.0;
        TRY(get_params_impl<double>(m, &w0, w.data(), v.data()));
        m->h_w0 = w0;
        m->h_w.swap(w);
        m->h_v.swap(v);
        m->host64_fresh = true;
    }
    const size_t n1 = (size_t)m->n1, nv = n1 * (size_t)m->k, nr = (size_t)std::max<int64_t>(d->n_rows, 1);
    TRY(m->als_w0.ensure(1));
    TRY(m->als_w.ensure(n1));
    TRY(m->als_v.ensure(nv));
    TRY(m->als_e.ensure(nr));
    TRY(m->als_q.ensure(nr * (size_t)m->k));
    TRY(m->als_part.ensure(2 * (size_t)kAlsMaxParts + 2));
    HIP_TRY(hipMemcpyAsync(m->als_w0.p, &m->h_w0, sizeof(double), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->als_w.p, m->h_w.data(), n1 * sizeof(double), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->als_v.p, m->h_v.data(), nv * sizeof(double), hipMemcpyHostToDevice, m->stream));
    if (d->n_rows > 0) {
        const BatchMeta &bm = d->batches[0];
        AlsArgs a{};
        a.k = m->k;
        a.num_attribute = m->n;
        a.n_rows = d->n_rows;
        a.nnz = d->nnz;
        a.row_ptr = d->row_ptr.p;
        a.col = d->col.p;
        a.val = d->val64.p;
        a.scol = d->scol.p;
        a.sval = d->sval64.p;
        a.y = d->y64.p;
        a.n_cols = bm.n_cols;
        a.cfeat = d->cfeat.p;
        a.cptr = d->cptr.p;
        a.crow = d->crow.p;
        a.cval = d->cval64.p;
        a.w0 = m->als_w0.p;
        a.w = m->als_w.p;
        a.v = m->als_v.p;
        a.reg0 = reg0;
        a.regw = regw;
        a.regv = regv;
        a.e = m->als_e.p;
        a.q = m->als_q.p;
        a.part = m->als_part.p;
        HIP_TRY(launch_als_epoch(a, d->h_cfeat.data(), d->h_cptr.data(), m->stream));
    }
    std::vector<double> w(n1), v(nv);
    double w0 = 0.0;
    HIP_TRY(hipMemcpyAsync(&w0, m->als_w0.p, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(w.data(), m->als_w.p, n1 * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(v.data(), m->als_v.p, nv * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return set_params_impl<double>(m, w0, w.data(), v.data());   // refreshes the fp64 masters and the fp32 device copy
}

// ---- data-parallel split step

int fmhip_grad_floats(fmhip_model_t m, int64_t *n_floats) {
    if (!m || !n_floats) return fail(FMHIP_ERR_INVALID, "NULL argument");
    *n_floats = (int64_t)m->grad_floats();
    return FMHIP_OK;
}

int fmhip_grad_bind(fmhip_model_t m, void *device_ptr) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15u))
        return fail(FMHIP_ERR_INVALID, "gradient buffer must be 16-byte aligned");
    m->grad = device_ptr ? static_cast<float *>(device_ptr) : m->grad_own.p;
    m->grad_dirty = false;
    return FMHIP_OK;
}

int fmhip_grad_ptr(fmhip_model_t m, void **device_ptr) {
    if (!m || !device_ptr) return fail(FMHIP_ERR_INVALID, "NULL argument");
    *device_ptr = m->grad;
    return FMHIP_OK;
}

int fmhip_step_compute(fmhip_model_t m, fmhip_dataset_t d, int64_t batch) {
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    return step_compute(m, d, batch, nullptr);
}

int fmhip_step_forward(fmhip_model_t m, fmhip_dataset_t d, int64_t batch) {
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    return step_forward(m, d, batch);
}

int fmhip_step_backward(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, int64_t feat_lo, int64_t feat_hi, int finish) {
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    if (feat_lo < 0 || feat_hi < feat_lo) return fail(FMHIP_ERR_INVALID, "bad feature interval [%lld, %lld)", (long long)feat_lo, (long long)feat_hi);
    // intervals must come in DESCENDING order and tile [0, n+1): a range straddling two intervals is
    // walked with the upper one, whose partials the lower one's fixup then reads
    if (m->bw_next_hi < 0) return fail(FMHIP_ERR_INVALID, "fmhip_step_backward without fmhip_step_forward");
    if (m->bw_next_hi == INT64_MAX ? feat_hi < m->n1 : feat_hi != m->bw_next_hi)
        return fail(FMHIP_ERR_INVALID, "feature intervals must tile [0, n+1) in descending order (expected hi = %lld, got %lld)",
                    (long long)(m->bw_next_hi == INT64_MAX ? m->n1 : m->bw_next_hi), (long long)feat_hi);
    if (finish && feat_lo != 0) return fail(FMHIP_ERR_INVALID, "finish = 1 belongs to the interval that starts at feature 0");
    TRY(step_backward(m, d, batch, feat_lo, feat_hi, finish != 0, nullptr, nullptr));
    m->bw_next_hi = feat_lo == 0 ? -1 : feat_lo;
    return FMHIP_OK;
}

int fmhip_grad_layout(fmhip_model_t m, int64_t *row_floats, int64_t *gv_offset) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (row_floats) *row_floats = m->Kp;
    if (gv_offset) *gv_offset = (int64_t)m->head_floats();
    return FMHIP_OK;
}

int fmhip_step_apply(fmhip_model_t m, double eta, double reg0, double regw, double regv) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    return step_apply(m, eta, reg0, regw, regv);
}

int fmhip_step_stats(fmhip_model_t m, fmhip_stats *stats) {
    if (!m || !stats) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(m->device));
    memset(stats, 0, sizeof *stats);
    TRY(read_scal(m, stats));
    stats->nnz = m->last_nnz;
    stats->steps = 1;
    return FMHIP_OK;
}

// ---- measurement

int fmhip_dataset_layout(fmhip_dataset_t d, int32_t *n_hot, int32_t *hot_ids, int64_t *nnz_sparse) {
    if (!d) return fail(FMHIP_ERR_INVALID, "dataset is NULL");
    int32_t n = 0;
    for (int h = 0; h < (int)std::min<size_t>(d->hot_ids.size(), kHotT); ++h)      // page 0: the two-sided page
        if (d->hot_ids[(size_t)h] >= 0) {
            if (hot_ids) hot_ids[n] = d->hot_ids[(size_t)h];
            ++n;
        }
    if (n_hot) *n_hot = n;
    if (nnz_sparse) *nnz_sparse = d->nnz_sparse;
    return FMHIP_OK;
}

int fmhip_dataset_hot_pages(fmhip_dataset_t d, int32_t *n_pages, int32_t *n_ids, int32_t *ids, int64_t *nnz_sparse_backward) {
    if (!d) return fail(FMHIP_ERR_INVALID, "dataset is NULL");
    int32_t n = 0;
    for (size_t h = 0; h < d->hot_ids.size(); ++h)
        if (d->hot_ids[h] >= 0) {
            if (ids) ids[n] = d->hot_ids[h];
            ++n;
        }
    if (n_pages) *n_pages = d->hot_pages;
    if (n_ids) *n_ids = n;
    if (nnz_sparse_backward) *nnz_sparse_backward = d->scoring_only ? 0 : d->nnz_sparse_bwd;
    return FMHIP_OK;
}

int fmhip_profile_begin(fmhip_model_t m) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    for (auto &r : m->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->prof.clear();
    m->profiling = true;
    m->prof_rotate = false;
    m->prof_period = 1;
    m->prof_step = 0;
    return FMHIP_OK;
}

int fmhip_profile_begin_rotating(fmhip_model_t m) { return fmhip_profile_begin_sampled(m, 1); }

int fmhip_profile_begin_sampled(fmhip_model_t m, int period) {
    if (period < 1) return fail(FMHIP_ERR_INVALID, "period must be >= 1");
    int rc = fmhip_profile_begin(m);
    if (rc == FMHIP_OK) {
        m->prof_rotate = true;
        m->prof_period = period;
    }
    return rc;
}

int fmhip_profile_end(fmhip_model_t m, fmhip_profile *p) {
    if (!m || !p) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(m->device));
    m->profiling = false;
    memset(p, 0, sizeof *p);
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (auto &r : m->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            p->ms[r.kind] += ms;
            p->launches[r.kind] += 1;
            p->nnz[r.kind] += r.nnz;
            p->rows[r.kind] += r.rows;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->prof.clear();
    return FMHIP_OK;
}

// ---- feature relabelling by frequency (host arithmetic; see include/fmhip.h) -----------------------------------
int fmhip_feature_counts(int64_t nnz, const int32_t *col, int64_t n1, int64_t *counts) {
    if (nnz < 0 || n1 < 1 || n1 > INT32_MAX || !counts || (nnz > 0 && !col)) return fail(FMHIP_ERR_INVALID, "bad arguments");
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
    if (bad.load() >= 0) return fail(FMHIP_ERR_INVALID, "col[%lld] = %d outside [0, %lld)", (long long)bad.load(), col[bad.load()], (long long)n1);
    if (private_tables)
        parallel_chunks(n1, T, [&](int, int64_t lo, int64_t hi) {
            for (int t = 0; t < T; ++t) {
                const int64_t *src = part[(size_t)t].data();
                for (int64_t f = lo; f < hi; ++f) counts[f] += src[f];
            }
        });
    return FMHIP_OK;
}

int fmhip_rank_from_counts(int64_t n1, const int64_t *counts, int32_t *rank, int32_t *by_rank) {
    if (n1 < 1 || n1 > INT32_MAX || !counts || !rank) return fail(FMHIP_ERR_INVALID, "bad arguments");
    std::vector<int32_t> order((size_t)n1);
    std::iota(order.begin(), order.end(), 0);
    // descending count, ties by ascending id: every rank of a job derives the same order from the same counts
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return counts[a] > counts[b]; });
    for (int64_t r = 0; r < n1; ++r) {
        rank[order[(size_t)r]] = (int32_t)r;
        if (by_rank) by_rank[r] = order[(size_t)r];
    }
    return FMHIP_OK;
}

int fmhip_relabel_columns(int64_t nnz, const int32_t *col, int64_t n1, const int32_t *rank, int32_t *out) {
    if (nnz < 0 || n1 < 1 || !rank || (nnz > 0 && (!col || !out))) return fail(FMHIP_ERR_INVALID, "bad arguments");
    std::atomic<int64_t> bad{-1};
    parallel_chunks(nnz, host_threads(nnz), [&](int, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = col[i];
            if (c < 0 || c >= n1) { bad.store(i); return; }
            out[i] = rank[c];
        }
    });
    if (bad.load() >= 0) return fail(FMHIP_ERR_INVALID, "col[%lld] outside [0, %lld): nothing can be relied on in `out`", (long long)bad.load(), (long long)n1);
    return FMHIP_OK;
}

}  // extern "C"
