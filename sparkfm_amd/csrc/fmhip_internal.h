// fmhip_internal.h — host-side state behind the opaque handles of include/fmhip.h / fmhip_experimental.h, shared by the
// translation units of libfmhip.so (fmhip_api.hip: the C ABI of models, scoring and training; fmhip_dataset.hip:
// datasets; fmhip_step.hip: the single-GPU step in pieces; fmhip_comm.hip: the data-parallel step).
// Not installed, not part of the ABI.
#pragma once
#include "../../include/fmhip_experimental.h"   // (includes fmhip.h: the library implements both surfaces)
#include "fm_kernels.h"
#include "fmhip_host.h"   // the pure host arithmetic (shards, relabelling, batch metadata, band plan, ALS levels, the dp plan's cuts and shares)

#include <cstdarg>
#include <cstdint>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

namespace fmhip {
namespace host {

// records the calling thread's error message (fmhip_last_error) and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(_e == hipErrorOutOfMemory ? FMHIP_ERR_NOMEM : FMHIP_ERR_HIP,         \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define TRY(expr)               \
    do {                        \
        int _r = (expr);        \
        if (_r != FMHIP_OK) return _r; \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        release();
        if (count == 0) return FMHIP_OK;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FMHIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return FMHIP_OK;
    }
    int ensure(size_t count) { return count <= n ? FMHIP_OK : alloc(count); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

constexpr int64_t kAlsMaxNnz = (int64_t)1 << 27;

struct BatchMeta {
    int64_t row0 = 0, rows = 0;
    int64_t nnz0 = 0;   // offset of the batch in the global CSR/CSC entry arrays
    int32_t nnz = 0;        // entries of the batch in the CSR stream (what the forward walks)
    int32_t cnnz = 0;       // ... in the CSC stream (<= nnz: the gradient-side hot pages' entries are not in it)
    int32_t n_cols = 0;     // compressed columns (features present in the batch)
    int64_t col_off = 0;    // offset into cfeat; cptr offset is col_off + batch index
    int32_t n_ranges = 0;
    int64_t range_off = 0;
    int32_t n_split = 0;          // cut columns spanning > 8 ranges
    int64_t split_off = 0;
    int32_t n_split_short = 0;    // cut columns spanning <= 8 ranges
    int64_t split_short_off = 0;
    int32_t n_feats = 0;    // distinct features present (== n_cols unless the stream is row-blocked)
    int32_t n_mp = 0;       // features cut into several pieces (one per row block they occur in)
    int64_t mp_off = 0;     // offset into mp_feat; mp_ptr offset is mp_off + batch index
    int32_t n_pieces = 0;   // piece rows those features need
    int64_t nnz_total = 0;  // stored nonzeros of the batch incl. those held in the dense hot block
    unsigned __int128 hot_mask = 0;  // hot slots (all pages: up to 128) with at least one nonzero in this batch
    int64_t own_off = -1;   // offset (in words) of the batch's bitmap of fixup-owned features, -1 = none
    // band-affine placement of the backward's ranges (fmhip_dataset.hip: plan_bands): per XCD a list of range ids, the
    // ranges of its own row bands first; xoff[x] = offset of list x in fmhip_dataset::xlist, -1 = no plan for this batch
    int64_t xoff[fmhip::kXcds] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int32_t xlen[fmhip::kXcds] = {};
    int32_t xseg[fmhip::kXcds][fmhip::kXSegs + 1] = {};   // list x = runs [xseg[x][s], xseg[x][s+1]) of ascending range ids (one per band, then the rest)
    int32_t x_affine = 0;   // ranges that were placed by their band (the rest fill the lists evenly)
};

// Where a backward delivers its gradient rows when NOT into the model's packed buffer: the touched-rows exchange
// (fmhip_comm.hip) points the column walk at a compact buffer that holds one row per feature of the step's union
struct GradView {
    float *scal, *Gw, *Gb, *GV;
    const int32_t *cdst;       // per compressed column of the batch: its row in GV / Gw / Gb
    const int32_t *hot_pos;    // per slot of the dense hot block: its row (-1 = unused slot)
};

struct ProfRec {
    int kind;
    hipEvent_t a, b;
    int64_t nnz, rows;
    int64_t step;       // the step (fmhip_model::prof_step) the launch belongs to
};

// What ONE scoring call (predict / rmse / residual / term_q) works in: its own stream and workspace, so that any number of
// host threads may score through one frozen model at once (the reference's `predict` runs on executor task threads over a
// read-only model, S/Model.scala:14 under `local[*]`, S/driver.scala:14).  Pooled per model; a call takes a free one or
// makes one (fmhip_api.hip: ScoreLease).
struct ScoreCtx {
    hipStream_t s = nullptr;
    hipEvent_t ev = nullptr;        // orders the call behind what the model's own stream has queued (an asynchronous step)
    DevBuf<float> P, e, yhat;
    DevBuf<double> bsum, acc;
    ~ScoreCtx() {
        if (ev) (void)hipEventDestroy(ev);
        if (s) (void)hipStreamDestroy(s);
    }
};

}  // namespace host
}  // namespace fmhip

struct fmhip_dataset {
    using BatchMeta = fmhip::host::BatchMeta;
    template <typename T> using DevBuf = fmhip::host::DevBuf<T>;
    int device = 0;
    int64_t n_rows = 0, nnz = 0, dimension = 0, batch_rows = 0;
    int64_t max_rows = 0;
    int32_t max_ranges = 0;
    std::vector<BatchMeta> batches;
    DevBuf<int64_t> row_ptr;
    DevBuf<int32_t> col;
    DevBuf<float> val, y;
    DevBuf<uint32_t> crow;
    DevBuf<float> cval;
    DevBuf<int32_t> row_order;   // per batch: its rows' local ids sorted by stored length, longest first (forward walk order)
    // two-pass forward (fmhip_dataset_partition_rows): every row's stored entries stably partitioned at feature id
    // split_cut — [row_ptr[r], row_split[r]) hold the features below it, [row_split[r], row_ptr[r+1]) the others
    DevBuf<int64_t> row_split;
    int64_t split_cut = -1;      // -1: not partitioned
    // ... in a COPY of the CSR stream that only the two-pass forward reads (col / val themselves never move once the dataset
    // is built: other threads may be scoring or training with them).  part_mu guards the partition's making; part_users counts
    // the pipelined runs that are walking it (it is not re-made for another cut while one does)
    DevBuf<int32_t> col_part;
    DevBuf<float> val_part;
    std::mutex part_mu;
    int part_users = 0;
    int32_t hot0_max_id = -1;    // the largest feature id of the dense hot block's FORWARD page (page 0), -1 = none
    DevBuf<int32_t> cfeat, cptr, range_seg, split_seg, split_short, cdst, mp_feat, mp_ptr;
    // per batch: bitmap over the feature ids [0, dimension] of the rows whose gradient the FIXUP launch assembles (cut
    // columns, hot block) — the merged finish lets those update themselves and skips them in its dense pass
    DevBuf<uint32_t> own_bits;
    int64_t own_words = 0;       // words per batch
    DevBuf<int32_t> xlist;       // band-affine range lists of all batches (BatchMeta::xoff)
    std::vector<int32_t> h_xlist;   // host copy (a feature-interval launch searches the runs for its range window)
    std::vector<int32_t> h_cfeat, h_cptr, h_split, h_split_short;   // host copies (feature-chunked backward needs them)
    int64_t rb_rows = 0;       // rows per row block of the transposes (0 = not row-blocked)
    int32_t max_pieces = 0;
    // dense hot block: the entries of the most frequent features are held as dense [n_rows][kHotT] fp32 pages
    // (0 where the feature is absent).  Page 0 (up to kHotT features) is out of both sparse streams; pages 1.. are out
    // of the CSC stream only (fm_kernels.h, kHotPages)
    int32_t hot_T = 0;                 // 0 = no hot block
    int32_t hot_pages = 0;             // pages in use (0 = no hot block)
    int64_t hot_max_id = -1;           // the largest feature id held in any page
    std::vector<int32_t> hot_ids;      // [hot_pages * kHotT] feature id per slot, -1 = unused slot
    DevBuf<float> xhot;                // [hot_pages][n_rows][kHotT]
    DevBuf<int32_t> d_hot_ids;
    int64_t nnz_sparse = 0;            // entries of the CSR stream
    int64_t nnz_sparse_bwd = 0;        // entries of the CSC streams
    bool scoring_only = false;   // rows + labels only (fmhip_rows_create): no transposes, cannot train
    // fp64 copies of the values (CSR order, CSC order) and labels for the fp64 ALS learner; kept only
    // for single-batch datasets of at most kAlsMaxNnz stored nonzeros
    DevBuf<double> val64, cval64, y64;
    // ... and the rows with their entries sorted by feature id (the per-row order of the reference's transposed
    // q pass); als_dup: some row stores a feature twice (the column walk updates a row once per step: refused)
    DevBuf<int32_t> scol;
    DevBuf<double> sval64;
    bool als_dup = false;
    // level schedule of the ALS sweep (single-batch datasets): columns that share no row commute exactly, so the sweep may
    // take them side by side — level(c) = 1 + the largest level of an earlier column (smaller id) sharing a row with c.
    // als_lev_cols: the compressed columns sorted by (level, id); als_lev_ptr[l] .. [l + 1]: level l's slice of it
    DevBuf<int32_t> als_lev_cols;
    std::vector<int32_t> als_lev_ptr, h_als_lev_cols;
};

struct fmhip_model {
    template <typename T> using DevBuf = fmhip::host::DevBuf<T>;
    using ProfRec = fmhip::host::ProfRec;
    static constexpr int kGradHead = fmhip::kGradHead;
    // rows allocated (and kept zero) behind row n1p of V and of the library's own packed gradient: the sharded exchange
    // (fmhip_comm.hip) cuts [0, n+1) into `world` equal shares, so its last share may reach up to world - 1 rows past n1p
    static constexpr int kSlackRows = 64;
    // Threading rule of the C ABI (include/fmhip.h): calls that CHANGE a model (parameters, training, tuning, gradient
    // buffer, profiling, the data-parallel step) hold `mu` exclusively; the scoring calls and the parameter reads hold it
    // shared and work in a ScoreCtx of their own — they run side by side and never beside a writer.
    std::shared_mutex mu;
    std::mutex pool_mu;                                             // guards the two lists below
    std::vector<std::unique_ptr<fmhip::host::ScoreCtx>> ctx_all;
    std::vector<fmhip::host::ScoreCtx *> ctx_free;
    int device = 0;
    int64_t n = 0, n1 = 0, n1p = 0;
    int32_t k = 0, Kp = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf<float> V, w, w0;
    DevBuf<float> grad_own;
    float *grad = nullptr;        // packed gradient in use (own or bound)
    const fmhip::host::GradView *view = nullptr;   // set around a backward that writes elsewhere (see GradView)
    bool grad_dirty = false;      // holds a gradient that has not been applied/zeroed
    DevBuf<float> P, e, part, pieces, hot_part;
    DevBuf<float> part_sl;        // two-pass forward: pass A's {sum_f s_f, linear term} per row
    bool hot_pending = false;   // the dense hot block's gradient of the current step is still to be formed
    DevBuf<double> acc;           // {sum e, sum e^2, rows, nonfinite}
    DevBuf<double> bsum;          // k_forward's per-block statistic partials
    int64_t last_nnz = 0, last_rows = 0;
    int fwd_parts = 0;            // per-block statistic partials the last forward launch wrote
    // lazy weight decay (fm_apply.hip): the device tables hold U with V = sv*U, w = sw*(stored w); both are
    // exactly 1 unless rows-only updates with decay are pending.  Tracked in fp64 on the host, so the
    // scale itself accumulates no fp32 rounding from step to step.
    double sv = 1.0, sw = 1.0;
    int64_t bw_next_hi = -1;      // feature-chunked backward: the next interval must end here (-1: none pending)
    bool bw_up = false;           // ... ascending intervals instead (the next one must START here)
    // fp64 master copy of the parameters (reference layout): exact round trip of what the caller set,
    // and the state the fp64 ALS learner trains; stale once an fp32 SGD step has run
    std::vector<double> h_w, h_v;
    double h_w0 = 0.0;
    bool host64_fresh = false;
    DevBuf<double> als_w0, als_w, als_v, als_e, als_q, als_part;
    // per-model overrides of the tuning keys (fmhip_model_tune); -1 = the process-wide default (fmhip_tune) as it stands
    // at the time of the launch
    int tune[fmhip::kTuneCount];
    int tv(int key) const { return tune[key] >= 0 ? tune[key] : fmhip::tune_default(key); }
    bool profiling = false;
    bool prof_rotate = false;     // time one kernel kind per step, rotating
    int prof_period = 1;          // ... and only on every prof_period-th step
    int64_t prof_step = 0;
    std::vector<ProfRec> prof;

    // packed gradient: [ scalars (kGradHead floats, 8 used) | G_w (n1p) | G_b (n1p) | pad | G_V (n1p*Kp) ]: the
    // small head sits next to the G_V rows of the LOWEST feature ids, which the feature-chunked backward
    // finishes last, so a data-parallel host moves head + last interval in one collective
    size_t head_floats() const { return ((size_t)kGradHead + 2 * (size_t)n1p + 31) / 32 * 32; }
    float *scal() const { return grad; }
    float *Gw() const { return grad + kGradHead; }
    float *Gb() const { return grad + kGradHead + n1p; }
    float *GV() const { return grad + head_floats(); }
    int32_t pack_k() const { return k < Kp ? k : -1; }   // packed rows: slot k of a V row holds w_i
    size_t grad_floats() const { return head_floats() + (size_t)n1p * Kp; }
};


namespace fmhip {
namespace host {

extern thread_local std::string g_err;    // the calling thread's last error message (fmhip_api.hip)
int set_device(int device);
// the model's lock for the length of a C-ABI call (a NULL model is left to the call's own argument check)
struct WriteLock {
    std::unique_lock<std::shared_mutex> lk;
    explicit WriteLock(fmhip_model_t m) { if (m) lk = std::unique_lock<std::shared_mutex>(m->mu); }
};
struct ReadLock {
    std::shared_lock<std::shared_mutex> lk;
    explicit ReadLock(fmhip_model_t m) { if (m) lk = std::shared_lock<std::shared_mutex>(m->mu); }
};
// ---- fmhip_dataset.hip
int partition_rows_locked(fmhip_dataset_t d, int64_t cut_feature);      // fmhip_dataset_partition_rows with d->part_mu held by the caller
// ---- fmhip_step.hip
int ensure_workspace(fmhip_model_t m, fmhip_dataset_t d);
FwdArgs fwd_args(fmhip_model_t m, fmhip_dataset_t d, const BatchMeta &bm);
int check_pair(fmhip_model_t m, fmhip_dataset_t d);
int check_train(fmhip_model_t m, fmhip_dataset_t d);      // + the dataset must have its transposes
int check_batch(fmhip_dataset_t d, int64_t batch);
// the pieces of one mini-batch step, all asynchronous on m->stream (fmhip_api.hip)
int step_forward(fmhip_model_t m, fmhip_dataset_t d, int64_t b);
// a step whose update happens inside the backward (mode 1: every finished gradient row, FMHIP_TUNE_FUSED_UPDATE) or
// inside the fixup launch (mode 2, the merged finish: dense update beside the fixups, FMHIP_TUNE_MERGED_FINISH) — fmhip_api.hip
struct FusedPlan {
    int mode = 0;
    double eta = 0.0, reg0 = 0.0, regw = 0.0, regv = 0.0;
    double sv_out = 1.0, sw_out = 1.0;    // the tables' scales after the step
    FusedUpd upd{};
};
// Which of the two ranges that straddle a feature interval's edges a step_backward call walks: a straddling range belongs to
// whichever of its two intervals is walked FIRST (the other one's fixup then finds its partials).  Descending calls own their
// lower edge, ascending calls their upper one; a schedule that mixes the orders says so per call.
enum { kOwnLower = 1, kOwnUpper = 2 };
// the training forward in two passes (pass 0 = A: features below the dataset's split cut; pass 1 = B: the others + the row's finish)
int step_forward_pass(fmhip_model_t m, fmhip_dataset_t d, int64_t b, int pass);
int step_backward(fmhip_model_t m, fmhip_dataset_t d, int64_t b, int64_t feat_lo, int64_t feat_hi, bool finish, double *acc,
                  const FusedPlan *fused = nullptr, int own = kOwnLower);
// forward + backward + fixup of one batch into the packed gradient (fused: straight into the parameters)
int step_compute(fmhip_model_t m, fmhip_dataset_t d, int64_t b, double *acc, const FusedPlan *fused = nullptr);
// can this step's update run inside the fixup launch / the column walk?  (fills *p; false: a launch of its own, step_apply)
bool plan_fused(fmhip_model_t m, fmhip_dataset_t d, int64_t b, double eta, double reg0, double regw, double regv, FusedPlan *p);
int finish_fused(fmhip_model_t m, const FusedPlan &p);      // what step_apply leaves behind, for a step planned fused
int fold_scales(fmhip_model_t m);                           // lazily decayed tables back to scale 1
int read_acc(fmhip_model_t m, fmhip_stats *st);             // the fp64 epoch accumulators (synchronises)
int step_apply(fmhip_model_t m, double eta, double reg0, double regw, double regv, fmhip_dataset_t d = nullptr, int64_t b = -1);
int step_apply_interval(fmhip_model_t m, double eta, double reg0, double regw, double regv, int64_t lo, int64_t hi,
                        const float *rows, bool last);
// The sharded update of the feature interval [lo, hi) on stream `s` (one launch): V rows [vlo, vhi) (this rank's share) get
// the dense update, every linear weight of the interval is stepped, the G_V rows of [lo, hi_r) outside the share are zeroed
// (hi_r >= hi: the top interval's equal shares reach into the slack rows).  `last`: also steps w0 and closes the step.
int step_apply_shard(fmhip_model_t m, double eta, double reg0, double regw, double regv, int64_t lo, int64_t hi, int64_t hi_r,
                     int64_t vlo, int64_t vhi, const float *rows, bool last, hipStream_t s);
// the rows-only (lazy-decay) update of the feature rows listed on the device (ids < 0 are skipped), |B| from `rows`
// (device float): the touched-rows exchange of the data-parallel step applies the union of all ranks' rows with it
// view given: the gradient rows are read from (and zeroed in) its compact arrays, row j belonging to feature feat[j]
// off / last: the rows [off, off + n_feat) of the list (and of a compact view) only — the touched-rows exchange updates a
// feature interval as soon as its slice has arrived; the step's bookkeeping (w0, the tables' scale) moves with the LAST slice
int step_apply_rows(fmhip_model_t m, double eta, double reg0, double regw, double regv, const int32_t *feat, int32_t n_feat,
                    const float *rows, const GradView *view = nullptr, int64_t off = 0, bool last = true);
bool lazy_decay_ok(fmhip_model_t m, double eta, double regw, double regv);
int read_scal(fmhip_model_t m, fmhip_stats *st);

}  // namespace host
}  // namespace fmhip
