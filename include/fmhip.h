/*
 * fmhip.h — C ABI of the MI355X-native FM trainer (libfmhip.so).
 *
 * This is the drop-in boundary for SparkFM's learner plug-in point
 *     abstract class FMLearn { def learn(fm: FMModel, dataset: DataSet): FMModel }
 *     (S/fm/FMLearn.scala:10-12, invoked at S/fm/impl/FactorizationMachines.scala:45)
 * and for the scoring calls under it (FMModel.predict S/fm/FMModel.scala:34-63,
 * Model.computeRMSE S/Model.scala:13-19).  A JVM-side `HipSGD extends FMLearn`
 * binds exactly these symbols through JNI; the Python façade in
 * sparkfm_amd/ binds them through ctypes.
 *
 * S/ = src/main/scala/io/edstud/spark/ of edmundhung/SparkFM.
 *
 * This header is the PRODUCT surface: what INTEGRATION.md section 1 maps to a reference interface, plus the data-parallel
 * step.  Everything that exists to MEASURE or to experiment — tuning keys, per-kernel profiling, the exchange's timers and
 * emulation, layout queries, the two-pass forward on its own, a transport of the caller's instead of RCCL — lives in
 * fmhip_experimental.h (same library, same calling conventions; it includes this header).
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types; every call returns an int status
 *    (0 = FMHIP_OK, negative = error class) and never throws or aborts;
 *    fmhip_last_error() returns the calling thread's last message.
 *  - host parameter layout is the reference's: w has n+1 slots (n = num_attribute =
 *    largest feature index, S/fm/FMModel.scala:18); v is breeze's column-major
 *    DenseMatrix(k, n+1) (S/fm/FMModel.scala:19): element (f, i) at v[f + i*k].
 *  - host rows are CSR: row_ptr[n_rows+1] (int64), col[nnz] (int32, the breeze
 *    SparseVector `index` array, stored order, need not be sorted), val[nnz], y[n_rows]
 *    — the RDD[(Double, SparseVector[Double])] of S/DataSet.scala:42, flattened.
 *  - device arithmetic is fp32, indices int32; host I/O is fp64 (as the reference) or
 *    fp32 (the *_f32 entry points).
 *  - one model/dataset lives on ONE GPU.  Data-parallel training runs one process per
 *    GPU: fmhip_comm_create joins the ranks into an RCCL communicator and fmhip_dp_step
 *    runs forward -> backward -> all-reduce of the packed gradient (overlapped with the
 *    backward) -> update entirely inside the library — the reduction the reference's
 *    learner does itself inside `learn` (S/fm/lib/ALS.scala:153 `error.reduce(_+_)`).
 *    fmhip_dp_exchange switches the step to exchanging only the rows some rank touched
 *    (models far wider than a batch); fmhip_comm_create_external runs the same step over
 *    a collective of the caller's instead of RCCL.  The packed gradient is also exposed
 *    (fmhip_grad_*) for a host that orchestrates the step itself
 *    (fmhip_step_compute -> host all-reduce -> fmhip_step_apply).
 *  - threads.  Different handles are independent: any number of host threads may work on their own models / datasets /
 *    communicators at once, on one GPU or several (the reference's `local[*]` runs its tasks as threads of one JVM,
 *    S/driver.scala:14).  ONE model is guarded by a reader-writer lock inside the library: the scoring calls
 *    (fmhip_predict, fmhip_predict_rows, fmhip_rmse, fmhip_residual, fmhip_term_q) and the parameter reads
 *    (fmhip_model_get_params*, fmhip_model_get_rows, fmhip_model_info) are RE-ENTRANT — each call works on a stream and in
 *    a workspace of its own, so executor threads score through one frozen model side by side (S/Model.scala:14) and each
 *    gets the bits a lone caller would get; every call that changes the model (set_params, init, training, the split and
 *    the data-parallel step, gradient binding; tuning and profiling in fmhip_experimental.h) takes the lock exclusively and so runs alone, after the
 *    readers before it and before those behind it.  A dataset is immutable once created and may be shared by any number of
 *    threads.  A communicator belongs to the thread that drives its model.  Destroying a
 *    handle while another thread still uses it is the caller's bug.
 */
#ifndef FMHIP_H
#define FMHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMHIP_VERSION 500 /* 0.5.0: the measurement / experiment entry points moved to fmhip_experimental.h */

enum {
    FMHIP_OK = 0,
    FMHIP_ERR_INVALID = -1,      /* bad argument (NULL, negative size, unsorted row_ptr, index < 0 ...) */
    FMHIP_ERR_HIP = -2,          /* a HIP runtime call failed; message has hipGetErrorString */
    FMHIP_ERR_NOMEM = -3,        /* host or device allocation failed */
    FMHIP_ERR_SHAPE = -4,        /* dataset has a feature index > model.num_attribute */
    FMHIP_ERR_UNSUPPORTED = -5,  /* num_factor > FMHIP_MAX_FACTORS, batch too large ... */
    FMHIP_ERR_COMM = -6          /* RCCL could not be loaded, or a collective call failed */
};

#define FMHIP_MAX_FACTORS 256
/* CSC work split: every column of a batch is processed in pieces of this many entries */
#define FMHIP_RANGE_LEN 64

typedef struct fmhip_model *fmhip_model_t;
typedef struct fmhip_dataset *fmhip_dataset_t;
typedef struct fmhip_comm *fmhip_comm_t;

typedef struct fmhip_stats {
    double sse;        /* sum over processed rows of e^2, e = yhat - y (S/fm/lib/ALS.scala:143) */
    double sum_e;      /* sum of e */
    int64_t rows;      /* rows processed */
    int64_t nnz;       /* stored nonzeros processed */
    int64_t nonfinite; /* rows whose prediction was NaN/Inf (never masked; cf. S/fm/lib/ALS.scala:190-192) */
    int64_t steps;     /* mini-batch steps taken */
} fmhip_stats;

/* ---- library ----------------------------------------------------------------- */
int fmhip_version(void);
const char *fmhip_last_error(void);
int fmhip_device_count(int *count);

/* ---- model: `new FMModel(num_attribute, num_factor)`  S/fm/FMModel.scala:9-22 -- */
/* Parameters start at zero; the reference's unseeded N(0, 0.01) init (quirk Q2) is the
 * caller's job: draw on the host, then fmhip_model_set_params — or fmhip_model_init_normal.  `stream` is a
 * hipStream_t (NULL = the library creates its own non-blocking stream). */
int fmhip_model_create(int device, int64_t num_attribute, int32_t num_factor, void *stream, fmhip_model_t *out);
int fmhip_model_destroy(fmhip_model_t m);
/* padded_factors: floats per device row (whole 128-B lines: 32, 64, 128 or 256) */
int fmhip_model_info(fmhip_model_t m, int64_t *num_attribute, int32_t *num_factor, int32_t *padded_factors);
/* The reference's own initialisation, drawn on the device: w0 = 0, w = 0, v ~ N(mean, stdev)
 * (S/fm/FMModel.scala:17-22; its draw is unseeded — quirk Q2 — here `seed` makes it reproducible: element
 * (f, i) depends only on (seed, f, i)).  For models too wide to stage on the host (2^25 x 64: 17 GB fp64). */
int fmhip_model_init_normal(fmhip_model_t m, uint64_t seed, double mean, double stdev);
/* w: n+1 doubles, v: k*(n+1) doubles at v[f + i*k]  (FMModel.w0 / .w / .v, S/fm/FMModel.scala:17-19) */
int fmhip_model_set_params(fmhip_model_t m, double w0, const double *w, const double *v);
int fmhip_model_get_params(fmhip_model_t m, double *w0, double *w, double *v);
/* The parameters of `n` selected features only (ids need not be sorted or distinct): w[j] and the k factors
 * v[f + j*k] of feature ids[j] — what `fm.w(i)` / `fm.v(::, i)` read in the reference (S/fm/FMModel.scala:18-19).
 * For models too wide to copy whole. */
int fmhip_model_get_rows(fmhip_model_t m, int64_t n, const int32_t *ids, double *w, double *v);
int fmhip_model_set_params_f32(fmhip_model_t m, float w0, const float *w, const float *v);
int fmhip_model_get_params_f32(fmhip_model_t m, float *w0, float *w, float *v);
int fmhip_synchronize(fmhip_model_t m);

/* ---- dataset: DataSet(rdd).cache() + transposeInput  S/DataSet.scala:42-62,31-38 */
/* Copies the rows to the GPU, cuts them into mini-batches of `batch_rows` consecutive
 * rows (<= 0: one batch) and builds each batch's row->column transpose (the CSC the
 * atomics-free backward walks).  Caller keeps ownership of the host arrays. */
int fmhip_dataset_create(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                         const double *val, const double *y, int64_t batch_rows, fmhip_dataset_t *out);
int fmhip_dataset_create_f32(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                             const float *val, const float *y, int64_t batch_rows, fmhip_dataset_t *out);
/* The same with the layout choices spelled out per dataset instead of read from the process-wide tuning
 * keys FMHIP_TUNE_ROW_BLOCK and FMHIP_TUNE_HOT_BLOCK (fmhip_experimental.h; they remain the defaults, for A/B tools).  struct_size = sizeof(fmhip_dataset_opts). */
typedef struct fmhip_dataset_opts {
    int32_t struct_size;
    int32_t hot_block;       /* dense hot block: -1 = library default (FMHIP_TUNE_HOT_BLOCK, FMHIP_TUNE_HOT_PAGES), 0 = off, n >= 1 = on with up
                              * to n pages of 16 features (1 = the two-sided page only, 2..8 = gradient-side pages too).
                              * A dataset of ONE batch with at most 2^27 stored nonzeros is one fmhip_als_epoch can walk, and
                              * the default leaves its transpose whole (no block); n >= 1 asks for the block anyway — full-batch
                              * SGD, ~20 % faster at C4's width — and fmhip_als_epoch then refuses the dataset */
    int64_t batch_rows;      /* <= 0: one batch */
    int64_t row_block_rows;  /* rows per row block of the transposes: -1 = library default (FMHIP_TUNE_ROW_BLOCK), 0 = none */
} fmhip_dataset_opts;
int fmhip_dataset_create_opts(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                              const double *y, const fmhip_dataset_opts *opts, fmhip_dataset_t *out);
/* Rows + labels only, for scoring held-out data (`fm.computeRMSE(test)`, S/driver.scala:100-112): no
 * transposes, no mini-batches; y may be NULL (labels = 0: predict only).  Training calls refuse it. */
int fmhip_rows_create(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                      const double *y, fmhip_dataset_t *out);
int fmhip_rows_create_f32(int device, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const float *val,
                          const float *y, fmhip_dataset_t *out);
int fmhip_dataset_destroy(fmhip_dataset_t d);
/* size = rdd.count (S/DataSet.scala:23-25); dimension = max feature index (:27-29) */
int fmhip_dataset_info(fmhip_dataset_t d, int64_t *n_rows, int64_t *nnz, int64_t *dimension,
                       int64_t *batch_rows, int64_t *n_batches);
int fmhip_dataset_batch_info(fmhip_dataset_t d, int64_t batch, int64_t *row0, int64_t *rows, int64_t *nnz,
                             int64_t *n_columns);
/* Reads one batch's device-resident transpose back (parity of the index gathers):
 * feat[n_columns] ascending feature ids present in the batch, ptr[n_columns+1] offsets,
 * rows[nnz] batch-local row index (ascending inside a column), vals[nnz]. */
int fmhip_dataset_get_transpose(fmhip_dataset_t d, int64_t batch, int32_t *feat, int32_t *ptr, int32_t *rows,
                                float *vals);

/* ---- scoring ------------------------------------------------------------------ */
/* FMModel.predict mapped over the rows (S/fm/FMModel.scala:34-63; dataset.rdd.mapValues(predict)) */
int fmhip_predict(fmhip_model_t m, fmhip_dataset_t d, double *yhat /* n_rows */);
/* FMModel.predict(SparseVector) for ad-hoc host rows (S/fm/FMModel.scala:34): uploads, scores, frees */
int fmhip_predict_rows(fmhip_model_t m, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                       double *yhat /* n_rows */);
/* Model.computeRMSE (S/Model.scala:13-19) */
int fmhip_rmse(fmhip_model_t m, fmhip_dataset_t d, double *rmse, fmhip_stats *stats /* nullable */);
/* ALS.precomputeTermE (S/fm/lib/ALS.scala:142-144): e_r = yhat_r - y_r */
int fmhip_residual(fmhip_model_t m, fmhip_dataset_t d, double *e /* n_rows */);
/* ALS.precomputeTermQ for every factor (S/fm/lib/ALS.scala:146-150): q[r*k + f] */
int fmhip_term_q(fmhip_model_t m, fmhip_dataset_t d, double *q /* n_rows*k */);

/* ---- training (build-defined mini-batch SGD; SparkFM itself only ships ALS) ---- */
/*   g_theta = sum_{r in batch} e_r * h_r(theta),  h from S/fm/lib/ALS.scala:56-58 (V), :40 (w), :21 (w0)
 *   theta  <- theta - eta * (g_theta / |batch| + reg_theta * theta)                                  */
int fmhip_sgd_step(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, double eta, double reg0, double regw,
                   double regv, fmhip_stats *stats /* nullable: skips the host sync */);
/* one pass over all batches; order[n_batches] = batch visiting order (NULL = ascending) */
int fmhip_sgd_epoch(fmhip_model_t m, fmhip_dataset_t d, double eta, double reg0, double regw, double regv,
                    const int64_t *order, fmhip_stats *stats /* nullable */);
/* Gradient of one batch at the current parameters, host layout as the parameters
 * (gv[f + i*k], gw[i]); does not update anything. */
int fmhip_batch_grad(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, double *gv, double *gw, double *gw0,
                     fmhip_stats *stats);

/* ---- ALS: SparkFM's own learner -----------------------------------------------------
 * One ALS.learn pass (S/fm/lib/ALS.scala:15-75, closed-form steps :152-198) in fp64 on the GPU,
 * including the reference's quirk that slot `num_attribute` is never trained (:38,:52).  Needs a
 * single-batch dataset (batch_rows <= 0).  The parameters live in an fp64 master copy, so
 * fmhip_model_get_params returns the fp64 result exactly; the fp32 device copy used by the scoring
 * calls is refreshed from it. */
int fmhip_als_epoch(fmhip_model_t m, fmhip_dataset_t d, double reg0, double regw, double regv);

/* ---- data-parallel split step ---------------------------------------------------
 * packed fp32 gradient: [ scalars (32, 8 used) | G_w (n1p) | G_b (n1p) | pad to 32 | G_V (n1p*Kp) ],
 * n1p = n+1 rounded up to 4, Kp = padded factors; fmhip_grad_layout returns Kp and the offset of
 * G_V.  G_V holds sum e*x*q; G_b holds sum e*x^2 (the -x^2*v term of h is applied in
 * fmhip_step_apply, so the packed buffer is a plain sum over rows and all-reduces with `sum`).
 * scalars = {sum e, sum e^2, rows, nonfinite}.  The head (everything before G_V) lies next to the
 * G_V rows of the lowest feature ids — the interval a feature-chunked backward finishes last — so
 * the last collective of a step can cover head + interval in one message. */
int fmhip_grad_floats(fmhip_model_t m, int64_t *n_floats);
/* use caller-owned DEVICE memory (e.g. a torch tensor the host all-reduces); NULL = internal.
 * The buffer must be zero-filled by the caller before the first step. */
int fmhip_grad_bind(fmhip_model_t m, void *device_ptr);
int fmhip_grad_ptr(fmhip_model_t m, void **device_ptr);
int fmhip_step_compute(fmhip_model_t m, fmhip_dataset_t d, int64_t batch);
/* The same work in pieces, so the host can overlap the all-reduce with the backward:
 *   fmhip_step_forward                       forward of the batch
 *   fmhip_step_backward(.., lo, hi, finish)  gradient rows of the features lo <= id < hi; call it for
 *                                            disjoint intervals covering [0, n+1) in DESCENDING order
 *                                            (the cold, high-id features first: most of the gradient
 *                                            volume, least of the work) or, starting at feature 0, in
 *                                            ASCENDING order (same gradient, bit for bit); finish = 1 on
 *                                            the interval that starts at feature 0
 * After a call returns, floats [gv_offset + lo*row_floats, gv_offset + hi*row_floats) of the packed
 * buffer are final and can be all-reduced while the next interval computes; the head, floats
 * [0, gv_offset) (scalars | G_w | G_b), is final after the call with finish = 1. */
int fmhip_step_forward(fmhip_model_t m, fmhip_dataset_t d, int64_t batch);
int fmhip_step_backward(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, int64_t feat_lo, int64_t feat_hi,
                        int finish);
int fmhip_grad_layout(fmhip_model_t m, int64_t *row_floats, int64_t *gv_offset);
/* applies the packed gradient (after the host's all-reduce, if any), then zeroes it */
int fmhip_step_apply(fmhip_model_t m, double eta, double reg0, double regw, double regv);
/* scalars of the packed gradient as last computed/all-reduced (synchronises) */
int fmhip_step_stats(fmhip_model_t m, fmhip_stats *stats);

/* ---- data-parallel training inside the library (RCCL over xGMI) ------------------------------------
 * One process (or host thread) per GPU, each with its own model replica and its own row shard.  The
 * reference's learner reduces inside `learn` (S/fm/lib/ALS.scala:153, :34, :139) and is called once per
 * iteration by the driver (S/fm/impl/FactorizationMachines.scala:45); likewise a `HipSGD.learn` on N GPUs
 * calls fmhip_dp_epoch on every rank and nothing else.  RCCL (librccl.so.1) is loaded on first use;
 * single-GPU users never need it.
 *
 *   rank 0:    fmhip_comm_unique_id(id)          -> ship the 128 bytes to every rank (Spark broadcast, ...)
 *   each rank: fmhip_comm_create(m, id, rank, world, &c)
 *              fmhip_dp_epoch(m, d, c, ...)  or  fmhip_dp_step(m, d, batch, c, ...) in lock-step
 *
 * A step: forward of this rank's mini-batch; backward of the coldest (highest-id) feature interval; its
 * slice of the packed gradient is all-reduced on a second stream while the next interval's backward runs;
 * the last message carries the head with the hottest interval; every rank applies the identical update
 * with |B| = the summed row count, so the replicas stay bit-identical.  `batch` < 0: this rank has run
 * out of rows and contributes zeros.  All ranks must call with the same (eta, reg*) and the same cuts
 * (fmhip_dp_plan).  The summed row count travels as one fp32 word: a global batch (rows x world) must stay
 * below 2^24 rows — fmhip_dp_plan agrees the largest batch of any rank and refuses on EVERY rank alike; a
 * check that can only fail on one rank (a batch the plan has not seen) makes that rank contribute zeros to
 * the step and return the error afterwards, so no peer is left waiting in a collective. */
#define FMHIP_UNIQUE_ID_BYTES 128
int fmhip_comm_unique_id(void *id /* FMHIP_UNIQUE_ID_BYTES out */);
int fmhip_comm_create(fmhip_model_t m, const void *id, int rank, int world, fmhip_comm_t *out);
int fmhip_comm_destroy(fmhip_comm_t c);
int fmhip_comm_info(fmhip_comm_t c, int *rank, int *world);
/* Collective (every rank calls it): known patterns through each collective kind the plan and the step issue, on the
 * communicator's own stream with the step's own calls (the grouped three-region all-reduce of a gradient slice, the
 * in-place reduce-scatter / all-gathers of the sharded update, the plan's int64 maximum / broadcast and id all-gather).
 * All ranks get the same verdict: FMHIP_OK, or FMHIP_ERR_COMM with bit FMHIP_COLL_* of *failed_kinds (may be NULL) set for
 * every kind that left wrong elements on ANY rank.  A caller runs it once after fmhip_comm_create — a transport that moves
 * the wrong elements fails here and not as replicas that quietly drift apart (the reference's reductions cannot go wrong
 * this way: S/fm/lib/ALS.scala:153 is a JVM-side reduce). */
int fmhip_comm_selftest(fmhip_comm_t c, int *failed_kinds);
/* The collective kinds a step and a plan issue (the bits of fmhip_comm_selftest's failed_kinds; also what a transport of the
 * caller's own implements, fmhip_comm_create_external in fmhip_experimental.h):
 *   FMHIP_COLL_SUM_F32     element-wise sum of `count` floats (the gradient slices, the row count)
 *   FMHIP_COLL_MAX_I64     element-wise maximum of `count` int64 (step counts)
 *   FMHIP_COLL_BCAST0_I64  every rank receives rank 0's `count` int64 (the cuts) */
#define FMHIP_COLL_SUM_F32 0
#define FMHIP_COLL_MAX_I64 1
#define FMHIP_COLL_BCAST0_I64 2
#define FMHIP_COLL_ALLGATHER_I32 3 /* device_buf holds world x count int32, rank r's own at r * count: fill in the others' */
/* the sharded update (FMHIP_EXCHANGE_SHARDED): device_buf holds world x count floats, segment r at r * count */
#define FMHIP_COLL_REDUCE_SCATTER_F32 4 /* on return rank r's segment holds the sum over all ranks of that segment (the other segments: unspecified) */
#define FMHIP_COLL_ALLGATHER_F32 5      /* rank r's own segment is in place: fill in the others' */
/* What travels in a data-parallel step (set on every rank, before fmhip_dp_plan):
 *   FMHIP_EXCHANGE_DENSE    (default) the whole packed gradient, 4(n+1)(k+1) bytes, in slices overlapped with the backward —
 *                           north_star's dense all-reduce; right when a global batch touches most of the model (C4)
 *   FMHIP_EXCHANGE_TOUCHED  only the gradient rows some rank's mini-batch touched.  A dataset's mini-batches are fixed, so
 *                           fmhip_dp_plan forms ONCE, for every step t of the lock-step schedule "step t = every rank's batch t",
 *                           the sorted union U_t of the feature ids those batches touch (all-gather of the ids, sort, unique)
 *                           and where this rank's columns lie in it; a step then writes its gradient straight into a COMPACT
 *                           buffer [scalars | G_w | G_b | G_V rows of U_t], all-reduces that buffer (its size is known on the
 *                           host: no read-back, no synchronisation) and applies the rows-only update (weight decay rides in
 *                           the tables' scale: 0.5 <= 1 - eta*reg <= 1 required).  For models far wider than a global batch —
 *                           C5's 2^25 x 64 gradient is 8.9 GB dense and ~0.1 GB here.  The compact rows are sorted by feature id,
 *                           so the plan's cuts apply here too: the all-reduce of an interval's slice of rows runs beside the
 *                           backward of the next interval and the interval is updated when its slice has arrived, as in the dense
 *                           mode.  The plan is per POSITION of the lock-step schedule (position t = every rank's batch t):
 *                           fmhip_dp_epoch walks the positions in order, fmhip_dp_epoch_order / fmhip_dp_step_at in any order every
 *                           rank names alike; fmhip_dp_step takes batch t at step t (or -1 on a rank without it), in order.
 *   FMHIP_EXCHANGE_SHARDED  the dense exchange with the UPDATE sharded too: each interval's G_V slice is reduce-scattered (rank r
 *                           receives the summed rows of its 1/world share of the interval), rank r updates just those rows of V and
 *                           zeroes them, the updated rows are all-gathered in place into every replica's V.  Same bytes on the wire as
 *                           the all-reduce (it IS a ring all-reduce cut in two), but the update and the zeroing of the gradient — 4 x
 *                           the model's bytes, paid by EVERY rank in the dense mode — shrink world-fold, and the replicas are identical
 *                           by construction (one writer per row).  G_w / G_b (1/32 of the bytes) are still all-reduced and every rank
 *                           steps all of w.  Interval edges are rounded to multiples of `world`; needs the library's own gradient
 *                           buffer (fmhip_grad_bind: only if n+1 is a multiple of world) and world <= 64.
 * fmhip_dp_exchange_info: the mode, the id slots per rank agreed by the plan, the mean |U_t| over the planned steps. */
#define FMHIP_EXCHANGE_DENSE 0
#define FMHIP_EXCHANGE_TOUCHED 1
#define FMHIP_EXCHANGE_SHARDED 2
/*   FMHIP_EXCHANGE_PIPELINED the dense exchange, CONSECUTIVE STEPS OVERLAPPED: the coldest interval (most of the bytes, a few per cent
 *                           of the work) is walked and sent LAST — the others before it, from the second-coldest down to feature 0,
 *                           updated as they arrive — and while that slice travels the NEXT position's forward runs over every
 *                           feature below the top cut (the two-pass forward, fmhip_step_forward_pass; the run partitions the rows'
 *                           entries at fmhip_dp_plan's top cut).  When the slice has arrived its rows are updated and the second
 *                           pass finishes the rows.  The same sums and the same update as the dense mode (the forward's fp32 sums in
 *                           another order); needs at least one cut and a model of up to 64 padded factors (else: the dense step).
 *                           The overlap needs the next position: fmhip_dp_epoch / _epoch_order / fmhip_dp_steps have it, a single
 *                           fmhip_dp_step(_at) is the same step without it. */
#define FMHIP_EXCHANGE_PIPELINED 3
int fmhip_dp_exchange(fmhip_comm_t c, int mode);
int fmhip_dp_exchange_info(fmhip_comm_t c, int *mode, int64_t *id_slots_per_rank, double *mean_union_rows);
/* Chooses the feature ids that cut the backward into intervals and broadcasts them from rank 0 (collective).
 * upper_fractions[i], ascending: the share of rank 0's stored nonzeros that lies at or above cut i — e.g.
 * {0.25} = two intervals, the first (ids >= cut) a quarter of the work and nearly all of the gradient's
 * bytes; {0.15, 0.5} = three.  n_fractions = 0: whole backward, one all-reduce.  cuts (nullable, room for
 * n_fractions): the ids chosen, first cut first (0 = that cut collapsed).  If any rank's dataset was built
 * row-blocked (fmhip_dataset_opts.row_block_rows) nobody cuts: every rank issues the same collectives. */
#define FMHIP_DP_MAX_CUTS 7
int fmhip_dp_plan(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, int n_fractions, const double *upper_fractions,
                  int64_t *cuts);
int fmhip_dp_step(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0,
                  double regw, double regv);
/* max over ranks of the local batch count steps, ascending batches; stats (nullable) = the GLOBAL
 * sums over all ranks of the last step {sse, sum_e, rows, nonfinite} and steps taken */
int fmhip_dp_epoch(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, double eta, double reg0, double regw,
                   double regv, fmhip_stats *stats);
/* One step at a POSITION of the lock-step schedule that every rank names alike: this rank's batch `position` if it has that
 * many, a zero contribution otherwise.  What a permuted epoch calls (HipSGD.shuffle_seed, S/fm/lib/ALS.scala has no
 * counterpart): unlike fmhip_dp_step, a rank without rows still says WHICH position the step is, so the touched-rows exchange
 * can pick that position's union. */
int fmhip_dp_step_at(fmhip_model_t m, fmhip_dataset_t d, int64_t position, fmhip_comm_t c, double eta, double reg0,
                     double regw, double regv);
/* fmhip_dp_epoch with the positions visited in the caller's order: order[n_order] = a permutation of [0, steps), steps = the
 * largest batch count of any rank (fmhip_dp_plan_info), the SAME array on every rank.  order = NULL: ascending. */
int fmhip_dp_epoch_order(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, double eta, double reg0, double regw,
                         double regv, const int64_t *order, int64_t n_order, fmhip_stats *stats);
/* what the last fmhip_dp_plan agreed over all ranks: the lock-step steps of an epoch (the largest batch count of any rank)
 * and the largest mini-batch (rows) */
int fmhip_dp_plan_info(fmhip_comm_t c, int64_t *steps, int64_t *max_batch_rows);
/* `n` steps in one call, at the named positions of the lock-step schedule (every rank the same list): what a caller that knows
 * its next steps hands the pipelined exchange, which overlaps each step's last slice with the next position's forward; in the
 * other modes the same steps as n calls of fmhip_dp_step_at.  In EVERY mode a position whose batch fails a check only this rank
 * can see (a dataset the plan has not seen) contributes zeros and the run goes on to its last position — the peers are never
 * left alone in a later position's collectives; the first such error is returned afterwards. */
int fmhip_dp_steps(fmhip_model_t m, fmhip_dataset_t d, const int64_t *positions, int64_t n, fmhip_comm_t c, double eta, double reg0,
                   double regw, double regv);
/* Contiguous row shard [lo, hi) of `rank`, balanced by stored nonzeros (not by row count): the
 * partitioning a data-parallel caller applies before fmhip_dataset_create (pure host arithmetic). */
int fmhip_shard_rows(int64_t n_rows, const int64_t *row_ptr, int world, int rank, int64_t *lo, int64_t *hi);

/* ---- feature relabelling by frequency (pure host arithmetic, no GPU needed) ------------ */
/* The kernels do best when small feature ids are the frequent ones: the forward stages the linear weights of the
 * lowest ids in LDS, frequent rows then share cache lines and pages, and fmhip_dp_plan cuts the backward by id so
 * that the first interval is "few nonzeros, most of the gradient's rows".  Ids that come in another order (hashed,
 * dictionary order) cost ~20 % of the forward (tools/id_order_time.py).  A caller that wants the fast layout
 * relabels: rank = position of a feature in descending order of its stored-nonzero count (ties: ascending id), the
 * dataset is created from rank[col], and parameters move between the two numberings with `by_rank`
 * (internal row r = the caller's feature by_rank[r]).  A pure renaming: the model and its updates are the same.
 * Not for fmhip_als_epoch, whose Gauss-Seidel sweep runs in id order (S/fm/lib/ALS.scala:38,52).
 *
 *   fmhip_feature_counts   counts[c] += occurrences of c in col[0..nnz)   (caller zeroes `counts`; call once per
 *                          partition, or sum the tables of all ranks: every rank must end up with the SAME order)
 *   fmhip_rank_from_counts rank[n1] and, if not NULL, its inverse by_rank[n1]
 *   fmhip_relabel_columns  out[i] = rank[col[i]]; `out` may be `col` itself */
int fmhip_feature_counts(int64_t nnz, const int32_t *col, int64_t n1, int64_t *counts);
int fmhip_rank_from_counts(int64_t n1, const int64_t *counts, int32_t *rank, int32_t *by_rank);
int fmhip_relabel_columns(int64_t nnz, const int32_t *col, int64_t n1, const int32_t *rank, int32_t *out);
/* The same three steps ON THE GPU `device` (a histogram by atomic adds, a stable descending radix sort of (count, id), a gather;
 * the ids travel in chunks of 2^26): same arguments, same results BIT FOR BIT — the same order for the same counts, ties by
 * ascending id — for hosts whose cores are the slow part (2^25 slots x 210 M ids: 9.7 s of host arithmetic, under half a
 * second here).  The host versions above stay the reference and need no GPU. */
int fmhip_feature_counts_gpu(int device, int64_t nnz, const int32_t *col, int64_t n1, int64_t *counts);
int fmhip_rank_from_counts_gpu(int device, int64_t n1, const int64_t *counts, int32_t *rank, int32_t *by_rank);
int fmhip_relabel_columns_gpu(int device, int64_t nnz, const int32_t *col, int64_t n1, const int32_t *rank, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* FMHIP_H */
