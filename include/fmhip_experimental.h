/*
 * fmhip_experimental.h — the measurement and experiment surface of libfmhip.so.
 *
 * Nothing here is needed to train or score (include/fmhip.h is the product surface: what INTEGRATION.md section 1 maps to a
 * SparkFM interface, plus the data-parallel step).  These entry points exist so that the library can be MEASURED and its
 * variants compared on the same binary: named tuning keys, per-kernel HIP-event profiling, the exchange's own timers and
 * the emulation of collectives on a one-GPU box, how a dataset was laid out, the two-pass forward on its own, and the
 * data-parallel step over a transport of the caller's instead of RCCL (how the test suite runs 2-8 ranks on one GPU).
 * Same library, same conventions (int status codes, fmhip_last_error, never throws).  They may change between versions.
 */
#ifndef FMHIP_EXPERIMENTAL_H
#define FMHIP_EXPERIMENTAL_H
#include "fmhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- build identity ----------------------------------------------------------------- */
/* 0 in a library anyone may train with.  Non-zero: this build carries a timing-only ablation of a kernel (a part of the
 * arithmetic or of the memory traffic compiled out to measure what it costs, FMHIP_EXP_* in the kernel sources): its results
 * are wrong by construction and it exists only as an A/B variant beside the real library (tools/build_variant.sh). */
int fmhip_ablation_mask(void);

/* ---- tuning keys -------------------------------------------------------------------------
 * fmhip_tune sets the PROCESS-WIDE DEFAULT of a key, fmhip_model_tune overrides it for one model (value < 0: back to the
 * default); a launch reads the model's value if it has one, else the default as it stands then.  Results are identical across
 * variants up to fp32 rounding.  Each key is an atomic word: any thread may set or read one at any time (a launch that is
 * being prepared sees the old or the new value, never a torn one). */
typedef enum fmhip_tune_key {
    /* forward kernel: 60 = LDS w-tile (default: the linear weights of the 6144 lowest feature ids are staged in LDS), 0 = plain
     * global-memory gathers, 20 = LDS V-tile (rows of the lowest-id features of V staged in LDS); the tiles only help when ids
     * are frequency-ranked (hot = low id), results are the same either way */
    FMHIP_TUNE_FORWARD_KERNEL = 0,
    /* backward kernel: 1 = pipelined buffer-load walk (default), 0 = plain walk */
    FMHIP_TUNE_BACKWARD_KERNEL = 1,
    /* LDS tile rows: 0 = auto (V-tile: as many rows as fit 128 KiB; w-tile: 6144) */
    FMHIP_TUNE_TILE_ROWS = 2,
    /* default rows per row block of the transposes fmhip_dataset_create builds (0 = off): entries sorted by (row block,
     * feature) so a block's slice of P stays L2-resident in the backward; features occurring in several blocks are summed by
     * an extra fixup pass.  A DEFAULT of fmhip_dataset_create only (fmhip_dataset_opts states it per dataset) */
    FMHIP_TUNE_ROW_BLOCK = 3,
    /* placement of the backward's workgroups on the eight XCDs (each has its own 4 MiB L2): 2 = band-affine (default):
     * fmhip_dataset_create plans, per batch of at least 1024 ranges, one range list per XCD that starts with the ranges of
     * long columns whose rows fall into that XCD's own row bands (a column's entries ascend by row, so such a range gathers P
     * rows from a 2 MB band that stays in that L2); 0 = ranges in stream order; 1 = XCD x walks the x-th eighth of the stream
     * (pays with row-blocked transposes, FMHIP_TUNE_ROW_BLOCK) */
    FMHIP_TUNE_XCD_PLACEMENT = 4,
    /* default for the dense hot block of the datasets fmhip_dataset_create builds (1 = on; 0 = off): in a dataset of more than
     * one mini-batch the (at most 16) features present in >= 10 % of the rows — none that occurs twice in a row or with a
     * stored zero — leave the sparse streams for a dense [rows][16] fp32 array; their V rows are served from LDS in the
     * forward and their gradient rows are a small dense product (MFMA) in the backward.  Invisible at the product interface:
     * batch_info, get_transpose, statistics and gradients report every stored nonzero.  A DEFAULT of fmhip_dataset_create only */
    FMHIP_TUNE_HOT_BLOCK = 5,
    /* cap on the forward's resident workgroups per CU (0 = all that fit, default): measurement knob */
    FMHIP_TUNE_FORWARD_OCCUPANCY = 6,
    /* models with k > 32: the forward walks each batch's rows longest-first (1 = default; 0 = stored order), so the slots of a
     * wave walk rows of equal length; narrower models gain nothing from it and always walk in stored order */
    FMHIP_TUNE_ROW_ORDER = 7,
    /* 1 = take the flat 64-bit-address kernels that tables of 4 GiB and more need (V in the forward, P in the backward)
     * whatever the size; 0 = by size (default).  Test knob: reaches the paths of Criteo-width models on small inputs */
    FMHIP_TUNE_FLAT_ADDRESS = 8,
    /* lazy weight decay (1 = on, default): the fused step (fmhip_sgd_step / _epoch) updates only the rows a batch touched
     * even with regw/regv > 0 — the decay of every row rides in a scale factor of the tables (fm_apply.hip); 0 = dense update
     * whenever there is decay.  Equal up to fp32 rounding */
    FMHIP_TUNE_LAZY_DECAY = 9,
    /* fused update (0 = off, default): 1 = the fused step applies every finished gradient row to its parameter row inside the
     * backward / fixup launches whenever the rows-only update is legal.  Bit-identical to the separate update launch; off by
     * default because the read-modify-write of the parameter row sits in the column walk's dependent chain */
    FMHIP_TUNE_FUSED_UPDATE = 10,
    /* merged finish (1 = on, default): when the fused step's update is the dense pass it runs inside the fixup launch — the
     * rows the fixups assemble update themselves from registers, every other row is updated by extra workgroups beside them —
     * instead of as a launch of its own.  Bit-identical; 0 = separate update launch */
    FMHIP_TUNE_MERGED_FINISH = 11,
    /* pages of the dense hot block (1..8, default 4): 1 = the two-sided page only; more = the next most frequent features
     * that pass the density test are dense on the gradient side (fmhip_dataset_hot_pages).  Models of up to 32 (padded)
     * factors form up to 8 pages in one pass over P, wider ones 4 pages per pass.  A DEFAULT of fmhip_dataset_create only */
    FMHIP_TUNE_HOT_PAGES = 12,
    FMHIP_TUNE_KEY_COUNT = 13
} fmhip_tune_key;
/* FMHIP_TUNE_ROW_BLOCK, _HOT_BLOCK and _HOT_PAGES are only the DEFAULTS of fmhip_dataset_create (read at the time of the
 * call; they decide the layout of the dataset being built and nothing else) — fmhip_dataset_create_opts states them per
 * dataset and fmhip_model_tune refuses them; every other key is read by the next launch. */
int fmhip_tune(int key /* fmhip_tune_key */, int value);
/* overrides a tuning key for THIS model only; value < 0 = follow the process-wide default again */
int fmhip_model_tune(fmhip_model_t m, int key /* fmhip_tune_key */, int value);

/* ---- per-kernel device time --------------------------------------------------------- */
/* per-kernel device time, HIP events on the model's stream (fmhip_profile_*).  FMHIP_K_REDUCE is
 * kept for ABI stability: the statistics reduction now runs inside the fixup launch. */
enum { FMHIP_K_FORWARD = 0, FMHIP_K_REDUCE = 1, FMHIP_K_BACKWARD = 2, FMHIP_K_FIXUP = 3, FMHIP_K_APPLY = 4, FMHIP_K_COUNT = 5 };
typedef struct fmhip_profile {
    double ms[FMHIP_K_COUNT];        /* summed elapsed per kernel kind */
    int64_t launches[FMHIP_K_COUNT];
    int64_t nnz[FMHIP_K_COUNT];      /* stored nonzeros those launches covered */
    int64_t rows[FMHIP_K_COUNT];
    int64_t steps[FMHIP_K_COUNT];    /* steps in which the kind was timed: a data-parallel step launches its backward / fixup /
                                      * update once per feature interval, so ms / steps is the kind's time PER STEP */
} fmhip_profile;
int fmhip_profile_begin(fmhip_model_t m);                  /* start recording HIP events around every kernel */
/* same, but each SGD step times only ONE kernel kind, rotating forward -> backward -> fixup ->
 * apply from step to step: 2 event records per step instead of 8, so the timed region
 * is barely perturbed (event records cost ~4 us each on the stream) */
int fmhip_profile_begin_rotating(fmhip_model_t m);
/* the same on every `period`-th step only (step 0 forward, step `period` backward, ...): one pair of event
 * records per `period` steps — period 4 keeps the perturbation of a 0.3 ms step under 1 % */
int fmhip_profile_begin_sampled(fmhip_model_t m, int period);
int fmhip_profile_end(fmhip_model_t m, fmhip_profile *p);  /* synchronise, sum, stop recording */

/* ---- how the library laid a dataset out (byte accounting, parity tests) ------------------ */
/* How the library laid a dataset out (for byte accounting; not needed to use it): the number of
 * features held in the dense hot block (0 = none) with their ids (ids: room for 16, nullable), and
 * the stored nonzeros that stayed in the sparse streams. */
int fmhip_dataset_layout(fmhip_dataset_t d, int32_t *n_hot, int32_t *hot_ids, int64_t *nnz_sparse);
/* The dense hot block has up to FMHIP_HOT_PAGES pages of 16 features.  Page 0 (what fmhip_dataset_layout reports)
 * is dense for the forward and the backward; the features of pages 1.. are dense on the gradient side only: their
 * entries stay in the rows the forward walks and leave the transposes the backward walks.  n_pages; n_ids and ids
 * (room for 16 * FMHIP_HOT_PAGES, nullable): the features of ALL pages; nnz_sparse_backward: the stored nonzeros left
 * in the transposes. */
#define FMHIP_HOT_PAGES 8
int fmhip_dataset_hot_pages(fmhip_dataset_t d, int32_t *n_pages, int32_t *n_ids, int32_t *ids, int64_t *nnz_sparse_backward);
/* The band-affine plan of the backward's ranges (fmhip_tune key 4) summed over the batches: ranges (64-entry pieces of the
 * transposes) in all, those with a plan, and those placed by the row band they cover (the rest fill the XCDs' lists evenly). */
int fmhip_dataset_band_plan(fmhip_dataset_t d, int64_t *n_ranges, int64_t *planned_ranges, int64_t *band_affine_ranges);
/* The sweep's level schedule, made when a single-batch dataset is created: the reference walks the features in id order
 * (S/fm/lib/ALS.scala:38,52) and every step sees the residuals the previous one left — but two columns WITHOUT A COMMON ROW
 * touch disjoint residuals, so their steps commute exactly.  level(c) = 1 + the largest level of an earlier column sharing a
 * row with c; fmhip_als_epoch takes the levels one launch each, all columns of a level side by side, whenever they hold 16
 * columns or more on average (one-hot fields — the reference's MovieLens demo, S/driver.scala:73-113 — give one level per
 * field) and leaves the same bits as the sequential walk.  n_levels = 0: no schedule (not a single-batch dataset). */
int fmhip_dataset_als_levels(fmhip_dataset_t d, int64_t *n_levels, int64_t *n_columns, int64_t *widest_level);

/* ---- the two-pass forward on its own (the pipelined exchange runs it by itself) ---------- */
/* The same forward in TWO passes over every row's entries — pass 0: the features below a cut, pass 1: the others and the row's
 * finish — so that pass 0 can run while the rows of V at or above the cut are still being exchanged (the pipelined
 * data-parallel schedule, FMHIP_EXCHANGE_PIPELINED).  fmhip_dataset_partition_rows(d, cut) prepares the dataset: a stable
 * partition of each row's stored entries at feature id `cut`, made in a COPY of the stream that only the two-pass forward reads
 * (the dataset's own streams never move: other threads may go on scoring and training with it; one partition per dataset — it
 * is re-made for another cut, but not while a pipelined run of some model is walking it: that call fails and says so).  A
 * feature of the dense hot block's forward page at or above the cut moves the block's prologue from pass 0 to pass 1.  Pass 0
 * then pass 1 = fmhip_step_forward up to the order of the fp32 sums; models of up to 64 padded factors.  (FMModel.predict's sum over a row's entries, S/fm/FMModel.scala:41-46,57-63, taken in two parts.) */
int fmhip_dataset_partition_rows(fmhip_dataset_t d, int64_t cut_feature);
int fmhip_step_forward_pass(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, int pass);

/* ---- the data-parallel step over a transport of the caller's own ------------------------- */
/* The same communicator over a transport of the caller's own instead of RCCL (MPI, UCX, a JVM-side channel; the
 * two-ranks-on-one-GPU test of this repo stages through the host and torch.distributed/gloo).  `fn` is called from
 * fmhip_dp_step / _plan / _epoch on the calling thread and must leave in EVERY rank's `device_buf` the result over all
 * ranks, in place, ordered after the work already queued on `hip_stream` (a hipStream_t) and before anything queued
 * later — it may enqueue its own kernels there, or wait for the stream and work from the host (the overlap with the
 * backward is then lost, the result is the same).  All ranks see the same sequence of calls.  Return 0 = done.
 *   (the kinds: FMHIP_COLL_* in fmhip.h) */
typedef int (*fmhip_collective_fn)(void *ctx, void *device_buf, size_t count, int kind, void *hip_stream);
int fmhip_comm_create_external(fmhip_model_t m, int rank, int world, fmhip_collective_fn fn, void *ctx, fmhip_comm_t *out);
/* What a host-staged transport needs and cannot reach from the JVM / ctypes by itself: wait for a stream; copy
 * device -> host / host -> device behind the work queued on the stream (both return when the copy has finished). */
int fmhip_stream_wait(void *hip_stream);
int fmhip_device_read(void *host_dst, const void *device_src, size_t bytes, void *hip_stream);
int fmhip_device_write(void *device_dst, const void *host_src, size_t bytes, void *hip_stream);

/* ---- the exchange's timers, and collectives emulated on a one-GPU box -------------------- */
/* device time of the exchange as the compute stream saw it (HIP events, summed over the steps since
 * _begin): exposed_ms = time the update waited for the last collective after the backward had finished;
 * comm_ms = busy time of the collectives on their own stream; bytes = payload all-reduced per rank */
typedef struct fmhip_comm_profile {
    double exposed_ms, comm_ms;
    int64_t steps, bytes;
} fmhip_comm_profile;
/* Measurement aid for boxes with fewer GPUs than the job: after every collective the comm stream is held
 * for payload_bytes / (payload_gb_per_s GB/s) by a one-wave delay kernel — the time a real all-reduce of
 * that payload would take at that rate — so the overlap schedule can be timed with one rank.  0 = off.
 * (Optimistic: a real collective also takes CUs and memory bandwidth from the backward beside it.) */
int fmhip_comm_emulate(fmhip_comm_t c, double payload_gb_per_s);
/* ... with the collective's FOOTPRINT on this GPU instead of an idle wait: for the emulated duration `workgroups` workgroups
 * (RCCL keeps a few dozen resident) stream the payload through HBM — read and written back unchanged, twice for an all-reduce,
 * once for a reduce-scatter or an all-gather — so the backward beside it loses the CU slots and the memory bandwidth a real
 * collective takes.  A grouped call (an interval's three regions) is ONE such kernel: the call's whole duration, its footprint on
 * the largest region — as RCCL runs a group.  0 = the idle wait (default). */
int fmhip_comm_emulate_load(fmhip_comm_t c, int workgroups);
/* ... and for the sharded update: pretend to be rank 0 of `ranks` (one real rank only): intervals are cut into `ranks` shares,
 * this rank updates and zeroes only the first, the reduce-scatter / all-gather delays are those of `ranks` GPUs (half an
 * all-reduce each).  The rows of the other shares are NOT updated — a timing aid, not a training mode.  0 = off. */
int fmhip_comm_emulate_ranks(fmhip_comm_t c, int ranks);
int fmhip_comm_profile_begin(fmhip_comm_t c);
int fmhip_comm_profile_end(fmhip_comm_t c, fmhip_comm_profile *p);

#ifdef __cplusplus
}
#endif
#endif /* FMHIP_EXPERIMENTAL_H */
