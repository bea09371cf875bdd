// fm_kernels.h — launchers of the gfx950 kernels (internal to libfmhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "fm_constants.h"   // kRangeLen, kXcds, kXSegs, kRowBands, kExtend

namespace fmhip {

constexpr int kPartPad = 4;        // partial row = Kp floats + {sum e*x, sum e*x^2, pad, pad}
constexpr int kScalars = 8;        // packed-gradient tail: {sum e, sum e^2, rows, nonfinite, ...}
constexpr int kGradHead = 32;      // floats reserved for them at the front of the packed gradient (one 128-B line)

// runtime kernel-variant knobs (diagnostics / A-B benchmarking; fmhip_tune)
enum { kTuneFwd = 0, kTuneBwd = 1, kTuneTile = 2, kTuneRowBlock = 3, kTuneXcd = 4, kTuneHot = 5, kTuneFwdOcc = 6, kTuneRowOrder = 7,
       kTuneFlat = 8, kTuneLazy = 9, kTuneFused = 10, kTuneMerged = 11, kTuneHotPages = 12, kTuneCount = 13 };
constexpr int kHotT = 16;           // slots of one page of the dense hot block (fp32 per row: one 64-B half line)
// Pages of the dense hot block.  Page 0 (the 16 most frequent features) is dense on BOTH sides: its entries leave the
// CSR and the CSC streams.  Pages 1.. (the next most frequent ones that still pass the density test) are dense on the
// GRADIENT side only: their entries stay in the CSR stream the forward walks (a longer dense prologue costs the forward
// its occupancy — profiles/r02_experiments.md §18) but leave the CSC stream, where every entry costs the backward a P-row
// gather; the MFMA block product below forms their gradient rows in the same pass over P as page 0's.
constexpr int kHotPages = 8;
// pages the gradient-side block product carries through ONE pass over P (its accumulators: pages x Kp/16 x 4 VGPRs: 64 at
// 8 pages x Kp = 32 and at 4 pages x Kp = 64); a dataset with more pages than that takes several passes
constexpr int hot_pages_max(int Kp) { return Kp <= 32 ? 8 : (Kp <= 64 ? 4 : 1); }
extern std::atomic<int> g_tune[kTuneCount];     // process-wide DEFAULTS (fmhip_tune; atomic: any thread may set or read one); a model may override a key (fmhip_model_tune)
inline int tune_default(int key) { return g_tune[key].load(std::memory_order_relaxed); }

// padded factor count: 4 * LPN * J with LPN = lanes per row-slot (<= 16), J float4 per lane
int padded_factors(int k);
// grid of k_forward for a batch (also the number of per-block statistic partials it writes)
constexpr int kMaxFwdBlocks = 16384;
int forward_blocks(int Kp, int64_t n_rows);
int forward_blocks_lds(int64_t n_rows);           // grid of the LDS V-tile forward
int forward_blocks_wt(int Kp, int64_t n_rows, int occ_cap = 0);    // grid of the w-tile forward

// kFwdPartA / kFwdPartB: the training forward in two passes over a row's entries [row_ptr, row_split) and [row_split, row_end)
// (the pipelined data-parallel schedule, fmhip_comm.hip: the rows' entries are partitioned at the top cut of the plan; pass A —
// every feature below the cut — runs while the coldest slice of the previous step is still on the wire, pass B adds the cold
// features' terms once their rows are final and finishes the row).  A leaves the row's raw q in its P row and {sum_f s_f, linear
// term} in part_sl; B starts from them.  Kp <= 64 only.
enum FwdMode { kFwdTrain = 0, kFwdResidual = 1, kFwdQ = 2, kFwdPartA = 3, kFwdPartB = 4 };

struct FwdArgs {
    const int64_t *row_ptr;  // global CSR offsets (device), indexed row0 + r
    const int32_t *col;
    const float *val;
    const float *y;
    const float *V;   // [(n+1)][Kp]
    uint32_t v_bytes; // size of V in bytes, or 0 if it does not fit a 32-bit buffer descriptor
    const float *w;   // [n+1]
    const float *w0;  // [1]
    int64_t row0;
    int64_t nz0;           // row_ptr[row0]: the batch's first entry (k_forward / k_forward_wt walk 32-bit positions relative to it)
    const int64_t *row_split;  // kFwdPartA / kFwdPartB: where a row's entries of features >= the cut begin (global offsets, indexed row0 + r)
    float *part_sl;            // kFwdPartA writes, kFwdPartB reads: [rows][2] = {sum over factors of s_f, linear term} of pass A
    int32_t hot_in_b;          // two-pass forward: 1 = the dense hot block's prologue runs in pass B, not in pass A (a hot feature's id is at or above the cut)
    const int32_t *order;  // [n_rows] batch-local row ids, longest row first (NULL = 0, 1, 2, ..)
    int32_t n_rows;
    float *P;     // train: [rows][Kp] = e*q ; q-mode: [rows][Kp] = q
    float *e;     // [rows] e = yhat - y   (residual / train)
    float *yhat;  // optional [rows]
    double *bsum; // optional [forward_blocks][4] per-block {sum e, sum e^2, nonfinite, 0}
    int32_t tile_rows; // LDS V-tile: rows of V (feature ids < tile_rows) staged in LDS; 0 = off
    int32_t wt_rows;   // LDS w-tile: linear weights of feature ids < wt_rows staged in LDS
    int32_t pack_k;    // >= 0: packed rows — slot pack_k of a V row is w_i, of a P row is e (k < Kp); -1: off
    // dense hot block: xhot[r][h] = value of feature hot_ids[h] in row r (0 = absent), h < kHotT
    int32_t hot_T;           // 0 = none
    const float *xhot;       // [n_rows][kHotT], this batch's slice
    const int32_t *hot_ids;  // [kHotT], -1 = unused slot
    // lazy weight decay: the stored tables hold U with V = sv * U and w = sw * (stored w) (both 1 unless
    // rows-only updates with decay are pending, fm_apply.hip); the row epilogue applies them
    float sv, sw;
    // host-side launch choices (the model's tuning keys 0 and 6; the kernels never read them)
    int32_t variant;   // 60 = LDS w-tile kernel, 20 = LDS V-tile kernel, 0 = plain
    int32_t occ_cap;   // cap on the w-tile kernel's resident workgroups per CU (0 = all that fit)
};

// Fused update (single-GPU step, nothing to exchange): a finished gradient row is applied to its parameter row on
// the spot — the rows-only form of fm_apply.hip, same operations on the same values — instead of being stored to
// the packed gradient, read back, and zeroed by a separate launch.  V == NULL: store the gradient.
struct FusedUpd {
    float *V, *w;                 // parameter tables (stored scale sv: V = sv * stored)
    float sv, eta_v, eta_w, invb; // eta_* = eta / (scale after this step); invb = 1 / rows of the batch
};

// dense hot block, gradient side: G rows of the hot features = xhot^T . P (plus their two scalar sums)
struct HotArgs {
    const float *P;          // [n_rows][Kp] = e*q (slot pack_k = e for packed rows)
    const float *e;          // [n_rows]
    const float *xhot;       // page p: [n_rows][kHotT] at xhot + p * page_stride (this batch's slice)
    int64_t page_stride;     // floats between the pages of xhot
    const int32_t *hot_ids;  // [pages * kHotT], -1 = unused slot
    float *part;             // [hot_blocks][pages * kHotT][Kp + kPartPad] per-workgroup partial sums
    float *GV, *Gw, *Gb;
    int32_t n_rows;
    int32_t pack_k;
    int32_t nblk;            // hot_blocks(Kp, n_rows)
    int32_t pages;           // 1 .. hot_pages_max(Kp)
    FusedUpd upd;
};
struct ApplyArgs {
    float *V, *w, *w0;
    float *GV, *Gw, *Gb;
    const float *scal;  // {sum e, sum e^2, rows, ...}
    const float *rows;  // the batch's (global) row count |B|: scal + 2 unless it was exchanged on its own
    int64_t n1;         // n+1 (rows of V actually used)
    int64_t row_lo, row_hi;   // dense pass: the feature rows [row_lo, row_hi) (whole model: 0, n1)
    int32_t do_w0;      // this launch also steps w0 (once per step)
    float invb_val;     // use_invb_val: 1/|B| given by the host (the statistics of the step are still being summed)
    int32_t use_invb_val;
    int32_t pack_k;     // >= 0: packed rows — slot pack_k of a V row is the linear weight
    float eta, reg0, regw, regv;
    // scale of the stored tables on entry (V = sv_in * stored, w = sw_in * stored; fm_apply.hip).  The dense
    // pass leaves them at scale 1; the rows-only pass leaves them at sv_in * (1 - eta*regv) (resp. w) and
    // uses eta_v = eta / that (eta_w likewise) as its step on the stored values
    float sv_in, sw_in, eta_v, eta_w;
    // rows-only variant (feat != NULL): just the listed distinct features and the hot block's ids
    int32_t rows_only;       // 1: update only the rows in feat[] and hot_ids[] (k_apply_rows; decay rides in the tables' scale) — stated, not
                             // inferred from feat: a batch whose features all sit in the dense hot block has n_feat = 0 and may have feat = NULL
    const int32_t *feat;     // [n_feat] distinct feature ids of the batch
    int32_t n_feat;
    const int32_t *hot_ids;  // [n_hot], -1 = unused
    int32_t n_hot;
    int32_t g_compact;       // rows-only variant: GV / Gw / Gb are COMPACT arrays indexed by the position in `feat`, not by feature id
    // k_apply_shard (sharded update of one feature interval): V rows [row_lo, row_hi) = this rank's share are updated,
    // the linear weights of [w_lo, w_hi) = the whole interval are stepped, the G_V rows of [w_lo, z_hi) outside the share zeroed
    int64_t w_lo, w_hi, z_hi;
};

struct BwdArgs {
    const uint32_t *crow;      // batch CSC: bit31 = first entry of its column, low bits = batch-local row
    const float *cval;
    const int32_t *range_seg;  // [n_ranges] compressed column index holding entry rho*kRangeLen
    const int32_t *cfeat;      // [n_cols] feature id of compressed column
    const int32_t *cdst;       // [n_cols] >= 0: G row (= feature id) | < 0: piece row -1-dst (feature has several pieces)
    const int32_t *cptr;       // [n_cols+1]
    const int32_t *split_seg;  // [n_split] cut columns spanning > 8 ranges (a wave each in k_fixup)
    const int32_t *split_short; // [n_split_short] cut columns spanning <= 8 ranges (a slot each)
    int32_t nnz;
    int32_t n_ranges;
    int32_t rho_lo, rho_hi;    // ranges this launch walks (whole batch: 0, n_ranges)
    int32_t xcd_chunk;         // > 0: XCD-aware workgroup placement (set by the launcher)
    // Band-affine placement (xlist != NULL): workgroup b — dispatched round-robin, so on XCD b % 8 — walks the ranges at
    // positions (b / 8) * SLOTS + slot of XCD b % 8's list, which is the concatenation of up to kXSegs runs of xlist
    // (xseg_off / xseg_len: one run per row band the XCD owns, then its share of the other ranges; a feature-interval
    // launch passes the sub-runs that fall into its range window).  A range that lies inside one long column covers a
    // narrow band of rows (a column's entries ascend by row): the plan gives XCD x the ranges of ITS row bands first, so
    // their P rows — 2 MB per band — stay in that XCD's L2 instead of coming from the Infinity Cache, the latency that
    // bounds the gather rate.  Wave sums are off in such a launch (no_wave_sum also tells k_fixup).
    const int32_t *xlist;
    int32_t xseg_off[kXcds][kXSegs], xseg_len[kXcds][kXSegs];
    int32_t no_wave_sum;
    int32_t n_split;
    int32_t n_split_short;
    const float *P;            // [rows][Kp]
    uint32_t p_bytes;          // size of P in bytes, or 0 if it does not fit a 32-bit buffer descriptor
    const float *e;            // [rows]
    float *GV;                 // [n1p][Kp]
    float *Gw;                 // [n1p]
    float *Gb;                 // [n1p]
    float *part;               // [n_ranges][2][Kp + kPartPad]
    float *pieces;             // [n_pieces][Kp + kPartPad] column pieces of multi-piece features (row-blocked streams)
    const int32_t *mp_feat;    // [n_mp] features with several pieces
    const int32_t *mp_ptr;     // [n_mp + 1] their piece-row intervals
    int32_t n_mp;
    int32_t pack_k;            // >= 0: packed rows (see FwdArgs)
    // optional: k_fixup's extra last block also sums the forward's per-block statistics
    const double *red_bsum;
    int32_t red_nblocks, red_rows;
    float *red_scal;
    double *red_acc;
    // merged finish (single-GPU dense step): the fixup launch also performs the parameter update — its own rows
    // (cut columns, hot block) straight from registers, every other row in fin_blocks extra workgroups that run
    // beside the latency-bound fixups; fin_own = bitmap of the features the fixup part owns
    ApplyArgs fin;
    int32_t fin_blocks;
    const uint32_t *fin_own;
    int32_t fin_own_bits;
    float *red_w0;             // fused update: the statistics block also steps w0 (NULL = leave it to k_apply)
    float red_eta, red_reg0;
    FusedUpd upd;
    // dense hot block (hot_blocks > 0): the first hot_blocks workgroups of the backward launch form its
    // partial sums while the others walk the sparse stream; pages * kHotT extra workgroups of k_fixup finish it
    HotArgs hot;
    int32_t hot_blocks;
    int32_t hot_first;         // block index at which the hot workgroups start (0 = the front of the launch; set by the launcher)
    int32_t pipelined;         // host-side launch choice (the model's tuning key 1): 1 = k_backward_p where it applies
};


int hot_blocks(int Kp, int64_t n_rows);
// bit mask of the result-changing timing ablations (FMHIP_EXP_*) compiled into the kernels; 0 in every shipped build
int forward_ablations();
int backward_ablations();

// n_partials (optional): the number of per-block statistic partials the launch writes to a.bsum
hipError_t launch_forward(int Kp, FwdMode mode, const FwdArgs &a, hipStream_t s, int *n_partials = nullptr);
// V[i][f] = mean + stdev * N(0,1) for f < k (a hash of (seed, i, f) through Box-Muller), padding and w = 0
hipError_t launch_init_normal(int Kp, float *V, float *w, float *w0, int64_t n1, int64_t n1p, int32_t k, uint64_t seed, float mean,
                              float stdev, hipStream_t s);
// out_v[j*Kp + f] = V[ids[j]][f], out_w[j] = w[ids[j]] (raw stored values; the caller applies scales / packed slots)
hipError_t launch_gather_rows(int Kp, const float *V, const float *w, const int32_t *ids, int64_t n, float *out_v, float *out_w,
                              hipStream_t s);
// dense V *= sv, w *= sw (packed rows: the w slot of a V row by sw): brings lazily decayed tables back to scale 1
hipError_t launch_rescale(int Kp, float *V, float *w, int64_t n1, int32_t pack_k, float sv, float sw, hipStream_t s);
hipError_t launch_backward(int Kp, const BwdArgs &a, hipStream_t s);
hipError_t launch_fixup(int Kp, const BwdArgs &a, hipStream_t s);
hipError_t launch_fixup2(int Kp, const BwdArgs &a, hipStream_t s);   // sums the pieces of multi-piece features
hipError_t launch_apply(int Kp, const ApplyArgs &a, hipStream_t s);
hipError_t launch_apply_shard(int Kp, const ApplyArgs &a, hipStream_t s);        // see ApplyArgs::w_lo
// scal[0..3] = {sum e, sum e^2, n_rows, nonfinite} (optional); acc (optional, 4 doubles) += the same
hipError_t launch_reduce_blocks(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal, double *acc,
                                hipStream_t s);

}  // namespace fmhip
