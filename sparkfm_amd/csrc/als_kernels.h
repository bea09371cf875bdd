// als_kernels.h — launcher of the fp64 ALS epoch (internal to libfmhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmhip {

struct AlsArgs {
    int32_t k;
    int64_t num_attribute;   // loop bound of the sweeps (arrays have num_attribute + 1 slots)
    int64_t n_rows;
    int64_t nnz;
    // rows (CSR, stored order) — fp64 values
    const int64_t *row_ptr;
    const int32_t *col;
    const double *val;
    const double *y;
    // the same rows with every row's entries sorted by feature id (stable): the per-row order in which the
    // reference's transposed q pass (S/fm/lib/ALS.scala:146-150) adds a row's terms
    const int32_t *scol;
    const double *sval;
    // transpose (compressed columns of the single batch) — fp64 values
    int32_t n_cols;
    const int32_t *cfeat;
    const int32_t *cptr;
    const uint32_t *crow;
    const double *cval;
    // fp64 parameters, reference layout v[f + i*k]
    double *w0, *w, *v;
    double reg0, regw, regv;
    // workspace: e[n_rows]; q[n_rows * k] (every factor's q comes from an up-front pass over the feature-sorted rows);
    // part[2 * kAlsMaxParts + 2]: per-workgroup partial sums of a long column's chip-wide step
    double *e, *q, *part;
    // level schedule (optional): the compressed columns sorted by (level, id) — columns of one level share no row and
    // their steps commute exactly (fmhip_dataset::als_lev_cols)
    const int32_t *lev_cols;
};

constexpr int kAlsMaxParts = 512;       // workgroups of a chip-wide column step
// Columns of at least this many entries take the chip-wide two-launch step, shorter ones the one-workgroup walk
// (FMHIP_ALS_LONG in the environment overrides it: a test / measurement knob)
constexpr int kAlsLongColumn = 8192;

// h_cfeat / h_cptr: HOST copies of cfeat / cptr (the sweep's launch plan follows the column lengths)
// h_lev_ptr / n_levels: the level schedule's offsets into a.lev_cols (NULL / 0: none); h_lev_cols: host copy of a.lev_cols.  Taken when the levels hold at least
// 16 columns on average (FMHIP_ALS_LEVELS=0 / 1 in the environment: never / always — a test and measurement knob).
hipError_t launch_als_epoch(const AlsArgs &a, const int32_t *h_cfeat, const int32_t *h_cptr, const int32_t *h_lev_ptr,
                            const int32_t *h_lev_cols, int n_levels, hipStream_t s);

}  // namespace fmhip
