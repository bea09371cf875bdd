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
};

constexpr int kAlsMaxParts = 512;       // workgroups of a chip-wide column step
// Columns of at least this many entries take the chip-wide two-launch step, shorter ones the one-workgroup walk
// (FMHIP_ALS_LONG in the environment overrides it: a test / measurement knob)
constexpr int kAlsLongColumn = 8192;

// h_cfeat / h_cptr: HOST copies of cfeat / cptr (the sweep's launch plan follows the column lengths)
hipError_t launch_als_epoch(const AlsArgs &a, const int32_t *h_cfeat, const int32_t *h_cptr, hipStream_t s);

}  // namespace fmhip
