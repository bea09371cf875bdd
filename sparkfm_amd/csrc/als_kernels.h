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
    // workspace: e[n_rows]; q[n_rows * k] (the LDS sweep takes every factor's q from an up-front pass; the
    // fallback sweep uses the first n_rows)
    double *e, *q;
};

hipError_t launch_als_epoch(const AlsArgs &a, hipStream_t s);

}  // namespace fmhip
