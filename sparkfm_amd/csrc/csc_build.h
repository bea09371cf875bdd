// csc_build.h — device-side row->column transpose of one mini-batch (internal to libfmhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmhip {

// scratch sized for the largest batch; owned by the caller for the duration of the build
struct CscScratch {
    int32_t *keys_a = nullptr, *keys_b = nullptr;   // [max_nnz] feature id per entry (unsorted / sorted)
    uint32_t *idx_a = nullptr, *idx_b = nullptr;    // [max_nnz] batch-local entry index (payload of the sort)
    int32_t *rowid = nullptr;                       // [max_nnz] batch-local row of every CSR entry
    uint8_t *flags = nullptr;                       // [max_nnz] 1 = first entry of its column
    int32_t *starts = nullptr;                      // [max_cols + 1] offsets of the column starts
    int32_t *feats = nullptr;                       // [max_cols] feature id of every column
    int32_t *count = nullptr;                       // [1] number of columns
    void *tmp = nullptr;                            // rocPRIM temporary storage
    size_t tmp_bytes = 0;
};

// temporary-storage requirement of the sort / select for a batch of `max_nnz` entries
hipError_t csc_scratch_bytes(size_t max_nnz, int key_bits, size_t *bytes);

// Transposes the batch whose CSR entries are [nnz0, nnz0 + nnz) of col/val (rows row0 .. row0+rows):
//   crow[nnz0 + p] = batch-local row | (first entry of its column) << 31, cval[nnz0 + p] = value,
// sorted by (row block = local row / rb_rows, feature id), rows ascending inside a column piece (stable sort of the CSR order) — the same
// result as SparkFM's zipWithIndex + flatMap + groupByKey (S/DataSet.scala:31-38) with the
// unspecified groupByKey order fixed to ascending rows.  Leaves the column starts / feature ids /
// column count in the scratch (device) for the caller to read back.
// `drop` (nullable): bitmap over the feature ids whose entries are to stay OUT of the transpose (the gradient-side pages
// of the dense hot block): they are keyed `drop_key` (an id above every real one, inside key_bits; not with row blocks)
// and so form one last pseudo-column that the caller cuts off.
hipError_t csc_build_batch(hipStream_t s, const CscScratch &sc, const int64_t *row_ptr, const int32_t *col,
                           const float *val, const double *val64, int64_t row0, int64_t rows, int64_t nnz0,
                           int32_t nnz, int key_bits, int32_t rb_rows, int rb_bits, uint32_t *crow, float *cval,
                           double *cval64, const uint32_t *drop = nullptr, int32_t drop_key = 0);

// first[rho], last[rho] = the batch-local rows of the first and the last entry of range rho (kRangeLen entries each) of
// the transposed stream crow[0 .. nnz): what the band-affine placement of the backward's ranges is planned from
hipError_t csc_range_rows(hipStream_t s, const uint32_t *crow, int32_t nnz, int32_t range_len, int32_t n_ranges, int32_t *first,
                          int32_t *last);

}  // namespace fmhip
