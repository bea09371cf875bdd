// csc_build.hip — per-batch row->column transpose on the GPU (SURVEY §8(f) "next #2").
//
// Replaces the reference's shuffle-based transposeRDD (S/DataSet.scala:31-38), which then collects
// the whole transposed dataset to the driver (S/fm/lib/ALS.scala:34).  The entries of a batch are
// keyed by feature id and sorted with a STABLE LSD radix sort (rocPRIM device_radix_sort — a setup
// step, not the hot path), so inside a column the rows keep their CSR (ascending) order; the payload
// is the entry's original index, from which row id and value(s) are gathered afterwards.
#include "csc_build.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace fmhip {
namespace {

__global__ __launch_bounds__(256) void k_expand(const int64_t *row_ptr, const int32_t *col, int64_t row0, int64_t rows,
                                                int64_t nnz0, int32_t rb_rows, int key_bits, int32_t *keys, uint32_t *idx,
                                                int32_t *rowid, const uint32_t *drop, int32_t drop_key) {
    // one 8-lane group per row: coalesced 32-B pieces of the row's entries
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    if (r >= rows) return;
    const int64_t p0 = row_ptr[row0 + r], p1 = row_ptr[row0 + r + 1];
    for (int64_t p = p0 + l; p < p1; p += 8) {
        const int64_t o = p - nnz0;
        // sort key: (row block, feature id) — row blocks keep a block's P rows L2-resident in the backward
        const int32_t c = col[p];
        const bool out = drop && (drop[c >> 5] >> (c & 31) & 1u);
        keys[o] = out ? drop_key : (c | (int32_t)((r / rb_rows) << key_bits));
        idx[o] = (uint32_t)o;
        rowid[o] = (int32_t)r;
    }
}

__global__ __launch_bounds__(256) void k_unpack(const int32_t *keys, const uint32_t *idx, const int32_t *rowid,
                                                const float *val, const double *val64, int64_t nnz0, int32_t nnz,
                                                uint32_t *crow, float *cval, double *cval64, uint8_t *flags) {
    const int32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint32_t o = idx[p];
    const bool first = p == 0 || keys[p] != keys[p - 1];
    crow[nnz0 + p] = (uint32_t)rowid[o] | (first ? 0x80000000u : 0u);
    cval[nnz0 + p] = val[nnz0 + o];
    if (cval64) cval64[nnz0 + p] = val64[nnz0 + o];
    flags[p] = first ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_gather_feats(const int32_t *keys, const int32_t *starts, const int32_t *count,
                                                      int32_t feat_mask, int32_t *feats) {
    const int32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s < *count) feats[s] = keys[starts[s]] & feat_mask;
}

}  // namespace

hipError_t csc_scratch_bytes(size_t max_nnz, int key_bits, size_t *bytes) {
    size_t a = 0, b = 0;
    int32_t *k = nullptr;
    uint32_t *v = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, max_nnz, 0, (unsigned)key_bits, (hipStream_t)0);
    if (e != hipSuccess) return e;
    uint8_t *f = nullptr;
    int32_t *o = nullptr;
    e = rocprim::select(nullptr, b, rocprim::counting_iterator<int32_t>(0), f, o, o, max_nnz, (hipStream_t)0);
    if (e != hipSuccess) return e;
    *bytes = a > b ? a : b;
    return hipSuccess;
}

hipError_t csc_build_batch(hipStream_t s, const CscScratch &sc, const int64_t *row_ptr, const int32_t *col,
                           const float *val, const double *val64, int64_t row0, int64_t rows, int64_t nnz0,
                           int32_t nnz, int key_bits, int32_t rb_rows, int rb_bits, uint32_t *crow, float *cval,
                           double *cval64, const uint32_t *drop, int32_t drop_key) {
    hipError_t e = hipMemsetAsync(sc.count, 0, sizeof(int32_t), s);
    if (e != hipSuccess || nnz == 0) return e;
    {
        const int64_t threads = rows * 8;
        hipLaunchKernelGGL(k_expand, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, row_ptr, col, row0, rows, nnz0,
                           rb_rows, key_bits, sc.keys_a, sc.idx_a, sc.rowid, drop, drop_key);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    size_t tb = sc.tmp_bytes;
    e = rocprim::radix_sort_pairs(sc.tmp, tb, sc.keys_a, sc.keys_b, sc.idx_a, sc.idx_b, (size_t)nnz, 0, (unsigned)(key_bits + rb_bits), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, sc.keys_b, sc.idx_b, sc.rowid, val, val64,
                       nnz0, nnz, crow, cval, cval64, sc.flags);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    tb = sc.tmp_bytes;
    e = rocprim::select(sc.tmp, tb, rocprim::counting_iterator<int32_t>(0), sc.flags, sc.starts, sc.count, (size_t)nnz, s);
    if (e != hipSuccess) return e;
    // at most min(nnz, max_cols) columns; the launch covers that bound and reads the count on device
    hipLaunchKernelGGL(k_gather_feats, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, sc.keys_b, sc.starts, sc.count,
                       (int32_t)(((int64_t)1 << key_bits) - 1), sc.feats);
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void k_range_rows(const uint32_t *crow, int32_t nnz, int32_t range_len, int32_t n_ranges, int32_t *first,
                                                    int32_t *last) {
    const int32_t rho = (int32_t)(blockIdx.x * 256 + threadIdx.x);
    if (rho >= n_ranges) return;
    const int64_t beg = (int64_t)rho * range_len;
    const int64_t end = beg + range_len < nnz ? beg + range_len : nnz;
    first[rho] = (int32_t)(crow[beg] & 0x7fffffffu);
    last[rho] = (int32_t)(crow[end - 1] & 0x7fffffffu);
}
}  // namespace

hipError_t csc_range_rows(hipStream_t s, const uint32_t *crow, int32_t nnz, int32_t range_len, int32_t n_ranges, int32_t *first,
                          int32_t *last) {
    if (n_ranges < 1) return hipSuccess;
    hipLaunchKernelGGL(k_range_rows, dim3((unsigned)((n_ranges + 255) / 256)), dim3(256), 0, s, crow, nnz, range_len, n_ranges, first, last);
    return hipGetLastError();
}

}  // namespace fmhip
