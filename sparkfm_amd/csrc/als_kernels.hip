// als_kernels.hip — SparkFM's own learner, ALS / coordinate descent (S/fm/lib/ALS.scala:15-75,
// 152-198), on the GPU in fp64.
//
// The algorithm is Gauss-Seidel over the features: every closed-form update sees the residuals
// left by the previous one, so features are visited strictly in order by ONE persistent
// workgroup; the parallelism is inside a column (its entries are spread over the 1024 threads,
// sums are tree-reduced in a fixed order).  All arithmetic is fp64 without fma contraction — the
// reference runs on the JVM, which never fuses — so an epoch tracks the fp64 oracle to ~1e-13.
// This is a fidelity path (the reference's `fit`), not a throughput path: ~3 barriers per
// (feature, factor).
#include "als_kernels.h"

namespace fmhip {
namespace {


// S/fm/lib/ALS.scala:190-192
__device__ __forceinline__ bool is_updatable(double nv, double ov) { return !isnan(nv) && !isinf(nv) && nv != ov; }

// S/fm/lib/ALS.scala:167-176
__device__ __forceinline__ double compute_theta(double theta, double reg, double sum_e_h, double sum_h_sqr) {
#pragma clang fp contract(off)
    const double theta_new = -(sum_e_h - theta * sum_h_sqr) / (reg + sum_h_sqr);
    return is_updatable(theta_new, theta) ? theta_new : theta;
}

// ALS.precomputeTermE (S/fm/lib/ALS.scala:142-144) with FMModel.predict's exact order of operations
// (S/fm/FMModel.scala:34-63): one thread per row, sequential sums in stored order.
__global__ __launch_bounds__(256) void k_als_residual(AlsArgs a) {
#pragma clang fp contract(off)
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.n_rows; r += (int64_t)gridDim.x * 256) {
        const int64_t p0 = a.row_ptr[r], p1 = a.row_ptr[r + 1];
        double result = 0.0;
        result += *a.w0;
        if (p1 > p0) {
            double lin = a.w[a.col[p0]] * a.val[p0];
            for (int64_t p = p0 + 1; p < p1; ++p) lin += a.w[a.col[p]] * a.val[p];
            result += lin;
            for (int f = 0; f < a.k; ++f) {
                double t = a.v[f + (int64_t)a.col[p0] * a.k] * a.val[p0];
                double sum_f = t, sum_sqr_f = t * t;
                for (int64_t p = p0 + 1; p < p1; ++p) {
                    t = a.v[f + (int64_t)a.col[p] * a.k] * a.val[p];
                    sum_f += t;
                    sum_sqr_f += t * t;
                }
                result += 0.5 * (sum_f * sum_f - sum_sqr_f);
            }
        }
        a.e[r] = result - a.y[r];
    }
}

// fixed-order block sum of two doubles; result valid in every thread
template <int kAlsBlock>
__device__ __forceinline__ void block_sum2(double &x, double &y, double (*sh)[kAlsBlock / 64]) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x += __shfl_xor(x, m, 64);
        y += __shfl_xor(y, m, 64);
    }
    const int wv = threadIdx.x >> 6;
    __syncthreads();   // sh may still be read from the previous call
    if ((threadIdx.x & 63) == 0) { sh[0][wv] = x; sh[1][wv] = y; }
    __syncthreads();
    double tx = 0.0, ty = 0.0;
#pragma unroll
    for (int i = 0; i < kAlsBlock / 64; ++i) { tx += sh[0][i]; ty += sh[1][i]; }
    x = tx;
    y = ty;
}

// One ALS.learn pass after the residuals: w0 (:19-28), the linear weights (:36-43), then for every
// factor the q term (:50,146-150) and the factor sweep (:52-68).
template <int kAlsBlock>   // threads of the single workgroup: 256 for short columns (cheaper barriers), 1024 for long ones
__global__ __launch_bounds__(kAlsBlock) void k_als_sweep(AlsArgs a) {
#pragma clang fp contract(off)
    __shared__ double sh[2][kAlsBlock / 64];
    const int tid = threadIdx.x;
    const int k = a.k;
    // ---- global bias: drawGlobalBias :152-154 = computeTheta(w0, reg0, sum e, size)
    {
        double se = 0.0, dummy = 0.0;
        for (int64_t r = tid; r < a.n_rows; r += kAlsBlock) se += a.e[r];
        block_sum2<kAlsBlock>(se, dummy, sh);
        const double w0 = *a.w0;
        const double w0n = compute_theta(w0, a.reg0, se, (double)a.n_rows);
        if (is_updatable(w0n, w0)) {
            const double d = w0n - w0;
            for (int64_t r = tid; r < a.n_rows; r += kAlsBlock) a.e[r] = a.e[r] + d;
        }
        __syncthreads();
        if (tid == 0) *a.w0 = w0n;
        __syncthreads();
    }
    // ---- linear weights: `0 until num_attribute` (quirk Q1: slot n is never trained)
    for (int s = 0; s < a.n_cols; ++s) {
        const int64_t i = a.cfeat[s];
        if (i >= a.num_attribute) continue;
        const int c0 = a.cptr[s], c1 = a.cptr[s + 1];
        double shs = 0.0, seh = 0.0;
        for (int p = c0 + tid; p < c1; p += kAlsBlock) {
            const double x = a.cval[p];
            shs += x * x;
            seh += a.e[a.crow[p] & 0x7fffffffu] * x;
        }
        block_sum2<kAlsBlock>(shs, seh, sh);
        const double th = a.w[i];
        const double thn = compute_theta(th, a.regw, seh, shs);
        if (is_updatable(thn, th)) {
            const double d = thn - th;
            for (int p = c0 + tid; p < c1; p += kAlsBlock) a.e[a.crow[p] & 0x7fffffffu] += a.cval[p] * d;
        }
        __syncthreads();
        if (tid == 0) a.w[i] = thn;
        __syncthreads();
    }
    // ---- factors
    for (int f = 0; f < k; ++f) {
        for (int64_t r = tid; r < a.n_rows; r += kAlsBlock) a.q[r] = 0.0;
        __syncthreads();
        for (int s = 0; s < a.n_cols; ++s) {                    // precomputeTermQ: every slot, ascending feature id
            const double vfi = a.v[f + (int64_t)a.cfeat[s] * k];
            const int c0 = a.cptr[s], c1 = a.cptr[s + 1];
            for (int p = c0 + tid; p < c1; p += kAlsBlock) a.q[a.crow[p] & 0x7fffffffu] += vfi * a.cval[p];
            __syncthreads();
        }
        for (int s = 0; s < a.n_cols; ++s) {
            const int64_t i = a.cfeat[s];
            if (i >= a.num_attribute) continue;
            const int c0 = a.cptr[s], c1 = a.cptr[s + 1];
            const double vfi = a.v[f + i * k];
            double shs = 0.0, seh = 0.0;
            for (int p = c0 + tid; p < c1; p += kAlsBlock) {
                const uint32_t r = a.crow[p] & 0x7fffffffu;
                const double x = a.cval[p];
                const double h = x * a.q[r] - x * x * vfi;        // :56-58
                shs += h * h;
                seh += a.e[r] * h;
            }
            block_sum2<kAlsBlock>(shs, seh, sh);
            const double vn = compute_theta(vfi, a.regv, seh, shs);
            const double d = vn - vfi;
            const bool upd = is_updatable(vn, vfi);
            for (int p = c0 + tid; p < c1; p += kAlsBlock) {
                const uint32_t r = a.crow[p] & 0x7fffffffu;
                const double x = a.cval[p];
                if (upd) {
                    const double h = x * a.q[r] - x * x * vfi;
                    a.e[r] += h * d;                              // updateError :194-198
                }
                a.q[r] += x * d;                                  // :60-62
            }
            __syncthreads();
            if (tid == 0) a.v[f + i * k] = vn;                    // :64
            __syncthreads();
        }
    }
}

}  // namespace

hipError_t launch_als_epoch(const AlsArgs &a, hipStream_t s) {
    int64_t blocks = (a.n_rows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_als_residual, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // mean column length decides the workgroup size (a barrier over 4 waves is ~3x cheaper than over 16)
    const int64_t nnz = a.n_cols > 0 ? a.nnz : 0;
    if (a.n_cols > 0 && nnz / a.n_cols < 512) hipLaunchKernelGGL(k_als_sweep<256>, dim3(1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_als_sweep<1024>, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

}  // namespace fmhip
