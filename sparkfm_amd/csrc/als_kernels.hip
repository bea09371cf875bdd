// als_kernels.hip — SparkFM's own learner, ALS / coordinate descent (S/fm/lib/ALS.scala:15-75,
// 152-198), on the GPU in fp64.
//
// The algorithm is Gauss-Seidel over the features: every closed-form update sees the residuals
// left by the previous one, so features are visited strictly in order by ONE persistent
// workgroup; the parallelism is inside a column (its entries are spread over the 1024 threads,
// sums are tree-reduced in a fixed order).  All arithmetic is fp64 without fma contraction — the
// reference runs on the JVM, which never fuses — so an epoch tracks the fp64 oracle to ~1e-13.
// Three shapes of the sweep:
//  * k_als_sweep_lds — residuals e and the factor's q live in LDS (n_rows * 16 B <= 160,000 B: BASELINE config 1's
//    10,000 rows fill it exactly); ONE wave walks the columns (a column of ~100 entries is two wave-iterations: no
//    workgroup barrier per feature, wave sums by DPP row reductions), the next column's entries are prefetched
//    while the current one is reduced.
//  Larger datasets keep e and q in global memory and follow a launch plan made from the column lengths:
//  * k_als_cols_wg — a RUN of consecutive short columns by one workgroup (~3 barriers per column);
//  * k_als_col_sums + k_als_col_update — ONE long column (>= kAlsLongColumn entries) on the whole chip: every
//    workgroup sums its slice (sum h^2, sum e*h), the second launch adds the partials in a fixed tree (every workgroup
//    the same bits), forms theta* and updates its slice of e and q.  The dependency between consecutive columns is the
//    launch boundary — no device-side grid barrier to get wrong.  A column of 10^5 entries costs a CPU core ~0.2 ms per
//    step and this path two launches.
// Every factor's q comes from one up-front pass over the feature-sorted rows (k_als_q_all), the per-row order of the
// reference's transposed pass.
#include "als_kernels.h"

#include <cstdlib>

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

// drawGlobalBias :152-154 = computeTheta(w0, reg0, sum e, size) and `fm.w0 = w0` (:19-27) — once per epoch, one workgroup.
// The residuals are NOT shifted here: as Spark evaluates :23-31 the lazy `error` RDD is materialised after `fm.w0 = w0`, so
// its `fm.predict` already carries the new bias and the mapped term `w0 - fm.w0` is zero — the
// launcher simply runs k_als_residual again with the new bias.
template <int kAlsBlock>
__global__ __launch_bounds__(kAlsBlock) void k_als_w0(AlsArgs a) {
#pragma clang fp contract(off)
    __shared__ double sh[2][kAlsBlock / 64];
    const int tid = threadIdx.x;
    double se = 0.0, dummy = 0.0;
    for (int64_t r = tid; r < a.n_rows; r += kAlsBlock) se += a.e[r];
    block_sum2<kAlsBlock>(se, dummy, sh);
    const double w0 = *a.w0;
    const double w0n = compute_theta(w0, a.reg0, se, (double)a.n_rows);
    if (tid == 0) *a.w0 = w0n;
}

// The closed-form steps of the consecutive columns [s_lo, s_hi) by ONE workgroup, in order (S/fm/lib/ALS.scala:36-43
// linear weights, :52-68 factor f): the run of SHORT columns between two long ones.  e and q in global memory.
template <int kAlsBlock, bool FACTOR>   // threads: 256 for short columns (cheaper barriers), 1024 for longer ones
__global__ __launch_bounds__(kAlsBlock) void k_als_cols_wg(AlsArgs a, double *q, int s_lo, int s_hi, int f) {
#pragma clang fp contract(off)
    __shared__ double sh[2][kAlsBlock / 64];
    const int tid = threadIdx.x;
    const int k = a.k;
    const double reg = FACTOR ? a.regv : a.regw;
    for (int s = s_lo; s < s_hi; ++s) {
        const int64_t i = a.cfeat[s];
        if (i >= a.num_attribute) continue;                       // `0 until num_attribute` (quirk Q1: slot n is never trained)
        const int c0 = a.cptr[s], c1 = a.cptr[s + 1];
        double *par = FACTOR ? a.v + f + i * k : a.w + i;
        const double th = *par;
        double shs = 0.0, seh = 0.0;
        for (int p = c0 + tid; p < c1; p += kAlsBlock) {
            const uint32_t r = a.crow[p] & 0x7fffffffu;
            const double x = a.cval[p];
            const double h = FACTOR ? x * q[r] - x * x * th : x;      // :56-58 / :40
            shs += h * h;
            seh += a.e[r] * h;
        }
        block_sum2<kAlsBlock>(shs, seh, sh);
        const double tn = compute_theta(th, reg, seh, shs);
        const double d = tn - th;
        const bool upd = is_updatable(tn, th);
        for (int p = c0 + tid; p < c1; p += kAlsBlock) {
            const uint32_t r = a.crow[p] & 0x7fffffffu;
            const double x = a.cval[p];
            if (upd) {
                const double h = FACTOR ? x * q[r] - x * x * th : x;
                a.e[r] += h * d;                                  // updateError :194-198
            }
            if (FACTOR) q[r] += x * d;                            // :60-62
        }
        __syncthreads();
        if (tid == 0) *par = tn;                                  // :40 / :64
        __syncthreads();
    }
}

// ---- one LONG column on the whole chip: two launches -------------------------------------------------
constexpr int kColBlock = 256;

// launch 1: workgroup g sums its contiguous slice of the column -> part[2g], part[2g + 1]; part[2G] = theta
template <bool FACTOR>
__global__ __launch_bounds__(kColBlock) void k_als_col_sums(AlsArgs a, const double *q, int c0, int c1, int64_t i, int f) {
#pragma clang fp contract(off)
    __shared__ double sh[2][kColBlock / 64];
    const int tid = threadIdx.x, G = (int)gridDim.x;
    const int per = (c1 - c0 + G - 1) / G;
    const int lo = c0 + (int)blockIdx.x * per, hi = lo + per < c1 ? lo + per : c1;
    const double th = FACTOR ? a.v[f + i * a.k] : a.w[i];
    double shs = 0.0, seh = 0.0;
    for (int p = lo + tid; p < hi; p += kColBlock) {
        const uint32_t r = a.crow[p] & 0x7fffffffu;
        const double x = a.cval[p];
        const double h = FACTOR ? x * q[r] - x * x * th : x;
        shs += h * h;
        seh += a.e[r] * h;
    }
    block_sum2<kColBlock>(shs, seh, sh);
    if (tid == 0) {
        a.part[2 * blockIdx.x] = shs;
        a.part[2 * blockIdx.x + 1] = seh;
        if (blockIdx.x == 0) a.part[2 * G] = th;      // the update launch must not read a parameter another workgroup is writing
    }
}

// launch 2: every workgroup adds the G partials in the same fixed tree (the same bits everywhere), forms theta* and
// updates its slice of e (and q); workgroup 0 stores the parameter
template <bool FACTOR>
__global__ __launch_bounds__(kColBlock) void k_als_col_update(AlsArgs a, double *q, int c0, int c1, int64_t i, int f) {
#pragma clang fp contract(off)
    __shared__ double sh[2][kColBlock / 64];
    const int tid = threadIdx.x, G = (int)gridDim.x;
    double shs = 0.0, seh = 0.0;
    for (int g = tid; g < G; g += kColBlock) {       // G <= kAlsMaxParts = 2 * kColBlock: at most two terms per thread
        shs += a.part[2 * g];
        seh += a.part[2 * g + 1];
    }
    block_sum2<kColBlock>(shs, seh, sh);
    const double th = a.part[2 * G];
    const double tn = compute_theta(th, FACTOR ? a.regv : a.regw, seh, shs);
    const double d = tn - th;
    const bool upd = is_updatable(tn, th);
    const int per = (c1 - c0 + G - 1) / G;
    const int lo = c0 + (int)blockIdx.x * per, hi = lo + per < c1 ? lo + per : c1;
    for (int p = lo + tid; p < hi; p += kColBlock) {
        const uint32_t r = a.crow[p] & 0x7fffffffu;       // the rows of a column are distinct (als_dup is refused): no two writers
        const double x = a.cval[p];
        if (upd) {
            const double h = FACTOR ? x * q[r] - x * x * th : x;
            a.e[r] += h * d;
        }
        if (FACTOR) q[r] += x * d;
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (FACTOR) a.v[f + i * a.k] = tn;
        else a.w[i] = tn;
    }
}


// precomputeTermQ (S/fm/lib/ALS.scala:146-150) of EVERY factor before the sweeps, on the whole chip: q_f only
// reads v[f, :], which nothing modifies before factor f's own sweep.  One thread per (row, factor); a row's terms
// are added in ascending feature order (the feature-sorted copy of the rows) — the order of the reference's
// transposed pass.  qall[f * n_rows + r].
__global__ __launch_bounds__(256) void k_als_q_all(AlsArgs a, double *qall) {
#pragma clang fp contract(off)
    const int64_t total = a.n_rows * a.k;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / a.k;
        const int f = (int)(idx % a.k);
        double qv = 0.0;
        for (int64_t p = a.row_ptr[r]; p < a.row_ptr[r + 1]; ++p) qv += a.v[f + (int64_t)a.scol[p] * a.k] * a.sval[p];
        qall[(int64_t)f * a.n_rows + r] = qv;
    }
}

// ---- LDS-resident sweep -------------------------------------------------------------------------------
constexpr int kLdsSweepThreads = 1024;
constexpr int kLevelBlock = 256;            // level sweep: four waves = four columns per workgroup
constexpr size_t kAlsLdsBytes = 160000;     // e + q of at most 10,000 rows (the static reduction scratch fits beside it)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, identical in every lane; fixed order: quad butterfly, half-row and row mirrors on
// the vector ALU (DPP), then the four row sums through scalar registers
__device__ __forceinline__ double wave_sum(double v) {
#pragma clang fp contract(off)
    v += dpp_f64<0xB1>(v);     // quad_perm:[1,0,3,2]
    v += dpp_f64<0x4E>(v);     // quad_perm:[2,3,0,1]
    v += dpp_f64<0x141>(v);    // row_half_mirror
    v += dpp_f64<0x140>(v);    // row_mirror
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * i), __builtin_amdgcn_readlane(__double2loint(v), 16 * i));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// The first 128 entries of a column as one wave holds them: lane l has entries l and 64 + l (raw CSC words: bit 31 of
// a row word still carries the column-start flag).  Loaded a step ahead, at clamped addresses and without touching the
// values, so that the requests stay in flight across the step (a mask or a branch on a loaded value makes the
// compiler wait for it on the spot).
struct ColHead {
    uint32_t r0, r1;
    double x0, x1;
};

__device__ __forceinline__ ColHead load_head(const AlsArgs &a, int c0, int lane) {
    const int64_t last = a.nnz - 1;
    const int64_t p0 = (int64_t)c0 + lane < last ? (int64_t)c0 + lane : last;
    const int64_t p1 = (int64_t)c0 + 64 + lane < last ? (int64_t)c0 + 64 + lane : last;
    ColHead h;
    h.r0 = a.crow[p0];
    h.r1 = a.crow[p1];
    h.x0 = a.cval[p0];
    h.x1 = a.cval[p1];
    return h;
}

// One closed-form step on column [c0, c1) by ONE wave (S/fm/lib/ALS.scala:36-43 linear, :52-68 factor):
// h_r, the sums sum h^2 / sum e*h, theta*, then e += h*(theta* - theta) (and q += x*(theta* - theta)).
// The first two wave-iterations of the column (<= 128 entries: every column of config 1) stay in registers
// between the sums and the update — e and q are read from LDS once; `head` arrives prefetched and leaves holding
// the next column's.
template <bool FACTOR, bool NEXT = true>
__device__ __forceinline__ double column_step(const AlsArgs &a, double *e, double *q, int c0, int c1, double theta, double reg,
                                              ColHead &head, int lane) {
#pragma clang fp contract(off)
    const int n = c1 - c0;
    const ColHead next = NEXT ? load_head(a, c1, lane) : head;      // in flight while this column is reduced
    const bool v0 = lane < n, v1 = 64 + lane < n;
    const uint32_t ra = v0 ? head.r0 & 0x7fffffffu : 0u, rb = v1 ? head.r1 & 0x7fffffffu : 0u;
    const double xa = head.x0, xb = head.x1;
    const double ea = e[ra], eb = e[rb];
    const double qa = FACTOR ? q[ra] : 0.0, qb = FACTOR ? q[rb] : 0.0;
    double ha = FACTOR ? xa * qa - xa * xa * theta : xa;            // :56-58 / :40
    double hb = FACTOR ? xb * qb - xb * xb * theta : xb;
    ha = v0 ? ha : 0.0;
    hb = v1 ? hb : 0.0;
    double shs = 0.0, seh = 0.0;
    shs += ha * ha;
    seh += (v0 ? ea : 0.0) * ha;
    shs += hb * hb;
    seh += (v1 ? eb : 0.0) * hb;
    for (int p = c0 + 128 + lane; p < c1; p += 64) {               // long columns: re-read in the update pass
        const uint32_t r = a.crow[p] & 0x7fffffffu;
        const double x = a.cval[p];
        const double h = FACTOR ? x * q[r] - x * x * theta : x;
        shs += h * h;
        seh += e[r] * h;
    }
    shs = wave_sum(shs);
    seh = wave_sum(seh);
    const double tn = compute_theta(theta, reg, seh, shs);
    const double d = tn - theta;
    const bool upd = is_updatable(tn, theta);
    if (v0) {
        if (upd) e[ra] = ea + ha * d;                              // updateError :194-198
        if (FACTOR) q[ra] = qa + xa * d;                           // :60-62
    }
    if (v1) {
        if (upd) e[rb] = eb + hb * d;
        if (FACTOR) q[rb] = qb + xb * d;
    }
    for (int p = c0 + 128 + lane; p < c1; p += 64) {
        const uint32_t r = a.crow[p] & 0x7fffffffu;
        const double x = a.cval[p];
        if (upd) {
            const double h = FACTOR ? x * q[r] - x * x * theta : x;
            e[r] += h * d;
        }
        if (FACTOR) q[r] += x * d;
    }
    head = next;
    return tn;
}

// The column walk of one pass (linear weights, or factor f) by wave 0.  Everything a step needs from global memory
// is requested a step (the column's entries, theta) or two (its feature id and end offset) ahead, so the chain of a
// step is LDS + vector ALU only.  Valid because a pass touches every parameter exactly once.
template <bool FACTOR>
__device__ __forceinline__ void walk_columns(const AlsArgs &a, double *e, double *q, int f, int lane) {
    const int k = a.k, nc = a.n_cols;
    if (nc < 1 || a.nnz < 1) return;
    double *par = FACTOR ? a.v + f : a.w;                          // parameter of feature i: par[i * stride]
    const int64_t stride = FACTOR ? k : 1;
    const double reg = FACTOR ? a.regv : a.regw;
    int c0 = a.cptr[0], c1 = a.cptr[1];
    int64_t i_cur = a.cfeat[0], i_nx = a.cfeat[nc > 1 ? 1 : 0];
    int c_nx1 = a.cptr[nc > 1 ? 2 : 1];
    double th_cur = par[i_cur * stride];
    ColHead head = load_head(a, c0, lane);
    for (int s = 0; s < nc; ++s) {
        const double th_nx = par[i_nx * stride];                   // step s+1's parameter
        const int64_t i_nn = a.cfeat[s + 2 < nc ? s + 2 : nc - 1]; // step s+2's feature id and end offset (clamped, unconditional)
        const int c_nn1 = a.cptr[s + 3 < nc ? s + 3 : nc];
        if (i_cur < a.num_attribute) {                             // `0 until num_attribute`: slot n is never trained (quirk Q1)
            const double tn = column_step<FACTOR>(a, e, q, c0, c1, th_cur, reg, head, lane);
            if (lane == 0) par[i_cur * stride] = tn;               // :40 / :64
        } else {
            head = load_head(a, c1, lane);                         // skipped column: still hand the next column's head on
        }
        c0 = c1; c1 = c_nx1; c_nx1 = c_nn1;
        i_cur = i_nx; i_nx = i_nn;
        th_cur = th_nx;
    }
}

// ---- level sweep ----------------------------------------------------------------------------------------
// The columns of ONE level of the schedule side by side, a wave each: they share no row, so each step reads and writes
// residuals / q entries no other step of the launch touches, and all the columns it depends on (smaller ids sharing a
// row) sit in earlier levels = earlier launches.  The step itself is column_step — the operations and the summation order
// of the one-wave LDS walk, on global memory — so a level-scheduled pass leaves the bits the sequential walk leaves.
// One-hot fields (the reference's own MovieLens demo, S/driver.scala:73-113: a user field and an item field) are the
// case this is for: all columns of a field form one level.
template <bool FACTOR>
__global__ __launch_bounds__(kLevelBlock) void k_als_level(AlsArgs a, double *q, const int32_t *cols, int n, int f, int long_col) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int j = (int)blockIdx.x * (kLevelBlock / 64) + (int)(threadIdx.x >> 6);
    if (j >= n) return;
    const int s = cols[j];
    const int64_t i = a.cfeat[s];
    if (i >= a.num_attribute) return;                              // `0 until num_attribute` (quirk Q1)
    const int c0 = a.cptr[s], c1 = a.cptr[s + 1];
    if (c1 - c0 >= long_col) return;                               // a long column of the level: the chip-wide step behind this launch
    double *par = FACTOR ? a.v + f + i * a.k : a.w + i;
    ColHead head = load_head(a, c0, lane);
    const double tn = column_step<FACTOR, false>(a, a.e, q, c0, c1, *par, FACTOR ? a.regv : a.regw, head, lane);
    if (lane == 0) *par = tn;                                      // :40 / :64
}

__global__ __launch_bounds__(kLdsSweepThreads) void k_als_sweep_lds(AlsArgs a, const double *qall) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) double lds_eq[];
    double *e = lds_eq, *q = lds_eq + a.n_rows;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool walker = tid < 64;                                  // wave 0 walks the columns
    for (int64_t r = tid; r < a.n_rows; r += kLdsSweepThreads) e[r] = a.e[r];
    __syncthreads();
    // (the global bias step ran before this launch: k_als_w0, then the residuals again with the new bias)
    // ---- linear weights (:36-43)
    if (walker) walk_columns<false>(a, e, q, 0, lane);
    // ---- factors (:50-68): q of the factor comes in from the up-front pass, every thread copies its share
    for (int f = 0; f < a.k; ++f) {
        __syncthreads();
        for (int64_t r = tid; r < a.n_rows; r += kLdsSweepThreads) q[r] = qall[(int64_t)f * a.n_rows + r];
        __syncthreads();
        if (walker) walk_columns<true>(a, e, q, f, lane);
    }
}

}  // namespace

namespace {

// one chip-wide closed-form step of a long column (two launches; see k_als_col_sums)
template <bool FACTOR>
void long_column_step(const AlsArgs &a, double *q, int c0, int c1, int64_t i, int f, hipStream_t s) {
    int G = (c1 - c0 + 2047) / 2048;
    if (G > kAlsMaxParts) G = kAlsMaxParts;
    hipLaunchKernelGGL((k_als_col_sums<FACTOR>), dim3((unsigned)G), dim3(kColBlock), 0, s, a, q, c0, c1, i, f);
    hipLaunchKernelGGL((k_als_col_update<FACTOR>), dim3((unsigned)G), dim3(kColBlock), 0, s, a, q, c0, c1, i, f);
}

template <bool FACTOR>
hipError_t sweep_pass(const AlsArgs &a, const int32_t *h_cfeat, const int32_t *h_cptr, int f, int long_col, bool wide_wg, hipStream_t s) {
    double *q = a.q + (FACTOR ? (int64_t)f * a.n_rows : 0);
    int s0 = 0;
    while (s0 < a.n_cols) {
        const int len = h_cptr[s0 + 1] - h_cptr[s0];
        if (len >= long_col) {
            if (h_cfeat[s0] < a.num_attribute) long_column_step<FACTOR>(a, q, h_cptr[s0], h_cptr[s0 + 1], (int64_t)h_cfeat[s0], f, s);
            ++s0;
            continue;
        }
        int s1 = s0 + 1;
        while (s1 < a.n_cols && h_cptr[s1 + 1] - h_cptr[s1] < long_col) ++s1;
        if (wide_wg) hipLaunchKernelGGL((k_als_cols_wg<1024, FACTOR>), dim3(1), dim3(1024), 0, s, a, q, s0, s1, f);
        else hipLaunchKernelGGL((k_als_cols_wg<256, FACTOR>), dim3(1), dim3(256), 0, s, a, q, s0, s1, f);
        s0 = s1;
    }
    return hipGetLastError();
}

}  // namespace

namespace {
template <bool FACTOR>
hipError_t level_pass(const AlsArgs &a, const int32_t *h_cfeat, const int32_t *h_cptr, const int32_t *h_lev_ptr, const int32_t *h_lev_cols,
                      int n_levels, int f, int long_col, hipStream_t s) {
    double *q = a.q + (FACTOR ? (int64_t)f * a.n_rows : 0);
    for (int l = 0; l < n_levels; ++l) {
        const int n = h_lev_ptr[l + 1] - h_lev_ptr[l];
        if (n < 1) continue;
        hipLaunchKernelGGL((k_als_level<FACTOR>), dim3((unsigned)((n + kLevelBlock / 64 - 1) / (kLevelBlock / 64))), dim3(kLevelBlock), 0, s, a, q,
                           a.lev_cols + h_lev_ptr[l], n, f, long_col);
        // the level's long columns share no row with its other columns either: their chip-wide steps follow, in any order
        for (int j = h_lev_ptr[l]; j < h_lev_ptr[l + 1]; ++j) {
            const int c = h_lev_cols[j];
            if (h_cptr[c + 1] - h_cptr[c] >= long_col && h_cfeat[c] < a.num_attribute)
                long_column_step<FACTOR>(a, q, h_cptr[c], h_cptr[c + 1], (int64_t)h_cfeat[c], f, s);
        }
    }
    return hipGetLastError();
}
}  // namespace

hipError_t launch_als_epoch(const AlsArgs &a, const int32_t *h_cfeat, const int32_t *h_cptr, const int32_t *h_lev_ptr,
                            const int32_t *h_lev_cols, int n_levels, hipStream_t s) {
    int64_t blocks = (a.n_rows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    // precomputeTermE (:17) with the old bias feeds the bias step (:19-27); the residuals the sweeps start from are
    // evaluated after `fm.w0 = w0` (see k_als_w0)
    hipLaunchKernelGGL(k_als_residual, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_als_w0<1024>, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_als_residual, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int64_t qb = (a.n_rows * a.k + 255) / 256;
    if (qb > 4096) qb = 4096;
    hipLaunchKernelGGL(k_als_q_all, dim3((unsigned)(qb < 1 ? 1 : qb)), dim3(256), 0, s, a, a.q);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    int long_col = kAlsLongColumn;
    if (const char *ev = getenv("FMHIP_ALS_LONG")) long_col = atoi(ev) > 0 ? atoi(ev) : long_col;
    {
        // level schedule: a launch per level and pass, every column of the level on a wave of its own — pays when the
        // levels are wide (one-hot fields: two levels of thousands of columns); a dense conflict graph (config 1: every
        // column shares rows with most others) has about as many levels as columns and keeps the sequential walks below.
        // A level's columns long enough for the chip-wide step take it, behind the level's launch.
        const bool have = h_lev_ptr && h_lev_cols && a.lev_cols && n_levels > 0;
        bool levels = have && (int64_t)n_levels * 16 <= a.n_cols;
        if (const char *ev = getenv("FMHIP_ALS_LEVELS")) levels = have && atoi(ev) != 0;
        if (levels) {
            if ((e = level_pass<false>(a, h_cfeat, h_cptr, h_lev_ptr, h_lev_cols, n_levels, 0, long_col, s)) != hipSuccess) return e;
            for (int f = 0; f < a.k; ++f)
                if ((e = level_pass<true>(a, h_cfeat, h_cptr, h_lev_ptr, h_lev_cols, n_levels, f, long_col, s)) != hipSuccess) return e;
            return hipSuccess;
        }
    }
    if ((size_t)a.n_rows * 2 * sizeof(double) <= kAlsLdsBytes && !getenv("FMHIP_ALS_NO_LDS")) {
        // e and q fit the LDS of one CU: the one-wave column walk
        e = hipFuncSetAttribute((const void *)k_als_sweep_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAlsLdsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_als_sweep_lds, dim3(1), dim3(kLdsSweepThreads), (size_t)a.n_rows * 2 * sizeof(double), s, a, a.q);
        return hipGetLastError();
    }
    // e and q in global memory: one pass per parameter group following the column lengths — a run of short columns is one
    // launch of one workgroup, a long column two chip-wide launches
    // the short columns' mean length decides their workgroup size (a barrier over 4 waves is ~3x cheaper than over 16)
    int64_t short_nnz = 0, n_short = 0;
    for (int c = 0; c < a.n_cols; ++c) {
        const int len = h_cptr[c + 1] - h_cptr[c];
        if (len < long_col) { short_nnz += len; ++n_short; }
    }
    const bool wide_wg = n_short > 0 && short_nnz / n_short >= 512;
    if ((e = sweep_pass<false>(a, h_cfeat, h_cptr, 0, long_col, wide_wg, s)) != hipSuccess) return e;
    for (int f = 0; f < a.k; ++f)
        if ((e = sweep_pass<true>(a, h_cfeat, h_cptr, f, long_col, wide_wg, s)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace fmhip
