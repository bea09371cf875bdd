// fm_apply.hip — the parameter update of the FM step (dense, or the rows a batch touched), fused with
// zeroing the packed gradient.
#include "fm_device.h"

namespace fmhip {
namespace {

// (apply_piece / apply_w0 live in fm_device.h: k_fixup's merged finish uses them too)

template <int KP>
__global__ __launch_bounds__(kBlock) void k_apply(ApplyArgs a) {
    constexpr int LPR = KP / 4;  // lanes per feature row (<= 64, divides the wave)
    const float invb = apply_invb(a);
    const int64_t total = (a.row_hi - a.row_lo) * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock)
        apply_piece<KP, false>(a, a.row_lo + idx / LPR, (int)(idx % LPR), invb);
    if (a.do_w0) apply_w0(a, invb);
}

// The same update restricted to the rows a batch touched (its distinct features + the dense hot
// block's); every other row has a zero gradient and its decay rides in the tables' scale (see above).
// Matters when the model is far wider than a batch (Criteo-like widths: 2^25 rows of V, 8.6 GB,
// against ~2 M touched).
template <int KP>
__global__ __launch_bounds__(kBlock) void k_apply_rows(ApplyArgs a) {
    constexpr int LPR = KP / 4;
    const float invb = apply_invb(a);
    const int64_t total = ((int64_t)a.n_feat + a.n_hot) * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t j = idx / LPR;
        const int32_t i = j < a.n_feat ? a.feat[j] : a.hot_ids[j - a.n_feat];
        if (i >= 0) apply_piece<KP, true>(a, i, (int)(idx % LPR), invb, a.g_compact ? j : (int64_t)i);
    }
    if (a.do_w0) apply_w0(a, invb);
}

// Sharded update (fmhip_comm.hip, FMHIP_EXCHANGE_SHARDED): one launch per feature interval [w_lo, z_hi), queued on the
// collectives' stream between the interval's reduce-scatter and its all-gather.
//   * the V rows of THIS rank's share [row_lo, row_hi) get the dense update (apply_piece<KP, false>: the same operations
//     as the unsharded pass) and come back to every replica through the all-gather — one writer per row;
//   * the other shares' rows of G_V hold this rank's local (unsummed) gradient: zeroed for the next step;
//   * w (4 bytes per feature against 4*KP) is stepped by EVERY rank from the all-reduced G_w over [w_lo, w_hi) — the
//     operations of apply_piece's linear-weight branch, so every replica holds the same bits.
template <int KP>
__global__ __launch_bounds__(kBlock) void k_apply_shard(ApplyArgs a) {
    constexpr int LPR = KP / 4;
    const float invb = apply_invb(a);
    const int64_t total = (a.z_hi - a.w_lo) * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t i = a.w_lo + idx / LPR;
        const int c = (int)(idx % LPR);
        if (i >= a.row_lo && i < a.row_hi) {
            apply_piece<KP, false>(a, i, c, invb);
            continue;
        }
        reinterpret_cast<float4 *>(a.GV)[i * LPR + c] = f4zero();
        if (c == 0 && i < a.w_hi) {
            const float us = a.w[i], gi = a.Gw[i] * invb;
            const float wi = us * a.sw_in;
            a.w[i] = wi - a.eta * fmaf(a.regw, wi, gi);
            a.Gw[i] = 0.f;
            a.Gb[i] = 0.f;
        }
    }
    if (a.do_w0) apply_w0(a, invb);
}


// ------------------------------------------------------------------ init
// `new FMModel(n, k)` on the device (S/fm/FMModel.scala:17-22): v ~ N(mean, stdev), w = 0, w0 = 0.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(kBlock) void k_init_normal(float *V, float *w, float *w0, int64_t n1, int64_t n1p, int32_t k, int32_t kp,
                                                       uint64_t seed, float mean, float stdev) {
    const int64_t total = n1p * kp;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t i = idx / kp;
        const int f = (int)(idx % kp);
        float v = 0.f;
        if (i < n1 && f < k) {
            const uint64_t h = mix64(seed ^ mix64((uint64_t)i * (uint64_t)k + (uint64_t)f + 0x9E3779B97F4A7C15ull));
            const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);         // (0, 1)
            const float u2 = ((float)(uint32_t)((h >> 8) & 0xffffffu)) * (1.0f / 16777216.0f);   // [0, 1)
            v = mean + stdev * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
        }
        V[idx] = v;
        if (f == 0) w[i] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *w0 = 0.f;
}

__global__ __launch_bounds__(kBlock) void k_gather_rows(const float *V, const float *w, const int32_t *ids, int64_t n, int32_t kp,
                                                       float *out_v, float *out_w) {
    const int lpr = kp / 4;
    const int64_t total = n * lpr;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t j = idx / lpr;
        const int c = (int)(idx % lpr);
        const int64_t i = ids[j];
        reinterpret_cast<float4 *>(out_v)[idx] = reinterpret_cast<const float4 *>(V)[i * lpr + c];
        if (c == 0) out_w[j] = w[i];
    }
}

}  // namespace

hipError_t launch_gather_rows(int Kp, const float *V, const float *w, const int32_t *ids, int64_t n, float *out_v, float *out_w,
                              hipStream_t s) {
    int64_t blocks = (n * (Kp / 4) + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks), dim3(kBlock), 0, s, V, w, ids, n, Kp, out_v, out_w);
    return hipGetLastError();
}

hipError_t launch_init_normal(int Kp, float *V, float *w, float *w0, int64_t n1, int64_t n1p, int32_t k, uint64_t seed, float mean,
                              float stdev, hipStream_t s) {
    int64_t blocks = (n1p * Kp + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_init_normal, dim3((unsigned)blocks), dim3(kBlock), 0, s, V, w, w0, n1, n1p, k, Kp, seed, mean, stdev);
    return hipGetLastError();
}

hipError_t launch_apply_shard(int Kp, const ApplyArgs &a, hipStream_t s) {
    int64_t blocks = ((a.z_hi - a.w_lo) * (Kp / 4) + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    dim3 g((unsigned)blocks), b(kBlock);
    switch (Kp) {
        case 32: hipLaunchKernelGGL((k_apply_shard<32>), g, b, 0, s, a); break;
        case 64: hipLaunchKernelGGL((k_apply_shard<64>), g, b, 0, s, a); break;
        case 128: hipLaunchKernelGGL((k_apply_shard<128>), g, b, 0, s, a); break;
        case 256: hipLaunchKernelGGL((k_apply_shard<256>), g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_apply(int Kp, const ApplyArgs &a, hipStream_t s) {
    const bool rows_only = a.rows_only != 0;
    int64_t total = (rows_only ? (int64_t)a.n_feat + a.n_hot : a.row_hi - a.row_lo) * (Kp / 4);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    dim3 g((unsigned)blocks), b(kBlock);
#define FMHIP_AP(KP_)                                                    \
    if (rows_only) hipLaunchKernelGGL((k_apply_rows<KP_>), g, b, 0, s, a); \
    else hipLaunchKernelGGL((k_apply<KP_>), g, b, 0, s, a)
    switch (Kp) {
        case 32: FMHIP_AP(32); break;
        case 64: FMHIP_AP(64); break;
        case 128: FMHIP_AP(128); break;
        case 256: FMHIP_AP(256); break;
        default: return hipErrorInvalidValue;
    }
#undef FMHIP_AP
    return hipGetLastError();
}

}  // namespace fmhip
