// fm_apply.hip — the parameter update of the FM step (dense, or the rows a batch touched), fused with
// zeroing the packed gradient.
#include "fm_device.h"

namespace fmhip {
namespace {

// ------------------------------------------------------------------ apply
// theta <- theta - eta*(g/|B| + reg*theta) with g_V = G_V - v*G_b (S/fm/lib/ALS.scala:56-58:
// sum e*(x*q - x^2*v)); the packed gradient is zeroed on the way out.
// one float4 of one feature row: theta <- theta - eta*(g/|B| + lambda*theta), then the gradient is zeroed
template <int KP>
__device__ __forceinline__ void apply_piece(const ApplyArgs &a, int64_t i, int c, float invb) {
    constexpr int LPR = KP / 4;
    float4 *V4 = reinterpret_cast<float4 *>(a.V) + i * LPR + c;
    float4 *G4 = reinterpret_cast<float4 *>(a.GV) + i * LPR + c;
    const float b = a.Gb[i];
    float4 g = *G4, v = *V4;
    float wslot = 0.f;
    const bool has_w = a.pack_k >= 0 && c == (a.pack_k >> 2);
    if (has_w) {   // packed rows: this float4 holds the linear weight in component pack_k & 3
        const float wi = f4pick(v, a.pack_k & 3);
        wslot = wi - a.eta * fmaf(a.regw, wi, f4pick(g, a.pack_k & 3) * invb);
    }
    v.x -= a.eta * fmaf(a.regv, v.x, (g.x - v.x * b) * invb);
    v.y -= a.eta * fmaf(a.regv, v.y, (g.y - v.y * b) * invb);
    v.z -= a.eta * fmaf(a.regv, v.z, (g.z - v.z * b) * invb);
    v.w -= a.eta * fmaf(a.regv, v.w, (g.w - v.w * b) * invb);
    if (has_w) f4set(v, a.pack_k & 3, wslot);
    *V4 = v;
    *G4 = f4zero();
    if (c == 0) {
        const float wi = a.w[i];
        a.w[i] = wi - a.eta * fmaf(a.regw, wi, a.Gw[i] * invb);
        a.Gw[i] = 0.f;
        a.Gb[i] = 0.f;  // same wave already holds its copy of b (all lanes of a row share a wave)
    }
}

__device__ __forceinline__ void apply_w0(const ApplyArgs &a, float invb) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float w0 = *a.w0;
        *a.w0 = w0 - a.eta * fmaf(a.reg0, w0, a.scal[0] * invb);
    }
}

template <int KP>
__global__ __launch_bounds__(kBlock) void k_apply(ApplyArgs a) {
    constexpr int LPR = KP / 4;  // lanes per feature row (<= 64, divides the wave)
    const float invb = a.scal[2] > 0.f ? 1.0f / a.scal[2] : 0.f;
    const int64_t total = a.n1 * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock)
        apply_piece<KP>(a, idx / LPR, (int)(idx % LPR), invb);
    apply_w0(a, invb);
}

// The same update restricted to the rows a batch touched (its distinct features + the dense hot
// block's): with regw = regv = 0 every other row has a zero gradient and no decay, so the dense pass
// would rewrite it unchanged.  Matters when the model is far wider than a batch (Criteo-like widths:
// 2^25 rows of V, 8.6 GB, against ~2 M touched).
template <int KP>
__global__ __launch_bounds__(kBlock) void k_apply_rows(ApplyArgs a) {
    constexpr int LPR = KP / 4;
    const float invb = a.scal[2] > 0.f ? 1.0f / a.scal[2] : 0.f;
    const int64_t total = ((int64_t)a.n_feat + a.n_hot) * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t j = idx / LPR;
        const int32_t i = j < a.n_feat ? a.feat[j] : a.hot_ids[j - a.n_feat];
        if (i >= 0) apply_piece<KP>(a, i, (int)(idx % LPR), invb);
    }
    apply_w0(a, invb);
}


}  // namespace

hipError_t launch_apply(int Kp, const ApplyArgs &a, hipStream_t s) {
    const bool rows_only = a.feat != nullptr;
    int64_t total = (rows_only ? (int64_t)a.n_feat + a.n_hot : a.n1) * (Kp / 4);
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
