// fm_apply.hip — the parameter update of the FM step (dense, or the rows a batch touched), fused with
// zeroing the packed gradient.
#include "fm_device.h"

namespace fmhip {
namespace {

// ------------------------------------------------------------------ apply
// theta <- theta - eta*(g/|B| + reg*theta) with g_V = G_V - v*G_b (S/fm/lib/ALS.scala:56-58:
// sum e*(x*q - x^2*v)); the packed gradient is zeroed on the way out.
//
// Lazy weight decay.  The tables hold U with V = sv*U (and w = sw*stored w); normally sv = sw = 1.  The
// update (1 - eta*reg)*theta - eta*g/|B| touches EVERY row through its decay factor; the rows-only pass
// instead multiplies the scale, sv' = sv*(1 - eta*regv), and updates just the rows with a gradient:
//     U_i <- U_i - (eta/sv') * (G_V - (sv*U_i)*G_b)/|B|        (so that sv'*U_i' is the eager result)
// — no per-row timestamps and no catch-up pass: the forward multiplies its row sums by the scale
// (row_finish) and the next dense pass (below) folds it back in.  With no decay and sv = 1 both passes
// perform the same operations on the same values (bit-identical; tested).
//
// one float4 of one feature row; ROWS = the rows-only (lazy) form
template <int KP, bool ROWS>
__device__ __forceinline__ void apply_piece(const ApplyArgs &a, int64_t i, int c, float invb) {
    constexpr int LPR = KP / 4;
    float4 *V4 = reinterpret_cast<float4 *>(a.V) + i * LPR + c;
    float4 *G4 = reinterpret_cast<float4 *>(a.GV) + i * LPR + c;
    const float b = a.Gb[i];
    float4 g = *G4, u = *V4;
    float4 v = f4mul(u, a.sv_in);                      // the parameter values (x 1 is exact)
    float wslot = 0.f;
    const bool has_w = a.pack_k >= 0 && c == (a.pack_k >> 2);
    if (has_w) {   // packed rows: this float4 holds the linear weight in component pack_k & 3
        const float us = f4pick(u, a.pack_k & 3), gi = f4pick(g, a.pack_k & 3) * invb;
        const float wi = us * a.sw_in;
        wslot = ROWS ? us - a.eta_w * gi : wi - a.eta * fmaf(a.regw, wi, gi);
    }
    if (ROWS) {
        u.x -= a.eta_v * ((g.x - v.x * b) * invb);
        u.y -= a.eta_v * ((g.y - v.y * b) * invb);
        u.z -= a.eta_v * ((g.z - v.z * b) * invb);
        u.w -= a.eta_v * ((g.w - v.w * b) * invb);
    } else {
        u.x = v.x - a.eta * fmaf(a.regv, v.x, (g.x - v.x * b) * invb);
        u.y = v.y - a.eta * fmaf(a.regv, v.y, (g.y - v.y * b) * invb);
        u.z = v.z - a.eta * fmaf(a.regv, v.z, (g.z - v.z * b) * invb);
        u.w = v.w - a.eta * fmaf(a.regv, v.w, (g.w - v.w * b) * invb);
    }
    if (has_w) f4set(u, a.pack_k & 3, wslot);
    *V4 = u;
    *G4 = f4zero();
    if (c == 0) {
        const float us = a.w[i], gi = a.Gw[i] * invb;
        const float wi = us * a.sw_in;
        a.w[i] = ROWS ? us - a.eta_w * gi : wi - a.eta * fmaf(a.regw, wi, gi);
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
    const float rows = *a.rows, invb = rows > 0.f ? 1.0f / rows : 0.f;
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
    const float rows = *a.rows, invb = rows > 0.f ? 1.0f / rows : 0.f;
    const int64_t total = ((int64_t)a.n_feat + a.n_hot) * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t j = idx / LPR;
        const int32_t i = j < a.n_feat ? a.feat[j] : a.hot_ids[j - a.n_feat];
        if (i >= 0) apply_piece<KP, true>(a, i, (int)(idx % LPR), invb);
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

}  // namespace

hipError_t launch_init_normal(int Kp, float *V, float *w, float *w0, int64_t n1, int64_t n1p, int32_t k, uint64_t seed, float mean,
                              float stdev, hipStream_t s) {
    int64_t blocks = (n1p * Kp + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_init_normal, dim3((unsigned)blocks), dim3(kBlock), 0, s, V, w, w0, n1, n1p, k, Kp, seed, mean, stdev);
    return hipGetLastError();
}

hipError_t launch_apply(int Kp, const ApplyArgs &a, hipStream_t s) {
    const bool rows_only = a.feat != nullptr;
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
