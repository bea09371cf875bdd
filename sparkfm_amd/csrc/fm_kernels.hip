// fm_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the FM mini-batch SGD step.
//
// Work shapes (HBM/L2-bound gather + stream at ~1 flop/B; the one GEMM-shaped piece, the gradient of
// the dense hot block, is fp32 MFMA):
//   k_forward  CSR rows  : per stored nonzero gather one Kp-float row of V (128 B at Kp=32); the
//                          dense hot block's features come from LDS
//   k_backward CSC ranges: per stored nonzero gather one Kp-float row of P = e*q; its first workgroups
//                          form the hot block's gradient, xhot^T . P, instead
//   k_fixup    sums the partials of columns cut across ranges and of the hot block (fixed order ->
//              deterministic)
//   k_apply    SGD update fused with zeroing the packed gradient (dense, or the touched rows only)
//
// Lane geometry: a "slot" = LPN consecutive lanes (8 at Kp = 32, 16 above) owning one CSR row
// (forward) or one CSC range (backward); lane l of a slot holds factors 4*(l + jj*LPN) .. +3 for
// jj < J, so one wave-instruction moves 64/LPN whole rows of Kp = 4*LPN*J floats (whole 128-B
// lines: the texture addresser charges ~2 cycles per distinct line, whatever the bytes used), each
// row a contiguous, 16-B-per-lane coalesced segment.  Index/value streams are read LPN entries at a time (one
// per lane, contiguous) and broadcast inside the slot (two DPP moves for 8-lane slots, ds_bpermute above).
//
// Formulas restated from SparkFM (S/ = src/main/scala/io/edstud/spark/):
//   forward   S/fm/FMModel.scala:34-63   yhat = w0 + sum w x + 0.5*sum_f[(sum v x)^2 - sum (v x)^2]
//   residual  S/fm/lib/ALS.scala:142-144 e = yhat - y
//   q         S/fm/lib/ALS.scala:146-150 q_f = sum_i v_fi x_i
//   gradient  S/fm/lib/ALS.scala:56-58   h(v_fi) = x*q_f - x^2*v_fi ; :40 h(w_i) = x ; :21 h(w0) = 1
#include "fm_kernels.h"
#ifndef FMHIP_DPP_BCAST
#define FMHIP_DPP_BCAST 1
#endif

namespace fmhip {

int g_tune[kTuneCount] = {60, 1, 0, 0, 0, 1, 0, 1};   // forward: w-tile kernel; backward: pipelined; tile rows: auto; row blocks, XCD placement: off; hot block: on

int padded_factors(int k) {
    int kp = 32;   // a row is at least one 128-B line: the cost of a gather is per line, not per byte
    while (kp < k) kp <<= 1;
    return kp;
}

int forward_wt_occupancy(int Kp);   // workgroups of k_forward_wt one CU holds (defined next to the kernel)

int forward_blocks_wt(int Kp, int64_t n_rows) {
    // persistent: as many workgroups as the chip holds at once (24 KiB of LDS each; the register count
    // decides: 5 per CU at Kp = 32, fewer for wider rows), rows grid-strided
    const int64_t need = forward_blocks(Kp, n_rows);
    int occ = forward_wt_occupancy(Kp);
    if (g_tune[kTuneFwdOcc] > 0 && g_tune[kTuneFwdOcc] < occ) occ = g_tune[kTuneFwdOcc];   // experiment knob: fewer resident workgroups
    const int64_t cap = (int64_t)256 * occ;
    return (int)(need < cap ? need : cap);
}

int forward_blocks_lds(int64_t n_rows) {
    (void)n_rows;
    return 256;   // one 1024-thread workgroup per CU, rows grid-strided
}

int forward_blocks(int Kp, int64_t n_rows) {
    const int lpn = Kp <= 64 ? 8 : 16;      // the forward's slot width (launch_forward)
    const int slots = 256 / lpn;
    int64_t blocks = (n_rows + slots - 1) / slots;
    if (blocks > kMaxFwdBlocks) blocks = kMaxFwdBlocks;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

namespace {

constexpr int kBlock = 256;

// The CSR / CSC index and value streams are read exactly once per step: load them non-temporally
// so they do not evict the gathered tables (V, P) from L2.
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
#ifdef FMHIP_STREAM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4mul(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ void f4fma(float4 &acc, float4 a, float s) {
    acc.x = fmaf(a.x, s, acc.x); acc.y = fmaf(a.y, s, acc.y); acc.z = fmaf(a.z, s, acc.z); acc.w = fmaf(a.w, s, acc.w);
}
__device__ __forceinline__ void f4add(float4 &acc, float4 a) { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
__device__ __forceinline__ void f4sqacc(float4 &acc, float4 a) {
    acc.x = fmaf(a.x, a.x, acc.x); acc.y = fmaf(a.y, a.y, acc.y); acc.z = fmaf(a.z, a.z, acc.z); acc.w = fmaf(a.w, a.w, acc.w);
}
// component c (0..3) of a float4 without dynamic register indexing
__device__ __forceinline__ float f4pick(float4 v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }
__device__ __forceinline__ void f4set(float4 &v, int c, float x) {
    if (c == 0) v.x = x; else if (c == 1) v.y = x; else if (c == 2) v.z = x; else v.w = x;
}

// (q*q - s) with the product rounded BEFORE the subtraction (no fma contraction): for a
// single-nonzero row q = v*x and s = round((v*x)^2), so this is exactly 0 (quirk Q6).
// (HIP's __fmul_rn/__fsub_rn are plain operators that hipcc would still contract into one fma,
// hence the explicit contract(off).)
__device__ __forceinline__ float sq_minus(float q, float s) {
#pragma clang fp contract(off)
    const float qq = q * q;
    return qq - s;
}
__device__ __forceinline__ float f4sqminus(float4 q, float4 s) {
    return (sq_minus(q.x, s.x) + sq_minus(q.y, s.y)) + (sq_minus(q.z, s.z) + sq_minus(q.w, s.w));
}

// Raw buffer view of a row table (V or P): a load whose byte offset is >= `bytes` returns 0 and
// fetches nothing, so padding entries of a lane group cost no memory traffic and need no mask.
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0xffffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}

// ------------------------------------------------------------------ forward
// LDS V-tile variant: a 1024-thread workgroup (one per CU) first stages the rows of the T hottest
// features — ids < T, i.e. frequency-ranked ids — of V (and of w) into LDS, then walks its rows
// like k_forward_p.  A nonzero whose feature id is < T reads its factor row with ds_read_b128 and
// issues NO global request (its buffer offset is out of range); only the colder ids go to L2.
// With power-law ids the tile absorbs most gathers (58 % at T = 1024 on the C3 workload).
constexpr int kLdsBlock = 1024;

// ------------------------------------------------------------------ dense hot block (forward side)
// The kHotT most frequent features of a dataset are not in its sparse streams: their values sit in
// xhot[row][slot] (0 = absent) and their V rows / linear weights are staged in LDS once per
// workgroup, so a hot nonzero costs an LDS read instead of a 128-B gather through the texture
// addresser (profiles/r01_experiments.md §12/§15).  Contributions enter q, s and the linear term
// exactly as a stored nonzero's would (FMModel.scala:41-46,57-63); an absent feature adds nothing.
template <int KP>
__device__ __forceinline__ bool hot_stage(const FwdArgs &a, float *vh, float *wh) {
    int bad = 0;
    for (int i = threadIdx.x; i < kHotT * (KP / 4); i += blockDim.x) {
        const int h = i / (KP / 4), c = i % (KP / 4);
        const int id = a.hot_ids[h];
        const float4 t = id >= 0 ? reinterpret_cast<const float4 *>(a.V + (size_t)id * KP)[c] : f4zero();
        reinterpret_cast<float4 *>(vh)[i] = t;
        bad |= !(isfinite(t.x) && isfinite(t.y) && isfinite(t.z) && isfinite(t.w));
    }
    if (threadIdx.x < kHotT) {
        const int id = a.hot_ids[threadIdx.x];
        const float t = id >= 0 ? a.w[id] : 0.f;
        wh[threadIdx.x] = t;
        bad |= !isfinite(t);
    }
    // (also the barrier that publishes the tile) all staged parameters finite: 0 * v is exactly 0 and
    // an absent slot needs no masking; otherwise the masked prologue keeps absent features out
    return __syncthreads_or(bad) == 0;
}

// The row's kHotT values, issued together with the row's offsets so their latency is paid once.  Lane
// l of the slot loads the float4 of hot slots 4*(l&3)..+3 — 16 B per lane: the texture path returns
// 64 B/clk/CU whatever the coalescing, so every lane loading all 64 B would cost four times as much —
// and the prologue broadcasts each value inside the quad with a DPP move (no LDS crossbar).
__device__ __forceinline__ float4 hot_load(const FwdArgs &a, int r, int l) {
    return reinterpret_cast<const float4 *>(a.xhot + (size_t)r * kHotT)[l & 3];
}

template <int G>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), G * 0x55, 0xf, 0xf, false));   // quad_perm:[G,G,G,G]
}

constexpr bool g_dpp_bcast = FMHIP_DPP_BCAST;
// Broadcast of lane `SRC` of every 8-lane slot to the slot's lanes with two DPP moves on the vector ALU
// (quad broadcast, then a 4-lane row shift into the other quad of the slot) instead of a ds_bpermute
// through the one LDS crossbar per CU — the pipe the per-entry broadcasts used to keep 35-56 % busy.
template <int SRC>
__device__ __forceinline__ int slot8_bcast_i(int v) {
    const int t = __builtin_amdgcn_update_dpp(0, v, (SRC & 3) * 0x55, 0xf, 0xf, false);       // quad_perm:[s,s,s,s]
    if (SRC < 4) return __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xa, false);             // row_shr:4 into quads 1,3
    return __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);                           // row_shl:4 into quads 0,2
}

template <int LPN>
__device__ __forceinline__ int slot_bcast(int v, int src) {
    if (LPN == 8 && g_dpp_bcast) {
        switch (src) {
            case 0: return slot8_bcast_i<0>(v);
            case 1: return slot8_bcast_i<1>(v);
            case 2: return slot8_bcast_i<2>(v);
            case 3: return slot8_bcast_i<3>(v);
            case 4: return slot8_bcast_i<4>(v);
            case 5: return slot8_bcast_i<5>(v);
            case 6: return slot8_bcast_i<6>(v);
            default: return slot8_bcast_i<7>(v);
        }
    }
    return __shfl(v, src, LPN);
}
template <int LPN>
__device__ __forceinline__ float slot_bcast(float v, int src) { return __int_as_float(slot_bcast<LPN>(__float_as_int(v), src)); }
template <int LPN>
__device__ __forceinline__ uint32_t slot_bcast(uint32_t v, int src) { return (uint32_t)slot_bcast<LPN>((int)v, src); }

template <int LPN, int J, bool WITH_LIN, bool MASKED, int G>
__device__ __forceinline__ void hot_group(const float4 xq, const float *vh, const float *wh, int l, float4 (&q)[J],
                                          float4 (&s)[J], float &lin) {
    constexpr int KP = 4 * LPN * J;
    const float xs[4] = {quad_bcast<G>(xq.x), quad_bcast<G>(xq.y), quad_bcast<G>(xq.z), quad_bcast<G>(xq.w)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int h = G * 4 + c;
        const float x = xs[c];
        const bool live = x != 0.f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            float4 tv = f4mul(reinterpret_cast<const float4 *>(vh + h * KP)[jj * LPN + l], x);
            if (MASKED && !live) tv = f4zero();
            f4add(q[jj], tv);
            f4sqacc(s[jj], tv);
        }
        if (WITH_LIN && (h & (LPN - 1)) == l && (!MASKED || live)) lin = fmaf(wh[h], x, lin);
    }
    // four slots' LDS reads in flight at a time: the prologue must not raise the kernel's register count
    __builtin_amdgcn_sched_barrier(0);
}

template <int LPN, int J, bool WITH_LIN, bool MASKED>
__device__ __forceinline__ void hot_prologue(const float4 xq, const float *vh, const float *wh, int l,
                                             float4 (&q)[J], float4 (&s)[J], float &lin) {
    static_assert(kHotT == 16, "one float4 per quad lane");
    hot_group<LPN, J, WITH_LIN, MASKED, 0>(xq, vh, wh, l, q, s, lin);
    hot_group<LPN, J, WITH_LIN, MASKED, 1>(xq, vh, wh, l, q, s, lin);
    hot_group<LPN, J, WITH_LIN, MASKED, 2>(xq, vh, wh, l, q, s, lin);
    hot_group<LPN, J, WITH_LIN, MASKED, 3>(xq, vh, wh, l, q, s, lin);
}

// One step of a row walk: the slot's LPN entries (c, x: one per lane) are broadcast, their V rows
// gathered CH at a time and accumulated in stored order (q_f: FMModel.scala:59, sum_sqr_f: :60).
// MASKED = the row's last, partial step (entries >= cnt are dead); full steps carry no per-entry
// compare/select — the vector ALU, not the memory path, is what the forward saturates
// (profiles/r01_experiments.md, section 23).
template <int LPN, int J, int CH, bool MASKED>
__device__ __forceinline__ void fwd_step(const float *V, int c, float x, int cnt, int l, float4 (&q)[J], float4 (&s)[J]) {
    constexpr int KP = 4 * LPN * J;
#pragma unroll
    for (int c0 = 0; c0 < LPN; c0 += CH) {
        float4 t[CH][J];
        float xs[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int cj = slot_bcast<LPN>(c, c0 + j);
            xs[j] = slot_bcast<LPN>(x, c0 + j);
            const float4 *vr = reinterpret_cast<const float4 *>(V + (size_t)(uint32_t)cj * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) t[j][jj] = vr[jj * LPN];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool live = c0 + j < cnt;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                float4 tv = f4mul(t[j][jj], xs[j]);
                if (MASKED && !live) tv = f4zero();
                f4add(q[jj], tv);
                f4sqacc(s[jj], tv);
            }
        }
    }
}

template <int LPN, int J, int MODE>
__global__ __launch_bounds__(kLdsBlock) void k_forward_lds(FwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    constexpr int SLOTS = kLdsBlock / LPN;
    constexpr int CH = (LPN * J > 8) ? ((8 / J) > 0 ? (8 / J) : 1) : LPN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.tile_rows;
    float *vt = lds;                    // [T][KP]
    float *wt = lds + (size_t)T * KP;   // [T]
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.V);
        float4 *dst = reinterpret_cast<float4 *>(vt);
        const int n4 = T * (KP / 4);
        for (int i = threadIdx.x; i < n4; i += kLdsBlock) dst[i] = src[i];
        for (int i = threadIdx.x; i < T; i += kLdsBlock) wt[i] = a.w[i];
    }
    __syncthreads();
    const int l = threadIdx.x & (LPN - 1);
    const int slot = threadIdx.x / LPN;
    const float w0 = *a.w0;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(a.V, a.v_bytes);
    float st1 = 0.f, st2 = 0.f, stbad = 0.f;
    // rows are taken in the dataset's length-sorted order (longest first): the slots of a wave walk rows of
    // (nearly) equal length, so no lane idles while a neighbour finishes a longer row, and every slot's
    // share — one row per length stratum — weighs the same
    for (int ri = blockIdx.x * SLOTS + slot; ri < a.n_rows; ri += gridDim.x * SLOTS) {
        const int r = a.order ? a.order[ri] : ri;
        const int64_t p0 = a.row_ptr[a.row0 + r], p1 = a.row_ptr[a.row0 + r + 1];
        float4 q[J], s[J];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) { q[jj] = f4zero(); s[jj] = f4zero(); }
        float lin = 0.f;
        for (int64_t base = p0; base < p1; base += LPN) {
            const int64_t p = base + l;
            int c = -1;
            float x = 0.f;
            if (p < p1) {
                c = stream_load(a.col + p);
                x = stream_load(a.val + p);
                const float wv = c < T ? wt[c] : a.w[c];
                lin = fmaf(wv, x, lin);
            }
#pragma unroll
            for (int c0 = 0; c0 < LPN; c0 += CH) {
                float4 tg[CH][J], tl[CH][J];
                float xs[CH];
                bool hot[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int cj = slot_bcast<LPN>(c, c0 + j);
                    xs[j] = slot_bcast<LPN>(x, c0 + j);
                    hot[j] = (unsigned)cj < (unsigned)T;            // false for dead entries (cj = -1)
                    const uint32_t off = (uint32_t)cj * (KP * 4u) + (uint32_t)l * 16u;
                    const float4 *lp = reinterpret_cast<const float4 *>(vt + (size_t)(hot[j] ? cj : 0) * KP) + l;
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) {
                        tg[j][jj] = buf_load4(vr, (hot[j] || cj < 0) ? kOob : off + jj * LPN * 16u);
                        tl[j][jj] = lp[jj * LPN];
                    }
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) {
                        const float4 t = hot[j] ? tl[j][jj] : tg[j][jj];
                        const float4 tv = f4mul(t, xs[j]);          // dead entries: 0 * 0
                        f4add(q[jj], tv);
                        f4sqacc(s[jj], tv);
                    }
                }
            }
        }
        float u = 0.f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) u += f4sqminus(q[jj], s[jj]);
        float tot = fmaf(0.5f, u, lin);
#pragma unroll
        for (int m = LPN >> 1; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, LPN);
        const float yhat = w0 + tot;
        const float e = yhat - a.y[a.row0 + r];
        if (MODE == kFwdTrain) {
            float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = f4mul(q[jj], e);
        } else if (MODE == kFwdQ) {
            float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = q[jj];
        }
        if (l == 0) {
            if (a.e) a.e[r] = e;
            if (a.yhat) a.yhat[r] = yhat;
            st1 += e;
            st2 = fmaf(e, e, st2);
            if (!isfinite(e)) stbad += 1.f;
        }
    }
    if (a.bsum) {
        __shared__ double sh[3][kLdsBlock / 64];
        double d1 = st1, d2 = st2, db = stbad;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            d1 += __shfl_xor(d1, m, 64);
            d2 += __shfl_xor(d2, m, 64);
            db += __shfl_xor(db, m, 64);
        }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[0][wv] = d1; sh[1][wv] = d2; sh[2][wv] = db; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t1 = 0.0, t2 = 0.0, tb = 0.0;
#pragma unroll
            for (int i = 0; i < kLdsBlock / 64; ++i) { t1 += sh[0][i]; t2 += sh[1][i]; tb += sh[2][i]; }
            double *o = a.bsum + (size_t)blockIdx.x * 4;
            o[0] = t1; o[1] = t2; o[2] = tb; o[3] = 0.0;
        }
    }
}

template <int LPN, int J, int MODE, bool PACKED, bool HOT>
__global__ __launch_bounds__(kBlock) void k_forward(FwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    constexpr int SLOTS = kBlock / LPN;
    constexpr int CH = (LPN * J > 16) ? (16 / J) : LPN;  // entries whose V rows are in flight together
    const int l = threadIdx.x & (LPN - 1);
    const int slot = threadIdx.x / LPN;
    const float w0 = *a.w0;
    // Packed rows (k < Kp): slot k of every V row holds the feature's linear weight w_i, so q_k
    // accumulates sum w_i x_i — the linear term — for free and there is no separate w gather (a
    // 64-lane scalar gather costs the texture addresser as much as the whole row gather); slot k of
    // the P row carries e to the backward the same way.
    constexpr bool packed = PACKED;
    const int kl = packed ? (a.pack_k >> 2) & (LPN - 1) : 0, kj = packed ? (a.pack_k >> 2) / LPN : 0, kc = a.pack_k & 3;
    __shared__ __attribute__((aligned(16))) float vh[HOT ? kHotT * KP : 4];
    __shared__ float wh[HOT ? kHotT : 1];
    bool hot_plain = false;
    if (HOT) hot_plain = hot_stage<KP>(a, vh, wh);
    float st1 = 0.f, st2 = 0.f, stbad = 0.f;   // this thread's share of {sum e, sum e^2, nonfinite}
    // rows are taken in the dataset's length-sorted order (longest first): the slots of a wave walk rows of
    // (nearly) equal length, so no lane idles while a neighbour finishes a longer row, and every slot's
    // share — one row per length stratum — weighs the same
    for (int ri = blockIdx.x * SLOTS + slot; ri < a.n_rows; ri += gridDim.x * SLOTS) {
        const int r = a.order ? a.order[ri] : ri;
        float4 xh = f4zero();
        if (HOT) xh = hot_load(a, r, l);
        const int64_t p0 = a.row_ptr[a.row0 + r], p1 = a.row_ptr[a.row0 + r + 1];
        float4 q[J], s[J];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) { q[jj] = f4zero(); s[jj] = f4zero(); }
        float lin = 0.f;
        if (HOT) {
            if (hot_plain) hot_prologue<LPN, J, !PACKED, false>(xh, vh, wh, l, q, s, lin);
            else hot_prologue<LPN, J, !PACKED, true>(xh, vh, wh, l, q, s, lin);
        }
        int64_t base = p0;
        for (; base + LPN <= p1; base += LPN) {        // full steps
            const int c = stream_load(a.col + base + l);
            const float x = stream_load(a.val + base + l);
            float wv = 0.f;
            if (!packed) wv = a.w[c];
            fwd_step<LPN, J, CH, false>(a.V, c, x, LPN, l, q, s);
            if (!packed) lin = fmaf(wv, x, lin);
        }
        if (base < p1) {                               // the row's last, partial step
            const int64_t p = base + l;
            int c = 0;
            float x = 0.f, wv = 0.f;
            if (p < p1) {
                c = stream_load(a.col + p);
                x = stream_load(a.val + p);
                if (!packed) wv = a.w[c];
            }
            fwd_step<LPN, J, CH, true>(a.V, c, x, (int)(p1 - base), l, q, s);
            if (!packed) lin = fmaf(wv, x, lin);
        }
        float lin_all = 0.f;   // packed: the complete linear term, identical in every lane of the slot
        if (packed) {
            float lk = 0.f;
#pragma unroll
            for (int jj = 0; jj < J; ++jj)
                if (jj == kj) {
                    lk = f4pick(q[jj], kc);
                    if (l == kl) { f4set(q[jj], kc, 0.f); f4set(s[jj], kc, 0.f); }   // slot k is not a factor
                }
            lin_all = __shfl(lk, kl, LPN);
        }
        float u = 0.f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) u += f4sqminus(q[jj], s[jj]);
        float tot = fmaf(0.5f, u, lin);
#pragma unroll
        for (int m = LPN >> 1; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, LPN);
        const float yhat = w0 + (tot + lin_all);
        const float e = yhat - a.y[a.row0 + r];
        if (MODE == kFwdTrain) {
            float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                float4 o = f4mul(q[jj], e);
                if (packed && jj == kj && l == kl) f4set(o, kc, e);   // slot k of the P row carries e
                pr[jj * LPN] = o;
            }
        } else if (MODE == kFwdQ) {
            float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = q[jj];
        }
        if (l == 0) {
            if (a.e) a.e[r] = e;
            if (a.yhat) a.yhat[r] = yhat;
            st1 += e;
            st2 = fmaf(e, e, st2);
            if (!isfinite(e)) stbad += 1.f;
        }
    }
    // block partial of the residual statistics (fixed order; k_reduce_blocks finishes the sum)
    if (a.bsum) {
        __shared__ double sh[3][kBlock / 64];
        double d1 = st1, d2 = st2, db = stbad;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            d1 += __shfl_xor(d1, m, 64);
            d2 += __shfl_xor(d2, m, 64);
            db += __shfl_xor(db, m, 64);
        }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[0][wv] = d1; sh[1][wv] = d2; sh[2][wv] = db; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t1 = 0.0, t2 = 0.0, tb = 0.0;
#pragma unroll
            for (int i = 0; i < kBlock / 64; ++i) { t1 += sh[0][i]; t2 += sh[1][i]; tb += sh[2][i]; }
            double *o = a.bsum + (size_t)blockIdx.x * 4;
            o[0] = t1; o[1] = t2; o[2] = tb; o[3] = 0.0;
        }
    }
}

// k_forward with an LDS-resident tile of the hot linear weights (see the comment in the body)
// (second launch bound = waves per SIMD: the persistent grid of forward_blocks_wt is sized for 5, and a
// register count that admits only 4 would run it in two rounds)
template <int LPN, int J, int MODE, bool HOT>
__global__ __launch_bounds__(kBlock, (LPN * J <= 8 ? 5 : 1)) void k_forward_wt(FwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    constexpr int SLOTS = kBlock / LPN;
    constexpr int CH = (LPN * J > 16) ? (16 / J) : LPN;  // entries whose V rows are in flight together
    const int l = threadIdx.x & (LPN - 1);
    const int slot = threadIdx.x / LPN;
    const float w0 = *a.w0;
    // w-tile: the linear weights of the wt_rows lowest (= hottest, for frequency-ranked ids) feature
    // ids live in LDS.  A 64-lane gather of w costs ~2 TA cycles per distinct line touched — as much
    // per nonzero as the whole 128-B V-row gather (profiles/r01_experiments.md §13); lanes whose id is
    // in the tile read LDS instead and drop out of the global gather.
    extern __shared__ __attribute__((aligned(16))) float wt[];
    const int T = a.wt_rows;
    __shared__ __attribute__((aligned(16))) float vh[HOT ? kHotT * KP : 4];
    __shared__ float wh[HOT ? kHotT : 1];
    for (int i = threadIdx.x; i < T; i += kBlock) wt[i] = a.w[i];
    bool hot_plain = false;
    if (HOT) hot_plain = hot_stage<KP>(a, vh, wh);
    else __syncthreads();
    float st1 = 0.f, st2 = 0.f, stbad = 0.f;   // this thread's share of {sum e, sum e^2, nonfinite}
    // rows are taken in the dataset's length-sorted order (longest first): the slots of a wave walk rows of
    // (nearly) equal length, so no lane idles while a neighbour finishes a longer row, and every slot's
    // share — one row per length stratum — weighs the same
    for (int ri = blockIdx.x * SLOTS + slot; ri < a.n_rows; ri += gridDim.x * SLOTS) {
        const int r = a.order ? a.order[ri] : ri;
        float4 xh = f4zero();
        if (HOT) xh = hot_load(a, r, l);
        const int64_t p0 = a.row_ptr[a.row0 + r], p1 = a.row_ptr[a.row0 + r + 1];
        float4 q[J], s[J];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) { q[jj] = f4zero(); s[jj] = f4zero(); }
        float lin = 0.f;
        if (HOT) {
            if (hot_plain) hot_prologue<LPN, J, true, false>(xh, vh, wh, l, q, s, lin);
            else hot_prologue<LPN, J, true, true>(xh, vh, wh, l, q, s, lin);
        }
        int64_t base = p0;
        for (; base + LPN <= p1; base += LPN) {        // full steps
            const int c = stream_load(a.col + base + l);
            const float x = stream_load(a.val + base + l);
            const float wv = c < T ? wt[c] : a.w[c];
            fwd_step<LPN, J, CH, false>(a.V, c, x, LPN, l, q, s);
            lin = fmaf(wv, x, lin);                    // consumed after the gathers are on their way
        }
        if (base < p1) {                               // the row's last, partial step
            const int64_t p = base + l;
            int c = 0;
            float x = 0.f, wv = 0.f;
            if (p < p1) {
                c = stream_load(a.col + p);
                x = stream_load(a.val + p);
                wv = c < T ? wt[c] : a.w[c];
            }
            fwd_step<LPN, J, CH, true>(a.V, c, x, (int)(p1 - base), l, q, s);
            lin = fmaf(wv, x, lin);
        }
        float u = 0.f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) u += f4sqminus(q[jj], s[jj]);
        float tot = fmaf(0.5f, u, lin);
#pragma unroll
        for (int m = LPN >> 1; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, LPN);
        const float yhat = w0 + tot;
        const float e = yhat - a.y[a.row0 + r];
        if (MODE == kFwdTrain) {
            float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = f4mul(q[jj], e);
        } else if (MODE == kFwdQ) {
            float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = q[jj];
        }
        if (l == 0) {
            if (a.e) a.e[r] = e;
            if (a.yhat) a.yhat[r] = yhat;
            st1 += e;
            st2 = fmaf(e, e, st2);
            if (!isfinite(e)) stbad += 1.f;
        }
    }
    // block partial of the residual statistics (fixed order; k_reduce_blocks finishes the sum)
    if (a.bsum) {
        __shared__ double sh[3][kBlock / 64];
        double d1 = st1, d2 = st2, db = stbad;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            d1 += __shfl_xor(d1, m, 64);
            d2 += __shfl_xor(d2, m, 64);
            db += __shfl_xor(db, m, 64);
        }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[0][wv] = d1; sh[1][wv] = d2; sh[2][wv] = db; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t1 = 0.0, t2 = 0.0, tb = 0.0;
#pragma unroll
            for (int i = 0; i < kBlock / 64; ++i) { t1 += sh[0][i]; t2 += sh[1][i]; tb += sh[2][i]; }
            double *o = a.bsum + (size_t)blockIdx.x * 4;
            o[0] = t1; o[1] = t2; o[2] = tb; o[3] = 0.0;
        }
    }
}

// ------------------------------------------------------------------ reduce
// one block, fixed order: sums the forward's per-block partials into
// scal = {sum e, sum e^2, rows, nonfinite} (fp32, part of the packed gradient) and acc (fp64, +=)
__device__ __forceinline__ void reduce_blocks_body(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal,
                                                   double *acc, double (*sh)[kBlock / 64]) {
    double s1 = 0.0, s2 = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) {
        const double4 b = reinterpret_cast<const double4 *>(bsum)[i];
        s1 += b.x;
        s2 += b.y;
        bad += b.z;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
        bad += __shfl_xor(bad, m, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wv] = s1; sh[1][wv] = s2; sh[2][wv] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0, tb = 0.0;
        for (int i = 0; i < kBlock / 64; ++i) { t1 += sh[0][i]; t2 += sh[1][i]; tb += sh[2][i]; }
        if (scal) { scal[0] = (float)t1; scal[1] = (float)t2; scal[2] = (float)n_rows; scal[3] = (float)tb; }
        if (acc) { acc[0] += t1; acc[1] += t2; acc[2] += (double)n_rows; acc[3] += tb; }
    }
}

__global__ __launch_bounds__(kBlock) void k_reduce_blocks(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal,
                                                         double *acc) {
    __shared__ double sh[3][kBlock / 64];
    reduce_blocks_body(bsum, nblocks, n_rows, scal, acc, sh);
}

// ------------------------------------------------------------------ backward
template <int LPN, int J>
__device__ __forceinline__ void store_row(float *dst, int l, const float4 (&acc)[J], float sa, float sb, float *dsa, float *dsb,
                                          int sl = 0) {
    float4 *d4 = reinterpret_cast<float4 *>(dst) + l;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) d4[jj * LPN] = acc[jj];
    if (l == sl) { *dsa = sa; *dsb = sb; }   // sl: the lane whose scalar sums are the real ones
}

// A finished column piece goes to its destination: the G row of its feature when the feature has a
// single piece in the batch, else a piece row that k_fixup2 sums per feature (row-blocked streams).
template <int LPN, int J>
__device__ __forceinline__ void store_seg(const BwdArgs &a, int seg, int l, const float4 (&acc)[J], float sa, float sb,
                                          int sl = 0) {
    constexpr int KP = 4 * LPN * J;
    const int dst = a.cdst[seg];
    if (dst >= 0) {
        store_row<LPN, J>(a.GV + (size_t)dst * KP, l, acc, sa, sb, a.Gw + dst, a.Gb + dst, sl);
    } else {
        float *pr = a.pieces + (size_t)(-1 - dst) * (KP + kPartPad);
        store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
    }
}

// sa += e*x (-> G_w, h(w_i) = x); sb += e*x^2 (-> G_b, the -x^2*v term of h(v)).  Written with
// explicit fma's so that every code path (plain-chunk fast path, flush path, both kernels) rounds
// identically whatever the compiler's contraction choices: results do not depend on which path a
// wave happened to take.
__device__ __forceinline__ void accum_scalars(float &sa, float &sb, float e, float x) {
    sa = fmaf(e, x, sa);
    sb = fmaf(__fmul_rn(e, x), x, sb);
}

template <int LPN, int J>
__device__ __forceinline__ void slots_reduce(float4 (&acc)[J], float &sa, float &sb) {
#pragma unroll
    for (int m = 32; m >= LPN; m >>= 1) {
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            acc[jj].x += __shfl_xor(acc[jj].x, m, 64);
            acc[jj].y += __shfl_xor(acc[jj].y, m, 64);
            acc[jj].z += __shfl_xor(acc[jj].z, m, 64);
            acc[jj].w += __shfl_xor(acc[jj].w, m, 64);
        }
        sa += __shfl_xor(sa, m, 64);
        sb += __shfl_xor(sb, m, 64);
    }
}

// ------------------------------------------------------------------ dense hot block (gradient side)
// G_V[hot h][f] = sum_r xhot[r][h] * P[r][f], G_w = sum_r e_r x, G_b = sum_r e_r x^2: a dense
// [kHotT x rows] . [rows x Kp] product, the one GEMM-shaped piece of the path, streamed once over P
// and xhot.  Each wave owns a contiguous run of rows and feeds them four at a time to
// v_mfma_f32_16x16x4_f32 (exact f32, a k-ordered fmaf chain): A[h][k] = xhot[r0+k][h] is ONE
// coalesced dword load per lane (lane l <-> xhot[r0*16 + l]), B[k][f] = P[r0+k][16j + f]; the 16 x Kp
// result lives in Kp/16 accumulators of 4 registers.  The two scalar sums ride on the A operand's
// lanes.  Waves of a workgroup are summed through LDS in wave order into one partial per workgroup;
// hot_reduce_body sums the partials in workgroup order: fixed orders, bit-identical run to run, no atomics.
// The body runs in the FIRST hot_blocks workgroups of the backward launch (HBM streaming next to the
// gather-bound column walk of the other workgroups, no extra launch); the reduction rides in k_fixup.
template <int NJ>
__device__ __forceinline__ void hot_backward_body(const HotArgs &a, int bx, int nbx) {
    static_assert(kHotT == 16 && kBlock == 256, "tile mapping below assumes a 16-slot hot block and 4 waves");
    constexpr int KP = 16 * NJ, PR = KP + kPartPad, W = kBlock / 64;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kk = lane >> 4, jj = lane & 15;
    const int nw = nbx * W;
    const int per = (((a.n_rows + nw - 1) / nw) + 3) & ~3;      // rows per wave, a multiple of 4
    const int64_t beg64 = (int64_t)(bx * W + wv) * per;
    const int r_beg = beg64 < a.n_rows ? (int)beg64 : a.n_rows;
    const int r_end = r_beg + per < a.n_rows ? r_beg + per : a.n_rows;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sa = 0.f, sb = 0.f;
    constexpr int U = NJ <= 4 ? 4 : 2;                         // 4-row steps whose loads are in flight together
    for (int r0 = r_beg; r0 < r_end; r0 += 4 * U) {
        float x[U], e[U], b[U][NJ];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 4 * u + kk;
            x[u] = 0.f;
            e[u] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[u][j] = 0.f;
            if (r < r_end) {
                x[u] = a.xhot[(size_t)r * kHotT + jj];
                const float *pr = a.P + (size_t)r * KP;
                e[u] = a.pack_k >= 0 ? pr[a.pack_k] : a.e[r];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[u][j] = pr[j * 16 + jj];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            accum_scalars(sa, sb, e[u], x[u]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u], b[u][j], acc[j], 0, 0, 0);
        }
    }
    sa += __shfl_xor(sa, 16, 64);
    sb += __shfl_xor(sb, 16, 64);
    sa += __shfl_xor(sa, 32, 64);
    sb += __shfl_xor(sb, 32, 64);                              // lanes 0..15: the sums of hot slot `lane`
    __shared__ float red[W][kHotT][17];
    __shared__ float reds[W][kHotT][2];
    float *out = a.part + (size_t)bx * kHotT * PR;
    const int oh = threadIdx.x >> 4, of = threadIdx.x & 15;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j) __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wv][kk * 4 + reg][jj] = acc[j][reg];   // C/D: row = (lane>>4)*4 + reg, col = lane&15
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) t += red[w][oh][of];
        out[(size_t)oh * PR + j * 16 + of] = t;
    }
    if (lane < kHotT) { reds[wv][lane][0] = sa; reds[wv][lane][1] = sb; }
    __syncthreads();
    if (threadIdx.x < kHotT * 2) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) t += reds[w][threadIdx.x >> 1][threadIdx.x & 1];
        out[(size_t)(threadIdx.x >> 1) * PR + KP + (threadIdx.x & 1)] = t;
    }
}

// one workgroup per hot slot: sums that slot's partial rows over the hot workgroups (groups of threads
// take interleaved partials, then the groups are summed in order) and stores the G row
__device__ __forceinline__ void hot_reduce_body(const HotArgs &a, int h, int kp) {
    const int id = a.hot_ids[h];
    if (id < 0) return;
    const int PR = kp + kPartPad, R4 = PR / 4;
    __shared__ float4 sh[kBlock];
    const int G = kBlock / R4 < 1 ? 1 : kBlock / R4;   // thread groups (R4 <= 65 <= kBlock)
    const int f = threadIdx.x % R4, g = threadIdx.x / R4;
    float4 t = f4zero();
    if (g < G)
        for (int b = g; b < a.nblk; b += G) f4add(t, reinterpret_cast<const float4 *>(a.part + ((size_t)b * kHotT + h) * PR)[f]);
    sh[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < R4) {
        float4 u = f4zero();
        for (int gg = 0; gg < G; ++gg) f4add(u, sh[gg * R4 + threadIdx.x]);
        if (threadIdx.x < kp / 4) {
            reinterpret_cast<float4 *>(a.GV + (size_t)id * kp)[threadIdx.x] = u;
        } else {
            a.Gw[id] = a.pack_k >= 0 ? 0.f : u.x;
            a.Gb[id] = u.y;
        }
    }
}

// One slot walks kRangeLen consecutive entries of the batch's CSC stream.  Column
// boundaries inside the range are handled serially (flush + reset), so every slot does the
// same amount of work whatever the column-length skew (power-law features), no atomics are
// needed and the summation order is fixed.  Outputs per closed column piece:
//   whole column inside the slot's walk         -> G rows directly
//   piece of a column begun in an earlier range -> part[rho][0]  ("head")
//   last piece, column continues past the range -> part[rho][1]  ("tail")
// Two rules keep the number of partials (and the fixup pass) small:
//   extension  a column that starts in range rho and ends within kExtend entries of the next
//              range is finished by slot rho (slot rho+1 skips those entries): short columns
//              straddling a range boundary produce no partial at all;
//   wave sum   when the whole wave's span (64/LPN ranges) lies inside ONE column the slots are
//              tree-summed in registers and a single partial is written for the wave.
// k_fixup and the host-side split list (fmhip_api.hip) apply the same two predicates.
template <int LPN, int J, bool PACKED, bool HOT>
__global__ __launch_bounds__(kBlock) void k_backward(BwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    if (HOT && (int)blockIdx.x < a.hot_blocks) {
        hot_backward_body<KP / 16>(a.hot, (int)blockIdx.x, a.hot_blocks);
        return;
    }
    const int bx = HOT ? (int)blockIdx.x - a.hot_blocks : (int)blockIdx.x;
    constexpr int SLOTS = kBlock / LPN;
    constexpr int WS = 64 / LPN;                        // slots (ranges) per wave
    constexpr int PR = KP + kPartPad;
    constexpr int CH = (LPN * J > 16) ? (16 / J) : LPN;  // entries whose P rows are in flight together
    const int l = threadIdx.x & (LPN - 1);
    // packed rows (k < Kp): slot k of the P row is e, so slot k of acc IS sum e*x (the w gradient) and
    // there is no e gather; sum e*x^2 is formed from that slot in the lane that owns it (lane sl)
    constexpr bool packed = PACKED;
    const int kl = packed ? (a.pack_k >> 2) & (LPN - 1) : 0, kj = packed ? (a.pack_k >> 2) / LPN : 0, kc = a.pack_k & 3;
    const int sl = kl;
    // a launch may cover only the ranges [rho_lo, rho_hi) (feature-chunked backward); block
    // numbering stays aligned to the global range numbering so the wave-sum predicate is unchanged.
    // xcd_chunk > 0: XCD-aware placement — workgroups b, b+8, b+16, .. share an XCD (round-robin
    // dispatch), so XCD x is given the x-th contiguous eighth of the stream: with a row-blocked
    // stream that is a few whole row blocks, whose slice of P then lives in that XCD's L2 only.
    const int blk = a.xcd_chunk > 0 ? (bx & 7) * a.xcd_chunk + (bx >> 3) : bx;
    const int rho = (a.rho_lo / SLOTS + blk) * SLOTS + threadIdx.x / LPN;
    if (rho < a.rho_lo || rho >= a.rho_hi) return;
    const int beg = rho * kRangeLen;
    const int end = (beg + kRangeLen < a.nnz) ? beg + kRangeLen : a.nnz;
    int seg = a.range_seg[rho];
    const int ca = a.cptr[seg], cb = a.cptr[seg + 1];   // the column open at `beg`
    const int wbeg = (rho - (int)((threadIdx.x & 63) / LPN)) * kRangeLen;
    const bool clean = (ca <= wbeg) && (cb >= wbeg + WS * kRangeLen);   // wave-uniform by construction
    bool is_head = ca < beg;
    int p0 = beg, stop = end;
    bool tail_partial = false;
    if (!clean) {
        if (is_head && ca >= beg - kRangeLen && cb - beg <= kExtend) {
            p0 = cb;            // slot rho-1 finishes that column
            ++seg;
            is_head = false;
        }
        if (end < a.nnz) {
            const int sn = a.range_seg[rho + 1];
            const int ca2 = a.cptr[sn], cb2 = a.cptr[sn + 1];   // the column open at `end`
            if (ca2 < end) {
                if (ca2 >= beg && cb2 - end <= kExtend) stop = cb2;   // finish it here
                else tail_partial = true;
            }
        }
    }
    float4 acc[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
    float sa = 0.f, sb = 0.f;
    for (int base = p0; base < stop; base += LPN) {
        const int p = base + l;
        uint32_t rf = 0u;
        float x = 0.f, ee = 0.f;
        if (p < stop) {
            rf = stream_load(a.crow + p);
            x = stream_load(a.cval + p);
            if (!packed) ee = a.e[rf & 0x7fffffffu];
        }
        const int cnt = (stop - base) < LPN ? (stop - base) : LPN;
#pragma unroll
        for (int c0 = 0; c0 < LPN; c0 += CH) {
            float4 pv[CH][J];
            uint32_t rj[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                rj[j] = slot_bcast<LPN>(rf, c0 + j);
                const float4 *pr = reinterpret_cast<const float4 *>(a.P + (size_t)(rj[j] & 0x7fffffffu) * KP) + l;
#pragma unroll
                for (int jj = 0; jj < J; ++jj) pv[j][jj] = pr[jj * LPN];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float xj = slot_bcast<LPN>(x, c0 + j);
                const float ej = packed ? 0.f : slot_bcast<LPN>(ee, c0 + j);
                if (c0 + j < cnt) {
                    if ((rj[j] >> 31) && (base + c0 + j != p0)) {
                        // the open column ends here: flush it
                        if (is_head) {
                            float *pr = a.part + ((size_t)rho * 2) * PR;
                            store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
                        } else {
                            store_seg<LPN, J>(a, seg, l, acc, sa, sb, sl);
                        }
                        is_head = false;
                        ++seg;
#pragma unroll
                        for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
                        sa = 0.f;
                        sb = 0.f;
                    }
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) f4fma(acc[jj], pv[j][jj], xj);  // sum x * (e*q)
                    if (packed) {
                        float pk = 0.f;
#pragma unroll
                        for (int jj = 0; jj < J; ++jj)
                            if (jj == kj) pk = f4pick(pv[j][jj], kc);
                        sb = fmaf(__fmul_rn(pk, xj), xj, sb);
                    } else {
                        accum_scalars(sa, sb, ej, xj);
                    }
                }
            }
        }
    }
    if (clean) {
        slots_reduce<LPN, J>(acc, sa, sb);
        if (beg == wbeg) {
            float *pr = a.part + ((size_t)rho * 2 + (ca == wbeg ? 1 : 0)) * PR;
            store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
        }
        return;
    }
    if (p0 >= stop) return;   // everything in this range belonged to the previous slot
    if (is_head) {
        float *pr = a.part + ((size_t)rho * 2) * PR;
        store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
    } else if (tail_partial) {
        float *pr = a.part + ((size_t)rho * 2 + 1) * PR;
        store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
    } else {
        store_seg<LPN, J>(a, seg, l, acc, sa, sb, sl);
    }
}

// Pipelined variant of k_backward (same walk, same predicates, same outputs): the CSC index /
// value / e loads of a whole super-group (up to 64 entries) are issued up front, the P-row
// gathers go through a buffer descriptor (dead entries fetch nothing) and are double-buffered in
// chunks of CHB entries so chunk c+1 is in flight while chunk c is accumulated.
// (HOT: the waves-per-SIMD bound keeps the MFMA accumulators of the hot body from costing the walkers
// their third wave)
template <int LPN, int J, bool PACKED, bool HOT>
__global__ __launch_bounds__(kBlock, (HOT && J == 1 ? 3 : 1)) void k_backward_p(BwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    if (HOT && (int)blockIdx.x < a.hot_blocks) {
        hot_backward_body<KP / 16>(a.hot, (int)blockIdx.x, a.hot_blocks);
        return;
    }
    const int bx = HOT ? (int)blockIdx.x - a.hot_blocks : (int)blockIdx.x;
    constexpr int SLOTS = kBlock / LPN;
    constexpr int WS = 64 / LPN;
    constexpr int PR = KP + kPartPad;
    constexpr int SG = (kRangeLen / LPN) < 8 ? (kRangeLen / LPN) : 8;         // lane-groups per super-group
    constexpr int CHB = (LPN * J > 8) ? ((8 / J) > 0 ? (8 / J) : 1) : LPN;   // entries per gather chunk
    constexpr int NCH = SG * LPN / CHB;                                       // chunks per super-group
    const int l = threadIdx.x & (LPN - 1);
    // packed rows (k < Kp): slot k of the P row is e, so slot k of acc IS sum e*x (the w gradient) and
    // there is no e gather; sum e*x^2 is formed from that slot in the lane that owns it (lane sl)
    constexpr bool packed = PACKED;
    const int kl = packed ? (a.pack_k >> 2) & (LPN - 1) : 0, kj = packed ? (a.pack_k >> 2) / LPN : 0, kc = a.pack_k & 3;
    const int sl = kl;
    // a launch may cover only the ranges [rho_lo, rho_hi) (feature-chunked backward); block
    // numbering stays aligned to the global range numbering so the wave-sum predicate is unchanged.
    // xcd_chunk > 0: XCD-aware placement — workgroups b, b+8, b+16, .. share an XCD (round-robin
    // dispatch), so XCD x is given the x-th contiguous eighth of the stream: with a row-blocked
    // stream that is a few whole row blocks, whose slice of P then lives in that XCD's L2 only.
    const int blk = a.xcd_chunk > 0 ? (bx & 7) * a.xcd_chunk + (bx >> 3) : bx;
    const int rho = (a.rho_lo / SLOTS + blk) * SLOTS + threadIdx.x / LPN;
    if (rho < a.rho_lo || rho >= a.rho_hi) return;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.P, a.p_bytes);
    const int beg = rho * kRangeLen;
    const int end = (beg + kRangeLen < a.nnz) ? beg + kRangeLen : a.nnz;
    int seg = a.range_seg[rho];
    const int ca = a.cptr[seg], cb = a.cptr[seg + 1];
    const int wbeg = (rho - (int)((threadIdx.x & 63) / LPN)) * kRangeLen;
    const bool clean = (ca <= wbeg) && (cb >= wbeg + WS * kRangeLen);
    bool is_head = ca < beg;
    int p0 = beg, stop = end;
    bool tail_partial = false;
    if (!clean) {
        if (is_head && ca >= beg - kRangeLen && cb - beg <= kExtend) {
            p0 = cb;
            ++seg;
            is_head = false;
        }
        if (end < a.nnz) {
            const int sn = a.range_seg[rho + 1];
            const int ca2 = a.cptr[sn], cb2 = a.cptr[sn + 1];
            if (ca2 < end) {
                if (ca2 >= beg && cb2 - end <= kExtend) stop = cb2;
                else tail_partial = true;
            }
        }
    }
    float4 acc[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
    float sa = 0.f, sb = 0.f;
    for (int sbase = p0; sbase < stop; sbase += SG * LPN) {
        uint32_t rf[SG];
        float x[SG], ee[SG];
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            const int p = sbase + g * LPN + l;
            rf[g] = 0u;
            x[g] = 0.f;
            if (p < stop) { rf[g] = stream_load(a.crow + p); x[g] = stream_load(a.cval + p); }
        }
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            const int p = sbase + g * LPN + l;
            ee[g] = 0.f;
            if (!packed && p < stop) ee[g] = a.e[rf[g] & 0x7fffffffu];
        }
        float4 pv[2][CHB][J];
        uint32_t rj[2][CHB];
        auto issue = [&](int ch, int buf) {
#pragma unroll
            for (int j = 0; j < CHB; ++j) {
                const int ent = ch * CHB + j;             // entry index inside the super-group
                const int g = ent / LPN, jl = ent % LPN;
                rj[buf][j] = slot_bcast<LPN>(rf[g], jl);
                const bool live = sbase + ent < stop;
                const uint32_t off = (rj[buf][j] & 0x7fffffffu) * (KP * 4u) + (uint32_t)l * 16u;
#pragma unroll
                for (int jj = 0; jj < J; ++jj) pv[buf][j][jj] = buf_load4(prs, live ? off + jj * LPN * 16u : kOob);
            }
        };
        issue(0, 0);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int buf = ch & 1;
            if (ch + 1 < NCH) issue(ch + 1, buf ^ 1);
            // A chunk is "plain" for a slot when all its entries are live and none of them closes
            // a column (the very first entry walked never does).  If that holds for every slot of
            // the wave the chunk is accumulated by straight-line code: no exec-mask juggling, no
            // flush paths — the common case inside long (hot) columns.
            const int cpos = sbase + ch * CHB;
            uint32_t fl = 0u;
#pragma unroll
            for (int j = 0; j < CHB; ++j) fl |= (cpos + j == p0) ? 0u : rj[buf][j];
            const bool plain = (cpos + CHB <= stop) && !(fl >> 31);
            if (__all(plain)) {
#pragma unroll
                for (int j = 0; j < CHB; ++j) {
                    const int ent = ch * CHB + j;
                    const int g = ent / LPN, jl = ent % LPN;
                    const float xj = slot_bcast<LPN>(x[g], jl);
                    const float ej = packed ? 0.f : slot_bcast<LPN>(ee[g], jl);
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) f4fma(acc[jj], pv[buf][j][jj], xj);
                    if (packed) {
                        float pk = 0.f;
#pragma unroll
                        for (int jj = 0; jj < J; ++jj)
                            if (jj == kj) pk = f4pick(pv[buf][j][jj], kc);
                        sb = fmaf(__fmul_rn(pk, xj), xj, sb);
                    } else {
                        accum_scalars(sa, sb, ej, xj);
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < CHB; ++j) {
                const int ent = ch * CHB + j;
                const int g = ent / LPN, jl = ent % LPN;
                const float xj = slot_bcast<LPN>(x[g], jl);
                const float ej = packed ? 0.f : slot_bcast<LPN>(ee[g], jl);
                if (sbase + ent < stop) {
                    if ((rj[buf][j] >> 31) && (sbase + ent != p0)) {
                        if (is_head) {
                            float *pr = a.part + ((size_t)rho * 2) * PR;
                            store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
                        } else {
                            store_seg<LPN, J>(a, seg, l, acc, sa, sb, sl);
                        }
                        is_head = false;
                        ++seg;
#pragma unroll
                        for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
                        sa = 0.f;
                        sb = 0.f;
                    }
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) f4fma(acc[jj], pv[buf][j][jj], xj);
                    if (packed) {
                        float pk = 0.f;
#pragma unroll
                        for (int jj = 0; jj < J; ++jj)
                            if (jj == kj) pk = f4pick(pv[buf][j][jj], kc);
                        sb = fmaf(__fmul_rn(pk, xj), xj, sb);
                    } else {
                        accum_scalars(sa, sb, ej, xj);
                    }
                }
            }
        }
    }
    if (clean) {
        slots_reduce<LPN, J>(acc, sa, sb);
        if (beg == wbeg) {
            float *pr = a.part + ((size_t)rho * 2 + (ca == wbeg ? 1 : 0)) * PR;
            store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
        }
        return;
    }
    if (p0 >= stop) return;
    if (is_head) {
        float *pr = a.part + ((size_t)rho * 2) * PR;
        store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
    } else if (tail_partial) {
        float *pr = a.part + ((size_t)rho * 2 + 1) * PR;
        store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
    } else {
        store_seg<LPN, J>(a, seg, l, acc, sa, sb, sl);
    }
}

// Sums the partials of the columns that were cut across ranges.  The column [ca, cb) spans ranges
// ra..rb; its units are, in order: the ranges before the first wave-aligned range, one wave-sum per
// wave lying wholly inside the column, the ranges after the last such wave.
//   * columns spanning <= 8 ranges (the vast majority: short columns straddling a boundary) are
//     summed by ONE SLOT each, units in order;
//   * longer columns by a whole workgroup: units strided over its slots (4 in flight per slot),
//     tree-summed per wave, the wave sums added in wave order.
// Both are fixed orders, so results are run-to-run identical.  The last block of the launch
// optionally finishes the step's residual statistics.
template <int LPN, int J>
struct ColumnUnits {
    static constexpr int KP = 4 * LPN * J;
    static constexpr int PR = KP + kPartPad;
    static constexpr int WS = 64 / LPN;
    static constexpr int WSPAN = WS * kRangeLen;
    int ca, ra, w_lo, nw, nl, r2, count;
    __device__ __forceinline__ ColumnUnits(int ca_, int cb) : ca(ca_) {
        ra = ca / kRangeLen;
        const int rb = (cb - 1) / kRangeLen;
        w_lo = (ca + WSPAN - 1) / WSPAN;
        const int w_hi = cb / WSPAN;                              // clean waves [w_lo, w_hi)
        nw = w_hi > w_lo ? w_hi - w_lo : 0;
        nl = nw ? w_lo * WS - ra : rb - ra + 1;                   // leading single ranges
        r2 = w_hi * WS;                                           // first trailing range
        count = nw ? nl + nw + (rb - r2 + 1) : nl;
    }
    __device__ __forceinline__ const float *row(const float *part, int t) const {
        const int rho = t < nl ? ra + t : (t < nl + nw ? (w_lo + (t - nl)) * WS : r2 + (t - nl - nw));
        return part + ((size_t)rho * 2 + (ca >= rho * kRangeLen ? 1 : 0)) * PR;
    }
};

template <int LPN, int J, bool HOT>
__global__ __launch_bounds__(kBlock) void k_fixup(BwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    constexpr int SLOTS = kBlock / LPN;
    constexpr int WS = 64 / LPN;
    if (a.red_bsum && blockIdx.x == gridDim.x - 1) {
        // the extra last block finishes the residual statistics of this step (saves a launch)
        __shared__ double sh[3][kBlock / 64];
        reduce_blocks_body(a.red_bsum, a.red_nblocks, a.red_rows, a.red_scal, a.red_acc, sh);
        return;
    }
    if (HOT) {
        // kHotT more workgroups finish the dense hot block's gradient rows
        const int hot0 = (int)gridDim.x - (a.red_bsum ? 1 : 0) - kHotT;
        if ((int)blockIdx.x >= hot0) {
            hot_reduce_body(a.hot, (int)blockIdx.x - hot0, KP);
            return;
        }
    }
    const int lane = threadIdx.x & 63;
    const int l = lane & (LPN - 1);
    float4 acc[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
    float sa = 0.f, sb = 0.f;
    const int blocks_short = (a.n_split_short + SLOTS - 1) / SLOTS;
    if ((int)blockIdx.x < blocks_short) {
        // ---- one slot per short column
        const int idx = blockIdx.x * SLOTS + threadIdx.x / LPN;
        if (idx >= a.n_split_short) return;
        const int seg = a.split_short[idx];
        const ColumnUnits<LPN, J> cu(a.cptr[seg], a.cptr[seg + 1]);
        for (int t = 0; t < cu.count; ++t) {
            const float *pr = cu.row(a.part, t);
            const float4 *p4 = reinterpret_cast<const float4 *>(pr) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) f4add(acc[jj], p4[jj * LPN]);
            sa += pr[KP];
            sb += pr[KP + 1];
        }
        store_seg<LPN, J>(a, seg, l, acc, sa, sb);
        return;
    }
    // ---- one WORKGROUP per long column: units strided over the 4 waves x WS slots, four in flight per
    // slot; slots tree-summed inside each wave, the 4 wave sums added in wave order through LDS
    const int ws = lane / LPN;
    const int wv = threadIdx.x >> 6;
    const int idx = (int)blockIdx.x - blocks_short;
    if (idx >= a.n_split) return;
    const int seg = a.split_seg[idx];
    const ColumnUnits<LPN, J> cu(a.cptr[seg], a.cptr[seg + 1]);
    constexpr int STRIDE = (kBlock / 64) * WS;
    int t = wv * WS + ws;
    for (; t + 3 * STRIDE < cu.count; t += 4 * STRIDE) {
        const float *pr[4];
        float4 v[4][J];
        float va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pr[u] = cu.row(a.part, t + u * STRIDE);
            const float4 *p4 = reinterpret_cast<const float4 *>(pr[u]) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) v[u][jj] = p4[jj * LPN];
            va[u] = pr[u][KP];
            vb[u] = pr[u][KP + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int jj = 0; jj < J; ++jj) f4add(acc[jj], v[u][jj]);
            sa += va[u];
            sb += vb[u];
        }
    }
    for (; t < cu.count; t += STRIDE) {
        const float *pr = cu.row(a.part, t);
        const float4 *p4 = reinterpret_cast<const float4 *>(pr) + l;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) f4add(acc[jj], p4[jj * LPN]);
        sa += pr[KP];
        sb += pr[KP + 1];
    }
    slots_reduce<LPN, J>(acc, sa, sb);
    __shared__ float4 wsum[kBlock / 64][J][LPN];
    __shared__ float wsc[kBlock / 64][2];
    if (ws == 0) {
#pragma unroll
        for (int jj = 0; jj < J; ++jj) wsum[wv][jj][l] = acc[jj];
        if (l == 0) { wsc[wv][0] = sa; wsc[wv][1] = sb; }
    }
    __syncthreads();
    if (wv == 0 && ws == 0) {
#pragma unroll
        for (int jj = 0; jj < J; ++jj) acc[jj] = wsum[0][jj][l];
        sa = wsc[0][0];
        sb = wsc[0][1];
#pragma unroll
        for (int w2 = 1; w2 < kBlock / 64; ++w2) {
#pragma unroll
            for (int jj = 0; jj < J; ++jj) f4add(acc[jj], wsum[w2][jj][l]);
            sa += wsc[w2][0];
            sb += wsc[w2][1];
        }
        store_seg<LPN, J>(a, seg, l, acc, sa, sb);
    }
}

// One slot per feature whose column was cut into several pieces (one per row block): the pieces lie
// next to each other in the piece buffer, in row-block order; summed serially -> fixed order.
template <int LPN, int J>
__global__ __launch_bounds__(kBlock) void k_fixup2(BwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    constexpr int SLOTS = kBlock / LPN;
    constexpr int PR = KP + kPartPad;
    const int l = threadIdx.x & (LPN - 1);
    const int m = blockIdx.x * SLOTS + threadIdx.x / LPN;
    if (m >= a.n_mp) return;
    const int p0 = a.mp_ptr[m], p1 = a.mp_ptr[m + 1];
    float4 acc[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
    float sa = 0.f, sb = 0.f;
    for (int p = p0; p < p1; ++p) {
        const float *pr = a.pieces + (size_t)p * PR;
        const float4 *p4 = reinterpret_cast<const float4 *>(pr) + l;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) f4add(acc[jj], p4[jj * LPN]);
        sa += pr[KP];
        sb += pr[KP + 1];
    }
    const int i = a.mp_feat[m];
    store_row<LPN, J>(a.GV + (size_t)i * KP, l, acc, sa, sb, a.Gw + i, a.Gb + i);
}

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


template <int LPN, int J>
hipError_t fwd_dispatch(FwdMode mode, const FwdArgs &a, hipStream_t s) {
    int64_t blocks = forward_blocks(4 * LPN * J, a.n_rows);
    dim3 g((unsigned)blocks), b(kBlock);
    int var = g_tune[kTuneFwd];
    if (a.pack_k >= 0) var = 0;              // packed rows carry w in the row: only the plain kernel handles them
    if (var == 20 && (!a.v_bytes || a.hot_T)) var = a.hot_T ? 60 : 0;   // the LDS V-tile kernel needs V to fit a 32-bit buffer view; it has no hot-block prologue
    if (var == 60 && a.wt_rows > 0) {
        const size_t lds_bytes = (size_t)a.wt_rows * sizeof(float);
        int64_t nb = forward_blocks_wt(4 * LPN * J, a.n_rows);
        dim3 gw((unsigned)nb);
#define FMHIP_WT(MODE_)                                                                                  \
    if (a.hot_T) hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE_, true>), gw, b, lds_bytes, s, a);         \
    else hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE_, false>), gw, b, lds_bytes, s, a)
        switch (mode) {
            case kFwdTrain: FMHIP_WT(kFwdTrain); break;
            case kFwdResidual: FMHIP_WT(kFwdResidual); break;
            case kFwdQ: FMHIP_WT(kFwdQ); break;
        }
#undef FMHIP_WT
        return hipGetLastError();
    }
    if (var == 20 && a.tile_rows > 0) {
        const size_t lds_bytes = (size_t)a.tile_rows * (4 * LPN * J + 1) * sizeof(float);
        dim3 gl((unsigned)forward_blocks_lds(a.n_rows)), bl(kLdsBlock);
        hipError_t e = hipSuccess;
        switch (mode) {
            case kFwdTrain:
                e = hipFuncSetAttribute((const void *)k_forward_lds<LPN, J, kFwdTrain>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e == hipSuccess) hipLaunchKernelGGL((k_forward_lds<LPN, J, kFwdTrain>), gl, bl, lds_bytes, s, a);
                break;
            case kFwdResidual:
                e = hipFuncSetAttribute((const void *)k_forward_lds<LPN, J, kFwdResidual>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e == hipSuccess) hipLaunchKernelGGL((k_forward_lds<LPN, J, kFwdResidual>), gl, bl, lds_bytes, s, a);
                break;
            case kFwdQ:
                e = hipFuncSetAttribute((const void *)k_forward_lds<LPN, J, kFwdQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                if (e == hipSuccess) hipLaunchKernelGGL((k_forward_lds<LPN, J, kFwdQ>), gl, bl, lds_bytes, s, a);
                break;
        }
        return e != hipSuccess ? e : hipGetLastError();
    }
#define FMHIP_FW(MODE_)                                                                                   \
    if (a.pack_k >= 0) {                                                                                  \
        if (a.hot_T) hipLaunchKernelGGL((k_forward<LPN, J, MODE_, true, true>), g, b, 0, s, a);           \
        else hipLaunchKernelGGL((k_forward<LPN, J, MODE_, true, false>), g, b, 0, s, a);                  \
    } else {                                                                                              \
        if (a.hot_T) hipLaunchKernelGGL((k_forward<LPN, J, MODE_, false, true>), g, b, 0, s, a);          \
        else hipLaunchKernelGGL((k_forward<LPN, J, MODE_, false, false>), g, b, 0, s, a);                 \
    }
    switch (mode) {
        case kFwdTrain: FMHIP_FW(kFwdTrain) break;
        case kFwdResidual: FMHIP_FW(kFwdResidual) break;
        case kFwdQ: FMHIP_FW(kFwdQ) break;
    }
#undef FMHIP_FW
    return hipGetLastError();
}

template <int LPN, int J>
hipError_t bwd_dispatch(const BwdArgs &a, hipStream_t s) {
    constexpr int SLOTS = kBlock / LPN;
    if (a.rho_hi <= a.rho_lo && a.hot_blocks < 1) return hipSuccess;
    int nblk = 0;
    if (a.rho_hi > a.rho_lo) nblk = (a.rho_hi - 1) / SLOTS - a.rho_lo / SLOTS + 1;
    BwdArgs a2 = a;
    a2.xcd_chunk = a.xcd_chunk > 0 ? (nblk + 7) / 8 : 0;      // blocks per XCD
    dim3 g((unsigned)((a2.xcd_chunk > 0 ? a2.xcd_chunk * 8 : nblk) + a.hot_blocks)), b(kBlock);
    // the pipelined kernel needs P to fit a 32-bit buffer view (< 4 GiB per batch); for k > 64 (J > 1)
    // its register footprint spills, so those sizes take the plain walk
    const bool pipe = J == 1 && a.p_bytes && g_tune[kTuneBwd] == 1;
#define FMHIP_BW(PACKED_, HOT_)                                                               \
    if (pipe) hipLaunchKernelGGL((k_backward_p<LPN, J, PACKED_, HOT_>), g, b, 0, s, a2);      \
    else hipLaunchKernelGGL((k_backward<LPN, J, PACKED_, HOT_>), g, b, 0, s, a2)
    if (a.pack_k >= 0) {
        if (a.hot_blocks > 0) { FMHIP_BW(true, true); } else { FMHIP_BW(true, false); }
    } else {
        if (a.hot_blocks > 0) { FMHIP_BW(false, true); } else { FMHIP_BW(false, false); }
    }
#undef FMHIP_BW
    return hipGetLastError();
}

template <int LPN, int J>
hipError_t fix_dispatch(const BwdArgs &a, hipStream_t s) {
    constexpr int SLOTS = kBlock / LPN;
    const int extra = a.red_bsum ? 1 : 0;
    const int hot = a.hot_blocks > 0 ? kHotT : 0;
    if (a.n_split < 1 && a.n_split_short < 1 && !extra && !hot) return hipSuccess;
    dim3 g((unsigned)((a.n_split_short + SLOTS - 1) / SLOTS + a.n_split + hot + extra)), b(kBlock);
    if (hot) hipLaunchKernelGGL((k_fixup<LPN, J, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_fixup<LPN, J, false>), g, b, 0, s, a);
    return hipGetLastError();
}

}  // namespace

#define FMHIP_KP_SWITCH(KPV, CALL)                       \
    switch (KPV) {                                       \
        case 32: return CALL(8, 1);                      \
        case 64: return CALL(16, 1);                     \
        case 128: return CALL(16, 2);                    \
        case 256: return CALL(16, 4);                    \
        default: return hipErrorInvalidValue;            \
    }

template <int LPN, int J>
static int wt_occupancy() {
    // the training-mode kernels decide (the scoring modes need no more registers); w-tile of 6144 floats
    int n0 = 0, n1 = 0;
    const size_t lds = 6144 * sizeof(float);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, (const void *)k_forward_wt<LPN, J, kFwdTrain, false>, kBlock, lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, (const void *)k_forward_wt<LPN, J, kFwdTrain, true>, kBlock, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    const int n = n0 < n1 ? n0 : n1;
    return n < 1 ? 1 : (n > 5 ? 5 : n);
}

int forward_wt_occupancy(int Kp) {
    static int cache[4] = {0, 0, 0, 0};
    const int idx = Kp == 32 ? 0 : Kp == 64 ? 1 : Kp == 128 ? 2 : 3;
    if (!cache[idx]) {
        switch (Kp) {
            case 32: cache[idx] = wt_occupancy<8, 1>(); break;
            case 64: cache[idx] = wt_occupancy<8, 2>(); break;
            case 128: cache[idx] = wt_occupancy<16, 2>(); break;
            default: cache[idx] = wt_occupancy<16, 4>(); break;
        }
    }
    return cache[idx];
}

// The forward walks Kp = 64 rows with 8-lane slots holding two float4 per lane (DPP broadcasts, eight rows
// per wave: 189 -> 177 us); the backward keeps 16-lane slots there (its pipelined kernel needs J = 1).
// Row layouts in memory do not depend on the lane geometry, so the two may differ.
hipError_t launch_forward(int Kp, FwdMode mode, const FwdArgs &a, hipStream_t s) {
    switch (Kp) {
        case 32: return fwd_dispatch<8, 1>(mode, a, s);
        case 64: return fwd_dispatch<8, 2>(mode, a, s);
        case 128: return fwd_dispatch<16, 2>(mode, a, s);
        case 256: return fwd_dispatch<16, 4>(mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_backward(int Kp, const BwdArgs &a, hipStream_t s) {
#define CALL(L_, J_) bwd_dispatch<L_, J_>(a, s)
    FMHIP_KP_SWITCH(Kp, CALL)
#undef CALL
}

hipError_t launch_fixup(int Kp, const BwdArgs &a, hipStream_t s) {
#define CALL(L_, J_) fix_dispatch<L_, J_>(a, s)
    FMHIP_KP_SWITCH(Kp, CALL)
#undef CALL
}

template <int LPN, int J>
hipError_t fix2_dispatch(const BwdArgs &a, hipStream_t s) {
    constexpr int SLOTS = kBlock / LPN;
    if (a.n_mp < 1) return hipSuccess;
    dim3 g((unsigned)((a.n_mp + SLOTS - 1) / SLOTS)), b(kBlock);
    hipLaunchKernelGGL((k_fixup2<LPN, J>), g, b, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fixup2(int Kp, const BwdArgs &a, hipStream_t s) {
#define CALL(L_, J_) fix2_dispatch<L_, J_>(a, s)
    FMHIP_KP_SWITCH(Kp, CALL)
#undef CALL
}

int hot_blocks(int Kp, int64_t n_rows) {
    (void)Kp;
    int64_t b = (n_rows + 63) / 64;       // at least 16 rows per wave
    if (b > 256) b = 256;                 // one hot workgroup per CU, next to the column walkers
    return b < 1 ? 1 : (int)b;
}

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

hipError_t launch_reduce_blocks(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal, double *acc,
                                hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_blocks, dim3(1), dim3(kBlock), 0, s, bsum, nblocks, n_rows, scal, acc);
    return hipGetLastError();
}

}  // namespace fmhip
