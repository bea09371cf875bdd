// fm_forward.hip — the forward of the FM step: row walks over the CSR stream (k_forward, k_forward_wt,
// k_forward_lds), the dense hot block's prologue, the per-block statistics reduction, and the launch
// geometry that goes with them.  Lane geometry and formulas: fm_device.h.
#include "fm_device.h"

#ifndef FMHIP_FWD_LDS_SHARE
#define FMHIP_FWD_LDS_SHARE 1
#endif
// timing-only ablations (results wrong by construction; tools/build_variant.sh with EXTRA_FLAGS)
#ifndef FMHIP_FWD_AHEAD
#define FMHIP_FWD_AHEAD 1
#endif
#ifndef FMHIP_FWD_W_SPLIT
#define FMHIP_FWD_W_SPLIT 1
#endif
#ifndef FMHIP_EXP_NO_STREAM
#define FMHIP_EXP_NO_STREAM 0
#endif
#ifndef FMHIP_EXP_NO_MATH
#define FMHIP_EXP_NO_MATH 0
#endif
#ifndef FMHIP_EXP_NO_W
#define FMHIP_EXP_NO_W 0
#endif
#ifndef FMHIP_EXP_NO_GATHER
#define FMHIP_EXP_NO_GATHER 0
#endif
#ifndef FMHIP_EXP_L1_GATHER
#define FMHIP_EXP_L1_GATHER 0
#endif
#ifndef FMHIP_EXP_NO_HOT
#define FMHIP_EXP_NO_HOT 0
#endif
// ... none of which may reach a library anyone trains with: a build that sets one must say so (FMHIP_ABLATION_BUILD, which
// tools/build_variant.sh passes for A/B variants), and fmhip_ablation_mask() reports it at run time (tests/test_host_cpu.py
// asserts the shipped library's mask is 0)
#define FMHIP_FWD_ABLATIONS ((FMHIP_EXP_NO_STREAM ? 1 : 0) | (FMHIP_EXP_NO_MATH ? 2 : 0) | (FMHIP_EXP_NO_W ? 4 : 0) | (FMHIP_EXP_NO_GATHER ? 8 : 0) | \
                             (FMHIP_EXP_L1_GATHER ? 16 : 0) | (FMHIP_EXP_NO_HOT ? 32 : 0))
#if FMHIP_FWD_ABLATIONS && !defined(FMHIP_ABLATION_BUILD)
#error "a result-changing FMHIP_EXP_* ablation is set without FMHIP_ABLATION_BUILD: timing-only variants are built by tools/build_variant.sh"
#endif

namespace fmhip {
int forward_ablations() { return FMHIP_FWD_ABLATIONS; }
}

namespace fmhip {

std::atomic<int> g_tune[kTuneCount] = {60, 1, 0, 0, 2, 1, 0, 1, 0, 1, 0, 1, 4};   // forward: w-tile kernel; backward: pipelined; tile rows: auto; row blocks: off; backward placement: band-affine; hot block: on; row order: on; forced flat loads: off; lazy decay: on; fused update: off; merged finish: on; hot pages: 4 of up to kHotPages (pages 5-8 measured: no gain, profiles/r03_experiments.md)

int padded_factors(int k) {
    int kp = 32;   // a row is at least one 128-B line: the cost of a gather is per line, not per byte
    while (kp < k) kp <<= 1;
    return kp;
}

int forward_wt_occupancy(int Kp);   // workgroups of k_forward_wt one CU holds (defined next to the kernel)

int forward_blocks_wt(int Kp, int64_t n_rows, int occ_cap) {
    // persistent: as many workgroups as the chip holds at once (24 KiB of LDS each; the register count
    // decides: 5 per CU at Kp = 32, fewer for wider rows), rows grid-strided
    const int64_t need = forward_blocks(Kp, n_rows);
    int occ = forward_wt_occupancy(Kp);
    if (occ_cap > 0 && occ_cap < occ) occ = occ_cap;   // experiment knob (tuning key 6): fewer resident workgroups
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

// q += v*x and s += (v*x)^2 for one float4 of one entry, spelled as the six packed-fp32 operations it should be
// (v_pk_mul_f32 for the rounded product, v_pk_fma_f32 into q and into s): left to the vectoriser the same source comes out
// packed or scalar depending on the surrounding register pressure.  A single-nonzero row still has q = round(v*x) and
// s = round(q^2), so its interaction is exactly 0 (quirk Q6).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void acc_entry(float4 &q, float4 &s, const float4 t, const float x) {
    const v2f xs = {x, x};
    const v2f tlo = {t.x, t.y}, thi = {t.z, t.w};
    v2f qlo = {q.x, q.y}, qhi = {q.z, q.w}, slo = {s.x, s.y}, shi = {s.z, s.w};
    const v2f plo = tlo * xs, phi = thi * xs;
    qlo = __builtin_elementwise_fma(tlo, xs, qlo);
    qhi = __builtin_elementwise_fma(thi, xs, qhi);
    slo = __builtin_elementwise_fma(plo, plo, slo);
    shi = __builtin_elementwise_fma(phi, phi, shi);
    q = make_float4(qlo.x, qlo.y, qhi.x, qhi.y);
    s = make_float4(slo.x, slo.y, shi.x, shi.y);
}

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
            const float4 vrow = reinterpret_cast<const float4 *>(vh + h * KP)[jj * LPN + l];
            if (MASKED) {
                float4 tv = f4mul(vrow, x);
                if (!live) tv = f4zero();
                f4add(q[jj], tv);
                f4sqacc(s[jj], tv);
            } else {
                acc_entry(q[jj], s[jj], vrow, x);
            }
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

// The slot's LPN entries (c, x: one per lane) are handed to every lane of the slot through LDS: each lane writes its
// entry into the slot's 2 x LPN words of the workgroup's staging area (slot_publish) and reads the others back four at a
// time with ds_read_b128 (the lanes of a slot read the same address: a broadcast) — 1 + LPN/2 LDS instructions per step
// instead of ~6 vector-ALU moves per entry (two DPP moves per broadcast value plus the copies DPP's tied operand
// needs), on a kernel whose vector ALU is the busiest unit (profiles/r03_experiments.md §12).  LDS operations of one
// wave complete in order and the lanes of a slot always run together, so no barrier is needed — only the compiler
// kept from reordering the accesses.
template <int LPN>
__device__ __forceinline__ const int *slot_publish(int *stage, int c, float x, int l) {
    if (!FMHIP_FWD_LDS_SHARE) return nullptr;
    int *sc = stage + (threadIdx.x & ~(LPN - 1)) * 2;      // this slot's [LPN] ids, then its [LPN] values
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous step's reads come first
    sc[l] = c;
    sc[LPN + l] = __float_as_int(x);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return sc;
}
template <int LPN, int CH>
__device__ __forceinline__ void slot_entries(const int *sc, int c, int c0, int (&out)[CH]) {
    static_assert(CH % 4 == 0, "four staged words per LDS read");
    if (FMHIP_FWD_LDS_SHARE) {
#pragma unroll
        for (int j = 0; j < CH; j += 4) {
            const int4 t = *reinterpret_cast<const int4 *>(sc + c0 + j);
            out[j] = t.x; out[j + 1] = t.y; out[j + 2] = t.z; out[j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < CH; ++j) out[j] = slot_bcast<LPN>(c, c0 + j);
    }
}

// One step of a row walk: the V rows of the slot's LPN entries are gathered CH at a time and accumulated in stored
// order (q_f: FMModel.scala:59, sum_sqr_f: :60).  The entries' values are fetched from the staging area only once the
// gathers are on their way (their registers are not live while the addresses are).
// MASKED = the row's last, partial step (entries >= cnt are dead); full steps carry no per-entry
// compare/select — the vector ALU, not the memory path, is what the forward saturates
// (profiles/r01_experiments.md, section 23).
// BUF: V fits a 32-bit buffer view (< 4 GiB): a row's address is ONE 32-bit shift-add per entry (the
// descriptor lives in scalar registers) instead of a 64-bit multiply-add pair, and the dead entries of a
// partial step fetch nothing (out-of-range offset).  Wider tables take flat 64-bit addresses.
template <int LPN, int J, int CH, bool MASKED, bool BUF>
__device__ __forceinline__ void fwd_step(const float *V, __amdgpu_buffer_rsrc_t vr, const int *sc, int c, float x, int cnt, int l, float4 (&q)[J],
                                         float4 (&s)[J]) {
    constexpr int KP = 4 * LPN * J;
#pragma unroll
    for (int c0 = 0; c0 < LPN; c0 += CH) {
        float4 t[CH][J];
        {
            int cj[CH];
            slot_entries<LPN, CH>(sc, c, c0, cj);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (FMHIP_EXP_NO_GATHER) {
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) t[j][jj] = make_float4(__int_as_float(cj[j]), 1.f, 2.f, (float)l);
                } else if (BUF) {
                    const uint32_t off = (uint32_t)(FMHIP_EXP_L1_GATHER ? (cj[j] & 63) : cj[j]) * (KP * 4u) + (uint32_t)l * 16u;
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) t[j][jj] = buf_load4(vr, (MASKED && c0 + j >= cnt) ? kOob : off + jj * LPN * 16u);
                } else {
                    const float4 *vp = reinterpret_cast<const float4 *>(V + (size_t)(uint32_t)cj[j] * KP) + l;
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) t[j][jj] = vp[jj * LPN];
                }
            }
        }
        if (FMHIP_FWD_LDS_SHARE) __builtin_amdgcn_sched_barrier(0);
        int xi[CH];
        slot_entries<LPN, CH>(sc ? sc + LPN : nullptr, __float_as_int(x), c0, xi);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool live = c0 + j < cnt;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                if (MASKED) {
                    float4 tv = f4mul(t[j][jj], __int_as_float(xi[j]));
                    if (!live) tv = f4zero();
                    f4add(q[jj], tv);
                    f4sqacc(s[jj], tv);
                } else if (FMHIP_EXP_NO_MATH) {
                    q[jj].x += t[j][jj].x * __int_as_float(xi[j]);
                } else {
                    acc_entry(q[jj], s[jj], t[j][jj], __int_as_float(xi[j]));
                }
            }
        }
    }
}

// What every forward kernel does once a row's sums are complete (FMModel.scala:48-55): the prediction, the
// residual (ALS.scala:143), the row of P for the backward (or q itself, ALS.scala:146-150) and this
// thread's share of the residual statistics.  q, s and lin are in units of the STORED tables; the scales
// of a lazily decayed model (FwdArgs.sv / .sw; both 1 otherwise, and multiplying by 1 is exact) enter here:
// the interaction is homogeneous of degree 2 in V, so a single-nonzero row still gives exactly 0 (quirk Q6).
template <int LPN, int J, int MODE, bool PACKED>
__device__ __forceinline__ void row_finish(const FwdArgs &a, int r, int l, float4 (&q)[J], float4 (&s)[J], float lin, float w0,
                                           float &st1, float &st2, float &stbad) {
    constexpr int KP = 4 * LPN * J;
    // Packed rows (k < Kp): slot k of every V row holds the feature's linear weight w_i, so q_k
    // accumulated sum w_i x_i — the linear term — and there was no separate w gather; slot k of the P row
    // carries e to the backward the same way.
    const int kl = PACKED ? (a.pack_k >> 2) & (LPN - 1) : 0, kj = PACKED ? (a.pack_k >> 2) / LPN : 0, kc = a.pack_k & 3;
    if (MODE == kFwdPartA) {
        // pass A of a two-pass forward: the row's raw sums go out as they are — q (slot k of a packed row: the linear term so
        // far) into the P row, sum_f s_f and the lanes' linear terms, each summed over the slot, into part_sl
        float sp = 0.f, lp = lin;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            float4 sj = s[jj];
            if (PACKED && jj == kj && l == kl) f4set(sj, kc, 0.f);        // slot k is not a factor
            sp += (sj.x + sj.y) + (sj.z + sj.w);
        }
#pragma unroll
        for (int m = LPN >> 1; m >= 1; m >>= 1) { sp += __shfl_xor(sp, m, LPN); lp += __shfl_xor(lp, m, LPN); }
        float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = q[jj];
        if (l == 0) { a.part_sl[2 * (size_t)r] = sp; a.part_sl[2 * (size_t)r + 1] = lp; }
        return;
    }
    float lin_all = 0.f;   // packed: the complete linear term, identical in every lane of the slot
    if (PACKED) {
        float lk = 0.f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj)
            if (jj == kj) {
                lk = f4pick(q[jj], kc);
                if (l == kl) { f4set(q[jj], kc, 0.f); f4set(s[jj], kc, 0.f); }   // slot k is not a factor
            }
        lin_all = a.sw * __shfl(lk, kl, LPN);
    }
    float u = 0.f;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) u += f4sqminus(q[jj], s[jj]);
    float tot = fmaf(0.5f * a.sv * a.sv, u, a.sw * lin);
    // pass B: q already holds pass A's share (the row started from it); its sum_f s_f and linear term enter once, through lane 0
    if (MODE == kFwdPartB && l == 0) tot += fmaf(-0.5f * a.sv * a.sv, a.part_sl[2 * (size_t)r], a.sw * a.part_sl[2 * (size_t)r + 1]);
#pragma unroll
    for (int m = LPN >> 1; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, LPN);
    const float yhat = w0 + (tot + lin_all);
    const float e = yhat - a.y[a.row0 + r];
    if (MODE == kFwdTrain || MODE == kFwdPartB) {
        float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
        const float es = e * a.sv;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            float4 o = f4mul(q[jj], es);
            if (PACKED && jj == kj && l == kl) f4set(o, kc, e);   // slot k of the P row carries e
            if (!PACKED && kEInP && jj == 0 && l < 8) o = embed_bits4(o, __float_as_uint(e) >> (4 * l));   // no spare slot: e rides in the LSBs (fm_device.h)
            p_store(pr + jj * LPN, o);
        }
    } else if (MODE == kFwdQ) {
        float4 *pr = reinterpret_cast<float4 *>(a.P + (size_t)r * KP) + l;
#pragma unroll
        for (int jj = 0; jj < J; ++jj) pr[jj * LPN] = f4mul(q[jj], a.sv);
    }
    if (l == 0) {
        if (a.e) a.e[r] = e;
        if (a.yhat) a.yhat[r] = yhat;
        st1 += e;
        st2 = fmaf(e, e, st2);
        if (!isfinite(e)) stbad += 1.f;
    }
}

// block partial of the residual statistics (fixed order; the fixup launch / k_reduce_blocks finishes the sum)
template <int NT>
__device__ __forceinline__ void block_stats(double *bsum, float st1, float st2, float stbad) {
    if (!bsum) return;
    __shared__ double sh[3][NT / 64];
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
        for (int i = 0; i < NT / 64; ++i) { t1 += sh[0][i]; t2 += sh[1][i]; tb += sh[2][i]; }
        double *o = bsum + (size_t)blockIdx.x * 4;
        o[0] = t1; o[1] = t2; o[2] = tb; o[3] = 0.0;
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
        row_finish<LPN, J, MODE, false>(a, r, l, q, s, lin, w0, st1, st2, stbad);
    }
    block_stats<kLdsBlock>(a.bsum, st1, st2, stbad);
}

// The linear weight of feature c: from the LDS tile when c < T, else from the table.  Written as one LDS read and one
// buffer load whose tile lanes are out of range (they fetch nothing) rather than a select between two addresses, which
// the compiler turns into a flat load that sends all 64 lanes through the texture addresser.
template <bool WT, bool BUF>
__device__ __forceinline__ float w_lookup(const FwdArgs &a, __amdgpu_buffer_rsrc_t wr, const float *wt, int T, int c) {
    // c == -1 (the dead entry of a pipelined step) yields 0 without a branch: every load below is unconditional
    if (BUF && FMHIP_FWD_W_SPLIT) {
        const bool in_tile = WT && (uint32_t)c < (uint32_t)T;
        const float wg = buf_load1(wr, in_tile ? kOob : (uint32_t)c * 4u);     // id -1: past the end of w, returns 0
        if (!WT) return wg;
        const float wl = wt[in_tile ? c : 0];
        return in_tile ? wl : wg;
    }
    const int cc = c < 0 ? 0 : c;
    const float v = (WT && cc < T) ? wt[cc] : a.w[cc];
    return c < 0 ? 0.f : v;
}

// The row walk of k_forward / k_forward_wt.  Rows are taken in the dataset's length-sorted order when one
// is given (longest first): the slots of a wave then walk rows of (nearly) equal length, so no lane idles
// while a neighbour finishes a longer row.  WT: the linear weights of the wt_rows lowest (= hottest, for
// frequency-ranked ids) feature ids are read from the LDS tile `wt` — a 64-lane gather of w costs ~2 TA
// cycles per distinct line touched, as much per nonzero as the whole 128-B V-row gather
// (profiles/r01_experiments.md §13); lanes whose id is in the tile drop out of the global gather.
template <int LPN, int J, int MODE, bool PACKED, bool HOT, bool WT, bool BUF>
__device__ __forceinline__ void forward_rows(const FwdArgs &a, const float *wt, const float *vh, const float *wh, bool hot_plain) {
    __shared__ __attribute__((aligned(16))) int stage[FMHIP_FWD_LDS_SHARE ? 2 * kBlock : 4];   // slot_share
    constexpr int SLOTS = kBlock / LPN;
    constexpr int KPW = 4 * LPN * J;
    constexpr int CH = (LPN * J > 16) ? (16 / J) : LPN;  // entries whose V rows are in flight together
    const int l = threadIdx.x & (LPN - 1);
    const int slot = threadIdx.x / LPN;
    const float w0 = *a.w0;
    const int T = WT ? a.wt_rows : 0;
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(a.V, a.v_bytes);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, a.v_bytes / KPW);   // (n+1) floats, when V fits a buffer view
    const int32_t *colb = a.col + a.nz0;
    const float *valb = a.val + a.nz0;
    float st1 = 0.f, st2 = 0.f, stbad = 0.f;   // this thread's share of {sum e, sum e^2, nonfinite}
    if (BUF && FMHIP_FWD_AHEAD) {
        // Through a buffer view one loop serves full and partial steps — a dead entry is id -1 with value 0: its row offset
        // lies past the end of V (a multiple of the row size below 2^32), the gather fetches nothing and returns 0, and 0 * 0
        // adds exactly nothing — and the entries of the row's NEXT step are requested before this step's gathers, so the
        // stream's latency is not paid between steps.  What was measured around it (profiles/r03_experiments.md §12): looking
        // further ahead (the next row's offsets, first entries, dense-block values) is slower at C3 (+5 us), and so is
        // requesting the row's first entries before the dense-block prologue; packed rows (no w lookup) prefer the requests
        // unconditional through opaque addresses (C2 forward 108 -> 97 us), rows with a w lookup the plain conditional form.
        for (int ri = blockIdx.x * SLOTS + slot; ri < a.n_rows; ri += gridDim.x * SLOTS) {
            const int r = a.order ? a.order[ri] : ri;
            float4 xh = f4zero();
            if (HOT) xh = hot_load(a, r, l);
            const uint32_t p0 = (uint32_t)((MODE == kFwdPartB ? a.row_split[a.row0 + r] : a.row_ptr[a.row0 + r]) - a.nz0),
                           p1 = (uint32_t)((MODE == kFwdPartA ? a.row_split[a.row0 + r] : a.row_ptr[a.row0 + r + 1]) - a.nz0);
            float4 q[J], s[J];
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                q[jj] = MODE == kFwdPartB ? reinterpret_cast<const float4 *>(a.P + (size_t)r * KPW)[jj * LPN + l] : f4zero();
                s[jj] = f4zero();
            }
            float lin = 0.f;
            if (HOT && !FMHIP_EXP_NO_HOT) {
                if (hot_plain) hot_prologue<LPN, J, !PACKED, false>(xh, vh, wh, l, q, s, lin);
                else hot_prologue<LPN, J, !PACKED, true>(xh, vh, wh, l, q, s, lin);
            }
            int c_n = -1;
            float x_n = 0.f;
            bool live_n = p0 + l < p1;   // c_n / x_n hold a loaded entry (else a dead lane's filler)
            if (live_n) { c_n = stream_load(colb + (p0 + l)); x_n = stream_load(valb + (p0 + l)); }
            for (uint32_t base = p0; base < p1; base += LPN) {
                const int c = live_n ? c_n : -1;
                const float x = live_n ? x_n : 0.f;
                const uint32_t pos = base + LPN + l;
                if (PACKED) {
                    // unconditional loads (a dead lane reads a harmless word), the dead-lane select left to the consumer
                    live_n = pos < p1;
                    typedef __attribute__((address_space(1))) const int32_t gint;
                    typedef __attribute__((address_space(1))) const float gflt;
                    gint *cp = (gint *)(live_n ? colb + pos : reinterpret_cast<const int32_t *>(a.w0));
                    gflt *xp = (gflt *)(live_n ? valb + pos : a.w0);
                    asm volatile("" : "+v"(cp), "+v"(xp));   // opaque: or the select + load is turned back into a branch
                    c_n = *cp;
                    x_n = *xp;
                } else {
                    live_n = true; c_n = -1; x_n = 0.f;
                    if (pos < p1) { c_n = stream_load(colb + pos); x_n = stream_load(valb + pos); }
                }
                float wv = 0.f;
                if (!PACKED && !FMHIP_EXP_NO_W && c >= 0) wv = w_lookup<WT, BUF>(a, wr, wt, T, c);
                fwd_step<LPN, J, CH, false, BUF>(a.V, vr, slot_publish<LPN>(stage, c, x, l), c, x, LPN, l, q, s);
                if (!PACKED) lin = fmaf(wv, x, lin);
            }
            row_finish<LPN, J, MODE, PACKED>(a, r, l, q, s, lin, w0, st1, st2, stbad);
        }
        block_stats<kBlock>(a.bsum, st1, st2, stbad);
        return;
    }
    for (int ri = blockIdx.x * SLOTS + slot; ri < a.n_rows; ri += gridDim.x * SLOTS) {
        const int r = a.order ? a.order[ri] : ri;
        float4 xh = f4zero();
        if (HOT) xh = hot_load(a, r, l);
        // entry positions relative to the batch's first entry: 32-bit walk state (a batch holds < 2^31 entries)
        const uint32_t p0 = (uint32_t)((MODE == kFwdPartB ? a.row_split[a.row0 + r] : a.row_ptr[a.row0 + r]) - a.nz0),
                       p1 = (uint32_t)((MODE == kFwdPartA ? a.row_split[a.row0 + r] : a.row_ptr[a.row0 + r + 1]) - a.nz0);
        float4 q[J], s[J];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            q[jj] = MODE == kFwdPartB ? reinterpret_cast<const float4 *>(a.P + (size_t)r * KPW)[jj * LPN + l] : f4zero();
            s[jj] = f4zero();
        }
        float lin = 0.f;
        if (HOT && !FMHIP_EXP_NO_HOT) {
            if (hot_plain) hot_prologue<LPN, J, !PACKED, false>(xh, vh, wh, l, q, s, lin);
            else hot_prologue<LPN, J, !PACKED, true>(xh, vh, wh, l, q, s, lin);
        }
        uint32_t base = p0;
        for (; base + LPN <= p1; base += LPN) {        // full steps
            const int c = FMHIP_EXP_NO_STREAM ? (int)((base + l) & 1023u) : stream_load(colb + (base + l));
            const float x = FMHIP_EXP_NO_STREAM ? 1.f : stream_load(valb + (base + l));
            float wv = 0.f;
            if (!PACKED && !FMHIP_EXP_NO_W) wv = w_lookup<WT, BUF>(a, wr, wt, T, c);
            fwd_step<LPN, J, CH, false, BUF>(a.V, vr, slot_publish<LPN>(stage, c, x, l), c, x, LPN, l, q, s);
            if (!PACKED) lin = fmaf(wv, x, lin);       // consumed after the gathers are on their way
        }
        if (base < p1) {                               // the row's last, partial step
            // Through a buffer view a dead entry is id -1 with value 0: its row offset lies past the end of V (a multiple
            // of the row size below 2^32), so the gather fetches nothing and returns 0, and 0 * 0 adds exactly nothing —
            // the step needs no per-entry mask.  Flat addresses keep the masked step (a dead entry reads row 0).
            const uint32_t p = base + l;
            int c = BUF ? -1 : 0;
            float x = 0.f, wv = 0.f;
            if (p < p1) {
                c = stream_load(colb + p);
                x = stream_load(valb + p);
                if (!PACKED && !FMHIP_EXP_NO_W) wv = w_lookup<WT, BUF>(a, wr, wt, T, c);
            }
            fwd_step<LPN, J, CH, !BUF, BUF>(a.V, vr, slot_publish<LPN>(stage, c, x, l), c, x, (int)(p1 - base), l, q, s);
            if (!PACKED) lin = fmaf(wv, x, lin);
        }
        row_finish<LPN, J, MODE, PACKED>(a, r, l, q, s, lin, w0, st1, st2, stbad);
    }
    block_stats<kBlock>(a.bsum, st1, st2, stbad);
}

// (Second launch bound, Kp = 32 with the hot-block prologue: left to itself the compiler takes 100 registers — four waves per
// SIMD; budgeted for five it needs 94 and C2's forward, k = 16 in packed rows, goes from 98.3 to 93.0 us with the same bits.
// Without the prologue the kernel sits at five already and the bound only perturbs it: r04_experiments.md section 20.)
#ifndef FMHIP_FWD_PLAIN_WGS
#define FMHIP_FWD_PLAIN_WGS 5
#endif
template <int LPN, int J, int MODE, bool PACKED, bool HOT, bool BUF>
__global__ __launch_bounds__(kBlock, (LPN * J <= 8 && HOT ? FMHIP_FWD_PLAIN_WGS : 1)) void k_forward(FwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    __shared__ __attribute__((aligned(16))) float vh[HOT ? kHotT * KP : 4];
    __shared__ float wh[HOT ? kHotT : 1];
    bool hot_plain = false;
    if (HOT) hot_plain = hot_stage<KP>(a, vh, wh);
    forward_rows<LPN, J, MODE, PACKED, HOT, false, BUF>(a, nullptr, vh, wh, hot_plain);
}

// k_forward with an LDS-resident tile of the hot linear weights.  (Second launch bound = waves per SIMD:
// the persistent grid of forward_blocks_wt is sized for 5, and a register count that admits only 4 would
// run it in two rounds.)
#ifndef FMHIP_FWD_WGS
#define FMHIP_FWD_WGS 5       // waves per SIMD the compiler budgets registers for (Kp = 32); 6 was measured: r04_experiments.md section 18
#endif
template <int LPN, int J, int MODE, bool HOT, bool BUF>
__global__ __launch_bounds__(kBlock, (LPN * J <= 8 ? FMHIP_FWD_WGS : 1)) void k_forward_wt(FwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    extern __shared__ __attribute__((aligned(16))) float wt[];
    __shared__ __attribute__((aligned(16))) float vh[HOT ? kHotT * KP : 4];
    __shared__ float wh[HOT ? kHotT : 1];
    for (int i = threadIdx.x; i < a.wt_rows; i += kBlock) wt[i] = a.w[i];
    bool hot_plain = false;
    if (HOT) hot_plain = hot_stage<KP>(a, vh, wh);
    else __syncthreads();
    forward_rows<LPN, J, MODE, false, HOT, true, BUF>(a, wt, vh, wh, hot_plain);
}

// Pass B of the two-pass forward (kFwdPartB): a row has a handful of cold entries — the features at or above the top cut, a few per
// cent of the nonzeros — so the generic walk (eight entries per step, the next step's entries requested ahead, 86 registers, five
// waves per SIMD) is the wrong shape.  Here an 8-lane slot takes a row and its entries ONE AT A TIME (the slot's lanes read the
// entry's id and value from one address — a broadcast — and 16 B of the V row each); the body is lean enough for more waves, and
// what pass B waits for is the chain of dependent loads per row (extent, entry, V row; the row's partial q from P), which more rows in
// flight hide.  Same arithmetic per entry (acc_entry), same finish (row_finish); the linear term's few addends are summed by one lane.
template <int LPN, int J, bool PACKED>
__global__ __launch_bounds__(kBlock) void k_forward_pass_b(FwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    constexpr int SLOTS = kBlock / LPN;
    const int l = threadIdx.x & (LPN - 1);
    const int slot = threadIdx.x / LPN;
    const float w0 = *a.w0;
    const int32_t *colb = a.col + a.nz0;
    const float *valb = a.val + a.nz0;
    float st1 = 0.f, st2 = 0.f, stbad = 0.f;
    for (int ri = blockIdx.x * SLOTS + slot; ri < a.n_rows; ri += gridDim.x * SLOTS) {
        const int r = a.order ? a.order[ri] : ri;
        const uint32_t p0 = (uint32_t)(a.row_split[a.row0 + r] - a.nz0), p1 = (uint32_t)(a.row_ptr[a.row0 + r + 1] - a.nz0);
        float4 q[J], s[J];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            q[jj] = reinterpret_cast<const float4 *>(a.P + (size_t)r * KP)[jj * LPN + l];
            s[jj] = f4zero();
        }
        float lin = 0.f;
        for (uint32_t p = p0; p < p1; ++p) {
            const int c = colb[p];
            const float x = valb[p];
            const float4 *vp = reinterpret_cast<const float4 *>(a.V + (size_t)(uint32_t)c * KP) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) acc_entry(q[jj], s[jj], vp[jj * LPN], x);
            if (!PACKED && l == 0) lin = fmaf(a.w[c], x, lin);      // (row_finish sums the lanes' linear terms)
        }
        row_finish<LPN, J, kFwdPartB, PACKED>(a, r, l, q, s, lin, w0, st1, st2, stbad);
    }
    block_stats<kBlock>(a.bsum, st1, st2, stbad);
}

__global__ __launch_bounds__(kBlock) void k_reduce_blocks(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal,
                                                         double *acc) {
    __shared__ double sh[3][kBlock / 64];
    reduce_blocks_body(bsum, nblocks, n_rows, scal, acc, sh);
}

// dense V *= sv, w *= sw (lazily decayed tables back to scale 1)
template <int KP>
__global__ __launch_bounds__(kBlock) void k_rescale(float *V, float *w, int64_t n1, int32_t pack_k, float sv, float sw) {
    constexpr int LPR = KP / 4;
    const int64_t total = n1 * LPR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t i = idx / LPR;
        const int c = (int)(idx % LPR);
        float4 *p = reinterpret_cast<float4 *>(V) + idx;
        float4 v = *p;
        const float keep = f4pick(v, pack_k & 3);
        v = f4mul(v, sv);
        if (pack_k >= 0 && c == (pack_k >> 2)) f4set(v, pack_k & 3, keep * sw);
        *p = v;
        if (c == 0) w[i] *= sw;
    }
}

// The one place that decides which forward kernel runs and how large its grid is (the number of
// per-block statistic partials the launch writes): packed rows carry w in the row, so only the plain
// kernel handles them; the LDS V-tile kernel needs V to fit a 32-bit buffer view and has no hot-block
// prologue.
struct FwdPlan { int var; int blocks; };
template <int LPN, int J>
FwdPlan fwd_plan(const FwdArgs &a) {
    constexpr int KP = 4 * LPN * J;
    int var = a.variant;
    if (a.pack_k >= 0) var = 0;
    if (var == 20 && (!a.v_bytes || a.hot_T || a.tile_rows < 1)) var = a.hot_T ? 60 : 0;
    if (var == 60 && a.wt_rows < 1) var = 0;
    if (var != 20 && var != 60) var = 0;
    const int blocks = var == 60 ? forward_blocks_wt(KP, a.n_rows, a.occ_cap) : var == 20 ? forward_blocks_lds(a.n_rows) : forward_blocks(KP, a.n_rows);
    return {var, blocks};
}

template <int LPN, int J, int MODE>
hipError_t fwd_launch(const FwdArgs &a, hipStream_t s, int *n_partials) {
    const FwdPlan pl = fwd_plan<LPN, J>(a);
    if (n_partials) *n_partials = pl.blocks;
    const dim3 g((unsigned)pl.blocks), b(kBlock);
    const bool buf = a.v_bytes != 0;
    if (pl.var == 60) {
        const size_t lds_bytes = (size_t)a.wt_rows * sizeof(float);
#define FMHIP_WT(HOT_, BUF_) hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE, HOT_, BUF_>), g, b, lds_bytes, s, a)
        if (a.hot_T) { if (buf) FMHIP_WT(true, true); else FMHIP_WT(true, false); }
        else { if (buf) FMHIP_WT(false, true); else FMHIP_WT(false, false); }
#undef FMHIP_WT
        return hipGetLastError();
    }
    if (pl.var == 20) {
        const size_t lds_bytes = (size_t)a.tile_rows * (4 * LPN * J + 1) * sizeof(float);
        hipError_t e = hipFuncSetAttribute((const void *)k_forward_lds<LPN, J, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_forward_lds<LPN, J, MODE>), g, dim3(kLdsBlock), lds_bytes, s, a);
        return hipGetLastError();
    }
#define FMHIP_FW(PACKED_, HOT_)                                                                   \
    do {                                                                                          \
        if (buf) hipLaunchKernelGGL((k_forward<LPN, J, MODE, PACKED_, HOT_, true>), g, b, 0, s, a);  \
        else hipLaunchKernelGGL((k_forward<LPN, J, MODE, PACKED_, HOT_, false>), g, b, 0, s, a);     \
    } while (0)
    if (a.pack_k >= 0) { if (a.hot_T) FMHIP_FW(true, true); else FMHIP_FW(true, false); }
    else { if (a.hot_T) FMHIP_FW(false, true); else FMHIP_FW(false, false); }
#undef FMHIP_FW
    return hipGetLastError();
}

// The two passes of the pipelined schedule's forward (Kp <= 64).  Pass A is the training forward's own choice of kernel (w-tile
// or plain, with the hot-block prologue) minus the LDS V-tile variant; pass B — the few cold entries of a row — is
// k_forward_pass_b, without a prologue: the dense hot block's features are the most frequent ones and, with ids ranked by
// frequency, lie below any cut.  When they do NOT (hashed or field-ordered ids: a feature in a tenth of the rows may carry
// the highest id) the prologue would read parameter rows at or above the cut in pass A, one update behind — FwdArgs::hot_in_b
// moves it into pass B, which then takes the generic row walk with the prologue.
template <int LPN, int J, int MODE>
hipError_t fwd_launch_pass(const FwdArgs &a0, hipStream_t s, int *n_partials) {
    if constexpr (LPN * J > 16) {
        return hipErrorInvalidValue;
    } else {
        FwdArgs a = a0;
        const bool hot_here = a.hot_T && (MODE == kFwdPartB) == (a.hot_in_b != 0);      // the pass that runs the prologue
        if (!hot_here) a.hot_T = 0;
        if (MODE == kFwdPartB) a.variant = 0;
        if (a.variant == 20) a.variant = a.hot_T ? 60 : 0;
        const FwdPlan pl = fwd_plan<LPN, J>(a);
        if (n_partials) *n_partials = pl.blocks;
        const dim3 g((unsigned)pl.blocks), b(kBlock);
        const bool buf = a.v_bytes != 0;
        if constexpr (MODE == kFwdPartB) {
            if (!a.hot_T) {
                // rows in STORED order: the length-sorted order balances the slots of a wave over rows of 20..60 entries; pass
                // B's rows hold a handful, and in stored order a wave's eight rows read their extents, partial sums, labels and
                // P rows from neighbouring addresses and write e and P there (forward 378 -> 373.5 us per step at C4 / 625k rows)
                a.order = nullptr;
                // (the generic walk in this mode, A/B on one box: forward 409 -> 401 us per step, step 1.066 -> 1.058 ms)
                if (a.pack_k >= 0) hipLaunchKernelGGL((k_forward_pass_b<LPN, J, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL((k_forward_pass_b<LPN, J, false>), g, b, 0, s, a);
                return hipGetLastError();
            }
        }
        if constexpr (MODE == kFwdPartA) {
            if (pl.var == 60) {
                const size_t lds_bytes = (size_t)a.wt_rows * sizeof(float);
                if (a.hot_T) {
                    if (buf) hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE, true, true>), g, b, lds_bytes, s, a);
                    else hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE, true, false>), g, b, lds_bytes, s, a);
                } else {
                    if (buf) hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE, false, true>), g, b, lds_bytes, s, a);
                    else hipLaunchKernelGGL((k_forward_wt<LPN, J, MODE, false, false>), g, b, lds_bytes, s, a);
                }
                return hipGetLastError();
            }
        }
        if (a.hot_T) {
            if (a.pack_k >= 0) {
                if (buf) hipLaunchKernelGGL((k_forward<LPN, J, MODE, true, true, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL((k_forward<LPN, J, MODE, true, true, false>), g, b, 0, s, a);
            } else {
                if (buf) hipLaunchKernelGGL((k_forward<LPN, J, MODE, false, true, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL((k_forward<LPN, J, MODE, false, true, false>), g, b, 0, s, a);
            }
            return hipGetLastError();
        }
        if constexpr (MODE == kFwdPartA) {
            if (a.pack_k >= 0) {
                if (buf) hipLaunchKernelGGL((k_forward<LPN, J, MODE, true, false, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL((k_forward<LPN, J, MODE, true, false, false>), g, b, 0, s, a);
            } else {
                if (buf) hipLaunchKernelGGL((k_forward<LPN, J, MODE, false, false, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL((k_forward<LPN, J, MODE, false, false, false>), g, b, 0, s, a);
            }
            return hipGetLastError();
        }
        return hipErrorInvalidValue;       // (pass B without a prologue returned above)
    }
}

template <int LPN, int J>
hipError_t fwd_dispatch(FwdMode mode, const FwdArgs &a, hipStream_t s, int *n_partials) {
    switch (mode) {
        case kFwdTrain: return fwd_launch<LPN, J, kFwdTrain>(a, s, n_partials);
        case kFwdResidual: return fwd_launch<LPN, J, kFwdResidual>(a, s, n_partials);
        case kFwdPartA: return fwd_launch_pass<LPN, J, kFwdPartA>(a, s, n_partials);
        case kFwdPartB: return fwd_launch_pass<LPN, J, kFwdPartB>(a, s, n_partials);
        default: return fwd_launch<LPN, J, kFwdQ>(a, s, n_partials);
    }
}

}  // namespace

template <int LPN, int J>
static int wt_occupancy() {
    // the training-mode kernels decide (the scoring modes need no more registers); w-tile of 6144 floats
    int n = 5;
    const size_t lds = 6144 * sizeof(float);
    const void *fns[4] = {(const void *)k_forward_wt<LPN, J, kFwdTrain, false, false>, (const void *)k_forward_wt<LPN, J, kFwdTrain, true, false>,
                          (const void *)k_forward_wt<LPN, J, kFwdTrain, false, true>, (const void *)k_forward_wt<LPN, J, kFwdTrain, true, true>};
    for (const void *fn : fns) {
        int ni = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ni, fn, kBlock, lds) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        if (ni < n) n = ni;
    }
    return n < 1 ? 1 : n;
}

int forward_wt_occupancy(int Kp) {
    static std::atomic<int> cache[4] = {0, 0, 0, 0};      // filled on first use; two threads racing compute the same value
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
hipError_t launch_forward(int Kp, FwdMode mode, const FwdArgs &a, hipStream_t s, int *n_partials) {
    switch (Kp) {
        case 32: return fwd_dispatch<8, 1>(mode, a, s, n_partials);
        case 64: return fwd_dispatch<8, 2>(mode, a, s, n_partials);
        case 128: return fwd_dispatch<16, 2>(mode, a, s, n_partials);
        case 256: return fwd_dispatch<16, 4>(mode, a, s, n_partials);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reduce_blocks(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal, double *acc,
                                hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_blocks, dim3(1), dim3(kBlock), 0, s, bsum, nblocks, n_rows, scal, acc);
    return hipGetLastError();
}

hipError_t launch_rescale(int Kp, float *V, float *w, int64_t n1, int32_t pack_k, float sv, float sw, hipStream_t s) {
    int64_t blocks = (n1 * (Kp / 4) + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(kBlock);
    switch (Kp) {
        case 32: hipLaunchKernelGGL(k_rescale<32>, g, b, 0, s, V, w, n1, pack_k, sv, sw); break;
        case 64: hipLaunchKernelGGL(k_rescale<64>, g, b, 0, s, V, w, n1, pack_k, sv, sw); break;
        case 128: hipLaunchKernelGGL(k_rescale<128>, g, b, 0, s, V, w, n1, pack_k, sv, sw); break;
        case 256: hipLaunchKernelGGL(k_rescale<256>, g, b, 0, s, V, w, n1, pack_k, sv, sw); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fmhip
