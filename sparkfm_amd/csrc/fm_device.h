// fm_device.h — device-side helpers shared by fm_forward.hip / fm_backward.hip / fm_apply.hip, the
// hand-written gfx950 (CDNA4, wave64) kernels of the FM mini-batch SGD step.
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
// Lane geometry: a "slot" = LPN consecutive lanes (8 at Kp = 32 and in the Kp = 64 forward, 16 otherwise) owning one CSR row
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
#pragma once
#include "fm_kernels.h"
#ifndef FMHIP_DPP_BCAST
#define FMHIP_DPP_BCAST 1
#endif

namespace fmhip {
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

__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// P rows are written once by the forward and read by OTHER compute units in the backward (never from
// this XCD's L2, which is not coherent with theirs): optionally stored non-temporally so they do not
// displace the gathered V rows from L2 on their way out.
__device__ __forceinline__ void p_store(float4 *dst, float4 v) {
#ifdef FMHIP_P_NT
    __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(dst));
#else
    *dst = v;
#endif
}

// ---- the residual rides in the P row (k == Kp: no spare slot) -------------------------------------------------
// The backward needs e_r for every entry it walks (G_w += e x, G_b += e x^2).  When a P row has a spare slot (k < Kp) e sits
// there; when it has none (k = 32, 64, 128, 256) the walk used to gather e[row] beside the row — a 64-lane instruction touching
// up to 64 lines of an L2-resident table: 17-18 us of the 143-us walk at C3 (profiles/r04_experiments.md section 11).  Instead
// the forward now writes the 32 bits of e into the mantissa LSBs of the row's first 32 floats (lane l < 8 of the slot holds
// floats 4l .. 4l+3 and carries bits 4l .. 4l+3), and the walk reads them back from the row it has gathered anyway: three
// DPP moves and a dozen bit operations per entry on a vector ALU that idles 88 % of the time, no memory request at all.
// e itself is exact; a carrying P element moves to the nearest float with those low bits (carry_bits below), deterministically.
#ifndef FMHIP_E_IN_P
#define FMHIP_E_IN_P 1      // 0 = the separate gather (A/B builds); forward and backward must agree
#endif
constexpr bool kEInP = FMHIP_E_IN_P != 0;
#ifndef FMHIP_E_BITS
#define FMHIP_E_BITS 2      // bits of e per carrying float: 1 = all four floats of a lane, one bit each (<= 1 ulp on 32 floats);
                            // 2 = two floats, two bits each; 4 = one float, four bits (fewer bit operations per entry, a larger nudge)
#endif
// x -> the NEAREST float whose N low mantissa bits are b (ties to the even multiple of 2^N): an unbiased rounding, and half as
// far as overwriting the bits, which truncates towards a fixed residue (N = 2: <= 2 units in the last place, 2.4e-7 relative,
// against 3).  Zero, the tiniest denormals, the last floats below infinity and non-finite values get their bits overwritten.
#ifndef FMHIP_E_ROUND
#define FMHIP_E_ROUND 1     // 0 = overwrite the bits (A/B builds)
#endif
template <int N>
__device__ __forceinline__ float carry_bits(float x, uint32_t b) {
    constexpr uint32_t M = (1u << N) - 1u;
    const uint32_t u = __float_as_uint(x), m = u & 0x7fffffffu;
    uint32_t r = (m & ~M) | b;
    if (FMHIP_E_ROUND && m > M && m < 0x7f7ffff0u) {
        uint32_t y = m - b;
        y += (M >> 1) + ((y >> N) & 1u);
        r = (y & ~M) + b;
    }
    return __uint_as_float((u & 0x80000000u) | r);
}
__device__ __forceinline__ float4 embed_bits4(float4 v, uint32_t bits) {
    if (FMHIP_E_BITS == 4) {
        v.x = carry_bits<4>(v.x, bits & 15u);
    } else if (FMHIP_E_BITS == 2) {
        v.x = carry_bits<2>(v.x, bits & 3u);
        v.y = carry_bits<2>(v.y, (bits >> 2) & 3u);
    } else {
        v.x = carry_bits<1>(v.x, bits & 1u);
        v.y = carry_bits<1>(v.y, (bits >> 1) & 1u);
        v.z = carry_bits<1>(v.z, (bits >> 2) & 1u);
        v.w = carry_bits<1>(v.w, (bits >> 3) & 1u);
    }
    return v;
}
__device__ __forceinline__ uint32_t extract_bits4(float4 v) {
    if (FMHIP_E_BITS == 4) return __float_as_uint(v.x) & 15u;
    if (FMHIP_E_BITS == 2) return (__float_as_uint(v.x) & 3u) | ((__float_as_uint(v.y) & 3u) << 2);
    return (__float_as_uint(v.x) & 1u) | ((__float_as_uint(v.y) & 1u) << 1) | ((__float_as_uint(v.z) & 1u) << 2) |
           ((__float_as_uint(v.w) & 1u) << 3);
}
// OR over each aligned group of 8 lanes (quad butterfly, then the mirror of the half row), valid in all 8
__device__ __forceinline__ uint32_t or8(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);     // quad_perm:[1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);     // quad_perm:[2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);    // row_half_mirror
    return v;
}
// the residual of the row whose first float4 (per lane of the slot) is `first`; l = the lane's index in its slot
__device__ __forceinline__ float e_from_row(float4 first, int l) {
    return __uint_as_float(or8(l < 8 ? extract_bits4(first) << (4 * l) : 0u));
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
// gr: the row of the gradient arrays that belongs to parameter row i (the packed gradient: gr == i; the compact
// gradient of the touched-rows exchange: the position of feature i in the step's union)
template <int KP, bool ROWS>
__device__ __forceinline__ void apply_piece(const ApplyArgs &a, int64_t i, int c, float invb, int64_t gr) {
    constexpr int LPR = KP / 4;
    float4 *V4 = reinterpret_cast<float4 *>(a.V) + i * LPR + c;
    float4 *G4 = reinterpret_cast<float4 *>(a.GV) + gr * LPR + c;
    const float b = a.Gb[gr];
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
        const float us = a.w[i], gi = a.Gw[gr] * invb;
        const float wi = us * a.sw_in;
        a.w[i] = ROWS ? us - a.eta_w * gi : wi - a.eta * fmaf(a.regw, wi, gi);
        a.Gw[gr] = 0.f;
        a.Gb[gr] = 0.f;  // same wave already holds its copy of b (all lanes of a row share a wave)
    }
}
template <int KP, bool ROWS>
__device__ __forceinline__ void apply_piece(const ApplyArgs &a, int64_t i, int c, float invb) {
    apply_piece<KP, ROWS>(a, i, c, invb, i);
}

// 1/|B|: from the step's row count on the device, or given by the host
__device__ __forceinline__ float apply_invb(const ApplyArgs &a) {
    if (a.use_invb_val) return a.invb_val;
    const float rows = *a.rows;
    return rows > 0.f ? 1.0f / rows : 0.f;
}

__device__ __forceinline__ void apply_w0(const ApplyArgs &a, float invb) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float w0 = *a.w0;
        *a.w0 = w0 - a.eta * fmaf(a.reg0, w0, a.scal[0] * invb);
    }
}

// ------------------------------------------------------------------ reduce
// one block, fixed order: sums the forward's per-block partials into
// scal = {sum e, sum e^2, rows, nonfinite} (fp32, part of the packed gradient) and acc (fp64, +=)
// w0 (optional): also steps the bias, theta <- theta - eta*(sum e / rows + reg0*theta) — what k_apply does in the unfused path
__device__ __forceinline__ void reduce_blocks_body(const double *bsum, int32_t nblocks, int32_t n_rows, float *scal,
                                                   double *acc, double (*sh)[kBlock / 64], float *w0 = nullptr, float eta = 0.f,
                                                   float reg0 = 0.f) {
    double s1 = 0.0, s2 = 0.0, bad = 0.0;
    const double4 *b4 = reinterpret_cast<const double4 *>(bsum);
    int i = threadIdx.x;
    for (; i + 3 * kBlock < nblocks; i += 4 * kBlock) {   // four loads in flight; added in index order
        const double4 b0 = b4[i], b1 = b4[i + kBlock], b2 = b4[i + 2 * kBlock], b3 = b4[i + 3 * kBlock];
        s1 += b0.x; s2 += b0.y; bad += b0.z;
        s1 += b1.x; s2 += b1.y; bad += b1.z;
        s1 += b2.x; s2 += b2.y; bad += b2.z;
        s1 += b3.x; s2 += b3.y; bad += b3.z;
    }
    for (; i < nblocks; i += kBlock) {
        const double4 b = b4[i];
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
        if (w0) {
            const float rows = (float)n_rows, invb = rows > 0.f ? 1.0f / rows : 0.f, b0 = *w0;
            *w0 = b0 - eta * fmaf(reg0, b0, (float)t1 * invb);
        }
    }
}


}  // namespace
}  // namespace fmhip
