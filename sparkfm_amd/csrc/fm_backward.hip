// fm_backward.hip — the gradient of the FM step: column walks over the per-batch CSC stream (k_backward,
// k_backward_p), the dense hot block's MFMA product, and the fixed-order fixup passes.  Lane geometry
// and formulas: fm_device.h.
#include "fm_device.h"

#include <cstdlib>

namespace fmhip {
namespace {

// ------------------------------------------------------------------ backward
template <int LPN, int J>
__device__ __forceinline__ void store_row(float *dst, int l, const float4 (&acc)[J], float sa, float sb, float *dsa, float *dsb,
                                          int sl = 0) {
    float4 *d4 = reinterpret_cast<float4 *>(dst) + l;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) d4[jj * LPN] = acc[jj];
    if (l == sl) { *dsa = sa; *dsb = sb; }   // sl: the lane whose scalar sums are the real ones
}

// Fused update of parameter row i from its finished gradient (acc = G_V row, sa = G_w, sb = G_b; the scalars are
// valid in lane sl of the slot): U_i <- U_i - eta_v*((G_V - (sv*U_i)*G_b)/|B|), w_i likewise — apply_piece<KP, true>
// of fm_apply.hip, operation for operation.  pack_k >= 0: slot pack_k of the row is the linear weight.
template <int LPN, int J>
__device__ __forceinline__ void apply_row(const FusedUpd &u, int pack_k, int i, int l, const float4 (&acc)[J], float sa, float sb, int sl) {
    constexpr int KP = 4 * LPN * J;
    const float b = __shfl(sb, sl, LPN);
    float4 *V4 = reinterpret_cast<float4 *>(u.V + (size_t)i * KP) + l;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        float4 x = V4[jj * LPN];
        const float4 v = f4mul(x, u.sv), g = acc[jj];
        const bool has_w = pack_k >= 0 && (l + jj * LPN) == (pack_k >> 2);
        float wslot = 0.f;
        if (has_w) wslot = f4pick(x, pack_k & 3) - u.eta_w * (f4pick(g, pack_k & 3) * u.invb);
        x.x -= u.eta_v * ((g.x - v.x * b) * u.invb);
        x.y -= u.eta_v * ((g.y - v.y * b) * u.invb);
        x.z -= u.eta_v * ((g.z - v.z * b) * u.invb);
        x.w -= u.eta_v * ((g.w - v.w * b) * u.invb);
        if (has_w) f4set(x, pack_k & 3, wslot);
        V4[jj * LPN] = x;
    }
    if (l == sl && pack_k < 0) u.w[i] = u.w[i] - u.eta_w * (sa * u.invb);
}

// Merged finish: the dense update of parameter row i straight from the registers that hold its finished gradient
// (acc = G_V row, sa = G_w, sb = G_b; the scalars are valid in lane sl of the slot) — apply_piece<KP, false> of
// fm_device.h, operation for operation; the packed gradient's row is never written (it stays zero).
template <int LPN, int J>
__device__ __forceinline__ void finish_row(const ApplyArgs &f, int i, int l, const float4 (&acc)[J], float sa, float sb, int sl) {
    constexpr int KP = 4 * LPN * J;
    const float invb = f.invb_val;
    const float b = __shfl(sb, sl, LPN), gw = __shfl(sa, sl, LPN);
    float4 *V4 = reinterpret_cast<float4 *>(f.V + (size_t)i * KP) + l;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        float4 u = V4[jj * LPN];
        const float4 g = acc[jj];
        float4 v = f4mul(u, f.sv_in);
        const bool has_w = f.pack_k >= 0 && (l + jj * LPN) == (f.pack_k >> 2);
        float wslot = 0.f;
        if (has_w) {
            const float wi = f4pick(u, f.pack_k & 3) * f.sw_in;
            wslot = wi - f.eta * fmaf(f.regw, wi, f4pick(g, f.pack_k & 3) * invb);
        }
        u.x = v.x - f.eta * fmaf(f.regv, v.x, (g.x - v.x * b) * invb);
        u.y = v.y - f.eta * fmaf(f.regv, v.y, (g.y - v.y * b) * invb);
        u.z = v.z - f.eta * fmaf(f.regv, v.z, (g.z - v.z * b) * invb);
        u.w = v.w - f.eta * fmaf(f.regv, v.w, (g.w - v.w * b) * invb);
        if (has_w) f4set(u, f.pack_k & 3, wslot);
        V4[jj * LPN] = u;
    }
    if (l == sl) {
        const float wi = f.w[i] * f.sw_in;
        f.w[i] = wi - f.eta * fmaf(f.regw, wi, (f.pack_k >= 0 ? 0.f : gw) * invb);
    }
}

// A finished column piece goes to its destination: the G row of its feature (or, fused, straight into the
// parameters) when the feature has a single piece in the batch, else a piece row that k_fixup2 sums per
// feature (row-blocked streams).
// FIN: the caller is the fixup launch, which may be running the merged finish (the column walk never does)
template <int LPN, int J, bool FIN = false>
__device__ __forceinline__ void store_seg(const BwdArgs &a, int seg, int l, const float4 (&acc)[J], float sa, float sb,
                                          int sl = 0) {
    constexpr int KP = 4 * LPN * J;
    const int dst = a.cdst[seg];
    if (dst >= 0) {
        if (FIN && a.fin_blocks > 0) finish_row<LPN, J>(a.fin, dst, l, acc, sa, sb, sl);
        else if (a.upd.V) apply_row<LPN, J>(a.upd, a.pack_k, dst, l, acc, sa, sb, sl);
        else store_row<LPN, J>(a.GV + (size_t)dst * KP, l, acc, sa, sb, a.Gw + dst, a.Gb + dst, sl);
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
    static_assert(kHotT == 16 && kBlock == 256, "tile mapping below assumes 16-slot pages and 4 waves");
    constexpr int KP = 16 * NJ, PR = KP + kPartPad, W = kBlock / 64;
    constexpr int PG = hot_pages_max(KP);                      // pages one pass over P carries (its accumulators); wider
                                                               // rows take the pages one pass each
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kk = lane >> 4, jj = lane & 15;
    const int nw = nbx * W;
    const int per = (((a.n_rows + nw - 1) / nw) + 3) & ~3;      // rows per wave, a multiple of 4
    const int64_t beg64 = (int64_t)(bx * W + wv) * per;
    const int r_beg = beg64 < a.n_rows ? (int)beg64 : a.n_rows;
    const int r_end = r_beg + per < a.n_rows ? r_beg + per : a.n_rows;
    __shared__ float red[W][kHotT][17];
    __shared__ float reds[W][kHotT][2];
    float *out = a.part + (size_t)bx * (a.pages * kHotT) * PR;
    const int oh = threadIdx.x >> 4, of = threadIdx.x & 15;
    for (int p0 = 0; p0 < a.pages; p0 += PG) {
        const int pages = a.pages - p0 < PG ? a.pages - p0 : PG;   // wave-uniform: the branches on it below are scalar
        const float *xh = a.xhot + (size_t)p0 * a.page_stride;
        f32x4 acc[PG][NJ];
        float sa[PG], sb[PG];
#pragma unroll
        for (int p = 0; p < PG; ++p) {
            sa[p] = 0.f;
            sb[p] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        constexpr int U = (NJ <= 4 && PG <= 4) ? 4 : 2;        // 4-row steps whose loads are in flight together (registers: U x (PG + NJ))
        for (int r0 = r_beg; r0 < r_end; r0 += 4 * U) {
            float x[U][PG], e[U], b[U][NJ];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + 4 * u + kk;
                e[u] = 0.f;
#pragma unroll
                for (int p = 0; p < PG; ++p) x[u][p] = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[u][j] = 0.f;
                if (r < r_end) {
#pragma unroll
                    for (int p = 0; p < PG; ++p)
                        if (p < pages) x[u][p] = xh[(size_t)p * a.page_stride + (size_t)r * kHotT + jj];
                    const float *pr = a.P + (size_t)r * KP;
                    e[u] = a.pack_k >= 0 ? pr[a.pack_k] : a.e[r];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[u][j] = pr[j * 16 + jj];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int p = 0; p < PG; ++p) {
                    if (p >= pages) continue;
                    accum_scalars(sa[p], sb[p], e[u], x[u][p]);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][p], b[u][j], acc[p][j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PG; ++p) {
            if (p >= pages) continue;
            float s0 = sa[p], s1 = sb[p];
            s0 += __shfl_xor(s0, 16, 64);
            s1 += __shfl_xor(s1, 16, 64);
            s0 += __shfl_xor(s0, 32, 64);
            s1 += __shfl_xor(s1, 32, 64);                      // lanes 0..15: the sums of slot `lane` of this page
            float *po = out + (size_t)((p0 + p) * kHotT) * PR;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                __syncthreads();                               // the previous round's readers are done with `red`
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) red[wv][kk * 4 + reg][jj] = acc[p][j][reg];   // C/D: row = (lane>>4)*4 + reg, col = lane&15
                __syncthreads();
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < W; ++w) t += red[w][oh][of];
                po[(size_t)oh * PR + j * 16 + of] = t;
            }
            if (lane < kHotT) { reds[wv][lane][0] = s0; reds[wv][lane][1] = s1; }
            __syncthreads();
            if (threadIdx.x < kHotT * 2) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < W; ++w) t += reds[w][threadIdx.x >> 1][threadIdx.x & 1];
                po[(size_t)(threadIdx.x >> 1) * PR + KP + (threadIdx.x & 1)] = t;
            }
        }
    }
}

// one workgroup per hot slot: sums that slot's partial rows over the hot workgroups (groups of threads
// take interleaved partials, then the groups are summed in order) and stores the G row
__device__ __forceinline__ void hot_reduce_body(const HotArgs &a, int h, int kp, const ApplyArgs &fin, bool merged) {
    const int id = a.hot_ids[h];
    if (id < 0) return;
    const int PR = kp + kPartPad, R4 = PR / 4;
    __shared__ float4 sh[kBlock];
    const int G = kBlock / R4 < 1 ? 1 : kBlock / R4;   // thread groups (R4 <= 65 <= kBlock)
    const int f = threadIdx.x % R4, g = threadIdx.x / R4;
    float4 t = f4zero();
    if (g < G)
        for (int b = g; b < a.nblk; b += G) f4add(t, reinterpret_cast<const float4 *>(a.part + ((size_t)b * (a.pages * kHotT) + h) * PR)[f]);
    sh[threadIdx.x] = t;
    __syncthreads();
    float4 u = f4zero();
    if (threadIdx.x < R4)
        for (int gg = 0; gg < G; ++gg) f4add(u, sh[gg * R4 + threadIdx.x]);
    if (merged) {
        // merged finish: the dense update of the hot feature's parameter row (apply_piece<KP, false>, operation for operation)
        __shared__ float hf[2];
        if (threadIdx.x == kp / 4) { hf[0] = u.x; hf[1] = u.y; }
        __syncthreads();
        if (threadIdx.x < kp / 4) {
            const float invb = fin.invb_val, b = hf[1];
            float4 *V4 = reinterpret_cast<float4 *>(fin.V + (size_t)id * kp) + threadIdx.x;
            float4 x = *V4;
            const float4 v = f4mul(x, fin.sv_in);
            const bool has_w = fin.pack_k >= 0 && (int)threadIdx.x == (fin.pack_k >> 2);
            float wslot = 0.f;
            if (has_w) {
                const float wi = f4pick(x, fin.pack_k & 3) * fin.sw_in;
                wslot = wi - fin.eta * fmaf(fin.regw, wi, f4pick(u, fin.pack_k & 3) * invb);
            }
            x.x = v.x - fin.eta * fmaf(fin.regv, v.x, (u.x - v.x * b) * invb);
            x.y = v.y - fin.eta * fmaf(fin.regv, v.y, (u.y - v.y * b) * invb);
            x.z = v.z - fin.eta * fmaf(fin.regv, v.z, (u.z - v.z * b) * invb);
            x.w = v.w - fin.eta * fmaf(fin.regv, v.w, (u.w - v.w * b) * invb);
            if (has_w) f4set(x, fin.pack_k & 3, wslot);
            *V4 = x;
            if (threadIdx.x == 0) {
                const float wi = fin.w[id] * fin.sw_in;
                fin.w[id] = wi - fin.eta * fmaf(fin.regw, wi, (fin.pack_k >= 0 ? 0.f : hf[0]) * invb);
            }
        }
        return;
    }
    if (!a.upd.V) {
        if (threadIdx.x < kp / 4) {
            reinterpret_cast<float4 *>(a.GV + (size_t)id * kp)[threadIdx.x] = u;
        } else if (threadIdx.x < R4) {
            a.Gw[id] = a.pack_k >= 0 ? 0.f : u.x;
            a.Gb[id] = u.y;
        }
        return;
    }
    // fused update of the hot feature's parameter row (apply_piece<KP, true> of fm_apply.hip, operation for operation)
    __shared__ float hs[2];
    if (threadIdx.x == kp / 4) { hs[0] = u.x; hs[1] = u.y; }
    __syncthreads();
    if (threadIdx.x < kp / 4) {
        const FusedUpd &f = a.upd;
        const float b = hs[1];
        float4 *V4 = reinterpret_cast<float4 *>(f.V + (size_t)id * kp) + threadIdx.x;
        float4 x = *V4;
        const float4 v = f4mul(x, f.sv);
        const bool has_w = a.pack_k >= 0 && (int)threadIdx.x == (a.pack_k >> 2);
        float wslot = 0.f;
        if (has_w) wslot = f4pick(x, a.pack_k & 3) - f.eta_w * (f4pick(u, a.pack_k & 3) * f.invb);
        x.x -= f.eta_v * ((u.x - v.x * b) * f.invb);
        x.y -= f.eta_v * ((u.y - v.y * b) * f.invb);
        x.z -= f.eta_v * ((u.z - v.z * b) * f.invb);
        x.w -= f.eta_v * ((u.w - v.w * b) * f.invb);
        if (has_w) f4set(x, a.pack_k & 3, wslot);
        *V4 = x;
        if (threadIdx.x == 0 && a.pack_k < 0) f.w[id] = f.w[id] - f.eta_w * (hs[0] * f.invb);
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
//
// RangeWalk = what a slot knows about its range before and while it walks it; shared by the plain walk
// (k_backward) and the pipelined one (k_backward_p), which differ only in how the entries are fetched.
template <int LPN, int J, bool PACKED>
struct RangeWalk {
    static constexpr int KP = 4 * LPN * J;
    static constexpr int SLOTS = kBlock / LPN;
    static constexpr int WS = 64 / LPN;                       // slots (ranges) per wave
    static constexpr int PR = KP + kPartPad;
    int l, sl, kj, kc;          // lane in the slot; lane / float4 / component that carry the packed row's scalar slot
    int rho, beg, seg, ca, wbeg, p0, stop;
    bool clean, is_head, tail_partial;
    float4 acc[J];
    float sa, sb;

    // false: this slot has no range in the launch's interval
    __device__ __forceinline__ bool setup(const BwdArgs &a, int bx) {
        l = threadIdx.x & (LPN - 1);
        // packed rows (k < Kp): slot k of the P row is e, so slot k of acc IS sum e*x (the w gradient) and
        // there is no e gather; sum e*x^2 is formed from that slot in the lane that owns it (lane sl)
        sl = PACKED ? (a.pack_k >> 2) & (LPN - 1) : 0;
        kj = PACKED ? (a.pack_k >> 2) / LPN : 0;
        kc = a.pack_k & 3;
        // a launch may cover only the ranges [rho_lo, rho_hi) (feature-chunked backward); block
        // numbering stays aligned to the global range numbering so the wave-sum predicate is unchanged.
        // xcd_chunk > 0: XCD-aware placement — workgroups b, b+8, b+16, .. share an XCD (round-robin
        // dispatch), so XCD x is given the x-th contiguous eighth of the stream: with a row-blocked
        // stream that is a few whole row blocks, whose slice of P then lives in that XCD's L2 only.
        if (a.xlist) {
            // band-affine placement: this workgroup's ranges come from the list of the XCD it runs on (see BwdArgs::xlist)
            const int x = bx & (kXcds - 1);
            int j = (bx >> 3) * SLOTS + (int)(threadIdx.x / LPN), sg = 0;
            while (sg < kXSegs && j >= a.xseg_len[x][sg]) j -= a.xseg_len[x][sg++];
            if (sg == kXSegs) return false;
            rho = a.xlist[a.xseg_off[x][sg] + j];
        } else {
            const int blk = a.xcd_chunk > 0 ? (bx & 7) * a.xcd_chunk + (bx >> 3) : bx;
            rho = (a.rho_lo / SLOTS + blk) * SLOTS + threadIdx.x / LPN;
        }
        if (rho < a.rho_lo || rho >= a.rho_hi) return false;
        beg = rho * kRangeLen;
        const int end = (beg + kRangeLen < a.nnz) ? beg + kRangeLen : a.nnz;
        seg = a.range_seg[rho];
        ca = a.cptr[seg];
        const int cb = a.cptr[seg + 1];                       // the column open at `beg`
        wbeg = (rho - (int)((threadIdx.x & 63) / LPN)) * kRangeLen;
        clean = !a.no_wave_sum && (ca <= wbeg) && (cb >= wbeg + WS * kRangeLen);   // wave-uniform by construction
        is_head = ca < beg;
        p0 = beg;
        stop = end;
        tail_partial = false;
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
#pragma unroll
        for (int jj = 0; jj < J; ++jj) acc[jj] = f4zero();
        sa = 0.f;
        sb = 0.f;
        return true;
    }

    __device__ __forceinline__ float *part_row(const BwdArgs &a, int which) const { return a.part + ((size_t)rho * 2 + which) * PR; }

    // the open column ends in front of the entry being walked: its sum goes out, the accumulators restart
    __device__ __forceinline__ void flush(const BwdArgs &a) {
        if (is_head) {
            float *pr = part_row(a, 0);
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

    // acc += x * (the entry's P row); the two scalar sums (sum e*x -> G_w, sum e*x^2 -> G_b)
    __device__ __forceinline__ void add(const float4 (&pv)[J], float xj, float ej) {
#pragma unroll
        for (int jj = 0; jj < J; ++jj) f4fma(acc[jj], pv[jj], xj);  // sum x * (e*q)
        if (PACKED) {
            float pk = 0.f;
#pragma unroll
            for (int jj = 0; jj < J; ++jj)
                if (jj == kj) pk = f4pick(pv[jj], kc);
            sb = fmaf(__fmul_rn(pk, xj), xj, sb);
        } else {
            if (kEInP) ej = e_from_row(pv[0], l);      // the residual rides in the row just gathered (fm_device.h): no e gather
            accum_scalars(sa, sb, ej, xj);
        }
    }

    // one live entry at stream position `pos`: `flag` = bit 31 of its row word (first entry of its column)
    __device__ __forceinline__ void entry(const BwdArgs &a, int pos, uint32_t rj, const float4 (&pv)[J], float xj, float ej) {
        if ((rj >> 31) && pos != p0) flush(a);
        add(pv, xj, ej);
    }

    __device__ __forceinline__ void finish(const BwdArgs &a) {
        if (clean) {
            slots_reduce<LPN, J>(acc, sa, sb);
            if (beg == wbeg) {
                float *pr = part_row(a, ca == wbeg ? 1 : 0);
                store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
            }
            return;
        }
        if (p0 >= stop) return;   // everything in this range belonged to the previous slot
        if (is_head) {
            float *pr = part_row(a, 0);
            store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
        } else if (tail_partial) {
            float *pr = part_row(a, 1);
            store_row<LPN, J>(pr, l, acc, sa, sb, pr + KP, pr + KP + 1, sl);
        } else {
            store_seg<LPN, J>(a, seg, l, acc, sa, sb, sl);
        }
    }
};

// The plain walk: LPN entries per step (one per lane), their P rows gathered CH at a time with flat loads —
// the variant for P tables of 4 GiB and more, and for rows wider than the pipelined kernel's registers allow.
template <int LPN, int J, bool PACKED, bool HOT>
__global__ __launch_bounds__(kBlock) void k_backward(BwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    if (HOT && (int)blockIdx.x >= a.hot_first && (int)blockIdx.x < a.hot_first + a.hot_blocks) {
        hot_backward_body<KP / 16>(a.hot, (int)blockIdx.x - a.hot_first, a.hot_blocks);
        return;
    }
    constexpr int CH = (LPN * J > 16) ? (16 / J) : LPN;  // entries whose P rows are in flight together
    RangeWalk<LPN, J, PACKED> w;
    if (!w.setup(a, (HOT && (int)blockIdx.x >= a.hot_first) ? (int)blockIdx.x - a.hot_blocks : (int)blockIdx.x)) return;
    const int l = w.l;
    for (int base = w.p0; base < w.stop; base += LPN) {
        const int p = base + l;
        uint32_t rf = 0u;
        float x = 0.f, ee = 0.f;
        if (p < w.stop) {
            rf = stream_load(a.crow + p);
            x = stream_load(a.cval + p);
            if (!PACKED && !kEInP) ee = a.e[rf & 0x7fffffffu];
        }
        const int cnt = (w.stop - base) < LPN ? (w.stop - base) : LPN;
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
                const float ej = (PACKED || kEInP) ? 0.f : slot_bcast<LPN>(ee, c0 + j);
                if (c0 + j < cnt) w.entry(a, base + c0 + j, rj[j], pv[j], xj, ej);
            }
        }
    }
    w.finish(a);
}

// Pipelined variant of k_backward (same walk, same predicates, same outputs): the CSC index /
// value / e loads of a whole super-group (up to 64 entries) are issued up front, the P-row
// gathers go through a buffer descriptor (dead entries fetch nothing) and are double-buffered in
// chunks of CHB entries so chunk c+1 is in flight while chunk c is accumulated.
// (HOT: the waves-per-SIMD bound keeps the MFMA accumulators of the hot body from costing the walkers
// their third wave)
#ifndef FMHIP_EXP_FIX_SKIP
#define FMHIP_EXP_FIX_SKIP 0      // timing-only ablation of k_fixup's parts: 1 merged update, 2 hot rows, 4 short columns, 8 long columns
#endif
#ifndef FMHIP_EXP_NO_XE_BCAST
#define FMHIP_EXP_NO_XE_BCAST 0   // timing-only ablation: the value / residual broadcasts of the pipelined walk dropped (results wrong)
#endif
#ifndef FMHIP_EXP_NO_E
#define FMHIP_EXP_NO_E 0          // timing-only ablation: 1 = the residual gather of the pipelined walk dropped (G_w / G_b wrong);
                                  // 2 = the residual read at the entry's STREAM position instead of its row (coalesced: the cost of
                                  // the loads without the gather) — what an LDS-staged or stream-resident e could save at most
#endif
#define FMHIP_BWD_ABLATIONS ((FMHIP_EXP_FIX_SKIP ? 64 : 0) | (FMHIP_EXP_NO_XE_BCAST ? 128 : 0) | (FMHIP_EXP_NO_E ? 256 : 0))
#if FMHIP_BWD_ABLATIONS && !defined(FMHIP_ABLATION_BUILD)
#error "a result-changing FMHIP_EXP_* ablation is set without FMHIP_ABLATION_BUILD: timing-only variants are built by tools/build_variant.sh"
#endif
#ifndef FMHIP_BWD_WAVES
#define FMHIP_BWD_WAVES 3
#endif
#ifndef FMHIP_BWD_WAVES8
#define FMHIP_BWD_WAVES8 4    // the same bound for the 8-lane slots (Kp = 32) alone: four, with the 4-entry gather chunks below
#endif
template <int LPN, int J, bool PACKED, bool HOT>
__global__ __launch_bounds__(kBlock, (HOT && J == 1 ? (LPN == 8 ? FMHIP_BWD_WAVES8 : FMHIP_BWD_WAVES) : 1)) void k_backward_p(BwdArgs a) {
    constexpr int KP = 4 * LPN * J;
    if (HOT && (int)blockIdx.x >= a.hot_first && (int)blockIdx.x < a.hot_first + a.hot_blocks) {
        hot_backward_body<KP / 16>(a.hot, (int)blockIdx.x - a.hot_first, a.hot_blocks);
        return;
    }
#ifndef FMHIP_BWD_SG
#define FMHIP_BWD_SG 8
#endif
    constexpr int SG = (kRangeLen / LPN) < FMHIP_BWD_SG ? (kRangeLen / LPN) : FMHIP_BWD_SG;   // lane-groups per super-group
    // entries per gather chunk (two chunks are in flight).  8-lane slots (Kp = 32): 4 — with 8 the double buffer alone held 64
    // registers and the kernel three waves per SIMD; 4 fits four waves (127 registers) and the walk gained 3 % at C3, 5 % at
    // C2 (r04_experiments.md section 19; the same bits: only the grouping of the loads changed).  16-lane slots: FMHIP_BWD_CHB16.
#ifndef FMHIP_BWD_CHB
#define FMHIP_BWD_CHB 4
#endif
#ifndef FMHIP_BWD_CHB16
#define FMHIP_BWD_CHB16 0     // 0 = 8 / J
#endif
    constexpr int CHB = (LPN * J <= 8) ? FMHIP_BWD_CHB : ((J == 1 && FMHIP_BWD_CHB16 > 0) ? FMHIP_BWD_CHB16 : ((8 / J) > 0 ? (8 / J) : 1));
    constexpr int NCH = SG * LPN / CHB;                                       // chunks per super-group
    RangeWalk<LPN, J, PACKED> w;
    if (!w.setup(a, (HOT && (int)blockIdx.x >= a.hot_first) ? (int)blockIdx.x - a.hot_blocks : (int)blockIdx.x)) return;
    const int l = w.l;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(a.P, a.p_bytes);
    for (int sbase = w.p0; sbase < w.stop; sbase += SG * LPN) {
        uint32_t rf[SG];
        float x[SG], ee[SG];
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            const int p = sbase + g * LPN + l;
            rf[g] = 0u;
            x[g] = 0.f;
            if (p < w.stop) { rf[g] = stream_load(a.crow + p); x[g] = stream_load(a.cval + p); }
        }
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            const int p = sbase + g * LPN + l;
            ee[g] = 0.f;
            if (!PACKED && !kEInP && p < w.stop && FMHIP_EXP_NO_E != 1) ee[g] = a.e[FMHIP_EXP_NO_E == 2 ? (uint32_t)(p & 0xffff) : (rf[g] & 0x7fffffffu)];
        }
        float4 pv[2][CHB][J];
        uint32_t rj[2][CHB];
        auto issue = [&](int ch, int buf) {
#pragma unroll
            for (int j = 0; j < CHB; ++j) {
                const int ent = ch * CHB + j;             // entry index inside the super-group
                const int g = ent / LPN, jl = ent % LPN;
                rj[buf][j] = slot_bcast<LPN>(rf[g], jl);
                const bool live = sbase + ent < w.stop;
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
            for (int j = 0; j < CHB; ++j) fl |= (cpos + j == w.p0) ? 0u : rj[buf][j];
            const bool plain = (cpos + CHB <= w.stop) && !(fl >> 31);
            if (__all(plain)) {
#pragma unroll
                for (int j = 0; j < CHB; ++j) {
                    const int ent = ch * CHB + j;
                    const int g = ent / LPN, jl = ent % LPN;
                    w.add(pv[buf][j], FMHIP_EXP_NO_XE_BCAST ? x[g] : slot_bcast<LPN>(x[g], jl), (PACKED || kEInP) ? 0.f : (FMHIP_EXP_NO_XE_BCAST ? ee[g] : slot_bcast<LPN>(ee[g], jl)));
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < CHB; ++j) {
                const int ent = ch * CHB + j;
                const int g = ent / LPN, jl = ent % LPN;
                const float xj = FMHIP_EXP_NO_XE_BCAST ? x[g] : slot_bcast<LPN>(x[g], jl);
                const float ej = (PACKED || kEInP) ? 0.f : (FMHIP_EXP_NO_XE_BCAST ? ee[g] : slot_bcast<LPN>(ee[g], jl));
                if (sbase + ent < w.stop) w.entry(a, sbase + ent, rj[buf][j], pv[buf][j], xj, ej);
            }
        }
    }
    w.finish(a);
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
    // no_wave: the launch that wrote the partials formed no wave sums (band-affine placement): every range is a unit
    __device__ __forceinline__ ColumnUnits(int ca_, int cb, bool no_wave) : ca(ca_) {
        ra = ca / kRangeLen;
        const int rb = (cb - 1) / kRangeLen;
        w_lo = (ca + WSPAN - 1) / WSPAN;
        const int w_hi = cb / WSPAN;                              // clean waves [w_lo, w_hi)
        nw = (w_hi > w_lo && !no_wave) ? w_hi - w_lo : 0;
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
    // Block 0 (when asked for) finishes the residual statistics of this step (saves a launch).  It is one workgroup walking
    // the forward's per-block partials — a serial chain of loads — so it goes out FIRST and runs beside everything else;
    // as the last block it was the tail of the launch (k = 16, whose forward leaves 7,813 partials: +7 us).
    int bx = (int)blockIdx.x;
    const int nbx = (int)gridDim.x - (a.red_bsum ? 1 : 0);
    if (a.red_bsum) {
        if (bx == 0) {
            __shared__ double sh[3][kBlock / 64];
            reduce_blocks_body(a.red_bsum, a.red_nblocks, a.red_rows, a.red_scal, a.red_acc, sh, a.red_w0, a.red_eta, a.red_reg0);
            return;
        }
        bx -= 1;
    }
    // merged finish: the last fin_blocks workgroups update every parameter row the fixup part
    // does not own (those update themselves, straight from registers) — bandwidth-bound work beside latency-bound work
    const int fin0 = nbx - a.fin_blocks;
    if (a.fin_blocks > 0 && bx >= fin0) {
        if (FMHIP_EXP_FIX_SKIP & 1) return;
        constexpr int LPR = KP / 4;
        const ApplyArgs &f = a.fin;
        const int64_t total = (f.row_hi - f.row_lo) * LPR;
        for (int64_t idx = (int64_t)(bx - fin0) * kBlock + threadIdx.x; idx < total; idx += (int64_t)a.fin_blocks * kBlock) {
            const int64_t i = f.row_lo + idx / LPR;
            if (i < a.fin_own_bits && (a.fin_own[i >> 5] >> (i & 31) & 1u)) continue;
            apply_piece<KP, false>(f, i, (int)(idx % LPR), f.invb_val);
        }
        return;
    }
    if (HOT) {
        // one more workgroup per hot slot finishes the dense hot block's gradient rows
        const int hot0 = fin0 - a.hot.pages * kHotT;
        if (bx >= hot0) {
            if (FMHIP_EXP_FIX_SKIP & 2) return;
            hot_reduce_body(a.hot, bx - hot0, KP, a.fin, a.fin_blocks > 0);
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
    if (bx < blocks_short) {
        // ---- one slot per short column
        if (FMHIP_EXP_FIX_SKIP & 4) return;
        const int idx = bx * SLOTS + threadIdx.x / LPN;
        if (idx >= a.n_split_short) return;
        const int seg = a.split_short[idx];
        const ColumnUnits<LPN, J> cu(a.cptr[seg], a.cptr[seg + 1], a.no_wave_sum != 0);
        for (int t = 0; t < cu.count; ++t) {
            const float *pr = cu.row(a.part, t);
            const float4 *p4 = reinterpret_cast<const float4 *>(pr) + l;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) f4add(acc[jj], p4[jj * LPN]);
            sa += pr[KP];
            sb += pr[KP + 1];
        }
        store_seg<LPN, J, true>(a, seg, l, acc, sa, sb);
        return;
    }
    // ---- one WORKGROUP per long column: units strided over the 4 waves x WS slots, four in flight per
    // slot; slots tree-summed inside each wave, the 4 wave sums added in wave order through LDS
    const int ws = lane / LPN;
    const int wv = threadIdx.x >> 6;
    const int idx = bx - blocks_short;
    if (idx >= a.n_split || (FMHIP_EXP_FIX_SKIP & 8)) return;
    const int seg = a.split_seg[idx];
    const ColumnUnits<LPN, J> cu(a.cptr[seg], a.cptr[seg + 1], a.no_wave_sum != 0);
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
        store_seg<LPN, J, true>(a, seg, l, acc, sa, sb);
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

template <int LPN, int J>
hipError_t bwd_dispatch(const BwdArgs &a, hipStream_t s) {
    constexpr int SLOTS = kBlock / LPN;
    if (a.rho_hi <= a.rho_lo && a.hot_blocks < 1) return hipSuccess;
    int nblk = 0;
    if (a.rho_hi > a.rho_lo) nblk = (a.rho_hi - 1) / SLOTS - a.rho_lo / SLOTS + 1;
    BwdArgs a2 = a;
    a2.xcd_chunk = a.xcd_chunk > 0 ? (nblk + 7) / 8 : 0;      // blocks per XCD
    if (a.xlist) {
        // band-affine placement: kXcds interleaved lists, workgroup b takes SLOTS ranges of list b % 8
        int longest = 0;
        for (int x = 0; x < kXcds; ++x) {
            int len = 0;
            for (int sg = 0; sg < kXSegs; ++sg) len += a.xseg_len[x][sg];
            longest = len > longest ? len : longest;
        }
        nblk = kXcds * ((longest + SLOTS - 1) / SLOTS);
        a2.xcd_chunk = 0;
    }
    // Where the block product's workgroups sit in the launch: at the front by default; with the band-affine plan behind the
    // band-affine part of the lists (FMHIP_HOT_AT = percent of the walkers in front of them; measurement knob) so that their
    // 100 MB of streams do not pass through the L2s while the bands are meant to stay in them
    a2.hot_first = 0;
    if (a.xlist && a.hot_blocks > 0) {
        static const int hot_at = getenv("FMHIP_HOT_AT") ? atoi(getenv("FMHIP_HOT_AT")) : 0;
        a2.hot_first = (int)((int64_t)nblk * hot_at / 100) / kXcds * kXcds;
    }
    dim3 g((unsigned)((a2.xcd_chunk > 0 ? a2.xcd_chunk * 8 : nblk) + a.hot_blocks)), b(kBlock);
    // the pipelined kernel needs P to fit a 32-bit buffer view (< 4 GiB per batch); for k > 64 (J > 1)
    // its register footprint spills, so those sizes take the plain walk
    const bool pipe = J == 1 && a.p_bytes && a.pipelined == 1;
    static const size_t lds_pad = getenv("FMHIP_BWD_LDS_PAD") ? (size_t)atol(getenv("FMHIP_BWD_LDS_PAD")) : 0;   // experiment: fewer resident workgroups
#define FMHIP_BW(PACKED_, HOT_)                                                               \
    if (pipe) hipLaunchKernelGGL((k_backward_p<LPN, J, PACKED_, HOT_>), g, b, lds_pad, s, a2); \
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
    const int hot = a.hot_blocks > 0 ? a.hot.pages * kHotT : 0;
    if (a.n_split < 1 && a.n_split_short < 1 && !extra && !hot && a.fin_blocks < 1) return hipSuccess;
    dim3 g((unsigned)((a.n_split_short + SLOTS - 1) / SLOTS + a.n_split + hot + a.fin_blocks + extra)), b(kBlock);
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

int backward_ablations() { return FMHIP_BWD_ABLATIONS; }

int hot_blocks(int Kp, int64_t n_rows) {
    (void)Kp;
    int64_t b = (n_rows + 63) / 64;       // at least 16 rows per wave
    if (b > 256) b = 256;                 // one hot workgroup per CU, next to the column walkers (128 / 512 / 1024: no better)
    return b < 1 ? 1 : (int)b;
}

}  // namespace fmhip
