// fmhip_step.hip — the launch sequence of one mini-batch step on one GPU, in pieces:
//     step_forward -> step_backward (whole, or one feature interval at a time) -> [exchange, fmhip_comm.hip] -> step_apply*
// plus the argument blocks of the kernels (fwd_args / bwd_args), the planning of where the update runs (plan_fused: a launch
// of its own, inside the fixup launch, inside the column walk) and the lazily decayed tables' bookkeeping.  Everything is
// asynchronous on the model's stream; the arithmetic lives in fm_forward.hip / fm_backward.hip / fm_apply.hip.
#include "fmhip_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

using namespace fmhip;

#ifndef FMHIP_FIN_BLOCKS
#define FMHIP_FIN_BLOCKS 2048     // cap on the merged finish's update workgroups (beside ~2k fixup workgroups at C3)
#endif

namespace fmhip {
namespace host {

struct ProfScope {
    fmhip_model *m;
    ProfRec r{};
    bool on;
    ProfScope(fmhip_model *m_, int kind, int64_t nnz, int64_t rows) : m(m_), on(m_->profiling) {
        if (on && m->prof_rotate) {
            static const int live[4] = {FMHIP_K_FORWARD, FMHIP_K_BACKWARD, FMHIP_K_FIXUP, FMHIP_K_APPLY};
            const int64_t period = m->prof_period > 0 ? m->prof_period : 1;
            if (m->prof_step % period != 0 || live[(m->prof_step / period) % 4] != kind) on = false;
        }
        if (!on) return;
        r.kind = kind;
        r.nnz = nnz;
        r.rows = rows;
        r.step = m->prof_step;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.a, m->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, m->stream);
        m->prof.push_back(r);
    }
};

// ---- model helpers ----------------------------------------------------------------

int check_pair(fmhip_model_t m, fmhip_dataset_t d) {
    if (!m || !d) return fail(FMHIP_ERR_INVALID, "model or dataset is NULL");
    if (m->device != d->device) return fail(FMHIP_ERR_INVALID, "model on device %d, dataset on device %d", m->device, d->device);
    if (d->dimension > m->n)
        return fail(FMHIP_ERR_SHAPE, "dataset has feature index %lld but the model has num_attribute = %lld",
                    (long long)d->dimension, (long long)m->n);
    return set_device(m->device);
}

// training calls need the transposes a scoring-only dataset does not have
int check_train(fmhip_model_t m, fmhip_dataset_t d) {
    TRY(check_pair(m, d));
    if (d->scoring_only)
        return fail(FMHIP_ERR_UNSUPPORTED, "dataset was created with fmhip_rows_create (scoring only): it has no transposes to train on");
    return FMHIP_OK;
}

int check_batch(fmhip_dataset_t d, int64_t batch) {
    if (batch < 0 || batch >= (int64_t)d->batches.size())
        return fail(FMHIP_ERR_INVALID, "batch %lld out of range [0, %zu)", (long long)batch, d->batches.size());
    return FMHIP_OK;
}

int ensure_workspace(fmhip_model_t m, fmhip_dataset_t d) {
    TRY(m->P.ensure((size_t)std::max<int64_t>(d->max_rows, 1) * m->Kp));
    TRY(m->e.ensure((size_t)std::max<int64_t>(d->max_rows, 1)));
    TRY(m->part.ensure((size_t)std::max<int32_t>(d->max_ranges, 1) * 2 * (m->Kp + kPartPad)));
    TRY(m->pieces.ensure((size_t)std::max<int32_t>(d->max_pieces, 1) * (m->Kp + kPartPad)));
    TRY(m->bsum.ensure((size_t)kMaxFwdBlocks * 4));
    if (d->hot_T) TRY(m->hot_part.ensure((size_t)hot_blocks(m->Kp, d->max_rows) * d->hot_pages * kHotT * (m->Kp + kPartPad)));
    return FMHIP_OK;
}

FwdArgs fwd_args(fmhip_model_t m, fmhip_dataset_t d, const BatchMeta &bm) {
    FwdArgs a{};
    a.row_ptr = d->row_ptr.p;
    a.col = d->col.p;
    a.val = d->val.p;
    a.y = d->y.p;
    a.V = m->V.p;
    {
        // tables of 4 GiB and more do not fit a 32-bit buffer view and take the flat-address kernels
        // (FMHIP_TUNE_FLAT_ADDRESS forces those for any size, so that tests reach them on small inputs)
        const uint64_t vb = (uint64_t)m->n1p * m->Kp * sizeof(float);
        a.v_bytes = (vb < 0xffffffffull && !m->tv(kTuneFlat)) ? (uint32_t)vb : 0u;
    }
    a.sv = (float)m->sv;
    a.sw = (float)m->sw;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.row0 = bm.row0;
    a.nz0 = bm.nnz0;
    // longest-first row order: pays for wide rows only (k=64: -8 %); at Kp = 32 it changes nothing but the
    // locality of the per-row streams (forward FETCH_SIZE 184 -> 269 MB), so narrow models walk in stored order
    a.order = (m->tv(kTuneRowOrder) && (m->Kp >= 64 || getenv("FMHIP_ORDER_ALL"))) ? d->row_order.p + bm.row0 : nullptr;
    a.n_rows = (int32_t)bm.rows;
    a.P = m->P.p;
    a.e = m->e.p;
    a.yhat = nullptr;
    a.pack_k = m->pack_k();
    a.hot_T = d->hot_T;
    a.xhot = d->hot_T ? d->xhot.p + (size_t)bm.row0 * kHotT : nullptr;
    a.hot_ids = d->d_hot_ids.p;
    a.bsum = m->bsum.p;
    {
        // LDS V-tile size: as many hot rows as fit 128 KiB (+ their w), capped by the model
        int64_t t = (128 * 1024) / ((int64_t)m->Kp * 4);
        if (m->tv(kTuneTile) > 0) t = m->tv(kTuneTile);
        a.tile_rows = (int32_t)std::min<int64_t>(t, m->n1);
        a.wt_rows = (int32_t)std::min<int64_t>(m->tv(kTuneTile) > 0 ? m->tv(kTuneTile) : 6144, m->n1);   // 24 KiB
        a.variant = m->tv(kTuneFwd);
        a.occ_cap = m->tv(kTuneFwdOcc);
    }
    return a;
}

BwdArgs bwd_args(fmhip_model_t m, fmhip_dataset_t d, int64_t b) {
    const BatchMeta &bm = d->batches[(size_t)b];
    BwdArgs a{};
    a.crow = d->crow.p + bm.nnz0;
    a.cval = d->cval.p + bm.nnz0;
    a.range_seg = d->range_seg.p + bm.range_off;
    a.cfeat = d->cfeat.p + bm.col_off;
    a.cdst = d->cdst.p + bm.col_off;
    a.pieces = m->pieces.p;
    a.mp_feat = d->mp_feat.p + bm.mp_off;
    a.mp_ptr = d->mp_ptr.p + bm.mp_off + b;
    a.n_mp = bm.n_mp;
    a.pack_k = m->pack_k();
    a.cptr = d->cptr.p + bm.col_off + b;
    a.split_seg = d->split_seg.p + bm.split_off;
    a.split_short = d->split_short.p + bm.split_short_off;
    a.n_split_short = bm.n_split_short;
    a.nnz = bm.cnnz;
    a.n_ranges = bm.n_ranges;
    a.rho_lo = 0;
    a.rho_hi = bm.n_ranges;
    a.xcd_chunk = m->tv(kTuneXcd) == 1 ? 1 : 0;
    a.pipelined = m->tv(kTuneBwd);
    a.n_split = bm.n_split;
    a.P = m->P.p;
    {
        const uint64_t pb = (uint64_t)bm.rows * m->Kp * sizeof(float);
        a.p_bytes = (pb < 0xffffffffull && !m->tv(kTuneFlat)) ? (uint32_t)pb : 0u;
    }
    a.e = m->e.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    if (m->view) {     // the rows go to a compact buffer (touched-rows exchange): column s -> row view->cdst[s]
        a.GV = m->view->GV;
        a.Gw = m->view->Gw;
        a.Gb = m->view->Gb;
        a.cdst = m->view->cdst;
    }
    a.part = m->part.p;
    return a;
}

// forward of one batch: P = e*q, e, per-block statistics partials
int step_forward(fmhip_model_t m, fmhip_dataset_t d, int64_t b) {
    const BatchMeta &bm = d->batches[(size_t)b];
    TRY(ensure_workspace(m, d));
    if (m->grad_dirty) {
        HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
        m->grad_dirty = false;
    }
    {
        ProfScope ps(m, FMHIP_K_FORWARD, bm.nnz_total, bm.rows);
        HIP_TRY(launch_forward(m->Kp, kFwdTrain, fwd_args(m, d, bm), m->stream, &m->fwd_parts));
    }
    m->grad_dirty = true;
    m->last_nnz = bm.nnz_total;
    m->last_rows = bm.rows;
    m->bw_next_hi = INT64_MAX;
    m->hot_pending = d->hot_T > 0;
    return FMHIP_OK;
}

// The training forward in two passes over every row's entries (FwdMode kFwdPartA / kFwdPartB; the dataset's rows are partitioned
// at split_cut).  Pass A touches nothing but P / part_sl — in the pipelined data-parallel schedule the coldest slice of the
// previous step's gradient is still being exchanged in place while it runs — pass B is where the step's state changes as after
// step_forward.  A then B = the forward, up to the order of the fp32 sums (A's terms first).
int step_forward_pass(fmhip_model_t m, fmhip_dataset_t d, int64_t b, int pass) {
    const BatchMeta &bm = d->batches[(size_t)b];
    if (d->split_cut < 0) return fail(FMHIP_ERR_INVALID, "the dataset's rows are not partitioned (fmhip_dataset_partition_rows)");
    if (m->Kp > 64) return fail(FMHIP_ERR_UNSUPPORTED, "the two-pass forward serves models of up to 64 padded factors (this one: %d)", m->Kp);
    TRY(ensure_workspace(m, d));
    TRY(m->part_sl.ensure((size_t)std::max<int64_t>(d->max_rows, 1) * 2));
    FwdArgs a = fwd_args(m, d, bm);
    a.col = d->col_part.p;               // the partitioned copy of the stream (the dataset's own never moves)
    a.val = d->val_part.p;
    a.row_split = d->row_split.p;
    a.part_sl = m->part_sl.p;
    // pass A runs while the parameter rows at or above the cut are still being exchanged: a hot feature up there (ids not
    // ranked by frequency) moves the dense block's prologue into pass B — a local choice, no collective depends on it
    a.hot_in_b = d->hot_T > 0 && (int64_t)d->hot0_max_id >= d->split_cut;
    if (pass == 0) {
        a.bsum = nullptr;
        ProfScope ps(m, FMHIP_K_FORWARD, bm.nnz_total, bm.rows);
        HIP_TRY(launch_forward(m->Kp, kFwdPartA, a, m->stream, nullptr));
        return FMHIP_OK;
    }
    if (m->grad_dirty) {
        HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
        m->grad_dirty = false;
    }
    {
        ProfScope ps(m, FMHIP_K_FORWARD, 0, bm.rows);
        HIP_TRY(launch_forward(m->Kp, kFwdPartB, a, m->stream, &m->fwd_parts));
    }
    m->grad_dirty = true;
    m->last_nnz = bm.nnz_total;
    m->last_rows = bm.rows;
    m->bw_next_hi = INT64_MAX;
    m->hot_pending = d->hot_T > 0;
    return FMHIP_OK;
}

// gradient rows of the dense hot block (whole batch; they do not depend on the feature interval, so
// the first backward call of a step forms them and every later interval finds them complete): the
// work rides in that call's backward and fixup launches
void hot_attach(fmhip_model_t m, fmhip_dataset_t d, const BatchMeta &bm, BwdArgs &ba) {
    if (!m->hot_pending) return;
    HotArgs &h = ba.hot;
    h.P = m->P.p;
    h.e = m->e.p;
    h.xhot = d->xhot.p + (size_t)bm.row0 * kHotT;
    h.page_stride = std::max<int64_t>(d->n_rows, 1) * kHotT;
    h.pages = d->hot_pages;
    h.hot_ids = m->view ? m->view->hot_pos : d->d_hot_ids.p;
    h.part = m->hot_part.p;
    h.GV = ba.GV;
    h.Gw = ba.Gw;
    h.Gb = ba.Gb;
    h.n_rows = (int32_t)bm.rows;
    h.pack_k = m->pack_k();
    h.nblk = hot_blocks(m->Kp, bm.rows);
    h.upd = ba.upd;
    ba.hot_blocks = h.nblk;
    m->hot_pending = false;
}

// backward + fixup of the columns whose feature id lies in [feat_lo, feat_hi) into the packed
// gradient.  The CSC stream is sorted by feature, so the interval is a contiguous run of entries;
// the range holding its first entry is walked by THIS call in full (the entries of lower features
// in it produce G rows / head partials that the call covering them consumes later), the range
// holding the first entry of feat_hi is left to the call that covers feat_hi.  `finish` adds the
// residual-statistics reduction (once per step, with the last interval).
// `own` (kOwnLower | kOwnUpper): which straddling ranges this call walks.  A range that straddles two intervals goes with the
// one walked FIRST — descending callers own their lower edge (the rule above), ascending callers their upper one, and the
// pipelined data-parallel schedule (second-coldest interval first, then down to feature 0, the coldest last) says it per call.
// What a range does never depends on which call walks it, so the gradient is the same bits in any order.
int step_backward(fmhip_model_t m, fmhip_dataset_t d, int64_t b, int64_t feat_lo, int64_t feat_hi, bool finish,
                  double *acc, const FusedPlan *fused, int own) {
    const BatchMeta &bm = d->batches[(size_t)b];
    BwdArgs ba = bwd_args(m, d, b);
    if (fused) {
        if (fused->mode == 1) ba.upd = fused->upd;
        if (finish) { ba.red_w0 = m->w0.p; ba.red_eta = (float)fused->eta; ba.red_reg0 = (float)fused->reg0; }
    }
    const bool whole = feat_lo <= 0 && feat_hi >= m->n1;
    // the block product rides in the first call whose interval reaches down to the highest hot id (callers that cut the
    // backward go from the top down: every hot row is complete before the interval holding it is exchanged, and the
    // cold intervals in front — whose exchange the rest of the backward hides — are not held up by it)
    if (d->rb_rows > 0 && !whole)      // refused before anything of the step's state (hot_pending) is consumed
        return fail(FMHIP_ERR_UNSUPPORTED, "feature-interval backward is not available on a row-blocked dataset");
    if (whole || finish || d->hot_max_id >= feat_lo) hot_attach(m, d, bm, ba);
    // band-affine placement (FMHIP_TUNE_XCD_PLACEMENT = 2): XCD x walks the ranges of its own row bands first (BwdArgs::xlist); the same
    // choice for every launch of a step — the partials of a cut column are written and read under one rule (no wave sums)
    const bool banded = m->tv(kTuneXcd) == 2 && bm.xoff[0] >= 0 && d->rb_rows == 0;
    if (banded) {
        ba.xlist = d->xlist.p;
        ba.no_wave_sum = 1;
        ba.xcd_chunk = 0;
    }
    if (whole) {   // the common case needs no host-side searches
        if (banded)
            for (int x = 0; x < kXcds; ++x)
                for (int sg = 0; sg < kXSegs; ++sg) {
                    ba.xseg_off[x][sg] = (int32_t)bm.xoff[x] + bm.xseg[x][sg];
                    ba.xseg_len[x][sg] = bm.xseg[x][sg + 1] - bm.xseg[x][sg];
                }
        if (finish) {
            ba.red_bsum = m->bsum.p;
            ba.red_nblocks = m->fwd_parts;
            ba.red_rows = (int32_t)bm.rows;
            ba.red_scal = m->view ? m->view->scal : m->scal();
            ba.red_acc = acc;
        }
        {
            ProfScope ps(m, FMHIP_K_BACKWARD, bm.nnz_total, bm.rows);
            HIP_TRY(launch_backward(m->Kp, ba, m->stream));
        }
        if (fused && fused->mode == 2) {
            // merged finish: the fixup launch also updates the parameters (its own rows from registers, the rest in
            // extra workgroups beside it); the column walk above stored its gradient rows as usual
            ApplyArgs &f = ba.fin;
            f.V = m->V.p;
            f.w = m->w.p;
            f.w0 = m->w0.p;
            f.GV = m->GV();
            f.Gw = m->Gw();
            f.Gb = m->Gb();
            f.scal = m->scal();
            f.rows = m->scal() + 2;
            f.n1 = m->n1;
            f.row_lo = 0;
            f.row_hi = m->n1;
            f.do_w0 = 0;                                   // the statistics block steps w0 (red_w0)
            f.pack_k = m->pack_k();
            f.eta = (float)fused->eta;
            f.reg0 = (float)fused->reg0;
            f.regw = (float)fused->regw;
            f.regv = (float)fused->regv;
            f.sv_in = (float)m->sv;
            f.sw_in = (float)m->sw;
            f.eta_v = f.eta_w = f.eta;
            f.invb_val = fused->upd.invb;
            f.use_invb_val = 1;
            int64_t blocks = (m->n1 * (m->Kp / 4) + 255) / 256;
            ba.fin_blocks = (int32_t)std::min<int64_t>(std::max<int64_t>(blocks, 1), FMHIP_FIN_BLOCKS);
            ba.fin_own = d->own_bits.p + bm.own_off;
            ba.fin_own_bits = (int32_t)std::min<int64_t>(d->own_words * 32, INT32_MAX);
        }
        {
            ProfScope ps(m, FMHIP_K_FIXUP, bm.nnz_total, bm.rows);
            HIP_TRY(launch_fixup(m->Kp, ba, m->stream));
            HIP_TRY(launch_fixup2(m->Kp, ba, m->stream));
        }
        return FMHIP_OK;
    }
    const int32_t *hf = d->h_cfeat.data() + bm.col_off, *hp = d->h_cptr.data() + bm.col_off + b;
    const int32_t *hs = d->h_split.data() + bm.split_off;
    const int32_t s_lo = (int32_t)(std::lower_bound(hf, hf + bm.n_cols, (int32_t)std::min<int64_t>(feat_lo, INT32_MAX)) - hf);
    const int32_t s_hi = (int32_t)(std::lower_bound(hf, hf + bm.n_cols, (int32_t)std::min<int64_t>(feat_hi, INT32_MAX)) - hf);
    const int32_t e_lo = hp[s_lo], e_hi = hp[s_hi];            // entry interval of the columns
    // lower edge owned: from the range holding entry e_lo; else from the first range that starts at or after it (the range
    // holding entry e_lo - 1 goes with the interval below).  Upper edge likewise.
    ba.rho_lo = (own & kOwnLower) ? e_lo / kRangeLen : (e_lo + kRangeLen - 1) / kRangeLen;
    ba.rho_hi = s_hi >= bm.n_cols ? bm.n_ranges : ((own & kOwnUpper) ? (e_hi + kRangeLen - 1) / kRangeLen : e_hi / kRangeLen);
    if (ba.rho_hi < ba.rho_lo) ba.rho_hi = ba.rho_lo;                  // (an interval inside one range that a neighbour owns)
    if (banded) {
        // every run of the plan holds ascending range ids: the part of it inside [rho_lo, rho_hi) is one sub-run
        for (int x = 0; x < kXcds; ++x)
            for (int sg = 0; sg < kXSegs; ++sg) {
                const int32_t *r0 = d->h_xlist.data() + bm.xoff[x] + bm.xseg[x][sg], *r1 = d->h_xlist.data() + bm.xoff[x] + bm.xseg[x][sg + 1];
                const int32_t *lo = std::lower_bound(r0, r1, ba.rho_lo), *hi = std::lower_bound(r0, r1, ba.rho_hi);
                ba.xseg_off[x][sg] = (int32_t)(lo - d->h_xlist.data());
                ba.xseg_len[x][sg] = (int32_t)(hi - lo);
            }
    }
    const int32_t sp_lo = (int32_t)(std::lower_bound(hs, hs + bm.n_split, s_lo) - hs);
    const int32_t sp_hi = (int32_t)(std::lower_bound(hs, hs + bm.n_split, s_hi) - hs);
    ba.split_seg += sp_lo;
    ba.n_split = sp_hi - sp_lo;
    {
        const int32_t *hss = d->h_split_short.data() + bm.split_short_off;
        const int32_t q_lo = (int32_t)(std::lower_bound(hss, hss + bm.n_split_short, s_lo) - hss);
        const int32_t q_hi = (int32_t)(std::lower_bound(hss, hss + bm.n_split_short, s_hi) - hss);
        ba.split_short += q_lo;
        ba.n_split_short = q_hi - q_lo;
    }
    if (finish) {
        ba.red_bsum = m->bsum.p;
        ba.red_nblocks = m->fwd_parts;
        ba.red_rows = (int32_t)bm.rows;
        ba.red_scal = m->view ? m->view->scal : m->scal();
        ba.red_acc = acc;
    }
    const int64_t nnz_part = (int64_t)e_hi - e_lo;
    {
        ProfScope ps(m, FMHIP_K_BACKWARD, nnz_part, bm.rows);
        HIP_TRY(launch_backward(m->Kp, ba, m->stream));
    }
    {
        ProfScope ps(m, FMHIP_K_FIXUP, nnz_part, bm.rows);
        HIP_TRY(launch_fixup(m->Kp, ba, m->stream));
    }
    return FMHIP_OK;
}

// forward + backward + fixup of one batch into the packed gradient (fused: straight into the parameters)
int step_compute(fmhip_model_t m, fmhip_dataset_t d, int64_t b, double *acc, const FusedPlan *fused) {
    TRY(step_forward(m, d, b));
    return step_backward(m, d, b, 0, INT64_MAX, true, acc, fused);
}

// Can this step apply its gradient rows inside the backward (no exchange, no separate update launch)?  It is the
// rows-only update, so weight decay must be expressible through the tables' scale (lazy decay, fm_apply.hip).
bool plan_fused(fmhip_model_t m, fmhip_dataset_t d, int64_t b, double eta, double reg0, double regw, double regv, FusedPlan *p) {
    const double dv = 1.0 - eta * regv, dw = 1.0 - eta * regw;
    const bool decay = regw != 0.0 || regv != 0.0;
    const bool lazy_ok = !decay || (m->tv(kTuneLazy) && dv >= 0.5 && dw >= 0.5 && dv <= 1.0 && dw <= 1.0);
    const BatchMeta &bm0 = d->batches[(size_t)b];
    p->eta = eta;
    p->reg0 = reg0;
    p->regw = regw;
    p->regv = regv;
    {
        const float rows = (float)bm0.rows;
        p->upd.invb = rows > 0.f ? 1.0f / rows : 0.f;
    }
    // merged finish (FMHIP_TUNE_MERGED_FINISH): when the step's update is the DENSE pass (the batch touches most of the model, or decay
    // cannot ride in the scale) it runs inside the fixup launch, beside the fixups, instead of as a launch of its own
    const int64_t touched = (int64_t)bm0.n_cols + d->hot_pages * kHotT;
    const bool rows_only = lazy_ok && touched * 2 <= m->n1;
    if (m->tv(kTuneMerged) && !m->tv(kTuneFused) && d->rb_rows == 0 && !rows_only && bm0.own_off >= 0 && d->dimension <= m->n) {
        p->mode = 2;
        p->sv_out = p->sw_out = 1.0;      // the dense pass folds the scale
        return true;
    }
    if (!m->tv(kTuneFused) || d->rb_rows != 0) return false;
    if (!lazy_ok) return false;
    p->mode = 1;
    p->sv_out = m->sv * dv;
    p->sw_out = m->sw * dw;
    p->upd.V = m->V.p;
    p->upd.w = m->w.p;
    p->upd.sv = (float)m->sv;
    p->upd.eta_v = (float)(eta / p->sv_out);
    p->upd.eta_w = (float)(eta / p->sw_out);
    return true;
}

// brings lazily decayed tables back to scale 1 (dense pass)
int fold_scales(fmhip_model_t m) {
    if (m->sv == 1.0 && m->sw == 1.0) return FMHIP_OK;
    HIP_TRY(launch_rescale(m->Kp, m->V.p, m->w.p, m->n1, m->pack_k(), (float)m->sv, (float)m->sw, m->stream));
    m->sv = m->sw = 1.0;
    return FMHIP_OK;
}

// what step_apply leaves behind, for a step whose update already happened inside the backward
int finish_fused(fmhip_model_t m, const FusedPlan &p) {
    m->sv = p.sv_out;
    m->sw = p.sw_out;
    if (m->sv < 0x1p-24 || m->sw < 0x1p-24) TRY(fold_scales(m));
    m->grad_dirty = false;        // nothing but the statistics head was written
    m->host64_fresh = false;
    ++m->prof_step;
    return FMHIP_OK;
}

// `d`/`b` given: the gradient in the buffer is exactly batch b's (no exchange happened), so the update
// may be restricted to the rows that batch touched — their decay, and everyone else's, rides in the
// tables' scale (lazy weight decay, fm_apply.hip).  Otherwise the dense pass, which also folds a pending
// scale back to 1.
int step_apply(fmhip_model_t m, double eta, double reg0, double regw, double regv, fmhip_dataset_t d, int64_t b) {
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    double sv_out = 1.0, sw_out = 1.0;
    const double dv = 1.0 - eta * regv, dw = 1.0 - eta * regw;
    const bool decay = regw != 0.0 || regv != 0.0;
    if (d && b >= 0 && d->rb_rows == 0 && (!decay || (m->tv(kTuneLazy) && dv >= 0.5 && dw >= 0.5 && dv <= 1.0 && dw <= 1.0))) {
        const BatchMeta &bm = d->batches[(size_t)b];
        const int64_t touched = (int64_t)bm.n_cols + d->hot_pages * kHotT;
        if (touched * 2 <= m->n1) {     // otherwise the dense, perfectly coalesced pass is as cheap
            a.rows_only = 1;
            a.feat = d->cfeat.p + bm.col_off;
            a.n_feat = bm.n_cols;
            a.hot_ids = d->d_hot_ids.p;
            a.n_hot = d->hot_pages * kHotT;
            sv_out = m->sv * dv;
            sw_out = m->sw * dw;
        }
    }
    a.eta_v = (float)(eta / sv_out);
    a.eta_w = (float)(eta / sw_out);
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    a.scal = m->scal();
    a.rows = m->scal() + 2;
    a.n1 = m->n1;
    a.row_lo = 0;
    a.row_hi = m->n1;
    a.do_w0 = 1;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    {
        ProfScope ps(m, FMHIP_K_APPLY, m->last_nnz, m->last_rows);
        HIP_TRY(launch_apply(m->Kp, a, m->stream));
    }
    m->sv = sv_out;
    m->sw = sw_out;
    // fp32 tables lose nothing to a small scale until their values approach the denormal range; fold long before
    if (m->sv < 0x1p-24 || m->sw < 0x1p-24) TRY(fold_scales(m));
    m->grad_dirty = false;
    m->host64_fresh = false;
    ++m->prof_step;
    return FMHIP_OK;
}

// The dense update of the feature rows [lo, hi) only — the data-parallel step applies an interval as soon as its
// slice of the gradient has been exchanged (fmhip_comm.hip).  `rows`: device float holding the global row count;
// `last`: the final interval of the step (also steps w0 from the head's scalars and closes the step's bookkeeping).
int step_apply_interval(fmhip_model_t m, double eta, double reg0, double regw, double regv, int64_t lo, int64_t hi,
                        const float *rows, bool last) {
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    a.eta_v = a.eta_w = (float)eta;
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    a.scal = m->scal();
    a.rows = rows;
    a.n1 = m->n1;
    a.row_lo = lo;
    a.row_hi = hi;
    a.do_w0 = last ? 1 : 0;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    if (hi > lo || last) {
        ProfScope ps(m, FMHIP_K_APPLY, m->last_nnz, m->last_rows);
        HIP_TRY(launch_apply(m->Kp, a, m->stream));
    }
    if (last) {
        m->sv = m->sw = 1.0;      // every interval folded the pending scale
        m->grad_dirty = false;
        m->host64_fresh = false;
        ++m->prof_step;
    }
    return FMHIP_OK;
}

int step_apply_shard(fmhip_model_t m, double eta, double reg0, double regw, double regv, int64_t lo, int64_t hi, int64_t hi_r,
                     int64_t vlo, int64_t vhi, const float *rows, bool last, hipStream_t s) {
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    a.eta_v = a.eta_w = (float)eta;
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = m->GV();
    a.Gw = m->Gw();
    a.Gb = m->Gb();
    a.scal = m->scal();
    a.rows = rows;
    a.n1 = m->n1;
    hi = std::min(hi, m->n1);
    a.row_lo = std::min(std::max(vlo, lo), hi);
    a.row_hi = std::min(std::max(vhi, a.row_lo), hi);
    a.w_lo = lo;
    a.w_hi = hi;
    a.z_hi = std::max(hi_r, hi);
    a.do_w0 = last ? 1 : 0;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    HIP_TRY(launch_apply_shard(m->Kp, a, s));
    if (last) {
        m->sv = m->sw = 1.0;      // every share folded the pending scale; the all-gather spreads the folded rows
        m->host64_fresh = false;
        ++m->prof_step;
    }
    return FMHIP_OK;
}

// can weight decay ride in the tables' scale for this (eta, reg)?  (no decay at all: trivially)
bool lazy_decay_ok(fmhip_model_t m, double eta, double regw, double regv) {
    const double dv = 1.0 - eta * regv, dw = 1.0 - eta * regw;
    if (regw == 0.0 && regv == 0.0) return true;
    return m->tv(kTuneLazy) && dv >= 0.5 && dw >= 0.5 && dv <= 1.0 && dw <= 1.0;
}

int step_apply_rows(fmhip_model_t m, double eta, double reg0, double regw, double regv, const int32_t *feat, int32_t n_feat,
                    const float *rows, const GradView *view, int64_t off, bool last) {
    if (!lazy_decay_ok(m, eta, regw, regv))
        return fail(FMHIP_ERR_UNSUPPORTED, "a rows-only update needs weight decay that fits the tables' scale (0.5 <= 1 - eta*reg <= 1)");
    // every slice of a step starts from the scale the step began with (m->sv / m->sw move with the LAST slice only)
    const double sv_out = m->sv * (1.0 - eta * regv), sw_out = m->sw * (1.0 - eta * regw);
    ApplyArgs a{};
    a.sv_in = (float)m->sv;
    a.sw_in = (float)m->sw;
    a.rows_only = 1;
    a.feat = feat + off;
    a.n_feat = n_feat;
    a.hot_ids = nullptr;
    a.n_hot = 0;
    a.eta_v = (float)(eta / sv_out);
    a.eta_w = (float)(eta / sw_out);
    a.V = m->V.p;
    a.w = m->w.p;
    a.w0 = m->w0.p;
    a.GV = (view ? view->GV : m->GV()) + (view ? (size_t)off * m->Kp : 0);
    a.Gw = (view ? view->Gw : m->Gw()) + (view ? off : 0);
    a.Gb = (view ? view->Gb : m->Gb()) + (view ? off : 0);
    a.scal = view ? view->scal : m->scal();
    a.g_compact = view ? 1 : 0;
    a.rows = rows;
    a.n1 = m->n1;
    a.row_lo = 0;
    a.row_hi = m->n1;
    a.do_w0 = last ? 1 : 0;
    a.pack_k = m->pack_k();
    a.eta = (float)eta;
    a.reg0 = (float)reg0;
    a.regw = (float)regw;
    a.regv = (float)regv;
    if (n_feat > 0 || last) {
        ProfScope ps(m, FMHIP_K_APPLY, m->last_nnz, m->last_rows);
        HIP_TRY(launch_apply(m->Kp, a, m->stream));
    }
    if (!last) return FMHIP_OK;
    m->sv = sv_out;
    m->sw = sw_out;
    if (m->sv < 0x1p-24 || m->sw < 0x1p-24) TRY(fold_scales(m));
    m->grad_dirty = false;
    m->host64_fresh = false;
    ++m->prof_step;
    return FMHIP_OK;
}

int read_scal(fmhip_model_t m, fmhip_stats *st) {
    float h[4];
    HIP_TRY(hipMemcpyAsync(h, m->scal(), sizeof h, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    st->sum_e = h[0];
    st->sse = h[1];
    st->rows = (int64_t)llround(h[2]);
    st->nonfinite = (int64_t)llround(h[3]);
    return FMHIP_OK;
}

int read_acc(fmhip_model_t m, fmhip_stats *st) {
    double h[4];
    HIP_TRY(hipMemcpyAsync(h, m->acc.p, sizeof h, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    st->sum_e = h[0];
    st->sse = h[1];
    st->rows = (int64_t)llround(h[2]);
    st->nonfinite = (int64_t)llround(h[3]);
    return FMHIP_OK;
}

}  // namespace host
}  // namespace fmhip
