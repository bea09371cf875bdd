// fmhip_api.hip — the C ABI declared in include/fmhip.h: library / model entry points, scoring, the training calls, the
// reference's ALS learner, the split step a host-orchestrated exchange drives, profiling.  Plumbing only:
//   fmhip_dataset.hip   datasets (host passes, device transposes, layout queries, feature relabelling)
//   fmhip_step.hip      the launch sequence of one mini-batch step
//   fmhip_comm.hip      the data-parallel step (RCCL / caller's transport)
//   fm_forward / fm_backward / fm_apply / als_kernels / csc_build .hip   the kernels
#include "fmhip_internal.h"
#include "als_kernels.h"

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

namespace fmhip {
namespace host {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace host
}  // namespace fmhip

using namespace fmhip::host;

namespace fmhip {
namespace host {

int set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return FMHIP_OK;
}

template <typename FT>
int set_params_impl(fmhip_model_t m, FT w0, const FT *w, const FT *v) {
    if (!m || !w || !v) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(m->device));
    std::vector<float> hV((size_t)m->n1p * m->Kp, 0.f), hw((size_t)m->n1p, 0.f);
    for (int64_t i = 0; i < m->n1; ++i) {
        hw[(size_t)i] = (float)w[i];
        for (int f = 0; f < m->k; ++f) hV[(size_t)i * m->Kp + f] = (float)v[f + i * (int64_t)m->k];
        if (m->pack_k() >= 0) hV[(size_t)i * m->Kp + m->k] = (float)w[i];   // packed rows: w_i rides in slot k
    }
    const float hw0 = (float)w0;
    m->h_w0 = (double)w0;
    m->h_w.assign((size_t)m->n1, 0.0);
    m->h_v.assign((size_t)m->n1 * m->k, 0.0);
    for (int64_t i = 0; i < m->n1; ++i) m->h_w[(size_t)i] = (double)w[i];
    for (int64_t j = 0; j < m->n1 * m->k; ++j) m->h_v[(size_t)j] = (double)v[j];
    m->host64_fresh = true;
    m->sv = m->sw = 1.0;
    HIP_TRY(hipMemcpyAsync(m->V.p, hV.data(), hV.size() * sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->w.p, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->w0.p, &hw0, sizeof(float), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return FMHIP_OK;
}

template <typename FT>
int get_params_impl(fmhip_model_t m, FT *w0, FT *w, FT *v) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    if (m->host64_fresh) {   // nothing has trained in fp32 since the masters were written: return them exactly
        if (w0) *w0 = (FT)m->h_w0;
        if (w) for (int64_t i = 0; i < m->n1; ++i) w[i] = (FT)m->h_w[(size_t)i];
        if (v) for (int64_t j = 0; j < m->n1 * m->k; ++j) v[j] = (FT)m->h_v[(size_t)j];
        return FMHIP_OK;
    }
    std::vector<float> hV((size_t)m->n1p * m->Kp), hw((size_t)m->n1p);
    float hw0 = 0.f;
    HIP_TRY(hipMemcpyAsync(hV.data(), m->V.p, hV.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hw.data(), m->w.p, hw.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(&hw0, m->w0.p, sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (w0) *w0 = (FT)hw0;
    // a lazily decayed model stores U with V = sv*U (fm_apply.hip); with sv = sw = 1 the products are exact
    for (int64_t i = 0; i < m->n1; ++i) {
        if (w) w[i] = (FT)((double)(m->pack_k() >= 0 ? hV[(size_t)i * m->Kp + m->k] : hw[(size_t)i]) * m->sw);
        if (v)
            for (int f = 0; f < m->k; ++f) v[f + i * (int64_t)m->k] = (FT)((double)hV[(size_t)i * m->Kp + f] * m->sv);
    }
    return FMHIP_OK;
}

}  // namespace host
}  // namespace fmhip

// =================================================================== C ABI

extern "C" {

int fmhip_version(void) { return FMHIP_VERSION; }

int fmhip_ablation_mask(void) { return forward_ablations() | backward_ablations(); }

const char *fmhip_last_error(void) { return g_err.c_str(); }

// enum fmhip_tune_key (include/fmhip_experimental.h) names the kernels' own keys (fm_kernels.h)
static_assert(FMHIP_TUNE_FORWARD_KERNEL == kTuneFwd && FMHIP_TUNE_BACKWARD_KERNEL == kTuneBwd && FMHIP_TUNE_TILE_ROWS == kTuneTile &&
              FMHIP_TUNE_ROW_BLOCK == kTuneRowBlock && FMHIP_TUNE_XCD_PLACEMENT == kTuneXcd && FMHIP_TUNE_HOT_BLOCK == kTuneHot &&
              FMHIP_TUNE_FORWARD_OCCUPANCY == kTuneFwdOcc && FMHIP_TUNE_ROW_ORDER == kTuneRowOrder && FMHIP_TUNE_FLAT_ADDRESS == kTuneFlat &&
              FMHIP_TUNE_LAZY_DECAY == kTuneLazy && FMHIP_TUNE_FUSED_UPDATE == kTuneFused && FMHIP_TUNE_MERGED_FINISH == kTuneMerged &&
              FMHIP_TUNE_HOT_PAGES == kTuneHotPages && FMHIP_TUNE_KEY_COUNT == kTuneCount, "fmhip_experimental.h and fm_kernels.h disagree on the tuning keys");

int fmhip_tune(int key, int value) {
    if (key < 0 || key >= kTuneCount) return fail(FMHIP_ERR_INVALID, "unknown tuning key %d", key);
    g_tune[key] = value;
    return FMHIP_OK;
}

int fmhip_model_tune(fmhip_model_t m, int key, int value) {
    WriteLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (key < 0 || key >= kTuneCount) return fail(FMHIP_ERR_INVALID, "unknown tuning key %d", key);
    if (key == kTuneRowBlock || key == kTuneHot || key == kTuneHotPages)
        return fail(FMHIP_ERR_INVALID, "tuning key %d decides a DATASET's layout: state it in fmhip_dataset_opts (or the process default, fmhip_tune)", key);
    m->tune[key] = value < 0 ? -1 : value;
    return FMHIP_OK;
}

int fmhip_device_count(int *count) {
    if (!count) return fail(FMHIP_ERR_INVALID, "count is NULL");
    *count = 0;
    HIP_TRY(hipGetDeviceCount(count));
    return FMHIP_OK;
}

int fmhip_model_create(int device, int64_t num_attribute, int32_t num_factor, void *stream, fmhip_model_t *out) {
    if (!out) return fail(FMHIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (num_attribute < 0 || num_attribute >= ((int64_t)1 << 31) - 8)
        return fail(FMHIP_ERR_INVALID, "num_attribute %lld out of range", (long long)num_attribute);
    if (num_factor < 1) return fail(FMHIP_ERR_INVALID, "num_factor must be >= 1");
    if (num_factor > FMHIP_MAX_FACTORS)
        return fail(FMHIP_ERR_UNSUPPORTED, "num_factor %d > FMHIP_MAX_FACTORS (%d)", num_factor, FMHIP_MAX_FACTORS);
    TRY(set_device(device));
    fmhip_model *m = new (std::nothrow) fmhip_model();
    if (!m) return fail(FMHIP_ERR_NOMEM, "out of host memory");
    m->device = device;
    for (int &t : m->tune) t = -1;          // every key follows the process-wide default until fmhip_model_tune says otherwise
    m->n = num_attribute;
    m->n1 = num_attribute + 1;
    m->n1p = (m->n1 + 3) & ~(int64_t)3;
    m->k = num_factor;
    m->Kp = padded_factors(num_factor);
    if (stream) {
        m->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete m;
            return fail(FMHIP_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        m->own_stream = true;
    }
    int rc;
    const size_t slack = (size_t)fmhip_model::kSlackRows * m->Kp;       // zero rows behind the tables (sharded exchange)
    if ((rc = m->V.alloc((size_t)m->n1p * m->Kp + slack)) || (rc = m->w.alloc((size_t)m->n1p)) || (rc = m->w0.alloc(1)) ||
        (rc = m->grad_own.alloc(m->grad_floats() + slack)) || (rc = m->acc.alloc(4))) {
        fmhip_model_destroy(m);
        return rc;
    }
    m->grad = m->grad_own.p;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMemsetAsync(m->V.p, 0, m->V.n * sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->w.p, 0, m->w.n * sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->w0.p, 0, sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->grad, 0, m->grad_own.n * sizeof(float), m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->acc.p, 0, 4 * sizeof(double), m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
        fmhip_model_destroy(m);
        return fail(FMHIP_ERR_HIP, "zero-initialisation failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return FMHIP_OK;
}

int fmhip_model_destroy(fmhip_model_t m) {
    if (!m) return FMHIP_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (auto &r : m->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return FMHIP_OK;
}

int fmhip_model_info(fmhip_model_t m, int64_t *num_attribute, int32_t *num_factor, int32_t *padded) {
    ReadLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (num_attribute) *num_attribute = m->n;
    if (num_factor) *num_factor = m->k;
    if (padded) *padded = m->Kp;
    return FMHIP_OK;
}

int fmhip_model_init_normal(fmhip_model_t m, uint64_t seed, double mean, double stdev) {
    WriteLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    HIP_TRY(launch_init_normal(m->Kp, m->V.p, m->w.p, m->w0.p, m->n1, m->n1p, m->k, seed, (float)mean, (float)stdev, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->sv = m->sw = 1.0;
    m->host64_fresh = false;     // the device holds the parameters; the fp64 masters are refreshed on demand
    std::vector<double>().swap(m->h_w);
    std::vector<double>().swap(m->h_v);
    return FMHIP_OK;
}

int fmhip_model_get_rows(fmhip_model_t m, int64_t n, const int32_t *ids, double *w, double *v) {
    ReadLock lock(m);
    if (!m || n < 0 || (n > 0 && !ids)) return fail(FMHIP_ERR_INVALID, "NULL argument or negative count");
    for (int64_t j = 0; j < n; ++j)
        if (ids[j] < 0 || ids[j] > m->n) return fail(FMHIP_ERR_SHAPE, "feature id %d outside [0, %lld]", ids[j], (long long)m->n);
    if (n == 0) return FMHIP_OK;
    if (m->host64_fresh) {   // nothing has trained in fp32 since the masters were written: the exact fp64 values, as get_params
        for (int64_t j = 0; j < n; ++j) {
            if (w) w[j] = m->h_w[(size_t)ids[j]];
            if (v)
                for (int f = 0; f < m->k; ++f) v[f + j * (int64_t)m->k] = m->h_v[(size_t)f + (size_t)ids[j] * m->k];
        }
        return FMHIP_OK;
    }
    TRY(set_device(m->device));
    DevBuf<int32_t> dids;
    DevBuf<float> dv, dw;
    TRY(dids.alloc((size_t)n));
    TRY(dv.alloc((size_t)n * m->Kp));
    TRY(dw.alloc((size_t)n));
    HIP_TRY(hipMemcpyAsync(dids.p, ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(launch_gather_rows(m->Kp, m->V.p, m->w.p, dids.p, n, dv.p, dw.p, m->stream));
    std::vector<float> hv((size_t)n * m->Kp), hw((size_t)n);
    HIP_TRY(hipMemcpyAsync(hv.data(), dv.p, hv.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hw.data(), dw.p, hw.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (int64_t j = 0; j < n; ++j) {       // scales of a lazily decayed model, packed rows: as get_params
        if (w) w[j] = (double)(m->pack_k() >= 0 ? hv[(size_t)j * m->Kp + m->k] : hw[(size_t)j]) * m->sw;
        if (v)
            for (int f = 0; f < m->k; ++f) v[f + j * (int64_t)m->k] = (double)hv[(size_t)j * m->Kp + f] * m->sv;
    }
    return FMHIP_OK;
}

int fmhip_model_set_params(fmhip_model_t m, double w0, const double *w, const double *v) {
    WriteLock lock(m);
    return set_params_impl<double>(m, w0, w, v);
}
int fmhip_model_get_params(fmhip_model_t m, double *w0, double *w, double *v) {
    ReadLock lock(m);
    return get_params_impl<double>(m, w0, w, v);
}
int fmhip_model_set_params_f32(fmhip_model_t m, float w0, const float *w, const float *v) {
    WriteLock lock(m);
    return set_params_impl<float>(m, w0, w, v);
}
int fmhip_model_get_params_f32(fmhip_model_t m, float *w0, float *w, float *v) {
    ReadLock lock(m);
    return get_params_impl<float>(m, w0, w, v);
}

int fmhip_synchronize(fmhip_model_t m) {
    ReadLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return FMHIP_OK;
}

// ---- scoring

// A scoring call's workspace: taken from the model's pool (or made), given back when the call returns.
namespace {
struct ScoreLease {
    fmhip_model_t m;
    ScoreCtx *cx = nullptr;
    explicit ScoreLease(fmhip_model_t m_) : m(m_) {}
    int take() {
        {
            std::lock_guard<std::mutex> g(m->pool_mu);
            if (!m->ctx_free.empty()) {
                cx = m->ctx_free.back();
                m->ctx_free.pop_back();
                return FMHIP_OK;
            }
        }
        std::unique_ptr<ScoreCtx> fresh(new (std::nothrow) ScoreCtx());
        if (!fresh) return fail(FMHIP_ERR_NOMEM, "out of host memory");
        HIP_TRY(hipStreamCreateWithFlags(&fresh->s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&fresh->ev, hipEventDisableTiming));
        std::lock_guard<std::mutex> g(m->pool_mu);
        cx = fresh.get();
        m->ctx_all.push_back(std::move(fresh));
        return FMHIP_OK;
    }
    ~ScoreLease() {
        if (!cx) return;
        (void)hipStreamSynchronize(cx->s);       // an early error return must not hand a busy workspace to the next call
        std::lock_guard<std::mutex> g(m->pool_mu);
        m->ctx_free.push_back(cx);
    }
};
}  // namespace

// One pass of FMModel.predict over a dataset's rows.  Re-entrant: the caller holds the model's lock SHARED, every call
// works on a stream and in buffers of its own (ScoreCtx) and touches nothing of the model but its parameters.
static int score_pass(fmhip_model_t m, fmhip_dataset_t d, double *yhat, double *e_out, double *q_out, fmhip_stats *st) {
    TRY(check_pair(m, d));
    ScoreLease lease(m);
    TRY(lease.take());
    ScoreCtx &cx = *lease.cx;
    const size_t rows_max = (size_t)std::max<int64_t>(d->max_rows, 1);
    TRY(cx.e.ensure(rows_max));
    TRY(cx.bsum.ensure((size_t)kMaxFwdBlocks * 4));
    TRY(cx.acc.ensure(4));
    if (yhat) TRY(cx.yhat.ensure(rows_max));
    if (q_out) TRY(cx.P.ensure(rows_max * m->Kp));
    // behind whatever the model's own stream still has queued (a training step returns before it has run)
    HIP_TRY(hipEventRecord(cx.ev, m->stream));
    HIP_TRY(hipStreamWaitEvent(cx.s, cx.ev, 0));
    HIP_TRY(hipMemsetAsync(cx.acc.p, 0, 4 * sizeof(double), cx.s));
    std::vector<float> hbuf;
    for (size_t b = 0; b < d->batches.size(); ++b) {
        const BatchMeta &bm = d->batches[b];
        FwdArgs a = fwd_args(m, d, bm);
        a.P = q_out ? cx.P.p : nullptr;          // the scoring modes write P only to hand q back
        a.e = cx.e.p;
        a.bsum = cx.bsum.p;
        a.yhat = yhat ? cx.yhat.p : nullptr;
        int parts = 0;
        HIP_TRY(launch_forward(m->Kp, q_out ? kFwdQ : kFwdResidual, a, cx.s, &parts));
        HIP_TRY(launch_reduce_blocks(cx.bsum.p, parts, (int32_t)bm.rows, nullptr, cx.acc.p, cx.s));
        if (yhat || e_out) {
            hbuf.resize((size_t)bm.rows);
            if (yhat) {
                HIP_TRY(hipMemcpyAsync(hbuf.data(), cx.yhat.p, (size_t)bm.rows * sizeof(float), hipMemcpyDeviceToHost, cx.s));
                HIP_TRY(hipStreamSynchronize(cx.s));
                for (int64_t r = 0; r < bm.rows; ++r) yhat[bm.row0 + r] = hbuf[(size_t)r];
            }
            if (e_out) {
                HIP_TRY(hipMemcpyAsync(hbuf.data(), cx.e.p, (size_t)bm.rows * sizeof(float), hipMemcpyDeviceToHost, cx.s));
                HIP_TRY(hipStreamSynchronize(cx.s));
                for (int64_t r = 0; r < bm.rows; ++r) e_out[bm.row0 + r] = hbuf[(size_t)r];
            }
        }
        if (q_out) {
            hbuf.resize((size_t)bm.rows * m->Kp);
            HIP_TRY(hipMemcpyAsync(hbuf.data(), cx.P.p, hbuf.size() * sizeof(float), hipMemcpyDeviceToHost, cx.s));
            HIP_TRY(hipStreamSynchronize(cx.s));
            for (int64_t r = 0; r < bm.rows; ++r)
                for (int f = 0; f < m->k; ++f) q_out[(bm.row0 + r) * m->k + f] = hbuf[(size_t)r * m->Kp + f];
        }
    }
    if (st) {
        memset(st, 0, sizeof *st);
        double h[4];
        HIP_TRY(hipMemcpyAsync(h, cx.acc.p, sizeof h, hipMemcpyDeviceToHost, cx.s));
        HIP_TRY(hipStreamSynchronize(cx.s));
        st->sum_e = h[0];
        st->sse = h[1];
        st->rows = (int64_t)llround(h[2]);
        st->nonfinite = (int64_t)llround(h[3]);
        st->nnz = d->nnz;
    }
    return FMHIP_OK;
}

int fmhip_predict(fmhip_model_t m, fmhip_dataset_t d, double *yhat) {
    ReadLock lock(m);
    if (!yhat) return fail(FMHIP_ERR_INVALID, "yhat is NULL");
    return score_pass(m, d, yhat, nullptr, nullptr, nullptr);
}

int fmhip_predict_rows(fmhip_model_t m, int64_t n_rows, const int64_t *row_ptr, const int32_t *col, const double *val,
                       double *yhat) {
    ReadLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (n_rows > 0 && !yhat) return fail(FMHIP_ERR_INVALID, "yhat is NULL");
    fmhip_dataset_t d = nullptr;
    TRY(fmhip_rows_create(m->device, n_rows, row_ptr, col, val, nullptr, &d));      // scoring-only upload (fmhip_dataset.hip)
    const int rc = n_rows > 0 ? score_pass(m, d, yhat, nullptr, nullptr, nullptr) : FMHIP_OK;
    fmhip_dataset_destroy(d);
    return rc;
}

int fmhip_residual(fmhip_model_t m, fmhip_dataset_t d, double *e) {
    ReadLock lock(m);
    if (!e) return fail(FMHIP_ERR_INVALID, "e is NULL");
    return score_pass(m, d, nullptr, e, nullptr, nullptr);
}

int fmhip_term_q(fmhip_model_t m, fmhip_dataset_t d, double *q) {
    ReadLock lock(m);
    if (!q) return fail(FMHIP_ERR_INVALID, "q is NULL");
    return score_pass(m, d, nullptr, nullptr, q, nullptr);
}

int fmhip_rmse(fmhip_model_t m, fmhip_dataset_t d, double *rmse, fmhip_stats *stats) {
    ReadLock lock(m);
    if (!rmse) return fail(FMHIP_ERR_INVALID, "rmse is NULL");
    fmhip_stats st;
    TRY(score_pass(m, d, nullptr, nullptr, nullptr, &st));
    // S/Model.scala:13-19: sqrt(sum (y - yhat)^2 / size); (y - yhat)^2 == e^2
    *rmse = st.rows > 0 ? std::sqrt(st.sse / (double)st.rows) : 0.0;
    if (stats) *stats = st;
    return FMHIP_OK;
}

// ---- training

int fmhip_sgd_step(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, double eta, double reg0, double regw,
                   double regv, fmhip_stats *stats) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    FusedPlan fp{};
    const bool fused = plan_fused(m, d, batch, eta, reg0, regw, regv, &fp);
    TRY(step_compute(m, d, batch, nullptr, fused ? &fp : nullptr));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        TRY(read_scal(m, stats));
        stats->nnz = d->batches[(size_t)batch].nnz_total;
        stats->steps = 1;
    }
    return fused ? finish_fused(m, fp) : step_apply(m, eta, reg0, regw, regv, d, batch);
}

int fmhip_sgd_epoch(fmhip_model_t m, fmhip_dataset_t d, double eta, double reg0, double regw, double regv,
                    const int64_t *order, fmhip_stats *stats) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    const int64_t nb = (int64_t)d->batches.size();
    if (order)
        for (int64_t j = 0; j < nb; ++j) TRY(check_batch(d, order[j]));
    HIP_TRY(hipMemsetAsync(m->acc.p, 0, 4 * sizeof(double), m->stream));
    for (int64_t j = 0; j < nb; ++j) {
        const int64_t b = order ? order[j] : j;
        FusedPlan fp{};
        const bool fused = plan_fused(m, d, b, eta, reg0, regw, regv, &fp);
        TRY(step_compute(m, d, b, m->acc.p, fused ? &fp : nullptr));
        TRY(fused ? finish_fused(m, fp) : step_apply(m, eta, reg0, regw, regv, d, b));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        TRY(read_acc(m, stats));
        stats->nnz = d->nnz;
        stats->steps = nb;
    }
    return FMHIP_OK;
}

int fmhip_batch_grad(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, double *gv, double *gw, double *gw0,
                     fmhip_stats *stats) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    TRY(step_compute(m, d, batch, nullptr));
    std::vector<float> hG(m->grad_floats()), hV((size_t)m->n1p * m->Kp);
    HIP_TRY(hipMemcpyAsync(hG.data(), m->grad, hG.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hV.data(), m->V.p, hV.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->grad_dirty = false;
    const float *sc = hG.data(), *Gw = sc + kGradHead, *Gb = Gw + m->n1p, *GV = sc + m->head_floats();
    for (int64_t i = 0; i < m->n1; ++i) {
        if (gw) gw[i] = m->pack_k() >= 0 ? GV[(size_t)i * m->Kp + m->k] : Gw[i];
        if (gv)
            for (int f = 0; f < m->k; ++f)
                gv[f + i * (int64_t)m->k] = (double)GV[(size_t)i * m->Kp + f] - (double)hV[(size_t)i * m->Kp + f] * m->sv * (double)Gb[i];
    }
    if (gw0) *gw0 = sc[0];
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->sum_e = sc[0];
        stats->sse = sc[1];
        stats->rows = (int64_t)llround(sc[2]);
        stats->nonfinite = (int64_t)llround(sc[3]);
        stats->nnz = d->batches[(size_t)batch].nnz_total;
    }
    return FMHIP_OK;
}

// ---- ALS (the reference's own learner), fp64

int fmhip_als_epoch(fmhip_model_t m, fmhip_dataset_t d, double reg0, double regw, double regv) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    if (d->batches.size() > 1 || (d->nnz > 0 && !d->val64.p))
        return fail(FMHIP_ERR_UNSUPPORTED, "ALS walks the whole-dataset transpose: create the dataset with batch_rows <= 0 "
                                           "(single batch, at most 2^27 stored nonzeros) and without asking for the dense hot "
                                           "block (fmhip_dataset_opts::hot_block <= 0)");
    if (d->rb_rows > 0) return fail(FMHIP_ERR_UNSUPPORTED, "ALS needs a dataset without row blocks (FMHIP_TUNE_ROW_BLOCK = 0)");
    if (d->als_dup)
        return fail(FMHIP_ERR_UNSUPPORTED, "ALS: a row stores the same feature index twice; the column walk updates every row of a "
                                           "column at once and needs the (row, feature) pairs to be distinct");
    if (!m->host64_fresh) {   // parameters last changed by fp32 SGD: start from their fp64 widening
        std::vector<double> w((size_t)m->n1), v((size_t)m->n1 * m->k);
        double w0 = 0.0;
        TRY(get_params_impl<double>(m, &w0, w.data(), v.data()));
        m->h_w0 = w0;
        m->h_w.swap(w);
        m->h_v.swap(v);
        m->host64_fresh = true;
    }
    const size_t n1 = (size_t)m->n1, nv = n1 * (size_t)m->k, nr = (size_t)std::max<int64_t>(d->n_rows, 1);
    TRY(m->als_w0.ensure(1));
    TRY(m->als_w.ensure(n1));
    TRY(m->als_v.ensure(nv));
    TRY(m->als_e.ensure(nr));
    TRY(m->als_q.ensure(nr * (size_t)m->k));
    TRY(m->als_part.ensure(2 * (size_t)kAlsMaxParts + 2));
    HIP_TRY(hipMemcpyAsync(m->als_w0.p, &m->h_w0, sizeof(double), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->als_w.p, m->h_w.data(), n1 * sizeof(double), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->als_v.p, m->h_v.data(), nv * sizeof(double), hipMemcpyHostToDevice, m->stream));
    if (d->n_rows > 0) {
        const BatchMeta &bm = d->batches[0];
        AlsArgs a{};
        a.k = m->k;
        a.num_attribute = m->n;
        a.n_rows = d->n_rows;
        a.nnz = d->nnz;
        a.row_ptr = d->row_ptr.p;
        a.col = d->col.p;
        a.val = d->val64.p;
        a.scol = d->scol.p;
        a.sval = d->sval64.p;
        a.y = d->y64.p;
        a.n_cols = bm.n_cols;
        a.cfeat = d->cfeat.p;
        a.cptr = d->cptr.p;
        a.crow = d->crow.p;
        a.cval = d->cval64.p;
        a.w0 = m->als_w0.p;
        a.w = m->als_w.p;
        a.v = m->als_v.p;
        a.reg0 = reg0;
        a.regw = regw;
        a.regv = regv;
        a.e = m->als_e.p;
        a.q = m->als_q.p;
        a.part = m->als_part.p;
        a.lev_cols = d->als_lev_cols.p;
        const int n_levels = d->als_lev_ptr.empty() ? 0 : (int)d->als_lev_ptr.size() - 1;
        HIP_TRY(launch_als_epoch(a, d->h_cfeat.data(), d->h_cptr.data(), n_levels ? d->als_lev_ptr.data() : nullptr, d->h_als_lev_cols.data(), n_levels,
                                 m->stream));
    }
    std::vector<double> w(n1), v(nv);
    double w0 = 0.0;
    HIP_TRY(hipMemcpyAsync(&w0, m->als_w0.p, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(w.data(), m->als_w.p, n1 * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(v.data(), m->als_v.p, nv * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return set_params_impl<double>(m, w0, w.data(), v.data());   // refreshes the fp64 masters and the fp32 device copy
}

// ---- data-parallel split step

int fmhip_grad_floats(fmhip_model_t m, int64_t *n_floats) {
    ReadLock lock(m);
    if (!m || !n_floats) return fail(FMHIP_ERR_INVALID, "NULL argument");
    *n_floats = (int64_t)m->grad_floats();
    return FMHIP_OK;
}

int fmhip_grad_bind(fmhip_model_t m, void *device_ptr) {
    WriteLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15u))
        return fail(FMHIP_ERR_INVALID, "gradient buffer must be 16-byte aligned");
    m->grad = device_ptr ? static_cast<float *>(device_ptr) : m->grad_own.p;
    m->grad_dirty = false;
    return FMHIP_OK;
}

int fmhip_grad_ptr(fmhip_model_t m, void **device_ptr) {
    ReadLock lock(m);
    if (!m || !device_ptr) return fail(FMHIP_ERR_INVALID, "NULL argument");
    *device_ptr = m->grad;
    return FMHIP_OK;
}

int fmhip_step_compute(fmhip_model_t m, fmhip_dataset_t d, int64_t batch) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    return step_compute(m, d, batch, nullptr);
}

int fmhip_step_forward(fmhip_model_t m, fmhip_dataset_t d, int64_t batch) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    return step_forward(m, d, batch);
}

int fmhip_step_forward_pass(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, int pass) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    if (pass != 0 && pass != 1) return fail(FMHIP_ERR_INVALID, "pass must be 0 (features below the cut) or 1 (the others, and the row's finish)");
    return step_forward_pass(m, d, batch, pass);
}

int fmhip_step_backward(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, int64_t feat_lo, int64_t feat_hi, int finish) {
    WriteLock lock(m);
    TRY(check_train(m, d));
    TRY(check_batch(d, batch));
    if (feat_lo < 0 || feat_hi < feat_lo) return fail(FMHIP_ERR_INVALID, "bad feature interval [%lld, %lld)", (long long)feat_lo, (long long)feat_hi);
    // intervals tile [0, n+1) in DESCENDING order (a range straddling two intervals is walked with the upper one, whose
    // partials the lower one's fixup then reads) or, from feature 0 up, in ASCENDING order (the mirror rule); the first call
    // after the forward says which: it ends at n+1 or starts at 0
    if (m->bw_next_hi < 0) return fail(FMHIP_ERR_INVALID, "fmhip_step_backward without fmhip_step_forward");
    if (finish && feat_lo != 0) return fail(FMHIP_ERR_INVALID, "finish = 1 belongs to the interval that starts at feature 0");
    if (m->bw_next_hi == INT64_MAX) m->bw_up = feat_hi < m->n1 && feat_lo == 0;
    if (m->bw_up) {
        const int64_t want = m->bw_next_hi == INT64_MAX ? 0 : m->bw_next_hi;
        if (feat_lo != want)
            return fail(FMHIP_ERR_INVALID, "feature intervals must tile [0, n+1) in ascending order (expected lo = %lld, got %lld)", (long long)want, (long long)feat_lo);
        TRY(step_backward(m, d, batch, feat_lo, feat_hi, finish != 0, nullptr, nullptr, kOwnUpper));
        m->bw_next_hi = feat_hi >= m->n1 ? -1 : feat_hi;
        return FMHIP_OK;
    }
    if (m->bw_next_hi == INT64_MAX ? feat_hi < m->n1 : feat_hi != m->bw_next_hi)
        return fail(FMHIP_ERR_INVALID, "feature intervals must tile [0, n+1) in descending order (expected hi = %lld, got %lld), or start at feature 0 and ascend",
                    (long long)(m->bw_next_hi == INT64_MAX ? m->n1 : m->bw_next_hi), (long long)feat_hi);
    TRY(step_backward(m, d, batch, feat_lo, feat_hi, finish != 0, nullptr, nullptr));
    m->bw_next_hi = feat_lo == 0 ? -1 : feat_lo;
    return FMHIP_OK;
}

int fmhip_grad_layout(fmhip_model_t m, int64_t *row_floats, int64_t *gv_offset) {
    ReadLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (row_floats) *row_floats = m->Kp;
    if (gv_offset) *gv_offset = (int64_t)m->head_floats();
    return FMHIP_OK;
}

int fmhip_step_apply(fmhip_model_t m, double eta, double reg0, double regw, double regv) {
    WriteLock lock(m);
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    TRY(set_device(m->device));
    return step_apply(m, eta, reg0, regw, regv);
}

int fmhip_step_stats(fmhip_model_t m, fmhip_stats *stats) {
    WriteLock lock(m);
    if (!m || !stats) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(m->device));
    memset(stats, 0, sizeof *stats);
    TRY(read_scal(m, stats));
    stats->nnz = m->last_nnz;
    stats->steps = 1;
    return FMHIP_OK;
}

// ---- measurement

static int profile_begin(fmhip_model_t m, bool rotate, int period) {
    if (!m) return fail(FMHIP_ERR_INVALID, "model is NULL");
    if (period < 1) return fail(FMHIP_ERR_INVALID, "period must be >= 1");
    WriteLock lock(m);
    for (auto &r : m->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->prof.clear();
    m->profiling = true;
    m->prof_rotate = rotate;
    m->prof_period = period;
    m->prof_step = 0;
    return FMHIP_OK;
}

int fmhip_profile_begin(fmhip_model_t m) { return profile_begin(m, false, 1); }
int fmhip_profile_begin_rotating(fmhip_model_t m) { return profile_begin(m, true, 1); }
int fmhip_profile_begin_sampled(fmhip_model_t m, int period) { return profile_begin(m, true, period); }

int fmhip_profile_end(fmhip_model_t m, fmhip_profile *p) {
    WriteLock lock(m);
    if (!m || !p) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(m->device));
    m->profiling = false;
    memset(p, 0, sizeof *p);
    HIP_TRY(hipStreamSynchronize(m->stream));
    int64_t last_step[FMHIP_K_COUNT];
    for (int64_t &x : last_step) x = -1;
    for (auto &r : m->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            p->ms[r.kind] += ms;
            p->launches[r.kind] += 1;
            p->nnz[r.kind] += r.nnz;
            p->rows[r.kind] += r.rows;
            if (r.step != last_step[r.kind]) {       // records are in launch order: a new step index = one more timed step
                p->steps[r.kind] += 1;
                last_step[r.kind] = r.step;
            }
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->prof.clear();
    return FMHIP_OK;
}

}  // extern "C"
