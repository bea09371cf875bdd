/*
 * fm_oracle.c — CPU (fp64) restatement of SparkFM's FM arithmetic.
 * TEST INFRASTRUCTURE ONLY — see fm_oracle.h for the rules and the parity status
 * ("parity unpinned by the reference": SparkFM ships no tests / golden vectors).
 *
 * Every function cites the reference lines it restates.
 * S/ = /root/reference/src/main/scala/io/edstud/spark/
 */
#include "fm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int fmo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int clamp_threads(int threads) {
    int m = fmo_max_threads();
    if (threads <= 0 || threads > m) threads = m;
    return threads;
}

/* S/fm/FMModel.scala:34-55 (predict) + :57-63 (computeFactorComponents).
 * Summation order is the reference's: w0, then the linear terms reduced left to
 * right in stored order (:45), then for each factor i = 0..k-1 (:48) the in-order
 * sums sum_f and sum_sqr_f (:58-60), adding 0.5*(sum_f^2 - sum_sqr_f) (:50).
 * An empty row returns w0 only (:42, quirk Q6). */
double fmo_predict_row(int k, double w0, const double *w, const double *v,
                       int64_t nnz, const int32_t *idx, const double *val) {
    double result = 0.0;
    result += w0;                                   /* :38-40, k0 = true */
    if (nnz > 0) {                                  /* :42 */
        double lin = w[idx[0]] * val[0];            /* :45 reduce(_+_) */
        for (int64_t a = 1; a < nnz; ++a) lin += w[idx[a]] * val[a];
        result += lin;
        for (int f = 0; f < k; ++f) {               /* :48 */
            double t = v[f + (int64_t)idx[0] * k] * val[0];   /* :58 */
            double sum_f = t;                       /* :59 */
            double sum_sqr_f = t * t;               /* :60 */
            for (int64_t a = 1; a < nnz; ++a) {
                t = v[f + (int64_t)idx[a] * k] * val[a];
                sum_f += t;
                sum_sqr_f += t * t;
            }
            result += 0.5 * (sum_f * sum_f - sum_sqr_f);      /* :50 */
        }
    }
    return result;
}

void fmo_predict(int k, double w0, const double *w, const double *v,
                 int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                 const double *val, double *out, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t p = row_ptr[r];
        out[r] = fmo_predict_row(k, w0, w, v, row_ptr[r + 1] - p, col + p, val + p);
    }
}

/* S/Model.scala:13-19: rmse_sqr = sum over rows of (y - predict)^2 (note the
 * sign: target minus prediction, quirk Q3), rmse = sqrt(rmse_sqr / size).
 * RDD.sum() has no fixed order; here: in-order per thread chunk, chunks added in
 * thread order (deterministic for a given thread count). */
double fmo_rmse(int k, double w0, const double *w, const double *v,
                int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                const double *val, const double *y, int threads) {
    threads = clamp_threads(threads);
    double total = 0.0;
    double *part = (double *)calloc((size_t)threads, sizeof(double));
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        int64_t lo = n_rows * t / nt, hi = n_rows * (t + 1) / nt;
        double s = 0.0;
        for (int64_t r = lo; r < hi; ++r) {
            int64_t p = row_ptr[r];
            double d = y[r] - fmo_predict_row(k, w0, w, v, row_ptr[r + 1] - p, col + p, val + p);
            s += d * d;
        }
        part[t] = s;
    }
    for (int t = 0; t < threads; ++t) total += part[t];
    free(part);
    return n_rows > 0 ? sqrt(total / (double)n_rows) : 0.0;
}

/* S/fm/lib/ALS.scala:142-144: e_r = predict(x_r) - y_r. */
void fmo_residual(int k, double w0, const double *w, const double *v,
                  int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                  const double *val, const double *y, double *e, int threads) {
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t p = row_ptr[r];
        e[r] = fmo_predict_row(k, w0, w, v, row_ptr[r + 1] - p, col + p, val + p) - y[r];
    }
}

/* S/DataSet.scala:31-38: zipWithIndex + flatMap (featureId -> (rowIdx, value)) +
 * groupByKey.  groupByKey's value order is unspecified in Spark; this
 * restatement fixes it to ascending row index (a stable counting sort). */
void fmo_transpose(int64_t n_rows, int64_t n1, const int64_t *row_ptr,
                   const int32_t *col, const double *val,
                   int64_t *col_ptr, int32_t *rows, double *cval) {
    int64_t nnz = row_ptr[n_rows];
    memset(col_ptr, 0, (size_t)(n1 + 1) * sizeof(int64_t));
    for (int64_t p = 0; p < nnz; ++p) col_ptr[col[p] + 1]++;
    for (int64_t i = 0; i < n1; ++i) col_ptr[i + 1] += col_ptr[i];
    int64_t *cur = (int64_t *)malloc((size_t)n1 * sizeof(int64_t));
    memcpy(cur, col_ptr, (size_t)n1 * sizeof(int64_t));
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
            int64_t d = cur[col[p]]++;
            rows[d] = (int32_t)r;
            cval[d] = val[p];
        }
    free(cur);
}

/* S/DataSet.scala:27-29: rdd.map(_.index.max).reduce(math.max); 0 when empty
 * (:28 `else 0`).  breeze's `index` is the full backing array; rows here carry
 * exactly their stored entries.  An all-empty dataset yields 0. */
int32_t fmo_dimension(int64_t n_rows, const int64_t *row_ptr, const int32_t *col) {
    int32_t m = 0;
    int64_t nnz = n_rows > 0 ? row_ptr[n_rows] : 0;
    for (int64_t p = 0; p < nnz; ++p)
        if (col[p] > m) m = col[p];
    return m;
}

/* S/fm/lib/ALS.scala:146-150: flatMap over the transposed columns emitting
 * (rowIdx, v(f, fid) * x) and groupByKey(rowIdx).sum.  Accumulation order fixed
 * here to ascending feature id. */
void fmo_term_q(int k, int f, const double *v, int64_t n_rows, int64_t n1,
                const int64_t *col_ptr, const int32_t *rows, const double *cval,
                double *q) {
    for (int64_t r = 0; r < n_rows; ++r) q[r] = 0.0;
    for (int64_t i = 0; i < n1; ++i) {
        double vfi = v[f + i * k];
        for (int64_t p = col_ptr[i]; p < col_ptr[i + 1]; ++p) q[rows[p]] += vfi * cval[p];
    }
}

/* ---- mini-batch gradient ------------------------------------------------ */
#define FMO_MAX_K_STACK 256

/* per-row forward that also returns q_f = sum_i v_fi x_i (the quantity
 * S/fm/lib/ALS.scala:146-150 calls q) so the gradient h of :56-58 can be formed. */
static double row_forward_q(int k, double w0, const double *w, const double *v,
                            int64_t nnz, const int32_t *idx, const double *val, double *q) {
    /* Same arithmetic as fmo_predict_row with the two loops interchanged (nonzeros outer,
     * factors inner, so the inner loop is contiguous and vectorises).  Every factor's sums
     * still accumulate over the nonzeros in stored order and the factors are still added to
     * the result in the order 0..k-1, so the value is bit-identical to fmo_predict_row
     * (tests/test_oracle_kat.py checks that). */
    double sq[FMO_MAX_K_STACK];
    double *s = k <= FMO_MAX_K_STACK ? sq : (double *)malloc((size_t)k * sizeof(double));
    double result = w0;
    for (int f = 0; f < k; ++f) { q[f] = 0.0; s[f] = 0.0; }
    if (nnz > 0) {
        double lin = w[idx[0]] * val[0];
        for (int64_t a = 1; a < nnz; ++a) lin += w[idx[a]] * val[a];
        result += lin;
        {
            const double *vi = v + (int64_t)idx[0] * k;
            const double x = val[0];
            for (int f = 0; f < k; ++f) { double t = vi[f] * x; q[f] = t; s[f] = t * t; }
        }
        for (int64_t a = 1; a < nnz; ++a) {
            const double *vi = v + (int64_t)idx[a] * k;
            const double x = val[a];
            for (int f = 0; f < k; ++f) { double t = vi[f] * x; q[f] += t; s[f] += t * t; }
        }
        for (int f = 0; f < k; ++f) result += 0.5 * (q[f] * q[f] - s[f]);
    }
    if (s != sq) free(s);
    return result;
}

/* Per-thread accumulation workspace, reused across the steps of an epoch: thread
 * t > 0 owns a dense buffer plus the list of features it touched, so that the
 * cross-thread reduction and the re-zeroing cost O(touched), not O(n). */
typedef struct {
    int threads;
    size_t gsz;
    int64_t n1;
    double **tgv, **tgw;
    unsigned char **seen;
    int32_t **tl;
    int64_t *nt;
    double *tg0, *tsse;
} grad_ws;

static grad_ws *ws_new(int threads, int k, int64_t n1) {
    grad_ws *ws = (grad_ws *)calloc(1, sizeof(grad_ws));
    ws->threads = threads;
    ws->gsz = (size_t)k * (size_t)n1;
    ws->n1 = n1;
    ws->tgv = (double **)calloc((size_t)threads, sizeof(double *));
    ws->tgw = (double **)calloc((size_t)threads, sizeof(double *));
    ws->seen = (unsigned char **)calloc((size_t)threads, sizeof(unsigned char *));
    ws->tl = (int32_t **)calloc((size_t)threads, sizeof(int32_t *));
    ws->nt = (int64_t *)calloc((size_t)threads, sizeof(int64_t));
    ws->tg0 = (double *)calloc((size_t)threads, sizeof(double));
    ws->tsse = (double *)calloc((size_t)threads, sizeof(double));
    for (int t = 1; t < threads; ++t) {
        ws->tgv[t] = (double *)calloc(ws->gsz, sizeof(double));
        ws->tgw[t] = (double *)calloc((size_t)n1, sizeof(double));
        ws->seen[t] = (unsigned char *)calloc((size_t)n1, 1);
        ws->tl[t] = (int32_t *)malloc((size_t)n1 * sizeof(int32_t));
    }
    return ws;
}

static void ws_free(grad_ws *ws) {
    for (int t = 1; t < ws->threads; ++t) {
        free(ws->tgv[t]); free(ws->tgw[t]); free(ws->seen[t]); free(ws->tl[t]);
    }
    free(ws->tgv); free(ws->tgw); free(ws->seen); free(ws->tl);
    free(ws->nt); free(ws->tg0); free(ws->tsse);
    free(ws);
}

/* g_theta = sum_r e_r * h_r(theta):
 *   h_r(v_fi) = x_ri*q_rf - x_ri^2*v_fi   S/fm/lib/ALS.scala:56-58
 *   h_r(w_i)  = x_ri                      S/fm/lib/ALS.scala:40 (h = features(id))
 *   h_r(w0)   = 1                         S/fm/lib/ALS.scala:21,152-154
 *   e_r       = yhat_r - y_r              S/fm/lib/ALS.scala:142-144
 * Rows are split into contiguous per-thread chunks; each thread accumulates in
 * row order into its own buffer (thread 0 straight into the output); buffers are
 * added in thread order, so the result is deterministic for a given thread
 * count and equals plain row-order accumulation at threads = 1. */
static void batch_grad_ws(grad_ws *ws, int k, int64_t n1, double w0, const double *w,
                          const double *v, int64_t r0, int64_t r1, const int64_t *row_ptr,
                          const int32_t *col, const double *val, const double *y,
                          double *gv, double *gw, double *gw0, double *sse, double *e_out) {
    int threads = ws->threads;
    int64_t nb = r1 - r0;
    memset(gv, 0, ws->gsz * sizeof(double));
    memset(gw, 0, (size_t)n1 * sizeof(double));
    ws->tgv[0] = gv;
    ws->tgw[0] = gw;
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        int64_t lo = r0 + nb * t / threads, hi = r0 + nb * (t + 1) / threads;
        double *q = (double *)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
        double *mgv = ws->tgv[t], *mgw = ws->tgw[t];
        unsigned char *seen = ws->seen[t];
        int32_t *tl = ws->tl[t];
        int64_t nt = 0;
        double g0 = 0.0, s2 = 0.0;
        for (int64_t r = lo; r < hi; ++r) {
            int64_t p = row_ptr[r], nnz = row_ptr[r + 1] - p;
            double yhat = row_forward_q(k, w0, w, v, nnz, col + p, val + p, q);
            double e = yhat - y[r];
            if (e_out) e_out[r - r0] = e;
            g0 += e;
            s2 += e * e;
            for (int64_t a = 0; a < nnz; ++a) {
                int32_t i = col[p + a];
                double x = val[p + a];
                if (seen && !seen[i]) { seen[i] = 1; tl[nt++] = i; }
                mgw[i] += e * x;
                const double *vi = v + (int64_t)i * k;
                double *gi = mgv + (int64_t)i * k;
                for (int f = 0; f < k; ++f) gi[f] += e * (x * q[f] - x * x * vi[f]);
            }
        }
        ws->tg0[t] = g0;
        ws->tsse[t] = s2;
        ws->nt[t] = nt;
        free(q);
    }
    double g0 = 0.0, s2 = 0.0;
    for (int t = 0; t < threads; ++t) { g0 += ws->tg0[t]; s2 += ws->tsse[t]; }
    /* cross-thread reduction, parallel over features; for every feature the thread buffers
     * are added in thread order (deterministic), and touched slots are re-zeroed */
    if (threads > 1) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t i = 0; i < n1; ++i) {
            for (int t = 1; t < threads; ++t) {
                if (!ws->seen[t][i]) continue;
                double *tv = ws->tgv[t], *tw = ws->tgw[t];
                gw[i] += tw[i];
                tw[i] = 0.0;
                double *gi = gv + i * k;
                double *ti = tv + i * k;
                for (int f = 0; f < k; ++f) { gi[f] += ti[f]; ti[f] = 0.0; }
                ws->seen[t][i] = 0;
            }
        }
    }
    *gw0 = g0;
    *sse = s2;
}

static int grad_threads(int threads, int64_t nb) {
    threads = clamp_threads(threads);
    if (threads > nb) threads = nb > 0 ? (int)nb : 1;
    return threads;
}

void fmo_batch_grad(int k, int64_t n1, double w0, const double *w, const double *v,
                    int64_t r0, int64_t r1, const int64_t *row_ptr,
                    const int32_t *col, const double *val, const double *y,
                    double *gv, double *gw, double *gw0, double *sse,
                    double *e_out, int threads) {
    grad_ws *ws = ws_new(grad_threads(threads, r1 - r0), k, n1);
    batch_grad_ws(ws, k, n1, w0, w, v, r0, r1, row_ptr, col, val, y, gv, gw, gw0, sse, e_out);
    ws_free(ws);
}

/* Build-defined SGD update (no SparkFM counterpart — SURVEY.md §0.1):
 *   theta <- theta - eta * (g_theta/|B| + lambda_theta*theta), all n+1 slots
 * (the ALS-only quirk Q1, "last slot never trained", does not apply here). */
static double sgd_step_ws(grad_ws *ws, int k, int64_t n1, double *w0, double *w, double *v,
                          int64_t r0, int64_t r1, const int64_t *row_ptr,
                          const int32_t *col, const double *val, const double *y,
                          double eta, double reg0, double regw, double regv,
                          double *buf, int threads) {
    size_t gsz = (size_t)k * (size_t)n1;
    double *gv = buf, *gw = buf + gsz, gw0 = 0.0, sse = 0.0;
    batch_grad_ws(ws, k, n1, *w0, w, v, r0, r1, row_ptr, col, val, y, gv, gw, &gw0, &sse, NULL);
    double invb = r1 > r0 ? 1.0 / (double)(r1 - r0) : 0.0;
    threads = clamp_threads(threads);
    *w0 -= eta * (gw0 * invb + reg0 * (*w0));
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n1; ++i) {
        w[i] -= eta * (gw[i] * invb + regw * w[i]);
        double *vi = v + i * k;
        const double *gi = gv + i * k;
        for (int f = 0; f < k; ++f) vi[f] -= eta * (gi[f] * invb + regv * vi[f]);
    }
    return sse;
}

double fmo_sgd_step(int k, int64_t n1, double *w0, double *w, double *v,
                    int64_t r0, int64_t r1, const int64_t *row_ptr,
                    const int32_t *col, const double *val, const double *y,
                    double eta, double reg0, double regw, double regv,
                    double *scratch, int threads) {
    size_t gsz = (size_t)k * (size_t)n1;
    double *buf = scratch ? scratch : (double *)malloc((gsz + (size_t)n1) * sizeof(double));
    grad_ws *ws = ws_new(grad_threads(threads, r1 - r0), k, n1);
    double sse = sgd_step_ws(ws, k, n1, w0, w, v, r0, r1, row_ptr, col, val, y,
                             eta, reg0, regw, regv, buf, threads);
    ws_free(ws);
    if (!scratch) free(buf);
    return sse;
}

double fmo_sgd_epoch(int k, int64_t n1, double *w0, double *w, double *v,
                     int64_t n_rows, int64_t batch_rows, const int64_t *order,
                     const int64_t *row_ptr, const int32_t *col, const double *val,
                     const double *y, double eta, double reg0, double regw,
                     double regv, int threads) {
    if (batch_rows <= 0) batch_rows = n_rows;
    if (n_rows <= 0) return 0.0;
    int64_t nb = (n_rows + batch_rows - 1) / batch_rows;
    size_t gsz = (size_t)k * (size_t)n1;
    double *scratch = (double *)malloc((gsz + (size_t)n1) * sizeof(double));
    int64_t min_rows = n_rows % batch_rows ? n_rows % batch_rows : batch_rows;
    grad_ws *ws = ws_new(grad_threads(threads, min_rows), k, n1);
    double total = 0.0;
    for (int64_t j = 0; j < nb; ++j) {
        int64_t b = order ? order[j] : j;
        int64_t r0 = b * batch_rows, r1 = r0 + batch_rows;
        if (r1 > n_rows) r1 = n_rows;
        total += sgd_step_ws(ws, k, n1, w0, w, v, r0, r1, row_ptr, col, val, y,
                             eta, reg0, regw, regv, scratch, threads);
    }
    ws_free(ws);
    free(scratch);
    return total;
}

/* ---- ALS (the reference's only learner) ---------------------------------- */

/* S/fm/lib/ALS.scala:190-192 */
static int is_updatable(double nv, double ov) { return !isnan(nv) && !isinf(nv) && nv != ov; }

/* S/fm/lib/ALS.scala:167-176 */
static double compute_theta(double theta, double reg, double sum_e_h, double sum_h_sqr) {
    double theta_new = -(sum_e_h - theta * sum_h_sqr) / (reg + sum_h_sqr);
    return is_updatable(theta_new, theta) ? theta_new : theta;
}

/* S/fm/lib/ALS.scala:15-75.  `e` is the driver-side map of :31 as an array. */
void fmo_als_epoch(int k, int64_t num_attribute, double *w0, double *w, double *v,
                   double reg0, double regw, double regv, int64_t n_rows,
                   const int64_t *row_ptr, const int32_t *col, const double *val,
                   const double *y, const int64_t *col_ptr, const int32_t *rows,
                   const double *cval, double *e) {
    int64_t n1 = num_attribute + 1;
    /* :17 precomputeTermE */
    fmo_residual(k, *w0, w, v, n_rows, row_ptr, col, val, y, e, 1);
    /* :19-28 global bias; drawGlobalBias :152-154 = computeTheta(theta, reg, sum e, size) */
    {
        double se = 0.0;
        if (n_rows > 0) { se = e[0]; for (int64_t r = 1; r < n_rows; ++r) se += e[r]; }
        double w0n = compute_theta(*w0, reg0, se, (double)n_rows);
        /* :23-27 as Spark EVALUATES them.  `error` is a lazy RDD whose closures hold `fm` by reference: `error.map(e => e +
         * (w0 - fm.w0))` (:24) is only built here, `fm.w0 = w0` (:27) runs next, and the job that materialises it (:31,
         * transformAsMap) serialises the closures AFTER that assignment.  So precomputeTermE's `fm.predict` already adds
         * the NEW bias, and the mapped term is (w0 - fm.w0) = w0n - w0n = 0: the residuals the sweeps start from are
         * predict_{new w0}(x) - y + 0, not e_old + (w0n - w0_old) — equal in exact arithmetic, a last-bit difference in
         * fp64. */
        int upd = is_updatable(w0n, *w0);
        *w0 = w0n;                                        /* :27 */
        if (upd) {
            fmo_residual(k, *w0, w, v, n_rows, row_ptr, col, val, y, e, 1);
            for (int64_t r = 0; r < n_rows; ++r) e[r] = e[r] + (w0n - *w0);   /* + 0.0: kept for the record */
        }
    }
    /* :36-43 linear weights; `0 until num_attribute` skips slot n (quirk Q1) */
    for (int64_t id = 0; id < num_attribute; ++id) {
        int64_t a = col_ptr[id], b = col_ptr[id + 1];
        if (b == a) continue;                             /* :39 features.contains(id) */
        /* drawTheta :156-165 with h = features(id) (:40); components :178-188 */
        double sum_h_sqr = cval[a] * cval[a], sum_e_h = e[rows[a]] * cval[a];
        for (int64_t p = a + 1; p < b; ++p) {
            sum_h_sqr += cval[p] * cval[p];
            sum_e_h += e[rows[p]] * cval[p];
        }
        double th = w[id], thn = compute_theta(th, regw, sum_e_h, sum_h_sqr);
        if (is_updatable(thn, th)) {                      /* :160-162, updateError :194-198 */
            double d = thn - th;
            for (int64_t p = a; p < b; ++p) e[rows[p]] += cval[p] * d;
        }
        w[id] = thn;
    }
    /* :45-70 factors */
    double *q = (double *)malloc((size_t)(n_rows > 0 ? n_rows : 1) * sizeof(double));
    double *h = NULL;
    int64_t hcap = 0;
    for (int f = 0; f < k; ++f) {
        fmo_term_q(k, f, v, n_rows, n1, col_ptr, rows, cval, q);   /* :50 */
        for (int64_t id = 0; id < num_attribute; ++id) {           /* :52 */
            int64_t a = col_ptr[id], b = col_ptr[id + 1];
            if (b == a) continue;                                  /* :54 */
            if (b - a > hcap) { hcap = b - a; h = (double *)realloc(h, (size_t)hcap * sizeof(double)); }
            double vfi = v[f + id * k];
            /* :56-58 h = x*q(row) - x*x*v(f,id) */
            for (int64_t p = a; p < b; ++p) h[p - a] = cval[p] * q[rows[p]] - cval[p] * cval[p] * vfi;
            double sum_h_sqr = h[0] * h[0], sum_e_h = e[rows[a]] * h[0];
            for (int64_t p = a + 1; p < b; ++p) {
                sum_h_sqr += h[p - a] * h[p - a];
                sum_e_h += e[rows[p]] * h[p - a];
            }
            double vn = compute_theta(vfi, regv, sum_e_h, sum_h_sqr);
            if (is_updatable(vn, vfi)) {
                double d = vn - vfi;
                for (int64_t p = a; p < b; ++p) e[rows[p]] += h[p - a] * d;
            }
            /* :60-62 q(row) += x * (v_new - v_old) */
            for (int64_t p = a; p < b; ++p) q[rows[p]] += cval[p] * (vn - vfi);
            v[f + id * k] = vn;                                    /* :64 */
        }
    }
    free(h);
    free(q);
}
