/*
 * fm_oracle.h — CPU (fp64) restatement of SparkFM's FM arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library.  The product path (sparkfm_amd/,
 * libfmhip.so) must never call into it.
 *
 * PARITY STATUS: "parity unpinned by the reference" — SparkFM has no tests, no
 * golden vectors and cannot be compiled or run here (no JVM).
 * This restatement is pinned instead by (1) the exact-rational known-answer
 * vectors of tests/golden/ (derived from the naive pairwise FM definition),
 * (2) an independent numpy twin (oracle/fm_oracle_np.py), (3) finite-difference
 * checks of the gradient.  The mini-batch SGD update has NO reference
 * counterpart (SparkFM only ships ALS); it is defined here, once, in fp64.
 *
 * Citations: S/ = /root/reference/src/main/scala/io/edstud/spark/
 *
 * Parameter layout (identical to breeze's column-major DenseMatrix(k, n+1),
 * S/fm/FMModel.scala:19): v[f + i*k] is factor f of feature i; w has n1 = n+1
 * slots where n = num_attribute = max feature index (S/fm/FMModel.scala:18).
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* S/fm/FMModel.scala:34-63 — FMModel.predict for one sparse row (stored order). */
double fmo_predict_row(int k, double w0, const double *w, const double *v,
                       int64_t nnz, const int32_t *idx, const double *val);

/* predict over CSR rows; out[r] = yhat_r.  Multi-threaded over rows (order
 * inside a row is the reference's; rows are independent). */
void fmo_predict(int k, double w0, const double *w, const double *v,
                 int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                 const double *val, double *out, int threads);

/* S/Model.scala:13-19 — sqrt(sum_r (y_r - yhat_r)^2 / N). */
double fmo_rmse(int k, double w0, const double *w, const double *v,
                int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                const double *val, const double *y, int threads);

/* S/fm/lib/ALS.scala:142-144 — e_r = yhat_r - y_r (prediction minus target). */
void fmo_residual(int k, double w0, const double *w, const double *v,
                  int64_t n_rows, const int64_t *row_ptr, const int32_t *col,
                  const double *val, const double *y, double *e, int threads);

/* S/DataSet.scala:31-38 — row->column transpose.  Outputs CSC: col_ptr[n1+1],
 * rows[nnz] (ascending inside a column), cval[nnz]. */
void fmo_transpose(int64_t n_rows, int64_t n1, const int64_t *row_ptr,
                   const int32_t *col, const double *val,
                   int64_t *col_ptr, int32_t *rows, double *cval);

/* S/DataSet.scala:27-29 — dimension = max over rows of max(index); 0 if empty. */
int32_t fmo_dimension(int64_t n_rows, const int64_t *row_ptr, const int32_t *col);

/* S/fm/lib/ALS.scala:146-150 — q_r = sum_i v[f,i] x_ri for one factor f,
 * computed from the TRANSPOSED data (feature-ascending accumulation). */
void fmo_term_q(int k, int f, const double *v, int64_t n_rows, int64_t n1,
                const int64_t *col_ptr, const int32_t *rows, const double *cval,
                double *q);

/* Gradient of 0.5*e^2 summed over rows [r0,r1): g = sum_r e_r * h_r(theta) with
 * h from S/fm/lib/ALS.scala:56-58 (V), :40 (w), :21 (w0).
 * gv[k*n1] (same layout as v), gw[n1], *gw0, *sse (sum e^2).  Outputs are
 * OVERWRITTEN.  Also returns e for the rows if e_out != NULL. */
void fmo_batch_grad(int k, int64_t n1, double w0, const double *w, const double *v,
                    int64_t r0, int64_t r1, const int64_t *row_ptr,
                    const int32_t *col, const double *val, const double *y,
                    double *gv, double *gw, double *gw0, double *sse,
                    double *e_out, int threads);

/* Build-defined mini-batch SGD step (no reference counterpart):
 *   theta <- theta - eta * ( g_theta / |B| + lambda_theta * theta )
 * over rows [r0,r1).  Returns sum e^2 of the batch (before the update). */
double fmo_sgd_step(int k, int64_t n1, double *w0, double *w, double *v,
                    int64_t r0, int64_t r1, const int64_t *row_ptr,
                    const int32_t *col, const double *val, const double *y,
                    double eta, double reg0, double regw, double regv,
                    double *scratch /* k*n1 + n1 doubles, or NULL */, int threads);

/* One epoch: batches are the contiguous row blocks [b*B, (b+1)*B) visited in the
 * order given by `order` (n_batches entries; NULL = ascending).  Returns the
 * sum over batches of sum e^2 (each measured before its own update). */
double fmo_sgd_epoch(int k, int64_t n1, double *w0, double *w, double *v,
                     int64_t n_rows, int64_t batch_rows, const int64_t *order,
                     const int64_t *row_ptr, const int32_t *col, const double *val,
                     const double *y, double eta, double reg0, double regw,
                     double regv, int threads);

/* S/fm/lib/ALS.scala:15-75,152-198 — one ALS.learn epoch, in place.
 * num_attribute = n (arrays have n+1 slots; the loop is `0 until num_attribute`
 * so the last slot is never trained — quirk Q1, reproduced).  CSC input is the
 * transpose of the data.  e_work: n_rows doubles of scratch. */
void fmo_als_epoch(int k, int64_t num_attribute, double *w0, double *w, double *v,
                   double reg0, double regw, double regv, int64_t n_rows,
                   const int64_t *row_ptr, const int32_t *col, const double *val,
                   const double *y, const int64_t *col_ptr, const int32_t *rows,
                   const double *cval, double *e_work);

int fmo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
