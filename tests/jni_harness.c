/*
 * tests/jni_harness.c — drives the compiled JNI shim (jvm/fmhip_jni.c) through an IN-MEMORY JNIEnv, without a JVM.
 *
 * The image has no JDK, so the shim cannot be loaded into a JVM here; what CAN be checked is everything the shim itself
 * does: which entry point of include/fmhip.h each native calls, with which arguments, and how it takes and releases the
 * Java arrays.  The JNIEnv below (tests/jni_stub/jni.h: a stand-in table, test infrastructure) behaves like the least
 * convenient VM the specification allows: Get<Type>ArrayElements hands out a COPY, Release<Type>ArrayElements writes it
 * back unless the mode is JNI_ABORT and then poisons and frees it — an output released with JNI_ABORT, an input modified
 * in place, a buffer used after its release all show up as wrong results.  Every native's outcome is compared with the
 * same call made straight through the C ABI.
 *
 *   jni_harness host   natives that are host arithmetic (no GPU): shardRows, featureCounts, rankFromCounts, relabelColumns
 *   jni_harness gpu    the model / dataset / training / scoring natives and a one-rank communicator (needs an MI355X)
 *
 * Built and run by tests/test_host_cpu.py (host) and tests/test_gpu_configs.py (gpu).
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmhip.h"

/* ---- the in-memory VM ------------------------------------------------------------------------------------------ */
typedef struct { int elem; jsize len; void *data; } harness_array;     /* elem: bytes per element */
typedef struct { char thrown[512]; int n_thrown; int live_copies; } harness_state;

static harness_state g_state;

static jarray new_array(int elem, jsize len, const void *init) {
    harness_array *a = (harness_array *)malloc(sizeof *a);
    a->elem = elem;
    a->len = len;
    a->data = calloc((size_t)(len > 0 ? len : 1), (size_t)elem);
    if (init && len > 0) memcpy(a->data, init, (size_t)len * (size_t)elem);
    return (jarray)a;
}
static void free_array(jarray j) {
    harness_array *a = (harness_array *)j;
    if (!a) return;
    free(a->data);
    free(a);
}
#define ARR(j) ((harness_array *)(j))

static jclass h_FindClass(JNIEnv *env, const char *name) { (void)env; return (jclass)name; }
static jint h_ThrowNew(JNIEnv *env, jclass c, const char *msg) {
    (void)env;
    snprintf(g_state.thrown, sizeof g_state.thrown, "%s: %s", (const char *)c, msg ? msg : "");
    ++g_state.n_thrown;
    return 0;
}
static jsize h_GetArrayLength(JNIEnv *env, jarray a) { (void)env; return ARR(a)->len; }
static jbyteArray h_NewByteArray(JNIEnv *env, jsize n) { (void)env; return new_array(1, n, NULL); }
static jlongArray h_NewLongArray(JNIEnv *env, jsize n) { (void)env; return new_array(8, n, NULL); }

static void *get_copy(jarray a, jboolean *is_copy) {
    const size_t bytes = (size_t)(ARR(a)->len > 0 ? ARR(a)->len : 1) * (size_t)ARR(a)->elem;
    void *p = malloc(bytes);
    memcpy(p, ARR(a)->data, bytes);
    if (is_copy) *is_copy = JNI_TRUE;
    ++g_state.live_copies;
    return p;
}
static void release_copy(jarray a, void *p, jint mode) {
    const size_t bytes = (size_t)(ARR(a)->len > 0 ? ARR(a)->len : 1) * (size_t)ARR(a)->elem;
    if (mode != JNI_ABORT) memcpy(ARR(a)->data, p, bytes);
    if (mode == JNI_COMMIT) return;
    memset(p, 0xA5, bytes);        /* a use after release reads garbage */
    free(p);
    --g_state.live_copies;
}
static jint *h_GetInt(JNIEnv *env, jintArray a, jboolean *c) { (void)env; return (jint *)get_copy(a, c); }
static jlong *h_GetLong(JNIEnv *env, jlongArray a, jboolean *c) { (void)env; return (jlong *)get_copy(a, c); }
static jdouble *h_GetDouble(JNIEnv *env, jdoubleArray a, jboolean *c) { (void)env; return (jdouble *)get_copy(a, c); }
static void h_RelInt(JNIEnv *env, jintArray a, jint *p, jint m) { (void)env; release_copy(a, p, m); }
static void h_RelLong(JNIEnv *env, jlongArray a, jlong *p, jint m) { (void)env; release_copy(a, p, m); }
static void h_RelDouble(JNIEnv *env, jdoubleArray a, jdouble *p, jint m) { (void)env; release_copy(a, p, m); }
static void region(jarray a, jsize start, jsize len, void *buf, int to_array) {
    if (start < 0 || len < 0 || start + len > ARR(a)->len) { fprintf(stderr, "harness: array region out of bounds\n"); exit(3); }
    char *at = (char *)ARR(a)->data + (size_t)start * (size_t)ARR(a)->elem;
    if (to_array) memcpy(at, buf, (size_t)len * (size_t)ARR(a)->elem);
    else memcpy(buf, at, (size_t)len * (size_t)ARR(a)->elem);
}
static void h_GetByteRegion(JNIEnv *env, jbyteArray a, jsize s, jsize n, jbyte *b) { (void)env; region(a, s, n, b, 0); }
static void h_SetByteRegion(JNIEnv *env, jbyteArray a, jsize s, jsize n, const jbyte *b) { (void)env; region(a, s, n, (void *)b, 1); }
static void h_GetDoubleRegion(JNIEnv *env, jdoubleArray a, jsize s, jsize n, jdouble *b) { (void)env; region(a, s, n, b, 0); }
static void h_SetDoubleRegion(JNIEnv *env, jdoubleArray a, jsize s, jsize n, const jdouble *b) { (void)env; region(a, s, n, (void *)b, 1); }
static void h_SetLongRegion(JNIEnv *env, jlongArray a, jsize s, jsize n, const jlong *b) { (void)env; region(a, s, n, (void *)b, 1); }
static void *h_GetCritical(JNIEnv *env, jarray a, jboolean *c) { (void)env; if (c) *c = JNI_FALSE; return ARR(a)->data; }
static void h_RelCritical(JNIEnv *env, jarray a, void *p, jint m) { (void)env; (void)a; (void)p; (void)m; }

static const struct JNINativeInterface_ g_table = {
    &g_state, h_FindClass, h_ThrowNew, h_GetArrayLength, h_NewByteArray, h_NewLongArray, h_GetInt, h_GetLong, h_GetDouble,
    h_RelInt, h_RelLong, h_RelDouble, h_GetByteRegion, h_SetByteRegion, h_GetDoubleRegion, h_SetDoubleRegion, h_SetLongRegion,
    h_GetCritical, h_RelCritical,
};
static JNIEnv g_env = &g_table;

/* ---- the natives under test (jvm/fmhip_jni.c) ------------------------------------------------------------------- */
#define N(name) Java_io_edstud_spark_fm_lib_HipSGD_00024_##name
jlong N(modelCreate)(JNIEnv *, jobject, jint, jlong, jint);
void N(modelDestroy)(JNIEnv *, jobject, jlong);
void N(setParams)(JNIEnv *, jobject, jlong, jdouble, jdoubleArray, jdoubleArray);
void N(getParams)(JNIEnv *, jobject, jlong, jdoubleArray, jdoubleArray, jdoubleArray);
jlong N(datasetCreate)(JNIEnv *, jobject, jint, jlong, jlongArray, jintArray, jdoubleArray, jdoubleArray, jlong);
jlong N(rowsCreate)(JNIEnv *, jobject, jint, jlong, jlongArray, jintArray, jdoubleArray, jdoubleArray);
void N(datasetDestroy)(JNIEnv *, jobject, jlong);
void N(sgdEpoch)(JNIEnv *, jobject, jlong, jlong, jdouble, jdouble, jdouble, jdouble);
jdouble N(rmse)(JNIEnv *, jobject, jlong, jlong);
void N(predict)(JNIEnv *, jobject, jlong, jlong, jdoubleArray);
void N(predictRows)(JNIEnv *, jobject, jlong, jlong, jlongArray, jintArray, jdoubleArray, jdoubleArray);
jint N(deviceCount)(JNIEnv *, jobject);
jbyteArray N(commUniqueId)(JNIEnv *, jobject);
jlong N(commCreate)(JNIEnv *, jobject, jlong, jbyteArray, jint, jint);
void N(commDestroy)(JNIEnv *, jobject, jlong);
void N(dpPlan)(JNIEnv *, jobject, jlong, jlong, jlong, jdoubleArray);
void N(dpEpoch)(JNIEnv *, jobject, jlong, jlong, jlong, jdouble, jdouble, jdouble, jdouble);
jlong N(dpPlanSteps)(JNIEnv *, jobject, jlong);
void N(dpEpochOrder)(JNIEnv *, jobject, jlong, jlong, jlong, jdouble, jdouble, jdouble, jdouble, jlongArray);
void N(dpExchange)(JNIEnv *, jobject, jlong, jint);
jlongArray N(shardRows)(JNIEnv *, jobject, jlongArray, jint, jint);
void N(featureCounts)(JNIEnv *, jobject, jintArray, jlong, jlongArray);
void N(rankFromCounts)(JNIEnv *, jobject, jlongArray, jintArray, jintArray);
void N(relabelColumns)(JNIEnv *, jobject, jintArray, jintArray);

/* ---- checks ------------------------------------------------------------------------------------------------------ */
static int g_checks;
#define CHECK(cond, ...)                                                             \
    do {                                                                             \
        ++g_checks;                                                                  \
        if (!(cond)) {                                                               \
            fprintf(stderr, "jni_harness: FAILED %s:%d: ", __FILE__, __LINE__);      \
            fprintf(stderr, __VA_ARGS__);                                            \
            fprintf(stderr, " (pending exception: %s)\n", g_state.thrown);           \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)
#define NO_THROW() CHECK(g_state.n_thrown == 0, "a native raised")
#define OK(call) CHECK((call) == FMHIP_OK, "%s: %s", #call, fmhip_last_error())

/* a small CSR problem from a fixed LCG: rows of 3..12 distinct features out of n1, a few dominating ones */
typedef struct { int64_t n_rows, nnz; int32_t n1, k; int64_t *rp; int32_t *col; double *val, *y, *w, *v; double w0; } problem;
static uint64_t g_lcg = 88172645463325252ull;
static double unif(void) {
    g_lcg = g_lcg * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(g_lcg >> 11) * (1.0 / 9007199254740992.0);
}
static problem make_problem(int64_t n_rows, int32_t n1, int32_t k) {
    problem p;
    p.n_rows = n_rows; p.n1 = n1; p.k = k;
    p.rp = (int64_t *)calloc((size_t)n_rows + 1, sizeof(int64_t));
    p.col = (int32_t *)malloc((size_t)n_rows * 12 * sizeof(int32_t));
    p.val = (double *)malloc((size_t)n_rows * 12 * sizeof(double));
    p.y = (double *)malloc((size_t)n_rows * sizeof(double));
    char *seen = (char *)calloc((size_t)n1, 1);
    int64_t at = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int len = 3 + (int)(unif() * 10.0);
        const int64_t start = at;
        for (int j = 0; j < len; ++j) {
            double u = unif();
            int32_t f = (int32_t)(u * u * u * (double)n1);        /* skewed towards small ids */
            if (f >= n1) f = n1 - 1;
            if (seen[f]) continue;
            seen[f] = 1;
            p.col[at] = f;
            p.val[at] = 0.25 + unif();
            ++at;
        }
        for (int64_t q = start; q < at; ++q) seen[p.col[q]] = 0;
        p.rp[r + 1] = at;
        p.y[r] = unif() - 0.5;
    }
    free(seen);
    p.nnz = at;
    p.w = (double *)malloc((size_t)n1 * sizeof(double));
    p.v = (double *)malloc((size_t)n1 * (size_t)k * sizeof(double));
    p.w0 = 0.125;
    for (int32_t i = 0; i < n1; ++i) p.w[i] = 0.1 * (unif() - 0.5);
    for (int64_t i = 0; i < (int64_t)n1 * k; ++i) p.v[i] = 0.2 * (unif() - 0.5);
    return p;
}

static void run_host(void) {
    problem p = make_problem(2000, 300, 4);
    /* shardRows == fmhip_shard_rows, for every rank of a world of 3 */
    jlongArray jrp = new_array(8, (jsize)(p.n_rows + 1), p.rp);
    for (int rank = 0; rank < 3; ++rank) {
        jlongArray got = N(shardRows)(&g_env, NULL, jrp, 3, rank);
        NO_THROW();
        int64_t lo = -1, hi = -1;
        OK(fmhip_shard_rows(p.n_rows, p.rp, 3, rank, &lo, &hi));
        CHECK(got && ARR(got)->len == 2 && ((jlong *)ARR(got)->data)[0] == lo && ((jlong *)ARR(got)->data)[1] == hi, "shardRows rank %d", rank);
        free_array(got);
    }
    CHECK(memcmp(ARR(jrp)->data, p.rp, (size_t)(p.n_rows + 1) * 8) == 0, "shardRows modified its input");
    /* a bad request raises and returns null */
    {
        jlongArray got = N(shardRows)(&g_env, NULL, jrp, 3, 7);
        CHECK(got == NULL && g_state.n_thrown == 1 && strstr(g_state.thrown, "RuntimeException"), "shardRows(rank 7 of 3) must raise");
        g_state.n_thrown = 0;
        g_state.thrown[0] = 0;
    }
    /* featureCounts ACCUMULATES into counts (two halves of the columns == one pass) and leaves col alone */
    const jsize half = (jsize)(p.nnz / 2);
    jintArray c1 = new_array(4, half, p.col), c2 = new_array(4, (jsize)p.nnz - half, p.col + half);
    jlongArray jcounts = new_array(8, p.n1, NULL);
    N(featureCounts)(&g_env, NULL, c1, p.n1, jcounts);
    N(featureCounts)(&g_env, NULL, c2, p.n1, jcounts);
    NO_THROW();
    int64_t *want_counts = (int64_t *)calloc((size_t)p.n1, 8);
    OK(fmhip_feature_counts(p.nnz, p.col, p.n1, want_counts));
    CHECK(memcmp(ARR(jcounts)->data, want_counts, (size_t)p.n1 * 8) == 0, "featureCounts");
    CHECK(memcmp(ARR(c1)->data, p.col, (size_t)half * 4) == 0, "featureCounts modified col");
    /* rankFromCounts fills both outputs */
    jintArray jrank = new_array(4, p.n1, NULL), jby = new_array(4, p.n1, NULL);
    N(rankFromCounts)(&g_env, NULL, jcounts, jrank, jby);
    NO_THROW();
    int32_t *want_rank = (int32_t *)malloc((size_t)p.n1 * 4), *want_by = (int32_t *)malloc((size_t)p.n1 * 4);
    OK(fmhip_rank_from_counts(p.n1, want_counts, want_rank, want_by));
    CHECK(memcmp(ARR(jrank)->data, want_rank, (size_t)p.n1 * 4) == 0 && memcmp(ARR(jby)->data, want_by, (size_t)p.n1 * 4) == 0, "rankFromCounts");
    CHECK(memcmp(ARR(jcounts)->data, want_counts, (size_t)p.n1 * 8) == 0, "rankFromCounts modified counts");
    /* relabelColumns rewrites col in place and leaves rank alone */
    jintArray jcol = new_array(4, (jsize)p.nnz, p.col);
    N(relabelColumns)(&g_env, NULL, jcol, jrank);
    NO_THROW();
    int32_t *want_col = (int32_t *)malloc((size_t)p.nnz * 4);
    OK(fmhip_relabel_columns(p.nnz, p.col, p.n1, want_rank, want_col));
    CHECK(memcmp(ARR(jcol)->data, want_col, (size_t)p.nnz * 4) == 0, "relabelColumns");
    CHECK(memcmp(ARR(jrank)->data, want_rank, (size_t)p.n1 * 4) == 0, "relabelColumns modified rank");
    CHECK(g_state.live_copies == 0, "%d array copies were never released", g_state.live_copies);
}

static void run_gpu(void) {
    problem p = make_problem(3000, 400, 8);
    const double eta = 0.05, r0 = 0.0, rw = 1e-3, rv = 1e-3;
    CHECK(N(deviceCount)(&g_env, NULL) >= 1, "deviceCount");
    NO_THROW();
    jlongArray jrp = new_array(8, (jsize)(p.n_rows + 1), p.rp);
    jintArray jcol = new_array(4, (jsize)p.nnz, p.col);
    jdoubleArray jval = new_array(8, (jsize)p.nnz, p.val), jy = new_array(8, (jsize)p.n_rows, p.y);
    jdoubleArray jw = new_array(8, p.n1, p.w), jv = new_array(8, p.n1 * p.k, p.v);

    /* the natives' model / dataset and their twins made through the C ABI */
    const jlong hm = N(modelCreate)(&g_env, NULL, 0, p.n1 - 1, p.k);
    const jlong hd = N(datasetCreate)(&g_env, NULL, 0, p.n_rows, jrp, jcol, jval, jy, 1000);
    N(setParams)(&g_env, NULL, hm, p.w0, jw, jv);
    NO_THROW();
    CHECK(hm && hd, "handles");
    CHECK(memcmp(ARR(jcol)->data, p.col, (size_t)p.nnz * 4) == 0 && memcmp(ARR(jval)->data, p.val, (size_t)p.nnz * 8) == 0 &&
              memcmp(ARR(jw)->data, p.w, (size_t)p.n1 * 8) == 0, "inputs were modified");
    fmhip_model_t tm = NULL;
    fmhip_dataset_t td = NULL;
    OK(fmhip_model_create(0, p.n1 - 1, p.k, NULL, &tm));
    OK(fmhip_dataset_create(0, p.n_rows, p.rp, p.col, p.val, p.y, 1000, &td));
    OK(fmhip_model_set_params(tm, p.w0, p.w, p.v));

    /* two epochs each way, then parameters, RMSE and predictions must agree bit for bit */
    for (int e = 0; e < 2; ++e) {
        N(sgdEpoch)(&g_env, NULL, hm, hd, eta, r0, rw, rv);
        OK(fmhip_sgd_epoch(tm, td, eta, r0, rw, rv, NULL, NULL));
    }
    NO_THROW();
    jdoubleArray o0 = new_array(8, 1, NULL), ow = new_array(8, p.n1, NULL), ov = new_array(8, p.n1 * p.k, NULL);
    N(getParams)(&g_env, NULL, hm, o0, ow, ov);
    NO_THROW();
    double t0 = 0.0, *tw = (double *)malloc((size_t)p.n1 * 8), *tv = (double *)malloc((size_t)p.n1 * p.k * 8);
    OK(fmhip_model_get_params(tm, &t0, tw, tv));
    CHECK(((double *)ARR(o0)->data)[0] == t0 && t0 != p.w0, "getParams: w0 %.17g vs %.17g", ((double *)ARR(o0)->data)[0], t0);
    CHECK(memcmp(ARR(ow)->data, tw, (size_t)p.n1 * 8) == 0 && memcmp(ARR(ov)->data, tv, (size_t)p.n1 * p.k * 8) == 0, "getParams: w / v differ");
    CHECK(memcmp(tv, p.v, (size_t)p.n1 * p.k * 8) != 0, "training left v unchanged");
    double trmse = 0.0;
    OK(fmhip_rmse(tm, td, &trmse, NULL));
    CHECK(N(rmse)(&g_env, NULL, hm, hd) == trmse && trmse > 0.0, "rmse");
    jdoubleArray jyhat = new_array(8, (jsize)p.n_rows, NULL);
    N(predict)(&g_env, NULL, hm, hd, jyhat);
    double *tyhat = (double *)malloc((size_t)p.n_rows * 8);
    OK(fmhip_predict(tm, td, tyhat));
    CHECK(memcmp(ARR(jyhat)->data, tyhat, (size_t)p.n_rows * 8) == 0, "predict");
    /* predictRows: the first 100 rows as loose arrays; a scoring-only dataset of the same rows */
    jlongArray jrp100 = new_array(8, 101, p.rp);
    jdoubleArray jy100 = new_array(8, 100, NULL);
    N(predictRows)(&g_env, NULL, hm, 100, jrp100, jcol, jval, jy100);
    NO_THROW();
    double t100[100];
    OK(fmhip_predict_rows(tm, 100, p.rp, p.col, p.val, t100));      /* (loose rows take the plain forward: another summation order than the dataset's) */
    CHECK(memcmp(ARR(jy100)->data, t100, 100 * 8) == 0, "predictRows");
    for (int r = 0; r < 100; ++r) CHECK(t100[r] - tyhat[r] < 1e-5 && tyhat[r] - t100[r] < 1e-5, "predictRows vs predict, row %d", r);
    const jlong hrows = N(rowsCreate)(&g_env, NULL, 0, p.n_rows, jrp, jcol, jval, jy);
    NO_THROW();
    fmhip_dataset_t trows = NULL;
    double trmse_rows = 0.0;
    OK(fmhip_rows_create(0, p.n_rows, p.rp, p.col, p.val, p.y, &trows));
    OK(fmhip_rmse(tm, trows, &trmse_rows, NULL));
    const double jrmse_rows = N(rmse)(&g_env, NULL, hm, hrows);
    CHECK(jrmse_rows == trmse_rows, "rmse over a scoring-only dataset");
    CHECK(jrmse_rows - trmse < 1e-6 && trmse - jrmse_rows < 1e-6, "scoring-only rows vs the training dataset: %.9g vs %.9g", jrmse_rows, trmse);
    N(datasetDestroy)(&g_env, NULL, hrows);
    OK(fmhip_dataset_destroy(trows));

    /* a one-rank communicator over RCCL: plan, an epoch, an epoch in a given order — against the same calls on the twin */
    jbyteArray jid = N(commUniqueId)(&g_env, NULL);
    NO_THROW();
    CHECK(jid && ARR(jid)->len == FMHIP_UNIQUE_ID_BYTES, "commUniqueId");
    const jlong hc = N(commCreate)(&g_env, NULL, hm, jid, 0, 1);
    NO_THROW();
    fmhip_comm_t tc = NULL;
    unsigned char tid[FMHIP_UNIQUE_ID_BYTES];
    OK(fmhip_comm_unique_id(tid));
    OK(fmhip_comm_create(tm, tid, 0, 1, &tc));
    const double fr[2] = {0.12, 0.4};
    jdoubleArray jfr = new_array(8, 2, fr);
    N(dpExchange)(&g_env, NULL, hc, FMHIP_EXCHANGE_DENSE);
    N(dpPlan)(&g_env, NULL, hm, hd, hc, jfr);
    NO_THROW();
    OK(fmhip_dp_exchange(tc, FMHIP_EXCHANGE_DENSE));
    OK(fmhip_dp_plan(tm, td, tc, 2, fr, NULL));
    CHECK(N(dpPlanSteps)(&g_env, NULL, hc) == 3, "dpPlanSteps");
    N(dpEpoch)(&g_env, NULL, hm, hd, hc, eta, r0, rw, rv);
    OK(fmhip_dp_epoch(tm, td, tc, eta, r0, rw, rv, NULL));
    const int64_t order[3] = {2, 0, 1};
    jlongArray jorder = new_array(8, 3, order);
    N(dpEpochOrder)(&g_env, NULL, hm, hd, hc, eta, r0, rw, rv, jorder);
    NO_THROW();
    OK(fmhip_dp_epoch_order(tm, td, tc, eta, r0, rw, rv, order, 3, NULL));
    N(getParams)(&g_env, NULL, hm, o0, ow, ov);
    OK(fmhip_model_get_params(tm, &t0, tw, tv));
    CHECK(((double *)ARR(o0)->data)[0] == t0 && memcmp(ARR(ow)->data, tw, (size_t)p.n1 * 8) == 0 &&
              memcmp(ARR(ov)->data, tv, (size_t)p.n1 * p.k * 8) == 0, "parameters after the data-parallel epochs differ");
    /* an error of the library becomes a RuntimeException that carries fmhip_last_error() */
    const int64_t bad_order[3] = {0, 0, 1};
    jlongArray jbad = new_array(8, 3, bad_order);
    N(dpEpochOrder)(&g_env, NULL, hm, hd, hc, eta, r0, rw, rv, jbad);
    CHECK(g_state.n_thrown == 1 && strstr(g_state.thrown, "RuntimeException") && strstr(g_state.thrown, "permutation"), "a bad order must raise");
    g_state.n_thrown = 0;
    g_state.thrown[0] = 0;
    jbyteArray jshort = new_array(1, 5, NULL);
    CHECK(N(commCreate)(&g_env, NULL, hm, jshort, 0, 1) == 0 && g_state.n_thrown == 1 && strstr(g_state.thrown, "IllegalArgumentException"),
          "a short unique id must raise");
    g_state.n_thrown = 0;
    g_state.thrown[0] = 0;

    N(commDestroy)(&g_env, NULL, hc);
    N(datasetDestroy)(&g_env, NULL, hd);
    N(modelDestroy)(&g_env, NULL, hm);
    NO_THROW();
    OK(fmhip_comm_destroy(tc));
    OK(fmhip_dataset_destroy(td));
    OK(fmhip_model_destroy(tm));
    CHECK(g_state.live_copies == 0, "%d array copies were never released", g_state.live_copies);
}

int main(int argc, char **argv) {
    if (argc != 2 || (strcmp(argv[1], "host") && strcmp(argv[1], "gpu"))) {
        fprintf(stderr, "usage: jni_harness host|gpu\n");
        return 2;
    }
    if (!strcmp(argv[1], "host")) run_host();
    else run_gpu();
    printf("jni_harness %s: %d checks ok\n", argv[1], g_checks);
    return 0;
}
