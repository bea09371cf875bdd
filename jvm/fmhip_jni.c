/*
 * fmhip_jni.c — JNI shim between jvm/HipSGD.scala and libfmhip.so (include/fmhip.h).
 *
 * The build image has no JDK (no jni.h): here this file is compiled (-Wall -Wextra -Werror) against the stand-in
 * tests/jni_stub/jni.h and its natives are driven, without a JVM, by tests/jni_harness.c through an in-memory JNIEnv that
 * copies arrays and poisons them on release; what that cannot cover is the real jni.h's ABI.  The file is
 * deliberately nothing but marshalling: every function obtains the Java arrays, calls ONE entry point of the
 * C ABI with plain pointers and sizes, releases them, and turns a non-zero status into a RuntimeException that
 * carries fmhip_last_error().
 *
 * Array access.  Calls that run long or block — the dataset build (a multi-second host pass, hipMalloc, stream syncs),
 * parameter transfers, scoring, the O(nnz) relabelling helpers — take the arrays with Get<Type>ArrayElements (the VM
 * may copy; the garbage collector keeps running).  GetPrimitiveArrayCritical is used only around the one short,
 * non-blocking call (shardRows: a binary search): a critical region locks out the collector for every thread of an
 * executor.
 *
 * Build on a machine with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jvm/fmhip_jni.c \
 *       -Lsparkfm_amd/lib -lfmhip -Wl,-rpath,'$ORIGIN' -o libfmhip_jni.so
 * Scala `object HipSGD` natives are static methods of class io.edstud.spark.fm.lib.HipSGD$ ("_00024" = '$').
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "fmhip.h"

#define JNI_FN(name) JNICALL Java_io_edstud_spark_fm_lib_HipSGD_00024_##name
#define H_MODEL(h) ((fmhip_model_t)(intptr_t)(h))
#define H_DATA(h) ((fmhip_dataset_t)(intptr_t)(h))
#define H_COMM(h) ((fmhip_comm_t)(intptr_t)(h))

static void raise(JNIEnv *env, int rc) {
    if (rc != FMHIP_OK)
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), fmhip_last_error());
}

/* views of the CSR arrays of one call (jlong == int64_t, jint == int32_t, jdouble == double); the calls that use them
 * are long ones: Get<Type>ArrayElements, not critical regions.  y_mode: how `y` is released (JNI_ABORT = an input,
 * 0 = an output to copy back) */
typedef struct { jlong *rp; jint *col; jdouble *val; jdouble *y; } csr_pins;

static int pin(JNIEnv *env, csr_pins *p, jlongArray rp, jintArray col, jdoubleArray val, jdoubleArray y) {
    p->rp = (*env)->GetLongArrayElements(env, rp, NULL);
    p->col = col ? (*env)->GetIntArrayElements(env, col, NULL) : NULL;
    p->val = val ? (*env)->GetDoubleArrayElements(env, val, NULL) : NULL;
    p->y = y ? (*env)->GetDoubleArrayElements(env, y, NULL) : NULL;
    return p->rp && (!col || p->col) && (!val || p->val) && (!y || p->y);     /* 0: OutOfMemoryError is pending */
}

static void unpin(JNIEnv *env, csr_pins *p, jlongArray rp, jintArray col, jdoubleArray val, jdoubleArray y, jint y_mode) {
    if (p->y) (*env)->ReleaseDoubleArrayElements(env, y, p->y, y_mode);
    if (p->val) (*env)->ReleaseDoubleArrayElements(env, val, p->val, JNI_ABORT);
    if (p->col) (*env)->ReleaseIntArrayElements(env, col, p->col, JNI_ABORT);
    if (p->rp) (*env)->ReleaseLongArrayElements(env, rp, p->rp, JNI_ABORT);
}

JNIEXPORT jlong JNI_FN(modelCreate)(JNIEnv *env, jobject o, jint dev, jlong n, jint k) {
    fmhip_model_t m = NULL;
    raise(env, fmhip_model_create(dev, n, k, NULL, &m));
    return (jlong)(intptr_t)m;
}

JNIEXPORT void JNI_FN(modelDestroy)(JNIEnv *env, jobject o, jlong h) { raise(env, fmhip_model_destroy(H_MODEL(h))); }

JNIEXPORT void JNI_FN(setParams)(JNIEnv *env, jobject o, jlong h, jdouble w0, jdoubleArray w, jdoubleArray v) {
    jdouble *pw = (*env)->GetDoubleArrayElements(env, w, NULL);
    jdouble *pv = (*env)->GetDoubleArrayElements(env, v, NULL);
    int rc = FMHIP_ERR_NOMEM;
    if (pw && pv) rc = fmhip_model_set_params(H_MODEL(h), w0, pw, pv);
    if (pv) (*env)->ReleaseDoubleArrayElements(env, v, pv, JNI_ABORT);
    if (pw) (*env)->ReleaseDoubleArrayElements(env, w, pw, JNI_ABORT);
    if (pw && pv) raise(env, rc);
}

JNIEXPORT void JNI_FN(getParams)(JNIEnv *env, jobject o, jlong h, jdoubleArray w0, jdoubleArray w, jdoubleArray v) {
    jdouble p0 = 0.0;
    jdouble *pw = (*env)->GetDoubleArrayElements(env, w, NULL);
    jdouble *pv = (*env)->GetDoubleArrayElements(env, v, NULL);
    int rc = FMHIP_ERR_NOMEM;
    if (pw && pv) rc = fmhip_model_get_params(H_MODEL(h), &p0, pw, pv);
    if (pv) (*env)->ReleaseDoubleArrayElements(env, v, pv, 0);        /* 0: copy back */
    if (pw) (*env)->ReleaseDoubleArrayElements(env, w, pw, 0);
    if (pw && pv) {
        (*env)->SetDoubleArrayRegion(env, w0, 0, 1, &p0);
        raise(env, rc);
    }
}

JNIEXPORT jlong JNI_FN(datasetCreate)(JNIEnv *env, jobject o, jint dev, jlong n_rows, jlongArray rp, jintArray col,
                                      jdoubleArray val, jdoubleArray y, jlong batch_rows) {
    csr_pins p;
    fmhip_dataset_t d = NULL;
    const int ok = pin(env, &p, rp, col, val, y);
    int rc = FMHIP_ERR_NOMEM;
    if (ok) rc = fmhip_dataset_create(dev, n_rows, (const int64_t *)p.rp, (const int32_t *)p.col, p.val, p.y, batch_rows, &d);
    unpin(env, &p, rp, col, val, y, JNI_ABORT);
    if (ok) raise(env, rc);
    return (jlong)(intptr_t)d;
}

JNIEXPORT jlong JNI_FN(rowsCreate)(JNIEnv *env, jobject o, jint dev, jlong n_rows, jlongArray rp, jintArray col,
                                   jdoubleArray val, jdoubleArray y) {
    csr_pins p;
    fmhip_dataset_t d = NULL;
    const int ok = pin(env, &p, rp, col, val, y);
    int rc = FMHIP_ERR_NOMEM;
    if (ok) rc = fmhip_rows_create(dev, n_rows, (const int64_t *)p.rp, (const int32_t *)p.col, p.val, p.y, &d);
    unpin(env, &p, rp, col, val, y, JNI_ABORT);
    if (ok) raise(env, rc);
    return (jlong)(intptr_t)d;
}

JNIEXPORT void JNI_FN(datasetDestroy)(JNIEnv *env, jobject o, jlong h) { raise(env, fmhip_dataset_destroy(H_DATA(h))); }

JNIEXPORT void JNI_FN(sgdEpoch)(JNIEnv *env, jobject o, jlong m, jlong d, jdouble eta, jdouble r0, jdouble rw, jdouble rv) {
    raise(env, fmhip_sgd_epoch(H_MODEL(m), H_DATA(d), eta, r0, rw, rv, NULL, NULL));
}

JNIEXPORT jdouble JNI_FN(rmse)(JNIEnv *env, jobject o, jlong m, jlong d) {
    double r = 0.0;
    raise(env, fmhip_rmse(H_MODEL(m), H_DATA(d), &r, NULL));
    return r;
}

JNIEXPORT void JNI_FN(predict)(JNIEnv *env, jobject o, jlong m, jlong d, jdoubleArray yhat) {
    jdouble *py = (*env)->GetDoubleArrayElements(env, yhat, NULL);
    if (!py) return;                                             /* OutOfMemoryError is pending */
    int rc = fmhip_predict(H_MODEL(m), H_DATA(d), py);
    (*env)->ReleaseDoubleArrayElements(env, yhat, py, 0);
    raise(env, rc);
}

JNIEXPORT void JNI_FN(predictRows)(JNIEnv *env, jobject o, jlong m, jlong n_rows, jlongArray rp, jintArray col,
                                   jdoubleArray val, jdoubleArray yhat) {
    csr_pins p;
    const int ok = pin(env, &p, rp, col, val, yhat);
    int rc = FMHIP_ERR_NOMEM;
    if (ok) rc = fmhip_predict_rows(H_MODEL(m), n_rows, (const int64_t *)p.rp, (const int32_t *)p.col, p.val, p.y);
    unpin(env, &p, rp, col, val, yhat, 0);     /* yhat is an output: copied back (mode 0); the inputs are released without a copy */
    if (ok) raise(env, rc);
}

JNIEXPORT jint JNI_FN(deviceCount)(JNIEnv *env, jobject o) {
    int n = 0;
    raise(env, fmhip_device_count(&n));
    return n;
}

JNIEXPORT jbyteArray JNI_FN(commUniqueId)(JNIEnv *env, jobject o) {
    jbyte id[FMHIP_UNIQUE_ID_BYTES];
    int rc = fmhip_comm_unique_id(id);
    raise(env, rc);
    if (rc != FMHIP_OK) return NULL;
    jbyteArray a = (*env)->NewByteArray(env, FMHIP_UNIQUE_ID_BYTES);
    if (a) (*env)->SetByteArrayRegion(env, a, 0, FMHIP_UNIQUE_ID_BYTES, id);
    return a;
}

JNIEXPORT jlong JNI_FN(commCreate)(JNIEnv *env, jobject o, jlong m, jbyteArray id, jint rank, jint world) {
    jbyte buf[FMHIP_UNIQUE_ID_BYTES];
    fmhip_comm_t c = NULL;
    if ((*env)->GetArrayLength(env, id) != FMHIP_UNIQUE_ID_BYTES) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "unique id must be 128 bytes");
        return 0;
    }
    (*env)->GetByteArrayRegion(env, id, 0, FMHIP_UNIQUE_ID_BYTES, buf);
    int rc = fmhip_comm_create(H_MODEL(m), buf, rank, world, &c);
    /* collective, like the creation: known patterns through every collective kind before a gradient is trusted to it */
    if (rc == FMHIP_OK && (rc = fmhip_comm_selftest(c, NULL)) != FMHIP_OK) {
        raise(env, rc);                  /* (the library's message first: destroying the communicator does not replace it) */
        fmhip_comm_destroy(c);
        return 0;
    }
    raise(env, rc);
    return (jlong)(intptr_t)c;
}

JNIEXPORT void JNI_FN(commDestroy)(JNIEnv *env, jobject o, jlong h) { raise(env, fmhip_comm_destroy(H_COMM(h))); }

JNIEXPORT void JNI_FN(dpPlan)(JNIEnv *env, jobject o, jlong m, jlong d, jlong c, jdoubleArray fractions) {
    jsize n = (*env)->GetArrayLength(env, fractions);
    jdouble f[FMHIP_DP_MAX_CUTS];
    if (n > FMHIP_DP_MAX_CUTS) n = FMHIP_DP_MAX_CUTS;
    (*env)->GetDoubleArrayRegion(env, fractions, 0, n, f);
    raise(env, fmhip_dp_plan(H_MODEL(m), H_DATA(d), H_COMM(c), (int)n, f, NULL));
}

JNIEXPORT void JNI_FN(dpEpoch)(JNIEnv *env, jobject o, jlong m, jlong d, jlong c, jdouble eta, jdouble r0, jdouble rw,
                               jdouble rv) {
    raise(env, fmhip_dp_epoch(H_MODEL(m), H_DATA(d), H_COMM(c), eta, r0, rw, rv, NULL));
}

JNIEXPORT jlong JNI_FN(dpPlanSteps)(JNIEnv *env, jobject o, jlong c) {
    int64_t steps = 0;
    raise(env, fmhip_dp_plan_info(H_COMM(c), &steps, NULL));
    return (jlong)steps;
}

JNIEXPORT void JNI_FN(dpEpochOrder)(JNIEnv *env, jobject o, jlong m, jlong d, jlong c, jdouble eta, jdouble r0, jdouble rw,
                                    jdouble rv, jlongArray order) {
    jsize n = (*env)->GetArrayLength(env, order);
    jlong *p = (*env)->GetLongArrayElements(env, order, NULL);       /* a long call: elements, not a critical region */
    if (!p) return;
    int rc = fmhip_dp_epoch_order(H_MODEL(m), H_DATA(d), H_COMM(c), eta, r0, rw, rv, (const int64_t *)p, (int64_t)n, NULL);
    (*env)->ReleaseLongArrayElements(env, order, p, JNI_ABORT);
    raise(env, rc);
}

/* what a data-parallel step exchanges: FMHIP_EXCHANGE_DENSE 0, _TOUCHED 1, _SHARDED 2 */
JNIEXPORT void JNI_FN(dpExchange)(JNIEnv *env, jobject o, jlong c, jint mode) { raise(env, fmhip_dp_exchange(H_COMM(c), mode)); }

JNIEXPORT jlongArray JNI_FN(shardRows)(JNIEnv *env, jobject o, jlongArray rp, jint world, jint rank) {
    jsize n = (*env)->GetArrayLength(env, rp);
    jlong *p = (*env)->GetPrimitiveArrayCritical(env, rp, NULL);
    int64_t lo = 0, hi = 0;
    int rc = fmhip_shard_rows((int64_t)n - 1, (const int64_t *)p, world, rank, &lo, &hi);
    (*env)->ReleasePrimitiveArrayCritical(env, rp, p, JNI_ABORT);
    raise(env, rc);
    if (rc != FMHIP_OK) return NULL;
    jlong out[2];
    out[0] = lo; out[1] = hi;
    jlongArray a = (*env)->NewLongArray(env, 2);
    if (a) (*env)->SetLongArrayRegion(env, a, 0, 2, out);
    return a;
}

/* feature relabelling by frequency (include/fmhip.h): counts accumulate into `counts`; rankFromCounts fills rank and
 * byRank; relabelColumns rewrites `col` in place */
JNIEXPORT void JNI_FN(featureCounts)(JNIEnv *env, jobject o, jintArray col, jlong n1, jlongArray counts) {
    jsize nnz = (*env)->GetArrayLength(env, col);
    jint *c = (*env)->GetIntArrayElements(env, col, NULL);
    jlong *k = (*env)->GetLongArrayElements(env, counts, NULL);
    int rc = FMHIP_ERR_NOMEM;
    if (c && k) rc = fmhip_feature_counts((int64_t)nnz, (const int32_t *)c, (int64_t)n1, (int64_t *)k);
    if (k) (*env)->ReleaseLongArrayElements(env, counts, k, 0);
    if (c) (*env)->ReleaseIntArrayElements(env, col, c, JNI_ABORT);
    if (c && k) raise(env, rc);
}

JNIEXPORT void JNI_FN(rankFromCounts)(JNIEnv *env, jobject o, jlongArray counts, jintArray rank, jintArray byRank) {
    jsize n1 = (*env)->GetArrayLength(env, counts);
    jlong *k = (*env)->GetLongArrayElements(env, counts, NULL);
    jint *r = (*env)->GetIntArrayElements(env, rank, NULL);
    jint *b = (*env)->GetIntArrayElements(env, byRank, NULL);
    int rc = FMHIP_ERR_NOMEM;
    if (k && r && b) rc = fmhip_rank_from_counts((int64_t)n1, (const int64_t *)k, (int32_t *)r, (int32_t *)b);
    if (b) (*env)->ReleaseIntArrayElements(env, byRank, b, 0);
    if (r) (*env)->ReleaseIntArrayElements(env, rank, r, 0);
    if (k) (*env)->ReleaseLongArrayElements(env, counts, k, JNI_ABORT);
    if (k && r && b) raise(env, rc);
}

JNIEXPORT void JNI_FN(relabelColumns)(JNIEnv *env, jobject o, jintArray col, jintArray rank) {
    jsize nnz = (*env)->GetArrayLength(env, col), n1 = (*env)->GetArrayLength(env, rank);
    jint *c = (*env)->GetIntArrayElements(env, col, NULL);
    jint *r = (*env)->GetIntArrayElements(env, rank, NULL);
    int rc = FMHIP_ERR_NOMEM;
    if (c && r) rc = fmhip_relabel_columns((int64_t)nnz, (const int32_t *)c, (int64_t)n1, (const int32_t *)r, (int32_t *)c);
    if (r) (*env)->ReleaseIntArrayElements(env, rank, r, JNI_ABORT);
    if (c) (*env)->ReleaseIntArrayElements(env, col, c, 0);
    if (c && r) raise(env, rc);
}
