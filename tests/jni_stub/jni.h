/*
 * tests/jni_stub/jni.h — COMPILE-CHECK STAND-IN for <jni.h>, test infrastructure only.
 *
 * The build image has no JDK.  This header declares, from the Java Native Interface specification's published C signatures,
 * just the types, macros and JNIEnv entries that jvm/fmhip_jni.c uses, so that tests/test_host_cpu.py can run the shim
 * through `gcc -fsyntax-only -Wall -Wextra -Werror`: every call into include/fmhip.h is then checked for argument count and
 * type (the drift a source-only file is exposed to), and tests/jni_harness.c can drive the compiled shim with an in-memory
 * JNIEnv.  It is NOT a JDK header, is never shipped, and nothing under jvm/ or sparkfm_amd/ includes it — a real build
 * uses $JAVA_HOME/include (jvm/fmhip_jni.c's header comment).  The order of the function table is not the JDK's: a library
 * compiled against this file must never be loaded into a JVM.
 */
#ifndef FMHIP_TEST_JNI_STUB_H
#define FMHIP_TEST_JNI_STUB_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1
#define JNI_FALSE 0
#define JNI_TRUE 1

/* jni_md.h, Linux LP64 */
typedef int jint;
typedef long jlong;
typedef signed char jbyte;
typedef unsigned char jboolean;
typedef double jdouble;
typedef float jfloat;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    void *test_state;       /* the harness's own: not part of JNI */
    jclass (JNICALL *FindClass)(JNIEnv *env, const char *name);
    jint (JNICALL *ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jsize (JNICALL *GetArrayLength)(JNIEnv *env, jarray array);
    jbyteArray (JNICALL *NewByteArray)(JNIEnv *env, jsize len);
    jlongArray (JNICALL *NewLongArray)(JNIEnv *env, jsize len);
    jint *(JNICALL *GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    jlong *(JNICALL *GetLongArrayElements)(JNIEnv *env, jlongArray array, jboolean *isCopy);
    jdouble *(JNICALL *GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
    void (JNICALL *ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    void (JNICALL *ReleaseLongArrayElements)(JNIEnv *env, jlongArray array, jlong *elems, jint mode);
    void (JNICALL *ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
    void (JNICALL *GetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, jbyte *buf);
    void (JNICALL *SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf);
    void (JNICALL *GetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, jdouble *buf);
    void (JNICALL *SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
    void (JNICALL *SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
    void *(JNICALL *GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);
    void (JNICALL *ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);
};

#endif
