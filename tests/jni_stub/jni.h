/* TEST STUB, not a JDK header: the few JNI names jvm/fmhip_jni.c uses, declared just precisely enough for
 * `gcc -fsyntax-only` to type-check that file in an image without a JDK (tests/test_host_cpu.py).  Nothing links
 * against it and nothing ships with it; a real build uses $JAVA_HOME/include/jni.h. */
#ifndef FMHIP_TEST_JNI_STUB_H
#define FMHIP_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef double jdouble;
typedef jint jsize;
typedef struct _jobject *jobject;
typedef jobject jclass, jarray, jbyteArray, jintArray, jlongArray, jdoubleArray;
#define JNIEXPORT
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jbyteArray (*NewByteArray)(JNIEnv *, jsize);
    jlongArray (*NewLongArray)(JNIEnv *, jsize);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, void *);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, void *);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, void *);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
    void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);
    void (*SetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, const jbyte *);
    void (*GetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, jdouble *);
    void (*SetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, const jdouble *);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
    void *(*GetPrimitiveArrayCritical)(JNIEnv *, jarray, void *);
    void (*ReleasePrimitiveArrayCritical)(JNIEnv *, jarray, void *, jint);
};
#endif
