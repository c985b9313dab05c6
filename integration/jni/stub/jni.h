/*
 * jni.h — STAND-IN for the JDK's header, for type-checking and exercising pinot_gpu_jni.c where no JDK exists (this image).
 *
 * NOT the JDK's file and not ABI-compatible with a JVM: the declarations below are written from the Java Native Interface
 * Specification (types of chapter 3, the functions of chapter 4 that pinot_gpu_jni.c calls, with their specified signatures), and
 * the function table holds ONLY those functions, in an order of its own.  Code compiled against it runs under the fake environment
 * of jni_fake_env_test.c, never under a JVM; on a JDK host the real <jni.h> is first on the include path and this file is unused.
 */
#ifndef PINOT_GPU_STUB_JNI_H
#define PINOT_GPU_STUB_JNI_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv* env, const char* name);
  jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* message);
  const char* (*GetStringUTFChars)(JNIEnv* env, jstring string, jboolean* isCopy);
  void (*ReleaseStringUTFChars)(JNIEnv* env, jstring string, const char* utf);
  jsize (*GetArrayLength)(JNIEnv* env, jarray array);
  jobject (*GetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index);
  jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
  jlong* (*GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* isCopy);
  void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  void (*ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
  void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
  void* (*GetPrimitiveArrayCritical)(JNIEnv* env, jarray array, jboolean* isCopy);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv* env, jarray array, void* carray, jint mode);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
  void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
  void (*SetDoubleArrayRegion)(JNIEnv* env, jdoubleArray array, jsize start, jsize len, const jdouble* buf);
  void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
  void (*GetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, jbyte* buf);
  void (*GetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, jint* buf);
  void (*GetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, jlong* buf);
  void (*GetDoubleArrayRegion)(JNIEnv* env, jdoubleArray array, jsize start, jsize len, jdouble* buf);
};
#endif
