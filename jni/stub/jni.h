/* jni/stub/jni.h — a compile-only stand-in for the JDK's <jni.h>, covering exactly what jni/demi_jni.c uses.
 * This image has no JDK; CI (tests/test_host_cpu.py) compiles the shim against this header so that it cannot rot.  On a
 * box with a JDK `make -C jni` uses $JAVA_HOME/include instead and this file is not involved.  Types and the JNIEnv calling
 * convention ((*env)->Fn(env, ...)) follow the JNI specification; the function table holds only the members used.   */
#ifndef DEMI_STUB_JNI_H
#define DEMI_STUB_JNI_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef int16_t jshort; typedef uint8_t jboolean; typedef jint jsize;
struct _jobject; typedef struct _jobject* jobject;
typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jarray jbyteArray; typedef jarray jshortArray;
typedef jarray jintArray; typedef jarray jlongArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_ERR (-1)
#define JNI_VERSION_1_6 0x00010006
struct JavaVM_; typedef struct JavaVM_ JavaVM;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jsize (*GetArrayLength)(JNIEnv*, jarray);
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  jbyte* (*GetByteArrayElements)(JNIEnv*, jbyteArray, jboolean*);
  jshort* (*GetShortArrayElements)(JNIEnv*, jshortArray, jboolean*);
  jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
  jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
  void (*ReleaseByteArrayElements)(JNIEnv*, jbyteArray, jbyte*, jint);
  void (*ReleaseShortArrayElements)(JNIEnv*, jshortArray, jshort*, jint);
  void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
  void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
  void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
  void (*GetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, jint*);
  void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*);
  void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
  jboolean (*ExceptionCheck)(JNIEnv*);
};
#endif
