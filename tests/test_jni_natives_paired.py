"""Every `public static native` method of PinotGpu.java has its JNI function in integration/jni/pinot_gpu_jni.c, with as many parameters
(JNIEnv*, jclass + the Java parameters), and the shim defines no function the Java class does not declare: the link error a JVM would
raise at first call (UnsatisfiedLinkError), found without a JVM."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "integration", "java", "org", "apache", "pinot", "gpu", "PinotGpu.java")
JNI = os.path.join(ROOT, "integration", "jni", "pinot_gpu_jni.c")


def java_natives():
    src = open(JAVA).read()
    out = {}
    for m in re.finditer(r"public\s+static\s+native\s+[\w\[\]<>]+\s+(\w+)\s*\(([^)]*)\)", src):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = len(params)
    return out


def jni_functions():
    src = open(JNI).read()
    out = {}
    # plain definitions: JNIEXPORT <type> JNICALL Java_org_apache_pinot_gpu_PinotGpu_<name>(JNIEnv* env, jclass c, ...)
    for m in re.finditer(r"Java_org_apache_pinot_gpu_PinotGpu_(\w+)\s*\(([^)]*)\)", src):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = len(params) - 2
    # the COPY_OUT / similar macros take `name(JNIEnv* env, jclass c, ...)` as their first argument
    for m in re.finditer(r"^[A-Z_]+\(\s*(\w+)\s*\((JNIEnv\*[^)]*)\)", src, re.M):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out.setdefault(m.group(1), len(params) - 2)
    return out


def test_every_native_has_its_jni_function():
    natives, jni = java_natives(), jni_functions()
    assert len(natives) > 40
    missing = sorted(set(natives) - set(jni))
    assert not missing, f"PinotGpu natives without a JNI function: {missing}"
    wrong = {n: (natives[n], jni[n]) for n in natives if natives[n] != jni[n]}
    assert not wrong, f"parameter counts differ (java, jni): {wrong}"
    extra = sorted(set(jni) - set(natives))
    assert not extra, f"JNI functions PinotGpu does not declare: {extra}"
