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


def _declared_methods(path):
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"(?:public|static|private|protected)[\w\s<>\[\],.?]*?\s(\w+)\s*\(([^)]*)\)\s*(?:throws [\w., ]+)?\s*[;{]", text):
        params = m.group(2)
        depth, n, cur = 0, 0, ""
        for ch in params:
            depth += ch in "<(" and 1 or 0
            depth -= ch in ">)" and 1 or 0
            if ch == "," and depth == 0:
                n += 1
                cur = ""
            else:
                cur += ch
        out.setdefault(m.group(1), set()).add(n + (1 if cur.strip() else 0))
    return out


def test_static_calls_between_the_plugin_classes_resolve():
    """`OtherClass.method(args)` between the classes of integration/java: declared there, with that many parameters."""
    src_dir = os.path.dirname(JAVA)
    classes = {n[:-5]: _declared_methods(os.path.join(src_dir, n)) for n in os.listdir(src_dir) if n.endswith(".java")}
    problems = []
    for name in sorted(classes):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(src_dir, name + ".java")).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
        for other in classes:
            if other == name:
                continue
            for m in re.finditer(r"(?<![\w.])" + other + r"\.(\w+)\s*\(", text):
                method = m.group(1)
                i, depth = m.end(), 1
                while depth and i < len(text):
                    depth += text[i] == "(" and 1 or 0
                    depth -= text[i] == ")" and 1 or 0
                    i += 1
                args = text[m.end():i - 1]
                d2, n, cur = 0, 0, ""
                for ch in args:
                    d2 += ch in "([{" and 1 or 0
                    d2 -= ch in ")]}" and 1 or 0
                    if ch == "," and d2 == 0:
                        n += 1
                        cur = ""
                    else:
                        cur += ch
                n_args = n + (1 if cur.strip() else 0)
                if method not in classes[other]:
                    problems.append(f"{name}.java: {other}.{method} is not declared")
                elif n_args not in classes[other][method]:
                    problems.append(f"{name}.java: {other}.{method} called with {n_args} arguments, declared with {sorted(classes[other][method])}")
    assert not problems, problems


def test_every_pinotgpu_call_site_names_a_declared_method_with_that_arity():
    """`PinotGpu.name(args)` in the other Java sources: the method exists in PinotGpu.java and takes that many arguments (javac's "cannot find
    symbol" / "method cannot be applied" for this class, without javac)."""
    src_dir = os.path.dirname(JAVA)
    pg = open(JAVA).read()
    declared = {}
    for m in re.finditer(r"(?:public|static|private)[\w\s<>\[\]]*?\s(\w+)\s*\(([^)]*)\)\s*(?:throws [\w., ]+)?\s*[;{]", pg):
        declared.setdefault(m.group(1), set()).add(len([p for p in m.group(2).split(",") if p.strip()]))

    def split_args(s):
        depth, n, cur = 0, 0, ""
        for ch in s:
            if ch in "([{<":
                depth += 1
            elif ch in ")]}>":
                depth -= 1
            if ch == "," and depth == 0:
                n += 1
                cur = ""
            else:
                cur += ch
        return n + (1 if cur.strip() else 0)
    problems = []
    for name in sorted(os.listdir(src_dir)):
        if not name.endswith(".java") or name == "PinotGpu.java":
            continue
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(src_dir, name)).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"PinotGpu\.(\w+)\s*\(", text):
            method = m.group(1)
            i, depth = m.end(), 1
            while depth and i < len(text):
                depth += text[i] in "(" and 1 or 0
                depth -= text[i] in ")" and 1 or 0
                i += 1
            n_args = split_args(text[m.end():i - 1])
            if method not in declared:
                problems.append(f"{name}: PinotGpu.{method} is not declared")
            elif n_args not in declared[method]:
                problems.append(f"{name}: PinotGpu.{method} called with {n_args} arguments, declared with {sorted(declared[method])}")
    assert not problems, problems
