/*
 * pinot_gpu_jni.c under a fake JNIEnv: the JNI functions themselves (not a restatement of what they do — jni_sequence_test.c is that)
 * are compiled against stub/jni.h and called the way the JVM calls them for org.apache.pinot.gpu.PinotGpu's native methods, with
 * strings, arrays and a direct buffer served by a small C environment that also records the exception a call left pending.
 *     init -> segmentCreate -> segmentAddColumn x2 -> queryParse(direct buffer) -> querySupported -> cancelCreate -> queryExec ->
 *     resultNumGroups -> resultGroupKeyType -> resultGroupDictIds -> resultKindOf / resultLongs / resultDoubles -> resultStats ->
 *     cancelRequest -> queryExec (EarlyTerminationException pending) -> resultFree -> queryFree -> segmentDestroy
 * Without a HIP device: init must leave a RuntimeException pending whose message says there is no CPU fallback; queryParse and its
 * IllegalArgumentException on a corrupt record run on the host alone.
 * Build: gcc -std=gnu11 -Wall -Wextra -Iintegration/jni/stub -Iinclude -Iintegration/jni integration/jni/jni_fake_env_test.c \
 *            integration/jni/pinot_gpu_shim.c -Lpinot_amd/csrc -lpinot_gpu -Wl,-rpath,$PWD/pinot_amd/csrc
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pinot_gpu_jni.c"   /* the unit under test, against stub/jni.h */

/* ---- fake objects --------------------------------------------------------------------------------------------------------------- */
enum { K_STRING = 1, K_ARRAY, K_BUFFER, K_CLASS };
struct _jobject { int kind; jsize length; size_t elem; void* data; };
static struct _jobject* new_object(int kind, jsize length, size_t elem, void* data) {
  struct _jobject* o = (struct _jobject*)calloc(1, sizeof *o);
  o->kind = kind; o->length = length; o->elem = elem; o->data = data;
  return o;
}
static jstring new_string(const char* s) { return new_object(K_STRING, (jsize)strlen(s), 1, strdup(s)); }
static jarray new_array(jsize n, size_t elem) { return new_object(K_ARRAY, n, elem, calloc((size_t)(n > 0 ? n : 1), elem)); }
static jobject new_direct_buffer(void* addr) { return new_object(K_BUFFER, 0, 1, addr); }

static char pending_class[256], pending_message[2048];
static int pins;   /* Get…Elements / Get…Critical / GetStringUTFChars not yet released */

static jclass f_FindClass(JNIEnv* env, const char* name) { (void)env; return new_object(K_CLASS, 0, 1, strdup(name)); }
static jint f_ThrowNew(JNIEnv* env, jclass c, const char* m) {
  (void)env;
  snprintf(pending_class, sizeof pending_class, "%s", (const char*)c->data);
  snprintf(pending_message, sizeof pending_message, "%s", m ? m : "");
  return 0;
}
static const char* f_GetStringUTFChars(JNIEnv* env, jstring s, jboolean* is_copy) { (void)env; if (is_copy) *is_copy = 0; pins++; return (const char*)s->data; }
static void f_ReleaseStringUTFChars(JNIEnv* env, jstring s, const char* utf) { (void)env; if (utf == (const char*)s->data) pins--; }
static jsize f_GetArrayLength(JNIEnv* env, jarray a) { (void)env; return a->length; }
static jobject f_GetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize i) { (void)env; return ((jobject*)a->data)[i]; }
static jint* f_GetIntArrayElements(JNIEnv* env, jintArray a, jboolean* c) { (void)env; if (c) *c = 0; pins++; return (jint*)a->data; }
static jlong* f_GetLongArrayElements(JNIEnv* env, jlongArray a, jboolean* c) { (void)env; if (c) *c = 0; pins++; return (jlong*)a->data; }
static void f_ReleaseIntArrayElements(JNIEnv* env, jintArray a, jint* e, jint mode) { (void)env; (void)mode; if (e == (jint*)a->data) pins--; }
static void f_ReleaseLongArrayElements(JNIEnv* env, jlongArray a, jlong* e, jint mode) { (void)env; (void)mode; if (e == (jlong*)a->data) pins--; }
static void f_SetLongArrayRegion(JNIEnv* env, jlongArray a, jsize start, jsize len, const jlong* buf) {
  (void)env;
  if (start < 0 || len < 0 || start + len > a->length) { snprintf(pending_class, sizeof pending_class, "java/lang/ArrayIndexOutOfBoundsException"); return; }
  memcpy((jlong*)a->data + start, buf, (size_t)len * sizeof(jlong));
}
static void* f_GetPrimitiveArrayCritical(JNIEnv* env, jarray a, jboolean* c) { (void)env; if (c) *c = 0; pins++; return a->data; }
static void f_ReleasePrimitiveArrayCritical(JNIEnv* env, jarray a, void* p, jint mode) { (void)env; (void)mode; if (p == a->data) pins--; }
static void* f_GetDirectBufferAddress(JNIEnv* env, jobject b) { (void)env; return b->kind == K_BUFFER ? b->data : NULL; }
#define F_SET_REGION(NAME, T)                                                                                                \
  static void NAME(JNIEnv* env, jarray a, jsize start, jsize len, const T* buf) {                                            \
    (void)env;                                                                                                               \
    if (start < 0 || len < 0 || start + len > a->length) { snprintf(pending_class, sizeof pending_class, "java/lang/ArrayIndexOutOfBoundsException"); return; } \
    memcpy((T*)a->data + start, buf, (size_t)len * sizeof(T));                                                               \
  }
F_SET_REGION(f_SetIntArrayRegion, jint)
F_SET_REGION(f_SetDoubleArrayRegion, jdouble)
F_SET_REGION(f_SetByteArrayRegion, jbyte)
#define F_GET_REGION(NAME, T)                                                                                                \
  static void NAME(JNIEnv* env, jarray a, jsize start, jsize len, T* buf) {                                                  \
    (void)env;                                                                                                               \
    if (start < 0 || len < 0 || start + len > a->length) { snprintf(pending_class, sizeof pending_class, "java/lang/ArrayIndexOutOfBoundsException"); return; } \
    memcpy(buf, (T*)a->data + start, (size_t)len * sizeof(T));                                                               \
  }
F_GET_REGION(f_GetByteArrayRegion, jbyte)
F_GET_REGION(f_GetIntArrayRegion, jint)
F_GET_REGION(f_GetLongArrayRegion, jlong)
F_GET_REGION(f_GetDoubleArrayRegion, jdouble)

static const struct JNINativeInterface_ fake_functions = {
  f_FindClass, f_ThrowNew, f_GetStringUTFChars, f_ReleaseStringUTFChars, f_GetArrayLength, f_GetObjectArrayElement, f_GetIntArrayElements,
  f_GetLongArrayElements, f_ReleaseIntArrayElements, f_ReleaseLongArrayElements, f_SetLongArrayRegion, f_GetPrimitiveArrayCritical,
  f_ReleasePrimitiveArrayCritical, f_GetDirectBufferAddress, f_SetIntArrayRegion, f_SetDoubleArrayRegion, f_SetByteArrayRegion, f_GetByteArrayRegion,
  f_GetIntArrayRegion, f_GetLongArrayRegion, f_GetDoubleArrayRegion,
};

static int pending(void) { return pending_class[0] != 0; }
static void clear_pending(void) { pending_class[0] = 0; pending_message[0] = 0; }
static int fail(const char* what) {
  fprintf(stderr, "%s: pending %s: %s (pins %d)\n", what, pending_class[0] ? pending_class : "(none)", pending_message, pins);
  return 1;
}

/* ---- the segment and the query of jni_sequence_test.c --------------------------------------------------------------------------- */
#define N_DOCS 4000
static void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static void put_le16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put_le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
typedef struct { uint8_t b[4096]; size_t n; } record;
static void w_i32(record* r, int32_t v) { put_le32(r->b + r->n, (uint32_t)v); r->n += 4; }
static void w_str(record* r, const char* s) {
  if (!s) { w_i32(r, -1); return; }
  const size_t len = strlen(s);
  w_i32(r, (int32_t)len);
  memcpy(r->b + r->n, s, len);
  r->n += len;
  while (r->n & 3) r->b[r->n++] = 0;
}
static void w_predicate(record* r, int type, const char* column, int n_values, const char* const* values, const char* lower, const char* upper) {
  w_i32(r, PG_FILTER_PREDICATE); w_i32(r, 0);
  w_i32(r, type); w_i32(r, n_values); w_str(r, column);
  for (int i = 0; i < n_values; i++) w_str(r, values[i]);
  w_str(r, lower); w_str(r, upper); w_i32(r, 1); w_i32(r, 1);
}
/* SELECT d, COUNT(*), SUM(m), MAX(m) FROM t WHERE d IN (20, 30) AND m BETWEEN 100 AND 2999 GROUP BY d — NativeQuery.java's record */
static void build_record(record* r, int32_t flags) {
  r->n = 0;
  w_i32(r, PGSHIM_QUERY_MAGIC); w_i32(r, flags); w_i32(r, 0); w_i32(r, 0);
  w_i32(r, 1); w_i32(r, 3); w_i32(r, 1); w_i32(r, 0);
  w_i32(r, 10); w_i32(r, -1);   /* limit, minSegmentGroupTrimSize (off) */
  w_str(r, "d");
  w_i32(r, PG_AGG_COUNT); w_i32(r, 0); w_str(r, "*");
  w_i32(r, PG_AGG_SUM); w_i32(r, 0); w_str(r, "m");
  w_i32(r, PG_AGG_MAX); w_i32(r, 0); w_str(r, "m");
  w_i32(r, PG_FILTER_AND); w_i32(r, 2);
  const char* in_values[2] = {"20", "30"};
  w_predicate(r, PG_PRED_IN, "d", 2, in_values, NULL, NULL);
  w_predicate(r, PG_PRED_RANGE, "m", 0, NULL, "100", "2999");
}

int main(void) {
  JNIEnv env_value = &fake_functions;
  JNIEnv* env = &env_value;
  jclass cls = NULL;

  if (Java_org_apache_pinot_gpu_PinotGpu_abiVersion(env, cls) != PG_ABI_VERSION) return fail("abiVersion");

  /* queryParse from a direct buffer, and its IllegalArgumentException */
  static record rec;
  build_record(&rec, 0);
  jobject direct = new_direct_buffer(rec.b);
  const jlong q = Java_org_apache_pinot_gpu_PinotGpu_queryParse(env, cls, direct, (jint)rec.n);
  if (!q || pending()) return fail("queryParse");
  rec.b[0] ^= 0xFF;
  if (Java_org_apache_pinot_gpu_PinotGpu_queryParse(env, cls, direct, (jint)rec.n) != 0 || strcmp(pending_class, "java/lang/IllegalArgumentException") != 0 ||
      !strstr(pending_message, "bad magic")) return fail("queryParse(corrupt)");
  rec.b[0] ^= 0xFF;
  clear_pending();

  const jint n_dev = Java_org_apache_pinot_gpu_PinotGpu_deviceCount(env, cls);
  if (pending()) return fail("deviceCount");
  if (n_dev <= 0) {   /* the product has no CPU path: init must say so through a RuntimeException */
    Java_org_apache_pinot_gpu_PinotGpu_init(env, cls, 0);
    if (strcmp(pending_class, "java/lang/RuntimeException") != 0 || !strstr(pending_message, "no CPU fallback")) return fail("init without a device");
    Java_org_apache_pinot_gpu_PinotGpu_queryFree(env, cls, q);
    printf("no HIP device: init left %s pending (%s); jni under the fake env (host part) ok\n", pending_class, pending_message);
    return pins == 0 ? 0 : 1;
  }

  Java_org_apache_pinot_gpu_PinotGpu_init(env, cls, 0);
  if (pending()) return fail("init");
  /* Pinot-format bytes of two columns: d (dictionary, 2 bits, inverted index), m (raw INT chunks) */
  static uint8_t dict[16], fwd[(N_DOCS * 2 + 7) / 8], inv[20 + 4 * (16 + 2 * (N_DOCS / 4))], raw[32 + 4 * N_DOCS];
  for (int i = 0; i < 4; i++) put_be32(dict + 4 * i, (uint32_t)(10 * (i + 1)));
  for (int doc = 0; doc < N_DOCS; doc++) { const int id = doc % 4, bit = doc * 2; fwd[bit >> 3] |= (uint8_t)(id << (6 - (bit & 7))); }
  size_t pos = 20;
  for (int id = 0; id < 4; id++) {
    put_be32(inv + 4 * id, (uint32_t)pos);
    uint8_t* b = inv + pos;
    put_le32(b, 12346); put_le32(b + 4, 1);
    put_le16(b + 8, 0); put_le16(b + 10, (uint16_t)(N_DOCS / 4 - 1));
    put_le32(b + 12, 16);
    for (int k = 0; k < N_DOCS / 4; k++) put_le16(b + 16 + 2 * k, (uint16_t)(4 * k + id));
    pos += 16 + 2 * (size_t)(N_DOCS / 4);
  }
  put_be32(inv + 16, (uint32_t)pos);
  const uint32_t hdr[8] = {2, 1, N_DOCS, 4, N_DOCS, 0, 28, 32};
  for (int i = 0; i < 8; i++) put_be32(raw + 4 * i, hdr[i]);
  for (int doc = 0; doc < N_DOCS; doc++) put_be32(raw + 32 + 4 * doc, (uint32_t)doc);

  const jlong seg = Java_org_apache_pinot_gpu_PinotGpu_segmentCreate(env, cls, new_string("jni_fake_env"), N_DOCS, 0);
  if (!seg || pending()) return fail("segmentCreate");
  Java_org_apache_pinot_gpu_PinotGpu_segmentAddColumn(env, cls, seg, new_string("d"), PG_TYPE_INT, PG_FWD_DICT_FIXED_BIT, 1, 4, 2, 0, 4, 0,
                                                      (jlong)(intptr_t)fwd, (jlong)sizeof fwd, (jlong)(intptr_t)dict, (jlong)sizeof dict, (jlong)(intptr_t)inv, (jlong)pos);
  if (pending()) return fail("segmentAddColumn(d)");
  Java_org_apache_pinot_gpu_PinotGpu_segmentAddColumn(env, cls, seg, new_string("m"), PG_TYPE_INT, PG_FWD_RAW_FIXED_BYTE_CHUNK, 0, 0, 0, 0, 0, 0,
                                                      (jlong)(intptr_t)raw, (jlong)sizeof raw, 0, 0, 0, 0);
  if (pending()) return fail("segmentAddColumn(m)");
  if (Java_org_apache_pinot_gpu_PinotGpu_segmentDeviceBytes(env, cls, seg) <= 0 || pending()) return fail("segmentDeviceBytes");
  /* a column that does not parse: RuntimeException with the library's message, nothing left pinned */
  Java_org_apache_pinot_gpu_PinotGpu_segmentAddColumn(env, cls, seg, new_string("broken"), PG_TYPE_INT, PG_FWD_RAW_FIXED_BYTE_CHUNK, 0, 0, 0, 0, 0, 0,
                                                      (jlong)(intptr_t)raw, 8, 0, 0, 0, 0);
  if (!pending() || pins != 0) return fail("segmentAddColumn(broken) should throw");
  clear_pending();

  if (Java_org_apache_pinot_gpu_PinotGpu_querySupported(env, cls, seg, q) != 0 || pending()) return fail("querySupported");
  const jlong cancel = Java_org_apache_pinot_gpu_PinotGpu_cancelCreate(env, cls);
  const jlong res = Java_org_apache_pinot_gpu_PinotGpu_queryExec(env, cls, seg, q, cancel);
  if (!res || pending()) return fail("queryExec");
  const jint ng = Java_org_apache_pinot_gpu_PinotGpu_resultNumGroups(env, cls, res);
  if (ng != 2 || Java_org_apache_pinot_gpu_PinotGpu_resultGroupKeyType(env, cls, res, 0) != PG_GROUP_KEY_DICT_IDS) return fail("resultNumGroups / KeyType");
  jintArray ids = new_array(ng, sizeof(jint));
  jlongArray counts = new_array(ng, sizeof(jlong)), stats = new_array(5, sizeof(jlong));
  jdoubleArray sums = new_array(ng, sizeof(jdouble)), maxs = new_array(ng, sizeof(jdouble));
  Java_org_apache_pinot_gpu_PinotGpu_resultGroupDictIds(env, cls, res, 0, ids);
  if (Java_org_apache_pinot_gpu_PinotGpu_resultKindOf(env, cls, res, 0) != PG_RESULT_LONG) return fail("resultKindOf");
  Java_org_apache_pinot_gpu_PinotGpu_resultLongs(env, cls, res, 0, 0, counts);
  Java_org_apache_pinot_gpu_PinotGpu_resultDoubles(env, cls, res, 1, 0, sums);
  Java_org_apache_pinot_gpu_PinotGpu_resultDoubles(env, cls, res, 2, 0, maxs);
  Java_org_apache_pinot_gpu_PinotGpu_resultStats(env, cls, res, stats);
  if (pending() || pins != 0) return fail("result accessors");
  int ok = 1;
  for (int g = 0; g < ng; g++) {
    const jint id = ((jint*)ids->data)[g];
    int64_t c = 0; double s = 0, mx = -1;
    for (int doc = 100; doc <= 2999; doc++) if (doc % 4 == id) { c++; s += doc; mx = doc; }
    printf("d=%d count=%lld sum=%.0f max=%.0f\n", 10 * (id + 1), (long long)((jlong*)counts->data)[g], ((jdouble*)sums->data)[g], ((jdouble*)maxs->data)[g]);
    ok = ok && (id == 1 || id == 2) && ((jlong*)counts->data)[g] == c && ((jdouble*)sums->data)[g] == s && ((jdouble*)maxs->data)[g] == mx;
  }
  const jlong* st = (const jlong*)stats->data;
  ok = ok && st[0] == 1450 && st[1] == 2000 && st[3] == N_DOCS;
  /* an array that is too short: the library's capacity check surfaces as an exception, the array stays unpinned */
  Java_org_apache_pinot_gpu_PinotGpu_resultLongs(env, cls, res, 0, 0, new_array(1, sizeof(jlong)));
  ok = ok && pending() && pins == 0;
  clear_pending();
  /* cancellation: EarlyTerminationException, as BaseOperator#nextBlock throws it; a reset token serves the next query */
  Java_org_apache_pinot_gpu_PinotGpu_cancelRequest(env, cls, cancel);
  ok = ok && Java_org_apache_pinot_gpu_PinotGpu_queryExec(env, cls, seg, q, cancel) == 0 && strstr(pending_class, "EarlyTerminationException") != NULL;
  clear_pending();
  Java_org_apache_pinot_gpu_PinotGpu_cancelReset(env, cls, cancel);
  {
    const jlong again = Java_org_apache_pinot_gpu_PinotGpu_queryExec(env, cls, seg, q, cancel);
    ok = ok && again != 0 && !pending();
    if (again) Java_org_apache_pinot_gpu_PinotGpu_resultFree(env, cls, again);
  }
  /* directBufferAddress: what GpuBuffers.address(PinotDataBuffer) takes from a one-byte view; a non-direct object throws */
  ok = ok && Java_org_apache_pinot_gpu_PinotGpu_directBufferAddress(env, cls, direct) == (jlong)(intptr_t)rec.b && !pending();
  ok = ok && Java_org_apache_pinot_gpu_PinotGpu_directBufferAddress(env, cls, new_string("heap")) == 0 && strcmp(pending_class, "java/lang/IllegalArgumentException") == 0;
  clear_pending();
  /* a range index that does not parse: the library's message in a RuntimeException, the column keeps its scan leaf */
  Java_org_apache_pinot_gpu_PinotGpu_segmentSetRangeIndex(env, cls, seg, new_string("m"), (jlong)(intptr_t)raw, 16);
  ok = ok && pending() && pins == 0;
  clear_pending();
  /* raw STRING group keys (resultGroupValuesBytes handed back the caller's empty arrays): a no-dictionary STRING column in the
     var-byte chunk layout (VarByteChunkForwardIndexWriter, v2, PASS_THROUGH), SELECT s, COUNT(*) FROM t GROUP BY s — the keys come back
     through resultGroupValuesBytesSize + resultGroupValuesBytes as offsets[numGroups + 1] and the values back to back */
  {
    static const char* const names[3] = {"a", "bb", "ccc"};
    enum { PER_CHUNK = 1000, CHUNKS = N_DOCS / PER_CHUNK };
    static uint8_t var[28 + 4 * CHUNKS + N_DOCS * 4 + N_DOCS * 3];
    const uint32_t vh[7] = {2, CHUNKS, PER_CHUNK, 3, N_DOCS, 0, 28};
    for (int i = 0; i < 7; i++) put_be32(var + 4 * i, vh[i]);
    size_t vpos = 28 + 4 * CHUNKS;
    for (int ch = 0; ch < CHUNKS; ch++) {
      put_be32(var + 28 + 4 * ch, (uint32_t)vpos);
      uint8_t* base = var + vpos;
      size_t at = 4 * PER_CHUNK;
      for (int k = 0; k < PER_CHUNK; k++) {
        const char* v = names[(ch * PER_CHUNK + k) % 3];
        put_be32(base + 4 * k, (uint32_t)at);
        memcpy(base + at, v, strlen(v));
        at += strlen(v);
      }
      vpos += at;
    }
    Java_org_apache_pinot_gpu_PinotGpu_segmentAddColumn(env, cls, seg, new_string("s"), PG_TYPE_STRING, PG_FWD_RAW_VAR_BYTE_CHUNK, 0, 0, 0, 0, 0, 0,
                                                        (jlong)(intptr_t)var, (jlong)vpos, 0, 0, 0, 0);
    ok = ok && !pending();
    static record rs;
    rs.n = 0;
    w_i32(&rs, PGSHIM_QUERY_MAGIC); w_i32(&rs, 0); w_i32(&rs, 0); w_i32(&rs, 0);
    w_i32(&rs, 1); w_i32(&rs, 1); w_i32(&rs, 0); w_i32(&rs, 0);
    w_i32(&rs, 10); w_i32(&rs, -1);
    w_str(&rs, "s");
    w_i32(&rs, PG_AGG_COUNT); w_i32(&rs, 0); w_str(&rs, "*");
    const jlong qs = Java_org_apache_pinot_gpu_PinotGpu_queryParse(env, cls, new_direct_buffer(rs.b), (jint)rs.n);
    const jlong rr = qs ? Java_org_apache_pinot_gpu_PinotGpu_queryExec(env, cls, seg, qs, 0) : 0;
    ok = ok && qs && rr && !pending();
    if (rr) {
      const jint n_keys = Java_org_apache_pinot_gpu_PinotGpu_resultNumGroups(env, cls, rr);
      const jlong total = Java_org_apache_pinot_gpu_PinotGpu_resultGroupValuesBytesSize(env, cls, rr, 0);
      ok = ok && n_keys == 3 && total == 6 && !pending();
      jlongArray offs = new_array(n_keys + 1, sizeof(jlong)), cnt = new_array(n_keys, sizeof(jlong));
      jbyteArray bytes = new_array((jsize)total, sizeof(jbyte));
      Java_org_apache_pinot_gpu_PinotGpu_resultGroupValuesBytes(env, cls, rr, 0, offs, bytes);
      Java_org_apache_pinot_gpu_PinotGpu_resultLongs(env, cls, rr, 0, 0, cnt);
      ok = ok && !pending() && pins == 0 && ((jlong*)offs->data)[0] == 0 && ((jlong*)offs->data)[n_keys] == total;
      int seen = 0;
      for (int g = 0; g < n_keys && ok; g++) {
        const jlong b0 = ((jlong*)offs->data)[g], b1 = ((jlong*)offs->data)[g + 1];
        const int len = (int)(b1 - b0);
        ok = ok && len >= 1 && len <= 3 && memcmp((const char*)bytes->data + b0, names[len - 1], (size_t)len) == 0;
        int64_t want = 0;
        for (int doc = 0; doc < N_DOCS; doc++) want += (doc % 3) == len - 1;
        ok = ok && ((jlong*)cnt->data)[g] == want;
        printf("s=%.*s count=%lld\n", len, (const char*)bytes->data + b0, (long long)((jlong*)cnt->data)[g]);
        seen |= 1 << (len - 1);
      }
      ok = ok && seen == 7;
      /* a byte array that is too short: the library's capacity check, as an exception */
      Java_org_apache_pinot_gpu_PinotGpu_resultGroupValuesBytes(env, cls, rr, 0, offs, new_array(2, sizeof(jbyte)));
      ok = ok && pending() && pins == 0;
      clear_pending();
      Java_org_apache_pinot_gpu_PinotGpu_resultFree(env, cls, rr);
    }
    if (qs) Java_org_apache_pinot_gpu_PinotGpu_queryFree(env, cls, qs);
    if (!ok) return fail("raw STRING group keys through resultGroupValuesBytes");
  }
  /* GroupByCombineOperator in the library: two results of the same query kept in HBM (PG_QUERY_FLAG_KEEP_DEVICE_TABLE) merge element-wise;
     a world-of-one communicator (pg_comm_init_all over device 0) all-reduces a result onto itself */
  {
    static record rec_keep;
    build_record(&rec_keep, PG_QUERY_FLAG_KEEP_DEVICE_TABLE);
    const jlong qk = Java_org_apache_pinot_gpu_PinotGpu_queryParse(env, cls, new_direct_buffer(rec_keep.b), (jint)rec_keep.n);
    const jlong ra = Java_org_apache_pinot_gpu_PinotGpu_queryExec(env, cls, seg, qk, 0), rb = Java_org_apache_pinot_gpu_PinotGpu_queryExec(env, cls, seg, qk, 0);
    ok = ok && qk && ra && rb && !pending();
    Java_org_apache_pinot_gpu_PinotGpu_resultMerge(env, cls, ra, rb);
    ok = ok && !pending();
    jlongArray merged = new_array(ng, sizeof(jlong));
    Java_org_apache_pinot_gpu_PinotGpu_resultLongs(env, cls, ra, 0, 0, merged);
    for (int g = 0; g < ng; g++) ok = ok && ((jlong*)merged->data)[g] == 2 * ((jlong*)counts->data)[g];
    jintArray devs = new_array(1, sizeof(jint));
    jlongArray comms = new_array(1, sizeof(jlong));
    Java_org_apache_pinot_gpu_PinotGpu_commInitAll(env, cls, devs, comms);
    if (pending()) {   /* no librccl on this host: the binding surfaced the library's error; nothing else to check here */
      printf("commInitAll: %s (%s)\n", pending_class, pending_message);
      clear_pending();
    } else {
      const jlong comm = ((jlong*)comms->data)[0];
      ok = ok && comm != 0 && Java_org_apache_pinot_gpu_PinotGpu_commWorldSize(env, cls, comm) == 1;
      Java_org_apache_pinot_gpu_PinotGpu_resultAllReduce(env, cls, rb, comm);
      ok = ok && !pending();
      Java_org_apache_pinot_gpu_PinotGpu_resultLongs(env, cls, rb, 0, 0, merged);
      for (int g = 0; g < ng; g++) ok = ok && ((jlong*)merged->data)[g] == ((jlong*)counts->data)[g];
      jbyteArray uid = new_array(PG_COMM_UNIQUE_ID_BYTES, sizeof(jbyte));
      Java_org_apache_pinot_gpu_PinotGpu_commGetUniqueId(env, cls, uid);
      ok = ok && !pending();
      Java_org_apache_pinot_gpu_PinotGpu_commGetUniqueId(env, cls, new_array(8, sizeof(jbyte)));
      ok = ok && strcmp(pending_class, "java/lang/IllegalArgumentException") == 0;
      clear_pending();
      Java_org_apache_pinot_gpu_PinotGpu_commDestroy(env, cls, comm);
      ok = ok && !pending();
    }
    Java_org_apache_pinot_gpu_PinotGpu_resultFree(env, cls, ra);
    Java_org_apache_pinot_gpu_PinotGpu_resultFree(env, cls, rb);
    Java_org_apache_pinot_gpu_PinotGpu_queryFree(env, cls, qk);
    ok = ok && !pending() && pins == 0;
  }
  Java_org_apache_pinot_gpu_PinotGpu_cancelDestroy(env, cls, cancel);
  Java_org_apache_pinot_gpu_PinotGpu_resultFree(env, cls, res);
  Java_org_apache_pinot_gpu_PinotGpu_queryFree(env, cls, q);
  Java_org_apache_pinot_gpu_PinotGpu_segmentDestroy(env, cls, seg);
  ok = ok && !pending() && pins == 0;
  printf(ok ? "jni under the fake env ok\n" : "jni under the fake env FAILED\n");
  return ok ? 0 : 1;
}
