/*
 * pinot_gpu_jni.c — JNI binding of libpinot_gpu.so for org.apache.pinot.gpu.PinotGpu (integration/java).  One JNI function per
 * pg_* entry point, no logic of its own: handles travel as jlong, buffers as (address, size) pairs of PinotDataBuffer
 * (pinot-segment-spi/.../memory/PinotDataBuffer.java:162-174 — columns exceed 2 GB, so no int-sized ByteBuffer views), results
 * are copied into caller-allocated primitive arrays (through a native staging buffer + Set<Type>ArrayRegion: the pg_* accessors may
 * launch kernels and synchronise a stream, which must not happen inside a Get/ReleasePrimitiveArrayCritical region), a status < 0 becomes a
 * RuntimeException carrying pg_last_error() (which BaseCombineOperator wraps with the segment name,
 * pinot-core/.../operator/combine/BaseCombineOperator.java:185-199); PG_ERR_CANCELLED becomes EarlyTerminationException
 * (pinot-core/.../operator/BaseOperator.java:44-46).
 *
 * Needs <jni.h>: there is no JDK in the image this repository is built in; build() compiles this file against the stand-in stub/jni.h
 * and runs it under the fake JNIEnv of jni_fake_env_test.c (type-checked and exercised, not JVM-linked).  The plain-C
 * half it relies on (pinot_gpu_shim.c) and the exact call sequence it performs are compiled and tested by jni_sequence_test.c.
 * Build on a JDK host:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include pinot_gpu_jni.c pinot_gpu_shim.c \
 *       -L../../pinot_amd/csrc -lpinot_gpu -o libpinot_gpu_jni.so
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "pinot_gpu.h"
#include "pinot_gpu_shim.h"

#define SEG(h) ((pg_segment_t)(intptr_t)(h))
#define RES(h) ((pg_result_t)(intptr_t)(h))
#define SET(h) ((pg_docidset_t)(intptr_t)(h))
#define BUF(addr, size) ((pg_buffer){(const void*)(intptr_t)(addr), (uint64_t)(size)})

static void throw_status(JNIEnv* env, int32_t status) {
  char msg[2048];
  pg_last_error(msg, sizeof msg);
  const char* cls = status == PG_ERR_CANCELLED ? "org/apache/pinot/core/query/exception/EarlyTerminationException"
                  : status == PG_ERR_UNSUPPORTED ? "java/lang/UnsupportedOperationException" : "java/lang/RuntimeException";
  jclass c = (*env)->FindClass(env, cls);
  if (c) (*env)->ThrowNew(env, c, msg);
}
#define CHECK(expr) do { int32_t _s = (expr); if (_s < 0) { throw_status(env, _s); return; } } while (0)
#define CHECK_RET(expr, ret) do { int32_t _s = (expr); if (_s < 0) { throw_status(env, _s); return (ret); } } while (0)

JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_abiVersion(JNIEnv* env, jclass c) { (void)env; (void)c; return pg_abi_version(); }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_init(JNIEnv* env, jclass c, jint device) { (void)c; CHECK(pg_init(device)); }
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_deviceCount(JNIEnv* env, jclass c) {
  (void)c; int32_t n = 0; CHECK_RET(pg_device_count(&n), 0); return n;
}

/* ---- segments ---------------------------------------------------------------------------------------------------------------- */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentCreate(JNIEnv* env, jclass c, jstring name, jint totalDocs, jint device) {
  (void)c;
  const char* n = (*env)->GetStringUTFChars(env, name, NULL);
  pg_segment_t s = NULL;
  const int32_t st = device < 0 ? pg_segment_create(n, totalDocs, &s) : pg_segment_create_on_device(n, totalDocs, device, &s);
  (*env)->ReleaseStringUTFChars(env, name, n);
  CHECK_RET(st, 0);
  return (jlong)(intptr_t)s;
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentAddColumn(JNIEnv* env, jclass c, jlong seg, jstring name, jint dataType,
    jint fwdEncoding, jboolean hasDictionary, jint cardinality, jint bitsPerValue, jboolean sorted, jint dictBytesPerValue,
    jint totalNumberOfEntries, jlong fwdAddr, jlong fwdSize, jlong dictAddr, jlong dictSize, jlong invAddr, jlong invSize) {
  (void)c;
  pg_column_desc d;
  memset(&d, 0, sizeof d);
  d.name = (*env)->GetStringUTFChars(env, name, NULL);
  d.data_type = dataType; d.fwd_encoding = fwdEncoding; d.has_dictionary = hasDictionary; d.cardinality = cardinality;
  d.bits_per_value = bitsPerValue; d.is_sorted = sorted; d.dict_bytes_per_value = dictBytesPerValue;
  d.total_number_of_entries = totalNumberOfEntries;   /* multi-value columns (PG_FWD_DICT_FIXED_BIT_MV): ColumnMetadata#getTotalNumberOfEntries */
  d.forward_index = BUF(fwdAddr, fwdSize); d.dictionary = BUF(dictAddr, dictSize); d.inverted_index = BUF(invAddr, invSize);
  const int32_t st = pg_segment_add_column(SEG(seg), &d);
  (*env)->ReleaseStringUTFChars(env, name, d.name);
  CHECK(st);
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentSetNullVector(JNIEnv* env, jclass c, jlong seg, jstring column, jlong addr, jlong size) {
  (void)c;
  const char* n = (*env)->GetStringUTFChars(env, column, NULL);
  const int32_t st = pg_segment_set_null_vector(SEG(seg), n, (const void*)(intptr_t)addr, (uint64_t)size);
  (*env)->ReleaseStringUTFChars(env, column, n);
  CHECK(st);
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentSetQueryableDocIds(JNIEnv* env, jclass c, jlong seg, jlong addr, jlong size) {
  (void)c; CHECK(pg_segment_set_queryable_doc_ids(SEG(seg), (const void*)(intptr_t)addr, (uint64_t)size));
}
/* dims / pairColumns: String[]; dimAddrSize / pairAddrSize: long[2n] = {address, size} per entry; pairFunctions / pairTypes: int[n] */
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentAddStarTree(JNIEnv* env, jclass c, jlong seg, jint numDocs, jint maxLeafRecords,
    jobjectArray dims, jlongArray dimAddrSize, jintArray pairFunctions, jintArray pairTypes, jobjectArray pairColumns,
    jlongArray pairAddrSize, jlong treeAddr, jlong treeSize) {
  (void)c;
  const jsize nd = (*env)->GetArrayLength(env, dims), np = (*env)->GetArrayLength(env, pairColumns);
  const char* dim_names[64]; pg_buffer dim_bufs[64]; pg_star_tree_pair pairs[64]; jstring held[128];
  if (nd > 64 || np > 64) { jclass x = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (x) (*env)->ThrowNew(env, x, "more than 64 star-tree dimensions / pairs"); return; }
  jlong* da = (*env)->GetLongArrayElements(env, dimAddrSize, NULL);
  jlong* pa = (*env)->GetLongArrayElements(env, pairAddrSize, NULL);
  jint* pf = (*env)->GetIntArrayElements(env, pairFunctions, NULL);
  jint* pt = (*env)->GetIntArrayElements(env, pairTypes, NULL);
  for (jsize i = 0; i < nd; i++) {
    held[i] = (jstring)(*env)->GetObjectArrayElement(env, dims, i);
    dim_names[i] = (*env)->GetStringUTFChars(env, held[i], NULL);
    dim_bufs[i] = BUF(da[2 * i], da[2 * i + 1]);
  }
  for (jsize i = 0; i < np; i++) {
    held[64 + i] = (jstring)(*env)->GetObjectArrayElement(env, pairColumns, i);
    pairs[i].function = pf[i]; pairs[i].data_type = pt[i];
    pairs[i].column = (*env)->GetStringUTFChars(env, held[64 + i], NULL);
    pairs[i].forward_index = BUF(pa[2 * i], pa[2 * i + 1]);
  }
  pg_star_tree_desc d;
  memset(&d, 0, sizeof d);
  d.num_docs = numDocs; d.n_dimensions = nd; d.n_pairs = np; d.max_leaf_records = maxLeafRecords;
  d.dimensions = dim_names; d.dimension_forward_indexes = dim_bufs; d.pairs = pairs; d.star_tree = BUF(treeAddr, treeSize);
  const int32_t st = pg_segment_add_star_tree(SEG(seg), &d);
  for (jsize i = 0; i < nd; i++) (*env)->ReleaseStringUTFChars(env, held[i], dim_names[i]);
  for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, held[64 + i], pairs[i].column);
  (*env)->ReleaseLongArrayElements(env, dimAddrSize, da, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, pairAddrSize, pa, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, pairFunctions, pf, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, pairTypes, pt, JNI_ABORT);
  CHECK(st);
}
/* DataSource#getRangeIndex: the `range_index` entry as BitSlicedRangeIndexReader reads it (BitSlicedRangeIndexReader.java:41-58) */
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentSetRangeIndex(JNIEnv* env, jclass c, jlong seg, jstring column, jlong addr, jlong size) {
  (void)c;
  const char* n = (*env)->GetStringUTFChars(env, column, NULL);
  const int32_t st = pg_segment_set_range_index(SEG(seg), n, (const void*)(intptr_t)addr, (uint64_t)size);
  (*env)->ReleaseStringUTFChars(env, column, n);
  CHECK(st);
}
/* native address of a direct ByteBuffer: GpuBuffers takes PinotDataBuffer addresses from one-byte views (toDirectByteBuffer(0, 1)) */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_directBufferAddress(JNIEnv* env, jclass c, jobject directBuffer) {
  (void)c;
  void* p = directBuffer ? (*env)->GetDirectBufferAddress(env, directBuffer) : NULL;
  if (!p) { jclass x = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (x) (*env)->ThrowNew(env, x, "not a direct buffer"); return 0; }
  return (jlong)(intptr_t)p;
}
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentDeviceBytes(JNIEnv* env, jclass c, jlong seg) {
  (void)c; uint64_t b = 0; CHECK_RET(pg_segment_device_bytes(SEG(seg), &b), 0); return (jlong)b;
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_segmentDestroy(JNIEnv* env, jclass c, jlong seg) { (void)c; CHECK(pg_segment_destroy(SEG(seg))); }

/* ---- queries: NativeQuery.java serialises the QueryContext into a direct ByteBuffer (pinot_gpu_shim.h) ------------------------------ */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_queryParse(JNIEnv* env, jclass c, jobject directBuffer, jint size) {
  (void)c;
  pgshim_query* q = NULL;
  char err[512];
  const int32_t st = pgshim_query_parse((*env)->GetDirectBufferAddress(env, directBuffer), (uint64_t)size, &q, err, sizeof err);
  if (st < 0) { jclass x = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (x) (*env)->ThrowNew(env, x, err); return 0; }
  return (jlong)(intptr_t)q;
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_queryFree(JNIEnv* env, jclass c, jlong q) { (void)env; (void)c; pgshim_query_free((pgshim_query*)(intptr_t)q); }
/* 0: the GPU plan takes the query; PG_ERR_UNSUPPORTED (-2): fall back to InstancePlanMakerImplV2; anything else throws */
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_querySupported(JNIEnv* env, jclass c, jlong seg, jlong q) {
  (void)c;
  const int32_t st = pg_query_supported(SEG(seg), pgshim_query_get((const pgshim_query*)(intptr_t)q));
  if (st < 0 && st != PG_ERR_UNSUPPORTED) throw_status(env, st);
  return st;
}
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_cancelCreate(JNIEnv* env, jclass c) { (void)c; pg_cancel_t t = NULL; CHECK_RET(pg_cancel_create(&t), 0); return (jlong)(intptr_t)t; }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_cancelRequest(JNIEnv* env, jclass c, jlong t) { (void)c; CHECK(pg_cancel_request((pg_cancel_t)(intptr_t)t)); }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_cancelReset(JNIEnv* env, jclass c, jlong t) { (void)c; CHECK(pg_cancel_reset((pg_cancel_t)(intptr_t)t)); }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_cancelDestroy(JNIEnv* env, jclass c, jlong t) { (void)c; CHECK(pg_cancel_destroy((pg_cancel_t)(intptr_t)t)); }
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_queryExec(JNIEnv* env, jclass c, jlong seg, jlong q, jlong cancel) {
  (void)c;
  pg_result_t r = NULL;
  CHECK_RET(pg_query_exec_cancellable(SEG(seg), pgshim_query_get((const pgshim_query*)(intptr_t)q), (pg_cancel_t)(intptr_t)cancel, &r), 0);
  return (jlong)(intptr_t)r;
}

/* ---- results ----------------------------------------------------------------------------------------------------------------------- */
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultNumGroups(JNIEnv* env, jclass c, jlong r) { (void)c; int32_t n = 0; CHECK_RET(pg_result_num_groups(RES(r), &n), 0); return n; }
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultKindOf(JNIEnv* env, jclass c, jlong r, jint agg) { (void)c; int32_t k = 0; CHECK_RET(pg_result_kind_of(RES(r), agg, &k), 0); return k; }
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultGroupKeyType(JNIEnv* env, jclass c, jlong r, jint col) { (void)c; int32_t k = 0; CHECK_RET(pg_result_group_key_type(RES(r), col, &k), 0); return k; }
/* Copy-out accessors: the native call fills a staging buffer (it may block: kernels, stream synchronisation, device copies — the JNI
 * specification forbids that between Get/ReleasePrimitiveArrayCritical), then one Set<Type>ArrayRegion moves the bytes.  The accessors
 * write only as many elements as the result holds, which may be fewer than the caller's array: the staging buffer therefore starts as a
 * copy of the array (Get<Type>ArrayRegion), so the elements the accessor leaves alone come back unchanged — never heap garbage. */
#include <stdlib.h>
#define COPY_OUT(NAME, JTYPE, CTYPE, GETREGION, SETREGION, CALL)                                                         \
  JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_##NAME {                                            \
    (void)c;                                                                                                    \
    if (!out) { jclass x = (*env)->FindClass(env, "java/lang/NullPointerException"); if (x) (*env)->ThrowNew(env, x, "output array"); return; } \
    const jsize n = (*env)->GetArrayLength(env, out);                                                           \
    CTYPE* p = (CTYPE*)malloc((size_t)(n > 0 ? n : 1) * sizeof(CTYPE));                                         \
    if (!p) { jclass x = (*env)->FindClass(env, "java/lang/OutOfMemoryError"); if (x) (*env)->ThrowNew(env, x, "staging buffer"); return; } \
    (*env)->GETREGION(env, out, 0, n, (JTYPE*)p);                                                               \
    const int32_t st = CALL;                                                                                    \
    if (st >= 0) (*env)->SETREGION(env, out, 0, n, (const JTYPE*)p);                                            \
    free(p);                                                                                                    \
    CHECK(st);                                                                                                  \
  }
COPY_OUT(resultGroupDictIds(JNIEnv* env, jclass c, jlong r, jint col, jintArray out), jint, int32_t, GetIntArrayRegion, SetIntArrayRegion, pg_result_group_dict_ids(RES(r), col, p, n))
COPY_OUT(resultGroupValuesLong(JNIEnv* env, jclass c, jlong r, jint col, jlongArray out), jlong, int64_t, GetLongArrayRegion, SetLongArrayRegion, pg_result_group_values_long(RES(r), col, p, n))
COPY_OUT(resultGroupValuesDouble(JNIEnv* env, jclass c, jlong r, jint col, jdoubleArray out), jdouble, double, GetDoubleArrayRegion, SetDoubleArrayRegion, pg_result_group_values_double(RES(r), col, p, n))
/* raw STRING / BYTES group keys: resultGroupValuesBytesSize, then offsets (numGroups + 1 longs) and the values back to back */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultGroupValuesBytesSize(JNIEnv* env, jclass c, jlong r, jint col) {
  (void)c;
  uint64_t total = 0;
  CHECK_RET(pg_result_group_values_bytes_size(RES(r), col, &total), 0);
  return (jlong)total;
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultGroupValuesBytes(JNIEnv* env, jclass c, jlong r, jint col, jlongArray offsets, jbyteArray out) {
  (void)c;
  if (!offsets || !out) { jclass x = (*env)->FindClass(env, "java/lang/NullPointerException"); if (x) (*env)->ThrowNew(env, x, "output array"); return; }
  const jsize n_off = (*env)->GetArrayLength(env, offsets), n_bytes = (*env)->GetArrayLength(env, out);
  int64_t* po = (int64_t*)calloc((size_t)(n_off > 0 ? n_off : 1), sizeof(int64_t));
  uint8_t* pb = (uint8_t*)calloc((size_t)(n_bytes > 0 ? n_bytes : 1), 1);
  if (!po || !pb) { free(po); free(pb); jclass x = (*env)->FindClass(env, "java/lang/OutOfMemoryError"); if (x) (*env)->ThrowNew(env, x, "staging buffer"); return; }
  /* the staging buffers start zeroed (calloc) and are written by the library, THEN copied out: nothing may touch them in between
   * (round 4 read the caller's still-empty arrays over the filled buffers here, so every raw STRING / BYTES key came back empty) */
  const int32_t st = pg_result_group_values_bytes(RES(r), col, po, (int32_t)n_off, pb, (uint64_t)n_bytes);
  if (st >= 0) {
    (*env)->SetLongArrayRegion(env, offsets, 0, n_off, (const jlong*)po);
    (*env)->SetByteArrayRegion(env, out, 0, n_bytes, (const jbyte*)pb);
  }
  free(po);
  free(pb);
  CHECK(st);
}
COPY_OUT(resultDoubles(JNIEnv* env, jclass c, jlong r, jint agg, jint comp, jdoubleArray out), jdouble, double, GetDoubleArrayRegion, SetDoubleArrayRegion, pg_result_doubles(RES(r), agg, comp, p, n))
COPY_OUT(resultLongs(JNIEnv* env, jclass c, jlong r, jint agg, jint comp, jlongArray out), jlong, int64_t, GetLongArrayRegion, SetLongArrayRegion, pg_result_longs(RES(r), agg, comp, p, n))
COPY_OUT(resultSetSizes(JNIEnv* env, jclass c, jlong r, jint agg, jintArray out), jint, int32_t, GetIntArrayRegion, SetIntArrayRegion, pg_result_set_sizes(RES(r), agg, p, n))
COPY_OUT(resultSetDictIds(JNIEnv* env, jclass c, jlong r, jint agg, jintArray out), jint, int32_t, GetIntArrayRegion, SetIntArrayRegion, pg_result_set_dict_ids(RES(r), agg, p, (int64_t)n))
COPY_OUT(resultSetValuesLong(JNIEnv* env, jclass c, jlong r, jint agg, jlongArray out), jlong, int64_t, GetLongArrayRegion, SetLongArrayRegion, pg_result_set_values_long(RES(r), agg, p, (int64_t)n))
COPY_OUT(resultSetValuesDouble(JNIEnv* env, jclass c, jlong r, jint agg, jdoubleArray out), jdouble, double, GetDoubleArrayRegion, SetDoubleArrayRegion, pg_result_set_values_double(RES(r), agg, p, (int64_t)n))
COPY_OUT(resultHllRegisters(JNIEnv* env, jclass c, jlong r, jint agg, jbyteArray out), jbyte, uint8_t, GetByteArrayRegion, SetByteArrayRegion, pg_result_hll_registers(RES(r), agg, p, (int64_t)n))
/* enableNullHandling: 1 where the group's result / key is NULL (pg_result_agg_nulls, pg_result_group_key_nulls) */
COPY_OUT(resultAggNulls(JNIEnv* env, jclass c, jlong r, jint agg, jbyteArray out), jbyte, uint8_t, GetByteArrayRegion, SetByteArrayRegion, pg_result_agg_nulls(RES(r), agg, p, n))
COPY_OUT(resultGroupKeyNulls(JNIEnv* env, jclass c, jlong r, jint col, jbyteArray out), jbyte, uint8_t, GetByteArrayRegion, SetByteArrayRegion, pg_result_group_key_nulls(RES(r), col, p, n))
/* The result's DataTableImplV4 bytes (pg_result_data_table_v4): size first, then into a byte[] of that size — DataTableFactory.getDataTable(bytes)
 * on the Java side gives the DataTable a results block hands to InstanceResponseOperator without boxing a group */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultDataTableV4Size(JNIEnv* env, jclass c, jlong r) {
  (void)c;
  int64_t n = 0;
  CHECK_RET(pg_result_data_table_v4(RES(r), NULL, 0, &n), 0);
  return (jlong)n;
}
COPY_OUT(resultDataTableV4(JNIEnv* env, jclass c, jlong r, jbyteArray out), jbyte, uint8_t, GetByteArrayRegion, SetByteArrayRegion, pg_result_data_table_v4(RES(r), p, (int64_t)n, &(int64_t){0}))
/* out[0..4] = numDocsScanned, numEntriesScannedInFilter, numEntriesScannedPostFilter, numTotalDocs, numGroupsLimitReached */
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultStats(JNIEnv* env, jclass c, jlong r, jlongArray out) {
  (void)c;
  pg_exec_stats s;
  CHECK(pg_result_stats(RES(r), &s));
  const jlong v[5] = {s.num_docs_scanned, s.num_entries_scanned_in_filter, s.num_entries_scanned_post_filter, s.num_total_docs, s.num_groups_limit_reached};
  (*env)->SetLongArrayRegion(env, out, 0, 5, v);
}
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultFree(JNIEnv* env, jclass c, jlong r) { (void)c; CHECK(pg_result_free(RES(r))); }

/* ---- filter-only offload (FilterOperatorUtils.setImplementation): BaseFilterOperator#getTrues as a bitmap ---------------------------- */
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_filterExec(JNIEnv* env, jclass c, jlong seg, jlong q) {
  (void)c;
  pg_docidset_t s = NULL;
  if (!q) { jclass x = (*env)->FindClass(env, "java/lang/NullPointerException"); if (x) (*env)->ThrowNew(env, x, "query"); return 0; }
  const pg_query* pq = pgshim_query_get((const pgshim_query*)(intptr_t)q);
  CHECK_RET(pg_filter_exec_flags(SEG(seg), pq->filter, pq->flags, &s), 0);   /* the query record's enableNullHandling: three-valued getTrues */
  return (jlong)(intptr_t)s;
}
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_docIdSetCardinality(JNIEnv* env, jclass c, jlong s) { (void)c; int64_t n = 0; CHECK_RET(pg_docidset_cardinality(SET(s), &n), 0); return n; }
COPY_OUT(docIdSetCopyWords(JNIEnv* env, jclass c, jlong s, jlongArray out), jlong, uint64_t, GetLongArrayRegion, SetLongArrayRegion, pg_docidset_copy_words(SET(s), p, (int64_t)n))
COPY_OUT(docIdSetCopyDocIds(JNIEnv* env, jclass c, jlong s, jintArray out), jint, int32_t, GetIntArrayRegion, SetIntArrayRegion, pg_docidset_copy_docids(SET(s), p, (int64_t)n))
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_docIdSetFree(JNIEnv* env, jclass c, jlong s) { (void)c; CHECK(pg_docidset_free(SET(s))); }

/* ---- GroupByCombineOperator in the library: segments sharing their key space merge element-wise in HBM (pinot_gpu.h) ------------------ */
#define COMM(h) ((pg_comm_t)(intptr_t)(h))
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultMerge(JNIEnv* env, jclass c, jlong dst, jlong src) { (void)c; CHECK(pg_result_merge(RES(dst), RES(src))); }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_resultAllReduce(JNIEnv* env, jclass c, jlong r, jlong comm) { (void)c; CHECK(pg_result_all_reduce(RES(r), COMM(comm))); }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_commGetUniqueId(JNIEnv* env, jclass c, jbyteArray out) {
  (void)c;
  uint8_t id[PG_COMM_UNIQUE_ID_BYTES];
  if (!out || (*env)->GetArrayLength(env, out) < PG_COMM_UNIQUE_ID_BYTES) {
    jclass x = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (x) (*env)->ThrowNew(env, x, "unique id needs 128 bytes"); return;
  }
  CHECK(pg_comm_get_unique_id(id));
  (*env)->SetByteArrayRegion(env, out, 0, PG_COMM_UNIQUE_ID_BYTES, (const jbyte*)id);
}
JNIEXPORT jlong JNICALL Java_org_apache_pinot_gpu_PinotGpu_commInitRank(JNIEnv* env, jclass c, jint device, jint world, jint rank, jbyteArray uniqueId) {
  (void)c;
  uint8_t id[PG_COMM_UNIQUE_ID_BYTES];
  if (!uniqueId || (*env)->GetArrayLength(env, uniqueId) < PG_COMM_UNIQUE_ID_BYTES) {
    jclass x = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (x) (*env)->ThrowNew(env, x, "unique id needs 128 bytes"); return 0;
  }
  (*env)->GetByteArrayRegion(env, uniqueId, 0, PG_COMM_UNIQUE_ID_BYTES, (jbyte*)id);
  pg_comm_t comm = NULL;
  CHECK_RET(pg_comm_init_rank(device, world, rank, id, &comm), 0);
  return (jlong)(intptr_t)comm;
}
/* one communicator per listed device (one JVM, N GPUs): outComms[i] belongs to devices[i] */
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_commInitAll(JNIEnv* env, jclass c, jintArray devices, jlongArray outComms) {
  (void)c;
  if (!devices || !outComms) { jclass x = (*env)->FindClass(env, "java/lang/NullPointerException"); if (x) (*env)->ThrowNew(env, x, "devices / outComms"); return; }
  const jsize n = (*env)->GetArrayLength(env, devices);
  if (n < 1 || n > 64 || (*env)->GetArrayLength(env, outComms) < n) {
    jclass x = (*env)->FindClass(env, "java/lang/IllegalArgumentException"); if (x) (*env)->ThrowNew(env, x, "1..64 devices and as many output slots"); return;
  }
  jint* d = (*env)->GetIntArrayElements(env, devices, NULL);
  int32_t ords[64];
  for (jsize i = 0; i < n; i++) ords[i] = d[i];
  (*env)->ReleaseIntArrayElements(env, devices, d, JNI_ABORT);
  pg_comm_t comms[64];
  CHECK(pg_comm_init_all(n, ords, comms));
  jlong handles[64];
  for (jsize i = 0; i < n; i++) handles[i] = (jlong)(intptr_t)comms[i];
  (*env)->SetLongArrayRegion(env, outComms, 0, n, handles);
}
JNIEXPORT jint JNICALL Java_org_apache_pinot_gpu_PinotGpu_commWorldSize(JNIEnv* env, jclass c, jlong comm) { (void)c; int32_t w = 0; CHECK_RET(pg_comm_world_size(COMM(comm), &w), 0); return w; }
JNIEXPORT void JNICALL Java_org_apache_pinot_gpu_PinotGpu_commDestroy(JNIEnv* env, jclass c, jlong comm) { (void)c; CHECK(pg_comm_destroy(COMM(comm))); }
