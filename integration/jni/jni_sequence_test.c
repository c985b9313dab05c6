/*
 * The JNI call sequence, without a JVM: what GpuSegmentRegistry / NativeQuery / GpuGroupByOperator (integration/java) make
 * pinot_gpu_jni.c do, driven from C with the same arguments — (address, size) buffers of Pinot-format bytes, the query as a
 * NativeQuery record in a flat byte buffer (pinot_gpu_shim.h), results into caller-allocated arrays.
 *     segmentCreate -> segmentAddColumn x2 -> queryParse -> querySupported -> cancelCreate -> queryExec -> resultNumGroups ->
 *     resultGroupDictIds -> resultKindOf / resultLongs / resultDoubles -> resultStats -> resultFree -> queryFree -> segmentDestroy
 * Without a GPU it checks the record parser (round trip, truncation, bad magic) and the loud failure of pg_init.
 * Build: gcc -std=c99 -Wall -Wextra -pedantic -Werror -Iinclude -Iintegration/jni integration/jni/jni_sequence_test.c \
 *            integration/jni/pinot_gpu_shim.c -Lpinot_amd/csrc -lpinot_gpu -Wl,-rpath,$PWD/pinot_amd/csrc
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pinot_gpu.h"
#include "pinot_gpu_shim.h"

#define N_DOCS 4000

static void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static void put_le16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put_le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* ---- NativeQuery.java's writer, in C ------------------------------------------------------------------------------------------- */
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
static void w_predicate(record* r, int predicate_type, const char* column, int n_values, const char* const* values, const char* lower,
                        const char* upper, int lower_inclusive, int upper_inclusive) {
  w_i32(r, PG_FILTER_PREDICATE); w_i32(r, 0);
  w_i32(r, predicate_type); w_i32(r, n_values); w_str(r, column);
  for (int i = 0; i < n_values; i++) w_str(r, values[i]);
  w_str(r, lower); w_str(r, upper); w_i32(r, lower_inclusive); w_i32(r, upper_inclusive);
}
/* SELECT d, COUNT(*), SUM(m), MAX(m) FROM t WHERE d IN (20, 30) AND m BETWEEN 100 AND 2999 GROUP BY d ORDER BY SUM(m) DESC, d LIMIT 7
 * with minSegmentGroupTrimSize 3 (the order-by block of PGQ2: segment-level group trim) */
static void build_record(record* r) {
  r->n = 0;
  w_i32(r, PGSHIM_QUERY_MAGIC); w_i32(r, 0); w_i32(r, 0); w_i32(r, 0);
  w_i32(r, 1); w_i32(r, 3); w_i32(r, 1); w_i32(r, 2);
  w_i32(r, 7); w_i32(r, 3);
  w_str(r, "d");
  w_i32(r, PG_AGG_COUNT); w_i32(r, 0); w_str(r, "*");
  w_i32(r, PG_AGG_SUM); w_i32(r, 0); w_str(r, "m");
  w_i32(r, PG_AGG_MAX); w_i32(r, 0); w_str(r, "m");
  w_i32(r, PG_ORDER_BY_AGGREGATION); w_i32(r, 1); w_i32(r, 0); w_i32(r, 0);
  w_i32(r, PG_ORDER_BY_GROUP_KEY); w_i32(r, 0); w_i32(r, 1); w_i32(r, 1);
  w_i32(r, PG_FILTER_AND); w_i32(r, 2);
  const char* in_values[2] = {"20", "30"};
  w_predicate(r, PG_PRED_IN, "d", 2, in_values, NULL, NULL, 0, 0);
  w_predicate(r, PG_PRED_RANGE, "m", 0, NULL, "100", "2999", 1, 1);
}

static int check(int32_t st, const char* what) {
  if (st == PG_OK) return 0;
  char msg[512];
  pg_last_error(msg, sizeof msg);
  fprintf(stderr, "%s failed (%d): %s\n", what, st, msg);
  return 1;
}

int main(void) {
  /* ---- the record parser (no device needed) ------------------------------------------------------------------------------------- */
  record rec;
  build_record(&rec);
  pgshim_query* nq = NULL;
  char err[256];
  if (pgshim_query_parse(rec.b, rec.n, &nq, err, sizeof err) != PG_OK) { fprintf(stderr, "parse: %s\n", err); return 1; }
  const pg_query* q = pgshim_query_get(nq);
  int ok = q->n_group_by == 1 && strcmp(q->group_by_columns[0], "d") == 0 && q->n_aggregations == 3 && q->n_order_by == 2 && q->limit == 7 &&
           q->min_segment_group_trim_size == 3 && q->order_by[0].kind == PG_ORDER_BY_AGGREGATION && q->order_by[0].index == 1 &&
           q->order_by[0].ascending == 0 && q->order_by[1].kind == PG_ORDER_BY_GROUP_KEY && q->order_by[1].nulls_last == 1 &&
           q->aggregations[1].function == PG_AGG_SUM && strcmp(q->aggregations[1].column, "m") == 0 && q->filter &&
           q->filter->type == PG_FILTER_AND && q->filter->n_children == 2 && q->filter->children[0].predicate_type == PG_PRED_IN &&
           q->filter->children[0].n_values == 2 && strcmp(q->filter->children[0].values[1], "30") == 0 &&
           q->filter->children[1].predicate_type == PG_PRED_RANGE && strcmp(q->filter->children[1].upper, "2999") == 0 &&
           q->filter->children[1].upper_inclusive == 1 && q->filter->children[0].lower == NULL;
  if (!ok) { fprintf(stderr, "record round trip mismatch\n"); return 1; }
  pgshim_query* bad = NULL;
  for (size_t cut = 0; cut < rec.n; cut += 7)   /* every truncation is refused, none crashes */
    if (pgshim_query_parse(rec.b, cut, &bad, err, sizeof err) == PG_OK) { fprintf(stderr, "truncated record accepted at %zu\n", cut); return 1; }
  rec.b[0] ^= 0xFF;
  if (pgshim_query_parse(rec.b, rec.n, &bad, err, sizeof err) == PG_OK || !strstr(err, "bad magic")) { fprintf(stderr, "bad magic accepted\n"); return 1; }
  rec.b[0] ^= 0xFF;
  printf("NativeQuery record: %zu bytes, parsed, truncations refused\n", rec.n);

  int32_t n_dev = 0;
  if (check(pg_device_count(&n_dev), "pg_device_count")) return 1;
  if (n_dev <= 0) {
    const int32_t st = pg_init(0);
    pgshim_query_free(nq);
    printf("no HIP device: pg_init -> %d; jni sequence (host part) ok\n", st);
    return st == PG_ERR_DEVICE ? 0 : 1;
  }

  /* ---- PinotGpu.init / GpuSegmentRegistry.handleFor ----------------------------------------------------------------------------------- */
  if (check(pg_init(0), "pg_init")) return 1;
  uint8_t dict[16];
  for (int i = 0; i < 4; i++) put_be32(dict + 4 * i, (uint32_t)(10 * (i + 1)));
  static uint8_t fwd[(N_DOCS * 2 + 7) / 8];
  for (int doc = 0; doc < N_DOCS; doc++) { const int id = doc % 4, bit = doc * 2; fwd[bit >> 3] |= (uint8_t)(id << (6 - (bit & 7))); }
  static uint8_t inv[20 + 4 * (16 + 2 * (N_DOCS / 4))];
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
  static uint8_t raw[32 + 4 * N_DOCS];
  const uint32_t hdr[8] = {2, 1, N_DOCS, 4, N_DOCS, 0, 28, 32};
  for (int i = 0; i < 8; i++) put_be32(raw + 4 * i, hdr[i]);
  for (int doc = 0; doc < N_DOCS; doc++) put_be32(raw + 32 + 4 * doc, (uint32_t)doc);

  pg_segment_t seg = NULL;
  if (check(pg_segment_create_on_device("jni_sequence", N_DOCS, 0, &seg), "segmentCreate")) return 1;
  pg_column_desc d;
  memset(&d, 0, sizeof d);
  d.name = "d"; d.data_type = PG_TYPE_INT; d.fwd_encoding = PG_FWD_DICT_FIXED_BIT; d.has_dictionary = 1; d.cardinality = 4;
  d.bits_per_value = 2; d.dict_bytes_per_value = 4;
  d.forward_index.addr = fwd; d.forward_index.size = sizeof fwd;
  d.dictionary.addr = dict; d.dictionary.size = sizeof dict;
  d.inverted_index.addr = inv; d.inverted_index.size = pos;
  if (check(pg_segment_add_column(seg, &d), "segmentAddColumn(d)")) return 1;
  memset(&d, 0, sizeof d);
  d.name = "m"; d.data_type = PG_TYPE_INT; d.fwd_encoding = PG_FWD_RAW_FIXED_BYTE_CHUNK;
  d.forward_index.addr = raw; d.forward_index.size = sizeof raw;
  if (check(pg_segment_add_column(seg, &d), "segmentAddColumn(m)")) return 1;

  /* ---- GpuInstancePlanMaker.makeSegmentPlanNode -> GpuGroupByOperator.getNextBlock -------------------------------------------------- */
  if (check(pg_query_supported(seg, q), "querySupported")) return 1;
  pg_cancel_t cancel = NULL;
  if (check(pg_cancel_create(&cancel), "cancelCreate")) return 1;
  pg_result_t res = NULL;
  if (check(pg_query_exec_cancellable(seg, q, cancel, &res), "queryExec")) return 1;
  int32_t ng = 0, kind = -1;
  pg_result_num_groups(res, &ng);
  int32_t ids[4]; int64_t counts[4]; double sums[4], maxs[4];
  ok = ng == 2 && pg_result_group_dict_ids(res, 0, ids, 4) == PG_OK && pg_result_kind_of(res, 0, &kind) == PG_OK && kind == PG_RESULT_LONG &&
       pg_result_longs(res, 0, 0, counts, 4) == PG_OK && pg_result_doubles(res, 1, 0, sums, 4) == PG_OK && pg_result_doubles(res, 2, 0, maxs, 4) == PG_OK;
  pg_exec_stats st;
  pg_result_stats(res, &st);
  for (int g = 0; g < ng && ok; g++) {
    /* docs with dictId id and 100 <= doc <= 2999: doc = 4k + id */
    int64_t c = 0; double s = 0, mx = -1;
    for (int doc = 100; doc <= 2999; doc++) if (doc % 4 == ids[g]) { c++; s += doc; mx = doc; }
    printf("d=%d count=%lld sum=%.0f max=%.0f\n", 10 * (ids[g] + 1), (long long)counts[g], sums[g], maxs[g]);
    ok = ok && (ids[g] == 1 || ids[g] == 2) && counts[g] == c && sums[g] == s && maxs[g] == mx;
  }
  /* AND(inverted, scan): the scan evaluates the 2000 docs the inverted index lets through */
  ok = ok && st.num_docs_scanned == 1450 && st.num_entries_scanned_in_filter == 2000 && st.num_total_docs == N_DOCS && st.stats_exact == 1;
  /* a cancelled token: EarlyTerminationException on the Java side */
  pg_cancel_request(cancel);
  pg_result_t none = NULL;
  ok = ok && pg_query_exec_cancellable(seg, q, cancel, &none) == PG_ERR_CANCELLED && none == NULL;
  pg_cancel_destroy(cancel);
  pg_result_free(res);
  pgshim_query_free(nq);
  pg_segment_destroy(seg);
  printf(ok ? "jni sequence ok\n" : "jni sequence FAILED\n");
  return ok ? 0 : 1;
}
