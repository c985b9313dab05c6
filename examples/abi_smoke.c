/*
 * Plain-C caller of the drop-in boundary (include/pinot_gpu.h), the way the JNI shim of INTEGRATION.md calls it: registers a
 * tiny segment (one dictionary column with an inverted index, one raw INT metric) from Pinot-format bytes built here, runs
 *     SELECT d, COUNT(*), SUM(m) FROM t WHERE d IN (1, 2) GROUP BY d
 * and prints the groups.  Without a GPU it stops after the ABI / error-path checks (the library has no CPU fallback).
 * Build: gcc -std=c99 -Wall -pedantic -Iinclude examples/abi_smoke.c -Lpinot_amd/csrc -lpinot_gpu -Wl,-rpath,$PWD/pinot_amd/csrc
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pinot_gpu.h"

#define N_DOCS 1000

static void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static void put_le16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put_le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

static int check(int32_t st, const char* what) {
  if (st == PG_OK) return 0;
  char msg[512];
  pg_last_error(msg, sizeof msg);
  fprintf(stderr, "%s failed (%d): %s\n", what, st, msg);
  return 1;
}

int main(void) {
  if (pg_abi_version() != PG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  int32_t n_dev = 0;
  if (check(pg_device_count(&n_dev), "pg_device_count")) return 1;
  if (n_dev <= 0) {
    int32_t st = pg_init(0);
    char msg[512];
    pg_last_error(msg, sizeof msg);
    printf("no HIP device: pg_init -> %d (%s); ABI v%d ok\n", st, msg, pg_abi_version());
    return st == PG_ERR_DEVICE ? 0 : 1;
  }
  if (check(pg_init(0), "pg_init")) return 1;

  /* column d: dictionary {10, 20, 30, 40} (INT, sorted, big-endian), dictId = doc % 4, 2 bits per value MSB first */
  uint8_t dict[16];
  for (int i = 0; i < 4; i++) put_be32(dict + 4 * i, (uint32_t)(10 * (i + 1)));
  uint8_t fwd[(N_DOCS * 2 + 7) / 8];
  memset(fwd, 0, sizeof fwd);
  for (int doc = 0; doc < N_DOCS; doc++) {
    int id = doc % 4, bit = doc * 2;
    fwd[bit >> 3] |= (uint8_t)(id << (6 - (bit & 7)));
  }
  /* inverted index: 5 BE offsets, then 4 portable RoaringBitmap blobs (cookie 12346, one array container each) */
  uint8_t inv[20 + 4 * (16 + 2 * (N_DOCS / 4))];
  size_t pos = 20;
  for (int id = 0; id < 4; id++) {
    put_be32(inv + 4 * id, (uint32_t)pos);
    uint8_t* b = inv + pos;
    put_le32(b, 12346); put_le32(b + 4, 1);                       /* cookie, one container */
    put_le16(b + 8, 0); put_le16(b + 10, (uint16_t)(N_DOCS / 4 - 1)); /* key 0, cardinality - 1 */
    put_le32(b + 12, 16);                                          /* offset of the container payload */
    for (int k = 0; k < N_DOCS / 4; k++) put_le16(b + 16 + 2 * k, (uint16_t)(4 * k + id));
    pos += 16 + 2 * (size_t)(N_DOCS / 4);
  }
  put_be32(inv + 16, (uint32_t)pos);
  /* column m: raw INT, FixedByteChunk v2 header (7 BE ints) + 1 chunk offset + big-endian values, m = doc */
  uint8_t raw[32 + 4 * N_DOCS];
  const uint32_t hdr[8] = {2, 1, N_DOCS, 4, N_DOCS, 0 /* PASS_THROUGH */, 28, 32};
  for (int i = 0; i < 8; i++) put_be32(raw + 4 * i, hdr[i]);
  for (int doc = 0; doc < N_DOCS; doc++) put_be32(raw + 32 + 4 * doc, (uint32_t)doc);

  pg_segment_t seg = NULL;
  if (check(pg_segment_create("abi_smoke", N_DOCS, &seg), "pg_segment_create")) return 1;
  pg_column_desc d;
  memset(&d, 0, sizeof d);
  d.name = "d"; d.data_type = PG_TYPE_INT; d.fwd_encoding = PG_FWD_DICT_FIXED_BIT; d.has_dictionary = 1; d.cardinality = 4;
  d.bits_per_value = 2; d.dict_bytes_per_value = 4;
  d.forward_index.addr = fwd; d.forward_index.size = sizeof fwd;
  d.dictionary.addr = dict; d.dictionary.size = sizeof dict;
  d.inverted_index.addr = inv; d.inverted_index.size = pos;
  if (check(pg_segment_add_column(seg, &d), "pg_segment_add_column(d)")) return 1;
  pg_column_desc m;
  memset(&m, 0, sizeof m);
  m.name = "m"; m.data_type = PG_TYPE_INT; m.fwd_encoding = PG_FWD_RAW_FIXED_BYTE_CHUNK;
  m.forward_index.addr = raw; m.forward_index.size = sizeof raw;
  if (check(pg_segment_add_column(seg, &m), "pg_segment_add_column(m)")) return 1;

  const char* in_values[2] = {"20", "30"};
  pg_filter_node pred;
  memset(&pred, 0, sizeof pred);
  pred.type = PG_FILTER_PREDICATE; pred.predicate_type = PG_PRED_IN; pred.column = "d"; pred.n_values = 2; pred.values = in_values;
  const char* group_by[1] = {"d"};
  pg_agg_spec aggs[2];
  memset(aggs, 0, sizeof aggs);
  aggs[0].function = PG_AGG_COUNT; aggs[0].column = "*";
  aggs[1].function = PG_AGG_SUM; aggs[1].column = "m";
  pg_query q;
  memset(&q, 0, sizeof q);
  q.filter = &pred; q.n_group_by = 1; q.group_by_columns = group_by; q.n_aggregations = 2; q.aggregations = aggs;
  pg_result_t res = NULL;
  if (check(pg_query_exec(seg, &q, &res), "pg_query_exec")) return 1;
  int32_t ng = 0;
  pg_result_num_groups(res, &ng);
  int32_t ids[4]; int64_t counts[4]; double sums[4];
  pg_result_group_dict_ids(res, 0, ids, 4);
  pg_result_longs(res, 0, 0, counts, 4);
  pg_result_doubles(res, 1, 0, sums, 4);
  pg_exec_stats st;
  pg_result_stats(res, &st);
  int ok = ng == 2;
  for (int g = 0; g < ng; g++) {
    /* docs with dictId id: id, id + 4, ...: 250 docs, sum = 250 * id + 4 * (0 + ... + 249) */
    const double expect = 250.0 * ids[g] + 4.0 * (249.0 * 250.0 / 2.0);
    printf("d=%d count=%lld sum=%.0f\n", 10 * (ids[g] + 1), (long long)counts[g], sums[g]);
    ok = ok && counts[g] == 250 && sums[g] == expect;
  }
  printf("docs scanned %lld, entries scanned in filter %lld, kernel %s\n", (long long)st.num_docs_scanned,
         (long long)st.num_entries_scanned_in_filter, st.kernel);
  ok = ok && st.num_docs_scanned == 500 && st.num_entries_scanned_in_filter == 0;
  pg_result_free(res);
  pg_segment_destroy(seg);
  printf(ok ? "abi smoke ok\n" : "abi smoke FAILED\n");
  return ok ? 0 : 1;
}
