/* See pinot_gpu_shim.h.  One arena per query: nodes, pointer arrays and NUL-terminated copies of the strings. */
#include "pinot_gpu_shim.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct pgshim_query {
  pg_query q;
  uint8_t* arena;
  size_t arena_size, arena_used;
};

typedef struct {
  const uint8_t* p;
  uint64_t size, pos;
  pgshim_query* out;
  char* err;
  size_t err_cap;
  int failed;
} reader;

static void fail(reader* r, const char* msg) {
  if (!r->failed && r->err && r->err_cap) snprintf(r->err, r->err_cap, "NativeQuery record: %s at byte %llu", msg, (unsigned long long)r->pos);
  r->failed = 1;
}
static int32_t rd_i32(reader* r) {
  if (r->failed || r->pos + 4 > r->size) { fail(r, "truncated"); return 0; }
  const uint8_t* b = r->p + r->pos;
  r->pos += 4;
  return (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
}
static void* arena_alloc(reader* r, size_t n) {
  pgshim_query* q = r->out;
  n = (n + 7) & ~(size_t)7;
  if (q->arena_used + n > q->arena_size) { fail(r, "arena exhausted"); return NULL; }
  void* p = q->arena + q->arena_used;
  q->arena_used += n;
  memset(p, 0, n);
  return p;
}
static const char* rd_string(reader* r) {
  const int32_t len = rd_i32(r);
  if (r->failed || len < 0) return NULL;
  const uint64_t padded = ((uint64_t)len + 3) & ~(uint64_t)3;
  if (r->pos + padded > r->size) { fail(r, "truncated string"); return NULL; }
  char* s = (char*)arena_alloc(r, (size_t)len + 1);
  if (!s) return NULL;
  memcpy(s, r->p + r->pos, (size_t)len);
  r->pos += padded;
  return s;
}
static void rd_node(reader* r, pg_filter_node* node, int depth) {
  if (depth > 64) { fail(r, "filter tree deeper than 64"); return; }
  node->type = rd_i32(r);
  node->n_children = rd_i32(r);
  if (r->failed) return;
  if (node->type < PG_FILTER_AND || node->type > PG_FILTER_CONSTANT_FALSE || node->n_children < 0 || node->n_children > 4096) { fail(r, "bad filter node"); return; }
  if (node->type == PG_FILTER_PREDICATE) {
    node->predicate_type = rd_i32(r);
    node->n_values = rd_i32(r);
    if (r->failed || node->n_values < 0 || node->n_values > (1 << 20)) { fail(r, "bad predicate"); return; }
    node->column = rd_string(r);
    if (node->n_values) {
      const char** values = (const char**)arena_alloc(r, sizeof(char*) * (size_t)node->n_values);
      if (!values) return;
      for (int32_t i = 0; i < node->n_values && !r->failed; i++) values[i] = rd_string(r);
      node->values = values;
    }
    node->lower = rd_string(r);
    node->upper = rd_string(r);
    node->lower_inclusive = rd_i32(r);
    node->upper_inclusive = rd_i32(r);
  }
  if (node->n_children) {
    pg_filter_node* kids = (pg_filter_node*)arena_alloc(r, sizeof(pg_filter_node) * (size_t)node->n_children);
    if (!kids) return;
    for (int32_t i = 0; i < node->n_children && !r->failed; i++) rd_node(r, &kids[i], depth + 1);
    node->children = kids;
  }
}

int32_t pgshim_query_parse(const void* record, uint64_t size, pgshim_query** out_query, char* err, size_t err_cap) {
  if (err && err_cap) err[0] = 0;
  if (!record || !out_query || size < 40) {
    if (err && err_cap) snprintf(err, err_cap, "NativeQuery record: null or shorter than its header");
    return PG_ERR_INVALID_ARGUMENT;
  }
  pgshim_query* q = (pgshim_query*)calloc(1, sizeof(*q));
  if (!q) return PG_ERR_OUT_OF_MEMORY;
  /* every int of the record becomes at most one 64-byte node / pointer slot, every byte at most one string byte + terminator */
  q->arena_size = (size_t)size * 20 + 1024;
  q->arena = (uint8_t*)malloc(q->arena_size);
  if (!q->arena) { free(q); return PG_ERR_OUT_OF_MEMORY; }
  reader r = {(const uint8_t*)record, size, 0, q, err, err_cap, 0};
  if (rd_i32(&r) != PGSHIM_QUERY_MAGIC) fail(&r, "bad magic");
  q->q.flags = rd_i32(&r);
  q->q.num_groups_limit = rd_i32(&r);
  q->q.max_initial_result_holder_capacity = rd_i32(&r);
  q->q.n_group_by = rd_i32(&r);
  q->q.n_aggregations = rd_i32(&r);
  const int32_t has_filter = rd_i32(&r);
  q->q.n_order_by = rd_i32(&r);
  q->q.limit = rd_i32(&r);
  q->q.min_segment_group_trim_size = rd_i32(&r);
  if (!r.failed && (q->q.n_group_by < 0 || q->q.n_group_by > 64 || q->q.n_aggregations < 0 || q->q.n_aggregations > 256 || q->q.n_order_by < 0 || q->q.n_order_by > 64))
    fail(&r, "bad counts");
  if (!r.failed && q->q.n_group_by) {
    const char** g = (const char**)arena_alloc(&r, sizeof(char*) * (size_t)q->q.n_group_by);
    for (int32_t i = 0; g && i < q->q.n_group_by && !r.failed; i++) g[i] = rd_string(&r);
    q->q.group_by_columns = g;
  }
  if (!r.failed && q->q.n_aggregations) {
    pg_agg_spec* a = (pg_agg_spec*)arena_alloc(&r, sizeof(pg_agg_spec) * (size_t)q->q.n_aggregations);
    for (int32_t i = 0; a && i < q->q.n_aggregations && !r.failed; i++) {
      a[i].function = rd_i32(&r);
      a[i].log2m = rd_i32(&r);
      a[i].column = rd_string(&r);
    }
    q->q.aggregations = a;
  }
  if (!r.failed && q->q.n_order_by) {
    pg_order_by* ob = (pg_order_by*)arena_alloc(&r, sizeof(pg_order_by) * (size_t)q->q.n_order_by);
    for (int32_t i = 0; ob && i < q->q.n_order_by && !r.failed; i++) {
      ob[i].kind = rd_i32(&r);
      ob[i].index = rd_i32(&r);
      ob[i].ascending = rd_i32(&r);
      ob[i].nulls_last = rd_i32(&r);
      if (!r.failed && (ob[i].kind < PG_ORDER_BY_GROUP_KEY || ob[i].kind > PG_ORDER_BY_AGGREGATION || ob[i].index < 0 ||
                        ob[i].index >= (ob[i].kind == PG_ORDER_BY_GROUP_KEY ? q->q.n_group_by : q->q.n_aggregations)))
        fail(&r, "bad order-by expression");
    }
    q->q.order_by = ob;
  }
  if (!r.failed && has_filter) {
    pg_filter_node* root = (pg_filter_node*)arena_alloc(&r, sizeof(pg_filter_node));
    if (root) rd_node(&r, root, 0);
    q->q.filter = root;
  }
  if (!r.failed && r.pos != size) fail(&r, "trailing bytes");
  if (r.failed) { pgshim_query_free(q); return PG_ERR_INVALID_ARGUMENT; }
  *out_query = q;
  return PG_OK;
}
const pg_query* pgshim_query_get(const pgshim_query* q) { return q ? &q->q : NULL; }
void pgshim_query_free(pgshim_query* q) {
  if (!q) return;
  free(q->arena);
  free(q);
}
