/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 * DocIdSetOperator → ProjectionOperator → GroupByOperator / AggregationOperator and the exported po_* entry points
 * (same shapes as include/pinot_gpu.h so the parity tests drive both sides with one set of structs).
 * SURVEY.md §8a rows a10, a11, a15–a23.
 */
#define _GNU_SOURCE   /* qsort_r */
#include <stdlib.h>
#include <math.h>
#include <stdio.h>
#include <time.h>

#include "po_internal.h"

const char* po_get_error(void);
int po_raw_parse_header(po_column* c);

#define DEFAULT_NUM_GROUPS_LIMIT 100000
#define DEFAULT_MAX_INITIAL_RESULT_HOLDER_CAPACITY 10000
#define DEFAULT_LOG2M 8

/* =====================================================================================================================
 * segment
 * ===================================================================================================================== */
po_column* po_segment_column(po_segment* seg, const char* name) {
  if (!name) return NULL;
  for (int i = 0; i < seg->n_columns; i++)
    if (strcmp(seg->columns[i]->name, name) == 0) return seg->columns[i];
  return NULL;
}

int32_t po_abi_version(void) { return PG_ABI_VERSION; }
int32_t po_init(int32_t d) { (void)d; return PG_OK; }
int32_t po_device_count(int32_t* n) { *n = 0; return PG_OK; }
int32_t po_last_error(char* buf, size_t cap) {
  const char* e = po_get_error();
  size_t n = strlen(e);
  if (cap) {
    size_t k = n < cap - 1 ? n : cap - 1;
    memcpy(buf, e, k);
    buf[k] = 0;
  }
  return (int32_t)n;
}

int32_t po_segment_create(const char* name, int32_t total_docs, void** out) {
  po_segment* s = (po_segment*)po_xcalloc(1, sizeof(*s));
  s->name = strdup(name ? name : "");
  s->total_docs = total_docs;
  *out = s;
  return PG_OK;
}

/* The oracle keeps pointers into the caller's buffers (they must outlive the segment). */
int32_t po_segment_add_column(void* segp, const pg_column_desc* d) {
  po_segment* seg = (po_segment*)segp;
  po_column* c = (po_column*)po_xcalloc(1, sizeof(*c));
  c->name = strdup(d->name);
  c->data_type = d->data_type;
  c->fwd_encoding = d->fwd_encoding;
  c->has_dictionary = d->has_dictionary;
  c->cardinality = d->cardinality;
  c->bits_per_value = d->bits_per_value;
  c->is_sorted = d->is_sorted;
  c->dict_bytes_per_value = d->dict_bytes_per_value;
  c->fwd = (const uint8_t*)d->forward_index.addr;
  c->fwd_len = d->forward_index.size;
  c->dict = (const uint8_t*)d->dictionary.addr;
  c->dict_len = d->dictionary.size;
  c->inv = (const uint8_t*)d->inverted_index.addr;
  c->inv_len = d->inverted_index.size;
  c->num_docs = seg->total_docs;
  /* a variable-length STRING dictionary (VarLengthValueReader: ".vl;", int version 1, int numValues, int dataSectionStartOffset, numValues + 1
   * absolute int offsets, the values; recognised by its magic like BaseImmutableDictionary.java:58-66 does) is read into the fixed-width,
   * zero-padded form the readers below index */
  if (c->has_dictionary && c->data_type == PG_TYPE_STRING && c->dict && c->dict_len >= 20 && !memcmp(c->dict, ".vl;", 4) && po_be32(c->dict + 4) == 1) {
    const uint32_t n = po_be32(c->dict + 8), start = po_be32(c->dict + 12);
    if ((int64_t)n != (int64_t)c->cardinality || (uint64_t)start + ((uint64_t)n + 1) * 4 > c->dict_len) {
      po_set_error("variable-length dictionary of %s: %u values, cardinality %d", c->name, n, c->cardinality);
      return PG_ERR_INVALID_ARGUMENT;
    }
    uint32_t width = 1;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t a = po_be32(c->dict + start + (uint64_t)i * 4), b = po_be32(c->dict + start + (uint64_t)(i + 1) * 4);
      if (b < a || b > c->dict_len) { po_set_error("variable-length dictionary of %s: bad offsets", c->name); return PG_ERR_INVALID_ARGUMENT; }
      if (b - a > width) width = b - a;
    }
    uint8_t* padded = (uint8_t*)po_xcalloc((size_t)n * width + 8, 1);
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t a = po_be32(c->dict + start + (uint64_t)i * 4), b = po_be32(c->dict + start + (uint64_t)(i + 1) * 4);
      memcpy(padded + (size_t)i * width, c->dict + a, b - a);
    }
    c->mv_owned_dict = padded;
    c->dict = padded;
    c->dict_len = (uint64_t)n * width;
    c->dict_bytes_per_value = (int32_t)width;
  }
  if ((c->fwd_encoding == PG_FWD_RAW_FIXED_BYTE_CHUNK || c->fwd_encoding == PG_FWD_RAW_VAR_BYTE_CHUNK) && po_raw_parse_header(c)) return PG_ERR_UNSUPPORTED;
  if (c->fwd_encoding == PG_FWD_RAW_MV_FIXED_BYTE_CHUNK || c->fwd_encoding == PG_FWD_RAW_MV_VAR_BYTE_CHUNK) {   /* FixedByteChunkMV / VarByteChunkMV readers: po_readers.c */
    if (c->fwd_encoding == PG_FWD_RAW_MV_VAR_BYTE_CHUNK ? po_raw_mv_attach_strings(c) : po_raw_mv_attach(c)) return PG_ERR_INVALID_ARGUMENT;
    if (d->total_number_of_entries > 0 && d->total_number_of_entries != c->total_entries) {
      po_set_error("raw multi-value index of %s holds %d entries, the metadata says %d", c->name, c->total_entries, d->total_number_of_entries);
      return PG_ERR_INVALID_ARGUMENT;
    }
    seg->columns = (po_column**)po_xrealloc(seg->columns, sizeof(po_column*) * (size_t)(seg->n_columns + 1));
    seg->columns[seg->n_columns++] = c;
    return PG_OK;
  }
  if (c->fwd_encoding == PG_FWD_DICT_FIXED_BIT_MV) {
    c->total_entries = d->total_number_of_entries;
    if (!c->has_dictionary) { po_set_error("raw multi-value column %s is outside the hot path", c->name); return PG_ERR_UNSUPPORTED; }
    if (c->fwd_len > 4 && po_be32(c->fwd) == 0xffabcdefu) {   /* ForwardIndexReaderFactory.java:82-86 checks this marker first */
      if (po_mv_entry_dict_attach(c)) return PG_ERR_INVALID_ARGUMENT;
      if (d->total_number_of_entries > 0 && d->total_number_of_entries != c->total_entries) {
        po_set_error("MV_ENTRY_DICT forward index of %s expands to %d entries, the metadata says %d", c->name, c->total_entries, d->total_number_of_entries);
        return PG_ERR_INVALID_ARGUMENT;
      }
    } else if (po_mv_parse(c)) return PG_ERR_INVALID_ARGUMENT;
  }
  if (c->fwd_encoding == PG_FWD_DICT_FIXED_BIT) {
    uint64_t need = ((uint64_t)seg->total_docs * (uint64_t)c->bits_per_value + 7) / 8;
    if (c->fwd_len < need) {
      po_set_error("forward index of %s is %llu bytes, need %llu", c->name, (unsigned long long)c->fwd_len,
                   (unsigned long long)need);
      return PG_ERR_INVALID_ARGUMENT;
    }
  }
  seg->columns = (po_column**)po_xrealloc(seg->columns, sizeof(po_column*) * (size_t)(seg->n_columns + 1));
  seg->columns[seg->n_columns++] = c;
  return PG_OK;
}
static po_bitmap* roaring_to_bitmap(const void* roaring, uint64_t size, int32_t total_docs) {
  po_bitmap* b = po_bitmap_new(total_docs);
  if (po_roaring_deserialize_or((const uint8_t*)roaring, size, b)) { po_bitmap_free(b); return NULL; }
  return b;
}
/* NullValueVectorReaderImpl: the null bitmap of one column */
int32_t po_segment_set_null_vector(void* segp, const char* column, const void* roaring, uint64_t size) {
  po_segment* seg = (po_segment*)segp;
  po_column* c = po_segment_column(seg, column);
  if (!c) { po_set_error("column not found: %s", column ? column : "(null)"); return PG_ERR_NOT_FOUND; }
  po_bitmap* b = roaring_to_bitmap(roaring, size, seg->total_docs);
  if (!b) return PG_ERR_INVALID_ARGUMENT;
  if (c->null_bitmap) po_bitmap_free(c->null_bitmap);
  c->null_bitmap = b;
  return PG_OK;
}
/* DataSource#getRangeIndex: the column's `range_index` entry (BitSlicedRangeIndexReader); the bytes are borrowed, like the other indexes */
int32_t po_segment_set_range_index(void* segp, const char* column, const void* bytes, uint64_t size) {
  po_segment* seg = (po_segment*)segp;
  po_column* c = po_segment_column(seg, column);
  if (!c) { po_set_error("column not found: %s", column ? column : "(null)"); return PG_ERR_NOT_FOUND; }
  c->range_idx = size ? (const uint8_t*)bytes : NULL;
  c->range_len = size;
  return PG_OK;
}
/* SegmentContext#getQueryableDocIdsSnapshot */
int32_t po_segment_set_queryable_doc_ids(void* segp, const void* roaring, uint64_t size) {
  po_segment* seg = (po_segment*)segp;
  po_bitmap* b = NULL;
  if (size) {
    b = roaring_to_bitmap(roaring, size, seg->total_docs);
    if (!b) return PG_ERR_INVALID_ARGUMENT;
  }
  if (seg->queryable_doc_ids) po_bitmap_free(seg->queryable_doc_ids);
  seg->queryable_doc_ids = b;
  return PG_OK;
}
int32_t po_segment_num_docs(void* s, int32_t* out) { *out = ((po_segment*)s)->total_docs; return PG_OK; }
int32_t po_segment_device_bytes(void* s, uint64_t* out) { (void)s; *out = 0; return PG_OK; }
int32_t po_segment_destroy(void* segp) {
  po_segment* seg = (po_segment*)segp;
  for (int i = 0; i < seg->n_columns; i++) {
    free(seg->columns[i]->name);
    if (seg->columns[i]->null_bitmap) po_bitmap_free(seg->columns[i]->null_bitmap);
    free(seg->columns[i]->raw_owned);
    free(seg->columns[i]->mv_owned_fwd);
    free(seg->columns[i]->mv_owned_dict);
    free(seg->columns[i]);
  }
  if (seg->queryable_doc_ids) po_bitmap_free(seg->queryable_doc_ids);
  free(seg->columns);
  free(seg->name);
  free(seg);
  return PG_OK;
}

/* =====================================================================================================================
 * filter-only API (FilterOperator + DocIdSetOperator)
 * ===================================================================================================================== */
typedef struct po_docidset_result {
  po_bitmap* bitmap;
  int64_t cardinality;
  int32_t num_docs;
  pg_exec_stats stats;
} po_docidset_result;

int32_t po_filter_exec_flags(void* segp, const pg_filter_node* filter, int32_t flags, void** out) {
  po_segment* seg = (po_segment*)segp;
  po_filter_op* op = po_filter_plan(seg, filter, (flags & PG_QUERY_FLAG_NULL_HANDLING) != 0);
  if (!op) return PG_ERR_INVALID_ARGUMENT;
  po_docidset* set = po_filter_get_trues(op);
  if (!set) return PG_ERR_INVALID_ARGUMENT;
  po_iter* it = set->iterator(set);
  po_docidset_result* r = (po_docidset_result*)po_xcalloc(1, sizeof(*r));
  r->bitmap = po_bitmap_new(seg->total_docs);
  r->num_docs = seg->total_docs;
  int32_t d;
  while ((d = it->next(it)) != PO_EOF) {
    po_bitmap_add(r->bitmap, d);
    r->cardinality++;
  }
  r->stats.num_docs_scanned = r->cardinality;
  r->stats.num_entries_scanned_in_filter = set->num_entries_scanned(set);
  r->stats.num_total_docs = seg->total_docs;
  r->stats.stats_exact = 1;
  r->stats.star_tree_index = -1;
  *out = r;
  return PG_OK;
}
int32_t po_filter_exec(void* segp, const pg_filter_node* filter, void** out) { return po_filter_exec_flags(segp, filter, 0, out); }
int32_t po_docidset_cardinality(void* s, int64_t* out) { *out = ((po_docidset_result*)s)->cardinality; return PG_OK; }
int32_t po_docidset_num_words(void* s, int64_t* out) {
  *out = (((po_docidset_result*)s)->num_docs + 63) / 64;
  return PG_OK;
}
int32_t po_docidset_copy_words(void* s, uint64_t* out, int64_t cap) {
  po_docidset_result* r = (po_docidset_result*)s;
  int64_t n = (r->num_docs + 63) / 64;
  if (cap < n) { po_set_error("capacity too small"); return PG_ERR_INVALID_ARGUMENT; }
  memcpy(out, r->bitmap->words, (size_t)n * 8);
  return PG_OK;
}
int32_t po_docidset_copy_docids(void* s, int32_t* out, int64_t cap) {
  po_docidset_result* r = (po_docidset_result*)s;
  if (cap < r->cardinality) { po_set_error("capacity too small"); return PG_ERR_INVALID_ARGUMENT; }
  int64_t k = 0;
  for (int64_t p = po_bitmap_next_set(r->bitmap, 0); p >= 0; p = po_bitmap_next_set(r->bitmap, p + 1)) out[k++] = (int32_t)p;
  return PG_OK;
}
int32_t po_docidset_stats(void* s, pg_exec_stats* out) { *out = ((po_docidset_result*)s)->stats; return PG_OK; }
int32_t po_docidset_free(void* s) {
  po_docidset_result* r = (po_docidset_result*)s;
  po_bitmap_free(r->bitmap);
  free(r);
  return PG_OK;
}

/* =====================================================================================================================
 * group key generation: DictionaryBasedGroupKeyGenerator
 * ===================================================================================================================== */
static inline uint32_t hash_common_mix(uint32_t x) { /* fastutil HashCommon.mix */
  uint32_t h = x * 0x9E3779B9u;
  return h ^ (h >> 16);
}

/* IntGroupIdMap, core/query/aggregation/groupby/DictionaryBasedGroupKeyGenerator.java:993-1084 */
typedef struct int_group_id_map { int32_t* kv; int32_t capacity, mask, max_entries, size; } int_group_id_map;
static void igm_init(int_group_id_map* m) {
  m->capacity = 1 << 9;
  int holder = m->capacity << 1;
  m->kv = (int32_t*)po_xcalloc((size_t)holder, 4);
  m->mask = holder - 1;
  m->max_entries = (int)(m->capacity * 0.75f);
  m->size = 0;
}
static void igm_expand(int_group_id_map* m) {
  m->capacity <<= 1;
  int holder = m->capacity << 1;
  int32_t* old = m->kv;
  m->kv = (int32_t*)po_xcalloc((size_t)holder, 4);
  m->mask = holder - 1;
  m->max_entries <<= 1;
  int old_index = 0;
  for (int i = 0; i < m->size; i++) {
    while (old[old_index] == 0) old_index += 2;
    int32_t key = old[old_index], value = old[old_index + 1];
    int ni = (int)((hash_common_mix((uint32_t)key) << 1) & (uint32_t)m->mask);
    while (m->kv[ni] != 0) ni = (ni + 2) & m->mask;
    m->kv[ni] = key;
    m->kv[ni + 1] = value;
    old_index += 2;
  }
  free(old);
}
static int32_t igm_get_group_id(int_group_id_map* m, int32_t raw_key, int32_t upper_bound) {
  int32_t internal = raw_key + 1;
  int index = (int)((hash_common_mix((uint32_t)internal) << 1) & (uint32_t)m->mask);
  while (1) {
    int32_t key = m->kv[index];
    if (key == internal) return m->kv[index + 1];
    if (key == 0) {
      if (m->size < upper_bound) {
        int32_t gid = m->size++;
        m->kv[index] = internal;
        m->kv[index + 1] = gid;
        if (m->size > m->max_entries) igm_expand(m);
        return gid;
      }
      return PO_INVALID_ID;
    }
    index = (index + 2) & m->mask;
  }
}

/* Long2IntOpenHashMap stand-in (fastutil; only map semantics matter: putIfAbsent(rawKey, numGroups)) */
typedef struct long_map { int64_t* keys; int32_t* vals; uint8_t* used; int64_t cap; int32_t size; } long_map;
static void lm_init(long_map* m) {
  m->cap = 1024;
  m->keys = (int64_t*)po_xcalloc((size_t)m->cap, 8);
  m->vals = (int32_t*)po_xcalloc((size_t)m->cap, 4);
  m->used = (uint8_t*)po_xcalloc((size_t)m->cap, 1);
  m->size = 0;
}
static uint64_t lm_hash(int64_t k) {
  uint64_t h = (uint64_t)k * 0x9E3779B97F4A7C15ULL;
  return h ^ (h >> 32);
}
static void lm_grow(long_map* m) {
  long_map n;
  n.cap = m->cap * 2;
  n.keys = (int64_t*)po_xcalloc((size_t)n.cap, 8);
  n.vals = (int32_t*)po_xcalloc((size_t)n.cap, 4);
  n.used = (uint8_t*)po_xcalloc((size_t)n.cap, 1);
  n.size = m->size;
  for (int64_t i = 0; i < m->cap; i++) {
    if (!m->used[i]) continue;
    uint64_t p = lm_hash(m->keys[i]) & (uint64_t)(n.cap - 1);
    while (n.used[p]) p = (p + 1) & (uint64_t)(n.cap - 1);
    n.used[p] = 1;
    n.keys[p] = m->keys[i];
    n.vals[p] = m->vals[i];
  }
  free(m->keys); free(m->vals); free(m->used);
  *m = n;
}
static int32_t lm_get_group_id(long_map* m, int64_t raw_key, int32_t upper_bound) { /* LongMapBasedHolder#getGroupId :652-660 */
  uint64_t p = lm_hash(raw_key) & (uint64_t)(m->cap - 1);
  while (m->used[p]) {
    if (m->keys[p] == raw_key) return m->vals[p];
    p = (p + 1) & (uint64_t)(m->cap - 1);
  }
  if (m->size < upper_bound) {
    m->used[p] = 1;
    m->keys[p] = raw_key;
    m->vals[p] = m->size;
    int32_t id = m->size++;
    if ((int64_t)m->size * 4 > m->cap * 3) lm_grow(m);
    return id;
  }
  return PO_INVALID_ID;
}

/* BaseOnTheFlyDictionary for STRING / BYTES (NoDictionaryMultiColumnGroupKeyGenerator's per-column value -> id maps,
 * Object2IntOpenHashMap in NoDictionarySingleColumnGroupKeyGenerator.java:132-140): ids in first-seen order; values borrowed from the
 * forward index buffer */
typedef struct bytes_dict { const uint8_t** vals; int32_t* lens; int32_t n, cap; int32_t* table; int32_t table_cap; } bytes_dict;
static uint64_t bytes_hash(const uint8_t* p, int32_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (int32_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ULL; }
  return h ^ (h >> 29);
}
static int32_t bytes_dict_index(bytes_dict* d, const uint8_t* p, int32_t n) {
  if (!d->table) {
    d->table_cap = 1024;
    d->table = (int32_t*)po_xcalloc((size_t)d->table_cap, 4);
    d->cap = 256;
    d->vals = (const uint8_t**)po_xmalloc(sizeof(void*) * (size_t)d->cap);
    d->lens = (int32_t*)po_xmalloc(4 * (size_t)d->cap);
  }
  uint64_t pos = bytes_hash(p, n) & (uint64_t)(d->table_cap - 1);
  while (d->table[pos]) {
    const int32_t id = d->table[pos] - 1;
    if (d->lens[id] == n && memcmp(d->vals[id], p, (size_t)n) == 0) return id;
    pos = (pos + 1) & (uint64_t)(d->table_cap - 1);
  }
  if (d->n == d->cap) {
    d->cap *= 2;
    d->vals = (const uint8_t**)po_xrealloc(d->vals, sizeof(void*) * (size_t)d->cap);
    d->lens = (int32_t*)po_xrealloc(d->lens, 4 * (size_t)d->cap);
  }
  const int32_t id = d->n++;
  d->vals[id] = p;
  d->lens[id] = n;
  d->table[pos] = id + 1;
  if ((int64_t)d->n * 2 > d->table_cap) {   /* rehash */
    free(d->table);
    d->table_cap *= 4;
    d->table = (int32_t*)po_xcalloc((size_t)d->table_cap, 4);
    for (int32_t k = 0; k < d->n; k++) {
      uint64_t q = bytes_hash(d->vals[k], d->lens[k]) & (uint64_t)(d->table_cap - 1);
      while (d->table[q]) q = (q + 1) & (uint64_t)(d->table_cap - 1);
      d->table[q] = k + 1;
    }
  }
  return id;
}

enum { HOLDER_ARRAY, HOLDER_INT_MAP, HOLDER_LONG_MAP,
       HOLDER_RAW_VALUES /* NoDictionarySingleColumnGroupKeyGenerator: value -> group id (Int2IntOpenHashMap / Long2IntOpenHashMap) */,
       HOLDER_TUPLES /* a raw FLOAT / DOUBLE column (Float2Int / Double2IntOpenHashMap), or NoDictionaryMultiColumnGroupKeyGenerator: per
                        column a dictId or the value's key (int value, floatToIntBits, doubleToLongBits), the tuple -> group id */ };

typedef struct group_key_gen {
  int n_cols;
  po_column** cols;
  int32_t* cardinalities;
  int holder;
  int32_t global_upper_bound;   /* _globalGroupIdUpperBound */
  /* array based */
  uint8_t* flags; int32_t num_keys;
  int_group_id_map imap;
  long_map lmap;
  /* reverse: raw key per group id for map holders */
  int64_t* raw_key_of_group; int32_t raw_cap;
  /* HOLDER_TUPLES: n_cols keys per group id, and an open-addressing table of group ids + 1 */
  int64_t* tuples; int32_t tuple_cap, n_tuples;
  int32_t* tuple_table; int32_t tuple_table_cap;
  bytes_dict* bytes_dicts;      /* HOLDER_TUPLES: per raw STRING / BYTES column its on-the-fly dictionary (the tuple holds the id) */
  int tuple_w;                  /* int64 slots per tuple: n_cols, or n_cols + 1 under null handling (the last slot = bit mask of the null raw columns) */
  const po_bitmap** raw_nulls;  /* null handling: per no-dictionary column its null bitmap (NULL: none) */
} group_key_gen;

/* constructor, DictionaryBasedGroupKeyGenerator.java:106-185 */
static int gkg_init(group_key_gen* g, int n_cols, po_column** cols, int32_t num_groups_limit, int32_t array_threshold, const po_bitmap** raw_nulls) {
  memset(g, 0, sizeof(*g));
  g->n_cols = n_cols;
  g->cols = cols;
  g->tuple_w = n_cols;
  int any_raw = 0, any_raw_null = 0;
  for (int i = 0; i < n_cols; i++) any_raw |= !cols[i]->has_dictionary;
  for (int i = 0; i < n_cols && raw_nulls; i++) any_raw_null |= !cols[i]->has_dictionary && raw_nulls[i] != NULL;
  if (any_raw_null) { g->raw_nulls = raw_nulls; g->tuple_w = n_cols + 1; }   /* a null raw value: key 0 + its bit in the mask slot */
  if (any_raw && (any_raw_null || !(n_cols == 1 && (cols[0]->data_type == PG_TYPE_INT || cols[0]->data_type == PG_TYPE_LONG)))) {
    /* DefaultGroupByExecutor.java:100-118: any group-by expression without a dictionary → the no-dictionary generators; both admit
     * new keys in docId order until numGroupsLimit (NoDictionaryMultiColumnGroupKeyGenerator.java:60-130,
     * NoDictionarySingleColumnGroupKeyGenerator.java:238-262) */
    g->holder = HOLDER_TUPLES;
    g->global_upper_bound = num_groups_limit;
    g->cardinalities = (int32_t*)po_xcalloc((size_t)n_cols + 1, 4);
    g->tuple_cap = 1024;
    g->tuples = (int64_t*)po_xmalloc(sizeof(int64_t) * (size_t)g->tuple_cap * (size_t)g->tuple_w);
    g->tuple_table_cap = 4096;
    g->tuple_table = (int32_t*)po_xcalloc((size_t)g->tuple_table_cap, 4);
    g->bytes_dicts = (bytes_dict*)po_xcalloc((size_t)n_cols + 1, sizeof(bytes_dict));
    return 0;
  }
  if (n_cols == 1 && !cols[0]->has_dictionary) {   /* NoDictionarySingleColumnGroupKeyGenerator ctor :69-84 */
    g->holder = HOLDER_RAW_VALUES;
    g->global_upper_bound = num_groups_limit;
    g->cardinalities = (int32_t*)po_xcalloc(2, 4);
    lm_init(&g->lmap);
    g->raw_cap = 1024;
    g->raw_key_of_group = (int64_t*)po_xmalloc(sizeof(int64_t) * (size_t)g->raw_cap);
    return 0;
  }
  g->cardinalities = (int32_t*)po_xcalloc((size_t)n_cols + 1, 4);
  int64_t product = 1;
  int long_overflow = 0;
  for (int i = 0; i < n_cols; i++) {
    int32_t card = cols[i]->cardinality;
    g->cardinalities[i] = card;
    if (!long_overflow) {
      if (card > 0 && product > INT64_MAX / card) long_overflow = 1;   /* card 0: an empty segment (pruned before planning in the reference) */
      else product *= card;
    }
  }
  if (long_overflow) {
    po_set_error("ArrayMapBasedHolder (cardinality product > 2^63) is outside the hot path");
    return -1;
  }
  if (product > INT32_MAX) {
    g->holder = HOLDER_LONG_MAP;
    g->global_upper_bound = num_groups_limit;
    lm_init(&g->lmap);
  } else {
    g->global_upper_bound = (int32_t)(product < num_groups_limit ? product : num_groups_limit);
    if (product > array_threshold || num_groups_limit < product) {
      g->holder = HOLDER_INT_MAP;
      igm_init(&g->imap);
    } else {
      g->holder = HOLDER_ARRAY;
      g->flags = (uint8_t*)po_xcalloc((size_t)g->global_upper_bound + 1, 1);
    }
  }
  if (g->holder != HOLDER_ARRAY) {
    g->raw_cap = 1024;
    g->raw_key_of_group = (int64_t*)po_xmalloc(sizeof(int64_t) * (size_t)g->raw_cap);
  }
  return 0;
}

static void gkg_remember(group_key_gen* g, int32_t gid, int64_t raw) {
  if (gid < 0) return;
  if (gid >= g->raw_cap) {
    while (gid >= g->raw_cap) g->raw_cap *= 2;
    g->raw_key_of_group = (int64_t*)po_xrealloc(g->raw_key_of_group, sizeof(int64_t) * (size_t)g->raw_cap);
  }
  g->raw_key_of_group[gid] = raw;
}

/* generateKeysForBlock → RawKeyHolder#processSingleValue (:285-354 array, :416-446 int map, :629-640 long map):
 * rawKey = sum_j dictId_j * prod_{i<j} card_i, built from the last column down */
static void gkg_generate(group_key_gen* g, int n_docs, int32_t** dict_ids, int32_t* out) {
  for (int i = 0; i < n_docs; i++) {
    int64_t raw = 0;
    for (int j = g->n_cols - 1; j >= 0; j--) raw = raw * g->cardinalities[j] + dict_ids[j][i];
    int32_t gid;
    if (g->holder == HOLDER_ARRAY) {
      gid = (int32_t)raw;
      if (!g->flags[gid]) { g->flags[gid] = 1; g->num_keys++; }
    } else if (g->holder == HOLDER_INT_MAP) {
      int32_t before = g->imap.size;
      gid = igm_get_group_id(&g->imap, (int32_t)raw, g->global_upper_bound);
      if (g->imap.size != before) gkg_remember(g, gid, raw);
    } else {
      int32_t before = g->lmap.size;
      gid = lm_get_group_id(&g->lmap, raw, g->global_upper_bound);
      if (g->lmap.size != before) gkg_remember(g, gid, raw);
    }
    out[i] = gid;
  }
}
/* NoDictionarySingleColumnGroupKeyGenerator#generateKeysForBlock :88-106 + getKeyForValue :241-265 (INT / LONG) */
static void gkg_generate_raw(group_key_gen* g, int n_docs, const int32_t* doc_ids, int32_t* out) {
  const po_column* c = g->cols[0];
  for (int i = 0; i < n_docs; i++) {
    int64_t v = c->data_type == PG_TYPE_INT ? (int64_t)po_raw_get_int(c, doc_ids[i]) : po_raw_get_long(c, doc_ids[i]);
    int32_t before = g->lmap.size;
    int32_t gid = lm_get_group_id(&g->lmap, v, g->global_upper_bound);
    if (g->lmap.size != before) gkg_remember(g, gid, v);
    out[i] = gid;
  }
}
/* the key a fastutil map compares: the int / long value, Float.floatToIntBits, Double.doubleToLongBits (one NaN; -0.0 != 0.0) */
static int64_t raw_key_of_doc(const po_column* c, int32_t doc) {
  switch (c->data_type) {
    case PG_TYPE_INT: return (int64_t)po_raw_get_int(c, doc);
    case PG_TYPE_LONG: return po_raw_get_long(c, doc);
    case PG_TYPE_FLOAT: { float f = po_raw_get_float(c, doc); uint32_t b; if (f != f) b = 0x7FC00000u; else memcpy(&b, &f, 4); return (int64_t)b; }
    default: { double d = po_raw_get_double(c, doc); uint64_t b; if (d != d) b = 0x7FF8000000000000ULL; else memcpy(&b, &d, 8); return (int64_t)b; }
  }
}
static uint64_t tuple_hash(const int64_t* t, int n) {
  uint64_t h = 0x9E3779B97F4A7C15ULL;
  for (int j = 0; j < n; j++) { h ^= (uint64_t)t[j] + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2); h *= 0xFF51AFD7ED558CCDULL; h ^= h >> 33; }
  return h;
}
static void tuple_table_insert(group_key_gen* g, int32_t gid) {
  uint64_t p = tuple_hash(g->tuples + (size_t)gid * (size_t)g->tuple_w, g->tuple_w) & (uint64_t)(g->tuple_table_cap - 1);
  while (g->tuple_table[p]) p = (p + 1) & (uint64_t)(g->tuple_table_cap - 1);
  g->tuple_table[p] = gid + 1;
}
/* generateKeysForBlock: existing tuple -> its id; a new tuple is admitted while fewer than numGroupsLimit groups exist, else INVALID_ID */
static void gkg_generate_tuples(group_key_gen* g, int n_docs, const int32_t* doc_ids, int32_t** dict_ids, int32_t* out) {
  const int nc = g->n_cols, tw = g->tuple_w;
  int64_t key[65];
  for (int i = 0; i < n_docs; i++) {
    int64_t null_mask = 0;
    for (int j = 0; j < nc; j++) {
      const po_column* c = g->cols[j];
      if (c->has_dictionary) key[j] = (int64_t)dict_ids[j][i];
      else if (g->raw_nulls && g->raw_nulls[j] && po_bitmap_contains(g->raw_nulls[j], doc_ids[i])) { key[j] = 0; null_mask |= (int64_t)1 << j; }
      else if (c->data_type > PG_TYPE_DOUBLE) {   /* STRING / BYTES: the on-the-fly dictionary's id of the value */
        int32_t len = 0;
        const uint8_t* v = po_raw_get_bytes(c, doc_ids[i], &len);
        key[j] = bytes_dict_index(&g->bytes_dicts[j], v, len);
      } else key[j] = raw_key_of_doc(c, doc_ids[i]);
    }
    if (tw > nc) key[nc] = null_mask;
    uint64_t p = tuple_hash(key, tw) & (uint64_t)(g->tuple_table_cap - 1);
    int32_t gid = PO_INVALID_ID;
    while (g->tuple_table[p]) {
      const int64_t* t = g->tuples + (size_t)(g->tuple_table[p] - 1) * (size_t)tw;
      if (memcmp(t, key, sizeof(int64_t) * (size_t)tw) == 0) { gid = g->tuple_table[p] - 1; break; }
      p = (p + 1) & (uint64_t)(g->tuple_table_cap - 1);
    }
    if (gid == PO_INVALID_ID && g->n_tuples < g->global_upper_bound) {
      if (g->n_tuples == g->tuple_cap) {
        g->tuple_cap *= 2;
        g->tuples = (int64_t*)po_xrealloc(g->tuples, sizeof(int64_t) * (size_t)g->tuple_cap * (size_t)tw);
      }
      gid = g->n_tuples++;
      memcpy(g->tuples + (size_t)gid * (size_t)tw, key, sizeof(int64_t) * (size_t)tw);
      if ((int64_t)g->n_tuples * 2 > g->tuple_table_cap) {   /* rehash */
        free(g->tuple_table);
        g->tuple_table_cap *= 4;
        g->tuple_table = (int32_t*)po_xcalloc((size_t)g->tuple_table_cap, 4);
        for (int32_t k = 0; k < g->n_tuples; k++) tuple_table_insert(g, k);
      } else {
        g->tuple_table[p] = gid + 1;
      }
    }
    out[i] = gid;
  }
}
static int32_t gkg_upper_bound(group_key_gen* g) { /* getCurrentGroupKeyUpperBound */
  if (g->holder == HOLDER_ARRAY) return g->global_upper_bound;
  if (g->holder == HOLDER_TUPLES) return g->n_tuples;
  return g->holder == HOLDER_INT_MAP ? g->imap.size : g->lmap.size;
}
static int32_t gkg_num_keys(group_key_gen* g) {
  if (g->holder == HOLDER_ARRAY) return g->num_keys;
  if (g->holder == HOLDER_TUPLES) return g->n_tuples;
  return g->holder == HOLDER_INT_MAP ? g->imap.size : g->lmap.size;
}

/* generateKeysForBlock(ValueBlock, int[][]) → RawKeyHolder#processMultiValue (:357-368 array, :449-459 int map, :648-660 long map) over
 * getIntRawKeys / getLongRawKeys (:504-573, :714-780): one raw key per combination of the doc's values, built from the last column
 * down — a single-value column (or a multi-value entry of one value) multiplies every key so far, a multi-value entry of k values
 * replaces the keys so far by k copies, copy j carrying value j.  Repeated values of a doc repeat their keys.
 * mv_off / mv_ids: per multi-value column the block's entries (offsets of n_docs + 1, dictIds back to back), NULL for single-value
 * columns.  out_off has n_docs + 1 entries; *out / *out_cap grow as needed. */
static void gkg_generate_mv(group_key_gen* g, int n_docs, int32_t** dict_ids, int32_t** mv_off, int32_t** mv_ids, int32_t* out_off,
                            int32_t** out, int32_t* out_cap) {
  int64_t* raw = NULL;
  int32_t raw_cap = 0, total = 0;
  for (int i = 0; i < n_docs; i++) {
    int32_t n_keys = 0;     /* 0: still the single rawKey */
    int64_t raw_key = 0;
    for (int j = g->n_cols - 1; j >= 0; j--) {
      const int64_t card = g->cardinalities[j];
      const int single = mv_off[j] == NULL;
      const int32_t n_values = single ? 1 : mv_off[j][i + 1] - mv_off[j][i];
      if (single || n_values == 1) {
        const int32_t d = single ? dict_ids[j][i] : mv_ids[j][mv_off[j][i]];
        if (n_keys == 0) raw_key = raw_key * card + d;
        else for (int32_t k = 0; k < n_keys; k++) raw[k] = raw[k] * card + d;
      } else {
        const int32_t* vals = mv_ids[j] + mv_off[j][i];
        const int32_t cur = n_keys == 0 ? 1 : n_keys, next = cur * n_values;
        if (next > raw_cap) { raw_cap = next * 2; raw = (int64_t*)po_xrealloc(raw, sizeof(int64_t) * (size_t)raw_cap); }
        if (n_keys == 0) raw[0] = raw_key;
        for (int32_t v = n_values - 1; v >= 0; v--)          /* copy v = the keys so far extended by value v (copy 0 last: in place) */
          for (int32_t k = 0; k < cur; k++) raw[v * cur + k] = raw[k] * card + vals[v];
        n_keys = next;
      }
    }
    if (n_keys == 0) {
      if (1 > raw_cap) { raw_cap = 16; raw = (int64_t*)po_xrealloc(raw, sizeof(int64_t) * (size_t)raw_cap); }
      raw[0] = raw_key;
      n_keys = 1;
    }
    if (total + n_keys > *out_cap) {
      *out_cap = (total + n_keys) * 2;
      *out = (int32_t*)po_xrealloc(*out, sizeof(int32_t) * (size_t)*out_cap);
    }
    out_off[i] = total;
    for (int32_t k = 0; k < n_keys; k++) {
      int32_t gid;
      if (g->holder == HOLDER_ARRAY) {
        gid = (int32_t)raw[k];
        if (!g->flags[gid]) { g->flags[gid] = 1; g->num_keys++; }
      } else if (g->holder == HOLDER_INT_MAP) {
        int32_t before = g->imap.size;
        gid = igm_get_group_id(&g->imap, (int32_t)raw[k], g->global_upper_bound);
        if (g->imap.size != before) gkg_remember(g, gid, raw[k]);
      } else {
        int32_t before = g->lmap.size;
        gid = lm_get_group_id(&g->lmap, raw[k], g->global_upper_bound);
        if (g->lmap.size != before) gkg_remember(g, gid, raw[k]);
      }
      (*out)[total++] = gid;
    }
  }
  out_off[n_docs] = total;
  free(raw);
}

/* the single-value function a multi-value function extends (CountMV extends Count, SumMV extends Sum, ...): holders, merge and
 * result extraction are the parent's */
static int sv_function_of(int f) {
  switch (f) {
    case PG_AGG_COUNTMV: return PG_AGG_COUNT;
    case PG_AGG_SUMMV: return PG_AGG_SUM;
    case PG_AGG_MINMV: return PG_AGG_MIN;
    case PG_AGG_MAXMV: return PG_AGG_MAX;
    case PG_AGG_AVGMV: return PG_AGG_AVG;
    case PG_AGG_MINMAXRANGEMV: return PG_AGG_MINMAXRANGE;
    case PG_AGG_DISTINCTCOUNTMV: return PG_AGG_DISTINCTCOUNT;
    case PG_AGG_DISTINCTCOUNTHLLMV: return PG_AGG_DISTINCTCOUNTHLL;
    default: return f;
  }
}
static int is_mv_function(int f) { return f >= PG_AGG_COUNTMV && f <= PG_AGG_DISTINCTCOUNTHLLMV; }

/* =====================================================================================================================
 * aggregation functions with their result holders
 * ===================================================================================================================== */
typedef struct agg_state {
  int function;
  po_column* col;          /* NULL for COUNT(*) */
  po_column* null_col;     /* COUNT(col) under null handling: the column whose nulls are not counted */
  int32_t log2m;
  int32_t capacity;        /* number of group slots */
  double* d0;              /* DoubleGroupByResultHolder / sum / min */
  double* d1;              /* max of MinMaxRangePair */
  int64_t* l0;             /* AvgPair count */
  uint8_t* has;            /* ObjectGroupByResultHolder: result != null */
  uint8_t* nn;             /* null handling: a non-null value reached the group (its result is not null) */
  int star;                /* aggregating a star-tree function-column pair column (pre-aggregated values) */
  po_bitmap** dict_bitmaps;/* DISTINCTCOUNT / HLL over dictionary columns: RoaringBitmap of dictIds */
  po_hll** hlls;           /* HLL over raw columns */
  struct po_vset* vsets;   /* DISTINCTCOUNT over a raw INT / LONG / FLOAT / DOUBLE column: the typed value sets (IntOpenHashSet ... DoubleOpenHashSet) */
} agg_state;

/* A value set of a raw column (BaseDistinctAggregateAggregationFunction.java:325-380: IntOpenHashSet / LongOpenHashSet / FloatOpenHashSet /
 * DoubleOpenHashSet, one per group).  Kept as order-preserving 64-bit keys — value ^ 2^63 for INT / LONG; for FLOAT (widened exactly) /
 * DOUBLE the IEEE bits with the sign folded in, so that -0.0 and 0.0 stay two elements as in fastutil's bit-wise equality — appended and
 * sort-uniqued when the array has doubled since the last compaction. */
typedef struct po_vset { uint64_t* k; int32_t n, cap, clean; } po_vset;
static int vset_cmp(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
static void vset_compact(po_vset* s) {
  if (s->n == s->clean) return;
  qsort(s->k, (size_t)s->n, 8, vset_cmp);
  int32_t m = 0;
  for (int32_t i = 0; i < s->n; i++) if (m == 0 || s->k[m - 1] != s->k[i]) s->k[m++] = s->k[i];
  s->n = s->clean = m;
}
static void vset_add(po_vset* s, uint64_t key) {
  if (s->n == s->cap) {
    if (s->n > 2 * s->clean + 64) vset_compact(s);
    if (s->n * 2 >= s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->k = (uint64_t*)po_xrealloc(s->k, 8 * (size_t)s->cap); }
  }
  s->k[s->n++] = key;
}
static uint64_t vset_key_of(const po_column* c, int32_t doc) {
  if (c->data_type == PG_TYPE_INT) return (uint64_t)(int64_t)po_raw_get_int(c, doc) ^ (1ULL << 63);
  if (c->data_type == PG_TYPE_LONG) return (uint64_t)po_raw_get_long(c, doc) ^ (1ULL << 63);
  double d = c->data_type == PG_TYPE_FLOAT ? (double)po_raw_get_float(c, doc) : po_raw_get_double(c, doc);
  uint64_t b; memcpy(&b, &d, 8);
  return (b >> 63) ? ~b : b ^ (1ULL << 63);
}
static int64_t vset_value_of(uint64_t key, int data_type, double* as_double) {   /* the value (INT / LONG) or the IEEE double bits behind a key */
  if (data_type == PG_TYPE_INT || data_type == PG_TYPE_LONG) { int64_t v = (int64_t)(key ^ (1ULL << 63)); *as_double = (double)v; return v; }
  uint64_t b = (key >> 63) ? key ^ (1ULL << 63) : ~key;
  memcpy(as_double, &b, 8);
  return (int64_t)b;
}

static void agg_ensure_capacity(agg_state* a, int32_t needed) { /* GroupByResultHolder#ensureCapacity */
  if (needed <= a->capacity) return;
  int32_t old = a->capacity;
  int32_t cap = old ? old : 16;
  while (cap < needed) cap *= 2;
  a->capacity = cap;
  double def0 = 0.0, def1 = 0.0;
  switch (sv_function_of(a->function)) {
    case PG_AGG_MIN: def0 = INFINITY; break;                 /* MinAggregationFunction DEFAULT_VALUE */
    case PG_AGG_MAX: def0 = -INFINITY; break;                /* MaxAggregationFunction.java:37 */
    default: break;
  }
  a->d0 = (double*)po_xrealloc(a->d0, sizeof(double) * (size_t)cap);
  a->d1 = (double*)po_xrealloc(a->d1, sizeof(double) * (size_t)cap);
  a->l0 = (int64_t*)po_xrealloc(a->l0, sizeof(int64_t) * (size_t)cap);
  a->has = (uint8_t*)po_xrealloc(a->has, (size_t)cap);
  a->nn = (uint8_t*)po_xrealloc(a->nn, (size_t)cap);
  for (int32_t i = old; i < cap; i++) { a->d0[i] = def0; a->d1[i] = def1; a->l0[i] = 0; a->has[i] = 0; a->nn[i] = 0; }
  if (sv_function_of(a->function) == PG_AGG_DISTINCTCOUNT || sv_function_of(a->function) == PG_AGG_DISTINCTCOUNTHLL) {
    a->dict_bitmaps = (po_bitmap**)po_xrealloc(a->dict_bitmaps, sizeof(void*) * (size_t)cap);
    a->hlls = (po_hll**)po_xrealloc(a->hlls, sizeof(void*) * (size_t)cap);
    a->vsets = (po_vset*)po_xrealloc(a->vsets, sizeof(po_vset) * (size_t)cap);
    for (int32_t i = old; i < cap; i++) { a->dict_bitmaps[i] = NULL; a->hlls[i] = NULL; memset(&a->vsets[i], 0, sizeof(po_vset)); }
  }
}

/* per-block column data = DataBlockCache (core/common/DataBlockCache.java:107-189) */
typedef struct block_col {
  po_column* col;
  int32_t* dict_ids;     /* valid if col->has_dictionary */
  double* doubles;       /* getDoubleValuesSV */
  int have_dict_ids, have_doubles;
  /* multi-value column: getDictionaryIdsMV / getDoubleValuesMV / getNumMVEntries of the block — offsets of n + 1, entries back to back;
   * the reader context lives as long as the column's ColumnValueReader (DataFetcher.java:317-333) */
  int32_t* mv_off; int32_t* mv_ids; double* mv_doubles; int32_t mv_cap;
  int have_mv, have_mv_doubles;
  po_mv_ctx mv_ctx;
} block_col;

static void fetch_mv_dict_ids(block_col* bc, const int32_t* doc_ids, int n) { /* DataFetcher.ColumnValueReader#readDictIdsMV :418-425 */
  if (bc->have_mv) return;
  po_column* c = bc->col;
  int32_t total = 0;
  for (int i = 0; i < n; i++) {
    if (total + c->mv_max_values > bc->mv_cap) {
      bc->mv_cap = (total + c->mv_max_values) * 2 + 16;
      bc->mv_ids = (int32_t*)po_xrealloc(bc->mv_ids, sizeof(int32_t) * (size_t)bc->mv_cap);
      bc->mv_doubles = (double*)po_xrealloc(bc->mv_doubles, sizeof(double) * (size_t)bc->mv_cap);
    }
    bc->mv_off[i] = total;
    total += po_mv_get_dict_ids(c, doc_ids[i], bc->mv_ids + total, &bc->mv_ctx);
  }
  bc->mv_off[n] = total;
  bc->have_mv = 1;
}
static void fetch_mv_doubles(block_col* bc, const int32_t* doc_ids, int n) { /* readDoubleValuesMV: dictionary.readDoubleValues per entry */
  if (bc->have_mv_doubles) return;
  fetch_mv_dict_ids(bc, doc_ids, n);
  for (int32_t k = 0; k < bc->mv_off[n]; k++) bc->mv_doubles[k] = po_dict_get_double(bc->col, bc->mv_ids[k]);
  bc->have_mv_doubles = 1;
}

static void fetch_dict_ids(block_col* bc, const int32_t* doc_ids, int n) { /* DataFetcher#fetchDictIds */
  if (bc->have_dict_ids) return;
  po_fwd_read_dict_ids(bc->col, doc_ids, n, bc->dict_ids);
  bc->have_dict_ids = 1;
}
static void fetch_doubles(block_col* bc, const int32_t* doc_ids, int n) { /* DataFetcher.ColumnValueReader#readDoubleValues :335-386 */
  if (bc->have_doubles) return;
  po_column* c = bc->col;
  if (c->has_dictionary) {
    fetch_dict_ids(bc, doc_ids, n);
    for (int i = 0; i < n; i++) bc->doubles[i] = po_dict_get_double(c, bc->dict_ids[i]);
  } else {
    switch (c->data_type) {
      case PG_TYPE_INT: for (int i = 0; i < n; i++) bc->doubles[i] = (double)po_raw_get_int(c, doc_ids[i]); break;
      case PG_TYPE_LONG: for (int i = 0; i < n; i++) bc->doubles[i] = (double)po_raw_get_long(c, doc_ids[i]); break;
      case PG_TYPE_FLOAT: for (int i = 0; i < n; i++) bc->doubles[i] = (double)po_raw_get_float(c, doc_ids[i]); break;
      default: for (int i = 0; i < n; i++) bc->doubles[i] = po_raw_get_double(c, doc_ids[i]); break;
    }
  }
  bc->have_doubles = 1;
}

static void hll_offer_raw(po_hll* h, po_column* c, int32_t doc_id) {
  switch (c->data_type) {
    case PG_TYPE_INT: po_hll_offer_int(h, po_raw_get_int(c, doc_id)); break;
    case PG_TYPE_LONG: po_hll_offer_long(h, po_raw_get_long(c, doc_id)); break;
    case PG_TYPE_FLOAT: po_hll_offer_float(h, po_raw_get_float(c, doc_id)); break;
    default: po_hll_offer_double(h, po_raw_get_double(c, doc_id)); break;
  }
}

/* aggregate / aggregateGroupBySV / aggregateGroupByMV of each function.  group_keys == NULL and mvk_off == NULL: the non-group-by
 * `aggregate` (single holder 0); mvk_off != NULL: the block's group keys are lists (a multi-value group-by column: every key of a doc
 * receives the doc's value, in list order).
 *   COUNT  CountAggregationFunction.java:110-143 (holder += 1 in double)      SUM   SumAggregationFunction.java:160-179
 *   MIN    MinAggregationFunction.java:163-188 (value < holder)               MAX   MaxAggregationFunction.java:163-188
 *   AVG    AvgAggregationFunction (AvgPair sum,count)                         MINMAXRANGE MinMaxRangeAggregationFunction
 *   DISTINCTCOUNT BaseDistinctAggregateAggregationFunction.java:306-345       HLL   DistinctCountHLLAggregationFunction.java:152-222
 *   the *MV forms: CountMV / SumMV / MinMV / MaxMV / AvgMV / MinMaxRangeMV AggregationFunction.java (whole files), DISTINCTCOUNTMV /
 *   DISTINCTCOUNTHLLMV: the dictId bitmap takes every dictId of the doc */
#define FOR_EACH_GROUP(i, g)                                                                                             \
  for (int32_t gk_ = 0, gk_n_ = mvk_off ? mvk_off[(i) + 1] - mvk_off[(i)] : 1; gk_ < gk_n_; gk_++)                        \
    for (int32_t g = mvk_off ? mvk[mvk_off[(i)] + gk_] : (group_keys ? group_keys[(i)] : 0), once_ = 1; once_ && g != PO_INVALID_ID; once_ = 0)
static void agg_process_block(agg_state* a, block_col* bc, const int32_t* doc_ids, int n, const int32_t* group_keys,
                              const int32_t* mvk_off, const int32_t* mvk) {
  const int grouped = group_keys != NULL || mvk_off != NULL;
  switch (a->function) {
    case PG_AGG_COUNT:
      if (a->star) { /* star-tree pre-aggregated values: CountAggregationFunction.java:99-106,134-141 */
        if (!grouped) {
          int64_t count = 0;
          for (int i = 0; i < n; i++) count += po_raw_get_long(a->col, doc_ids[i]);
          a->d0[0] = a->d0[0] + (double)count;
          return;
        }
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) a->d0[g] = a->d0[g] + (double)po_raw_get_long(a->col, doc_ids[i]);
        return;
      }
      if (!grouped) { a->d0[0] += n; return; }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) a->d0[g] += 1;
      return;
    case PG_AGG_SUM: {
      fetch_doubles(bc, doc_ids, n);
      if (!grouped) { /* SumAggregationFunction#aggregate: per-block inner sum, then holder += innerSum */
        double inner = 0;
        for (int i = 0; i < n; i++) inner += bc->doubles[i];
        a->d0[0] = inner + a->d0[0];
        return;
      }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) a->d0[g] = a->d0[g] + bc->doubles[i];
      return;
    }
    case PG_AGG_MIN:
      fetch_doubles(bc, doc_ids, n);
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) if (bc->doubles[i] < a->d0[g]) a->d0[g] = bc->doubles[i];
      return;
    case PG_AGG_MAX:
      fetch_doubles(bc, doc_ids, n);
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) if (bc->doubles[i] > a->d0[g]) a->d0[g] = bc->doubles[i];
      return;
    case PG_AGG_AVG:
      if (a->col->data_type == PG_TYPE_BYTES) {   /* serialized AvgPair (star-tree pair avg__x): AvgAggregationFunction.java:79-93,117-126 */
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
          int32_t len = 0;
          const uint8_t* blob = po_raw_get_bytes(a->col, doc_ids[i], &len);
          if (len < 16) continue;
          a->d0[g] += po_bef64(blob); a->l0[g] += (int64_t)po_be64(blob + 8); a->has[g] = 1;   /* AvgPair#apply(sum, count) */
        }
        return;
      }
      fetch_doubles(bc, doc_ids, n);
      if (!grouped) {
        double inner = 0;
        for (int i = 0; i < n; i++) inner += bc->doubles[i];
        a->d0[0] += inner; a->l0[0] += n; a->has[0] = 1;
        return;
      }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) { a->d0[g] += bc->doubles[i]; a->l0[g] += 1; a->has[g] = 1; }
      return;
    case PG_AGG_MINMAXRANGE:
      if (a->col->data_type == PG_TYPE_BYTES) {   /* serialized MinMaxRangePair (star-tree pair minMaxRange__x): MinMaxRangeAggregationFunction */
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
          int32_t len = 0;
          const uint8_t* blob = po_raw_get_bytes(a->col, doc_ids[i], &len);
          if (len < 16) continue;
          const double lo = po_bef64(blob), hi = po_bef64(blob + 8);
          if (!a->has[g]) { a->d0[g] = lo; a->d1[g] = hi; a->has[g] = 1; }
          else { if (lo < a->d0[g]) a->d0[g] = lo; if (hi > a->d1[g]) a->d1[g] = hi; }
        }
        return;
      }
      fetch_doubles(bc, doc_ids, n);
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
        double v = bc->doubles[i];
        if (!a->has[g]) { a->d0[g] = v; a->d1[g] = v; a->has[g] = 1; }
        else { if (v < a->d0[g]) a->d0[g] = v; if (v > a->d1[g]) a->d1[g] = v; }
      }
      return;
    case PG_AGG_DISTINCTCOUNT:
    case PG_AGG_DISTINCTCOUNTHLL: {
      po_column* c = a->col;
      if (c->data_type == PG_TYPE_BYTES) { /* serialized HyperLogLog (star-tree pair): DistinctCountHLLAggregationFunction.java:158-175 */
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
          int32_t len = 0;
          const uint8_t* blob = po_raw_get_bytes(c, doc_ids[i], &len);
          po_hll* v = po_hll_deserialize(blob, len);
          if (!v) continue;
          if (a->hlls[g]) { po_hll_merge(a->hlls[g], v); po_hll_free(v); }   /* hyperLogLog.addAll(value) */
          else a->hlls[g] = v;
        }
        return;
      }
      if (c->has_dictionary) {
        fetch_dict_ids(bc, doc_ids, n);
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
          if (!a->dict_bitmaps[g]) a->dict_bitmaps[g] = po_bitmap_new_small(c->cardinality);
          po_bitmap_add(a->dict_bitmaps[g], bc->dict_ids[i]);
        }
      } else if (a->function == PG_AGG_DISTINCTCOUNTHLL) {
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
          if (!a->hlls[g]) a->hlls[g] = po_hll_new(a->log2m);
          hll_offer_raw(a->hlls[g], c, doc_ids[i]);
        }
      } else {   /* DISTINCTCOUNT over a raw column: valueSet.add(value) per doc (BaseDistinctAggregateAggregationFunction.java:325-380) */
        for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) vset_add(&a->vsets[g], vset_key_of(c, doc_ids[i]));
      }
      return;
    }
    /* ---- the multi-value forms ------------------------------------------------------------------------------------------------- */
    case PG_AGG_COUNTMV: {   /* CountMVAggregationFunction.java:62-96: getNumMVEntries */
      fetch_mv_dict_ids(bc, doc_ids, n);
      if (!grouped) {
        int64_t count = 0;
        for (int i = 0; i < n; i++) count += bc->mv_off[i + 1] - bc->mv_off[i];
        a->d0[0] = a->d0[0] + (double)count;
        return;
      }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) a->d0[g] = a->d0[g] + (bc->mv_off[i + 1] - bc->mv_off[i]);
      return;
    }
    case PG_AGG_SUMMV:       /* SumMVAggregationFunction.java:41-84: sum = holder; sum += value for every value; holder = sum */
      fetch_mv_doubles(bc, doc_ids, n);
      if (!grouped) {
        double sum = a->d0[0];
        for (int32_t k = 0; k < bc->mv_off[n]; k++) sum += bc->mv_doubles[k];
        a->d0[0] = sum;
        return;
      }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
        double sum = a->d0[g];
        for (int32_t k = bc->mv_off[i]; k < bc->mv_off[i + 1]; k++) sum += bc->mv_doubles[k];
        a->d0[g] = sum;
      }
      return;
    case PG_AGG_MINMV:       /* MinMVAggregationFunction.java:41-93 */
      fetch_mv_doubles(bc, doc_ids, n);
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g)
        for (int32_t k = bc->mv_off[i]; k < bc->mv_off[i + 1]; k++) if (bc->mv_doubles[k] < a->d0[g]) a->d0[g] = bc->mv_doubles[k];
      return;
    case PG_AGG_MAXMV:       /* MaxMVAggregationFunction.java */
      fetch_mv_doubles(bc, doc_ids, n);
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g)
        for (int32_t k = bc->mv_off[i]; k < bc->mv_off[i + 1]; k++) if (bc->mv_doubles[k] > a->d0[g]) a->d0[g] = bc->mv_doubles[k];
      return;
    case PG_AGG_AVGMV:       /* AvgMVAggregationFunction.java:41-95: block sum + count (aggregate), per-doc sum + count (group-by) */
      fetch_mv_doubles(bc, doc_ids, n);
      if (!grouped) {
        double sum = 0.0;
        for (int32_t k = 0; k < bc->mv_off[n]; k++) sum += bc->mv_doubles[k];
        a->d0[0] += sum; a->l0[0] += bc->mv_off[n]; a->has[0] = 1;
        return;
      }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
        double sum = 0.0;
        for (int32_t k = bc->mv_off[i]; k < bc->mv_off[i + 1]; k++) sum += bc->mv_doubles[k];
        a->d0[g] += sum; a->l0[g] += bc->mv_off[i + 1] - bc->mv_off[i]; a->has[g] = 1;
      }
      return;
    case PG_AGG_MINMAXRANGEMV: {   /* MinMaxRangeMVAggregationFunction.java:41-106: (min, max) of the block / the doc, then the pair update */
      fetch_mv_doubles(bc, doc_ids, n);
      if (!grouped) {
        double lo = INFINITY, hi = -INFINITY;
        for (int32_t k = 0; k < bc->mv_off[n]; k++) { if (bc->mv_doubles[k] < lo) lo = bc->mv_doubles[k]; if (bc->mv_doubles[k] > hi) hi = bc->mv_doubles[k]; }
        if (!a->has[0]) { a->d0[0] = lo; a->d1[0] = hi; a->has[0] = 1; }
        else { if (lo < a->d0[0]) a->d0[0] = lo; if (hi > a->d1[0]) a->d1[0] = hi; }
        return;
      }
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
        double lo = INFINITY, hi = -INFINITY;
        for (int32_t k = bc->mv_off[i]; k < bc->mv_off[i + 1]; k++) { if (bc->mv_doubles[k] < lo) lo = bc->mv_doubles[k]; if (bc->mv_doubles[k] > hi) hi = bc->mv_doubles[k]; }
        if (!a->has[g]) { a->d0[g] = lo; a->d1[g] = hi; a->has[g] = 1; }
        else { if (lo < a->d0[g]) a->d0[g] = lo; if (hi > a->d1[g]) a->d1[g] = hi; }
      }
      return;
    }
    case PG_AGG_DISTINCTCOUNTMV:
    case PG_AGG_DISTINCTCOUNTHLLMV:   /* dictionary-encoded: RoaringBitmap#add(int[]) of the doc's dictIds */
      fetch_mv_dict_ids(bc, doc_ids, n);
      for (int i = 0; i < n; i++) FOR_EACH_GROUP(i, g) {
        if (!a->dict_bitmaps[g]) a->dict_bitmaps[g] = po_bitmap_new_small(a->col->cardinality);
        for (int32_t k = bc->mv_off[i]; k < bc->mv_off[i + 1]; k++) po_bitmap_add(a->dict_bitmaps[g], bc->mv_ids[k]);
      }
      return;
    default: return;
  }
}

/* DistinctCountHLLAggregationFunction#convertToHyperLogLog :457-466: offer dictionary.get(dictId) for every dictId */
static po_hll* hll_from_dict_bitmap(const po_bitmap* b, const po_column* c, int32_t log2m) {
  po_hll* h = po_hll_new(log2m);
  if (!b) return h;
  for (int64_t d = po_bitmap_next_set(b, 0); d >= 0 && d < c->cardinality; d = po_bitmap_next_set(b, d + 1)) {
    switch (c->data_type) {
      case PG_TYPE_INT: po_hll_offer_int(h, po_dict_get_int(c, (int32_t)d)); break;
      case PG_TYPE_LONG: po_hll_offer_long(h, po_dict_get_long(c, (int32_t)d)); break;
      case PG_TYPE_FLOAT: po_hll_offer_float(h, po_dict_get_float(c, (int32_t)d)); break;
      case PG_TYPE_DOUBLE: po_hll_offer_double(h, po_bef64(c->dict + d * 8)); break;
      default: {
        const uint8_t* e = c->dict + d * c->dict_bytes_per_value;
        int len = c->dict_bytes_per_value;
        while (len > 0 && e[len - 1] == 0) len--;
        po_hll_offer_string(h, e, len);
        break;
      }
    }
  }
  return h;
}

/* =====================================================================================================================
 * results
 * ===================================================================================================================== */
typedef struct po_agg_result {
  int kind;
  double* d[2];
  int64_t* l[2];
  int32_t* set_sizes; int32_t* set_ids; int64_t set_total;
  int64_t* set_l; double* set_d; int set_type;   /* PG_RESULT_VALUE_SET: the concatenated ascending values (set_type = the column's data type) */
  uint8_t* hll; int32_t log2m;
  uint8_t* nulls;            /* null handling: 1 where the group's result is null (NULL: none is) */
} po_agg_result;

typedef struct po_result_impl {
  int32_t num_groups, n_group_cols, n_aggs;
  int32_t** group_dict_ids;
  int64_t* group_values;     /* raw-value group keys (one no-dictionary INT / LONG group-by column), else NULL */
  int32_t* key_types;        /* HOLDER_TUPLES: PG_GROUP_KEY_* per group-by column, else NULL */
  int64_t** key_values;      /* HOLDER_TUPLES: per value-keyed column the groups' LONG values / DOUBLE bits */
  uint8_t** key_bytes;       /* HOLDER_TUPLES: per raw STRING / BYTES column the groups' values back to back, */
  int64_t** key_bytes_off;   /*   and their offsets (num_groups + 1) */
  uint8_t** key_nulls;       /* null handling: per group-by column 1 where the group's key is null (NULL: no null key) */
  po_agg_result* aggs;
  pg_exec_stats stats;
} po_result_impl;

static int result_kind(int function) {
  switch (sv_function_of(function)) {
    case PG_AGG_COUNT: return PG_RESULT_LONG;
    case PG_AGG_AVG: return PG_RESULT_AVG_PAIR;
    case PG_AGG_MINMAXRANGE: return PG_RESULT_MINMAX_PAIR;
    case PG_AGG_DISTINCTCOUNT: return PG_RESULT_DICTID_SET;
    case PG_AGG_DISTINCTCOUNTHLL: return PG_RESULT_HLL;
    default: return PG_RESULT_DOUBLE;
  }
}

static void extract_agg(po_agg_result* r, agg_state* a, int32_t n_groups, const int32_t* gid_of) {
  r->kind = result_kind(a->function);
  if (r->kind == PG_RESULT_DICTID_SET && a->col && !a->col->has_dictionary) r->kind = PG_RESULT_VALUE_SET;
  r->log2m = a->log2m;
  for (int k = 0; k < 2; k++) {
    r->d[k] = (double*)po_xcalloc((size_t)n_groups + 1, 8);
    r->l[k] = (int64_t*)po_xcalloc((size_t)n_groups + 1, 8);
  }
  if (r->kind == PG_RESULT_DICTID_SET) {
    r->set_sizes = (int32_t*)po_xcalloc((size_t)n_groups + 1, 4);
    int64_t total = 0;
    for (int32_t i = 0; i < n_groups; i++) {
      po_bitmap* b = a->dict_bitmaps[gid_of[i]];
      r->set_sizes[i] = b ? (int32_t)po_bitmap_cardinality(b) : 0;
      total += r->set_sizes[i];
    }
    r->set_total = total;
    r->set_ids = (int32_t*)po_xcalloc((size_t)total + 1, 4);
    int64_t k = 0;
    for (int32_t i = 0; i < n_groups; i++) {
      po_bitmap* b = a->dict_bitmaps[gid_of[i]];
      if (!b) continue;
      for (int64_t d = po_bitmap_next_set(b, 0); d >= 0; d = po_bitmap_next_set(b, d + 1)) r->set_ids[k++] = (int32_t)d;
    }
    return;
  }
  if (r->kind == PG_RESULT_VALUE_SET) {
    r->set_sizes = (int32_t*)po_xcalloc((size_t)n_groups + 1, 4);
    r->set_type = a->col->data_type;
    int64_t total = 0;
    for (int32_t i = 0; i < n_groups; i++) {
      po_vset* vs = a->vsets ? &a->vsets[gid_of[i]] : NULL;
      if (vs) vset_compact(vs);
      r->set_sizes[i] = vs ? vs->n : 0;
      total += r->set_sizes[i];
    }
    r->set_total = total;
    r->set_l = (int64_t*)po_xcalloc((size_t)total + 1, 8);
    r->set_d = (double*)po_xcalloc((size_t)total + 1, 8);
    int64_t k = 0;
    for (int32_t i = 0; i < n_groups; i++) {
      po_vset* vs = a->vsets ? &a->vsets[gid_of[i]] : NULL;
      for (int32_t e = 0; vs && e < vs->n; e++, k++) r->set_l[k] = vset_value_of(vs->k[e], r->set_type, &r->set_d[k]);
    }
    return;
  }
  if (r->kind == PG_RESULT_HLL) {
    int m = 1 << a->log2m;
    r->hll = (uint8_t*)po_xcalloc((size_t)n_groups * (size_t)m + 1, 1);
    for (int32_t i = 0; i < n_groups; i++) {
      int32_t g = gid_of[i];
      po_hll* h = a->col->has_dictionary ? hll_from_dict_bitmap(a->dict_bitmaps[g], a->col, a->log2m)
                                         : (a->hlls[g] ? a->hlls[g] : po_hll_new(a->log2m));
      memcpy(r->hll + (size_t)i * (size_t)m, h->regs, (size_t)m);
      if (a->col->has_dictionary || !a->hlls[g]) po_hll_free(h);
    }
    return;
  }
  for (int32_t i = 0; i < n_groups; i++) {
    int32_t g = gid_of[i];
    switch (sv_function_of(a->function)) {
      case PG_AGG_COUNT: r->l[0][i] = (int64_t)a->d0[g]; break;          /* extractGroupByResult: (long) double */
      case PG_AGG_AVG: r->d[0][i] = a->d0[g]; r->l[0][i] = a->l0[g]; break;
      case PG_AGG_MINMAXRANGE:   /* extractAggregationResult / extractGroupByResult (:162-181): no value → new MinMaxRangePair() = (+inf, -inf) */
        r->d[0][i] = a->has[g] ? a->d0[g] : INFINITY;
        r->d[1][i] = a->has[g] ? a->d1[g] : -INFINITY;
        break;
      default: r->d[0][i] = a->d0[g]; break;
    }
  }
}

/* =====================================================================================================================
 * segment-level group trim: GroupByOperator.java:120-133 -> GroupByUtils.getTableCapacity (core/util/GroupByUtils.java:45-57) ->
 * TableResizer#trimInSegmentResults (core/data/table/TableResizer.java:327-351) with the comparator of its constructor (:88-128, without
 * null handling) over the extractors of :129-161 / :406-445: a group-by expression's VALUE, an aggregation's extractFinalResult.
 * The reference keeps a heap of `size` records; which records TIED with the last one kept survive depends on the heap — here a stable
 * sort decides (tests put a unique value at the cut).
 * ===================================================================================================================== */
typedef struct order_value { int type; int64_t l; double d; const uint8_t* b; int32_t blen; int is_null; } order_value;   /* type 0 long / int, 1 double, 2 BYTES, 3 STRING */
/* String.compareTo (TableResizer's comparators on STRING keys) orders UTF-16 code units; UTF-8 byte order differs only where a supplementary
 * character (lead byte F0..F4: surrogates D800..DFFF) meets U+E000..U+FFFF (lead byte EE / EF), which sort behind it in UTF-16 */
static int utf16_unit_order(const uint8_t* a, int32_t alen, const uint8_t* b, int32_t blen) {
  const int32_t n = alen < blen ? alen : blen;
  for (int32_t i = 0; i < n; i++) {
    if (a[i] == b[i]) continue;
    const int x = a[i] == 0xEE || a[i] == 0xEF ? a[i] + 0x10 : a[i], y = b[i] == 0xEE || b[i] == 0xEF ? b[i] + 0x10 : b[i];
    return x < y ? -1 : 1;
  }
  return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
typedef struct order_ctx { const order_value* v; int n_ob; const int* asc; const int* nulls_last; } order_ctx;   /* nulls_last: NULL without null handling */
static int double_compare_java(double a, double b) {   /* Double.compare: -0.0 < 0.0, NaN above everything and equal to itself */
  if (a < b) return -1;
  if (a > b) return 1;
  int64_t x, y;
  memcpy(&x, &a, 8); memcpy(&y, &b, 8);
  if (a != a) x = INT64_MAX;   /* doubleToLongBits canonicalises NaN */
  if (b != b) y = INT64_MAX;
  return x == y ? 0 : (x < y ? -1 : 1);
}
static int order_cmp(const void* pa, const void* pb, void* ctxp) {
  const order_ctx* c = (const order_ctx*)ctxp;
  const int32_t ia = *(const int32_t*)pa, ib = *(const int32_t*)pb;
  for (int k = 0; k < c->n_ob; k++) {
    const order_value* a = &c->v[(size_t)ia * (size_t)c->n_ob + (size_t)k];
    const order_value* b = &c->v[(size_t)ib * (size_t)c->n_ob + (size_t)k];
    int r;
    if (c->nulls_last && (a->is_null || b->is_null)) {   /* TableResizer.java:98-116: nullComparisonResults[i] = isNullsLast ? -1 : 1, whatever the direction */
      if (a->is_null && b->is_null) continue;
      const int ncr = c->nulls_last[k] ? -1 : 1;
      return a->is_null ? -ncr : ncr;
    }
    if (a->type == 0) r = a->l < b->l ? -1 : (a->l > b->l ? 1 : 0);
    else if (a->type == 1) r = double_compare_java(a->d, b->d);
    else if (a->type == 3) r = utf16_unit_order(a->b, a->blen, b->b, b->blen);
    else {
      const int32_t n = a->blen < b->blen ? a->blen : b->blen;
      r = n ? memcmp(a->b, b->b, (size_t)n) : 0;   /* ByteArray.compare: unsigned bytes */
      if (r == 0) r = a->blen < b->blen ? -1 : (a->blen > b->blen ? 1 : 0);
    }
    if (r != 0) return c->asc[k] ? r : -r;
  }
  return ia < ib ? -1 : (ia > ib ? 1 : 0);   /* stable */
}
/* AggregationFunction#extractFinalResult of group `g`: COUNT Long; SUM / MIN / MAX Double; AVG sum / count (AvgAggregationFunction.java:
 * DEFAULT_FINAL_RESULT -inf for count 0); MINMAXRANGE max - min; DISTINCTCOUNT the set's size (Integer); DISTINCTCOUNTHLL cardinality (Long) */
static order_value agg_final_value(agg_state* a, int32_t g) {
  order_value v;
  memset(&v, 0, sizeof(v));
  switch (sv_function_of(a->function)) {
    case PG_AGG_COUNT: v.type = 0; v.l = (int64_t)a->d0[g]; break;
    case PG_AGG_AVG: v.type = 1; v.d = a->l0[g] == 0 ? -INFINITY : a->d0[g] / (double)a->l0[g]; break;
    case PG_AGG_MINMAXRANGE: v.type = 1; v.d = a->has[g] ? a->d1[g] - a->d0[g] : -INFINITY - INFINITY; break;
    case PG_AGG_DISTINCTCOUNT:
      v.type = 0;
      if (a->col && !a->col->has_dictionary && a->vsets) { vset_compact(&a->vsets[g]); v.l = a->vsets[g].n; }
      else v.l = a->dict_bitmaps[g] ? po_bitmap_cardinality(a->dict_bitmaps[g]) : 0;
      break;
    case PG_AGG_DISTINCTCOUNTHLL: {
      v.type = 0;
      po_hll* h = a->col->has_dictionary ? hll_from_dict_bitmap(a->dict_bitmaps[g], a->col, a->log2m) : (a->hlls[g] ? a->hlls[g] : po_hll_new(a->log2m));
      v.l = po_hll_cardinality(h);
      if (a->col->has_dictionary || !a->hlls[g]) po_hll_free(h);
      break;
    }
    default: v.type = 1; v.d = a->d0[g]; break;
  }
  return v;
}

/* =====================================================================================================================
 * query execution: GroupByOperator.getNextBlock (core/operator/query/GroupByOperator.java:100-140),
 * AggregationOperator.getNextBlock, FastFilteredCountOperator
 * ===================================================================================================================== */
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int32_t po_result_free(void* r);
/* PG_QUERY_FLAG_NULL_HANDLING (QueryContext#isNullHandlingEnabled), restated natively (doc at a time, like everything in this oracle): filters
 * in three-valued logic (po_filter.c), aggregations that skip the docs whose argument is null and are null over no value, null group keys.
 * null_check_* below answer "does the query read a column that holds a null" — which is when a star-tree may not answer (StarTreeUtils.java:381-418). */
static int null_check_column(po_segment* seg, const char* name) {
  if (!name || !strcmp(name, "*")) return 0;
  po_column* c = po_segment_column(seg, name);
  if (c && c->null_bitmap && po_bitmap_cardinality(c->null_bitmap) > 0) {
    po_set_error("enableNullHandling over column %s, which holds nulls in this segment", name);
    return 1;
  }
  return 0;
}
static int null_check_filter(po_segment* seg, const pg_filter_node* f) {
  if (!f) return 0;
  if (f->type == PG_FILTER_PREDICATE) return null_check_column(seg, f->column);
  for (int i = 0; i < f->n_children; i++) if (null_check_filter(seg, &f->children[i])) return 1;
  return 0;
}
static const po_bitmap* col_nulls(const po_column* c) {   /* NullValueVectorReader#getNullBitmap, non-empty */
  return c && c->null_bitmap && po_bitmap_cardinality(c->null_bitmap) > 0 ? c->null_bitmap : NULL;
}
/* What the restatement of query-level null handling leaves out (refused like the product path refuses it): nulls in multi-value columns and
 * in no-dictionary group-by columns, nulls in a group-by column next to a multi-value group-by column. */
static int null_handling_refused(po_segment* seg, const pg_query* q) {
  if (!(q->flags & PG_QUERY_FLAG_NULL_HANDLING)) return 0;
  int mv_gb = 0, null_gb = 0;
  for (int i = 0; i < q->n_group_by; i++) {
    po_column* c = po_segment_column(seg, q->group_by_columns[i]);
    if (!c) continue;
    mv_gb |= c->is_mv;
    if (col_nulls(c)) {
      null_gb = 1;
      if (c->is_mv) { po_set_error("enableNullHandling: nulls in the multi-value group-by column %s", c->name); return 1; }
    }
  }
  if (mv_gb && null_gb) { po_set_error("enableNullHandling: null group keys next to a multi-value group-by column"); return 1; }
  for (int i = 0; i < q->n_aggregations; i++) {
    const char* name = q->aggregations[i].column;
    po_column* c = name && strcmp(name, "*") ? po_segment_column(seg, name) : NULL;
    if (c && col_nulls(c) && c->is_mv) { po_set_error("enableNullHandling: nulls in the multi-value column %s", c->name); return 1; }
  }
  return 0;
}
/* the product library leaves these to the Java plan; the oracle refuses them alike so that the two sides answer the same queries */
static int trim_refused(const pg_query* q) {
  if (!(q->n_group_by > 0 && q->n_order_by > 0 && q->order_by && q->min_segment_group_trim_size > 0)) return 0;
  for (int32_t i = 0; i < q->n_order_by; i++)
    if (q->order_by[i].kind == PG_ORDER_BY_AGGREGATION && q->order_by[i].index >= 0 && q->order_by[i].index < q->n_aggregations) {
      const int f = sv_function_of(q->aggregations[q->order_by[i].index].function);
      if (!(f == PG_AGG_COUNT || f == PG_AGG_SUM || f == PG_AGG_MIN || f == PG_AGG_MAX || f == PG_AGG_AVG || f == PG_AGG_MINMAXRANGE || f == PG_AGG_DISTINCTCOUNT ||
            f == PG_AGG_DISTINCTCOUNTHLL)) {
        po_set_error("segment-level group trim ordered by aggregation function %d", f);
        return 1;
      }
    }
  return 0;
}
int32_t po_query_supported(void* seg, const pg_query* q) { return (null_handling_refused((po_segment*)seg, q) || trim_refused(q)) ? PG_ERR_UNSUPPORTED : PG_OK; }

int32_t po_query_exec(void* segp, const pg_query* q, void** out) {
  if (trim_refused(q)) return PG_ERR_UNSUPPORTED;
  double t0 = now_ms();
  po_segment* seg = (po_segment*)segp;
  int32_t num_groups_limit = q->num_groups_limit > 0 ? q->num_groups_limit : DEFAULT_NUM_GROUPS_LIMIT;
  int32_t max_init_cap = q->max_initial_result_holder_capacity > 0 ? q->max_initial_result_holder_capacity
                                                                   : DEFAULT_MAX_INITIAL_RESULT_HOLDER_CAPACITY;
  int n_aggs = q->n_aggregations, n_gb = q->n_group_by;
  if (n_aggs <= 0) { po_set_error("query has no aggregation"); return PG_ERR_INVALID_ARGUMENT; }
  if (null_handling_refused(seg, q)) return PG_ERR_UNSUPPORTED;

  po_filter_op* filter_op = po_filter_plan(seg, q->filter, (q->flags & PG_QUERY_FLAG_NULL_HANDLING) != 0);
  if (!filter_op) return PG_ERR_INVALID_ARGUMENT;

  po_result_impl* res = (po_result_impl*)po_xcalloc(1, sizeof(*res));
  res->n_aggs = n_aggs;
  res->n_group_cols = n_gb;
  res->stats.num_total_docs = seg->total_docs;
  res->stats.stats_exact = 1;
  res->stats.star_tree_index = -1;

  /* resolve columns */
  agg_state* aggs = (agg_state*)po_xcalloc((size_t)n_aggs, sizeof(agg_state));
  po_column** proj = (po_column**)po_xcalloc((size_t)(n_aggs + n_gb) * 2 + 1, sizeof(po_column*));
  int n_proj = 0;
  for (int i = 0; i < n_aggs; i++) {
    const pg_agg_spec* s = &q->aggregations[i];
    aggs[i].function = s->function;
    aggs[i].log2m = s->log2m > 0 ? s->log2m : DEFAULT_LOG2M;
    if (s->function == PG_AGG_COUNT) {   /* COUNT(*) takes no input expression; COUNT(col) under null handling counts the values that are
                                          * not null (CountAggregationFunction.java:88-98,118-131) */
      if ((q->flags & PG_QUERY_FLAG_NULL_HANDLING) && s->column && strcmp(s->column, "*")) {
        po_column* cc = po_segment_column(seg, s->column);
        if (cc && col_nulls(cc) && !cc->is_mv) aggs[i].null_col = cc;
      }
      continue;
    }
    po_column* c = po_segment_column(seg, s->column);
    if (!c) { po_set_error("column not found: %s", s->column ? s->column : "(null)"); return PG_ERR_NOT_FOUND; }
    if ((s->function == PG_AGG_DISTINCTCOUNT) && !c->has_dictionary &&
        (c->data_type > PG_TYPE_DOUBLE || c->is_mv || (q->flags & PG_QUERY_FLAG_NULL_HANDLING))) {   /* numeric single-value raw columns: typed value sets */
      po_set_error("DISTINCTCOUNT over the raw column %s is outside the hot path", c->name);
      return PG_ERR_UNSUPPORTED;
    }
    if (c->raw_mv && sv_function_of(s->function) == PG_AGG_DISTINCTCOUNT) {   /* its intermediate is a VALUE set: not built (po_raw_mv_attach) */
      po_set_error("DISTINCTCOUNTMV over the raw multi-value column %s is outside the hot path", c->name);
      return PG_ERR_UNSUPPORTED;
    }
    if (is_mv_function(s->function) != (c->is_mv != 0)) {   /* BlockValSet#getDoubleValuesSV / MV on the wrong kind of column throws */
      po_set_error("aggregation %d over %s column %s", s->function, c->is_mv ? "multi-value" : "single-value", c->name);
      return PG_ERR_INVALID_ARGUMENT;
    }
    aggs[i].col = c;
    int seen = 0;
    for (int k = 0; k < n_proj; k++) seen |= (proj[k] == c);
    if (!seen) proj[n_proj++] = c;
  }
  po_column** gcols = (po_column**)po_xcalloc((size_t)n_gb + 1, sizeof(po_column*));
  int mv_group_by = 0;   /* DefaultGroupByExecutor._hasMVGroupByExpression */
  for (int j = 0; j < n_gb; j++) {
    po_column* c = po_segment_column(seg, q->group_by_columns[j]);
    if (!c) { po_set_error("column not found: %s", q->group_by_columns[j]); return PG_ERR_NOT_FOUND; }
    if (!c->has_dictionary && c->data_type > PG_TYPE_DOUBLE && c->fwd_encoding != PG_FWD_RAW_VAR_BYTE_CHUNK) {
      po_set_error("no-dictionary group-by column %s is not a raw var-byte column", c->name);
      return PG_ERR_UNSUPPORTED;
    }
    gcols[j] = c;
    mv_group_by |= c->is_mv;
    int seen = 0;
    for (int k = 0; k < n_proj; k++) seen |= (proj[k] == c);
    if (!seen) proj[n_proj++] = c;
  }
  if (mv_group_by)
    for (int j = 0; j < n_gb; j++)
      if (!gcols[j]->has_dictionary) { po_set_error("multi-value group-by next to a no-dictionary column is outside the hot path"); return PG_ERR_UNSUPPORTED; }

  /* AggregationPlanNode: FastFilteredCountOperator (core/plan/AggregationPlanNode.java:106-108,192-196) */
  /* AggregationPlanNode.java:104-121: hasNullValues — null handling on and an aggregation argument with nulls — keeps the scanning plan */
  const int nh = (q->flags & PG_QUERY_FLAG_NULL_HANDLING) != 0;
  int has_null_values = 0;
  for (int i = 0; i < n_aggs && nh; i++) has_null_values |= col_nulls(aggs[i].col) != NULL || aggs[i].null_col != NULL;
  if (n_gb == 0 && n_aggs == 1 && aggs[0].function == PG_AGG_COUNT && !has_null_values && po_filter_can_optimize_count(filter_op)) {
    int32_t count = po_filter_num_matching_docs(filter_op);
    if (count < 0) return PG_ERR_INVALID_ARGUMENT;
    res->num_groups = 1;
    res->aggs = (po_agg_result*)po_xcalloc(1, sizeof(po_agg_result));
    res->aggs[0].kind = PG_RESULT_LONG;
    for (int k = 0; k < 2; k++) { res->aggs[0].d[k] = (double*)po_xcalloc(2, 8); res->aggs[0].l[k] = (int64_t*)po_xcalloc(2, 8); }
    res->aggs[0].l[0][0] = count;
    res->stats.num_docs_scanned = count;   /* FastFilteredCountOperator#getExecutionStatistics */
    res->stats.host_ms_total = (float)(now_ms() - t0);
    *out = res;
    return PG_OK;
  }

  /* NonScanBasedAggregationOperator (core/plan/AggregationPlanNode.java:110-120,165-190; core/operator/query/
   * NonScanBasedAggregationOperator.java:83-150,300-303): match-all filter, no GROUP BY, every aggregation is COUNT or a
   * dictionary-based function over a dictionary column → answered from the dictionaries, no doc is read */
  if (n_gb == 0 && filter_op->kind == PO_OP_MATCH_ALL && !has_null_values) {
    int fit = 1;
    for (int i = 0; i < n_aggs && fit; i++) {
      int f = sv_function_of(aggs[i].function);   /* DICTIONARY_BASED_FUNCTIONS holds MINMV / MAXMV / MINMAXRANGEMV / DISTINCTCOUNT(HLL)MV as well */
      if (aggs[i].function == PG_AGG_COUNT) continue;
      if (aggs[i].function == PG_AGG_COUNTMV || aggs[i].function == PG_AGG_SUMMV || aggs[i].function == PG_AGG_AVGMV) { fit = 0; break; }
      fit = aggs[i].col->has_dictionary && !aggs[i].col->raw_mv && (f == PG_AGG_MIN || f == PG_AGG_MAX || f == PG_AGG_MINMAXRANGE ||
                                            f == PG_AGG_DISTINCTCOUNT || f == PG_AGG_DISTINCTCOUNTHLL) &&
            (aggs[i].col->data_type <= PG_TYPE_DOUBLE || f == PG_AGG_DISTINCTCOUNT || f == PG_AGG_DISTINCTCOUNTHLL);
    }
    if (fit) {
      int32_t gid0 = 0;
      res->num_groups = 1;
      res->aggs = (po_agg_result*)po_xcalloc((size_t)n_aggs, sizeof(po_agg_result));
      for (int i = 0; i < n_aggs; i++) {
        agg_state* a = &aggs[i];
        agg_ensure_capacity(a, 1);
        po_column* c = a->col;
        switch (sv_function_of(a->function)) {
          case PG_AGG_COUNT: a->d0[0] = (double)seg->total_docs; break;
          case PG_AGG_MIN: a->d0[0] = po_dict_get_double(c, 0); break;
          case PG_AGG_MAX: a->d0[0] = po_dict_get_double(c, c->cardinality - 1); break;
          case PG_AGG_MINMAXRANGE: a->d0[0] = po_dict_get_double(c, 0); a->d1[0] = po_dict_get_double(c, c->cardinality - 1); a->has[0] = 1; break;
          default: /* DISTINCTCOUNT / DISTINCTCOUNTHLL: every dictionary value */
            a->dict_bitmaps[0] = po_bitmap_new_small(c->cardinality);
            po_bitmap_add_range(a->dict_bitmaps[0], 0, c->cardinality);
            break;
        }
        extract_agg(&res->aggs[i], a, 1, &gid0);
      }
      res->stats.num_docs_scanned = seg->total_docs;
      res->stats.host_ms_total = (float)(now_ms() - t0);
      *out = res;
      return PG_OK;
    }
  }

  /* AggregationFunctionUtils#buildAggregationInfo (:285-307): use a star-tree when the filter result is not empty and one
   * fits (StarTreeUtils#createStarTreeBasedProjectOperator); the operators then run over the star-tree's doc space. */
  /* StarTreeUtils.java:381-418: under null handling a star-tree answers only if no column the query reads holds a null in this segment */
  int star_tree_blocked = 0;
  if (q->flags & PG_QUERY_FLAG_NULL_HANDLING) {
    star_tree_blocked = null_check_filter(seg, q->filter);
    for (int i = 0; i < q->n_group_by && !star_tree_blocked; i++) star_tree_blocked = null_check_column(seg, q->group_by_columns[i]);
    for (int i = 0; i < q->n_aggregations && !star_tree_blocked; i++) star_tree_blocked = null_check_column(seg, q->aggregations[i].column);
  }
  if (!(q->flags & PG_QUERY_FLAG_SKIP_STAR_TREE) && filter_op->kind != PO_OP_EMPTY && !star_tree_blocked) {
    for (int t = 0; t < seg->n_star_trees; t++) {
      po_star_tree* st = seg->star_trees[t];
      po_filter_op* star_op = NULL;
      int fit = po_star_tree_plan(seg, st, q, &star_op);
      if (fit < 0) return fit;
      if (!fit) continue;
      res->stats.star_tree_index = t;
      filter_op = star_op;
      n_proj = 0;
      for (int i = 0; i < n_aggs; i++) {     /* StarTreeProjectPlanNode: function-column pair columns + group-by columns */
        const pg_agg_spec* s = &q->aggregations[i];
        po_column* c = st->pair_cols[po_star_tree_pair_index(st, s->function, s->column)];
        aggs[i].col = c;
        aggs[i].star = 1;
        int seen = 0;
        for (int k = 0; k < n_proj; k++) seen |= (proj[k] == c);
        if (!seen) proj[n_proj++] = c;
      }
      for (int j = 0; j < n_gb; j++) {
        po_column* c = po_segment_column(st->space, q->group_by_columns[j]);
        gcols[j] = c;
        int seen = 0;
        for (int k = 0; k < n_proj; k++) seen |= (proj[k] == c);
        if (!seen) proj[n_proj++] = c;
      }
      break;
    }
  }

  /* group key generator + holders (DefaultGroupByExecutor ctor, groupby/DefaultGroupByExecutor.java:79-140) */
  /* null handling: a null is a group key of its own (the reference generates the keys from VALUES then: DefaultGroupByExecutor.java:106-120,
   * NoDictionary*GroupKeyGenerator with a null key).  Restated over the dictIds: a nullable column takes one more id, `cardinality`, for null. */
  const po_bitmap** gnull = (const po_bitmap**)po_xcalloc((size_t)n_gb + 1, sizeof(po_bitmap*));
  po_column* gaug = (po_column*)po_xcalloc((size_t)n_gb + 1, sizeof(po_column));
  po_column** gkg_cols = (po_column**)po_xcalloc((size_t)n_gb + 1, sizeof(po_column*));
  int32_t** gbuf = (int32_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int32_t*));
  const po_bitmap** rnull = (const po_bitmap**)po_xcalloc((size_t)n_gb + 1, sizeof(po_bitmap*));   /* no-dictionary columns: a mask slot in the tuple */
  for (int j = 0; j < n_gb; j++) {
    gkg_cols[j] = gcols[j];
    if (!nh || res->stats.star_tree_index >= 0) continue;
    if (!gcols[j]->has_dictionary) { rnull[j] = col_nulls(gcols[j]); continue; }
    gnull[j] = col_nulls(gcols[j]);
    if (!gnull[j]) continue;
    gaug[j] = *gcols[j];
    gaug[j].cardinality += 1;
    gkg_cols[j] = &gaug[j];
    gbuf[j] = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);
  }
  group_key_gen gkg;
  if (n_gb > 0) {
    if (gkg_init(&gkg, n_gb, gkg_cols, num_groups_limit, max_init_cap, rnull)) return PG_ERR_UNSUPPORTED;
    int32_t max_results = gkg.global_upper_bound;
    int32_t initial = max_results < max_init_cap ? max_results : max_init_cap;
    for (int i = 0; i < n_aggs; i++) agg_ensure_capacity(&aggs[i], initial > 0 ? initial : 1);
  } else {
    for (int i = 0; i < n_aggs; i++) agg_ensure_capacity(&aggs[i], 1);
  }

  /* DocIdSetOperator + ProjectionOperator loop */
  po_docidset* set = po_filter_get_trues(filter_op);
  if (!set) return PG_ERR_INVALID_ARGUMENT;
  po_iter* it = set->iterator(set);
  int32_t* doc_ids = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);
  int32_t* group_keys = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);
  block_col* bcols = (block_col*)po_xcalloc((size_t)n_proj + 1, sizeof(block_col));
  for (int k = 0; k < n_proj; k++) {
    bcols[k].col = proj[k];
    bcols[k].dict_ids = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);
    bcols[k].doubles = (double*)po_xmalloc(sizeof(double) * PO_MAX_DOC_PER_CALL);
    if (proj[k]->is_mv) {
      bcols[k].mv_off = (int32_t*)po_xmalloc(sizeof(int32_t) * (PO_MAX_DOC_PER_CALL + 1));
      bcols[k].mv_ctx = (po_mv_ctx)PO_MV_CTX_INIT;
    }
  }
  int32_t** gmv_off = (int32_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int32_t*));
  int32_t** gmv_ids = (int32_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int32_t*));
  int32_t* mvk_off = mv_group_by ? (int32_t*)po_xmalloc(sizeof(int32_t) * (PO_MAX_DOC_PER_CALL + 1)) : NULL;
  int32_t* mvk = NULL;
  int32_t mvk_cap = 0;
  int32_t** gdict = (int32_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int32_t*));
  int32_t* cdocs = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);     /* null handling: the block's docs whose argument is not null */
  int32_t* ckeys = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);
  int32_t* cdict = (int32_t*)po_xmalloc(sizeof(int32_t) * PO_MAX_DOC_PER_CALL);
  double* cdoubles = (double*)po_xmalloc(sizeof(double) * PO_MAX_DOC_PER_CALL);
  int64_t num_docs_scanned = 0;
  int32_t cur = 0;
  while (cur != PO_EOF) {
    int pos = 0;
    for (int i = 0; i < PO_MAX_DOC_PER_CALL; i++) {  /* DocIdSetOperator.getNextBlock :74-80 */
      cur = it->next(it);
      if (cur == PO_EOF) break;
      doc_ids[pos++] = cur;
    }
    if (pos == 0) break;
    num_docs_scanned += pos;
    for (int k = 0; k < n_proj; k++) bcols[k].have_dict_ids = bcols[k].have_doubles = bcols[k].have_mv = bcols[k].have_mv_doubles = 0;
    const int32_t* keys = NULL;
    if (n_gb > 0 && mv_group_by) {   /* DefaultGroupByExecutor#process :150-158: generateKeysForBlock(valueBlock, _mvGroupKeys) */
      for (int j = 0; j < n_gb; j++)
        for (int k = 0; k < n_proj; k++) {
          if (bcols[k].col != gcols[j]) continue;
          if (gcols[j]->is_mv) { fetch_mv_dict_ids(&bcols[k], doc_ids, pos); gmv_off[j] = bcols[k].mv_off; gmv_ids[j] = bcols[k].mv_ids; }
          else { fetch_dict_ids(&bcols[k], doc_ids, pos); gdict[j] = bcols[k].dict_ids; gmv_off[j] = NULL; }
        }
      gkg_generate_mv(&gkg, pos, gdict, gmv_off, gmv_ids, mvk_off, &mvk, &mvk_cap);
      int32_t needed = gkg_upper_bound(&gkg);
      for (int i = 0; i < n_aggs; i++) agg_ensure_capacity(&aggs[i], needed);
    } else if (n_gb > 0) {
      if (gkg.holder == HOLDER_RAW_VALUES) {
        gkg_generate_raw(&gkg, pos, doc_ids, group_keys);
      } else if (gkg.holder == HOLDER_TUPLES) {
        for (int j = 0; j < n_gb; j++) {
          if (!gcols[j]->has_dictionary) continue;
          for (int k = 0; k < n_proj; k++)
            if (bcols[k].col == gcols[j]) { fetch_dict_ids(&bcols[k], doc_ids, pos); gdict[j] = bcols[k].dict_ids; }
          if (gnull[j]) {
            for (int i = 0; i < pos; i++) gbuf[j][i] = po_bitmap_contains(gnull[j], doc_ids[i]) ? gcols[j]->cardinality : gdict[j][i];
            gdict[j] = gbuf[j];
          }
        }
        gkg_generate_tuples(&gkg, pos, doc_ids, gdict, group_keys);
      } else {
        for (int j = 0; j < n_gb; j++) {
          for (int k = 0; k < n_proj; k++)
            if (bcols[k].col == gcols[j]) { fetch_dict_ids(&bcols[k], doc_ids, pos); gdict[j] = bcols[k].dict_ids; }
          if (gnull[j]) {
            for (int i = 0; i < pos; i++) gbuf[j][i] = po_bitmap_contains(gnull[j], doc_ids[i]) ? gcols[j]->cardinality : gdict[j][i];
            gdict[j] = gbuf[j];
          }
        }
        gkg_generate(&gkg, pos, gdict, group_keys);
      }
      keys = group_keys;
      int32_t needed = gkg_upper_bound(&gkg);
      for (int i = 0; i < n_aggs; i++) agg_ensure_capacity(&aggs[i], needed);
    }
    for (int i = 0; i < n_aggs; i++) {
      block_col* bc = NULL;
      for (int k = 0; k < n_proj; k++) if (bcols[k].col == aggs[i].col) bc = &bcols[k];
      /* null handling: the docs whose argument is null are skipped (NullableSingleInputAggregationFunction#forEachNotNull / foldNotNull: the
       * non-null ranges of the block in order; the per-range inner sums of SUM / AVG without GROUP BY are folded here in one pass over the
       * non-null docs — the same value whenever the partial sums are exact) */
      const po_bitmap* an = (nh && res->stats.star_tree_index < 0) ? col_nulls(aggs[i].null_col ? aggs[i].null_col : aggs[i].col) : NULL;
      if (an && !mv_group_by) {
        int cn = 0;
        for (int d = 0; d < pos; d++)
          if (!po_bitmap_contains(an, doc_ids[d])) { cdocs[cn] = doc_ids[d]; if (keys) ckeys[cn] = keys[d]; cn++; }
        if (cn == 0) continue;
        block_col tmp;
        memset(&tmp, 0, sizeof(tmp));
        if (bc) { tmp.col = bc->col; tmp.dict_ids = cdict; tmp.doubles = cdoubles; }
        agg_process_block(&aggs[i], bc ? &tmp : NULL, cdocs, cn, keys ? ckeys : NULL, NULL, NULL);
        if (keys) { for (int d = 0; d < cn; d++) if (ckeys[d] != PO_INVALID_ID) aggs[i].nn[ckeys[d]] = 1; }
        else aggs[i].nn[0] = 1;
        continue;
      }
      agg_process_block(&aggs[i], bc, doc_ids, pos, keys, mv_group_by ? mvk_off : NULL, mvk);
      if (nh && !mv_group_by) {   /* no nulls in the argument: every group a doc reached holds a value */
        if (keys) { for (int d = 0; d < pos; d++) if (keys[d] != PO_INVALID_ID) aggs[i].nn[keys[d]] = 1; }
        else aggs[i].nn[0] = 1;
      }
    }
  }

  /* results */
  int32_t n_groups;
  int32_t* gid_of;
  if (n_gb == 0) {
    n_groups = 1;
    gid_of = (int32_t*)po_xcalloc(1, 4);
  } else if (gkg.holder == HOLDER_ARRAY) {
    n_groups = gkg.num_keys;
    gid_of = (int32_t*)po_xcalloc((size_t)n_groups + 1, 4);
    int32_t k = 0;
    for (int32_t g = 0; g < gkg.global_upper_bound; g++) if (gkg.flags[g]) gid_of[k++] = g;
  } else {
    n_groups = gkg_num_keys(&gkg);
    gid_of = (int32_t*)po_xcalloc((size_t)n_groups + 1, 4);
    for (int32_t g = 0; g < n_groups; g++) gid_of[g] = g;
  }
  /* GroupByOperator.java:120-133: ORDER BY + minSegmentGroupTrimSize > 0 + more groups than trimSize -> keep the trimSize groups that sort first */
  if (n_gb > 0 && q->n_order_by > 0 && q->order_by && q->min_segment_group_trim_size > 0) {
    const int64_t by_limit = (int64_t)q->limit * 5;   /* GroupByUtils.getTableCapacity */
    const int32_t trim_size = by_limit > INT32_MAX ? INT32_MAX : ((int32_t)by_limit > q->min_segment_group_trim_size ? (int32_t)by_limit : q->min_segment_group_trim_size);
    if (n_groups > trim_size) {
      const int n_ob = q->n_order_by;
      order_value* vals = (order_value*)po_xcalloc((size_t)n_groups * (size_t)n_ob + 1, sizeof(order_value));
      int* asc = (int*)po_xcalloc((size_t)n_ob + 1, sizeof(int));
      int* nulls_last = (int*)po_xcalloc((size_t)n_ob + 1, sizeof(int));
      const int nh_trim = nh && res->stats.star_tree_index < 0;   /* null order-by values: a null group key, SUM / MIN / MAX / AVG / MINMAXRANGE over no value */
      for (int k = 0; k < n_ob; k++) {
        const pg_order_by* ob = &q->order_by[k];
        asc[k] = ob->ascending != 0;
        nulls_last[k] = ob->nulls_last != 0;
        if (ob->kind == PG_ORDER_BY_AGGREGATION) {
          if (ob->index < 0 || ob->index >= n_aggs) { free(vals); free(asc); free(nulls_last); free(gid_of); po_set_error("ORDER BY aggregation %d of %d", ob->index, n_aggs); return PG_ERR_INVALID_ARGUMENT; }
          agg_ensure_capacity(&aggs[ob->index], (gkg.holder == HOLDER_ARRAY ? gkg.global_upper_bound : n_groups) + 1);
          for (int32_t i = 0; i < n_groups; i++) vals[(size_t)i * (size_t)n_ob + (size_t)k] = agg_final_value(&aggs[ob->index], gid_of[i]);
          const int fo = sv_function_of(aggs[ob->index].function);
          if (nh_trim && !mv_group_by && (fo == PG_AGG_SUM || fo == PG_AGG_MIN || fo == PG_AGG_MAX || fo == PG_AGG_AVG || fo == PG_AGG_MINMAXRANGE))
            for (int32_t i = 0; i < n_groups; i++) vals[(size_t)i * (size_t)n_ob + (size_t)k].is_null = !aggs[ob->index].nn[gid_of[i]];
          continue;
        }
        if (ob->index < 0 || ob->index >= n_gb) { free(vals); free(asc); free(nulls_last); free(gid_of); po_set_error("ORDER BY group-by expression %d of %d", ob->index, n_gb); return PG_ERR_INVALID_ARGUMENT; }
        const int j = ob->index;
        const po_column* c = gcols[j];
        for (int32_t i = 0; i < n_groups; i++) {
          order_value* v = &vals[(size_t)i * (size_t)n_ob + (size_t)k];
          const int32_t g = gid_of[i];
          if (gkg.holder == HOLDER_RAW_VALUES) { v->type = 0; v->l = gkg.raw_key_of_group[g]; continue; }
          if (gkg.holder == HOLDER_TUPLES) {
            const int64_t t = gkg.tuples[(size_t)g * (size_t)gkg.tuple_w + (size_t)j];
            if (gnull[j] && t == c->cardinality) v->is_null = 1;   /* the id one past the dictionary */
            if (rnull[j] && gkg.tuple_w > n_gb && ((gkg.tuples[(size_t)g * (size_t)gkg.tuple_w + (size_t)n_gb] >> j) & 1)) v->is_null = 1;   /* the mask slot */
            if (c->has_dictionary || c->data_type <= PG_TYPE_LONG) { v->type = 0; v->l = t; }   /* dictIds order as the values do (sorted dictionaries) */
            else if (c->data_type == PG_TYPE_FLOAT) { uint32_t b = (uint32_t)t; float f; memcpy(&f, &b, 4); v->type = 1; v->d = (double)f; }
            else if (c->data_type == PG_TYPE_DOUBLE) { v->type = 1; memcpy(&v->d, &t, 8); }
            else { v->type = c->data_type == PG_TYPE_STRING ? 3 : 2; v->b = gkg.bytes_dicts[j].vals[t]; v->blen = gkg.bytes_dicts[j].lens[t]; }
            continue;
          }
          int64_t raw = (gkg.holder == HOLDER_ARRAY) ? g : gkg.raw_key_of_group[g];   /* getKeys :578-591: column 0 is least significant */
          for (int jj = 0; jj < j; jj++) raw /= gkg.cardinalities[jj];
          v->type = 0;
          v->l = raw % gkg.cardinalities[j];
          if (gnull[j] && v->l == c->cardinality) v->is_null = 1;
        }
      }
      int32_t* order = (int32_t*)po_xcalloc((size_t)n_groups + 1, 4);
      for (int32_t i = 0; i < n_groups; i++) order[i] = i;
      order_ctx octx = {vals, n_ob, asc, nh_trim ? nulls_last : NULL};
      qsort_r(order, (size_t)n_groups, 4, order_cmp, &octx);
      /* the survivors in group-id order again (the order of the result's rows carries no meaning) */
      uint8_t* keep = (uint8_t*)po_xcalloc((size_t)n_groups + 1, 1);
      for (int32_t i = 0; i < trim_size; i++) keep[order[i]] = 1;
      int32_t k2 = 0;
      for (int32_t i = 0; i < n_groups; i++) if (keep[i]) gid_of[k2++] = gid_of[i];
      n_groups = k2;
      free(keep); free(order); free(vals); free(asc); free(nulls_last);
    }
  }
  res->num_groups = n_groups;
  res->group_dict_ids = (int32_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int32_t*));
  for (int j = 0; j < n_gb; j++) res->group_dict_ids[j] = (int32_t*)po_xcalloc((size_t)n_groups + 1, 4);
  if (n_gb > 0 && gkg.holder == HOLDER_RAW_VALUES) {
    res->group_values = (int64_t*)po_xcalloc((size_t)n_groups + 1, 8);
    for (int32_t i = 0; i < n_groups; i++) res->group_values[i] = gkg.raw_key_of_group[gid_of[i]];
  }
  if (n_gb > 0 && gkg.holder == HOLDER_TUPLES) {
    res->key_types = (int32_t*)po_xcalloc((size_t)n_gb + 1, 4);
    res->key_values = (int64_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int64_t*));
    for (int j = 0; j < n_gb; j++) {
      const po_column* c = gcols[j];
      if (c->has_dictionary) {
        res->key_types[j] = PG_GROUP_KEY_DICT_IDS;
        for (int32_t i = 0; i < n_groups; i++) res->group_dict_ids[j][i] = (int32_t)gkg.tuples[(size_t)gid_of[i] * (size_t)gkg.tuple_w + (size_t)j];
        continue;
      }
      if (c->data_type > PG_TYPE_DOUBLE) {   /* the groups' byte strings, back to back */
        res->key_types[j] = PG_GROUP_KEY_BYTES_VALUES;
        if (!res->key_bytes) {
          res->key_bytes = (uint8_t**)po_xcalloc((size_t)n_gb + 1, sizeof(uint8_t*));
          res->key_bytes_off = (int64_t**)po_xcalloc((size_t)n_gb + 1, sizeof(int64_t*));
        }
        const bytes_dict* bd = &gkg.bytes_dicts[j];
        int64_t total = 0;
        res->key_bytes_off[j] = (int64_t*)po_xcalloc((size_t)n_groups + 2, 8);
        for (int32_t i = 0; i < n_groups; i++) {
          res->key_bytes_off[j][i] = total;
          total += bd->lens[gkg.tuples[(size_t)gid_of[i] * (size_t)gkg.tuple_w + (size_t)j]];
        }
        res->key_bytes_off[j][n_groups] = total;
        res->key_bytes[j] = (uint8_t*)po_xmalloc((size_t)total + 1);
        for (int32_t i = 0; i < n_groups; i++) {
          const int64_t id = gkg.tuples[(size_t)gid_of[i] * (size_t)gkg.tuple_w + (size_t)j];
          memcpy(res->key_bytes[j] + res->key_bytes_off[j][i], bd->vals[id], (size_t)bd->lens[id]);
        }
        continue;
      }
      res->key_types[j] = c->data_type <= PG_TYPE_LONG ? PG_GROUP_KEY_LONG_VALUES : PG_GROUP_KEY_DOUBLE_VALUES;
      res->key_values[j] = (int64_t*)po_xcalloc((size_t)n_groups + 1, 8);
      for (int32_t i = 0; i < n_groups; i++) {
        int64_t k = gkg.tuples[(size_t)gid_of[i] * (size_t)gkg.tuple_w + (size_t)j];
        if (c->data_type == PG_TYPE_FLOAT) { uint32_t b = (uint32_t)k; float f; memcpy(&f, &b, 4); double d = (double)f; memcpy(&k, &d, 8); }
        res->key_values[j][i] = k;
      }
    }
  }
  for (int32_t i = 0; i < n_groups && n_gb > 0 && gkg.holder != HOLDER_RAW_VALUES && gkg.holder != HOLDER_TUPLES; i++) {  /* getKeys :578-591: col 0 is least significant */
    int64_t raw = (gkg.holder == HOLDER_ARRAY) ? gid_of[i] : gkg.raw_key_of_group[gid_of[i]];
    for (int j = 0; j < n_gb; j++) {
      res->group_dict_ids[j][i] = (int32_t)(raw % gkg.cardinalities[j]);
      raw /= gkg.cardinalities[j];
    }
  }
  for (int j = 0; j < n_gb && n_gb > 0 && gkg.holder == HOLDER_TUPLES && gkg.tuple_w > n_gb; j++) {   /* the mask slot: null raw keys */
    if (!rnull[j]) continue;
    if (!res->key_nulls) res->key_nulls = (uint8_t**)po_xcalloc((size_t)n_gb + 1, sizeof(uint8_t*));
    res->key_nulls[j] = (uint8_t*)po_xcalloc((size_t)n_groups + 1, 1);
    for (int32_t i = 0; i < n_groups; i++)
      res->key_nulls[j][i] = (uint8_t)((gkg.tuples[(size_t)gid_of[i] * (size_t)gkg.tuple_w + (size_t)n_gb] >> j) & 1);
  }
  for (int j = 0; j < n_gb; j++) {   /* the id one past the dictionary is the null key */
    if (!gnull[j]) continue;
    if (!res->key_nulls) res->key_nulls = (uint8_t**)po_xcalloc((size_t)n_gb + 1, sizeof(uint8_t*));
    res->key_nulls[j] = (uint8_t*)po_xcalloc((size_t)n_groups + 1, 1);
    for (int32_t i = 0; i < n_groups; i++)
      if (res->group_dict_ids[j][i] == gcols[j]->cardinality) { res->key_nulls[j][i] = 1; res->group_dict_ids[j][i] = 0; }
  }
  res->aggs = (po_agg_result*)po_xcalloc((size_t)n_aggs, sizeof(po_agg_result));
  for (int i = 0; i < n_aggs; i++) {
    if (n_gb > 0) agg_ensure_capacity(&aggs[i], (gkg.holder == HOLDER_ARRAY ? gkg.global_upper_bound : n_groups) + 1);
    extract_agg(&res->aggs[i], &aggs[i], n_groups, gid_of);
    /* null handling: SUM / MIN / MAX / AVG / MINMAXRANGE of no value are null (the holders stay null: SumAggregationFunction.java:100-131,
     * 180-215; COUNT is 0, the distinct counts an empty set) */
    const int f = sv_function_of(aggs[i].function);
    if (nh && res->stats.star_tree_index < 0 && !mv_group_by && (f == PG_AGG_SUM || f == PG_AGG_MIN || f == PG_AGG_MAX || f == PG_AGG_AVG || f == PG_AGG_MINMAXRANGE)) {
      res->aggs[i].nulls = (uint8_t*)po_xcalloc((size_t)n_groups + 1, 1);
      for (int32_t g = 0; g < n_groups; g++)
        if (!aggs[i].nn[gid_of[g]]) {
          res->aggs[i].nulls[g] = 1;
          for (int k = 0; k < 2; k++) { res->aggs[i].d[k][g] = 0; res->aggs[i].l[k][g] = 0; }
        }
    }
  }
  res->stats.num_docs_scanned = num_docs_scanned;
  res->stats.num_entries_scanned_in_filter = set->num_entries_scanned(set);
  res->stats.num_entries_scanned_post_filter = num_docs_scanned * n_proj;
  if (n_gb > 0) res->stats.num_groups_limit_reached = gkg_num_keys(&gkg) >= num_groups_limit;
  res->stats.host_ms_total = (float)(now_ms() - t0);
  *out = res;
  return PG_OK;
}

/* ---- accessors ------------------------------------------------------------------------------------------------------------- */
#define RES(r) ((po_result_impl*)(r))
static int bad_agg(void* r, int32_t agg) {
  if (agg < 0 || agg >= RES(r)->n_aggs) { po_set_error("aggregation index out of range"); return 1; }
  return 0;
}
int32_t po_result_num_groups(void* r, int32_t* out) { *out = RES(r)->num_groups; return PG_OK; }
int32_t po_result_group_dict_ids(void* r, int32_t col, int32_t* out, int32_t cap) {
  if (col < 0 || col >= RES(r)->n_group_cols || cap < RES(r)->num_groups) { po_set_error("bad column/capacity"); return PG_ERR_INVALID_ARGUMENT; }
  memcpy(out, RES(r)->group_dict_ids[col], sizeof(int32_t) * (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_group_key_type(void* r, int32_t col, int32_t* out) {
  if (col < 0 || col >= RES(r)->n_group_cols) { po_set_error("group-by column index out of range"); return PG_ERR_INVALID_ARGUMENT; }
  if (RES(r)->key_types) *out = RES(r)->key_types[col];
  else *out = RES(r)->group_values ? PG_GROUP_KEY_LONG_VALUES : PG_GROUP_KEY_DICT_IDS;
  return PG_OK;
}
int32_t po_result_group_values_long(void* r, int32_t col, int64_t* out, int32_t cap) {
  if (col < 0 || col >= RES(r)->n_group_cols || cap < RES(r)->num_groups) { po_set_error("bad column/capacity"); return PG_ERR_INVALID_ARGUMENT; }
  const int64_t* v = NULL;
  if (RES(r)->key_types) v = RES(r)->key_types[col] == PG_GROUP_KEY_LONG_VALUES ? RES(r)->key_values[col] : NULL;
  else if (col == 0) v = RES(r)->group_values;
  if (!v) { po_set_error("group-by column does not have LONG values"); return PG_ERR_INVALID_ARGUMENT; }
  memcpy(out, v, sizeof(int64_t) * (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_group_values_bytes_size(void* r, int32_t col, uint64_t* out) {
  if (col < 0 || col >= RES(r)->n_group_cols || !RES(r)->key_types || RES(r)->key_types[col] != PG_GROUP_KEY_BYTES_VALUES) {
    po_set_error("group-by column does not have BYTES values");
    return PG_ERR_INVALID_ARGUMENT;
  }
  *out = (uint64_t)RES(r)->key_bytes_off[col][RES(r)->num_groups];
  return PG_OK;
}
int32_t po_result_group_values_bytes(void* r, int32_t col, int64_t* out_off, int32_t off_cap, uint8_t* out_bytes, uint64_t cap) {
  uint64_t total = 0;
  if (po_result_group_values_bytes_size(r, col, &total)) return PG_ERR_INVALID_ARGUMENT;
  if (off_cap < RES(r)->num_groups + 1 || cap < total) { po_set_error("capacity too small"); return PG_ERR_INVALID_ARGUMENT; }
  memcpy(out_off, RES(r)->key_bytes_off[col], sizeof(int64_t) * ((size_t)RES(r)->num_groups + 1));
  memcpy(out_bytes, RES(r)->key_bytes[col], (size_t)total);
  return PG_OK;
}
int32_t po_result_group_values_double(void* r, int32_t col, double* out, int32_t cap) {
  if (col < 0 || col >= RES(r)->n_group_cols || cap < RES(r)->num_groups) { po_set_error("bad column/capacity"); return PG_ERR_INVALID_ARGUMENT; }
  if (!RES(r)->key_types || RES(r)->key_types[col] != PG_GROUP_KEY_DOUBLE_VALUES) { po_set_error("group-by column does not have DOUBLE values"); return PG_ERR_INVALID_ARGUMENT; }
  memcpy(out, RES(r)->key_values[col], sizeof(double) * (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_kind_of(void* r, int32_t agg, int32_t* out) {
  if (bad_agg(r, agg)) return PG_ERR_INVALID_ARGUMENT;
  *out = RES(r)->aggs[agg].kind;
  return PG_OK;
}
int32_t po_result_doubles(void* r, int32_t agg, int32_t comp, double* out, int32_t cap) {
  if (bad_agg(r, agg) || comp < 0 || comp > 1 || cap < RES(r)->num_groups) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].d[comp], sizeof(double) * (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_longs(void* r, int32_t agg, int32_t comp, int64_t* out, int32_t cap) {
  if (bad_agg(r, agg) || comp < 0 || comp > 1 || cap < RES(r)->num_groups) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].l[comp], sizeof(int64_t) * (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_set_values_long(void* r, int32_t agg, int64_t* out, int64_t cap) {
  if (bad_agg(r, agg) || RES(r)->aggs[agg].kind != PG_RESULT_VALUE_SET || RES(r)->aggs[agg].set_type > PG_TYPE_LONG || cap < RES(r)->aggs[agg].set_total) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].set_l, 8 * (size_t)RES(r)->aggs[agg].set_total);
  return PG_OK;
}
int32_t po_result_set_values_double(void* r, int32_t agg, double* out, int64_t cap) {
  if (bad_agg(r, agg) || RES(r)->aggs[agg].kind != PG_RESULT_VALUE_SET || RES(r)->aggs[agg].set_type <= PG_TYPE_LONG || cap < RES(r)->aggs[agg].set_total) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].set_d, 8 * (size_t)RES(r)->aggs[agg].set_total);
  return PG_OK;
}
int32_t po_result_set_sizes(void* r, int32_t agg, int32_t* out, int32_t cap) {
  if (bad_agg(r, agg) || (RES(r)->aggs[agg].kind != PG_RESULT_DICTID_SET && RES(r)->aggs[agg].kind != PG_RESULT_VALUE_SET) || cap < RES(r)->num_groups) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].set_sizes, sizeof(int32_t) * (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_set_dict_ids(void* r, int32_t agg, int32_t* out, int64_t cap) {
  if (bad_agg(r, agg) || RES(r)->aggs[agg].kind != PG_RESULT_DICTID_SET || cap < RES(r)->aggs[agg].set_total) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].set_ids, sizeof(int32_t) * (size_t)RES(r)->aggs[agg].set_total);
  return PG_OK;
}
int32_t po_result_hll_registers(void* r, int32_t agg, uint8_t* out, int64_t cap) {
  if (bad_agg(r, agg) || RES(r)->aggs[agg].kind != PG_RESULT_HLL) return PG_ERR_INVALID_ARGUMENT;
  int64_t n = (int64_t)RES(r)->num_groups << RES(r)->aggs[agg].log2m;
  if (cap < n) return PG_ERR_INVALID_ARGUMENT;
  memcpy(out, RES(r)->aggs[agg].hll, (size_t)n);
  return PG_OK;
}
int32_t po_result_agg_nulls(void* r, int32_t agg, uint8_t* out, int32_t cap) {
  if (bad_agg(r, agg) || cap < RES(r)->num_groups) { po_set_error("bad aggregation/capacity"); return PG_ERR_INVALID_ARGUMENT; }
  if (RES(r)->aggs[agg].nulls) memcpy(out, RES(r)->aggs[agg].nulls, (size_t)RES(r)->num_groups);
  else memset(out, 0, (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_group_key_nulls(void* r, int32_t col, uint8_t* out, int32_t cap) {
  if (col < 0 || col >= RES(r)->n_group_cols || cap < RES(r)->num_groups) { po_set_error("bad column/capacity"); return PG_ERR_INVALID_ARGUMENT; }
  if (RES(r)->key_nulls && RES(r)->key_nulls[col]) memcpy(out, RES(r)->key_nulls[col], (size_t)RES(r)->num_groups);
  else memset(out, 0, (size_t)RES(r)->num_groups);
  return PG_OK;
}
int32_t po_result_stats(void* r, pg_exec_stats* out) { *out = RES(r)->stats; return PG_OK; }
int32_t po_result_free(void* r) {
  po_result_impl* res = RES(r);
  for (int i = 0; i < res->n_aggs && res->aggs; i++) {
    for (int k = 0; k < 2; k++) { free(res->aggs[i].d[k]); free(res->aggs[i].l[k]); }
    free(res->aggs[i].set_sizes); free(res->aggs[i].set_ids); free(res->aggs[i].set_l); free(res->aggs[i].set_d); free(res->aggs[i].hll); free(res->aggs[i].nulls);
  }
  free(res->aggs);
  for (int j = 0; j < res->n_group_cols && res->group_dict_ids; j++) free(res->group_dict_ids[j]);
  free(res->group_dict_ids);
  for (int j = 0; j < res->n_group_cols && res->key_nulls; j++) free(res->key_nulls[j]);
  free(res->key_nulls);
  free(res);
  return PG_OK;
}

/* ---- small helpers exported for the golden tests ---------------------------------------------------------------------------- */
int64_t po_hll_cardinality_from_registers(const uint8_t* regs, int32_t log2m) {
  po_hll h;
  h.log2m = log2m;
  h.m = 1 << log2m;
  h.regs = (uint8_t*)regs;
  return po_hll_cardinality(&h);
}
void po_hll_registers_for_values(const int64_t* values, int64_t n, int32_t as_int, int32_t log2m, uint8_t* out_regs) {
  po_hll* h = po_hll_new(log2m);
  for (int64_t i = 0; i < n; i++) {
    if (as_int) po_hll_offer_int(h, (int32_t)values[i]); else po_hll_offer_long(h, values[i]);
  }
  memcpy(out_regs, h->regs, (size_t)h->m);
  po_hll_free(h);
}
int32_t po_read_fixed_bit(const uint8_t* buf, int32_t bits, int32_t index) {
  po_column c;
  memset(&c, 0, sizeof(c));
  c.fwd = buf;
  c.bits_per_value = bits;
  return po_fixedbit_read(&c, index);
}
void po_read_fixed_bit_block(const uint8_t* buf, int32_t bits, const int32_t* doc_ids, int32_t n, int32_t* out) {
  po_column c;
  memset(&c, 0, sizeof(c));
  c.fwd = buf;
  c.bits_per_value = bits;
  c.fwd_encoding = PG_FWD_DICT_FIXED_BIT;
  po_fwd_read_dict_ids(&c, doc_ids, n, out);
}

/* test hook: VarByteChunkSVForwardIndexReader#getBytes over a raw var-byte chunk buffer (PASS_THROUGH); returns the length */
int32_t po_read_var_bytes(const uint8_t* buf, uint64_t len, int32_t doc_id, uint8_t* out, int32_t cap) {
  po_column c;
  memset(&c, 0, sizeof(c));
  c.name = (char*)"raw";
  c.fwd = buf;
  c.fwd_len = len;
  if (po_raw_parse_header(&c)) return -1;
  int32_t n = 0;
  const uint8_t* v = po_raw_get_bytes(&c, doc_id, &n);
  if (n <= cap) memcpy(out, v, (size_t)n);
  free(c.raw_owned);
  return n > cap ? -2 : n;
}
