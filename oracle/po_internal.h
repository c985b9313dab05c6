/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm for the hot path, written so that each function
 * follows one reference function line by line (cited at each definition; paths relative to /root/reference).
 * The reference is Java and cannot be compiled or run in this environment (no JVM), so parity is pinned by the
 * reference's own golden numbers instead: tests/test_oracle_goldens.py reproduces
 *   - InnerSegmentAggregationSingleValueQueriesTest (results AND ExecutionStatistics, incl. numEntriesScannedInFilter),
 *   - InterSegmentGroupBySingleValueQueriesTest / InterSegmentAggregationSingleValueQueriesTest (incl. HLL goldens),
 *   - FastFilteredCountTest, RangeQueriesTest formulaic fixtures, FixedByteChunkSVForwardIndexTest legacy blob.
 * Third-party arithmetic absent from the tree (RoaringBitmap 1.3.0 serialized format, stream-lib 2.9.8 HyperLogLog,
 * fastutil HashCommon.mix) is restated from the published algorithms; see the headers of po_bitmap.c / po_hll.c.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product path
 * (pinot_amd/, libpinot_gpu.so) never links, imports or calls it.
 */
#ifndef PO_INTERNAL_H_
#define PO_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pinot_gpu.h"

#define PO_EOF INT32_MIN                 /* Constants.EOF, pinot-segment-spi/.../Constants.java:25 */
#define PO_MAX_DOC_PER_CALL 10000        /* DocIdSetPlanNode.MAX_DOC_PER_CALL, core/plan/DocIdSetPlanNode.java:29 */
#define PO_SCAN_BATCH 256                /* BlockDocIdIterator.OPTIMAL_ITERATOR_BATCH_SIZE, core/common/BlockDocIdIterator.java:49 */
#define PO_INVALID_ID (-1)               /* GroupKeyGenerator.INVALID_ID */

void po_set_error(const char* fmt, ...);
void* po_xmalloc(size_t n);
void* po_xcalloc(size_t n, size_t sz);
void* po_xrealloc(void* p, size_t n);

/* ---- big-endian loads (PinotDataBuffer with ByteOrder.BIG_ENDIAN) -------------------------------------------------- */
static inline uint32_t po_be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
static inline uint64_t po_be64(const uint8_t* p) { return ((uint64_t)po_be32(p) << 32) | po_be32(p + 4); }
static inline uint16_t po_be16(const uint8_t* p) { return (uint16_t)(((uint16_t)p[0] << 8) | p[1]); }
static inline float po_bef32(const uint8_t* p) { uint32_t u = po_be32(p); float f; memcpy(&f, &u, 4); return f; }
static inline double po_bef64(const uint8_t* p) { uint64_t u = po_be64(p); double d; memcpy(&d, &u, 8); return d; }

/* ---- bitmap (stands in for Immutable/MutableRoaringBitmap; set semantics only) -------------------------------------- */
typedef struct po_bitmap {
  uint64_t* words;
  int64_t n_words;
  int64_t universe;   /* number of addressable bits (>= numDocs) */
  /* po_bitmap_new_small over a large universe starts as a sorted array (a RoaringBitmap array container, in effect): the dictIds
   * one group has seen are few, and a dense bitset per group over a 1 M-value dictionary would be 125 KB each.  Only add /
   * add_range / next_set / cardinality / free understand this form; it turns dense once it holds universe / 64 values. */
  int32_t* sparse;
  int32_t n_sparse, cap_sparse;
} po_bitmap;

po_bitmap* po_bitmap_new(int64_t universe);
po_bitmap* po_bitmap_new_small(int64_t universe);
po_bitmap* po_bitmap_clone(const po_bitmap* b);
void po_bitmap_free(po_bitmap* b);
void po_bitmap_add(po_bitmap* b, int32_t x);
void po_bitmap_add_range(po_bitmap* b, int64_t start, int64_t end_exclusive);
void po_bitmap_or(po_bitmap* dst, const po_bitmap* src);
void po_bitmap_and(po_bitmap* dst, const po_bitmap* src);
void po_bitmap_flip(po_bitmap* b, int64_t start, int64_t end_exclusive);
int64_t po_bitmap_cardinality(const po_bitmap* b);
int po_bitmap_contains(const po_bitmap* b, int32_t x);
/* first set bit >= from, or -1 */
int64_t po_bitmap_next_set(const po_bitmap* b, int64_t from);
/* ImmutableRoaringBitmap(ByteBuffer): parses the portable serialization into `dst` (OR-ing into it) */
int64_t po_snappy_uncompress(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap);
int64_t po_lz4_decompress(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap);
int64_t po_chunk_decompress(int32_t compression, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap);
int po_roaring_deserialize_or(const uint8_t* blob, uint64_t len, po_bitmap* dst);
int po_roaring_container_stats(const uint8_t* blob, uint64_t len, int* n_array, int* n_bitmap, int* n_run);

/* ---- column = DataSource -------------------------------------------------------------------------------------------- */
typedef struct po_column {
  char* name;
  int32_t data_type;
  int32_t fwd_encoding;
  int32_t has_dictionary;
  int32_t cardinality;
  int32_t bits_per_value;
  int32_t is_sorted;
  int32_t dict_bytes_per_value;
  const uint8_t* fwd;     uint64_t fwd_len;
  const uint8_t* dict;    uint64_t dict_len;
  const uint8_t* inv;     uint64_t inv_len;
  /* BaseChunkForwardIndexReader fields */
  int32_t raw_version, raw_num_chunks, raw_docs_per_chunk, raw_entry_len, raw_compression;
  const uint8_t* raw_data;   /* _rawData */
  uint8_t* raw_owned;        /* compressed chunks, decompressed once into the PASS_THROUGH layout (po_raw_parse_header) */
  int32_t num_docs;
  struct po_bitmap* null_bitmap;   /* NullValueVectorReader#getNullBitmap, NULL if the column has no null value vector */
  const uint8_t* range_idx; uint64_t range_len;   /* DataSource#getRangeIndex: BitSlicedRangeIndexReader bytes, NULL if none */
  /* FixedBitMVForwardIndexReader fields (PG_FWD_DICT_FIXED_BIT_MV): _numValues, _numDocsPerChunk, the three views of the buffer,
   * and ColumnMetadata#getMaxNumberOfMultiValues (found by one walk over the row-start bitmap when the column is added) */
  int32_t raw_mv;            /* the column came as a raw multi-value forward index (FixedByteChunkMVForwardIndexReader): see po_raw_mv_attach */
  uint8_t* mv_owned_fwd; uint8_t* mv_owned_dict;
  int32_t is_mv, total_entries, mv_docs_per_chunk, mv_max_values;
  const uint8_t* mv_chunk_offsets; const uint8_t* mv_bitmap; const uint8_t* mv_raw;
} po_column;

struct po_star_tree;
typedef struct po_segment {
  char* name;
  int32_t total_docs;
  int32_t n_columns;
  po_column** columns;
  int32_t n_star_trees;                 /* IndexSegment#getStarTrees */
  struct po_star_tree** star_trees;
  struct po_bitmap* queryable_doc_ids;  /* SegmentContext#getQueryableDocIdsSnapshot, NULL if none */
} po_segment;

/* StarTreeV2 (pinot-segment-local/.../startree/v2/store/StarTreeLoaderUtils.java:53-128): the tree, its metadata and
 * one DataSource per dimension / function-column pair over the star-tree doc space. */
typedef struct po_star_tree {
  po_segment* space;                    /* star-tree docs: dimension columns (parent dictionaries) + pair columns */
  int32_t num_docs;
  int32_t n_dims;
  char** dims;                          /* dimensionsSplitOrder == OffHeapStarTree#getDimensionNames */
  int32_t n_pairs;
  int32_t* pair_functions;              /* pg_agg_function */
  char** pair_columns;                  /* "*" for COUNT */
  po_column** pair_cols;                /* columns of `space` named like AggregationFunctionColumnPair#toColumnName */
  const uint8_t* nodes;                 /* little-endian node array, 7 ints per node */
  int32_t n_nodes;
} po_star_tree;

po_column* po_segment_column(po_segment* seg, const char* name);

/* FixedBitIntReader / FixedBitSVForwardIndexReaderV2 */
int32_t po_fixedbit_read(const po_column* c, int32_t index);
void po_fwd_read_dict_ids(const po_column* c, const int32_t* doc_ids, int32_t length, int32_t* out);
/* FixedBitMVForwardIndexReader: Context (_docId = -1, _endOffset = 0) + getDictIdMV(docId, dictIdBuffer, context) → number of values */
typedef struct po_mv_ctx { int32_t doc_id, end_offset; } po_mv_ctx;
#define PO_MV_CTX_INIT {-1, 0}
int po_mv_parse(po_column* c);
int po_raw_mv_attach(po_column* c);
int po_raw_mv_attach_strings(po_column* c);
int po_mv_entry_dict_attach(po_column* c);
int32_t po_mv_get_dict_ids(const po_column* c, int32_t doc_id, int32_t* buf, po_mv_ctx* ctx);
/* FixedByteChunkSVForwardIndexReader#getInt/getLong/getFloat/getDouble (PASS_THROUGH) */
int32_t po_raw_get_int(const po_column* c, int32_t doc_id);
int64_t po_raw_get_long(const po_column* c, int32_t doc_id);
float po_raw_get_float(const po_column* c, int32_t doc_id);
double po_raw_get_double(const po_column* c, int32_t doc_id);
/* VarByteChunkSVForwardIndexReader#getBytes (PASS_THROUGH): pointer into the index buffer + length */
const uint8_t* po_raw_get_bytes(const po_column* c, int32_t doc_id, int32_t* len);
/* SortedIndexReaderImpl#getDocIds */
void po_sorted_get_doc_ids(const po_column* c, int32_t dict_id, int32_t* start, int32_t* end_inclusive);
int32_t po_sorted_get_dict_id(const po_column* c, int32_t doc_id);
/* Dictionary */
int32_t po_dict_get_int(const po_column* c, int32_t dict_id);
int64_t po_dict_get_long(const po_column* c, int32_t dict_id);
float po_dict_get_float(const po_column* c, int32_t dict_id);
double po_dict_get_double(const po_column* c, int32_t dict_id);   /* Dictionary#getDoubleValue */
/* returns insertion index (>=0 found, else -(insertionPoint+1)); parse error => returns INT32_MIN and sets error */
int32_t po_dict_insertion_index_of(const po_column* c, const char* string_value);
/* BitmapInvertedIndexReader#getDocIds → OR into dst */
int po_inv_get_doc_ids_or(const po_column* c, int32_t dict_id, po_bitmap* dst);

/* ---- predicate evaluators -------------------------------------------------------------------------------------------- */
typedef struct po_pred_eval {
  int32_t pred_type;        /* pg_predicate_type */
  int dictionary_based;
  int always_true, always_false;
  int exclusive;            /* NOT_EQ / NOT_IN */
  /* dictionary based */
  int is_range;             /* SortedDictionaryBasedRangePredicateEvaluator */
  int32_t start_dict_id, end_dict_id;      /* [start, end) */
  int32_t n_matching; int32_t* matching_dict_ids;       /* sorted ascending; for exclusive these are the NON-matching ids' complement source, see po_predicate.c */
  int32_t n_non_matching; int32_t* non_matching_dict_ids;
  uint8_t* dict_id_match;   /* cardinality flags: applySV(dictId) */
  /* raw value based */
  int32_t data_type;
  int64_t lo_i, hi_i; double lo_d, hi_d; float lo_f, hi_f;   /* inclusive bounds */
  int32_t n_raw_values; int64_t* raw_i; double* raw_d;        /* EQ / IN value sets (sorted) */
  int32_t num_matching_items;   /* getNumMatchingItems */
  /* raw STRING column: the literals (EQ / NOT_EQ / IN / NOT_IN) or the bounds (RANGE), UTF-8 as the caller gave them */
  int32_t n_raw_s; char** raw_s; char* lo_s; char* hi_s; int lo_inc, hi_inc;   /* lo_s / hi_s NULL: unbounded */
} po_pred_eval;

po_pred_eval* po_pred_eval_create(const pg_filter_node* pred, const po_column* col);
void po_pred_eval_free(po_pred_eval* e);
int po_pred_apply_dict(const po_pred_eval* e, int32_t dict_id);
int po_pred_apply_int(const po_pred_eval* e, int32_t v);
int po_pred_apply_long(const po_pred_eval* e, int64_t v);
int po_pred_apply_float(const po_pred_eval* e, float v);
int po_pred_apply_double(const po_pred_eval* e, double v);
int po_pred_apply_string(const po_pred_eval* e, const uint8_t* v, int32_t len);   /* a raw STRING column's value (UTF-8 bytes) */
int po_utf16_unit_order(const uint8_t* a, int32_t alen, const uint8_t* b, int32_t blen);   /* String.compareTo over UTF-8 bytes */

/* ---- filter operators / docIdSets / iterators ----------------------------------------------------------------------- */
typedef struct po_iter po_iter;
typedef struct po_docidset po_docidset;
typedef struct po_filter_op po_filter_op;

enum { PO_IT_SCAN, PO_IT_BITMAP, PO_IT_RANGELESS_BITMAP, PO_IT_SORTED, PO_IT_AND, PO_IT_OR, PO_IT_NOT, PO_IT_MATCH_ALL,
       PO_IT_EMPTY };

struct po_iter {
  int kind;
  int32_t (*next)(po_iter*);
  int32_t (*advance)(po_iter*, int32_t target);
  void* state;
};

enum { PO_SET_SCAN, PO_SET_BITMAP, PO_SET_SORTED, PO_SET_AND, PO_SET_OR, PO_SET_NOT, PO_SET_MATCH_ALL, PO_SET_EMPTY };

struct po_docidset {
  int kind;
  po_iter* (*iterator)(po_docidset*);
  int64_t (*num_entries_scanned)(po_docidset*);
  void* state;
};

enum { PO_OP_EMPTY, PO_OP_MATCH_ALL, PO_OP_SCAN, PO_OP_INVERTED, PO_OP_SORTED, PO_OP_AND, PO_OP_OR, PO_OP_NOT,
       PO_OP_BITMAP /* BitmapBasedFilterOperator over a precomputed bitmap (star-tree traversal result) */,
       PO_OP_RANGE_INDEX /* RangeIndexBasedFilterOperator over an exact (bit-sliced) range index */ };
struct po_pred_eval;
po_bitmap* po_range_index_matching(const po_column* c, const struct po_pred_eval* e, int32_t num_docs);   /* po_rangeindex.c */

struct po_filter_op {
  int kind;
  int32_t num_docs;
  po_pred_eval* eval;
  const po_column* col;
  int n_children;
  po_filter_op** children;
  po_bitmap* bitmap;      /* PO_OP_BITMAP */
  int bitmap_exclusive;   /* BitmapBasedFilterOperator._exclusive */
  int null_handling;      /* BaseFilterOperator._nullHandlingEnabled (QueryContext#isNullHandlingEnabled) */
};

/* operator constructors shared with the star-tree filter (po_startree.c) */
po_filter_op* po_op_new(int kind, int32_t num_docs);
po_filter_op* po_leaf_filter_operator(po_pred_eval* eval, const po_column* col, int32_t num_docs);
po_filter_op* po_and_filter_operator(int n, po_filter_op** ops, int32_t num_docs);
po_filter_op* po_or_filter_operator(int n, po_filter_op** ops, int32_t num_docs);
po_filter_op* po_not_filter_operator(po_filter_op* child, int32_t num_docs);
/* StarTreeUtils#createStarTreeBasedProjectOperator + StarTreeFilterOperator: 1 = fit (*out_op is the filter over the
 * star-tree docs), 0 = not fit for this star-tree, < 0 = pg_status error */
int po_star_tree_plan(po_segment* seg, po_star_tree* st, const pg_query* q, po_filter_op** out_op);
int32_t po_star_tree_pair_index(const po_star_tree* st, int32_t function, const char* column);
/* FilterPlanNode.run */
po_filter_op* po_filter_plan(po_segment* seg, const pg_filter_node* filter, int null_handling);
po_docidset* po_filter_get_trues(po_filter_op* op);
int po_filter_can_optimize_count(po_filter_op* op);
int32_t po_filter_num_matching_docs(po_filter_op* op);
/* drains an iterator into a bitmap (BlockDocIdSet#toNonScanDocIdSet style) */

/* ---- HLL (stream-lib 2.9.8) ------------------------------------------------------------------------------------------ */
typedef struct po_hll { int32_t log2m; int32_t m; uint8_t* regs; } po_hll;
po_hll* po_hll_new(int32_t log2m);
void po_hll_free(po_hll* h);
void po_hll_offer_hash(po_hll* h, int32_t hash);
int32_t po_murmur_hash_long(int64_t v);
int32_t po_murmur_hash_bytes(const uint8_t* data, int32_t len);
void po_hll_offer_int(po_hll* h, int32_t v);
void po_hll_offer_long(po_hll* h, int64_t v);
void po_hll_offer_float(po_hll* h, float v);
void po_hll_offer_double(po_hll* h, double v);
void po_hll_offer_string(po_hll* h, const uint8_t* s, int32_t len);
void po_hll_merge(po_hll* dst, const po_hll* src);
/* ObjectSerDeUtils.HYPER_LOG_LOG_SER_DE#deserialize (HyperLogLog.Builder.build(bytes)); NULL on a malformed blob */
po_hll* po_hll_deserialize(const uint8_t* blob, int32_t len);
int64_t po_hll_cardinality(const po_hll* h);

#endif /* PO_INTERNAL_H_ */
