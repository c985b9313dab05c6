/*
 * pinot_gpu.h — C ABI of the MI355X-native Pinot segment query executor (libpinot_gpu.so).
 *
 * This is the drop-in boundary.  The reference (apachepinot/pinot 1.4.0-SNAPSHOT) is 100 % Java and
 * has no FFI for this path, so every entry point below names the Java interface it stands in for.  A JNI shim
 *  binds these 1:1; the same symbols are called by the Python/C++ parity and benchmark drivers.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; all pointers are HOST pointers unless stated otherwise;
 *   - every function returns pg_status (0 = ok, <0 = error); the message of the last error on the calling thread is
 *     returned by pg_last_error() (Java side: thrown as RuntimeException, which BaseCombineOperator wraps with the
 *     segment name — pinot-core/.../operator/combine/BaseCombineOperator.java:185-199);
 *   - buffers passed to pg_segment_add_column() are the *exact bytes* of the Pinot index entries (big-endian, as they
 *     sit in columns.psf behind the 8-byte magic marker; PinotDataBuffer address + long length —
 *     pinot-segment-spi/.../memory/PinotDataBuffer.java:162-174).  They are copied into HBM during the call and need not
 *     outlive it;
 *   - thread-safe and re-entrant: one segment may be queried from many host threads (one worker thread per segment task
 *     in the reference — BaseCombineOperator.java:97-142); each calling thread gets its own HIP stream and workspace;
 *   - results are copied into caller-allocated arrays (caller-allocated, callee-filled; JNI Get*ArrayCritical friendly).
 */
#ifndef PINOT_GPU_H_
#define PINOT_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 4

typedef enum pg_status {
  PG_OK = 0,
  PG_ERR_INVALID_ARGUMENT = -1,
  PG_ERR_UNSUPPORTED = -2,   /* query shape not handled by the GPU path: caller falls back to InstancePlanMakerImplV2 */
  PG_ERR_DEVICE = -3,        /* HIP error */
  PG_ERR_OUT_OF_MEMORY = -4,
  PG_ERR_NOT_FOUND = -5,     /* unknown column */
  PG_ERR_CANCELLED = -6,     /* EarlyTerminationException equivalent (BaseOperator.java:44-46) */
  PG_ERR_INTERNAL = -7
} pg_status;

/* Stored data types (FieldSpec.DataType#getStoredType, pinot-spi/.../data/FieldSpec.java). */
typedef enum pg_data_type {
  PG_TYPE_INT = 0,
  PG_TYPE_LONG = 1,
  PG_TYPE_FLOAT = 2,
  PG_TYPE_DOUBLE = 3,
  PG_TYPE_STRING = 4,
  PG_TYPE_BYTES = 5
} pg_data_type;

/* Forward index encodings on the path (ForwardIndexReaderFactory.java:74-109). */
typedef enum pg_fwd_encoding {
  /* FixedBitSVForwardIndexReaderV2: dictIds, MSB-first big-endian bit stream, ceil(numDocs*bits/8) bytes. */
  PG_FWD_DICT_FIXED_BIT = 0,
  /* FixedByteChunkSVForwardIndexReader: 7-int header + chunk offsets + big-endian values; chunks as written by any
   * ChunkCompressionType — PASS_THROUGH verbatim, SNAPPY / LZ4 / LZ4_LENGTH_PREFIXED decompressed in HBM, ZSTANDARD / GZIP decoded
   * on the host during pg_segment_add_column (libzstd.so.1 bound at run time: PG_ERR_UNSUPPORTED without it). */
  PG_FWD_RAW_FIXED_BYTE_CHUNK = 1,
  /* SortedIndexReaderImpl: 2 big-endian ints (startDocId, endDocId inclusive) per dictId; doubles as inverted index. */
  PG_FWD_DICT_SORTED = 2,
  /* FixedBitMVForwardIndexReader (multi-value dictionary column, .../readers/forward/FixedBitMVForwardIndexReader.java:57-76):
   * numChunks big-endian int chunk offsets | row-start bitmap of total_number_of_entries bits (MSB first, one set bit per doc) |
   * the dictIds of all docs back to back, bits_per_value each, MSB-first bit stream.  numDocsPerChunk =
   * ceil(2048 / (total_number_of_entries / numDocs)) with the reader's integer division. */
  PG_FWD_DICT_FIXED_BIT_MV = 3,
  /* VarByteChunkSVForwardIndexReader (raw STRING / BYTES column, writer versions 2 and 3): the 7-int header and chunk offsets of the
   * fixed-byte format, each chunk = numDocsPerChunk big-endian int offsets relative to the chunk start (0 for the absent rows of the
   * last chunk) followed by the values back to back.  PASS_THROUGH chunks only on the GPU path.  Such a column can be a GROUP BY key
   * (NoDictionarySingleColumnGroupKeyGenerator.java:132-140 / NoDictionaryMultiColumnGroupKeyGenerator's on-the-fly dictionaries). */
  PG_FWD_RAW_VAR_BYTE_CHUNK = 4,
  /* FixedByteChunkMVForwardIndexReader (raw, i.e. no-dictionary, multi-value column of INT / LONG / FLOAT / DOUBLE, writer versions
   * 2 and 3 — .../readers/forward/FixedByteChunkMVForwardIndexReader.java:35-140, written by MultiValueFixedByteRawIndexCreator through
   * VarByteChunkForwardIndexWriter#putIntMV ...): the var-byte chunk layout above whose value of doc d is
   * ArraySerDeUtils.serialize…ArrayWithLength = big-endian int numValues, then the values big-endian.  total_number_of_entries as for
   * PG_FWD_DICT_FIXED_BIT_MV.  PASS_THROUGH, LZ4, LZ4_LENGTH_PREFIXED, ZSTANDARD and GZIP chunks (decoded on the host at registration);
   * V4 / V5 chunk formats (VarByteChunkForwardIndexReaderV4) and SNAPPY are refused with PG_ERR_UNSUPPORTED.  The column answers
   * multi-value filters, GROUP BY keys and the *MV aggregations exactly like its dictionary-encoded twin would
   * (MultiValueRawQueriesTest asserts that equality query by query); DISTINCTCOUNTMV over it stays with the Java plan. */
  PG_FWD_RAW_MV_FIXED_BYTE_CHUNK = 5,
  /* VarByteChunkMVForwardIndexReader (raw multi-value STRING column, writer versions 2 and 3 —
   * .../readers/forward/VarByteChunkMVForwardIndexReader.java, written by MultiValueVarByteRawIndexCreator): the same var-byte chunk layout;
   * the value of doc d is ArraySerDeUtils.serializeStringArray (.../utils/ArraySerDeUtils.java:282-292): big-endian int numValues, numValues
   * big-endian int lengths, then the UTF-8 bytes.  Handled like encoding 5 (a dictionary-encoded twin built at registration, group keys back
   * as byte strings: PG_GROUP_KEY_BYTES_VALUES); raw multi-value BYTES columns are refused. */
  PG_FWD_RAW_MV_VAR_BYTE_CHUNK = 6
} pg_fwd_encoding;

typedef struct pg_buffer {
  const void* addr;
  uint64_t size;
} pg_buffer;

/*
 * One column = one DataSource (pinot-segment-spi/.../datasource/DataSource.java): forward index + optional dictionary
 * + optional inverted index.  Layouts: SURVEY.md §8a rows a12 (fixed-bit), a13 (raw chunk), a14 (dictionary),
 * a5 (BitmapInvertedIndexReader: (cardinality+1) BE uint32 offsets, then portable-format RoaringBitmap blobs).
 */
typedef struct pg_column_desc {
  const char* name;
  int32_t data_type;                /* pg_data_type (stored type) */
  int32_t fwd_encoding;             /* pg_fwd_encoding */
  int32_t has_dictionary;
  int32_t cardinality;              /* dictionary length; 0 for raw columns */
  int32_t bits_per_value;           /* PG_FWD_DICT_FIXED_BIT: PinotDataBitSet.getNumBitsPerValue(cardinality-1) */
  int32_t is_sorted;                /* DataSourceMetadata#isSorted */
  int32_t dict_bytes_per_value;     /* 4/8 numeric; padded length for fixed-width STRING/BYTES dictionaries */
  int32_t total_number_of_entries;  /* ColumnMetadata#getTotalNumberOfEntries: PG_FWD_DICT_FIXED_BIT_MV only (>= numDocs), else 0 */
  pg_buffer forward_index;
  pg_buffer dictionary;             /* sorted big-endian fixed-width values (BaseImmutableDictionary.java:45-58) */
  pg_buffer inverted_index;         /* size 0 if the column has no inverted index */
} pg_column_desc;

/* ------------------------------------------------------------------------------------------------------------------
 * Query description = the part of QueryContext the path reads (pinot-core/.../query/request/context/QueryContext.java):
 * FilterContext tree, group-by identifiers, aggregation functions and the group-by query options.
 * Literals stay strings exactly as in Predicate (pinot-common/.../request/context/predicate/{Eq,In,Range,...}Predicate.java); the native
 * PredicateEvaluator layer parses them against the column's stored type (PredicateEvaluatorProvider.java:45-96).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef enum pg_filter_type {   /* FilterContext.Type */
  PG_FILTER_AND = 0,
  PG_FILTER_OR = 1,
  PG_FILTER_NOT = 2,
  PG_FILTER_PREDICATE = 3,
  PG_FILTER_CONSTANT_TRUE = 4,
  PG_FILTER_CONSTANT_FALSE = 5
} pg_filter_type;

typedef enum pg_predicate_type {  /* Predicate.Type (subset on the path) */
  PG_PRED_EQ = 0,
  PG_PRED_NOT_EQ = 1,
  PG_PRED_IN = 2,
  PG_PRED_NOT_IN = 3,
  PG_PRED_RANGE = 4,
  PG_PRED_IS_NULL = 5,       /* BitmapBasedFilterOperator over the column's null value vector (FilterPlanNode.java:298-305); */
  PG_PRED_IS_NOT_NULL = 6    /* Empty / MatchAll when the column has none (:306-312) */
} pg_predicate_type;

#define PG_RANGE_UNBOUNDED "*"    /* RangePredicate.UNBOUNDED */

typedef struct pg_filter_node {
  int32_t type;                          /* pg_filter_type */
  int32_t n_children;                    /* AND / OR: >=1; NOT: 1 */
  const struct pg_filter_node* children; /* contiguous array of n_children nodes */
  /* PG_FILTER_PREDICATE only */
  int32_t predicate_type;                /* pg_predicate_type */
  int32_t n_values;                      /* EQ / NOT_EQ: 1; IN / NOT_IN: >=1 */
  const char* column;                    /* lhs identifier */
  const char* const* values;             /* literal strings */
  const char* lower;                     /* RANGE: lower bound or "*" */
  const char* upper;                     /* RANGE: upper bound or "*" */
  int32_t lower_inclusive;
  int32_t upper_inclusive;
} pg_filter_node;

typedef enum pg_agg_function {   /* AggregationFunctionType (subset named by north_star + AVG for the goldens) */
  PG_AGG_COUNT = 0,
  PG_AGG_SUM = 1,
  PG_AGG_MIN = 2,
  PG_AGG_MAX = 3,
  PG_AGG_AVG = 4,
  PG_AGG_DISTINCTCOUNT = 5,
  PG_AGG_DISTINCTCOUNTHLL = 6,
  PG_AGG_MINMAXRANGE = 7,
  /* the multi-value forms (CountMV / SumMV / MinMV / MaxMV / AvgMV / MinMaxRangeMV / DistinctCountMV / DistinctCountHLLMV
   * AggregationFunction.java): every value of every matching doc is aggregated; intermediate results as their single-value forms */
  PG_AGG_COUNTMV = 8,
  PG_AGG_SUMMV = 9,
  PG_AGG_MINMV = 10,
  PG_AGG_MAXMV = 11,
  PG_AGG_AVGMV = 12,
  PG_AGG_MINMAXRANGEMV = 13,
  PG_AGG_DISTINCTCOUNTMV = 14,
  PG_AGG_DISTINCTCOUNTHLLMV = 15
} pg_agg_function;

typedef struct pg_agg_spec {
  int32_t function;       /* pg_agg_function */
  int32_t log2m;          /* DISTINCTCOUNTHLL: 0 => CommonConstants.Helix.DEFAULT_HYPERLOGLOG_LOG2M (8) */
  const char* column;     /* NULL or "*" for COUNT(*) */
} pg_agg_spec;

/* One ORDER BY expression of a group-by query, as TableResizer resolves it (pinot-core/.../core/data/table/TableResizer.java:129-161):
 * a group-by expression (GroupByExpressionExtractor: the key's VALUE) or an aggregation (AggregationFunctionExtractor: the function's
 * FINAL result — COUNT a long, SUM / MIN / MAX / AVG / MINMAXRANGE a double, DISTINCTCOUNT the set's size, DISTINCTCOUNTHLL the
 * cardinality).  Post-aggregation expressions (SUM(a) + SUM(b)) and literals are not carried: such queries leave n_order_by = 0 and are
 * simply not trimmed per segment (trimming only ever DROPS groups the broker would drop anyway). */
typedef enum pg_order_by_kind {
  PG_ORDER_BY_GROUP_KEY = 0,     /* index: position in group_by_columns */
  PG_ORDER_BY_AGGREGATION = 1    /* index: position in aggregations */
} pg_order_by_kind;
typedef struct pg_order_by {
  int32_t kind;        /* pg_order_by_kind */
  int32_t index;
  int32_t ascending;   /* OrderByExpressionContext#isAsc */
  int32_t nulls_last;  /* OrderByExpressionContext#isNullsLast (read under PG_QUERY_FLAG_NULL_HANDLING only) */
} pg_order_by;

typedef struct pg_query {
  const pg_filter_node* filter;          /* NULL => MatchAllFilterOperator */
  int32_t n_group_by;                    /* 0 => AggregationOperator (no GROUP BY) */
  int32_t n_aggregations;                /* 0 => filter only (used by pg_filter_exec) */
  const char* const* group_by_columns;
  const pg_agg_spec* aggregations;
  /* InstancePlanMakerImplV2.java:75-96 defaults are applied for values <= 0 */
  int32_t num_groups_limit;                     /* DEFAULT_NUM_GROUPS_LIMIT = 100 000 */
  int32_t max_initial_result_holder_capacity;   /* DEFAULT_MAX_INITIAL_RESULT_HOLDER_CAPACITY = 10 000 */
  int32_t flags;                                /* PG_QUERY_* */
  /* Segment-level group trim (GroupByOperator.java:120-133, ABI 4): when the query has ORDER BY expressions, min_segment_group_trim_size > 0
   * (InstancePlanMakerImplV2 "min.segment.group.trim.size" / query option minSegmentGroupTrimSize; the reference's default is -1: off) and
   * the segment holds more groups than trimSize = max(5 x limit, min_segment_group_trim_size) (GroupByUtils.getTableCapacity :45-57), only
   * the trimSize groups that sort first under the ORDER BY come back (TableResizer#trimInSegmentResults :327-351; which of several
   * groups TIED at the cut survive is unspecified there — a heap — and here).  numGroupsLimitReached is decided before the trim.
   * Dense key spaces without DISTINCTCOUNT / HLL state select the survivors on the device: only they cross PCIe; ordered by a distinct count
   * (the set's size, HyperLogLog#cardinality: extractFinalResult, TableResizer.java:406-445) or a multi-value function the assembly trims. */
  int32_t n_order_by;                           /* 0 => no ORDER BY (no trim) */
  const pg_order_by* order_by;
  int32_t limit;                                /* QueryContext#getLimit (read with n_order_by > 0 only) */
  int32_t min_segment_group_trim_size;          /* <= 0: the segment's groups are never trimmed */
} pg_query;

#define PG_QUERY_FLAG_PROFILE 0x1          /* record per-kernel HIP-event timings into pg_exec_stats */
#define PG_QUERY_FLAG_SKIP_STAR_TREE 0x2   /* QueryContext#isSkipStarTree (query option useStarTree=false) */
#define PG_QUERY_FLAG_APPROX_FILTER_STATS 0x8 /* skip the exact numEntriesScannedInFilter of OR / NOT-over-scan shapes (stats_exact = 0) */
#define PG_QUERY_FLAG_EXACT_FILTER_STATS 0x10 /* compute it whatever the segment's size.  By default those shapes get the exact count (a) up to 2^27 docs
                                                 (PG_EXACT_STATS_DEVICE_MAX_DOCS) where the iterator automaton decomposes into tiles and is counted on the device —
                                                 an AND of scans, index leaves, NOTs over a leaf, ORs of leaves and of such ANDs, under drained ORs / NOTs: ~1 ms per 10^8 docs beside the
                                                 leaves' filter launches; (b) up to 2^22 docs (PG_EXACT_STATS_MAX_DOCS) otherwise — under an AND a NOT over a compound
                                                 child, an OR / NOT inside an OR: one bitmap copy to the host and a host walk per leaf, 0.2 - 3 s per 10^8 docs
                                                 (profiles/r06_filter_stats_device.txt; pg_exec_stats.filter_stats_path says which) */
#define PG_QUERY_FLAG_FINAL_DISTINCT 0x20  /* DISTINCTCOUNT / DISTINCTCOUNTHLL come back as their FINAL value (PG_RESULT_LONG: the set's size, HyperLogLog#cardinality —
                                              AggregationFunction#extractFinalResult), not as the intermediate set / registers: for a caller that
                                              merges nothing afterwards (one segment, or after pg_result_merge / _all_reduce).  The states stay in
                                              HBM; two integers per group come back (3.3 MB of registers -> 200 KB on BASELINE config 5) */
#define PG_QUERY_FLAG_NULL_HANDLING 0x40   /* QueryContext#isNullHandlingEnabled (query option enableNullHandling=true).  Filters are evaluated in
                                              three-valued logic (see pg_filter_exec_flags); an aggregation skips the docs whose argument is null
                                              (NullableSingleInputAggregationFunction#forEachNotNull) — COUNT(col) counts the values, SUM / MIN / MAX / AVG /
                                              MINMAXRANGE over no value are NULL (pg_result_agg_nulls), the distinct counts an empty set; a null is a
                                              group key of its own (pg_result_group_key_nulls).  A star-tree answers only when no column the query reads
                                              holds a null (StarTreeUtils.java:381-418); a lone COUNT(*) over an index-only filter is still
                                              FastFilteredCountOperator, which knows no nulls (AggregationPlanNode.java:104-108).  Refused
                                              (PG_ERR_UNSUPPORTED: the Java plan answers): nulls in a multi-value column, more than 3 nullable
                                              group-by columns, more groups than numGroupsLimit across their null partitions, PG_QUERY_FLAG_KEEP_DEVICE_TABLE next to nulls */
#define PG_QUERY_FLAG_KEEP_DEVICE_TABLE 0x4 /* keep the dense accumulator table in HBM with the result (pg_result_merge / _all_reduce) */

/* ExecutionStatistics (pinot-core/.../operator/ExecutionStatistics.java) + device timings. */
typedef struct pg_exec_stats {
  int64_t num_docs_scanned;
  int64_t num_entries_scanned_in_filter;
  int64_t num_entries_scanned_post_filter;
  int64_t num_total_docs;
  int32_t num_groups_limit_reached;
  int32_t stats_exact;            /* 1: num_entries_scanned_in_filter is the reference's count (see PG_QUERY_FLAG_EXACT_FILTER_STATS) */
  /* HIP-event timings on the stream the kernels ran on, milliseconds; 0 when not profiled */
  float device_ms_total;
  float device_ms_filter;
  float device_ms_aggregate;      /* fused scan+aggregate kernel when the plan is fused */
  float device_ms_reduce;
  float host_ms_plan;
  float host_ms_total;
  int64_t algorithmic_bytes;      /* bytes the plan must read once (columns + postings), for roofline accounting */
  char kernel[32];                /* name of the segment query kernel that ran (rocprofv3 kernel-trace name) */
  int32_t star_tree_index;        /* index of the star-tree the query ran on (StarTreeUtils.java:357-436), -1: none */
  int32_t filter_stats_path;      /* how num_entries_scanned_in_filter was counted — 0: by the query's own kernels (shapes with a closed form), 1: the reference's
                                     iterator automaton walked on the host over the leaves' match bitmaps, 2: the same automaton in tiles on the device
                                     (pg_filter_stats_tiles.h) */
} pg_exec_stats;

/* Intermediate result kinds (AggregationFunction#getIntermediateResultColumnType). */
typedef enum pg_result_kind {
  PG_RESULT_LONG = 0,      /* COUNT */
  PG_RESULT_DOUBLE = 1,    /* SUM / MIN / MAX */
  PG_RESULT_AVG_PAIR = 2,  /* AvgPair(sum, count) */
  PG_RESULT_MINMAX_PAIR = 3,
  PG_RESULT_DICTID_SET = 4,/* DISTINCTCOUNT over a dictionary column: set of dictIds (decoded by the caller) */
  PG_RESULT_HLL = 5,       /* HyperLogLog registers, m = 2^log2m bytes per group */
  PG_RESULT_VALUE_SET = 6  /* DISTINCTCOUNT over a raw (no-dictionary) INT / LONG / FLOAT / DOUBLE column: set of VALUES — the typed open-hash sets of
                              BaseDistinctAggregateAggregationFunction.java:325-380 (pg_result_set_sizes + pg_result_set_values_long / _double) */
} pg_result_kind;

/* ------------------------------------------------------------------------------------------------------------------
 * Star-tree index (StarTreeV2): a separate doc space of pre-aggregated records plus the tree that selects them.
 *   star_tree                   the `star_tree.index` entry exactly as OffHeapStarTree reads it — LITTLE-endian: long magic
 *                               0xBADDA55B00DAD00D, int version 1, int header size, int numDimensions, {int id, int len,
 *                               bytes}..., int numNodes, then numNodes x 7 ints {dimensionId, dimensionValue, startDocId,
 *                               endDocId, aggregatedDocId, firstChildId, lastChildId}
 *                               (pinot-segment-local/.../startree/OffHeapStarTree.java:38-85, OffHeapStarTreeNode.java:30-37,
 *                               StarTreeBuilderUtils.java:114-250)
 *   dimension_forward_indexes   FixedBitSVForwardIndexReaderV2 over the star-tree docs, bits = the parent column's
 *                               bitsPerElement, dictIds of the PARENT segment's dictionary, star stored as 0
 *                               (StarTreeLoaderUtils.java:70-78, StarTreeV2Constants.java:37-39)
 *   pairs                       one raw forward index per AggregationFunctionColumnPair ("count__*", "sum__m", ...):
 *                               LONG for COUNT, DOUBLE for SUM / MIN / MAX (fixed-byte chunk format, PASS_THROUGH), BYTES
 *                               for DISTINCTCOUNTHLL (var-byte chunk format v2/v3, PASS_THROUGH; every value a serialized
 *                               HyperLogLog: BE int log2m, BE int 4*ceil(2^log2m/6), RegisterSet words — ObjectSerDeUtils.java
 *                               :733-767); BYTES for AVG / MINMAXRANGE (16 bytes per value: AvgPair = BE double sum + BE long
 *                               count, MinMaxRangePair = BE double min + BE double max)
 *                               (StarTreeLoaderUtils.java:81-89, ValueAggregatorFactory#getAggregatedValueType)
 * The dimensions must already be registered as columns of the segment (their dictionaries are shared).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct pg_star_tree_pair {
  int32_t function;        /* pg_agg_function: COUNT / SUM / MIN / MAX / DISTINCTCOUNTHLL / AVG / MINMAXRANGE */
  int32_t data_type;       /* pg_data_type of the stored aggregate: LONG / DOUBLE / BYTES */
  const char* column;      /* "*" for COUNT */
  pg_buffer forward_index;
} pg_star_tree_pair;

typedef struct pg_star_tree_desc {
  int32_t num_docs;                          /* StarTreeV2Metadata#getNumDocs (startree.v2.N.total.docs) */
  int32_t n_dimensions;                      /* dimensionsSplitOrder */
  int32_t n_pairs;
  int32_t max_leaf_records;                  /* informational */
  const char* const* dimensions;
  const pg_buffer* dimension_forward_indexes;
  const pg_star_tree_pair* pairs;
  pg_buffer star_tree;
} pg_star_tree_desc;

typedef struct pg_segment_s* pg_segment_t;
typedef struct pg_result_s* pg_result_t;
typedef struct pg_docidset_s* pg_docidset_t;
typedef struct pg_cancel_s* pg_cancel_t;
typedef struct pg_comm_s* pg_comm_t;

/* ---- library ---------------------------------------------------------------------------------------------------- */
int32_t pg_abi_version(void);
/* Tuning / measurement knobs are PG_* environment variables (pinot_amd/csrc/pg_internal.hpp: struct Knobs lists every one), read ONCE at
 * pg_init (or at the first call that needs them).  pg_options_reload re-reads the environment: for A/B measurements and tests that flip
 * a knob inside one process.  It must not run concurrently with queries; a server never needs it. */
int32_t pg_options_reload(void);
/* Selects the DEFAULT HIP device — the one pg_segment_create() pins its segment on (one process per GPU: pg_init(LOCAL_RANK)).
 * Fails loudly if no device is present.  A server process that spreads its segments over several GPUs — the reference runs
 * every segment of a server inside one JVM, one worker task per segment (BaseCombineOperator.java:81-142) — names the device
 * per segment with pg_segment_create_on_device() instead; every entry point then runs on the device of the segment it is
 * given, on a stream private to the (calling thread, device) pair. */
int32_t pg_init(int32_t device_ordinal);
int32_t pg_device_count(int32_t* out_count);
/* Copies the calling thread's last error message into buf (NUL terminated, truncated to cap). Returns its length. */
int32_t pg_last_error(char* buf, size_t cap);

/* ---- segment life cycle: IndexSegment / ImmutableSegmentLoader.load → pin in HBM; IndexSegment#destroy ------------
 * (pinot-segment-spi/.../IndexSegment.java:137-142). */
int32_t pg_segment_create(const char* segment_name, int32_t total_docs, pg_segment_t* out_segment);
/* The segment -> GPU map of a multi-GPU server (SURVEY.md §8b "Threading"): the segment's columns, indexes and plans live in
 * the HBM of `device_ordinal`; queries on it run there whatever device other segments of the process use. */
int32_t pg_segment_create_on_device(const char* segment_name, int32_t total_docs, int32_t device_ordinal, pg_segment_t* out_segment);
int32_t pg_segment_device(pg_segment_t segment, int32_t* out_device_ordinal);
int32_t pg_segment_add_column(pg_segment_t segment, const pg_column_desc* column);
/* StarTreeLoaderUtils#loadStarTreeV2: registers star-tree number `IndexSegment#getStarTrees().size()` of the segment. */
int32_t pg_segment_add_star_tree(pg_segment_t segment, const pg_star_tree_desc* star_tree);

/* NullValueVectorReader#getNullBitmap (pinot-segment-local/.../index/readers/NullValueVectorReaderImpl.java:26-44): the column's
 * null value vector, one portable-format RoaringBitmap (the `nullvalue_vector` entry of `index_map`, file extension `.bitmap.nullvalue` in v1).  Read by IS_NULL / IS_NOT_NULL only
 * (query-level null handling is outside the path).  The bytes are copied. */
int32_t pg_segment_set_null_vector(pg_segment_t segment, const char* column, const void* roaring, uint64_t size);

/* DataSource#getRangeIndex: the column's `range_index` entry exactly as BitSlicedRangeIndexReader reads it
 * (pinot-segment-local/.../index/readers/BitSlicedRangeIndexReader.java:41-58): big-endian int version (2) and long min, then a
 * RoaringBitmap `RangeBitmap` (little-endian bit-sliced index: slice i = rows whose stored value has bit i clear; stored value =
 * dictId, value - min, or FPOrdering ordinal).  RANGE predicates on the column — and EQ when it has no inverted index — then take
 * RangeIndexBasedFilterOperator's place in the plan (FilterOperatorUtils.java:99-131; numEntriesScannedInFilter 0) instead of a
 * scan.  The bytes are copied.  Legacy (version 1, inexact) range indexes are refused: the column keeps its scan leaf. */
int32_t pg_segment_set_range_index(pg_segment_t segment, const char* column, const void* range_index, uint64_t size);

/* SegmentContext#getQueryableDocIdsSnapshot (upsert validDocIds / queryableDocIds): FilterPlanNode.run ANDs it into every
 * filter as a BitmapBasedFilterOperator (pinot-core/.../plan/FilterPlanNode.java:88-106).  One portable-format RoaringBitmap,
 * copied; replaces the previous snapshot; size 0 clears it.  Queries already running keep the snapshot they started with. */
int32_t pg_segment_set_queryable_doc_ids(pg_segment_t segment, const void* roaring, uint64_t size);
int32_t pg_segment_num_docs(pg_segment_t segment, int32_t* out_num_docs);
int32_t pg_segment_device_bytes(pg_segment_t segment, uint64_t* out_bytes);
int32_t pg_segment_destroy(pg_segment_t segment);

/* ---- FilterOperator + DocIdSetOperator: BaseFilterOperator#nextBlock → FilterBlock#getBlockDocIdSet ----------------
 * (pinot-core/.../operator/filter/BaseFilterOperator.java, DocIdSetOperator.java:59-86). */
int32_t pg_filter_exec(pg_segment_t segment, const pg_filter_node* filter, pg_docidset_t* out_docidset);
/* The same under query options: `flags` takes PG_QUERY_FLAG_NULL_HANDLING (QueryContext#isNullHandlingEnabled) — the filter tree's getTrues
 * in three-valued logic: a column predicate is true where it holds and the value is not null (BaseColumnFilterOperator.java:45-72), NOT
 * matches where its child is false, not where it is null (BaseFilterOperator.java:105-122, AndFilterOperator.java:62-90,
 * OrFilterOperator.java:61-87, NotFilterOperator.java:52-63); an always-true predicate matches the docs that hold a value
 * (FilterOperatorUtils.java:78-88).  Other flags are ignored. */
int32_t pg_filter_exec_flags(pg_segment_t segment, const pg_filter_node* filter, int32_t flags, pg_docidset_t* out_docidset);
int32_t pg_docidset_cardinality(pg_docidset_t set, int64_t* out_cardinality);
int32_t pg_docidset_num_words(pg_docidset_t set, int64_t* out_num_words);        /* ceil(numDocs/64) */
int32_t pg_docidset_copy_words(pg_docidset_t set, uint64_t* out_words, int64_t capacity_words);
int32_t pg_docidset_copy_docids(pg_docidset_t set, int32_t* out_docids, int64_t capacity); /* ascending */
int32_t pg_docidset_stats(pg_docidset_t set, pg_exec_stats* out_stats);
int32_t pg_docidset_free(pg_docidset_t set);

/* ---- GroupByOperator / AggregationOperator: Operator#nextBlock → GroupByResultsBlock / AggregationResultsBlock ----
 * (pinot-core/.../operator/query/GroupByOperator.java:100-140, AggregationOperator.java).  Returns PG_ERR_UNSUPPORTED
 * when the query shape is outside the GPU path so that GpuInstancePlanMaker can fall back to the default plan. */
int32_t pg_query_supported(pg_segment_t segment, const pg_query* query);
int32_t pg_query_exec(pg_segment_t segment, const pg_query* query, pg_result_t* out_result);

/* ---- cancellation: BaseOperator#nextBlock checks Tracing.ThreadAccountantOps.isInterrupted() and throws
 * EarlyTerminationException (pinot-core/.../operator/BaseOperator.java:36-53); the scheduler interrupts the worker thread on
 * timeout / query kill.  Here the Java side owns a token per running query: the interrupting thread calls pg_cancel_request()
 * (any thread, any time, idempotent); pg_query_exec_cancellable() polls the token before planning, between kernel launches and
 * while it waits for the device, and returns PG_ERR_CANCELLED (no result) once it is set.  A kernel already running is left
 * to finish (milliseconds); the calling thread's work areas stay consistent.  `cancel` may be NULL (= pg_query_exec). */
int32_t pg_cancel_create(pg_cancel_t* out_cancel);
int32_t pg_cancel_request(pg_cancel_t cancel);
int32_t pg_cancel_reset(pg_cancel_t cancel);
int32_t pg_cancel_destroy(pg_cancel_t cancel);
int32_t pg_query_exec_cancellable(pg_segment_t segment, const pg_query* query, pg_cancel_t cancel, pg_result_t* out_result);

int32_t pg_result_num_groups(pg_result_t result, int32_t* out_num_groups);
/* dictIds of group-by column `col` for every group (GroupKeyGenerator#getGroupKeys; decoding via Dictionary is the
 * caller's job as in DictionaryBasedGroupKeyGenerator.java:578-606). */
int32_t pg_result_group_dict_ids(pg_result_t result, int32_t col, int32_t* out_dict_ids, int32_t capacity);
/* A group-by column without a dictionary has no dictIds (NoDictionarySingleColumnGroupKeyGenerator.java:53-90,238-262: value -> group
 * id maps per stored type; NoDictionaryMultiColumnGroupKeyGenerator.java:60-130: on-the-fly dictionaries per raw column):
 * pg_result_group_key_type reports, per group-by column, PG_GROUP_KEY_LONG_VALUES (raw INT / LONG: the groups' values from
 * pg_result_group_values_long) or PG_GROUP_KEY_DOUBLE_VALUES (raw FLOAT / DOUBLE: pg_result_group_values_double; a FLOAT widened
 * exactly; keys compare by floatToIntBits / doubleToLongBits as in the fastutil maps: every NaN is one key, -0.0 and 0.0 are two);
 * pg_result_group_dict_ids fails for such a column.  Dictionary-encoded columns of the same query keep PG_GROUP_KEY_DICT_IDS. */
#define PG_GROUP_KEY_DICT_IDS 0
#define PG_GROUP_KEY_LONG_VALUES 1
#define PG_GROUP_KEY_DOUBLE_VALUES 2
/* raw STRING / BYTES column: the groups' values as byte strings — pg_result_group_values_bytes_size gives the total length,
 * pg_result_group_values_bytes fills out_offsets[0 .. numGroups] (offsets[g + 1] - offsets[g] = length of group g's value) and the
 * values back to back (a STRING is its UTF-8 bytes, as the forward index stores it) */
#define PG_GROUP_KEY_BYTES_VALUES 3
int32_t pg_result_group_values_bytes_size(pg_result_t result, int32_t col, uint64_t* out_total_bytes);
int32_t pg_result_group_values_bytes(pg_result_t result, int32_t col, int64_t* out_offsets, int32_t offsets_capacity, uint8_t* out_bytes,
                                     uint64_t bytes_capacity);
int32_t pg_result_group_key_type(pg_result_t result, int32_t col, int32_t* out_type);
int32_t pg_result_group_values_long(pg_result_t result, int32_t col, int64_t* out_values, int32_t capacity);
int32_t pg_result_group_values_double(pg_result_t result, int32_t col, double* out_values, int32_t capacity);
int32_t pg_result_kind_of(pg_result_t result, int32_t agg, int32_t* out_kind);
int32_t pg_result_doubles(pg_result_t result, int32_t agg, int32_t component, double* out, int32_t capacity);
int32_t pg_result_longs(pg_result_t result, int32_t agg, int32_t component, int64_t* out, int32_t capacity);
/* DISTINCTCOUNT: sizes[g] then the concatenated ascending dictIds of every group */
int32_t pg_result_set_sizes(pg_result_t result, int32_t agg, int32_t* out_sizes, int32_t capacity);
int32_t pg_result_set_dict_ids(pg_result_t result, int32_t agg, int32_t* out_dict_ids, int64_t capacity);
/* DISTINCTCOUNT over a raw column (PG_RESULT_VALUE_SET): sizes[g] as above, then the concatenated ascending VALUES of every group —
 * _long for INT / LONG columns, _double for FLOAT / DOUBLE columns (a FLOAT widened exactly); the other one returns PG_ERR_INVALID_ARGUMENT */
int32_t pg_result_set_values_long(pg_result_t result, int32_t agg, int64_t* out_values, int64_t capacity);
int32_t pg_result_set_values_double(pg_result_t result, int32_t agg, double* out_values, int64_t capacity);
/* DISTINCTCOUNTHLL: num_groups * 2^log2m register bytes, group-major */
int32_t pg_result_hll_registers(pg_result_t result, int32_t agg, uint8_t* out_registers, int64_t capacity);
/* The result as the bytes of a DataTableImplV4 carrying INTERMEDIATE results — what GroupByResultsBlock#getDataTable
 * (pinot-core/.../operator/blocks/results/GroupByResultsBlock.java:186-236) / AggregationResultsBlock#getDataTable (:104-155) build and
 * DataTableImplV4#toBytes (pinot-common/.../common/datatable/DataTableImplV4.java:422-517) writes: group keys as typed columns (STRING
 * through the table's string dictionary), COUNT as LONG, SUM / MIN / MAX as DOUBLE, AVG / MINMAXRANGE / DISTINCTCOUNT / DISTINCTCOUNTHLL as
 * serialized objects (ObjectSerDeUtils: AvgPair, MinMaxRangePair, typed value sets, HyperLogLog).  Rows come in the result's group order,
 * set elements ascending, no metadata entries (DataTableFactory.getDataTable(bytes) reads it).  out == NULL asks for the size only.
 * Not for results executed with PG_QUERY_FLAG_FINAL_DISTINCT (those are final values). */
int32_t pg_result_data_table_v4(pg_result_t result, uint8_t* out, int64_t capacity, int64_t* out_size);
/* Query-level null handling (PG_QUERY_FLAG_NULL_HANDLING): out[g] = 1 where group g's result of aggregation `agg` is NULL — SUM / MIN / MAX /
 * AVG / MINMAXRANGE that saw no non-null value (the reference's holders stay null: SumAggregationFunction.java:100-131,180-215) — the value
 * arrays hold 0 there; all 0 without the flag.  pg_result_group_key_nulls: out[g] = 1 where group g's key in group-by column `col` is NULL
 * (the dictId / value arrays hold 0 there). */
int32_t pg_result_agg_nulls(pg_result_t result, int32_t agg, uint8_t* out, int32_t capacity);
int32_t pg_result_group_key_nulls(pg_result_t result, int32_t col, uint8_t* out, int32_t capacity);
int32_t pg_result_stats(pg_result_t result, pg_exec_stats* out_stats);
int32_t pg_result_free(pg_result_t result);

/* ---- GroupByCombineOperator across segments that share their key space -----------------------------------------------
 * (pinot-core/.../operator/combine/GroupByCombineOperator.java:102-165,191-222; merge functions SumAggregationFunction.java
 * :223-233, MaxAggregationFunction.java:237-251, MinAggregationFunction, CountAggregationFunction,
 * DistinctCountHLLAggregationFunction.java:333-350).  The reference merges by group VALUES because dictionaries are per
 * segment; when the segments of a table share identical dictionaries for the group-by (and DISTINCTCOUNT) columns — the
 * caller's responsibility; true for the synthetic gpuBench table and for tables with pre-built global dictionaries — raw
 * keys are equal across segments and the merge is element-wise over the dense accumulator table the query left in HBM:
 * `+` for COUNT / SUM limbs, min / max for MIN / MAX, register-wise max for HyperLogLogs, OR for dictId sets.  A query
 * executed with PG_QUERY_FLAG_KEEP_DEVICE_TABLE keeps that table (and its DISTINCTCOUNT / HLL state) on the device with
 * the result.  Everything else (different dictionaries, hashed key spaces, numGroupsLimit trimming in effect) returns
 * PG_ERR_UNSUPPORTED and is merged on the host by values, as IndexedTable#upsert does.
 *   pg_result_merge       dst <- merge(dst, src), both on the same device (several segments per GPU); ExecutionStatistics add up
 *   pg_result_all_reduce  every rank of `comm` calls it with its own result of the SAME query; on return each result holds
 *                         the merged table of all ranks (RCCL all-reduce over xGMI: ncclSum / ncclMin / ncclMax on int64,
 *                         ncclMax on uint8 registers, one grouped launch; KB..MB payloads, i.e. latency-bound)
 * After either call the result's accessors (num_groups, dictIds, values, stats) describe the merged table. */
int32_t pg_result_merge(pg_result_t dst, pg_result_t src);
int32_t pg_result_all_reduce(pg_result_t result, pg_comm_t comm);

/* RCCL communicators (librccl is loaded on first use; no RCCL symbol is needed by single-GPU callers).
 *   one process per GPU (bench.py under torch.distributed.run, one JVM per GPU): rank 0 calls pg_comm_get_unique_id and ships
 *     the PG_COMM_UNIQUE_ID_BYTES bytes to the others out of band; every rank then calls pg_comm_init_rank (collective);
 *   one process, N GPUs (one JVM per server): pg_comm_init_all creates one communicator per listed device; worker thread i
 *     uses out_comms[i] with the results of the segments pinned on device_ordinals[i]. */
#define PG_COMM_UNIQUE_ID_BYTES 128
int32_t pg_comm_get_unique_id(void* out_unique_id);
int32_t pg_comm_init_rank(int32_t device_ordinal, int32_t world_size, int32_t rank, const void* unique_id, pg_comm_t* out_comm);
int32_t pg_comm_init_all(int32_t n_devices, const int32_t* device_ordinals, pg_comm_t* out_comms);
int32_t pg_comm_world_size(pg_comm_t comm, int32_t* out_world_size);
int32_t pg_comm_destroy(pg_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* PINOT_GPU_H_ */
