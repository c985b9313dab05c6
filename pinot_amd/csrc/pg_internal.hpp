// Host-side internals of libpinot_gpu.so (below the C ABI in include/pinot_gpu.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pinot_gpu.h"
#include "pg_device.h"

namespace pg {

// ---- errors ------------------------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int32_t status;
  Error(int32_t s, const std::string& m) : std::runtime_error(m), status(s) {}
};
[[noreturn]] void fail(int32_t status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void set_last_error(const std::string& msg);
const std::string& last_error();

#define PG_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      ::pg::fail(_e == hipErrorOutOfMemory ? PG_ERR_OUT_OF_MEMORY : PG_ERR_DEVICE, "HIP error %s at %s:%d: %s", \
                 hipGetErrorName(_e), __FILE__, __LINE__, #expr);                                      \
  } while (0)

// ---- device memory -------------------------------------------------------------------------------------------------------
// Allocated on the device that is current at alloc() time (every entry point first makes the segment's device current) and
// freed on that same device whatever is current then.
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t size = 0;
  int device = -1;
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n, bool zero = false) { alloc(n, zero); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : ptr(o.ptr), size(o.size), device(o.device) { o.ptr = nullptr; o.size = 0; }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) { release(); ptr = o.ptr; size = o.size; device = o.device; o.ptr = nullptr; o.size = 0; }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  void alloc(size_t n, bool zero = false);
  void release();
  void upload(const void* src, size_t n, size_t dst_off = 0);
  template <typename T> T* as() const { return reinterpret_cast<T*>(ptr); }
};
template <typename T>
DeviceBuffer upload_vector(const std::vector<T>& v) {
  DeviceBuffer b(v.empty() ? sizeof(T) : v.size() * sizeof(T));
  if (!v.empty()) b.upload(v.data(), v.size() * sizeof(T));
  return b;
}

// ---- segment ----------------------------------------------------------------------------------------------------------------
struct Column {
  std::string name;
  int32_t data_type = 0;
  int32_t fwd_encoding = 0;
  bool has_dictionary = false;
  int32_t cardinality = 0;
  int32_t bits = 0;
  bool is_sorted = false;
  int32_t dict_bytes_per_value = 0;
  // host copies needed by the planner
  std::vector<uint8_t> dict_host;               // big-endian dictionary bytes (binary search, as the reference does)
  std::vector<int32_t> sorted_start, sorted_end; // SortedIndexReaderImpl pairs
  // device
  int32_t col_kind = PG_COL_FIXED_BIT;          // layout the kernels see
  int32_t val_type = PG_V_I32;
  DeviceBuffer fwd_dev;                         // doc 0 at offset 0, padded to whole tiles
  DeviceBuffer dict_dev;                        // native-endian values
  // inverted index (BitmapInvertedIndexReader): containers re-laid out 16-byte aligned
  bool has_inverted = false;
  std::vector<PgContainer> descs_host;
  std::vector<uint32_t> posting_begin;          // cardinality + 1 offsets into descs_host
  std::vector<int64_t> posting_card;            // docs per dictId (exact: planning estimates filter selectivity from it)
  DeviceBuffer containers_dev, descs_dev;
  // bit-sliced range index (BitSlicedRangeIndexReader over a RoaringBitmap RangeBitmap): one container per (2^16-row chunk, slice)
  bool has_range_index = false;
  int32_t ri_slices = 0, ri_chunks = 0;
  int64_t ri_min = 0;                            // stored value = value - ri_min (raw INT / LONG), dictId, or FPOrdering ordinal
  uint64_t ri_bytes = 0;
  DeviceBuffer ri_containers_dev, ri_descs_dev;
  uint64_t fwd_bytes_logical = 0;               // bytes of the forward index proper (for algorithmic byte accounting)
  // value statistics behind exact SUMs (fixed-point scale, digit count), computed once at registration
  int32_t fx_exp = 0;                           // FLOAT / DOUBLE: every finite |value| < 2^fx_exp (a multiple of 16)
  bool has_nonfinite = false;                   // NaN / +-Inf among the values: SUM falls back to IEEE double accumulation
  uint64_t max_abs_int = 0;                     // LONG: largest |value|
  bool has_int_range = false;                   // raw INT / LONG: smallest and largest value (tuple packing of the partition pipeline)
  int64_t int_min = 0, int_max = 0;
  bool dict_affine = false;                     // INT / LONG dictionary whose values are base + step x dictId (ids, dense enumerations)
  int64_t dict_base = 0, dict_step = 0;
  uint64_t dict_hash = 0;                       // FNV-1a of the dictionary bytes: tables merge element-wise only over equal dictionaries
  std::map<int, DeviceBuffer> hll_luts;         // per log2m: (register index | rank << 16) of every dictionary value
  // virtual dictionary of a raw group-by column (pg_vdict.hip): a Column of bit-packed ids whose `vdict_keys` are the distinct values
  // (order-preserving 64-bit keys, ascending); built at first use, under the segment's lock
  std::unique_ptr<Column> vdict;
  std::vector<uint64_t> vdict_keys;
  int vdict_kind = -1;                          // 0 INT, 1 LONG, 2 FLOAT, 3 DOUBLE (set on the id column)
  uint64_t vdict_hash = 0;
  int32_t hll_log2m = 0;                        // PG_COL_HLL_REGS: log2m of the serialized HyperLogLogs
  // multi-value dictionary column (FixedBitMVForwardIndexReader): fwd_dev holds the dictIds of all docs back to back (the bit stream
  // of the index's raw-data section), mv_offsets_dev the first entry of every doc (numDocs + 1 ints: the row-start bitmap expanded
  // once at registration, instead of the reader's chunk-offset + bitmap walk per doc)
  // raw STRING / BYTES column (VarByteChunkSVForwardIndexReader): fwd_dev = the values back to back (chunk headers dropped at
  // registration), vb_offsets_dev = int64 [numDocs + 1]
  DeviceBuffer vb_offsets_dev;
  uint64_t vb_total_bytes = 0;
  // ... and, on its virtual dictionary (vdict_kind 4): the distinct values back to back in id order + offsets (cardinality + 1)
  std::vector<uint8_t> vdict_bytes;
  std::vector<int64_t> vdict_bytes_off;
  bool is_mv = false;
  // raw (no-dictionary) multi-value column (FixedByteChunkMVForwardIndexReader): the values are turned into a dictionary-encoded
  // multi-value column ONCE at registration — sorted distinct values + bit-packed ids in the FixedBitMV layout — kept in `vdict`
  // (complete at registration, unlike a single-value column's lazily built one); the planner resolves the column to it for filters,
  // group keys and aggregation sources, and hands the group keys back as values.  has_dictionary stays false on THIS column: the
  // dictionary-only operators (NonScanBasedAggregationOperator) do not apply to a raw column.
  bool raw_mv = false;
  Column* public_col = nullptr;                 // on the internal twin: the column the segment knows by name (statistics count it once)
  int32_t total_entries = 0;                    // ColumnMetadata#getTotalNumberOfEntries
  int32_t max_entries_per_doc = 0;              // ColumnMetadata#getMaxNumberOfMultiValues
  DeviceBuffer mv_offsets_dev;
  std::vector<int32_t> mv_offsets_host;         // the same offsets for the host-side statistics automaton (pg_filter_stats.cpp)
};

struct CompiledPlan;
struct StarTree;

struct Segment {
  std::string name;
  int device = 0;                               // HIP device ordinal whose HBM holds the segment (segment -> GPU map)
  int32_t total_docs = 0;
  int32_t n_tiles = 0;
  std::map<std::string, std::unique_ptr<Column>> columns;
  std::shared_ptr<int> alive = std::make_shared<int>(1);   // results keep a weak reference: their schema points into the columns' host dictionaries
  uint64_t device_bytes = 0;
  std::mutex mu;
  std::unordered_map<std::string, std::shared_ptr<CompiledPlan>> plan_cache;
  std::vector<std::unique_ptr<StarTree>> star_trees;   // IndexSegment#getStarTrees
  // doc-id bitmaps handed over as portable RoaringBitmaps, kept as one-posting inverted indexes (the kernels' posting leaf):
  std::map<std::string, std::shared_ptr<Column>> null_vectors;   // NullValueVectorReader#getNullBitmap per column
  std::shared_ptr<Column> queryable_doc_ids;                     // SegmentContext#getQueryableDocIdsSnapshot
  Column* find(const char* name);
  Segment();
  ~Segment();
};

// StarTreeV2 (pinot-segment-local/.../startree/v2/store/StarTreeLoaderUtils.java:53-128): the tree (host), and the star-tree
// docs as a doc space of their own — dimension columns (fixed-bit dictIds of the parent's dictionaries) and one column per
// function-column pair, pinned in HBM like any other column.
struct StarTreePair {
  int32_t function = 0;      // pg_agg_function
  std::string column;        // "*" for COUNT
  Column* col = nullptr;     // column of `space` named AggregationFunctionColumnPair#toColumnName
  Column* col_b = nullptr;   // AVG / MINMAXRANGE pairs (BYTES: AvgPair / MinMaxRangePair) are split in two raw columns: col = sum / min, col_b = count / max
};
struct StarTree {
  Segment space;
  std::vector<std::string> dims;
  std::vector<StarTreePair> pairs;
  std::vector<int32_t> nodes;   // 7 ints per node (OffHeapStarTreeNode), native endian
  int32_t n_nodes = 0;
  int pair_index(int32_t function, const char* column) const;
};

void segment_add_column(Segment& seg, const pg_column_desc& d);
void ensure_virtual_dictionary(Segment& seg, Column& c);            // pg_vdict.hip
int64_t vdict_value_of_key(uint64_t key, int kind, double* as_double);   // the value behind an order-preserving key (FLOAT / DOUBLE: IEEE double bits)
double limbs_to_double(const int64_t* limbs, int n_limbs, int q);   // sum_j limbs[j] * 2^(32 j + q), correctly rounded (pg_plan.cpp)
void segment_add_star_tree(Segment& seg, const pg_star_tree_desc& d);
// ZSTANDARD / GZIP chunks: host decode at registration (pg_host_codecs.cpp)
bool host_codec(int compression);
void host_decompress_fixed_byte_chunks(int compression, const uint8_t* file, const std::vector<uint64_t>& offs, uint32_t chunk_bytes,
                                       uint64_t total_bytes, uint8_t* dst_device, const char* column);
// one var-byte chunk of any supported ChunkCompressionType -> its bytes (at most `capacity`); PG_ERR_UNSUPPORTED / INVALID_ARGUMENT otherwise
std::vector<uint8_t> host_decompress_chunk(int compression, const uint8_t* src, uint64_t n, uint64_t capacity, const char* column);
void decompress_fixed_byte_chunks(int compression, const uint8_t* file, const std::vector<uint64_t>& offs, uint32_t chunk_bytes,
                                  uint64_t total_bytes, uint8_t* dst, const char* column);
void segment_set_null_vector(Segment& seg, const char* column, const void* roaring, uint64_t size);
void segment_set_range_index(Segment& seg, const char* column, const void* bytes, uint64_t size);
void segment_set_queryable_doc_ids(Segment& seg, const void* roaring, uint64_t size);

// ---- predicate evaluation (host): PredicateEvaluatorProvider & factories ---------------------------------------------------
struct PredEval {
  int32_t pred_type = 0;
  bool dictionary_based = false;
  bool always_true = false, always_false = false;
  bool exclusive = false;
  // dictionary based
  bool is_range = false;
  int32_t start_dict_id = 0, end_dict_id = 0;   // [start, end)
  std::vector<uint8_t> match;                   // per dictId
  std::vector<int32_t> matching, non_matching;  // ascending
  // raw based
  int32_t data_type = 0;
  int64_t lo_i = 0, hi_i = 0;
  double lo_d = 0, hi_d = 0;
  std::vector<int64_t> set_i;
  std::vector<double> set_d;
};
PredEval make_pred_eval(const pg_filter_node& p, const Column& col);

// ---- host-side doc-id bitmaps (pg_filter_stats.cpp: numEntriesScannedInFilter of leapfrogged filter shapes) ---------------------
struct HostBits {
  std::vector<uint64_t> w;   // ceil(n_docs / 64) + 1 words, bits beyond n_docs clear
  int64_t n_docs = 0;
  int64_t next_set(int64_t from) const;   // first set bit >= from, -1 when none
  int64_t cardinality() const;
  void resize_for(int64_t docs);
  void add_range(int64_t lo, int64_t hi_inclusive);
};
struct FilterOp;
using StatLeafBits = std::unordered_map<const FilterOp*, HostBits>;
int64_t emulate_entries_scanned_in_filter(const FilterOp& root, const StatLeafBits& leaves, int32_t n_docs);
// The same count without the bitmaps leaving HBM, for the shapes whose automaton decomposes into tiles (pg_filter_stats_tiles.h): an AND whose
// children are scans, index-based leaves and flat ORs of both, under any nest of drained ORs / NOTs.  `filter_stats_on_device`: does the plan's
// tree have such a shape; `entries_scanned_on_device`: the count, given every Scan / Inverted / RangeIdx leaf's match bitmap on the device
// (`arena`: scratch the call may grow; it synchronizes `stream` once, for the 8-byte answer).
using StatLeafWords = std::unordered_map<const FilterOp*, const uint64_t*>;
bool filter_stats_on_device(const FilterOp& root, int32_t n_docs);
int64_t entries_scanned_on_device(const FilterOp& root, const StatLeafWords& leaves, int32_t n_docs, DeviceBuffer& arena, void* hip_stream);

// ---- compiled plan ------------------------------------------------------------------------------------------------------------
enum class OpKind { Empty, MatchAll, Scan, Inverted, Sorted, And, Or, Not, Bitmap, RangeIdx };

struct FilterOp {
  OpKind kind = OpKind::Empty;
  PredEval eval;
  Column* col = nullptr;
  std::shared_ptr<Column> bitmap_col;        // Inverted over a docId bitmap of the segment (null vector, queryableDocIds): keeps it alive
  std::vector<std::unique_ptr<FilterOp>> children;
  std::vector<int32_t> range_lo, range_hi;   // Bitmap (BitmapBasedFilterOperator): ascending disjoint inclusive docId ranges
};
using OpPtr = std::unique_ptr<FilterOp>;
OpPtr make_filter_op(OpKind k);
OpPtr and_operator(std::vector<OpPtr> ops);
OpPtr or_operator(std::vector<OpPtr> ops);
OpPtr not_operator(OpPtr child);
OpPtr leaf_operator(PredEval ev, Column* col, int32_t predicate_type);
// StarTreeUtils#createStarTreeBasedProjectOperator + StarTreeFilterOperator: the filter over the star-tree docs when the
// query is fit for the star-tree, nullptr otherwise
OpPtr star_tree_filter(Segment& seg, StarTree& st, const pg_filter_node* filter, const pg_query& q);

enum class ResultKind { Long, Double, AvgPair, MinMaxPair };

static const int32_t kCountFromStats = -2;   // AggOut::op_a/op_b, CompiledPlan::exist_op: the match count, not a table op

struct AggOut {          // how one requested aggregation maps onto accumulator ops
  int32_t function;
  int32_t op_a = -1, op_b = -1;   // indices into ops (AVG: sum,count; MINMAXRANGE: min,max; COUNT: count op)
  bool is_float = false;
  int32_t sum_limbs = 0;           // SUM / AVG: > 0: the sum is the fixed-point number sum_j ops[op_a + j] * 2^(32 j + fx_q), rounded once
  int32_t fx_q = 0;
  bool star_count = false;         // COUNT over a star-tree: op_a is the SUM of count__* (CountAggregationFunction.java:99-106)
  int32_t aux = -1;                // DISTINCTCOUNT / DISTINCTCOUNTHLL: index into PgQueryPlan::aux
  int32_t log2m = 0;
  Column* aux_col = nullptr;
};

struct CompiledPlan {
  PgQueryPlan dev{};                 // template; per-execution pointers are patched in
  std::vector<DeviceBuffer> keep;    // device allocations referenced by dev
  std::vector<std::shared_ptr<Column>> pinned;   // doc-id bitmaps (null vectors, queryableDocIds snapshot) the program reads
  int32_t n_stat_slots = 1;
  // candidates of the scan leaf per 1000 docs as the last execution counted them (-1: never run) — pg_fast_i32range_s streams every column
  // whole, pg_fast_i32range_p skips the quads without candidates: which of the two runs follows what the plan's filter lets through
  mutable std::atomic<int> observed_candidate_permille{-1}, observed_match_permille{-1};
  mutable std::atomic<uint64_t> last_used{0};   // the segment's plan cache evicts the plans used longest ago (pg_exec.hip, get_plan)
  int64_t full_scan_entries = 0;     // entries contributed by unmasked scans whose count is known (numDocs each)
  bool stats_exact = true;           // the kernels' own counters give numEntriesScannedInFilter (flat AND shapes, drained ORs)
  mutable std::atomic<int> stats_on_device{-1};   // !stats_exact: can the tile automaton count it on the device (filter_stats_on_device; -1: not judged yet)
  // otherwise: the physical filter tree and one filter-only plan per Scan / Inverted leaf — their match bitmaps feed the iterator
  // automaton of pg_filter_stats.cpp, which reproduces the reference's count exactly
  std::unique_ptr<FilterOp> root_op;
  std::vector<std::pair<const FilterOp*, std::shared_ptr<CompiledPlan>>> stat_leaves;
  int32_t fast_scan_bits = 0;        // bits per value of the single scan leaf's column (pg_dict_count_* is chosen by it)
  bool always_empty = false;
  bool match_all = false;            // the filter is MatchAllFilterOperator: no filter pass in front of the partition pipeline
  int32_t n_projected_columns = 0;
  int64_t algorithmic_bytes = 0;
  // aggregation
  std::vector<AggOut> aggs;
  std::vector<Column*> group_cols;
  std::vector<int32_t> group_cards;
  size_t lds_bytes = 0;
  int32_t num_groups_limit = 0;
  // segment-level group trim (GroupByOperator.java:120-133): the ORDER BY expressions and trimSize = max(5 x limit, minSegmentGroupTrimSize);
  // 0: no trim.  Applied at result assembly (pg_exec.hip trim_groups), on the device first where the table allows (device_trim).
  std::vector<pg_order_by> order_by;
  int32_t trim_size = 0;
  // result assembly, dense key spaces in which EVERY group exists (a star-tree's pre-aggregated docs, low-cardinality group-bys): the groups'
  // dictIds per group-by column are a function of the plan alone — decoded once, copied afterwards (12 800 groups x 4 columns were ~20 us of the
  // star-tree route's 55 us of assembly).  Filled under full_keys_once by the first result that needs them.
  mutable std::once_flag full_keys_once;
  mutable std::vector<std::vector<int32_t>> full_keys;
  int32_t exist_op = 0;              // accumulator whose value tells whether a group was touched
  bool raw_group = false;            // the single group-by column is a raw INT / LONG column: keys are values (hash group-by)
  std::vector<Column*> group_vdict;  // per group-by column: its virtual dictionary (raw column grouped through ids), or null
  int32_t first_doc_op = -1;         // MIN(docId) per group, present when the key space exceeds numGroupsLimit
  int32_t fast_filter = -2;          // -2: interpreter kernel; -1: index-only filter; >= 0: ScanKind of the one scan leaf
  bool fast_agg = true;              // aggregation fits the fast kernels (or there is none)
  bool aux_in_lds = false;           // DISTINCTCOUNT / HLL states live in the workgroups' LDS (merged by pg_reduce_aux_kernel)
  bool digit_ops = false;            // some SUM runs on digit accumulators (PgAccValueKind): the *d kernels
  bool wide_agg = false;             // LDS-table aggregation over wide group columns / 64-bit sources: the *_w kernels
  DeviceBuffer ops_dev;
  std::vector<size_t> aux_bytes;     // bytes of each auxiliary region (256-byte multiples)
  int32_t star_tree_index = -1;      // the star-tree whose doc space the plan runs on
  int32_t space_docs = 0;            // docs of that doc space (the segment's own when no star-tree is used)
  // what pg_result_merge / pg_result_all_reduce check beyond the layout: the dictionaries behind dictId-indexed state (group-by
  // columns, DISTINCTCOUNT sets, dictionary-fed HyperLogLogs) must be equal, and the summed accumulators must still hold the sum
  std::vector<uint64_t> dict_hashes;
  uint64_t sum_max_abs = 0;          // largest |value| an int64 SUM accumulator adds per doc (0: none)
  bool has_digit_sums = false;       // 32-bit digits summed in int64 accumulators: safe below 2^31 docs in total
  bool non_scan_based = false;       // NonScanBasedAggregationOperator: answered from dictionaries on the host
};

void hll_registers_of_dictionary(Column& c, int log2m, uint8_t* regs);
double dictionary_value_as_double(const Column& c, int32_t dict_id);
std::shared_ptr<CompiledPlan> compile_plan(Segment& seg, const pg_filter_node* filter, const pg_query* query, int32_t flags = 0);   // flags: of a filter-only plan (query == nullptr)
std::string query_signature(const pg_filter_node* filter, const pg_query* query, int32_t flags = 0);

// ---- results --------------------------------------------------------------------------------------------------------------------
// Page-locked host blocks for result copies that a result may keep (pooled: hipHostMalloc costs more than a query)
struct PinnedBlock {
  void* ptr = nullptr;
  size_t size = 0;
  ~PinnedBlock();   // back to the pool
};
std::shared_ptr<PinnedBlock> acquire_pinned(size_t bytes);
struct AggResult {
  int32_t kind = PG_RESULT_DOUBLE;
  std::vector<double> d[2];
  std::vector<int64_t> l[2];
  std::vector<int32_t> set_sizes, set_ids;   // PG_RESULT_DICTID_SET: per group sizes, concatenated ascending dictIds
  int32_t set_value_kind = -1;               // PG_RESULT_VALUE_SET (a raw column through its virtual dictionary): 0 INT, 1 LONG (values in l[0]), 2 FLOAT, 3 DOUBLE (in d[0]; l[0] the IEEE bits), parallel to set_ids
  std::vector<uint8_t> hll;                  // PG_RESULT_HLL: num_groups * 2^log2m registers, or — big states —
  // the registers stay where the device wrote them: a page-locked block shared with the result (copying 3.3 MB of freshly DMA'd,
  // cache-cold bytes costs more than the kernel that produced them); group i's registers = hll_regs + hll_gids[i] * hll_stride
  std::shared_ptr<struct PinnedBlock> hll_block;
  const uint8_t* hll_regs = nullptr;
  int64_t hll_stride = 0;
  std::vector<int64_t> hll_gids;
  int32_t log2m = 0;
};
// What pg_result_merge / pg_result_all_reduce need to merge two results of the same query and to re-assemble the groups:
// the dense accumulator table [n_ops][G] + the statistics counters + the DISTINCTCOUNT / HLL regions, left in HBM.
// The key space of a table merged BY VALUE across segments whose group-by dictionaries differ (pg_comm.cpp, union_key_space): per group-by
// column the sorted union of the ranks' dictionaries — kept as a virtual dictionary (Column::vdict_kind / vdict_keys / vdict_bytes), so the
// assembly hands the groups' VALUES over as it does for raw group-by columns — and the dense layout over the unions' cardinalities
// (column 0 least significant, as in the plan's own key space).
struct UnionKeys {
  std::vector<std::unique_ptr<Column>> dicts;
  std::vector<int32_t> cards;
  std::vector<int64_t> mults;
  int64_t n_groups = 1;
};
struct DeviceTable {
  std::shared_ptr<CompiledPlan> plan;
  std::unique_ptr<UnionKeys> keys;   // set once the table has been re-keyed by pg_result_all_reduce (null: the plan's own key space)
  int device = 0;
  int32_t n_group_by = 0, n_aggregations = 0;
  int64_t n_out = 0;            // n_ops * G int64 slots, followed by PG_MAX_STATS statistics counters
  size_t aux_total = 0;         // bytes of the merged auxiliary regions (replica 0 .. n_rep-1 of every op)
  DeviceBuffer table, aux;
  int64_t full_scan_entries = 0;   // summed over the merged segments
  int64_t num_total_docs = 0;
  int64_t tail_host[2] = {0, 0};   // source of the asynchronous copy of the two above behind the statistics counters
  // overflow guards of the summed accumulators, carried by the TABLE (a merged table folds segments with different value ranges:
  // the plan of its first segment no longer bounds what it holds): largest |value| an int64 SUM adds per doc over every segment
  // folded in so far, and whether 32-bit digits are summed in int64 accumulators
  uint64_t sum_max_abs = 0;
  bool has_digit_sums = false;
};
// What a data table needs to know about a result column (pg_datatable.cpp): the name, the column's stored type and its dictionary on
// the host (a pointer into the segment's Column: the segment outlives its results)
struct ResultColumn {
  std::string name;                 // group-by column / the aggregation's argument ("*" for COUNT)
  int32_t data_type = 0;            // pg_data_type of the column
  int32_t function = -1;            // pg_agg_function (aggregations)
  const uint8_t* dict = nullptr;    // big-endian dictionary values, dict_width bytes each
  int32_t dict_width = 0;
};
struct Result {
  std::vector<ResultColumn> schema_keys, schema_aggs;
  std::weak_ptr<int> schema_segment;   // the segment whose dictionaries the schema points into (pg_result_data_table_v4 refuses once it is destroyed)
  bool schema_null_handling = false;   // enableNullHandling: COUNT(col) keeps its argument in the column name (CountAggregationFunction.java:64-66)
  std::unique_ptr<DeviceTable> dev;    // PG_QUERY_FLAG_KEEP_DEVICE_TABLE
  int32_t num_groups = 0;
  std::vector<std::vector<int32_t>> group_dict_ids;
  // no-dictionary group-by columns: per column the key type (PG_GROUP_KEY_*) and, for value keys, the groups' values (LONG values,
  // or the IEEE bits of DOUBLE values); group_dict_ids[col] stays empty for them
  std::vector<int32_t> group_key_type;
  std::vector<std::vector<int64_t>> group_values;
  std::vector<std::vector<uint8_t>> group_bytes;        // PG_GROUP_KEY_BYTES_VALUES: the groups' values back to back,
  std::vector<std::vector<int64_t>> group_bytes_off;    //   offsets (num_groups + 1)
  std::vector<AggResult> aggs;
  // PG_QUERY_FLAG_NULL_HANDLING: per aggregation / per group-by column, 1 where the group's result / key is NULL (empty vector: none is)
  std::vector<std::vector<uint8_t>> agg_nulls, key_nulls;
  bool null_handling = false;   // the query ran with PG_QUERY_FLAG_NULL_HANDLING (its data table carries the columns' null bitmaps)
  pg_exec_stats stats{};
};
struct DocIdSet {
  int device = 0;
  int32_t num_docs = 0;
  int64_t cardinality = 0;
  DeviceBuffer words;          // tile padded
  std::vector<uint32_t> tile_counts;
  pg_exec_stats stats{};
};

// cancellation token (pg_cancel_*): set by any thread, polled by the executing one
struct CancelToken { std::atomic<int> requested{0}; };

// execution (pg_exec.hip)
void device_init(int ordinal);          // pg_init: validates + selects the default device
int default_device();                   // the device pg_segment_create pins on (0 unless pg_init chose another)
void use_device(int ordinal);           // makes `ordinal` current on the calling thread, initialising it on first use
std::unique_ptr<Result> execute_query(Segment& seg, const pg_query& q, const CancelToken* cancel);         // pg_nullaware.cpp: query-level null handling above ...
// internal query flag (never set by callers: pg_query_exec masks it): the query is a part of a null-partitioned one — its filter is evaluated in
// three-valued logic even where the reference's FastFilteredCountOperator would not (a lone COUNT(*) over an index-only filter)
constexpr int32_t kQueryFlagNullPartition = 0x40000000;
std::unique_ptr<Result> execute_query_plain(Segment& seg, const pg_query& q, const CancelToken* cancel);   // ... the executor proper (pg_exec.hip)
void fill_result_schema(Segment& seg, const pg_query& q, Result& r);
int64_t hll_cardinality(const uint8_t* regs, int log2m);   // HyperLogLog#cardinality of one register row (pg_exec.hip)
void result_merge(Result& dst, Result& src);
std::vector<uint8_t> result_data_table_v4(const Result& r);   // pg_datatable.cpp
void check_null_handling(Segment& seg, const pg_query& q);     // pg_nullaware.cpp: what PG_QUERY_FLAG_NULL_HANDLING leaves to the Java plan
// RCCL (pg_comm.cpp)
struct Comm;
void comm_unique_id(void* out128);
Comm* comm_init_rank(int device, int world, int rank, const void* id128);
void comm_init_all(int n, const int32_t* devices, Comm** out);
int comm_world(const Comm& c);
void comm_destroy(Comm* c);
void result_all_reduce(Result& r, Comm& c);
// shared by result_merge / result_all_reduce (pg_exec.hip): rebuild the host view of a result from its device table
void result_reassemble(Result& r);
int64_t table_signature(const DeviceTable& T);
void check_merge_bounds(uint64_t sum_max_abs, bool has_digit_sums, int64_t total_docs);
void device_table_tail_store(DeviceTable& T, hipStream_t stream);
void merge_sets_on_stream(uint32_t* dst, const uint32_t* gathered, int64_t n_words, int n_src, hipStream_t stream);
bool merge_rekey_by_value(DeviceTable& A, DeviceTable& B, hipStream_t stream);   // pg_comm.cpp: two tables of one device into the union of their group-by dictionaries
void remap_table_on_stream(const int64_t* src, int64_t* dst, int64_t G, int64_t G2, int n_ops, int n_cols, const int32_t* maps, const int64_t* geo,
                           const PgAccOp* ops, hipStream_t stream);
hipStream_t thread_stream(int device);
std::unique_ptr<DocIdSet> execute_filter(Segment& seg, const pg_filter_node* filter, int32_t flags = 0);   // flags: PG_QUERY_FLAG_NULL_HANDLING
std::shared_ptr<CompiledPlan> get_plan(Segment& seg, const pg_filter_node* filter, const pg_query* q, int32_t flags = 0);   // cached, under seg.mu
void docidset_copy_docids(DocIdSet& s, int32_t* out, int64_t cap);


// ---- measurement / test / tuning knobs ------------------------------------------------------------------------------------------------
// Every PG_* environment variable the library honours, read ONCE (at pg_init, or on first use) into this struct — round 4 had 44 getenv
// calls in the planner and the executor, several of them per plan or per query.  pg_options_reload() re-reads the environment: the A/B
// measurements (tools/) and the tests that compare two kernels inside one process call it after changing a variable; it must not run
// concurrently with queries.  Flags are "variable present"; integers keep `unset` (-1) when absent.
struct Knobs {
  // planner (pg_plan.cpp)
  bool no_oct = false, oct_no_affine = false, oct_byte_regs = false, oct_any_cardinality = false, no_oct_prune = false, oct_dword_regs = false;
  bool no_p2 = false, p2_no_fast_a = false, no_p2_oct = false, no_radix = false, no_radix_aux = false, no_part = false, no_radix_packed = false;
  bool no_pipe_general = false, no_pipe_wide = false, no_pipe_wide_double = false, mv_no_windows = false;
  int specd_wgs_per_cu = 1;   // PG_SPECD_WGS_PER_CU: workgroups of the pg_fast_dictrange_s family per CU (where their LDS fits)
  bool specw = false;   // PG_SPECW: the shared-stage frame (pg_fast_dictrange_w, pg_kernels_specw.hip) where it fits — a measurement variant, slower than pg_fast_dictrange_s
  bool p2_no_pack = false;   // PG_P2_NO_PACK: no shared COUNT + SUM atomic in the partition pipeline's aggregation pass
  bool no_mvg = false;   // PG_NO_MVG: GROUP BY one multi-value column through pg_mv_query_l (the interpreter's frame), not pg_mv_group_*
  bool specd_no_dma = false;   // PG_SPECD_NO_DMA: the register-staged kernels also where the LDS-DMA kernels (pg_fast_dictrange_s_*_dma) fit
  bool no_specd = false, specd_no_affine = false;   // PG_NO_SPECD: no pg_fast_dictrange_s family; PG_SPECD_NO_AFFINE: arithmetic dictionaries are gathered like any other
  int64_t oct_min_docs = -1;
  bool no_oct_count_kernel = false;                // PG_NO_OCT_COUNT_KERNEL: those plans run pg_oct_l (two load buffers) instead of pg_oct_c (four)
  bool no_oct_count = false;                       // PG_NO_OCT_COUNT: COUNT(*)-only group-bys stay on the quad-layout kernels
  int64_t oct_count_min_docs = (int64_t)1 << 22;   // PG_OCT_COUNT_MIN_DOCS
  int part_min = -1;
  // executor (pg_exec.hip)
  bool force_interpreter = false, no_scan_pipe = false, no_pipe = false, no_dense_fused = false, no_part_grid_clamp = false, no_spin_wait = false;
  bool trace_oct = false, no_tile_split = false, no_oct_exec = false, no_p2_simple = false, no_dense_count = false, no_direct_result = false, trace_host = false, no_limit_prefix = false, no_device_trim = false, no_fused_finish = false;
  bool wave_specialised = false, no_wave_specialised = false;   // PG_WAVE_SPECIALISED: pg_fast_i32range_s whatever the filter lets through; PG_NO_WAVE_SPECIALISED: never
  int wave_specialised_min_permille = 150;                     // PG_WAVE_SPECIALISED_MIN_PERMILLE: ... by default from this candidate rate on (profiles/r05_wave_specialised.txt)
  int max_inflight = 16;   // PG_MAX_INFLIGHT: queries between submission and result per device (<= 0: unbounded)
  int scan_wgs_per_cu = 1, pipe_wgs_per_cu = 1, wgs_per_cu = 1, p2_wgs_per_cu = 4, dense_count_wgs = 1, tile_split_max = -1, hash_first_buckets = -1;
  int64_t exact_stats_max_docs = (int64_t)1 << 22;
  int64_t exact_stats_device_max_docs = (int64_t)1 << 27;   // PG_EXACT_STATS_DEVICE_MAX_DOCS: ... and where the device counts it (pg_filter_stats_tiles.h: ~1 ms per 10^8 docs)
  bool filter_stats_host = false;   // PG_FILTER_STATS_HOST: the iterator automaton always walks on the host (pg_filter_stats.cpp), also for shapes the device counts
  int64_t limit_prefix_min_docs = (int64_t)1 << 20;   // PG_LIMIT_PREFIX_MIN_DOCS: smallest doc prefix of the numGroupsLimit admission pass (tests lower it)
  std::string oct_passes;      // PG_OCT_PASSES: cumulative fractions, e.g. "0.02,0.08,0.3,1"
  // pg_comm.cpp
  std::string rccl_library;    // PG_RCCL_LIBRARY: the collective library to bind instead of the system's RCCL
};
const Knobs& knobs();
void knobs_reload();
}  // namespace pg
