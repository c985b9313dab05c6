// Star-tree index on the host side of libpinot_gpu.so: registration (pins the star-tree docs in HBM as a doc space of their
// own), the "fit for star-tree" test, the predicate map and the tree traversal.  The traversal result (a docId set over the
// star-tree docs) and the remaining predicates become an ordinary filter program; group-by and aggregation then run in the
// same HIP kernels over the star-tree's columns.
//
// Mirrors (paths relative to the reference root):
//   pinot-segment-local/.../startree/v2/store/StarTreeLoaderUtils.java:53-128              loadStarTreeV2
//   pinot-segment-local/.../startree/OffHeapStarTree.java:38-85, OffHeapStarTreeNode.java:30-155   little-endian tree file
//   pinot-core/.../core/startree/StarTreeUtils.java:66-86,98-170,179-211,220-300,357-436     pairs / predicate map / fit test
//   pinot-core/.../core/startree/operator/StarTreeFilterOperator.java:155-200,208-370,386-470  filter + traversal
//   pinot-core/.../core/startree/CompositePredicateEvaluator.java:48-57
#include <algorithm>
#include <deque>
#include <set>

#include "pg_internal.hpp"

namespace pg {

Segment::Segment() = default;
Segment::~Segment() = default;

static inline int32_t le32(const uint8_t* p) {
  return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}
static inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
static inline uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }

static const int32_t kStar = -1;   // StarTreeNode.ALL
enum NodeField { kDimensionId, kDimensionValue, kStartDocId, kEndDocId, kAggregatedDocId, kFirstChildId, kLastChildId };

static const char* pair_function_name(int32_t fn) {   // AggregationFunctionType#getName
  switch (fn) {
    case PG_AGG_COUNT: return "count";
    case PG_AGG_SUM: return "sum";
    case PG_AGG_MIN: return "min";
    case PG_AGG_MAX: return "max";
    case PG_AGG_DISTINCTCOUNTHLL: return "distinctCountHLL";
    case PG_AGG_AVG: return "avg";
    case PG_AGG_MINMAXRANGE: return "minMaxRange";
    default: return nullptr;
  }
}

int StarTree::pair_index(int32_t function, const char* column) const {
  const char* col = function == PG_AGG_COUNT ? "*" : column;
  if (!col) return -1;
  for (size_t i = 0; i < pairs.size(); i++)
    if (pairs[i].function == function && pairs[i].column == col) return (int)i;
  return -1;
}

// ---- serialized HyperLogLog pair column (var-byte chunk format v2/v3, PASS_THROUGH) -----------------------------------------------
// VarByteChunkSVForwardIndexReader#getBytesUncompressed (:158-217) for every doc; each value = BE int log2m, BE int byte size,
// RegisterSet words (6 five-bit registers per BE int) — ObjectSerDeUtils.java:733-767.  Transcoded once to one byte per
// register so that a wavefront merges a doc's registers with one coalesced dword load per lane.
static void upload_hll_pair(Segment& space, Column& c, const uint8_t* fwd, uint64_t len) {
  if (len < 28) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s too short", c.name.c_str());
  const int32_t version = (int32_t)be32(fwd), num_chunks = (int32_t)be32(fwd + 4), docs_per_chunk = (int32_t)be32(fwd + 8);
  const int32_t compression = (int32_t)be32(fwd + 20), header_start = (int32_t)be32(fwd + 24);
  if (version < 2 || version > 3) fail(PG_ERR_UNSUPPORTED, "column %s: var-byte chunk writer version %d", c.name.c_str(), version);
  if (compression != 0) fail(PG_ERR_UNSUPPORTED, "column %s: chunk compression type %d (only PASS_THROUGH is on the GPU path)", c.name.c_str(), compression);
  const int off_size = version == 2 ? 4 : 8;
  if (docs_per_chunk <= 0 || (uint64_t)header_start + (uint64_t)num_chunks * off_size > len) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s corrupt", c.name.c_str());
  auto chunk_pos = [&](int32_t ch) -> uint64_t {
    const uint8_t* p = fwd + header_start + (uint64_t)ch * off_size;
    return off_size == 4 ? (uint64_t)be32(p) : be64(p);
  };
  const int32_t n = space.total_docs;
  int log2m = 0;
  std::vector<uint8_t> regs;
  size_t m = 0;
  for (int32_t doc = 0; doc < n; doc++) {
    const int32_t ch = doc / docs_per_chunk, row = doc % docs_per_chunk;
    if (ch >= num_chunks) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s truncated", c.name.c_str());
    const uint64_t cs = chunk_pos(ch), ce = ch + 1 < num_chunks ? chunk_pos(ch + 1) : len;
    if (cs + (uint64_t)docs_per_chunk * 4 > len || ce > len) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s corrupt", c.name.c_str());
    const uint64_t start = cs + be32(fwd + cs + (uint64_t)row * 4);
    uint64_t end = ce;
    if (row != docs_per_chunk - 1) {
      const uint32_t nxt = be32(fwd + cs + (uint64_t)(row + 1) * 4);
      if (nxt != 0) end = cs + nxt;
    }
    if (end > len || start + 8 > end) fail(PG_ERR_INVALID_ARGUMENT, "serialized HyperLogLog of %s doc %d is malformed", c.name.c_str(), doc);
    const uint8_t* b = fwd + start;
    const int lg = (int)be32(b), nbytes = (int)be32(b + 4);
    if (doc == 0) {
      if (lg < 4 || lg > 16) fail(PG_ERR_UNSUPPORTED, "column %s: HyperLogLog log2m %d (4..16 on the GPU path)", c.name.c_str(), lg);
      log2m = lg;
      m = (size_t)1 << lg;
      regs.assign(((size_t)space.n_tiles * PG_TILE_DOCS) * m + 256, 0);
    }
    const int n_words = (int)((m + 5) / 6);
    if (lg != log2m || nbytes != n_words * 4 || start + 8 + (uint64_t)nbytes > end)
      fail(PG_ERR_INVALID_ARGUMENT, "serialized HyperLogLog of %s doc %d is malformed", c.name.c_str(), doc);
    uint8_t* out = regs.data() + (size_t)doc * m;
    for (size_t i = 0; i < m; i++) out[i] = (uint8_t)((be32(b + 8 + (i / 6) * 4) >> (5 * (i % 6))) & 0x1F);
  }
  if (n == 0) { log2m = 8; regs.assign(256, 0); }
  c.fwd_dev.alloc(regs.size());
  c.fwd_dev.upload(regs.data(), regs.size());
  c.col_kind = PG_COL_HLL_REGS;
  c.hll_log2m = log2m;
  c.bits = log2m;
  c.fwd_bytes_logical = len;
}

// AvgPair (BE double sum, BE long count: AvgPair#toBytes) / MinMaxRangePair (BE double min, BE double max) stored as BYTES in a
// var-byte chunk forward index (AvgValueAggregator / MinMaxRangeValueAggregator: 16 bytes per star-tree doc): split into two raw
// 64-bit columns of the star-tree's doc space, which the kernels aggregate like any SUM / MIN / MAX source.
static void upload_pair16(Segment& space, const std::string& name, int32_t type_a, int32_t type_b, const char* suffix_a, const char* suffix_b,
                          const uint8_t* fwd, uint64_t len) {
  if (len < 28) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s too short", name.c_str());
  const int32_t version = (int32_t)be32(fwd), num_chunks = (int32_t)be32(fwd + 4), docs_per_chunk = (int32_t)be32(fwd + 8);
  const int32_t compression = (int32_t)be32(fwd + 20), header_start = (int32_t)be32(fwd + 24);
  if (version < 2 || version > 3) fail(PG_ERR_UNSUPPORTED, "column %s: var-byte chunk writer version %d", name.c_str(), version);
  if (compression != 0) fail(PG_ERR_UNSUPPORTED, "column %s: chunk compression type %d (only PASS_THROUGH is on the GPU path)", name.c_str(), compression);
  const int off_size = version == 2 ? 4 : 8;
  if (docs_per_chunk <= 0 || (uint64_t)header_start + (uint64_t)num_chunks * off_size > len) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s corrupt", name.c_str());
  auto chunk_pos = [&](int32_t ch) -> uint64_t {
    const uint8_t* p = fwd + header_start + (uint64_t)ch * off_size;
    return off_size == 4 ? (uint64_t)be32(p) : be64(p);
  };
  const int32_t n = space.total_docs;
  // two PASS_THROUGH fixed-byte forward indexes (version 3 header, one chunk), built on the host and registered like any raw column
  auto blob = [&](int which) {
    std::vector<uint8_t> b(28 + 8 + (size_t)std::max(n, 1) * 8, 0);
    const uint32_t hdr[7] = {3u, 1u, (uint32_t)std::max(n, 1), 8u, (uint32_t)n, 0u, 28u};
    for (int w = 0; w < 7; w++) for (int k = 0; k < 4; k++) b[(size_t)w * 4 + (size_t)k] = (uint8_t)(hdr[w] >> (24 - 8 * k));
    const uint64_t first = 28 + 8;
    for (int k = 0; k < 8; k++) b[28 + (size_t)k] = (uint8_t)(first >> (56 - 8 * k));
    (void)which;
    return b;
  };
  std::vector<uint8_t> a = blob(0), b = blob(1);
  for (int32_t doc = 0; doc < n; doc++) {
    const int32_t ch = doc / docs_per_chunk, row = doc % docs_per_chunk;
    if (ch >= num_chunks) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s truncated", name.c_str());
    const uint64_t cs = chunk_pos(ch), ce = ch + 1 < num_chunks ? chunk_pos(ch + 1) : len;
    if (cs + (uint64_t)docs_per_chunk * 4 > len || ce > len) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s corrupt", name.c_str());
    const uint64_t start = cs + be32(fwd + cs + (uint64_t)row * 4);
    uint64_t end = ce;
    if (row != docs_per_chunk - 1) {
      const uint32_t nxt = be32(fwd + cs + (uint64_t)(row + 1) * 4);
      if (nxt != 0) end = cs + nxt;
    }
    if (end > len || start + 16 > end) fail(PG_ERR_INVALID_ARGUMENT, "serialized pair of %s doc %d is malformed", name.c_str(), doc);
    memcpy(a.data() + 36 + (size_t)doc * 8, fwd + start, 8);        // stays big-endian, as a raw forward index is
    memcpy(b.data() + 36 + (size_t)doc * 8, fwd + start + 8, 8);
  }
  auto add = [&](const std::vector<uint8_t>& bytes, int32_t type, const char* suffix) {
    const std::string cn = name + suffix;
    pg_column_desc cd{};
    cd.name = cn.c_str();
    cd.data_type = type;
    cd.fwd_encoding = PG_FWD_RAW_FIXED_BYTE_CHUNK;
    cd.forward_index.addr = bytes.data();
    cd.forward_index.size = bytes.size();
    segment_add_column(space, cd);
  };
  add(a, type_a, suffix_a);
  add(b, type_b, suffix_b);
}

void segment_add_star_tree(Segment& seg, const pg_star_tree_desc& d) {
  if (d.n_dimensions <= 0 || !d.dimensions || !d.dimension_forward_indexes || !d.star_tree.addr || d.num_docs < 0 || (d.n_pairs > 0 && !d.pairs))
    fail(PG_ERR_INVALID_ARGUMENT, "bad star-tree descriptor");
  const uint8_t* t = (const uint8_t*)d.star_tree.addr;
  const uint64_t len = d.star_tree.size;
  if (len < 24 || ((uint64_t)(uint32_t)le32(t) | ((uint64_t)(uint32_t)le32(t + 4) << 32)) != 0xBADDA55B00DAD00DULL)
    fail(PG_ERR_INVALID_ARGUMENT, "Invalid magic marker in star-tree data buffer");
  if (le32(t + 8) != 1) fail(PG_ERR_INVALID_ARGUMENT, "Invalid version in star-tree data buffer");
  const int32_t root_offset = le32(t + 12), n_dims = le32(t + 16);
  if (n_dims != d.n_dimensions) fail(PG_ERR_INVALID_ARGUMENT, "star-tree has %d dimensions, descriptor %d", n_dims, d.n_dimensions);
  auto st = std::make_unique<StarTree>();
  st->dims.resize((size_t)n_dims);
  uint64_t off = 20;
  for (int i = 0; i < n_dims; i++) {
    if (off + 8 > len) fail(PG_ERR_INVALID_ARGUMENT, "star-tree header truncated");
    const int32_t id = le32(t + off), nb = le32(t + off + 4);
    off += 8;
    if (id < 0 || id >= n_dims || nb < 0 || off + (uint64_t)nb > len) fail(PG_ERR_INVALID_ARGUMENT, "star-tree header corrupt");
    st->dims[(size_t)id].assign((const char*)t + off, (size_t)nb);
    off += (uint64_t)nb;
  }
  if (off + 4 > len) fail(PG_ERR_INVALID_ARGUMENT, "star-tree header truncated");
  st->n_nodes = le32(t + off);
  off += 4;
  if ((int64_t)off != root_offset) fail(PG_ERR_INVALID_ARGUMENT, "Error loading star-tree, header length mis-match");
  if (st->n_nodes <= 0 || off + (uint64_t)st->n_nodes * 28 != len) fail(PG_ERR_INVALID_ARGUMENT, "Error loading star-tree, buffer size mis-match");
  st->nodes.resize((size_t)st->n_nodes * 7);
  for (size_t i = 0; i < st->nodes.size(); i++) st->nodes[i] = le32(t + off + i * 4);
  std::vector<uint8_t> has_parent((size_t)st->n_nodes, 0);
  for (int32_t nd = 0; nd < st->n_nodes; nd++) {   // the traversal trusts these
    const int32_t* f = &st->nodes[(size_t)nd * 7];
    const bool leaf = f[kFirstChildId] == -1;
    if (!leaf && (f[kFirstChildId] <= nd || f[kLastChildId] < f[kFirstChildId] || f[kLastChildId] >= st->n_nodes)) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d has bad child ids", nd);
    // the root splits on no dimension (OffHeapStarTreeNode: -1), every other node on one of the tree's; a node's children split on the NEXT
    // dimension and belong to it alone — the traversal indexes dims[dimensionId + 1] and walks child ranges breadth first: a root with a
    // flipped dimension id walked off the array (found by the malformed-buffer fuzz, round 5), shared children would multiply the walk
    if (nd == 0 && f[kDimensionId] != -1) fail(PG_ERR_INVALID_ARGUMENT, "star-tree root has dimension id %d", f[kDimensionId]);
    if (nd > 0 && (f[kDimensionId] < 0 || f[kDimensionId] >= n_dims)) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d has bad dimension id", nd);
    if (!leaf) {
      if (f[kDimensionId] + 1 >= n_dims) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d splits past the last dimension", nd);
      for (int32_t c = f[kFirstChildId]; c <= f[kLastChildId]; c++) {
        if (has_parent[(size_t)c]) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d has two parents", c);
        has_parent[(size_t)c] = 1;
        if (st->nodes[(size_t)c * 7 + kDimensionId] != f[kDimensionId] + 1) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d: child %d is not on the next dimension", nd, c);
      }
    }
    if (f[kAggregatedDocId] < 0 || f[kAggregatedDocId] >= d.num_docs) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d has bad aggregated docId", nd);
    // every node's doc range feeds range / match-word leaves; the builders leave the root's at (-1, -1) (never a leaf's range),
    // anything else must be a range inside the star-tree's docs
    const bool unset_root = nd == 0 && f[kStartDocId] == -1 && f[kEndDocId] == -1;
    if (!unset_root && (f[kStartDocId] < 0 || f[kEndDocId] > d.num_docs || f[kStartDocId] > f[kEndDocId])) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d has a bad doc range", nd);
  }

  Segment& sp = st->space;
  sp.name = seg.name;
  sp.device = seg.device;
  sp.total_docs = d.num_docs;
  sp.n_tiles = std::max<int32_t>(1, (int32_t)(((int64_t)d.num_docs + PG_TILE_DOCS - 1) / PG_TILE_DOCS));
  for (int i = 0; i < n_dims; i++) {
    if (!d.dimensions[i] || st->dims[(size_t)i] != d.dimensions[i])
      fail(PG_ERR_INVALID_ARGUMENT, "dimension %d is %s in the tree, %s in the descriptor", i, st->dims[(size_t)i].c_str(), d.dimensions[i] ? d.dimensions[i] : "(null)");
    Column* parent = seg.find(d.dimensions[i]);
    if (!parent || !parent->has_dictionary)
      fail(PG_ERR_INVALID_ARGUMENT, "star-tree dimension %s is not a dictionary column of the segment", d.dimensions[i]);
    pg_column_desc cd{};
    cd.name = d.dimensions[i];
    cd.data_type = parent->data_type;
    cd.fwd_encoding = PG_FWD_DICT_FIXED_BIT;
    cd.has_dictionary = 1;
    cd.cardinality = parent->cardinality;
    cd.bits_per_value = parent->bits;
    cd.dict_bytes_per_value = parent->dict_bytes_per_value;
    cd.forward_index = d.dimension_forward_indexes[i];
    cd.dictionary.addr = parent->dict_host.data();
    cd.dictionary.size = parent->dict_host.size();
    segment_add_column(sp, cd);
  }
  for (int i = 0; i < d.n_pairs; i++) {
    const pg_star_tree_pair& p = d.pairs[i];
    const char* fname = pair_function_name(p.function);
    if (!fname) fail(PG_ERR_UNSUPPORTED, "unsupported star-tree function %d", p.function);
    StarTreePair sp_pair;
    sp_pair.function = p.function;
    sp_pair.column = p.function == PG_AGG_COUNT ? "*" : (p.column ? p.column : "");
    const std::string name = std::string(fname) + "__" + sp_pair.column;   // AggregationFunctionColumnPair#toColumnName
    if (p.data_type == PG_TYPE_BYTES && (p.function == PG_AGG_AVG || p.function == PG_AGG_MINMAXRANGE)) {
      const bool avg = p.function == PG_AGG_AVG;
      upload_pair16(sp, name, PG_TYPE_DOUBLE, avg ? PG_TYPE_LONG : PG_TYPE_DOUBLE, avg ? "$sum" : "$min", avg ? "$count" : "$max",
                    (const uint8_t*)p.forward_index.addr, p.forward_index.size);
      sp_pair.col = sp.find((name + (avg ? "$sum" : "$min")).c_str());
      sp_pair.col_b = sp.find((name + (avg ? "$count" : "$max")).c_str());
      st->pairs.push_back(sp_pair);
      continue;
    }
    if (p.data_type == PG_TYPE_BYTES) {
      if (p.function != PG_AGG_DISTINCTCOUNTHLL) fail(PG_ERR_UNSUPPORTED, "star-tree pair %s (BYTES) is outside the hot path", name.c_str());
      auto col = std::make_unique<Column>();
      col->name = name;
      col->data_type = PG_TYPE_BYTES;
      col->fwd_encoding = PG_FWD_RAW_FIXED_BYTE_CHUNK;
      upload_hll_pair(sp, *col, (const uint8_t*)p.forward_index.addr, p.forward_index.size);
      sp.device_bytes += col->fwd_dev.size;
      sp.columns.emplace(name, std::move(col));
    } else {
      pg_column_desc cd{};
      cd.name = name.c_str();
      cd.data_type = p.data_type;
      cd.fwd_encoding = PG_FWD_RAW_FIXED_BYTE_CHUNK;
      cd.forward_index = p.forward_index;
      segment_add_column(sp, cd);
    }
    sp_pair.col = sp.find(name.c_str());
    st->pairs.push_back(sp_pair);
  }
  seg.device_bytes += sp.device_bytes;
  seg.star_trees.push_back(std::move(st));
}

// =====================================================================================================================
// predicate map: Map<String, List<CompositePredicateEvaluator>>
// =====================================================================================================================
struct Composite {   // predicate evaluators conjoined with OR, each possibly negated
  std::vector<PredEval> evals;
  std::vector<bool> negated;
  std::vector<int32_t> pred_types;
  bool apply(int32_t dict_id) const {   // CompositePredicateEvaluator#apply
    for (size_t i = 0; i < evals.size(); i++)
      if ((evals[i].match[(size_t)dict_id] != 0) != negated[i]) return true;
    return false;
  }
};
struct PredMap {
  std::vector<std::string> columns;               // first-insertion order
  std::vector<std::vector<Composite>> lists;
  std::vector<Composite>& at(const std::string& c) {
    for (size_t i = 0; i < columns.size(); i++) if (columns[i] == c) return lists[i];
    columns.push_back(c);
    lists.emplace_back();
    return lists.back();
  }
  const std::vector<Composite>* find(const std::string& c) const {
    for (size_t i = 0; i < columns.size(); i++) if (columns[i] == c) return &lists[i];
    return nullptr;
  }
};

// StarTreeUtils#getPredicateEvaluator: false when the predicate cannot be solved with the star-tree (no dictionary)
static bool star_pred_eval(Segment& seg, const pg_filter_node& p, PredEval* out) {
  Column* col = seg.find(p.column);
  if (!col) fail(PG_ERR_NOT_FOUND, "column not found: %s", p.column ? p.column : "(null)");
  if (!col->has_dictionary) return false;
  if (p.predicate_type == PG_PRED_IS_NULL || p.predicate_type == PG_PRED_IS_NOT_NULL) return false;   // StarTreeUtils.java:333-341
  *out = make_pred_eval(p, *col);
  return true;
}
static const pg_filter_node* unwrap_not(const pg_filter_node* f, bool* negated) {
  *negated = false;
  while (f->type == PG_FILTER_NOT) {
    if (f->n_children != 1 || !f->children) fail(PG_ERR_INVALID_ARGUMENT, "NOT needs exactly one child");
    *negated = !*negated;
    f = &f->children[0];
  }
  return f->type == PG_FILTER_PREDICATE ? f : nullptr;
}
static bool or_clause_predicates(const pg_filter_node& f, std::vector<std::pair<const pg_filter_node*, bool>>& out) {
  for (int i = 0; i < f.n_children; i++) {
    const pg_filter_node& c = f.children[i];
    if (c.type == PG_FILTER_AND) return false;
    if (c.type == PG_FILTER_OR) { if (!or_clause_predicates(c, out)) return false; continue; }
    bool neg = false;
    const pg_filter_node* p = &c;
    if (c.type == PG_FILTER_NOT) { p = unwrap_not(&c, &neg); if (!p) return false; }
    else if (c.type != PG_FILTER_PREDICATE) return false;
    out.emplace_back(p, neg);
  }
  return true;
}
static Composite single(PredEval e, bool negated, int32_t pred_type) {
  Composite c;
  c.evals.push_back(std::move(e));
  c.negated.push_back(negated);
  c.pred_types.push_back(pred_type);
  return c;
}

// extractPredicateEvaluatorsMap :98-170; false = the filter cannot be solved by the star-tree
static bool extract_pred_map(Segment& seg, const pg_filter_node* filter, PredMap& m) {
  if (!filter) return true;
  std::deque<const pg_filter_node*> queue{filter};
  while (!queue.empty()) {
    const pg_filter_node* f = queue.front();
    queue.pop_front();
    switch (f->type) {
      case PG_FILTER_AND:
        if (f->n_children < 1 || !f->children) fail(PG_ERR_INVALID_ARGUMENT, "AND/OR without children");
        for (int i = 0; i < f->n_children; i++) queue.push_back(&f->children[i]);
        break;
      case PG_FILTER_OR: {   // isOrClauseValidForStarTree :220-258
        if (f->n_children < 1 || !f->children) fail(PG_ERR_INVALID_ARGUMENT, "AND/OR without children");
        std::vector<std::pair<const pg_filter_node*, bool>> preds;
        if (!or_clause_predicates(*f, preds)) return false;
        const char* identifier = nullptr;
        Composite ce;
        bool always_true = false;
        for (auto& pn : preds) {
          PredEval e;
          if (!star_pred_eval(seg, *pn.first, &e)) return false;
          const bool neg = pn.second;
          if ((e.always_true && !neg) || (e.always_false && neg)) { always_true = true; break; }
          if ((e.always_true && neg) || (e.always_false && !neg)) continue;
          if (!identifier) identifier = pn.first->column;
          else if (strcmp(identifier, pn.first->column) != 0) return false;
          ce.evals.push_back(std::move(e));
          ce.negated.push_back(neg);
          ce.pred_types.push_back(pn.first->predicate_type);
        }
        if (always_true) break;
        if (ce.evals.empty()) return false;
        m.at(identifier).push_back(std::move(ce));
        break;
      }
      case PG_FILTER_NOT: {
        bool neg = false;
        const pg_filter_node* p = unwrap_not(f, &neg);
        if (!p) return false;
        PredEval e;
        if (!star_pred_eval(seg, *p, &e)) return false;
        if ((e.always_true && neg) || (e.always_false && !neg)) return false;
        if ((e.always_true && !neg) || (e.always_false && neg)) break;
        m.at(p->column).push_back(single(std::move(e), neg, p->predicate_type));
        break;
      }
      case PG_FILTER_PREDICATE: {
        PredEval e;
        if (!star_pred_eval(seg, *f, &e)) return false;
        if (e.always_false) return false;
        if (!e.always_true) m.at(f->column).push_back(single(std::move(e), false, f->predicate_type));
        break;
      }
      default: return false;
    }
  }
  return true;
}

// java.util.HashSet<String> iteration order: the order in which StarTreeFilterOperator#getFilterOperator turns the remaining
// predicate columns into child filters (it decides which scan sees which candidates, i.e. numEntriesScannedInFilter)
static void java_hashset_order(std::vector<std::string>& names) {
  size_t cap = 16;
  while (names.size() > cap * 3 / 4) cap *= 2;
  auto bucket = [&](const std::string& s) {
    uint32_t h = 0;
    for (unsigned char ch : s) h = 31u * h + ch;
    return (h ^ (h >> 16)) & (uint32_t)(cap - 1);
  };
  std::stable_sort(names.begin(), names.end(), [&](const std::string& a, const std::string& b) { return bucket(a) < bucket(b); });
}

// traverseStarTree :208-370.  Returns false when a predicate column has no matching dictId (empty result).
static bool traverse(Segment& seg, const StarTree& st, const PredMap& pm, const std::set<std::string>& group_by,
                     std::vector<std::pair<int32_t, int32_t>>& doc_ranges /* [start, end) */, std::vector<std::string>& remaining_out) {
  auto F = [&](int32_t node, int field) { return st.nodes[(size_t)node * 7 + (size_t)field]; };
  auto is_leaf = [&](int32_t node) { return F(node, kFirstChildId) == -1; };
  std::set<std::string> remaining_pred(pm.columns.begin(), pm.columns.end());
  std::set<std::string> remaining_gb(group_by);
  bool have_global = false;
  std::set<std::string> global_remaining;
  bool found_leaf = is_leaf(0);
  if (found_leaf) { global_remaining = remaining_pred; have_global = true; }
  std::deque<int32_t> queue{0};
  int32_t current_dim = -1;
  std::vector<uint8_t> matching;
  int32_t n_matching = 0;
  bool have_matching = false;
  while (!queue.empty()) {
    const int32_t node = queue.front();
    queue.pop_front();
    const int32_t dim = F(node, kDimensionId);
    if (dim > current_dim) {   // previous level finished
      remaining_pred.erase(st.dims[(size_t)dim]);
      remaining_gb.erase(st.dims[(size_t)dim]);
      if (found_leaf && !have_global) { global_remaining = remaining_pred; have_global = true; }
      have_matching = false;
      current_dim = dim;
    }
    if (remaining_pred.empty() && remaining_gb.empty()) {   // all matched: the aggregated document
      doc_ranges.emplace_back(F(node, kAggregatedDocId), F(node, kAggregatedDocId) + 1);
      continue;
    }
    if (is_leaf(node)) {
      if (F(node, kEndDocId) > F(node, kStartDocId)) doc_ranges.emplace_back(F(node, kStartDocId), F(node, kEndDocId));
      continue;
    }
    if (dim < -1 || dim + 1 >= (int32_t)st.dims.size()) fail(PG_ERR_INVALID_ARGUMENT, "star-tree node %d splits past the last dimension", node);
    const std::string& child_dim = st.dims[(size_t)dim + 1];
    const int32_t first = F(node, kFirstChildId), last = F(node, kLastChildId);
    int32_t star = -1;   // getChildForDimensionValue(ALL): children are sorted by value, the star child comes first
    if ((!have_global || !global_remaining.count(child_dim)) && !remaining_gb.count(child_dim) && F(first, kDimensionValue) == kStar) star = first;
    if (remaining_pred.count(child_dim)) {
      if (!have_matching) {   // getMatchingDictIds :386-470: the dictIds every composite evaluator of the column accepts
        Column* col = seg.find(child_dim.c_str());
        const std::vector<Composite>& list = *pm.find(child_dim);
        matching.assign((size_t)col->cardinality, 0);
        n_matching = 0;
        for (int32_t d = 0; d < col->cardinality; d++) {
          bool ok = true;
          for (const Composite& c : list) if (!c.apply(d)) { ok = false; break; }
          matching[(size_t)d] = ok;
          n_matching += ok;
        }
        have_matching = true;
        if (n_matching == 0) return false;
      }
      const int32_t n_children = last - first + 1;
      auto matches = [&](int32_t c) { const int32_t v = F(c, kDimensionValue); return v != kStar && v >= 0 && (size_t)v < matching.size() && matching[(size_t)v]; };
      // (binary search and scan select the same children; only the scan branch may substitute the star-node, and its
      //  condition numMatchingDictIds >= numChildren - 1 implies that branch)
      bool use_star = false;
      if (star >= 0 && n_matching >= n_children - 1) {
        int32_t hits = 0;
        for (int32_t c = first; c <= last; c++) hits += matches(c);
        use_star = hits == n_children - 1;
      }
      if (use_star) {
        queue.push_back(star);
        found_leaf |= is_leaf(star);
      } else {
        for (int32_t c = first; c <= last; c++)
          if (matches(c)) { queue.push_back(c); found_leaf |= is_leaf(c); }
      }
    } else if (star >= 0) {
      queue.push_back(star);
      found_leaf |= is_leaf(star);
    } else {
      for (int32_t c = first; c <= last; c++)
        if (F(c, kDimensionValue) != kStar) { queue.push_back(c); found_leaf |= is_leaf(c); }
    }
  }
  if (have_global) remaining_out.assign(global_remaining.begin(), global_remaining.end());
  return true;
}

OpPtr star_tree_filter(Segment& seg, StarTree& st, const pg_filter_node* filter, const pg_query& q) {
  // extractAggregationFunctionPairs + isFitForStarTree: every aggregation's stored pair must be in the tree
  for (int i = 0; i < q.n_aggregations; i++) {
    const pg_agg_spec& s = q.aggregations[i];
    if (s.function == PG_AGG_DISTINCTCOUNT) return nullptr;   // no star-tree value aggregator
    const int pi = st.pair_index(s.function, s.column);
    if (pi < 0) return nullptr;
    // DistinctCountHLLAggregationFunction#canUseStarTree (:373-383): the tree's log2m must equal the query's
    if (s.function == PG_AGG_DISTINCTCOUNTHLL && st.pairs[(size_t)pi].col->hll_log2m != (s.log2m > 0 ? s.log2m : 8)) return nullptr;
  }
  PredMap pm;
  if (!extract_pred_map(seg, filter, pm)) return nullptr;
  auto is_dim = [&](const std::string& c) { return std::find(st.dims.begin(), st.dims.end(), c) != st.dims.end(); };
  std::set<std::string> group_by;
  for (int j = 0; j < q.n_group_by; j++) {
    if (!q.group_by_columns[j] || !is_dim(q.group_by_columns[j])) return nullptr;
    group_by.insert(q.group_by_columns[j]);
  }
  for (const std::string& c : pm.columns) if (!is_dim(c)) return nullptr;

  std::vector<std::pair<int32_t, int32_t>> ranges;
  std::vector<std::string> remaining;
  if (!traverse(seg, st, pm, group_by, ranges, remaining)) return make_filter_op(OpKind::Empty);   // EmptyFilterOperator
  std::sort(ranges.begin(), ranges.end());
  auto bitmap = make_filter_op(OpKind::Bitmap);
  for (auto& r : ranges) {   // docId set of the traversal as ascending disjoint inclusive ranges
    if (!bitmap->range_lo.empty() && r.first <= bitmap->range_hi.back() + 1) bitmap->range_hi.back() = std::max(bitmap->range_hi.back(), r.second - 1);
    else { bitmap->range_lo.push_back(r.first); bitmap->range_hi.push_back(r.second - 1); }
  }
  std::vector<OpPtr> children;
  children.push_back(std::move(bitmap));
  java_hashset_order(remaining);
  for (const std::string& name : remaining) {
    Column* col = st.space.find(name.c_str());   // the star-tree's DataSource of the dimension
    for (const Composite& ce : *pm.find(name)) {
      std::vector<OpPtr> ors;
      for (size_t e = 0; e < ce.evals.size(); e++) {
        OpPtr leaf = leaf_operator(ce.evals[e], col, ce.pred_types[e]);
        ors.push_back(ce.negated[e] ? not_operator(std::move(leaf)) : std::move(leaf));
      }
      children.push_back(ors.size() == 1 ? std::move(ors[0]) : or_operator(std::move(ors)));
    }
  }
  return and_operator(std::move(children));
}

}  // namespace pg
