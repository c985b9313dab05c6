// Query planning on the host: FilterContext → physical filter operators → tile filter program + aggregation plan.
//
// Mirrors (all under pinot-core/src/main/java/org/apache/pinot/core/):
//   plan/FilterPlanNode.java:88-106,195-320                      constructPhysicalOperator
//   operator/filter/FilterOperatorUtils.java:74-133              leaf selection Sorted > Inverted > Scan (RANGE skips inverted)
//   operator/filter/FilterOperatorUtils.java:136-252             AND child pruning + priority reordering
//   operator/filter/predicate/*PredicateEvaluatorFactory.java    dictionary → dictId sets / [start,end); raw → inclusive bounds
//   operator/docidsets/AndDocIdSet.java:72-186                   index-based children first, then scans restricted to the
//                                                                surviving candidates (this is what PG_F_AND_SCAN does)
//   query/aggregation/groupby/DictionaryBasedGroupKeyGenerator.java:106-185,312-354   raw key = Σ dictId_j · Π card_<j
#include <algorithm>
#include <functional>
#include <cerrno>
#include <cmath>
#include <sstream>

#include "pg_internal.hpp"
#include "pg_fixed_point.h"

extern "C" const int pg_specd_waves_per_block;   // pg_kernels_specd.hip
extern "C" int pg_specw_stage_bytes(int scan_bits, int value_bits, int bits0, int bits1, int n_bitmaps);   // pg_kernels_specw.hip
extern "C" int pg_specw_list_bytes();
extern "C" const int pg_specw_waves_per_block;
namespace pg {

// =====================================================================================================================
// literal parsing (Integer.parseInt / Long.parseLong / Float.parseFloat / Double.parseDouble)
// =====================================================================================================================
static bool parse_i64(const char* s, int64_t lo, int64_t hi, int64_t* out) {
  if (!s || !*s) return false;
  const char* p = s;
  if (*p == '-' || *p == '+') p++;
  if (!*p) return false;
  for (const char* q = p; *q; q++)
    if (*q < '0' || *q > '9') return false;
  errno = 0;
  long long v = strtoll(s, nullptr, 10);
  if (errno || v < lo || v > hi) return false;
  *out = v;
  return true;
}
static int64_t parse_int_or_fail(const char* s, bool is_long) {
  int64_t v;
  if (!parse_i64(s, is_long ? INT64_MIN : INT32_MIN, is_long ? INT64_MAX : INT32_MAX, &v))
    fail(PG_ERR_INVALID_ARGUMENT, "NumberFormatException: For input string: \"%s\"", s ? s : "null");
  return v;
}
static double parse_double_or_fail(const char* s) {
  if (!s || !*s) fail(PG_ERR_INVALID_ARGUMENT, "NumberFormatException: empty String");
  char* end = nullptr;
  double v = strtod(s, &end);
  if (end == s || *end) fail(PG_ERR_INVALID_ARGUMENT, "NumberFormatException: For input string: \"%s\"", s);
  return v;
}
static float parse_float_or_fail(const char* s) {
  if (!s || !*s) fail(PG_ERR_INVALID_ARGUMENT, "NumberFormatException: empty String");
  char* end = nullptr;
  float v = strtof(s, &end);
  if (end == s || *end) fail(PG_ERR_INVALID_ARGUMENT, "NumberFormatException: For input string: \"%s\"", s);
  return v;
}

static inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
static inline uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }

// Dictionary#insertionIndexOf: >= 0 when found, else -(insertionPoint + 1) (BaseImmutableDictionary.java:124-245)
static int32_t dict_insertion_index(const Column& c, const char* sv) {
  int32_t low = 0, high = c.cardinality - 1;
  const uint8_t* d = c.dict_host.data();
  auto search = [&](auto value, auto get) -> int32_t {
    while (low <= high) {
      int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
      auto mv = get(mid);
      if (mv < value) low = mid + 1;
      else if (mv > value) high = mid - 1;
      else return mid;
    }
    return -(low + 1);
  };
  switch (c.data_type) {
    case PG_TYPE_INT: {
      int32_t v = (int32_t)parse_int_or_fail(sv, false);
      return search(v, [&](int32_t i) { return (int32_t)be32(d + (size_t)i * 4); });
    }
    case PG_TYPE_LONG: {
      int64_t v = parse_int_or_fail(sv, true);
      return search(v, [&](int32_t i) { return (int64_t)be64(d + (size_t)i * 8); });
    }
    case PG_TYPE_FLOAT: {
      float v = parse_float_or_fail(sv);
      return search(v, [&](int32_t i) { uint32_t u = be32(d + (size_t)i * 4); float f; memcpy(&f, &u, 4); return f; });
    }
    case PG_TYPE_DOUBLE: {
      double v = parse_double_or_fail(sv);
      return search(v, [&](int32_t i) { uint64_t u = be64(d + (size_t)i * 8); double f; memcpy(&f, &u, 8); return f; });
    }
    default: {  // STRING / BYTES fixed-width entries padded with zeros; compare the unpadded bytes
      const int w = c.dict_bytes_per_value;
      const size_t slen = strlen(sv);
      while (low <= high) {
        int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
        const uint8_t* e = d + (size_t)mid * w;
        size_t elen = (size_t)w;
        while (elen > 0 && e[elen - 1] == 0) elen--;
        size_t n = elen < slen ? elen : slen;
        int cmp = memcmp(e, sv, n);
        if (cmp == 0) cmp = elen == slen ? 0 : (elen < slen ? -1 : 1);
        if (cmp < 0) low = mid + 1;
        else if (cmp > 0) high = mid - 1;
        else return mid;
      }
      return -(low + 1);
    }
  }
}

// A predicate over a raw (no-dictionary) single-value STRING column, evaluated over the column's VIRTUAL dictionary (pg_vdict.hip: the distinct
// values + bit-packed ids): the raw-value evaluators of the reference — value.equals / set.contains / String#compareTo against the bounds
// (EqualsPredicateEvaluatorFactory.java, InPredicateEvaluatorFactory.java, RangePredicateEvaluatorFactory.java: the String raw evaluators) —
// applied ONCE per distinct value; the scan then tests ids.  No always-true / always-false folding: a raw evaluator has no dictionary to tell,
// the reference scans every doc (and counts it) whatever the literal.
static int utf16_order(const uint8_t* a, size_t alen, const uint8_t* b, size_t blen) {   // String.compareTo over UTF-8 bytes (see pg_exec.hip: utf16_unit_order)
  const size_t n = alen < blen ? alen : blen;
  for (size_t i = 0; i < n; i++) {
    if (a[i] == b[i]) continue;
    const int x = a[i] == 0xEE || a[i] == 0xEF ? a[i] + 0x10 : a[i], y = b[i] == 0xEE || b[i] == 0xEF ? b[i] + 0x10 : b[i];
    return x < y ? -1 : 1;
  }
  return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
static PredEval make_raw_string_eval(const pg_filter_node& p, const Column& twin) {
  PredEval e;
  e.pred_type = p.predicate_type;
  e.data_type = PG_TYPE_STRING;
  e.dictionary_based = true;
  const bool is_range = p.predicate_type == PG_PRED_RANGE;
  if (!is_range && (p.n_values < 1 || !p.values)) fail(PG_ERR_INVALID_ARGUMENT, "predicate on %s has no value", twin.name.c_str());
  for (int i = 0; !is_range && i < p.n_values; i++)
    if (!p.values[i]) fail(PG_ERR_INVALID_ARGUMENT, "predicate on %s: value %d is null", twin.name.c_str(), i);
  if (is_range && (!p.lower || !p.upper)) fail(PG_ERR_INVALID_ARGUMENT, "range predicate on %s without bounds", twin.name.c_str());
  const int32_t card = twin.cardinality;
  e.match.assign((size_t)card, 0);
  auto value = [&](int32_t id, size_t* len) { *len = (size_t)(twin.vdict_bytes_off[(size_t)id + 1] - twin.vdict_bytes_off[(size_t)id]); return twin.vdict_bytes.data() + twin.vdict_bytes_off[(size_t)id]; };
  switch (p.predicate_type) {
    case PG_PRED_EQ: case PG_PRED_NOT_EQ: case PG_PRED_IN: case PG_PRED_NOT_IN: {
      const bool neg = p.predicate_type == PG_PRED_NOT_EQ || p.predicate_type == PG_PRED_NOT_IN;
      const int nv = (p.predicate_type == PG_PRED_EQ || p.predicate_type == PG_PRED_NOT_EQ) ? 1 : p.n_values;
      for (int32_t d = 0; d < card; d++) {
        size_t len; const uint8_t* v = value(d, &len);
        bool found = false;
        for (int i = 0; i < nv && !found; i++) found = strlen(p.values[i]) == len && memcmp(p.values[i], v, len) == 0;
        e.match[(size_t)d] = found != neg;
      }
      e.exclusive = neg;
      break;
    }
    case PG_PRED_RANGE: {
      const bool lo_unb = strcmp(p.lower, PG_RANGE_UNBOUNDED) == 0, hi_unb = strcmp(p.upper, PG_RANGE_UNBOUNDED) == 0;
      for (int32_t d = 0; d < card; d++) {
        size_t len; const uint8_t* v = value(d, &len);
        bool ok = true;
        if (!lo_unb) { const int c = utf16_order(v, len, (const uint8_t*)p.lower, strlen(p.lower)); ok = c > 0 || (c == 0 && p.lower_inclusive); }
        if (ok && !hi_unb) { const int c = utf16_order(v, len, (const uint8_t*)p.upper, strlen(p.upper)); ok = c < 0 || (c == 0 && p.upper_inclusive); }
        e.match[(size_t)d] = ok;
      }
      break;
    }
    default: fail(PG_ERR_UNSUPPORTED, "predicate type %d over the raw STRING column %s", p.predicate_type, twin.name.c_str());
  }
  for (int32_t d = 0; d < card; d++) (e.match[(size_t)d] ? e.matching : e.non_matching).push_back(d);
  return e;
}

PredEval make_pred_eval(const pg_filter_node& p, const Column& col) {
  PredEval e;
  e.pred_type = p.predicate_type;
  e.data_type = col.data_type;
  const bool is_range = p.predicate_type == PG_PRED_RANGE;
  if (!is_range && (p.n_values < 1 || !p.values)) fail(PG_ERR_INVALID_ARGUMENT, "predicate on %s has no value", col.name.c_str());
  for (int i = 0; !is_range && i < p.n_values; i++)
    if (!p.values[i]) fail(PG_ERR_INVALID_ARGUMENT, "predicate on %s: value %d is null", col.name.c_str(), i);
  if (is_range && (!p.lower || !p.upper)) fail(PG_ERR_INVALID_ARGUMENT, "range predicate on %s without bounds", col.name.c_str());
  if (col.has_dictionary) {
    e.dictionary_based = true;
    const int32_t card = col.cardinality;
    e.match.assign((size_t)card, 0);
    switch (p.predicate_type) {
      case PG_PRED_EQ: {   // EqualsPredicateEvaluatorFactory.java:95-108
        int32_t idx = dict_insertion_index(col, p.values[0]);
        if (idx >= 0) { e.match[idx] = 1; if (card == 1) e.always_true = true; }
        else e.always_false = true;
        break;
      }
      case PG_PRED_NOT_EQ: {
        int32_t idx = dict_insertion_index(col, p.values[0]);
        std::fill(e.match.begin(), e.match.end(), 1);
        e.exclusive = true;
        if (idx >= 0) { e.match[idx] = 0; if (card == 1) e.always_false = true; }
        else e.always_true = true;
        break;
      }
      case PG_PRED_IN:
      case PG_PRED_NOT_IN: {   // InPredicateEvaluatorFactory.java:158-171, NotInPredicateEvaluatorFactory.java:158-171
        int32_t found = 0;
        for (int i = 0; i < p.n_values; i++) {
          int32_t idx = dict_insertion_index(col, p.values[i]);
          if (idx >= 0 && !e.match[idx]) { e.match[idx] = 1; found++; }
        }
        if (p.predicate_type == PG_PRED_IN) {
          if (found == 0) e.always_false = true;
          else if (found == card) e.always_true = true;
        } else {
          for (auto& m : e.match) m = !m;
          e.exclusive = true;
          if (found == 0) e.always_true = true;
          else if (found == card) e.always_false = true;
        }
        break;
      }
      case PG_PRED_RANGE: {   // SortedDictionaryBasedRangePredicateEvaluator, RangePredicateEvaluatorFactory.java:126-167
        e.is_range = true;
        if (strcmp(p.lower, PG_RANGE_UNBOUNDED) == 0) e.start_dict_id = 0;
        else {
          int32_t ins = dict_insertion_index(col, p.lower);
          e.start_dict_id = ins < 0 ? -(ins + 1) : (p.lower_inclusive ? ins : ins + 1);
        }
        if (strcmp(p.upper, PG_RANGE_UNBOUNDED) == 0) e.end_dict_id = card;
        else {
          int32_t ins = dict_insertion_index(col, p.upper);
          e.end_dict_id = ins < 0 ? -(ins + 1) : (p.upper_inclusive ? ins + 1 : ins);
        }
        int32_t n = std::max(e.end_dict_id - e.start_dict_id, 0);
        if (n == 0) e.always_false = true;
        else if (n == card) e.always_true = true;
        for (int32_t d = e.start_dict_id; d < e.end_dict_id; d++) e.match[d] = 1;
        break;
      }
      default: fail(PG_ERR_UNSUPPORTED, "predicate type %d", p.predicate_type);
    }
    for (int32_t d = 0; d < card; d++) (e.match[d] ? e.matching : e.non_matching).push_back(d);
    return e;
  }
  // raw value based (RangePredicateEvaluatorFactory.java:68-117,331-380 and the Eq/In raw evaluators)
  const int t = col.data_type;
  if (t != PG_TYPE_INT && t != PG_TYPE_LONG && t != PG_TYPE_FLOAT && t != PG_TYPE_DOUBLE)
    fail(PG_ERR_UNSUPPORTED, "raw predicate on column %s of type %d", col.name.c_str(), t);
  if (is_range) {
    const bool lo_unb = strcmp(p.lower, PG_RANGE_UNBOUNDED) == 0, hi_unb = strcmp(p.upper, PG_RANGE_UNBOUNDED) == 0;
    const bool lo_inc = lo_unb || p.lower_inclusive, hi_inc = hi_unb || p.upper_inclusive;
    if (t == PG_TYPE_INT || t == PG_TYPE_LONG) {
      const bool is_long = t == PG_TYPE_LONG;
      int64_t tmin = is_long ? INT64_MIN : INT32_MIN, tmax = is_long ? INT64_MAX : INT32_MAX;
      int64_t lo = lo_unb ? tmin : parse_int_or_fail(p.lower, is_long);
      int64_t hi = hi_unb ? tmax : parse_int_or_fail(p.upper, is_long);
      if (!lo_inc) { if (lo == tmax) fail(PG_ERR_INVALID_ARGUMENT, "Invalid range"); lo += 1; }
      if (!hi_inc) { if (hi == tmin) fail(PG_ERR_INVALID_ARGUMENT, "Invalid range"); hi -= 1; }
      e.lo_i = lo; e.hi_i = hi;
    } else if (t == PG_TYPE_FLOAT) {
      float lo = lo_unb ? -INFINITY : parse_float_or_fail(p.lower);
      float hi = hi_unb ? INFINITY : parse_float_or_fail(p.upper);
      // Math.nextUp / nextDown + checkArgument (RangePredicateEvaluatorFactory.java:449-456): an exclusive bound at its infinity (or NaN)
      // is "Invalid range", not an empty match
      if (!lo_inc) { const float n = std::nextafter(lo, INFINITY); if (!(n > lo)) fail(PG_ERR_INVALID_ARGUMENT, "Invalid range"); lo = n; }
      if (!hi_inc) { const float n = std::nextafter(hi, -INFINITY); if (!(n < hi)) fail(PG_ERR_INVALID_ARGUMENT, "Invalid range"); hi = n; }
      e.lo_d = lo; e.hi_d = hi;
    } else {
      double lo = lo_unb ? -INFINITY : parse_double_or_fail(p.lower);
      double hi = hi_unb ? INFINITY : parse_double_or_fail(p.upper);
      if (!lo_inc) { const double n = std::nextafter(lo, (double)INFINITY); if (!(n > lo)) fail(PG_ERR_INVALID_ARGUMENT, "Invalid range"); lo = n; }
      if (!hi_inc) { const double n = std::nextafter(hi, -(double)INFINITY); if (!(n < hi)) fail(PG_ERR_INVALID_ARGUMENT, "Invalid range"); hi = n; }
      e.lo_d = lo; e.hi_d = hi;
    }
    return e;
  }
  e.exclusive = (p.predicate_type == PG_PRED_NOT_EQ || p.predicate_type == PG_PRED_NOT_IN);
  for (int i = 0; i < p.n_values; i++) {
    if (t == PG_TYPE_INT) e.set_i.push_back(parse_int_or_fail(p.values[i], false));
    else if (t == PG_TYPE_LONG) e.set_i.push_back(parse_int_or_fail(p.values[i], true));
    else if (t == PG_TYPE_FLOAT) e.set_d.push_back((double)parse_float_or_fail(p.values[i]));
    else e.set_d.push_back(parse_double_or_fail(p.values[i]));
  }
  return e;
}

// =====================================================================================================================
// physical filter operators
// =====================================================================================================================
// the single-value function a multi-value aggregation function extends (CountMVAggregationFunction extends CountAggregationFunction, ...):
// intermediate results, merges and final results are the parent's; only what is aggregated per doc differs
static int32_t sv_function_of(int32_t f) {
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
static bool is_mv_function(int32_t f) { return f >= PG_AGG_COUNTMV && f <= PG_AGG_DISTINCTCOUNTHLLMV; }

OpPtr make_filter_op(OpKind k) { auto o = std::make_unique<FilterOp>(); o->kind = k; return o; }
static OpPtr make_op(OpKind k) { return make_filter_op(k); }

static int priority(const FilterOp& op) {   // PrioritizedFilterOperator.java:31-38
  switch (op.kind) {
    case OpKind::Sorted: return 0;
    case OpKind::Inverted: return 100;
    case OpKind::Bitmap: return 100;   // BitmapBasedFilterOperator: MEDIUM_PRIORITY
    case OpKind::RangeIdx: return 200; // RangeIndexBasedFilterOperator: LOW_PRIORITY (FilterOperatorUtils.java:224-230)
    case OpKind::And: return 300;
    case OpKind::Or: return 400;
    case OpKind::Not: return priority(*op.children[0]);
    case OpKind::Scan: return op.col && op.col->is_mv ? 550 : 500;   // getScanBasedFilterPriority (FilterOperatorUtils.java:253-265): multi-value scans last
    default: return 10000;
  }
}

OpPtr and_operator(std::vector<OpPtr> ops) {   // getAndFilterOperator
  std::vector<OpPtr> ch;
  for (auto& o : ops) {
    if (o->kind == OpKind::Empty) return make_op(OpKind::Empty);
    if (o->kind != OpKind::MatchAll) ch.push_back(std::move(o));
  }
  if (ch.empty()) return make_op(OpKind::MatchAll);
  if (ch.size() == 1) return std::move(ch[0]);
  std::stable_sort(ch.begin(), ch.end(), [](const OpPtr& a, const OpPtr& b) { return priority(*a) < priority(*b); });
  auto r = make_op(OpKind::And);
  r->children = std::move(ch);
  return r;
}
OpPtr or_operator(std::vector<OpPtr> ops) {    // getOrFilterOperator
  std::vector<OpPtr> ch;
  for (auto& o : ops) {
    if (o->kind == OpKind::MatchAll) return make_op(OpKind::MatchAll);
    if (o->kind != OpKind::Empty) ch.push_back(std::move(o));
  }
  if (ch.empty()) return make_op(OpKind::Empty);
  if (ch.size() == 1) return std::move(ch[0]);
  auto r = make_op(OpKind::Or);
  r->children = std::move(ch);
  return r;
}
OpPtr not_operator(OpPtr child) {              // getNotFilterOperator
  if (child->kind == OpKind::MatchAll) return make_op(OpKind::Empty);
  if (child->kind == OpKind::Empty) return make_op(OpKind::MatchAll);
  auto r = make_op(OpKind::Not);
  r->children.push_back(std::move(child));
  return r;
}

// FilterOperatorUtils.DefaultImplementation#getLeafFilterOperator (:74-133): Sorted > Inverted > Scan; RANGE skips the inverted index
OpPtr leaf_operator(PredEval ev, Column* col, int32_t predicate_type) {
  if (ev.always_false) return make_op(OpKind::Empty);
  if (ev.always_true) return make_op(OpKind::MatchAll);
  const bool sorted_ok = col->is_sorted && col->has_dictionary && !col->sorted_start.empty();
  OpKind k;
  // RANGE: Sorted > RangeIndex > Scan; other predicates: Sorted > Inverted > RangeIndex (EQ over an exact range index:
  // RangeIndexBasedFilterOperator.canEvaluate, RangeIndexBasedFilterOperator.java:58-63) > Scan
  if (predicate_type == PG_PRED_RANGE) k = sorted_ok ? OpKind::Sorted : (col->has_range_index ? OpKind::RangeIdx : OpKind::Scan);
  else k = sorted_ok ? OpKind::Sorted
                     : (col->has_inverted ? OpKind::Inverted : (col->has_range_index && predicate_type == PG_PRED_EQ ? OpKind::RangeIdx : OpKind::Scan));
  auto op = make_op(k);
  op->eval = std::move(ev);
  op->col = col;
  return op;
}

// BitmapBasedFilterOperator(docIds, exclusive) over a RoaringBitmap the segment holds: the posting leaf of a one-entry inverted
// index (same priority, same zero numEntriesScannedInFilter, same canOptimizeCount as InvertedIndexFilterOperator)
static OpPtr bitmap_operator(std::shared_ptr<Column> bitmap, bool exclusive) {
  auto op = make_op(OpKind::Inverted);
  op->col = bitmap.get();
  op->bitmap_col = std::move(bitmap);
  op->eval.dictionary_based = true;
  op->eval.exclusive = exclusive;
  (exclusive ? op->eval.non_matching : op->eval.matching).push_back(0);
  return op;
}

// the column's null value vector when it holds a null (NullValueVectorReader#getNullBitmap non-null and non-empty), nullptr otherwise
static std::shared_ptr<Column> null_vector_of(Segment& seg, const Column* col) {
  if (!col) return nullptr;
  const Column* named = col->public_col ? col->public_col : col;
  auto it = seg.null_vectors.find(named->name);
  if (it == seg.null_vectors.end() || !it->second) return nullptr;
  const Column& nv = *it->second;
  return !nv.posting_card.empty() && nv.posting_card[0] > 0 ? it->second : nullptr;
}

static OpPtr construct(Segment& seg, const pg_filter_node& f, bool nh) {   // FilterPlanNode#constructPhysicalOperator
  switch (f.type) {
    case PG_FILTER_AND:
    case PG_FILTER_OR: {
      if (f.n_children < 1 || !f.children) fail(PG_ERR_INVALID_ARGUMENT, "AND/OR without children");
      std::vector<OpPtr> ch;
      for (int i = 0; i < f.n_children; i++) ch.push_back(construct(seg, f.children[i], nh));
      return f.type == PG_FILTER_AND ? and_operator(std::move(ch)) : or_operator(std::move(ch));
    }
    case PG_FILTER_NOT:
      if (f.n_children != 1 || !f.children) fail(PG_ERR_INVALID_ARGUMENT, "NOT needs exactly one child");
      return not_operator(construct(seg, f.children[0], nh));
    case PG_FILTER_PREDICATE: {
      Column* col = seg.find(f.column);
      if (!col) fail(PG_ERR_NOT_FOUND, "column not found: %s", f.column ? f.column : "(null)");
      if (f.predicate_type == PG_PRED_IS_NULL || f.predicate_type == PG_PRED_IS_NOT_NULL) {   // FilterPlanNode.java:298-312
        const bool not_null = f.predicate_type == PG_PRED_IS_NOT_NULL;
        auto it = seg.null_vectors.find(col->name);
        if (it == seg.null_vectors.end()) return make_op(not_null ? OpKind::MatchAll : OpKind::Empty);
        return bitmap_operator(it->second, not_null);
      }
      // a raw multi-value column is evaluated on its internal dictionary-encoded twin (built at registration): the same docs match, and
      // the scan counts the same entries (MVScanDocIdIterator over a raw column counts entries as well)
      if (col->raw_mv) col = col->vdict.get();
      if (col->col_kind == PG_COL_VAR_BYTES && !col->has_dictionary && !col->is_mv && col->data_type == PG_TYPE_STRING) {
        // a raw STRING column: through its virtual dictionary (built once per column), as a dictId scan over the ids
        if (nh) fail(PG_ERR_UNSUPPORTED, "predicate over the raw STRING column %s under enableNullHandling", col->name.c_str());
        ensure_virtual_dictionary(seg, *col);
        Column* twin = col->vdict.get();
        return leaf_operator(make_raw_string_eval(f, *twin), twin, f.predicate_type);
      }
      PredEval ev = make_pred_eval(f, *col);
      // FilterOperatorUtils.java:78-88: under null handling an always-true predicate matches the docs that hold a value
      if (nh && ev.always_true && !ev.always_false)
        if (auto nv = null_vector_of(seg, col)) return bitmap_operator(nv, true);
      return leaf_operator(std::move(ev), col, f.predicate_type);
    }
    case PG_FILTER_CONSTANT_TRUE: return make_op(OpKind::MatchAll);
    case PG_FILTER_CONSTANT_FALSE: return make_op(OpKind::Empty);
    default: fail(PG_ERR_INVALID_ARGUMENT, "bad filter node type %d", f.type);
  }
}

// ---- query-level null handling (QueryContext#isNullHandlingEnabled) ------------------------------------------------------------------
// The reference evaluates the operator tree in three-valued logic through getTrues / getNulls / getFalses: a column leaf (Scan / Inverted /
// Sorted / RangeIndex: BaseColumnFilterOperator.java:45-72) is true where its predicate holds AND the value is not null, null where the value
// is null; AND / OR / NOT / bitmap leaves have no nulls of their own (BaseFilterOperator.java:98-100); falses = NOT(trues OR nulls)
// (BaseFilterOperator.java:105-122), AND / OR build theirs from their children's trues and nulls (AndFilterOperator.java:62-90,
// OrFilterOperator.java:61-87), NOT swaps the two (NotFilterOperator.java:52-63).  nh_trues / nh_falses restate that as a rewrite into the
// two-valued operators the emitter knows (the docId sets the reference builds are And / Or / Not sets over bitmaps as well).
static bool column_leaf(const FilterOp& op) {
  return (op.kind == OpKind::Scan || op.kind == OpKind::Inverted || op.kind == OpKind::Sorted || op.kind == OpKind::RangeIdx) && !op.bitmap_col;
}
static OpPtr node_of(OpKind k, std::vector<OpPtr> ch) {
  auto r = make_op(k);
  r->children = std::move(ch);
  return r;
}
static OpPtr nh_falses(Segment& seg, OpPtr op);
static OpPtr nh_trues(Segment& seg, OpPtr op) {
  switch (op->kind) {
    case OpKind::And:
    case OpKind::Or: {
      const OpKind k = op->kind;
      std::vector<OpPtr> ch;
      for (auto& c : op->children) ch.push_back(nh_trues(seg, std::move(c)));
      return node_of(k, std::move(ch));
    }
    case OpKind::Not:
      if (op->children[0]->kind == OpKind::Empty) return make_op(OpKind::MatchAll);   // isResultEmpty
      return nh_falses(seg, std::move(op->children[0]));
    default:
      if (column_leaf(*op))
        if (auto nv = null_vector_of(seg, op->col)) {   // excludeNulls: AndDocIdSet(block, flip(nullBitmap))
          std::vector<OpPtr> ch;
          ch.push_back(std::move(op));
          ch.push_back(bitmap_operator(nv, true));
          return node_of(OpKind::And, std::move(ch));
        }
      return op;
  }
}
static OpPtr nh_falses(Segment& seg, OpPtr op) {
  switch (op->kind) {
    case OpKind::Not: return nh_trues(seg, std::move(op->children[0]));
    case OpKind::And:
    case OpKind::Or: {
      const bool is_and = op->kind == OpKind::And;
      std::vector<OpPtr> xs;
      for (auto& c : op->children) {
        std::shared_ptr<Column> nv = column_leaf(*c) ? null_vector_of(seg, c->col) : nullptr;
        OpPtr t = nh_trues(seg, std::move(c));
        if (is_and) {
          if (t->kind == OpKind::Empty) return make_op(OpKind::MatchAll);
          if (t->kind == OpKind::MatchAll) continue;
        } else {
          if (t->kind == OpKind::MatchAll) return make_op(OpKind::Empty);
          if (t->kind == OpKind::Empty) continue;
        }
        if (nv) {   // the child's nulls are not false either
          std::vector<OpPtr> both;
          both.push_back(std::move(t));
          both.push_back(bitmap_operator(nv, false));
          t = node_of(OpKind::Or, std::move(both));
        }
        xs.push_back(std::move(t));
      }
      if (xs.empty()) return make_op(is_and ? OpKind::Empty : OpKind::MatchAll);
      OpPtr inner = xs.size() == 1 ? std::move(xs[0]) : node_of(is_and ? OpKind::And : OpKind::Or, std::move(xs));
      std::vector<OpPtr> one;
      one.push_back(std::move(inner));
      return node_of(OpKind::Not, std::move(one));
    }
    default: {
      std::shared_ptr<Column> nv = column_leaf(*op) ? null_vector_of(seg, op->col) : nullptr;
      OpPtr t = nh_trues(seg, std::move(op));
      if (t->kind == OpKind::MatchAll) return make_op(OpKind::Empty);
      if (nv) {
        std::vector<OpPtr> both;
        both.push_back(std::move(t));
        both.push_back(bitmap_operator(nv, false));
        t = node_of(OpKind::Or, std::move(both));
      } else if (t->kind == OpKind::Empty) {
        return make_op(OpKind::MatchAll);
      }
      std::vector<OpPtr> one;
      one.push_back(std::move(t));
      return node_of(OpKind::Not, std::move(one));
    }
  }
}

// =====================================================================================================================
// program emission
// =====================================================================================================================
struct Emitter {
  Segment& seg;
  CompiledPlan& plan;
  std::vector<PgFInstr> instrs;
  std::vector<PgScanLeaf> scans;
  std::vector<PgPostingLeaf> postings;
  std::vector<PgRangeLeaf> ranges;
  std::vector<PgRangeIdxLeaf> rangeidx;
  int sp = 0, max_sp = 0;
  int64_t alg_bytes = 0;
  std::vector<Column*> scanned_cols;

  void push() { sp++; max_sp = std::max(max_sp, sp); }
  int64_t plan_wtiles() const { return ((int64_t)seg.total_docs + PG_WAVE_DOCS - 1) / PG_WAVE_DOCS; }
  template <typename T> const T* keep(const std::vector<T>& v) {
    plan.keep.push_back(upload_vector(v));
    return plan.keep.back().as<T>();
  }

  void emit_ranges(std::vector<int32_t> lo, std::vector<int32_t> hi) {
    PgRangeLeaf L{};
    L.n = (int32_t)lo.size();
    if (lo.size() > 64) {   // many ranges (star-tree traversals): hand the kernels match words instead of a range list
      std::vector<uint32_t> words((size_t)plan_wtiles() * 64 + 64, 0);
      for (size_t r = 0; r < lo.size(); r++)
        for (int64_t d = lo[r]; d <= hi[r];) {
          const int64_t w = d >> 5;
          const int64_t last = std::min<int64_t>(hi[r], w * 32 + 31);
          words[(size_t)w] |= (0xFFFFFFFFu << (d & 31)) & (0xFFFFFFFFu >> (31 - (last & 31)));
          d = last + 1;
        }
      L.words = keep(words);
      alg_bytes += ((int64_t)seg.total_docs + 7) / 8;
    }
    L.lo = keep(lo);
    L.hi = keep(hi);
    ranges.push_back(L);
    instrs.push_back({L.words ? PG_F_PUSH_WORDS : PG_F_PUSH_RANGES, (int32_t)ranges.size() - 1});
    push();
  }

  // SortedIndexBasedFilterOperator#getTrues (operator/filter/SortedIndexBasedFilterOperator.java:57-130)
  void emit_sorted(const FilterOp& op) {
    const Column& c = *op.col;
    const PredEval& e = op.eval;
    std::vector<int32_t> lo, hi;
    if (e.is_range) {
      lo.push_back(c.sorted_start[e.start_dict_id]);
      hi.push_back(c.sorted_end[e.end_dict_id - 1]);
    } else {
      const std::vector<int32_t>& ids = e.exclusive ? e.non_matching : e.matching;
      std::vector<int32_t> rl, rh;
      for (int32_t id : ids) {
        int32_t s = c.sorted_start[id], en = c.sorted_end[id];
        if (!rl.empty() && s == rh.back() + 1) rh.back() = en;
        else { rl.push_back(s); rh.push_back(en); }
      }
      if (!e.exclusive) { lo = rl; hi = rh; }
      else {
        if (rl[0] > 0) { lo.push_back(0); hi.push_back(rl[0] - 1); }
        for (size_t i = 0; i + 1 < rl.size(); i++) { lo.push_back(rh[i] + 1); hi.push_back(rl[i + 1] - 1); }
        if (rh.back() < seg.total_docs - 1) { lo.push_back(rh.back() + 1); hi.push_back(seg.total_docs - 1); }
      }
    }
    emit_ranges(std::move(lo), std::move(hi));
  }

  // InvertedIndexFilterOperator (operator/filter/InvertedIndexFilterOperator.java:60-96): OR of the posting lists of the
  // (non-)matching dictIds, flipped over [0, numDocs) when exclusive
  void emit_inverted(const FilterOp& op) {
    Column& c = *op.col;
    if (op.bitmap_col) plan.pinned.push_back(op.bitmap_col);
    const PredEval& e = op.eval;
    const std::vector<int32_t>& ids = e.exclusive ? e.non_matching : e.matching;
    const int32_t n_chunks = (seg.n_tiles + PG_TILES_PER_CHUNK - 1) / PG_TILES_PER_CHUNK;
    // Dense postings: the leading containers of a dictId that are bitmap containers of chunks 0, 1, 2 ... stored back to
    // back (they are copied in key order at registration) are addressed as base + 8 KB * chunk by the kernels.
    auto dense_prefix = [&](int32_t id) {
      int32_t n = 0;
      for (uint32_t k = c.posting_begin[id]; k < c.posting_begin[id + 1]; k++, n++) {
        const PgContainer& pc = c.descs_host[k];
        if (pc.type != 1 || pc.key != (uint16_t)n || pc.offset != c.descs_host[c.posting_begin[id]].offset + (uint64_t)n * 8192) break;
      }
      return n;
    };
    const int32_t want = std::max(1, (int32_t)(((int64_t)seg.total_docs + PG_CHUNK_DOCS - 1) / PG_CHUNK_DOCS) - 1);
    std::vector<int32_t> dense_ids;
    int32_t dense_chunks = 0;
    for (int32_t id : ids) {
      if ((int)dense_ids.size() >= PG_MAX_DENSE) break;
      const int32_t n = dense_prefix(id);
      if (n >= want) {
        dense_chunks = dense_ids.empty() ? n : std::min(dense_chunks, n);
        dense_ids.push_back(id);
      }
    }
    auto is_dense = [&](int32_t id, const PgContainer& pc) {
      return pc.key < dense_chunks && std::find(dense_ids.begin(), dense_ids.end(), id) != dense_ids.end();
    };
    std::vector<uint32_t> chunk_start((size_t)n_chunks + 2, 0);
    for (int32_t id : ids)
      for (uint32_t k = c.posting_begin[id]; k < c.posting_begin[id + 1]; k++)
        if (!is_dense(id, c.descs_host[k])) chunk_start[c.descs_host[k].key + 1]++;
    for (size_t i = 1; i < chunk_start.size(); i++) chunk_start[i] += chunk_start[i - 1];
    std::vector<PgContainer> entries(chunk_start.back());
    std::vector<uint32_t> cursor(chunk_start.begin(), chunk_start.end() - 1);
    for (int32_t id : ids)
      for (uint32_t k = c.posting_begin[id]; k < c.posting_begin[id + 1]; k++) {
        const PgContainer& pc = c.descs_host[k];
        if (!is_dense(id, pc)) entries[cursor[pc.key]++] = pc;
        alg_bytes += pc.type == 1 ? 8192 : (pc.type == 0 ? 2 * (int64_t)pc.n : 4 * (int64_t)pc.n);
      }
    PgPostingLeaf L{};
    L.containers = c.containers_dev.as<uint8_t>();
    L.has_csr = entries.empty() ? 0 : 1;
    L.chunk_start = keep(chunk_start);
    L.entries = keep(entries);
    L.exclusive = e.exclusive ? 1 : 0;
    L.n_dense = (int32_t)dense_ids.size();
    L.dense_chunks = dense_ids.empty() ? 0 : dense_chunks;
    for (size_t j = 0; j < PG_MAX_DENSE; j++)   // unused slots repeat slot 0: the kernels OR all eight unconditionally
      L.dense[j] = dense_ids.empty() ? c.containers_dev.as<uint8_t>()
                                     : c.containers_dev.as<uint8_t>() + c.descs_host[c.posting_begin[dense_ids[j < dense_ids.size() ? j : 0]]].offset;
    postings.push_back(L);
    instrs.push_back({PG_F_PUSH_POSTINGS, (int32_t)postings.size() - 1});
    push();
  }

  // RangeIndexBasedFilterOperator#getMatchingDocIds over BitSlicedRangeIndexReader#getMatchingDocIds (:41-246): the predicate's
  // inclusive bounds in the index's stored domain — dictIds, value - min, FPOrdering ordinals — as  lte(hi) AND NOT lte(lo - 1)
  static uint64_t fp_ordinal(double v, bool is_float) {   // FPOrdering.ordinalOf
    if (is_float) {
      const float f = (float)v;
      if (f == INFINITY) return 0xFFFFFFFFULL;
      if (f == -INFINITY || f != f) return 0;
      uint32_t b;
      memcpy(&b, &f, 4);
      b = (b & 0x80000000u) ? (b == 0x80000000u ? 0x80000000u : ~b) : (b ^ 0x80000000u);
      return b;
    }
    if (v == (double)INFINITY) return ~0ULL;
    if (v == -(double)INFINITY || v != v) return 0;
    uint64_t b;
    memcpy(&b, &v, 8);
    return (b & (1ULL << 63)) ? (b == (1ULL << 63) ? (1ULL << 63) : ~b) : (b ^ (1ULL << 63));
  }
  void emit_rangeidx(const FilterOp& op) {
    const Column& c = *op.col;
    const PredEval& e = op.eval;
    const bool eq = e.pred_type == PG_PRED_EQ;
    const uint64_t range_mask = c.ri_slices == 64 ? ~0ULL : ((1ULL << c.ri_slices) - 1ULL);
    bool empty = false;
    uint64_t lo = 0, hi = 0;
    if (e.dictionary_based) {
      lo = (uint64_t)(eq ? e.matching[0] : e.start_dict_id);
      hi = (uint64_t)(eq ? e.matching[0] : e.end_dict_id - 1);
    } else if (c.val_type == PG_V_I32 || c.val_type == PG_V_I64) {
      const int64_t lo_v = eq ? e.set_i[0] : e.lo_i, hi_v = eq ? e.set_i[0] : e.hi_i;
      if (lo_v > hi_v || hi_v < c.ri_min) empty = true;
      else {
        lo = (uint64_t)((__int128)std::max(lo_v, c.ri_min) - c.ri_min);
        hi = (uint64_t)((__int128)hi_v - c.ri_min);
      }
    } else {
      const double lo_v = eq ? e.set_d[0] : e.lo_d, hi_v = eq ? e.set_d[0] : e.hi_d;
      if (lo_v > hi_v) empty = true;
      else { lo = fp_ordinal(lo_v, c.val_type == PG_V_F32); hi = fp_ordinal(hi_v, c.val_type == PG_V_F32); }
    }
    if (!empty && lo > range_mask) empty = true;   // beyond every stored value
    if (empty) { instrs.push_back({PG_F_PUSH_NONE, 0}); push(); return; }
    PgRangeIdxLeaf L{};
    L.containers = c.ri_containers_dev.as<uint8_t>();
    L.descs = c.ri_descs_dev.as<PgContainer>();
    L.n_slices = c.ri_slices;
    L.has_hi = hi < range_mask ? 1 : 0;
    L.hi = hi < range_mask ? hi : range_mask;
    L.has_lo = lo > 0 ? 1 : 0;
    L.lo_m1 = lo > 0 ? lo - 1 : 0;
    rangeidx.push_back(L);
    alg_bytes += (int64_t)c.ri_bytes;
    instrs.push_back({PG_F_PUSH_RANGEIDX, (int32_t)rangeidx.size() - 1});
    push();
  }

  void emit_scan(const FilterOp& op, bool masked) {
    Column& c = *op.col;
    if (c.col_kind == PG_COL_VAR_BYTES) fail(PG_ERR_UNSUPPORTED, "predicate over the raw STRING / BYTES column %s", c.name.c_str());
    const PredEval& e = op.eval;
    PgScanLeaf L{};
    L.data = c.fwd_dev.as<uint8_t>();
    L.col_kind = c.col_kind;
    L.bits = c.bits;
    L.val_type = c.val_type;
    if (e.dictionary_based) {
      if (e.is_range) { L.pred_kind = PG_P_RANGE; L.lo = e.start_dict_id; L.hi = e.end_dict_id - 1; }
      else if (e.matching.size() == 1) { L.pred_kind = PG_P_RANGE; L.lo = L.hi = e.matching[0]; }
      else {
        L.pred_kind = PG_P_DICT_LUT;
        std::vector<uint32_t> lut(((size_t)c.cardinality + 31) / 32 + 1, 0);
        for (int32_t d : e.matching) lut[d >> 5] |= 1u << (d & 31);
        L.lut = keep(lut);
      }
    } else if (e.pred_type == PG_PRED_RANGE) {
      L.pred_kind = PG_P_RANGE;
      if (c.val_type == PG_V_I32 || c.val_type == PG_V_I64) { L.lo = e.lo_i; L.hi = e.hi_i; }
      else { memcpy(&L.lo, &e.lo_d, 8); memcpy(&L.hi, &e.hi_d, 8); }
    } else {
      L.pred_kind = PG_P_SET;
      L.exclusive = e.exclusive ? 1 : 0;
      if (c.val_type == PG_V_I32 || c.val_type == PG_V_I64) { L.n_set = (int32_t)e.set_i.size(); L.set_values = keep(e.set_i); }
      else { L.n_set = (int32_t)e.set_d.size(); L.set_values = keep(e.set_d); }
    }
    if (c.is_mv) {   // MVScanDocIdIterator: ANY entry passes (ALL for the exclusive predicates); every entry of an evaluated doc counts
      if (!e.dictionary_based) fail(PG_ERR_UNSUPPORTED, "raw multi-value column %s", c.name.c_str());
      L.mv = 1;
      L.exclusive = e.exclusive ? 1 : 0;
      L.set_values = c.mv_offsets_dev.ptr;
      plan.dev.mv = 1;
    }
    if (masked) {
      if (plan.n_stat_slots >= PG_MAX_STATS) fail(PG_ERR_UNSUPPORTED, "more than %d restricted scans in one filter", PG_MAX_STATS - 1);
      L.stat_slot = plan.n_stat_slots++;
    } else {
      plan.full_scan_entries += c.is_mv ? c.total_entries : seg.total_docs;
    }
    if (std::find(scanned_cols.begin(), scanned_cols.end(), &c) == scanned_cols.end()) {
      scanned_cols.push_back(&c);
      alg_bytes += (int64_t)c.fwd_bytes_logical;
    }
    scans.push_back(L);
    instrs.push_back({masked ? PG_F_AND_SCAN : PG_F_PUSH_SCAN, (int32_t)scans.size() - 1});
    if (!masked) push();
  }

  static bool has_scan(const FilterOp& op) {
    if (op.kind == OpKind::Scan) return true;
    for (auto& c : op.children) if (has_scan(*c)) return true;
    return false;
  }

  void emit(const FilterOp& op, bool top_level) {
    switch (op.kind) {
      case OpKind::Empty: instrs.push_back({PG_F_PUSH_NONE, 0}); push(); break;
      case OpKind::MatchAll: instrs.push_back({PG_F_PUSH_ALL, 0}); push(); break;
      case OpKind::Sorted: emit_sorted(op); break;
      case OpKind::Bitmap: emit_ranges(op.range_lo, op.range_hi); break;
      case OpKind::Inverted: emit_inverted(op); break;
      case OpKind::RangeIdx: emit_rangeidx(op); break;
      case OpKind::Scan: emit_scan(op, false); break;
      case OpKind::Not:
        emit(*op.children[0], false);
        instrs.push_back({PG_F_NOT, 0});
        if (has_scan(op)) plan.stats_exact = false;
        break;
      case OpKind::Or:
        for (size_t i = 0; i < op.children.size(); i++) {
          emit(*op.children[i], false);
          if (i) { instrs.push_back({PG_F_OR, 0}); sp--; }
        }
        // OrDocIdIterator drains every scan child fully only when the OR itself is drained (top level)
        if (!top_level && has_scan(op)) plan.stats_exact = false;
        for (auto& c : op.children) if (c->kind != OpKind::Scan && has_scan(*c)) plan.stats_exact = false;
        break;
      case OpKind::And: {
        // AndDocIdSet.iterator(): index-based children → bitmap; scans applyAnd() on the survivors in list order;
        // the remaining (nested) children are intersected last.
        std::vector<const FilterOp*> index_based, scan_based, remaining;
        for (auto& c : op.children) {
          if (index_child(*c)) index_based.push_back(c.get());   // leaves, and compound children whose iterator is bitmap based
          else if (c->kind == OpKind::Scan) scan_based.push_back(c.get());
          else remaining.push_back(c.get());
        }
        // nested ANDs first, plain index leaves after them: AND(AND(index.., scans..), queryableDocIds) then reads
        // [index..][AND_SCAN..][PUSH_POSTINGS, AND], the form the chain kernels take (set algebra does not care about the order)
        std::stable_partition(index_based.begin(), index_based.end(), [](const FilterOp* c) { return c->kind == OpKind::And; });
        bool first = true;
        for (auto* c : index_based) {
          emit(*c, false);
          if (!first) { instrs.push_back({PG_F_AND, 0}); sp--; }
          first = false;
        }
        if (index_based.empty()) plan.stats_exact = plan.stats_exact && scan_based.empty() && true;
        for (auto* c : scan_based) {
          if (first) {
            emit_scan(*c, false);            // AndDocIdIterator leapfrog in the reference: counts are data dependent
            plan.stats_exact = false;
            first = false;
          } else {
            emit_scan(*c, true);
          }
        }
        for (auto* c : remaining) {
          emit(*c, false);
          if (has_scan(*c)) plan.stats_exact = false;
          if (!first) { instrs.push_back({PG_F_AND, 0}); sp--; }
          first = false;
        }
        if (!top_level && has_scan(op) && !yields_bitmap(op)) plan.stats_exact = false;
        break;
      }
    }
  }

  // Which compound operators hand their parent a bitmap-based iterator (the parent AND then restricts its scans by it):
  //  * AndDocIdSet.iterator() (AndDocIdSet.java:125-178) merges index-based and scan children into ONE RangelessBitmapDocIdIterator,
  //    eagerly, when there is an index-based child next to a scan (or two index-based ones) and nothing else — the nested AND of
  //    FilterPlanNode.run's queryableDocIds wrapper; every scan count stays exact;
  //  * OrDocIdSet.iterator() (OrDocIdSet.java:62-125) merges into a BitmapDocIdIterator only with >= 2 SORTED children and no
  //    scan / compound child (its bitmap list is never filled in this reference snapshot: `numSorted + 0 > 1`); any other OR is an
  //    OrDocIdIterator, which an enclosing AND leapfrogs.
  static bool index_child(const FilterOp& c) {
    return c.kind == OpKind::Sorted || c.kind == OpKind::Inverted || c.kind == OpKind::Bitmap || c.kind == OpKind::RangeIdx || yields_bitmap(c);
  }
  static bool yields_bitmap(const FilterOp& op) {
    if (op.kind == OpKind::And) {
      int n_index = 0, n_scan = 0;
      for (auto& c : op.children) {
        if (index_child(*c)) n_index++;
        else if (c->kind == OpKind::Scan) n_scan++;
        else return false;
      }
      return (n_index > 0 && n_scan > 0) || n_index > 1;
    }
    if (op.kind == OpKind::Or) {
      int n_sorted = 0;
      for (auto& c : op.children) {
        if (c->kind == OpKind::Sorted) n_sorted++;
        else if (!index_child(*c)) return false;
      }
      return n_sorted > 1;
    }
    return false;
  }
};

// =====================================================================================================================
// signature (plan cache key)
// =====================================================================================================================
// Every string goes in with its length, so that no two queries share a signature ("a,b" / "c" vs "a" / "b,c" as range bounds);
// NULL strings are legal here (validation happens in compile_plan) and get their own marker.
static void sig_str(std::ostringstream& o, const char* s) {
  if (!s) { o << "~;"; return; }
  o << strlen(s) << "'" << s << ";";
}
static void sig_filter(std::ostringstream& o, const pg_filter_node* f) {
  if (!f) { o << "*"; return; }
  o << "(" << f->type;
  if (f->type == PG_FILTER_PREDICATE) {
    o << ":" << f->predicate_type << ":";
    sig_str(o, f->column);
    if (f->predicate_type == PG_PRED_RANGE) {
      o << (f->lower_inclusive ? "[" : "(");
      sig_str(o, f->lower);
      sig_str(o, f->upper);
      o << (f->upper_inclusive ? "]" : ")");
    } else {
      o << f->n_values << ":";
      for (int i = 0; i < f->n_values && f->values; i++) sig_str(o, f->values[i]);
    }
  }
  for (int i = 0; i < f->n_children && f->children; i++) sig_filter(o, &f->children[i]);
  o << ")";
}
std::string query_signature(const pg_filter_node* filter, const pg_query* q, int32_t flags) {
  std::ostringstream o;
  sig_filter(o, filter);
  o << "|nh" << (((q ? q->flags : flags) & PG_QUERY_FLAG_NULL_HANDLING) ? 1 : 0) << (((q ? q->flags : flags) & kQueryFlagNullPartition) ? 1 : 0);
  if (q) {
    o << "|g" << q->n_group_by << ":";
    for (int i = 0; i < q->n_group_by && q->group_by_columns; i++) sig_str(o, q->group_by_columns[i]);
    o << "|a" << q->n_aggregations << ":";
    for (int i = 0; i < q->n_aggregations && q->aggregations; i++) {
      o << q->aggregations[i].function << "," << q->aggregations[i].log2m << ",";
      sig_str(o, q->aggregations[i].column ? q->aggregations[i].column : "*");
    }
    o << "|" << q->num_groups_limit << "," << q->max_initial_result_holder_capacity << "," << (q->flags & PG_QUERY_FLAG_SKIP_STAR_TREE);
    if (q->n_group_by > 0 && q->n_order_by > 0 && q->order_by && q->min_segment_group_trim_size > 0) {
      o << "|trim" << q->limit << "," << q->min_segment_group_trim_size;
      for (int32_t i = 0; i < q->n_order_by; i++) o << ";" << q->order_by[i].kind << "," << q->order_by[i].index << "," << (q->order_by[i].ascending != 0);
    }
  } else {
    o << "|filter-only";
  }
  return o.str();
}

// =====================================================================================================================
// plan compilation
// =====================================================================================================================
// ---- HyperLogLog (stream-lib 2.9.8, not in the reference tree; algorithm restated from SURVEY.md §9) ---------------------
// offer(o): x = MurmurHash.hash(o); j = x >>> (32 - log2m); r = numberOfLeadingZeros((x << log2m) | (1 << (log2m-1)) + 1) + 1
// MurmurHash.hash(Integer / Long) = hashLong(value); Float → raw int bits (sign-extended); Double → raw long bits;
// String / bytes → MurmurHash2 over the bytes with seed -1.
static uint32_t murmur_hash_long(int64_t data) {
  const uint32_t m = 0x5bd1e995u;
  uint32_t h = 0;
  uint32_t k = (uint32_t)(uint64_t)data * m;
  k ^= k >> 24;
  h ^= k * m;
  k = (uint32_t)((uint64_t)data >> 32) * m;
  k ^= k >> 24;
  h *= m;
  h ^= k * m;
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}
static uint32_t murmur_hash_bytes(const uint8_t* data, int32_t length) {
  const uint32_t m = 0x5bd1e995u;
  uint32_t h = 0xFFFFFFFFu ^ (uint32_t)length;
  const int len4 = length >> 2;
  for (int i = 0; i < len4; i++) {
    const int i4 = i << 2;
    uint32_t k = (uint32_t)data[i4] | ((uint32_t)data[i4 + 1] << 8) | ((uint32_t)data[i4 + 2] << 16) | ((uint32_t)data[i4 + 3] << 24);
    k *= m; k ^= k >> 24; k *= m;
    h *= m; h ^= k;
  }
  const int left = length - (len4 << 2);
  if (left != 0) {
    if (left >= 3) h ^= (uint32_t)((int32_t)(int8_t)data[length - 3] << 16);
    if (left >= 2) h ^= (uint32_t)((int32_t)(int8_t)data[length - 2] << 8);
    if (left >= 1) h ^= (uint32_t)(int32_t)(int8_t)data[length - 1];
    h *= m;
  }
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}
static uint32_t hll_index_rank(uint32_t x, int log2m) {   // register index | rank << 16
  const uint32_t j = x >> (32 - log2m);
  const uint32_t w = (x << log2m) | ((1u << (log2m - 1)) + 1u);
  const uint32_t r = (uint32_t)__builtin_clz(w) + 1u;
  return j | (r << 16);
}
// (index, rank) of every dictionary value: what DistinctCountHLLAggregationFunction.java:457-466 computes per dictId.
static const uint32_t* hll_dict_lut(Column& c, int log2m) {
  auto it = c.hll_luts.find(log2m);
  if (it != c.hll_luts.end()) return it->second.as<uint32_t>();
  std::vector<uint32_t> lut((size_t)c.cardinality);
  const uint8_t* d = c.dict_host.data();
  for (int32_t i = 0; i < c.cardinality; i++) {
    uint32_t x;
    switch (c.data_type) {
      case PG_TYPE_INT: x = murmur_hash_long((int64_t)(int32_t)be32(d + (size_t)i * 4)); break;
      case PG_TYPE_LONG: x = murmur_hash_long((int64_t)be64(d + (size_t)i * 8)); break;
      case PG_TYPE_FLOAT: x = murmur_hash_long((int64_t)(int32_t)be32(d + (size_t)i * 4)); break;
      case PG_TYPE_DOUBLE: x = murmur_hash_long((int64_t)be64(d + (size_t)i * 8)); break;
      default: {
        const uint8_t* e = d + (size_t)i * c.dict_bytes_per_value;
        int len = c.dict_bytes_per_value;
        while (len > 0 && e[len - 1] == 0) len--;
        x = murmur_hash_bytes(e, len);
        break;
      }
    }
    lut[i] = hll_index_rank(x, log2m);
  }
  auto ins = c.hll_luts.emplace(log2m, upload_vector(lut));
  return ins.first->second.as<uint32_t>();
}

// HyperLogLog registers after offering every dictionary value (NonScanBasedAggregationOperator#getDistinctCountHLLResult)
void hll_registers_of_dictionary(Column& c, int log2m, uint8_t* regs) {
  std::vector<uint32_t> lut((size_t)c.cardinality);
  const uint8_t* d = c.dict_host.data();
  memset(regs, 0, (size_t)1 << log2m);
  for (int32_t i = 0; i < c.cardinality; i++) {
    uint32_t x;
    switch (c.data_type) {
      case PG_TYPE_INT: x = murmur_hash_long((int64_t)(int32_t)be32(d + (size_t)i * 4)); break;
      case PG_TYPE_LONG: x = murmur_hash_long((int64_t)be64(d + (size_t)i * 8)); break;
      case PG_TYPE_FLOAT: x = murmur_hash_long((int64_t)(int32_t)be32(d + (size_t)i * 4)); break;
      case PG_TYPE_DOUBLE: x = murmur_hash_long((int64_t)be64(d + (size_t)i * 8)); break;
      default: {
        const uint8_t* e = d + (size_t)i * c.dict_bytes_per_value;
        int len = c.dict_bytes_per_value;
        while (len > 0 && e[len - 1] == 0) len--;
        x = murmur_hash_bytes(e, len);
        break;
      }
    }
    const uint32_t ir = hll_index_rank(x, log2m);
    uint8_t& r = regs[ir & 0xFFFFu];
    if ((ir >> 16) > r) r = (uint8_t)(ir >> 16);
  }
}
double dictionary_value_as_double(const Column& c, int32_t dict_id) {   // Dictionary#getDoubleValue
  const uint8_t* d = c.dict_host.data();
  switch (c.data_type) {
    case PG_TYPE_INT: return (double)(int32_t)be32(d + (size_t)dict_id * 4);
    case PG_TYPE_LONG: return (double)(int64_t)be64(d + (size_t)dict_id * 8);
    case PG_TYPE_FLOAT: { uint32_t u = be32(d + (size_t)dict_id * 4); float f; memcpy(&f, &u, 4); return (double)f; }
    default: { uint64_t u = be64(d + (size_t)dict_id * 8); double f; memcpy(&f, &u, 8); return f; }
  }
}

double limbs_to_double(const int64_t* limbs, int n_limbs, int q) { return pg_limbs_to_double(limbs, n_limbs, q); }

static const int64_t kLdsTableBudget = 144 * 1024;      // bytes of LDS for the accumulator table (one workgroup per CU)
static const int64_t kLdsReplicaBudget = 96 * 1024;
static const size_t kMaxAuxBytes = (size_t)2 << 30;
static const int64_t kMaxDenseGroups = 64LL << 20;      // dense HBM table limit (groups)

// Fraction of the docs a filter is expected to pass: exact for index leaves (posting cardinalities, sorted ranges), independence
// for AND / OR, 1/5 per raw-value scan whose outcome is unknown at plan time.  Only steers a choice between two correct plans.
static double estimate_selectivity(const FilterOp& op, double n_docs) {
  if (n_docs <= 0) return 0;
  switch (op.kind) {
    case OpKind::Empty: return 0;
    case OpKind::MatchAll: return 1;
    case OpKind::Inverted: {
      if (op.bitmap_col && op.eval.exclusive) return std::max(0.0, 1.0 - (double)op.col->posting_card[0] / n_docs);
      double m = 0;
      for (int32_t id : op.eval.matching) m += (double)op.col->posting_card[(size_t)id];
      return std::min(1.0, m / n_docs);
    }
    case OpKind::Sorted: {
      double m = 0;
      for (int32_t id : op.eval.matching) m += (double)(op.col->sorted_end[(size_t)id] - op.col->sorted_start[(size_t)id] + 1);
      return std::min(1.0, m / n_docs);
    }
    case OpKind::Bitmap: {
      double m = 0;
      for (size_t i = 0; i < op.range_lo.size(); i++) m += (double)(op.range_hi[i] - op.range_lo[i] + 1);
      return std::min(1.0, m / n_docs);
    }
    case OpKind::RangeIdx:
      if (op.eval.dictionary_based && op.col->cardinality > 0) return (double)op.eval.matching.size() / op.col->cardinality;
      return 0.3;
    case OpKind::Scan:
      if (op.eval.dictionary_based && op.col->cardinality > 0) return (double)op.eval.matching.size() / op.col->cardinality;   // uniform dictIds
      return 0.2;   // a raw-value predicate without column statistics: assume it is selective (the dense HBM table is the safe side)
    case OpKind::And: { double s = 1; for (auto& c : op.children) s *= estimate_selectivity(*c, n_docs); return s; }
    case OpKind::Or: { double s = 1; for (auto& c : op.children) s *= 1 - estimate_selectivity(*c, n_docs); return 1 - s; }
    case OpKind::Not: return 1 - estimate_selectivity(*op.children[0], n_docs);
  }
  return 0.5;
}

// canOptimizeCount (BaseFilterOperator.java:56-82 and overrides): index-only filters whose cardinality needs no scan
static bool can_optimize_count(const FilterOp& op) {
  switch (op.kind) {
    case OpKind::Scan: return false;
    case OpKind::And: case OpKind::Or: case OpKind::Not:
      for (auto& c : op.children) if (!can_optimize_count(*c)) return false;
      return true;
    default: return true;
  }
}

static std::shared_ptr<CompiledPlan> compile_in_space(Segment& seg, OpPtr root, const pg_query* q, const StarTree* st, int star_index);

// `seg`: the segment the query addresses.  AggregationPlanNode#buildNonFilteredAggOperator (:97-127) / GroupByPlanNode: the
// regular filter is planned first; FastFilteredCountOperator takes a lone COUNT(*) over an index-only filter; otherwise, when the
// filter result is not empty, the first star-tree the query fits answers it (AggregationFunctionUtils#buildAggregationInfo
// :285-307) and the operators run over that star-tree's doc space.
std::shared_ptr<CompiledPlan> compile_plan(Segment& seg, const pg_filter_node* filter, const pg_query* q, int32_t flags) {
  const bool nh = ((q ? q->flags : flags) & PG_QUERY_FLAG_NULL_HANDLING) != 0;
  OpPtr root = filter ? construct(seg, *filter, nh) : make_op(OpKind::MatchAll);
  if (seg.queryable_doc_ids) {   // FilterPlanNode.run (:88-106): AND(filter, BitmapBasedFilterOperator(queryableDocIds))
    std::vector<OpPtr> both;
    both.push_back(std::move(root));
    both.push_back(bitmap_operator(seg.queryable_doc_ids, false));
    root = and_operator(std::move(both));
  }
  // the root's getTrues in three-valued logic, as two-valued operators — except under FastFilteredCountOperator, which asks the operators for
  // getNumMatchingDocs / getBitmaps: those know no nulls (AggregationPlanNode.java:104-108, InvertedIndexFilterOperator.java:103-131,
  // AndFilterOperator.java:99-110), so a lone COUNT(*) over an index-only filter counts the docs whose stored default value matches as well
  if (nh) {
    const bool fast_count = q && q->n_group_by == 0 && q->n_aggregations == 1 && q->aggregations && q->aggregations[0].function == PG_AGG_COUNT &&
                            (!q->aggregations[0].column || !strcmp(q->aggregations[0].column, "*") || !null_vector_of(seg, seg.find(q->aggregations[0].column))) &&
                            can_optimize_count(*root) && !(q->flags & kQueryFlagNullPartition);
    if (!fast_count) root = nh_trues(seg, std::move(root));
  }
  // AggregationPlanNode#buildNonFilteredAggOperator (:97-127): FastFilteredCountOperator, then NonScanBasedAggregationOperator, then
  // the star-trees — a match-all MIN / MAX / DISTINCTCOUNT(HLL) over dictionary columns is answered from the dictionaries even when a
  // star-tree holds the pair (same values; numDocsScanned / star-tree statistics follow the reference)
  bool non_scan_fit = q && q->n_aggregations > 0 && q->aggregations && q->n_group_by == 0 && root->kind == OpKind::MatchAll &&
                      !(q->n_aggregations == 1 && q->aggregations[0].function == PG_AGG_COUNT);
  for (int i = 0; non_scan_fit && i < q->n_aggregations; i++) {
    const pg_agg_spec& a = q->aggregations[i];
    if (a.function == PG_AGG_COUNT) continue;
    Column* c = seg.find(a.column);
    const int32_t f = sv_function_of(a.function);   // DICTIONARY_BASED_FUNCTIONS holds the MV forms of these as well (AggregationPlanNode.java:51-55)
    const bool dict_fn = a.function != PG_AGG_COUNTMV && (f == PG_AGG_MIN || f == PG_AGG_MAX || f == PG_AGG_MINMAXRANGE ||
                                                          f == PG_AGG_DISTINCTCOUNT || f == PG_AGG_DISTINCTCOUNTHLL);
    non_scan_fit = c && dict_fn && c->has_dictionary && is_mv_function(a.function) == c->is_mv &&
                   (c->data_type <= PG_TYPE_DOUBLE || f == PG_AGG_DISTINCTCOUNT || f == PG_AGG_DISTINCTCOUNTHLL);
  }
  // StarTreeUtils.java:381-418: under null handling a star-tree answers only if no column the query reads holds a null in this segment
  bool star_tree_blocked = nh && q && (q->flags & kQueryFlagNullPartition) != 0;   // a part of a query that reads a column with nulls
  if (nh && q && !seg.star_trees.empty() && !star_tree_blocked) {
    std::function<bool(const pg_filter_node*)> filter_has_nulls = [&](const pg_filter_node* f) -> bool {
      if (!f) return false;
      if (f->type == PG_FILTER_PREDICATE) return f->column && null_vector_of(seg, seg.find(f->column)) != nullptr;
      for (int i = 0; i < f->n_children; i++) if (filter_has_nulls(&f->children[i])) return true;
      return false;
    };
    star_tree_blocked = filter_has_nulls(filter);
    for (int i = 0; i < q->n_group_by && !star_tree_blocked; i++) star_tree_blocked = null_vector_of(seg, seg.find(q->group_by_columns[i])) != nullptr;
    for (int i = 0; i < q->n_aggregations && !star_tree_blocked; i++) {
      const char* c = q->aggregations[i].column;
      star_tree_blocked = c && strcmp(c, "*") != 0 && null_vector_of(seg, seg.find(c)) != nullptr;
    }
  }
  if (!non_scan_fit && q && q->n_aggregations > 0 && q->aggregations && !(q->flags & PG_QUERY_FLAG_SKIP_STAR_TREE) && !seg.star_trees.empty() &&
      root->kind != OpKind::Empty && !star_tree_blocked) {
    const bool fast_count = q->n_group_by == 0 && q->n_aggregations == 1 && q->aggregations[0].function == PG_AGG_COUNT &&
                            can_optimize_count(*root);
    for (size_t t = 0; t < seg.star_trees.size() && !fast_count; t++) {
      OpPtr star_root = star_tree_filter(seg, *seg.star_trees[t], filter, *q);
      if (star_root) return compile_in_space(seg.star_trees[t]->space, std::move(star_root), q, seg.star_trees[t].get(), (int)t);
    }
  }
  return compile_in_space(seg, std::move(root), q, nullptr, -1);
}

// `seg`: the doc space the operators run over — the segment itself, or a star-tree's docs.
static void collect_stat_leaves(const FilterOp& op, std::vector<const FilterOp*>& out) {
  if (op.kind == OpKind::Scan || op.kind == OpKind::Inverted || op.kind == OpKind::RangeIdx) out.push_back(&op);
  for (auto& c : op.children) collect_stat_leaves(*c, out);
}
static OpPtr clone_leaf(const FilterOp& op) {
  auto c = make_op(op.kind);
  c->eval = op.eval;
  c->col = op.col;
  c->bitmap_col = op.bitmap_col;
  c->range_lo = op.range_lo;
  c->range_hi = op.range_hi;
  return c;
}

// Oct-layout kernels (pg_kernels_oct.hip): <= 4 group columns of <= 8 bits, COUNT at most among the accumulators, and ONE DISTINCTCOUNTHLL /
// DISTINCTCOUNT state over a bit-packed (<= 24 bits) dictionary column or a raw INT column.  oct = 1: the state already lives in the
// workgroup's LDS (aux_in_lds: the plan pg_generic_query_l ran before round 4); oct = 2: the partition pipeline's plan whose key space
// (32-bit COUNTs + one floor byte per group) fits LDS — the pruned-offer passes.  Called once every other decision of the plan is taken.
static void plan_oct(CompiledPlan& P, PgQueryPlan& D, const std::vector<Column*>& srcs, int64_t G, int64_t total_docs) {
  D.oct = 0;
  D.oct_dword = 0;
  if (knobs().no_oct) return;   // measurement / test knob: the round-3 kernels
  const bool count_only = D.n_aux == 0 && srcs.empty() && D.n_ops == 1 && D.n_group_cols >= 1;   // COUNT(*) GROUP BY: no source column at all
  if (D.mv || P.first_doc_op >= 0 || (!count_only && (D.n_aux != 1 || srcs.size() != 1)) || D.n_group_cols > 4 || D.n_ops > 1) return;
  for (int g = 0; g < D.n_group_cols; g++)
    if (D.gcols[g].col_kind != PG_COL_FIXED_BIT || D.gcols[g].bits < 1 || D.gcols[g].bits > 8 || D.gcols[g].mult >= ((int64_t)1 << 24) ||
        D.mv_gcol_offsets[g] != nullptr)
      return;
  for (int o = 0; o < D.n_ops; o++)
    if (D.ops[o].fn != PG_ACC_COUNT || D.ops[o].src >= 0) return;
  if (count_only) {
    // The lane-owns-8-docs decode without a source: the group columns cost one dword load per 8 docs and column instead of the quad
    // layout's one per 4, and COUNT is the same ds_add.  Only where the table is LDS-resident and the segment large enough for one
    // 16-wavefront workgroup per CU to have work (PG_OCT_COUNT_MIN_DOCS).
    if (knobs().no_oct_count || total_docs < knobs().oct_count_min_docs || !P.match_all) return;   // behind a filter the fused filter + COUNT kernels win (no mask pass)
    if (D.agg_mode != PG_AGG_LDS && D.agg_mode != PG_AGG_SINGLE) return;
    D.oct_src = -1;
    D.oct_src_kind = 0;
    D.oct_log2m = 0;
    D.oct = 1;
    return;
  }
  const PgAuxOp& A = D.aux[0];
  const Column* c = srcs[(size_t)A.src];
  if (c->is_mv) return;
  int kind = 0;
  if (A.kind == PG_AUX_DICT_SET) {
    if (c->col_kind != PG_COL_FIXED_BIT || c->bits < 1 || c->bits > 24) return;
    kind = 4;
  } else if (A.kind == PG_AUX_HLL_DICT) {
    if (c->col_kind != PG_COL_FIXED_BIT || c->bits < 1 || c->bits > 24 || !c->has_dictionary) return;
    if (c->val_type == PG_V_I32 && c->dict_affine && c->dict_step > 0 && c->dict_step < ((int64_t)1 << 24) && !knobs().oct_no_affine) {
      kind = 1;
      const uint32_t m = 0x5bd1e995u;
      D.oct_c0 = (uint32_t)(int32_t)c->dict_base * m;
      D.oct_c1 = (uint32_t)c->dict_step * m;
      D.oct_nonneg = c->dict_base >= 0 ? 1 : 0;
      D.oct_base = (int32_t)c->dict_base;
      D.oct_step = (int32_t)c->dict_step;
    } else {
      if (!A.lut) return;
      kind = 2;
      D.oct_lut = A.lut;
    }
  } else if (A.kind == PG_AUX_HLL_RAW) {
    if (c->col_kind != PG_COL_RAW32 || c->val_type != PG_V_I32) return;
    kind = 3;
  } else {
    return;
  }
  D.oct_src = A.src;
  D.oct_src_kind = kind;
  D.oct_log2m = A.log2m;
  if (P.aux_in_lds && (D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE) && A.lds_offset >= 0) {
    D.oct = 1;
    // HyperLogLog registers as DWORDS while they live in LDS, if the key space leaves room for four bytes per register (<= ~140 groups
    // of 256 registers): an offer is one ds_max_u32 without a return instead of a read, a compare and a compare-and-swap on the byte's dword
    D.oct_dword = kind != 4 && (int64_t)A.lds_offset + A.rep_bytes * 4 + 64 <= 156 * 1024 && !knobs().oct_byte_regs ? 1 : 0;
    return;
  }
  // pruned offers: HyperLogLog only, one-dword tuples of the partition pipeline, counters + floors of the whole key space in LDS
  // Pruning pays when the groups' registers fill up (floors rise): a source with few distinct values leaves registers at zero for ever
  // and every offer survives every pass — 3 x the cost of the plain partition pipeline (profiles/r04_a: DISTINCTCOUNTHLL over a 16-value
  // column).  What is known at plan time is the column's cardinality / value range: at least 16 values per register.
  const int64_t distinct_hint = c->has_dictionary ? (int64_t)c->cardinality : (c->has_int_range ? c->int_max - c->int_min : 0);
  if (distinct_hint < ((int64_t)16 << A.log2m) && !knobs().oct_any_cardinality) return;
  // ... and enough docs: the floors only rise once a register has seen several offers (pg_exec.hip, oct_pass_bounds) — below ~16 offers
  // per register over the whole segment the passes would forward nearly everything
  const int64_t min_docs = knobs().oct_min_docs >= 0 ? knobs().oct_min_docs : std::max<int64_t>((int64_t)1 << 20, 16 * (G << A.log2m));
  if (D.agg_mode == PG_AGG_RADIX && D.p2 && D.p2_planes == 1 && kind != 4 && D.n_group_cols >= 1 && total_docs >= min_docs &&
      G * 4 + ((G + 3) & ~(int64_t)3) + 16 * 4096 + 512 <= 156 * 1024 &&   /* counters + floors + the wavefronts' survivor rings */ G < ((int64_t)1 << (31 - (A.log2m + 5))) && D.pk_bits[0] == A.log2m + 5 &&
      D.p2_fkind[0] == PG_P2_F_HLL && !knobs().no_oct_prune) {
    D.oct = 2;
    // The aggregation pass of the pruned passes sees HyperLogLog offers only (COUNT stays in pg_oct_p's LDS table) and keeps the registers
    // as BYTES: a bucket is as many groups as fill the LDS with one byte per register — 512 groups x 256 registers, 25 buckets for
    // config 5 instead of 157 with an accumulator slot and dword registers per group.  Re-cut the key accordingly.
    if (!knobs().oct_dword_regs) {
      const int64_t budget = kLdsTableBudget - (int64_t)PG_P2_LIST * 4 - 256;
      int shift = 0;
      while (shift < 24 && ((int64_t)2 << shift) * (int64_t)A.stride <= budget && shift + 1 + (A.log2m + 5) <= 31) shift++;
      const int64_t nb = (G + ((int64_t)1 << shift) - 1) >> shift;
      if (shift > D.radix_shift && nb >= 1) {
        D.radix_shift = shift;
        D.radix_buckets = (int32_t)nb;
        D.pk_shift[0] = shift;   // plane 0: the local key's bits, then the (index, rank) field
        D.p2_byte_regs = 1;
        P.lds_bytes = ((size_t)A.stride << shift) + (size_t)PG_P2_LIST * 4;
      }
    }
  }
}

// Partition pipeline v2 (pg_kernels_part.hip) for a PG_AGG_RADIX plan: bit-packs what the aggregation pass needs from a doc into
// 1..4 dwords and chooses the bucket width.  Per group the aggregation workgroup keeps n_ops int64 accumulators and one DWORD per
// HyperLogLog register in LDS (a single ds_max_u32 per offer); radix_shift is the largest width whose table fits — lowered (more
// buckets, up to PG_P2_MAX_BUCKETS) when that lets a tuple fit ONE dword, which halves the bytes written and read back.
// Returns false when the shape is outside the pipeline (the round-2 radix passes then run).
static bool plan_partition_v2(CompiledPlan& P, PgQueryPlan& D, const std::vector<Column*>& srcs, const std::vector<PgAccOp>& ops, int64_t G) {
  if (knobs().no_p2) return false;
  if ((int)srcs.size() > PG_MAX_RADIX_SRCS) return false;
  int64_t per_group = (int64_t)ops.size() * 8;
  for (int x = 0; x < D.n_aux; x++) {
    if (D.aux[x].kind == PG_AUX_DICT_SET && D.n_aux == 1) { per_group += (int64_t)D.aux[x].stride * 4; continue; }   // a DISTINCTCOUNT's dictId sets: words as they are
    if (D.aux[x].kind != PG_AUX_HLL_DICT && D.aux[x].kind != PG_AUX_HLL_RAW) return false;
    per_group += (int64_t)4 << D.aux[x].log2m;
  }
  if (per_group <= 0) return false;
  const int64_t budget = kLdsTableBudget - (int64_t)PG_P2_LIST * 4 - 256;
  int shift_max = 0;
  while (shift_max < 24 && ((int64_t)2 << shift_max) * per_group <= budget) shift_max++;
  if (((int64_t)1 << shift_max) * per_group > budget) return false;
  struct Field { int kind, bits; int64_t bias; };
  std::vector<Field> fields(srcs.size());
  bool any_wide = false;
  int small_bits = 0;
  for (size_t si = 0; si < srcs.size(); si++) {
    const Column* c = srcs[si];
    int n_aux_here = 0, log2m = 0;
    bool by_op = false;
    for (int x = 0; x < D.n_aux; x++)
      if (D.aux[x].src == (int32_t)si) { n_aux_here++; log2m = D.aux[x].log2m; D.pk_lut[si] = D.aux[x].kind == PG_AUX_HLL_DICT ? D.aux[x].lut : nullptr; }
    for (auto& o : ops) by_op |= o.src == (int32_t)si;
    if (n_aux_here > 1 || (n_aux_here == 1 && by_op)) return false;   // one consumer per source column
    Field f{0, 0, 0};
    D.pk_hll[si] = 0;
    D.pk_affine[si] = 0;
    if (n_aux_here == 1 && D.aux[0].kind == PG_AUX_DICT_SET) {
      if (!(c->col_kind == PG_COL_FIXED_BIT && c->has_dictionary && c->bits <= 24)) return false;
      f = {PG_P2_F_DICTID, c->bits, 0};
    } else if (n_aux_here == 1) {
      if (!(c->val_type == PG_V_I32 || c->val_type == PG_V_I64) && c->has_dictionary) return false;   // dictionary LUTs exist for any type, but keep to what is tested
      if (!c->has_dictionary && c->data_type > PG_TYPE_DOUBLE) return false;
      if (!c->has_dictionary && (c->val_type == PG_V_F32 || c->val_type == PG_V_F64)) return false;     // Float / Double offers hash other bits: HBM-register path
      f = {PG_P2_F_HLL, log2m + 5, 0};
      D.pk_hll[si] = log2m;
      if (c->has_dictionary && c->dict_affine) {
        D.pk_affine[si] = 1;
        D.pk_base[si] = c->dict_base;
        D.pk_step[si] = c->dict_step;
        // INT dictionary with a small positive step: base + step x dictId in 32-bit arithmetic (every value is an int by definition)
        if (c->val_type == PG_V_I32 && c->dict_step > 0 && c->dict_step < (1 << 24) && c->bits <= 24) D.pk_affine[si] = 2;
      }
    } else if (c->col_kind == PG_COL_FIXED_BIT && c->has_dictionary && c->data_type <= PG_TYPE_DOUBLE) {
      f = {PG_P2_F_DICTID, c->bits, 0};
      // an arithmetic INT dictionary: value = base + step x dictId, computed in the aggregation pass (no look-up: the lean consumer, pg_p2_aggregate_*s)
      if (c->val_type == PG_V_I32 && c->dict_affine && c->dict_step > 0 && c->dict_step < (1 << 24) && c->bits <= 24 &&
          c->dict_base >= INT32_MIN && c->dict_base <= INT32_MAX && !knobs().p2_no_pack) {
        D.pk_affine[si] = 3;
        D.pk_base[si] = c->dict_base;
        D.pk_step[si] = c->dict_step;
      }
    } else if (c->col_kind == PG_COL_RAW32 && c->val_type == PG_V_I32) {
      f = {PG_P2_F_RAW32, 32, 0};
      if (c->has_int_range) {
        const uint64_t range = (uint64_t)(c->int_max - c->int_min);
        int b = 1;
        while (b < 32 && (range >> b) != 0) b++;
        f = {PG_P2_F_RAW32, b, c->int_min};
      }
    } else if (c->col_kind == PG_COL_RAW32 && c->val_type == PG_V_F32) {
      f = {PG_P2_F_RAW32, 32, 0};
    } else if (c->col_kind == PG_COL_RAW64 && (c->val_type == PG_V_I64 || c->val_type == PG_V_F64)) {
      f = {PG_P2_F_RAW64, 64, 0};
      any_wide = true;
    } else {
      return false;
    }
    fields[si] = f;
    if (f.bits < 64) small_bits += f.bits;
  }
  const bool need_docid = P.first_doc_op >= 0;
  auto buckets_of = [&](int shift) { return (G + ((int64_t)1 << shift) - 1) >> shift; };
  // one dword: the key's low bits + every field within 31 bits (bit 31 of plane 0 marks padding)
  int shift = -1, planes = 0;
  if (!any_wide && !need_docid)
    for (int sft = shift_max; sft >= 0 && buckets_of(sft) <= PG_P2_MAX_BUCKETS; sft--)
      if (sft + small_bits <= 31) { shift = sft; planes = 1; break; }
  std::vector<int> fplane(srcs.size(), 0), fshift(srcs.size(), 0);
  int docid_plane = -1;
  if (planes == 1) {
    int next_bit = shift;
    for (size_t si = 0; si < srcs.size(); si++) { fplane[si] = 0; fshift[si] = next_bit; next_bit += fields[si].bits; }
  } else {
    // several planes: first-fit of the fields into dwords (plane 0 keeps bit 31 clear), 64-bit values and the docId in planes of
    // their own; among the bucket widths down to shift_max - 3 the one with the fewest planes wins (a narrower key may free the bits
    // that save a plane), the widest among equals
    auto layout = [&](int sft, std::vector<int>& pl_of, std::vector<int>& sh_of, int& dpl) -> int {
      int used[PG_P2_MAX_PLANES + 2] = {0};
      used[0] = sft;
      int n_pl = 1;
      for (size_t si = 0; si < srcs.size(); si++) {
        if (fields[si].bits == 64) continue;
        int pl = 0;
        while (pl < PG_P2_MAX_PLANES && used[pl] + fields[si].bits > (pl == 0 ? 31 : 32)) pl++;
        if (pl >= PG_P2_MAX_PLANES) return 99;
        pl_of[si] = pl; sh_of[si] = used[pl]; used[pl] += fields[si].bits;
        n_pl = std::max(n_pl, pl + 1);
      }
      for (size_t si = 0; si < srcs.size(); si++) {
        if (fields[si].bits != 64) continue;
        pl_of[si] = n_pl; sh_of[si] = 0;
        n_pl += 2;
      }
      dpl = -1;
      if (need_docid) dpl = n_pl++;
      return n_pl;
    };
    int best = 99;
    for (int sft = shift_max; sft >= 0 && sft >= shift_max - 3; sft--) {
      if (sft > 31 || buckets_of(sft) > PG_P2_MAX_BUCKETS) continue;
      std::vector<int> pl_of(srcs.size(), 0), sh_of(srcs.size(), 0);
      int dpl = -1;
      const int n_pl = layout(sft, pl_of, sh_of, dpl);
      if (n_pl > PG_P2_MAX_PLANES || buckets_of(sft) * n_pl > PG_P2_MAX_BUCKETS) continue;
      if (n_pl < best) { best = n_pl; shift = sft; planes = n_pl; fplane = pl_of; fshift = sh_of; docid_plane = dpl; }
    }
    if (best == 99) return false;
  }
  const int64_t nb = buckets_of(shift);
  if (nb * planes > PG_P2_MAX_BUCKETS) return false;   // the scatter workgroup's leftover lines: [planes][buckets][32] dwords of LDS
  D.p2 = 1;
  D.p2_planes = planes;
  D.p2_docid_plane = docid_plane;
  // the scatter's batched loader (pg_kernels_part.hip "fast A"): <= 4 bit-packed group columns, the first <= 24 bits wide and the
  // others <= 8; at most one source, bit-packed <= 24 bits or raw 32-bit; one or two planes
  {
    bool fast = planes <= 2 && D.n_group_cols >= 1 && D.n_group_cols <= 4 && srcs.size() <= 1 && !knobs().p2_no_fast_a;
    for (int g = 0; g < D.n_group_cols && fast; g++)
      fast = D.gcols[g].col_kind == PG_COL_FIXED_BIT && D.gcols[g].bits <= (g == 0 ? 24 : 8) && (g > 0 || D.gcols[g].mult == 1) &&
             D.gcols[g].mult < (1 << 24);
    if (fast && srcs.size() == 1) {
      const Column* c = srcs[0];
      fast = (c->col_kind == PG_COL_FIXED_BIT && c->bits <= 24) || c->col_kind == PG_COL_RAW32;
    }
    D.p2_fast_a = fast ? 1 : 0;
    // the oct-layout phase A (pg_p2_scatter_o*): a subset of the above — one plane, the source not a HyperLogLog offer
    bool oct = fast && planes == 1 && !knobs().no_p2_oct;
    if (oct && srcs.size() == 1) {
      const Column* c = srcs[0];
      oct = (fields[0].kind == PG_P2_F_RAW32 && c->col_kind == PG_COL_RAW32 && c->val_type == PG_V_I32) ||
            (fields[0].kind == PG_P2_F_DICTID && c->col_kind == PG_COL_FIXED_BIT && c->bits <= 24);
    }
    D.p2_oct_a = oct ? 1 : 0;
  }
  D.radix_shift = shift;
  D.radix_buckets = (int32_t)nb;
  D.radix_packed = 0;
  for (size_t si = 0; si < srcs.size(); si++) {
    D.p2_fplane[si] = fplane[si];
    D.p2_fkind[si] = fields[si].kind;
    D.p2_fbias[si] = fields[si].bias;
    D.pk_shift[si] = fshift[si];
    D.pk_bits[si] = fields[si].bits == 64 ? 32 : fields[si].bits;
  }
  return true;
}

static std::shared_ptr<CompiledPlan> compile_in_space(Segment& seg, OpPtr root_owned, const pg_query* q, const StarTree* st, int star_index) {
  auto plan = std::make_shared<CompiledPlan>();
  CompiledPlan& P = *plan;
  P.root_op = std::move(root_owned);
  FilterOp* const root = P.root_op.get();
  P.star_tree_index = star_index;
  P.space_docs = seg.total_docs;
  Emitter em{seg, P};
  if (root->kind == OpKind::Empty) P.always_empty = true;
  P.match_all = root->kind == OpKind::MatchAll;
  em.emit(*root, true);
  if (em.sp != 1) fail(PG_ERR_INTERNAL, "filter program leaves %d entries on the stack", em.sp);
  if (em.max_sp > PG_MAX_STACK) fail(PG_ERR_UNSUPPORTED, "filter needs %d bitmap stack levels (max %d)", em.max_sp, PG_MAX_STACK);

  PgQueryPlan& D = P.dev;
  D.num_docs = seg.total_docs;
  D.n_tiles = seg.n_tiles;
  D.n_wtiles = (int32_t)(((int64_t)seg.total_docs + PG_WAVE_DOCS - 1) / PG_WAVE_DOCS);
  D.n_instr = (int32_t)em.instrs.size();
  D.stack_depth = em.max_sp;
  D.instrs = em.keep(em.instrs);
  D.scans = em.keep(em.scans);
  D.postings = em.keep(em.postings);
  D.ranges = em.keep(em.ranges);
  D.rangeidx = em.keep(em.rangeidx);
  D.agg_mode = PG_AGG_NONE;
  D.n_groups = 1;
  D.replicas = 1;
  P.algorithmic_bytes = em.alg_bytes;
  // ---- fast-path shape of the filter: [index-only program] (AND one scan of a specialised kind) ---------------------------
  {
    auto index_op = [](int32_t op) {
      return op == PG_F_PUSH_POSTINGS || op == PG_F_PUSH_RANGES || op == PG_F_PUSH_WORDS || op == PG_F_PUSH_RANGEIDX || op == PG_F_PUSH_ALL || op == PG_F_PUSH_NONE ||   // (WORDS / RANGEIDX: see below)
             op == PG_F_AND || op == PG_F_OR || op == PG_F_NOT;
    };
    size_t n_idx = 0;
    while (n_idx < em.instrs.size() && index_op(em.instrs[n_idx].op)) n_idx++;
    {   // the interpreter kernels evaluate the longest index-only prefix that leaves exactly ONE entry on the stack in linear
        // layout (one dword per 32 docs, one transpose for the whole prefix instead of one per leaf)
      int depth = 0;
      size_t best = 0;
      for (size_t i = 0; i < n_idx; i++) {
        const int op = em.instrs[i].op;
        if (op == PG_F_AND || op == PG_F_OR) depth--;
        else if (op != PG_F_NOT) depth++;
        if (depth == 1) best = i + 1;
      }
      D.n_lin_prefix = (int32_t)best;
    }
    // [index program] AND_SCAN ... AND_SCAN, PUSH_POSTINGS, AND: the nested AND of FilterPlanNode.run followed by the
    // queryableDocIds bitmap — the chain kernels AND that posting leaf in after the scans (the scans keep their exact counts)
    size_t n_total = em.instrs.size();
    D.tail_posting = -1;
    if (n_idx > 0 && n_total >= n_idx + 3 && em.instrs[n_total - 1].op == PG_F_AND && em.instrs[n_total - 2].op == PG_F_PUSH_POSTINGS) {
      bool scans_only = true;
      for (size_t i = n_idx; i + 2 < n_total; i++) scans_only &= em.instrs[i].op == PG_F_AND_SCAN;
      if (scans_only) {
        D.tail_posting = em.instrs[n_total - 2].arg;
        n_total -= 2;
      }
    }
    size_t rest = n_total - n_idx;
    bool has_words = false;   // match-word leaves (star-tree traversals with many ranges) are the interpreter kernels' business
    for (auto& in : em.instrs) has_words |= in.op == PG_F_PUSH_WORDS || in.op == PG_F_PUSH_RANGEIDX;   // range-index leaves too
    if (has_words) rest = 1000;
    P.fast_filter = -2;   // -2: interpreter; -1: no scan; >= 0: ScanKind of the single scan
    D.fast_scan = -1;
    auto scan_kind = [&](const PgScanLeaf& L) -> int {   // kinds with a specialised kernel (pg_kernels.hip ScanKind values)
      if (L.col_kind == PG_COL_FIXED_BIT && L.bits <= 8) return L.pred_kind == PG_P_RANGE ? 0 : (L.pred_kind == PG_P_DICT_LUT ? 2 : -2);
      if (L.col_kind == PG_COL_RAW32 && L.val_type == PG_V_I32 && L.pred_kind == PG_P_RANGE) return 4;
      return -2;
    };
    if (rest == 0) {
      P.fast_filter = -1;
      D.n_index_instr = (int32_t)n_idx;
      if (n_idx == 1 && em.instrs[0].op == PG_F_PUSH_ALL) D.n_index_instr = 0;   // match-all: the valid mask itself
    } else if (rest == 1 && D.tail_posting < 0 && ((n_idx == 0 && em.instrs[0].op == PG_F_PUSH_SCAN) || (n_idx > 0 && em.instrs[n_idx].op == PG_F_AND_SCAN))) {
      const int k = scan_kind(em.scans[em.instrs[n_idx].arg]);
      if (k >= 0) {
        P.fast_filter = k;
        D.n_index_instr = (int32_t)n_idx;
        D.fast_scan = em.instrs[n_idx].arg;
        P.fast_scan_bits = em.scans[(size_t)D.fast_scan].bits;
        D.fast_scan_pushed = n_idx == 0 ? 1 : 0;
        D.n_fast_scans = 1;   // the chain kernels (pg_fast_multi_*) can run the single scan as well
      }
    }
    if (P.fast_filter == -2 && rest >= 1 && rest <= PG_MAX_FAST_SCANS) {
      // [index-only program] AND scan AND scan ...: a chain of up to 4 scan leaves of the common kinds (dictionary range /
      // LUT of any width, raw INT / LONG / FLOAT / DOUBLE range), each restricted to the survivors of the previous one (AndDocIdSet applies
      // the scan iterators in list order) — pg_fast_multi_* dispatches the kind per leaf at run time
      auto multi_kind = [&](const PgScanLeaf& L) {
        if (L.col_kind == PG_COL_FIXED_BIT) return L.pred_kind == PG_P_RANGE || L.pred_kind == PG_P_DICT_LUT;
        return L.pred_kind == PG_P_RANGE;
      };
      bool ok = true;
      for (size_t i = n_idx; i < n_total && ok; i++) {
        const int op = em.instrs[i].op;
        ok = (op == PG_F_AND_SCAN || (i == 0 && op == PG_F_PUSH_SCAN)) && multi_kind(em.scans[em.instrs[i].arg]);
      }
      if (ok && n_idx > 0) {   // the prefix must leave exactly one entry (it does when the program is index ops then AND_SCANs)
        int depth = 0;
        for (size_t i = 0; i < n_idx; i++) depth += (em.instrs[i].op == PG_F_AND || em.instrs[i].op == PG_F_OR) ? -1 : (em.instrs[i].op == PG_F_NOT ? 0 : 1);
        ok = depth == 1;
      }
      if (ok) {
        P.fast_filter = 100;
        D.n_index_instr = (int32_t)n_idx;
        D.n_fast_scans = (int32_t)rest;
        D.fast_scan_pushed = n_idx == 0 ? 1 : 0;
        if (n_idx == 1 && em.instrs[0].op == PG_F_PUSH_ALL) { /* cannot happen: and_operator drops match-all children */ }
      } else {
        D.tail_posting = -1;
      }
    }
  }
  if (D.mv) P.fast_filter = -2;   // a multi-value scan leaf: the interpreter's frame only (pg_mv_query_*)
  if (P.fast_filter != 100) D.tail_posting = -1;
  D.dict_filter_only = 0;
  if (!D.mv && D.tail_posting < 0 && (P.fast_filter == 0 || (P.fast_filter == 100 && D.n_fast_scans == 1)) && (size_t)D.n_index_instr < em.instrs.size()) {
    const PgScanLeaf& SL = em.scans[(size_t)em.instrs[(size_t)D.n_index_instr].arg];
    if (SL.col_kind == PG_COL_FIXED_BIT && SL.pred_kind == PG_P_RANGE && SL.bits >= 1 && SL.bits <= 24 && !SL.mv) D.dict_filter_only = 1;
  }
  // Fused dense index program (pg_fast_i32range_d): the index-only prefix is PUSH_POSTINGS (AND PUSH_POSTINGS)* over leaves that are
  // entirely dense (no CSR containers, their dense prefix covers every chunk) with at most 8 pointers and 4 leaves in all
  if ((P.fast_filter == 4 || P.fast_filter == -1 || P.fast_filter == 100) && D.n_index_instr > 0 && !D.fast_scan_pushed) {
    std::vector<int> leaves;
    bool ok = true;
    int depth = 0;
    for (int i = 0; i < D.n_index_instr && ok; i++) {
      const PgFInstr& in = em.instrs[(size_t)i];
      if (in.op == PG_F_PUSH_POSTINGS) { leaves.push_back(in.arg); depth++; }
      else if (in.op == PG_F_AND) depth--;
      else ok = false;
    }
    const int32_t n_chunks = (int32_t)(((int64_t)seg.total_docs + PG_CHUNK_DOCS - 1) / PG_CHUNK_DOCS);
    int n_ptr = 0;
    for (int l : leaves) {
      const PgPostingLeaf& PL = em.postings[(size_t)l];
      ok = ok && !PL.has_csr && PL.n_dense > 0 && PL.dense_chunks >= n_chunks;
      n_ptr += PL.n_dense;
    }
    if (ok && depth == 1 && !leaves.empty() && leaves.size() <= 4 && n_ptr <= 8) {
      int j = 0;
      for (size_t g = 0; g < leaves.size(); g++) {
        const PgPostingLeaf& PL = em.postings[(size_t)leaves[g]];
        for (int k = 0; k < PL.n_dense; k++, j++) { D.dense_ptr[j] = PL.dense[k]; D.dense_group[j] = (int32_t)g; }
        if (PL.exclusive) D.dense_excl |= 1 << g;
      }
      for (; j < 8; j++) { D.dense_ptr[j] = D.dense_ptr[0]; D.dense_group[j] = D.dense_group[0]; }   // OR is idempotent
      D.dense_groups = (int32_t)leaves.size();
      D.dense_fused = 1;
    }
  }
  P.lds_bytes = 0;   // the filter stack lives in registers
  if (!P.stats_exact) {
    // numEntriesScannedInFilter of this shape depends on how the reference's iterators drive each other: every Scan / Inverted
    // leaf gets a filter-only plan of its own whose match bitmap feeds the iterator automaton (pg_filter_stats.cpp) at execution
    std::vector<const FilterOp*> leaves;
    collect_stat_leaves(*root, leaves);
    for (const FilterOp* leaf : leaves) {
      if (leaf->kind == OpKind::Inverted && (leaf->eval.exclusive ? leaf->eval.non_matching : leaf->eval.matching).empty()) continue;
      P.stat_leaves.push_back({leaf, compile_in_space(seg, clone_leaf(*leaf), nullptr, nullptr, -1)});
    }
  }
  if (!q || q->n_aggregations <= 0) return plan;

  // ---- NonScanBasedAggregationOperator (AggregationPlanNode.java:110-120,165-190): no GROUP BY, match-all filter, every
  // aggregation answerable from a dictionary (or COUNT).  A lone COUNT(*) is FastFilteredCountOperator's instead. -------------
  if (!st && q->n_group_by == 0 && root->kind == OpKind::MatchAll &&
      !(q->n_aggregations == 1 && q->aggregations[0].function == PG_AGG_COUNT)) {
    bool fit = true;
    for (int i = 0; i < q->n_aggregations && fit; i++) {
      const pg_agg_spec& s = q->aggregations[i];
      if (s.function == PG_AGG_COUNT) continue;
      Column* c = seg.find(s.column);
      const int32_t f = sv_function_of(s.function);
      const bool dict_fn = s.function != PG_AGG_COUNTMV && (f == PG_AGG_MIN || f == PG_AGG_MAX || f == PG_AGG_MINMAXRANGE ||
                                                            f == PG_AGG_DISTINCTCOUNT || f == PG_AGG_DISTINCTCOUNTHLL);
      fit = c && dict_fn && c->has_dictionary && is_mv_function(s.function) == c->is_mv &&
            (c->data_type <= PG_TYPE_DOUBLE || f == PG_AGG_DISTINCTCOUNT || f == PG_AGG_DISTINCTCOUNTHLL);
    }
    if (fit) {
      P.non_scan_based = true;
      for (int i = 0; i < q->n_aggregations; i++) {
        const pg_agg_spec& s = q->aggregations[i];
        AggOut out{};
        out.function = sv_function_of(s.function);
        out.log2m = s.log2m > 0 ? s.log2m : 8;
        out.aux_col = s.function == PG_AGG_COUNT ? nullptr : seg.find(s.column);
        if (out.function == PG_AGG_DISTINCTCOUNTHLL && (out.log2m < 4 || out.log2m > 16))
          fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNTHLL log2m %d (4..16 on the GPU path)", out.log2m);
        P.aggs.push_back(out);
      }
      return plan;
    }
  }

  // ---- aggregation plan ------------------------------------------------------------------------------------------
  P.num_groups_limit = q->num_groups_limit > 0 ? q->num_groups_limit : 100000;
  if (q->n_group_by > 0 && q->n_order_by > 0 && q->order_by && q->min_segment_group_trim_size > 0) {
    // GroupByOperator.java:120-133 / GroupByUtils.getTableCapacity :45-57
    // (under enableNullHandling only queries over columns without nulls arrive with their ORDER BY: pg_nullaware.cpp trims a joined result itself)
    for (int32_t i = 0; i < q->n_order_by; i++) {
      const pg_order_by& ob = q->order_by[i];
      if (ob.kind == PG_ORDER_BY_GROUP_KEY) {
        if (ob.index < 0 || ob.index >= q->n_group_by) fail(PG_ERR_INVALID_ARGUMENT, "ORDER BY group-by expression %d of %d", ob.index, q->n_group_by);
      } else if (ob.kind == PG_ORDER_BY_AGGREGATION) {
        if (ob.index < 0 || ob.index >= q->n_aggregations) fail(PG_ERR_INVALID_ARGUMENT, "ORDER BY aggregation %d of %d", ob.index, q->n_aggregations);
        // the functions' final results order the groups (TableResizer.java:406-445): the multi-value forms share the single-value functions'
        // intermediates, the distinct counts are ordered by set size / HyperLogLog#cardinality at assembly
        const int f = sv_function_of(q->aggregations[ob.index].function);
        if (!(f == PG_AGG_COUNT || f == PG_AGG_SUM || f == PG_AGG_MIN || f == PG_AGG_MAX || f == PG_AGG_AVG || f == PG_AGG_MINMAXRANGE || f == PG_AGG_DISTINCTCOUNT ||
              f == PG_AGG_DISTINCTCOUNTHLL))
          fail(PG_ERR_UNSUPPORTED, "segment-level group trim ordered by aggregation function %d is left to the Java plan", q->aggregations[ob.index].function);
      } else {
        fail(PG_ERR_INVALID_ARGUMENT, "ORDER BY expression kind %d", ob.kind);
      }
      P.order_by.push_back(ob);
    }
    const int64_t by_limit = (int64_t)std::max(q->limit, 0) * 5;
    P.trim_size = by_limit > INT32_MAX ? INT32_MAX : std::max((int32_t)by_limit, q->min_segment_group_trim_size);
  }
  if (q->n_group_by > PG_MAX_GROUP_COLS) fail(PG_ERR_UNSUPPORTED, "more than %d group-by columns", PG_MAX_GROUP_COLS);
  std::vector<Column*> projected;
  auto project = [&](Column* c) {
    if (c->public_col) c = c->public_col;   // the internal twin of a raw multi-value column counts as that column
    if (std::find(projected.begin(), projected.end(), c) == projected.end()) projected.push_back(c);
  };
  int64_t G = 1;
  bool huge_key_space = false;
  for (int j = 0; j < q->n_group_by; j++) {
    if (!q->group_by_columns) fail(PG_ERR_INVALID_ARGUMENT, "group_by_columns is null");
    Column* c = seg.find(q->group_by_columns[j]);
    if (!c) fail(PG_ERR_NOT_FOUND, "column not found: %s", q->group_by_columns[j] ? q->group_by_columns[j] : "(null)");
    // (a LONG column that may hold Long.MAX_VALUE — the hash table's empty marker once biased — goes through its virtual dictionary)
    const bool one_raw_int = !c->has_dictionary && q->n_group_by == 1 && (c->col_kind == PG_COL_RAW32 || c->col_kind == PG_COL_RAW64) &&
                             (c->data_type == PG_TYPE_INT || (c->data_type == PG_TYPE_LONG && c->max_abs_int < (uint64_t)INT64_MAX));
    if (one_raw_int) {
      // NoDictionarySingleColumnGroupKeyGenerator (core/query/aggregation/groupby/NoDictionarySingleColumnGroupKeyGenerator.java:53-90,
      // 241-265): one raw INT / LONG column, value -> group id; here the value is the 64-bit key of the hash group-by
      D.gcols[j].data = c->fwd_dev.as<uint8_t>();
      D.gcols[j].bits = 0;
      D.gcols[j].mult = 1;
      D.gcols[j].col_kind = c->col_kind;
      P.group_cols.push_back(c);
      P.group_cards.push_back(0);
      P.group_vdict.push_back(nullptr);
      P.raw_group = true;
      huge_key_space = true;
      G = (int64_t)1 << 40;   // unknown number of distinct values: everything that compares G with a limit sees "large"
      project(c);
      continue;
    }
    Column* raw = nullptr;
    if (c->raw_mv) {   // NoDictionary…GroupKeyGenerator over a raw multi-value column: grouped through the twin's ids, keys handed back as values
      raw = c;
      c = c->vdict.get();
    }
    if (c->is_mv) {   // DictionaryBasedGroupKeyGenerator#processMultiValue: every entry of the doc is a key digit (Cartesian over such columns)
      if (st) fail(PG_ERR_UNSUPPORTED, "multi-value group-by column %s over a star-tree", c->name.c_str());
      int n_mv = 0;
      for (int k = 0; k < j; k++) n_mv += D.mv_gcol_offsets[k] != nullptr;
      if (n_mv >= 4) fail(PG_ERR_UNSUPPORTED, "more than four multi-value group-by columns");   // PG_MV_MAX_GROUP_COLS (pg_kernels_mv.hip)
      D.mv_gcol_offsets[j] = c->mv_offsets_dev.as<int32_t>();
      D.mv = 1;
    }
    if (!c->has_dictionary) {
      // any other raw key — FLOAT / DOUBLE, or a raw column among several group-by columns (NoDictionaryMultiColumnGroupKeyGenerator's
      // on-the-fly dictionaries): grouped through the column's virtual dictionary (pg_vdict.hip), like a dictionary column from here on
      ensure_virtual_dictionary(seg, *c);
      raw = c;
      c = c->vdict.get();
    }
    P.group_vdict.push_back(raw ? c : nullptr);
    D.gcols[j].data = c->fwd_dev.as<uint8_t>();
    D.gcols[j].bits = c->bits;
    D.gcols[j].mult = G;
    D.gcols[j].col_kind = PG_COL_FIXED_BIT;
    P.group_cols.push_back(c);
    P.group_cards.push_back(c->cardinality);
    if (!raw) P.dict_hashes.push_back(c->dict_hash);
    // beyond any dense table (the reference's LongMapBasedHolder, DictionaryBasedGroupKeyGenerator.java:166-176): 64-bit raw
    // keys, hash-partitioned and aggregated in LDS hash tables (PG_AGG_RADIX_HASH); keys must stay below 2^62
    if (G > kMaxDenseGroups / std::max(c->cardinality, 1)) huge_key_space = true;
    if (G > ((int64_t)1 << 62) / std::max(c->cardinality, 1)) fail(PG_ERR_UNSUPPORTED, "group key space beyond 2^62 (ArrayMapBasedHolder) is outside the GPU path");
    G *= c->cardinality;
    project(raw ? raw : c);
  }
  D.n_group_cols = q->n_group_by;
  D.n_groups = huge_key_space ? 0 : (int32_t)G;

  std::vector<PgAccOp> ops;
  std::vector<Column*> srcs;
  std::vector<int> src_is_len;   // per source: 1 = the NUMBER of entries of a multi-value column (COUNTMV, AVGMV's count), not its values
  auto src_index_kind = [&](Column* c, int len) {
    for (size_t i = 0; i < srcs.size(); i++) if (srcs[i] == c && src_is_len[i] == len) return (int32_t)i;
    if (srcs.size() >= PG_MAX_SRCS) fail(PG_ERR_UNSUPPORTED, "more than %d aggregated columns", PG_MAX_SRCS);
    srcs.push_back(c);
    src_is_len.push_back(len);
    return (int32_t)srcs.size() - 1;
  };
  auto src_index = [&](Column* c) { return src_index_kind(c, 0); };
  // a multi-value function or column anywhere in the query: pg_mv_query_* (one doc at a time) run the plan
  for (int i = 0; i < q->n_aggregations; i++)
    if (is_mv_function(q->aggregations[i].function)) D.mv = 1;
  auto op_index_kind = [&](int32_t fn, int32_t src, int32_t kind, int32_t limb) {
    for (size_t i = 0; i < ops.size(); i++)
      if (ops[i].fn == fn && ops[i].src == src && ops[i].is_float == kind && ops[i].limb == limb) return (int32_t)i;
    if (ops.size() >= PG_MAX_OPS) fail(PG_ERR_UNSUPPORTED, "more than %d accumulators", PG_MAX_OPS);
    ops.push_back({fn, src, kind, limb});
    return (int32_t)ops.size() - 1;
  };
  auto op_index = [&](int32_t fn, int32_t src, bool is_float) { return op_index_kind(fn, src, is_float ? PG_ACCV_DOUBLE : PG_ACCV_INT, 0); };
  // SUM accumulators of a column (see PgAccValueKind): INT, and LONG whose sum cannot leave int64, add into one int64; other LONG
  // columns into two 32-bit digits; FLOAT / DOUBLE into 3 / 4 fixed-point digits below the column's largest magnitude (96 / 128 bits:
  // exact for every value within 2^-56 / 2^-59 of that magnitude, truncated below — far inside 1 ulp of the exact sum); a column
  // holding NaN / Inf keeps the reference's IEEE double addition.
  std::vector<int32_t> src_fx_q(PG_MAX_SRCS, 0);
  auto sum_ops = [&](Column* c, int32_t si, AggOut& out) {
    if (c->val_type == PG_V_I32) { out.op_a = op_index(PG_ACC_SUM, si, false); P.sum_max_abs = std::max<uint64_t>(P.sum_max_abs, (uint64_t)1 << 31); return; }
    if (c->val_type == PG_V_I64) {
      // one int64 holds the sum when docs x largest magnitude stays below 2^63
      if ((unsigned __int128)c->max_abs_int * (unsigned __int128)std::max(seg.total_docs, 1) < ((unsigned __int128)1 << 63)) {
        out.op_a = op_index(PG_ACC_SUM, si, false);
        P.sum_max_abs = std::max<uint64_t>(P.sum_max_abs, std::max<uint64_t>(c->max_abs_int, 1));
        return;
      }
      out.op_a = op_index_kind(PG_ACC_SUM, si, PG_ACCV_LONG_DIGIT, 0);
      (void)op_index_kind(PG_ACC_SUM, si, PG_ACCV_LONG_DIGIT, 1);
      out.sum_limbs = 2;
      out.fx_q = 0;
      P.has_digit_sums = true;
      return;
    }
    if (c->has_nonfinite) { out.op_a = op_index(PG_ACC_SUM, si, true); return; }
    P.has_digit_sums = true;
    const int limbs = c->val_type == PG_V_F32 ? 3 : 4;
    const int q = c->fx_exp - 32 * limbs + 1;     // |x| * 2^-q < 2^(32 limbs - 1): the top digit stays below 2^31
    src_fx_q[(size_t)si] = q;
    out.op_a = op_index_kind(PG_ACC_SUM, si, PG_ACCV_FIXED_DIGIT, 0);
    for (int j = 1; j < limbs; j++) (void)op_index_kind(PG_ACC_SUM, si, PG_ACCV_FIXED_DIGIT, j);
    out.sum_limbs = limbs;
    out.fx_q = q;
  };
  // Which groups exist?  ArrayBasedHolder keeps a flag per raw key; here a group exists iff its COUNT is > 0 or — when the
  // query has no COUNT/AVG but has a MIN/MAX over an INT source — iff that accumulator left its identity (saves one LDS
  // atomic per matching doc).
  bool need_count = false;
  for (int i = 0; i < q->n_aggregations; i++)
    need_count |= (q->aggregations[i].function == PG_AGG_COUNT || q->aggregations[i].function == PG_AGG_AVG);
  if (D.mv) need_count = true;   // a group exists iff its hidden COUNT is > 0: the one existence test the multi-value kernels are checked with
  bool has_int_minmax = false;
  for (int i = 0; i < q->n_aggregations && !need_count; i++) {
    const pg_agg_spec& s = q->aggregations[i];
    if (s.function == PG_AGG_MIN || s.function == PG_AGG_MAX || s.function == PG_AGG_MINMAXRANGE) {
      Column* c = st ? nullptr : seg.find(s.column);
      if (c && c->val_type == PG_V_I32 && c->data_type == PG_TYPE_INT) has_int_minmax = true;
    }
  }
  if (!has_int_minmax) need_count = true;
  // Without GROUP BY the count is the number of matching docs (stats slot 0): no accumulator (kCountFromStats).
  const int32_t count_op = q->n_group_by == 0 ? kCountFromStats : (need_count ? op_index(PG_ACC_COUNT, -1, false) : -1);
  for (int i = 0; i < q->n_aggregations; i++) {
    const pg_agg_spec& s = q->aggregations[i];
    AggOut out{};
    out.function = s.function;
    if (st) {
      // StarTreeAggregationExecutor / StarTreeGroupByExecutor: the aggregation reads its function-column pair column
      Column* pc = st->pairs[(size_t)st->pair_index(s.function, s.column)].col;
      project(pc);
      if (s.function == PG_AGG_COUNT) {   // CountAggregationFunction.java:99-106,134-141: sum of the pre-aggregated long counts
        sum_ops(pc, src_index(pc), out);   // LONG counts
        out.star_count = true;
        P.aggs.push_back(out);
        continue;
      }
      if (s.function == PG_AGG_AVG || s.function == PG_AGG_MINMAXRANGE) {
        // serialized AvgPair / MinMaxRangePair (AvgAggregationFunction.java:79-93,117-126): sum of the sums + sum of the counts,
        // min of the mins + max of the maxes — over the pair's two halves (pg_startree.cpp upload_pair16); ONE projected column
        const StarTreePair& sp = st->pairs[(size_t)st->pair_index(s.function, s.column)];
        if (!sp.col_b) fail(PG_ERR_UNSUPPORTED, "star-tree pair of function %d is not a BYTES pair", s.function);
        const int32_t sa = src_index(sp.col), sb = src_index(sp.col_b);
        out.is_float = true;
        if (s.function == PG_AGG_AVG) {
          sum_ops(sp.col, sa, out);
          AggOut cnt{};
          sum_ops(sp.col_b, sb, cnt);
          if (cnt.sum_limbs != 0) fail(PG_ERR_UNSUPPORTED, "star-tree AvgPair counts beyond int64 sums");
          out.op_b = cnt.op_a;
        } else {
          out.op_a = op_index(PG_ACC_MIN, sa, true);
          out.op_b = op_index(PG_ACC_MAX, sb, true);
        }
        P.aggs.push_back(out);
        continue;
      }
      if (s.function == PG_AGG_DISTINCTCOUNTHLL) {   // serialized HyperLogLogs: register-wise max (addAll)
        if (D.n_aux >= PG_MAX_AUX) fail(PG_ERR_UNSUPPORTED, "more than %d DISTINCTCOUNT / DISTINCTCOUNTHLL aggregations", PG_MAX_AUX);
        PgAuxOp& A = D.aux[D.n_aux];
        A.kind = PG_AUX_HLL_BYTES;
        A.src = src_index(pc);
        A.log2m = pc->hll_log2m;
        A.stride = 1 << pc->hll_log2m;
        out.aux = D.n_aux++;
        out.log2m = pc->hll_log2m;
        out.aux_col = pc;
        P.aggs.push_back(out);
        continue;
      }
    }
    if (s.function == PG_AGG_COUNT) { out.op_a = count_op; P.aggs.push_back(out); continue; }
    Column* c = st ? st->pairs[(size_t)st->pair_index(s.function, s.column)].col : seg.find(s.column);
    if (!c) fail(PG_ERR_NOT_FOUND, "column not found: %s", s.column ? s.column : "(null)");
    if (c->raw_mv) {
      // the *MV functions read the values through the twin's dictionary; DISTINCTCOUNTMV would hand back ids of a dictionary the caller
      // does not have (BaseDistinctAggregateAggregationFunction keeps value sets for raw columns): the Java plan answers it
      if (sv_function_of(s.function) == PG_AGG_DISTINCTCOUNT) fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNTMV over the raw multi-value column %s", c->name.c_str());
      c = c->vdict.get();
    }
    // BlockValSet#getDoubleValuesSV over a multi-value column (and ...MV over a single-value one) throws in the reference
    if (is_mv_function(s.function) != c->is_mv)
      fail(PG_ERR_INVALID_ARGUMENT, "aggregation function %d over the %s column %s", s.function, c->is_mv ? "multi-value" : "single-value", c->name.c_str());
    const int32_t fn_sv = sv_function_of(s.function);
    out.function = fn_sv;   // intermediate results, merges and the wire format are the single-value function's
    if (c->is_mv && s.function == PG_AGG_COUNTMV) {   // CountMVAggregationFunction.java:62-96: sum of getNumMVEntries
      project(c);
      out.op_a = op_index(PG_ACC_SUM, src_index_kind(c, 1), false);
      P.sum_max_abs = std::max<uint64_t>(P.sum_max_abs, (uint64_t)std::max(c->max_entries_per_doc, 1));
      P.aggs.push_back(out);
      continue;
    }
    if (fn_sv == PG_AGG_DISTINCTCOUNT || fn_sv == PG_AGG_DISTINCTCOUNTHLL) {
      if (D.n_aux >= PG_MAX_AUX) fail(PG_ERR_UNSUPPORTED, "more than %d DISTINCTCOUNT / DISTINCTCOUNTHLL aggregations", PG_MAX_AUX);
      const bool hll = fn_sv == PG_AGG_DISTINCTCOUNTHLL;
      if (!hll && !c->has_dictionary) {
        // BaseDistinctAggregateAggregationFunction.java:325-380 keeps typed VALUE sets for a raw column: the column's virtual dictionary
        // (pg_vdict.hip, built once per column) turns it into a dictId set, handed back as values (PG_RESULT_VALUE_SET)
        if (st || c->is_mv || !((c->col_kind == PG_COL_RAW32 || c->col_kind == PG_COL_RAW64) && c->data_type <= PG_TYPE_DOUBLE))
          fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNT over the raw column %s: only single-value INT / LONG / FLOAT / DOUBLE columns", c->name.c_str());
        if (q->flags & PG_QUERY_FLAG_NULL_HANDLING) fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNT over the raw column %s under enableNullHandling", c->name.c_str());
        ensure_virtual_dictionary(seg, *c);
        project(c);   // (the statistics count the column the segment knows by name)
        c = c->vdict.get();
      }
      if (hll && !c->has_dictionary && c->data_type > PG_TYPE_DOUBLE) fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNTHLL over a raw %d column", c->data_type);
      const int log2m = s.log2m > 0 ? s.log2m : 8;   // CommonConstants.Helix.DEFAULT_HYPERLOGLOG_LOG2M
      if (hll && (log2m < 4 || log2m > 16)) fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNTHLL log2m %d (4..16 on the GPU path)", log2m);
      if (c->vdict_kind < 0) project(c);
      PgAuxOp& A = D.aux[D.n_aux];
      A.src = src_index(c);
      A.log2m = hll ? log2m : 0;
      if (!hll) { A.kind = PG_AUX_DICT_SET; A.stride = (c->cardinality + 31) / 32; P.dict_hashes.push_back(c->vdict_kind >= 0 ? c->vdict_hash : c->dict_hash); }   // sets are indexed by dictId
      else if (c->has_dictionary) { A.kind = PG_AUX_HLL_DICT; A.stride = 1 << log2m; A.lut = hll_dict_lut(*c, log2m); }
      else { A.kind = PG_AUX_HLL_RAW; A.stride = 1 << log2m; }
      out.aux = D.n_aux++;
      out.log2m = log2m;
      out.aux_col = c;
      P.aggs.push_back(out);
      continue;
    }
    if (c->data_type > PG_TYPE_DOUBLE) fail(PG_ERR_INVALID_ARGUMENT, "Cannot compute aggregation for non-numeric type: column %s", c->name.c_str());
    project(c);
    const int32_t si = src_index(c);
    const bool fl = c->val_type == PG_V_F32 || c->val_type == PG_V_F64;
    out.is_float = fl;
    if (c->is_mv && (fn_sv == PG_AGG_SUM || fn_sv == PG_AGG_AVG)) {
      // SumMV / AvgMV: the doc's entries are summed in int64 on the device, then added once per key.  Floating entries would need the
      // digit accumulators per ENTRY: kept with the Java plan for now
      if (fl) {   // FLOAT / DOUBLE entries: the fixed-point digit accumulators of sum_ops, every entry of the doc cut into its digits (pg_kernels_mv.hip)
        if (c->has_nonfinite) fail(PG_ERR_UNSUPPORTED, "SUMMV / AVGMV over %s, which holds NaN / Inf", c->name.c_str());
        sum_ops(c, si, out);
        if (fn_sv == PG_AGG_AVG) out.op_b = op_index(PG_ACC_SUM, src_index_kind(c, 1), false);   // AvgMV: count += values.length
        P.aggs.push_back(out);
        continue;
      }
      const unsigned __int128 worst = (unsigned __int128)std::max<uint64_t>(c->val_type == PG_V_I32 ? (uint64_t)1 << 31 : c->max_abs_int, 1) *
                                      (unsigned __int128)std::max(c->total_entries, 1);
      if (worst >= ((unsigned __int128)1 << 63)) fail(PG_ERR_UNSUPPORTED, "SUMMV over %s may leave int64", c->name.c_str());
      out.op_a = op_index(PG_ACC_SUM, si, false);
      // what one doc can add to a group (the merge's overflow bound multiplies it by the docs)
      P.sum_max_abs = std::max<uint64_t>(P.sum_max_abs, (uint64_t)std::min<unsigned __int128>((unsigned __int128)std::max<uint64_t>(c->val_type == PG_V_I32 ? (uint64_t)1 << 31 : c->max_abs_int, 1) *
                                                                                              (unsigned __int128)std::max(c->max_entries_per_doc, 1), (unsigned __int128)1 << 62));
      if (fn_sv == PG_AGG_AVG) out.op_b = op_index(PG_ACC_SUM, src_index_kind(c, 1), false);   // AvgMV: count += values.length
      P.aggs.push_back(out);
      continue;
    }
    switch (fn_sv) {
      case PG_AGG_SUM: sum_ops(c, si, out); break;
      case PG_AGG_MIN: out.op_a = op_index(PG_ACC_MIN, si, fl); break;
      case PG_AGG_MAX: out.op_a = op_index(PG_ACC_MAX, si, fl); break;
      case PG_AGG_AVG: sum_ops(c, si, out); out.op_b = count_op; break;
      case PG_AGG_MINMAXRANGE: out.op_a = op_index(PG_ACC_MIN, si, fl); out.op_b = op_index(PG_ACC_MAX, si, fl); break;
      default: fail(PG_ERR_UNSUPPORTED, "aggregation function %d", s.function);
    }
    P.aggs.push_back(out);
  }
  // numGroupsLimit (InstancePlanMakerImplV2.java:75-96, DictionaryBasedGroupKeyGenerator.java:166-185,416-446): once
  // `limit` distinct keys have been seen — in docId order — new keys get INVALID_ID and their docs are dropped, while the
  // admitted groups keep aggregating.  Equivalent without an order of execution: keep the `limit` groups whose FIRST matching
  // docId is smallest.  Only when the key space can exceed the limit does the plan carry that MIN(docId) accumulator.
  int32_t first_doc_op_unsorted = -1;
  // (multi-value plans carry no such accumulator — a doc admits several keys, in entry order: a result with more groups than the limit
  // is refused at execution instead, pg_exec.hip)
  if (q->n_group_by > 0 && G > (int64_t)P.num_groups_limit && !D.mv) first_doc_op_unsorted = op_index(PG_ACC_MIN, -1, false);
  // kernels walk ops grouped by source: stable sort by src and remap
  std::vector<int32_t> order(ops.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return ops[a].src < ops[b].src; });
  std::vector<int32_t> remap(ops.size());
  std::vector<PgAccOp> sorted_ops(ops.size());
  for (size_t i = 0; i < order.size(); i++) { sorted_ops[i] = ops[order[i]]; remap[order[i]] = (int32_t)i; }
  for (auto& a : P.aggs) { if (a.op_a >= 0) a.op_a = remap[a.op_a]; if (a.op_b >= 0) a.op_b = remap[a.op_b]; }
  if (first_doc_op_unsorted >= 0) P.first_doc_op = remap[first_doc_op_unsorted];
  P.exist_op = q->n_group_by == 0 ? kCountFromStats : -1;
  for (size_t i = 0; i < sorted_ops.size(); i++) {
    if (P.exist_op == -1 && sorted_ops[i].fn == PG_ACC_COUNT) { P.exist_op = (int32_t)i; break; }
  }
  if (P.exist_op == -1)
    for (size_t i = 0; i < sorted_ops.size(); i++)
      if ((sorted_ops[i].fn == PG_ACC_MIN || sorted_ops[i].fn == PG_ACC_MAX) && !sorted_ops[i].is_float && sorted_ops[i].src >= 0 &&
          srcs[sorted_ops[i].src]->val_type == PG_V_I32) { P.exist_op = (int32_t)i; break; }
  if (P.exist_op == -1) fail(PG_ERR_INTERNAL, "no existence accumulator");
  D.n_ops = (int32_t)sorted_ops.size();
  for (int i = 0; i < D.n_ops; i++) D.ops[i] = sorted_ops[i];
  P.ops_dev = upload_vector(sorted_ops);
  D.n_srcs = (int32_t)srcs.size();
  for (size_t i = 0; i < srcs.size(); i++) {
    Column* c = srcs[i];
    D.srcs[i].data = c->fwd_dev.as<uint8_t>();
    D.srcs[i].dict = c->has_dictionary ? c->dict_dev.ptr : nullptr;
    D.srcs[i].col_kind = c->col_kind;
    D.srcs[i].bits = c->bits;
    D.srcs[i].val_type = c->val_type;
    D.srcs[i].fx_q = src_fx_q[i];
    if (c->is_mv) {
      D.mv_src_offsets[i] = c->mv_offsets_dev.as<int32_t>();
      D.mv_src_len[i] = src_is_len[i];
      D.mv = 1;
    }
  }
  if (D.mv) {
    // the multi-value kernels take dense key spaces (an LDS or HBM table) and admit every key: beyond that, the Java plan
    if (huge_key_space) fail(PG_ERR_UNSUPPORTED, "multi-value query over a group key space beyond 64 M keys");
    if (st) fail(PG_ERR_UNSUPPORTED, "multi-value query over a star-tree");
  }
  for (Column* c : projected) P.algorithmic_bytes += (int64_t)c->fwd_bytes_logical;
  P.n_projected_columns = (int32_t)projected.size();

  if (huge_key_space) {
    if (D.n_aux > 0) fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNT / DISTINCTCOUNTHLL over a group key space beyond 64 M keys");
    if ((int)srcs.size() > PG_MAX_RADIX_SRCS) fail(PG_ERR_UNSUPPORTED, "more than %d aggregated columns over a hashed key space", PG_MAX_RADIX_SRCS);
    D.agg_mode = PG_AGG_RADIX_HASH;
    D.replicas = 1;
    D.replica_shift = 0;
    int cap = 1024;   // slots of the per-bucket LDS hash table: 8 B key + 8 B per accumulator
    while (cap < 16384 && (int64_t)cap * 2 * (8 + 8 * (int64_t)D.n_ops) <= kLdsTableBudget) cap *= 2;
    D.hash_cap = cap;
    P.lds_bytes = (size_t)cap * (8 + 8 * (size_t)D.n_ops);
    P.fast_agg = false;
    return plan;
  }
  const int64_t table_bytes = G * D.n_ops * 8;
  // DISTINCTCOUNTHLL states that do not fit one workgroup's LDS next to the table (config 5 without its star-tree: 12 800 groups
  // x 256 registers = 3.2 MB): updating them in HBM costs a dependent random read per doc (~23 ps per doc measured).  The radix
  // pipeline carries the doc's value in the tuple and keeps the registers of one bucket of groups in LDS instead.
  bool hll_radix = false;
  int hll_shift = 0;
  if (q->n_group_by > 0 && D.n_aux > 0 && D.n_ops > 0 && !D.mv && !knobs().no_radix && !knobs().no_radix_aux) {
    int64_t per_group = (int64_t)D.n_ops * 8, state_bytes = 0;
    bool ok = (int)srcs.size() <= PG_MAX_RADIX_SRCS;
    // ... and DISTINCTCOUNT's dictId sets (round 6): ONE set op over a <= 24-bit dictionary column next to COUNTs only — the tuple carries the
    // dictId, the aggregation pass ORs bits into the bucket's sets in LDS (pg_p2_aggregate_1set); a group's set must fit one workgroup's LDS
    bool set_radix = D.n_aux == 1 && D.aux[0].kind == PG_AUX_DICT_SET && !knobs().no_p2 && P.first_doc_op < 0;
    if (set_radix) {
      const Column* c = srcs[(size_t)D.aux[0].src];
      set_radix = c->has_dictionary && !c->is_mv && c->col_kind == PG_COL_FIXED_BIT && c->bits <= 24 && (int64_t)D.aux[0].stride * 4 + 8 * (int64_t)D.n_ops <= kLdsTableBudget - (int64_t)PG_P2_LIST * 4 - 256;
      for (auto& o : sorted_ops) set_radix = set_radix && o.src < 0 && o.fn == PG_ACC_COUNT;
    }
    for (int x = 0; x < D.n_aux && ok; x++) {
      const PgAuxOp& A = D.aux[x];
      const Column* c = srcs[(size_t)A.src];
      if (set_radix) { per_group += (int64_t)A.stride * 4; state_bytes += G * (int64_t)A.stride * 4; continue; }
      ok = (A.kind == PG_AUX_HLL_DICT || A.kind == PG_AUX_HLL_RAW) && (c->val_type == PG_V_I32 || c->val_type == PG_V_I64) && A.log2m <= 10;
      per_group += A.stride;
      state_bytes += G * (int64_t)A.stride;
    }
    if (ok && table_bytes + state_bytes > kLdsTableBudget) {
      while (((int64_t)2 << hll_shift) * per_group <= kLdsTableBudget) hll_shift++;
      const int64_t buckets = (G + ((int64_t)1 << hll_shift) - 1) >> hll_shift;
      const double sel = estimate_selectivity(*root, (double)seg.total_docs);
      const double cost_radix = 4.0 + sel * ((4.0 + 8.0 * (double)srcs.size()) * 0.25 + (double)D.n_ops * 0.95 + (double)D.n_aux * 1.5);
      const double cost_hbm = 1.25 + sel * (double)D.n_aux * 20.0;
      hll_radix = buckets <= PG_MAX_RADIX_BUCKETS && cost_radix < cost_hbm;
      if (set_radix) hll_radix = hll_radix && buckets <= PG_P2_MAX_BUCKETS;   // (only the partition pipeline v2 aggregates sets)
    }
  }
  if (D.n_ops == 0 && D.n_aux == 0) {
    D.agg_mode = PG_AGG_NONE;        // COUNT(*) only: nothing to accumulate beyond the match count
  } else if (q->n_group_by == 0) {
    D.agg_mode = PG_AGG_SINGLE;
    D.replicas = PG_BLOCK;          // one private slot per thread: no atomic conflicts
  } else if (hll_radix) {
    D.agg_mode = PG_AGG_RADIX;
    D.radix_shift = hll_shift;
    D.radix_buckets = (int32_t)((G + ((int64_t)1 << hll_shift) - 1) >> hll_shift);
    D.replicas = 1;
  } else if (table_bytes <= kLdsTableBudget) {
    D.agg_mode = PG_AGG_LDS;
    int r = 1;   // replicas spread same-group updates of the 16 wavefronts over distinct LDS addresses
    while (r < 64 && table_bytes * (r * 2) <= kLdsReplicaBudget) r *= 2;
    D.replicas = r;
  } else {
    // beyond one LDS table: range-partitioned LDS tables when <= 32 ranges cover the key space (LDS atomics sustain ~2e12/s,
    // memory-side atomics on an HBM table ~2.4e10/s — tools/probes/atomic_scope.hip), else the dense HBM table
    int parts = (int)std::min<int64_t>(32, std::max<int64_t>(2, (table_bytes + kLdsTableBudget - 1) / kLdsTableBudget));
    if (knobs().part_min >= 0) parts = std::max(parts, std::min(32, knobs().part_min));   // measurement knob
    // every range's workgroups visit every doc (~4e11 doc visits/s measured), the dense HBM table pays per matching doc and
    // accumulator (memory-side atomics, 2.4e10/s): partition when  parts * N / 4e11  <  matched * ops / 2.4e10, with the
    // match count estimated from the filter (exact for index leaves)
    const double sel = estimate_selectivity(*root, (double)seg.total_docs);
    // three ways, cost per doc of the segment (seconds x 1e12, measured rates — profiles/r01_v4_variants_200m.txt):
    //   dense HBM table       filter pass + memory-side atomics per matching doc and accumulator (2.4e10/s)
    //   range partitions      every range's workgroups visit every doc (4.2e11 visits/s)
    //   radix partition       filter pass, two passes over the matches' keys, tuples written and read once, LDS atomics
    const int n_srcs_radix = (int)srcs.size();
    int radix_shift = 0;
    while (((int64_t)2 << radix_shift) * std::max(D.n_ops, 1) * 8 <= kLdsTableBudget) radix_shift++;
    const int64_t radix_buckets = (G + ((int64_t)1 << radix_shift) - 1) >> radix_shift;
    const bool radix_ok = D.n_aux == 0 && n_srcs_radix <= PG_MAX_RADIX_SRCS && D.n_ops > 0 &&
                          radix_buckets <= PG_MAX_RADIX_BUCKETS && !knobs().no_radix;
    const bool part_ok = table_bytes <= kLdsTableBudget * parts && !knobs().no_part;
    const double ops_d = (double)std::max(D.n_ops, 1);
    const double cost_global = 1.25 + sel * ops_d * 41.7;
    const double cost_part = part_ok ? parts * 2.4 : 1e30;
    const double cost_radix = radix_ok ? 4.0 + sel * ((4.0 + 8.0 * n_srcs_radix) * 0.25 + ops_d * 0.95) : 1e30;   // 200 M rows, 40k keys: 0.92 / 1.78 ms at 12.5 / 100 %
    if (cost_radix <= cost_part && cost_radix <= cost_global) {
      D.agg_mode = PG_AGG_RADIX;
      D.radix_shift = radix_shift;
      D.radix_buckets = (int32_t)radix_buckets;
    } else if (cost_part <= cost_global) {
      D.agg_mode = PG_AGG_LDS_PART;
      D.n_parts = parts;
      D.part_groups = (int32_t)((G + parts - 1) / parts);
    } else {
      D.agg_mode = PG_AGG_GLOBAL;
    }
    D.replicas = 1;
  }
  if (D.mv && (D.agg_mode == PG_AGG_RADIX || D.agg_mode == PG_AGG_LDS_PART)) {   // pg_mv_query_*: an LDS table or the dense HBM table
    D.agg_mode = PG_AGG_GLOBAL;
    D.replicas = 1;
  }
  if (D.agg_mode == PG_AGG_RADIX) (void)plan_partition_v2(P, D, srcs, sorted_ops, G);
  D.replica_shift = 0;
  while ((1 << D.replica_shift) < D.replicas) D.replica_shift++;
  if (D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE) P.lds_bytes += (size_t)G * D.replicas * D.n_ops * 8;
  if (D.agg_mode == PG_AGG_LDS_PART) P.lds_bytes += (size_t)D.part_groups * D.n_ops * 8;
  if (D.agg_mode == PG_AGG_RADIX) {
    P.lds_bytes += ((size_t)D.n_ops << D.radix_shift) * 8;
    for (int x = 0; x < D.n_aux; x++) P.lds_bytes += ((size_t)D.aux[x].stride << D.radix_shift) * (D.p2 || D.aux[x].kind == PG_AUX_DICT_SET ? 4 : 1);   // p2: one dword per register; sets: words
    if (D.p2) P.lds_bytes += (size_t)PG_P2_LIST * 4;
  }
  // auxiliary regions (HBM): sizes per op, patched into the plan at execution
  for (int x = 0; x < D.n_aux; x++) {
    size_t bytes = D.aux[x].kind == PG_AUX_DICT_SET ? (size_t)G * D.aux[x].stride * 4 : (size_t)G * D.aux[x].stride;
    if (bytes > kMaxAuxBytes) fail(PG_ERR_UNSUPPORTED, "DISTINCTCOUNT state of %zu bytes exceeds the GPU path's limit", bytes);
    bytes = (bytes + 255) & ~(size_t)255;
    int n_rep = 1;   // replicate small states (up to 256 KB in total, at most one replica per workgroup)
    while (n_rep < 512 && bytes * (size_t)(n_rep * 2) <= ((size_t)256 << 10)) n_rep *= 2;
    if (D.agg_mode == PG_AGG_RADIX) n_rep = 1;   // one merged region, written by pg_radix_reduce_aux_kernel
    D.aux[x].n_rep = n_rep;
    D.aux[x].rep_bytes = (int64_t)bytes;
    D.aux[x].lds_offset = -1;
    P.aux_bytes.push_back(bytes * (size_t)n_rep);
  }
  // States that fit the workgroup's LDS next to the accumulator table (a DISTINCTCOUNT over <= ~1 M dictIds without GROUP BY,
  // HyperLogLog registers of <= ~500 groups ...) are kept there: LDS atomics sustain ~2e12/s, memory-side ones 2.4e10/s.
  {
    size_t aux_lds = 0;
    for (int x = 0; x < D.n_aux; x++) aux_lds += (size_t)D.aux[x].rep_bytes;
    const bool lds_table = D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE;
    if (D.n_aux > 0 && lds_table && P.lds_bytes + aux_lds <= (size_t)kLdsTableBudget) {
      size_t off = (P.lds_bytes + 255) & ~(size_t)255;
      for (int x = 0; x < D.n_aux; x++) {
        D.aux[x].lds_offset = (int32_t)off;
        D.aux[x].n_rep = 1;
        off += (size_t)D.aux[x].rep_bytes;
        P.aux_bytes[(size_t)x] = (size_t)D.aux[x].rep_bytes;   // the merged state; per-workgroup partials are sized at launch
      }
      P.lds_bytes = off;
      P.aux_in_lds = true;
    }
  }
  // Packed 4-byte radix tuples (pg_kernels.hip "Packed radix tuples"): the local key plus, per source, the (index, rank) a
  // DISTINCTCOUNTHLL offers or the dictId of a dictionary-encoded source, when that fits 32 bits; staged in LDS by at most 32 buckets
  if (D.agg_mode == PG_AGG_RADIX && !D.p2 && P.first_doc_op < 0 && D.radix_buckets <= 64 && !knobs().no_radix_packed) {
    bool ok = true;
    int next_bit = D.radix_shift;
    for (size_t si = 0; si < srcs.size() && ok; si++) {
      bool by_aux = false, by_op = false;
      int log2m = 0;
      for (int x = 0; x < D.n_aux; x++)
        if (D.aux[x].src == (int32_t)si) {
          ok = ok && !by_aux && (D.aux[x].kind == PG_AUX_HLL_DICT || D.aux[x].kind == PG_AUX_HLL_RAW);   // one HyperLogLog per source column
          by_aux = true;
          log2m = D.aux[x].log2m;
          D.pk_lut[si] = D.aux[x].kind == PG_AUX_HLL_DICT ? D.aux[x].lut : nullptr;
        }
      for (auto& o : sorted_ops) by_op |= o.src == (int32_t)si;
      const Column* c = srcs[si];
      if (by_aux && by_op) ok = false;
      else if (by_aux) {
        D.pk_hll[si] = log2m;
        D.pk_bits[si] = log2m + 5;
        if (c->has_dictionary && c->dict_affine) { D.pk_affine[si] = 1; D.pk_base[si] = c->dict_base; D.pk_step[si] = c->dict_step; }
      }
      else if (c->col_kind == PG_COL_FIXED_BIT && c->has_dictionary && c->data_type <= PG_TYPE_DOUBLE) { D.pk_hll[si] = 0; D.pk_bits[si] = c->bits; }
      else ok = false;
      D.pk_shift[si] = next_bit;
      next_bit += D.pk_bits[si];
    }
    if (ok && next_bit < 32) D.radix_packed = 1;   // bit 31 stays clear: a tuple never equals PG_RADIX_INVALID_KEY (the padding marker)
  }
  plan_oct(P, D, srcs, G, seg.total_docs);
  // fast aggregation: LDS table, slots fit 16 bits, <= 8-bit group columns, 32-bit value sources
  P.fast_agg = D.agg_mode != PG_AGG_GLOBAL && D.agg_mode != PG_AGG_LDS_PART && D.agg_mode != PG_AGG_RADIX && (int64_t)G * D.replicas <= 65536;   // (trivially true without a table)
  if (D.n_aux > 0) P.fast_agg = false;   // set / HLL accumulators run in the interpreter kernel
  for (auto& o : sorted_ops) if (o.is_float >= PG_ACCV_FIXED_DIGIT) { P.fast_agg = false; P.digit_ops = true; }   // digit accumulators: the general aggregator
  if (P.first_doc_op >= 0) P.fast_agg = false;
  for (Column* c : P.group_cols) if (c->bits > 8) P.fast_agg = false;
  for (Column* c : srcs)
    if (!(c->col_kind == PG_COL_RAW32 || (c->col_kind == PG_COL_FIXED_BIT && (c->val_type == PG_V_I32 || c->val_type == PG_V_F32))))
      P.fast_agg = false;
  D.fast_agg_shape = (P.fast_agg && (D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE)) ? 1 : 0;
  // the software-pipelined headline kernel keeps one value column and up to two group columns of a tile in registers
  D.pipe_fit = 0;
  D.pipe_src = -1;
  if (D.dense_fused && P.fast_agg && D.agg_mode == PG_AGG_LDS && D.n_group_cols >= 1 && D.n_group_cols <= 2) {
    bool ok = true;
    int src = -1;
    for (int o = 0; o < D.n_ops && ok; o++) {
      if (D.ops[o].src < 0) continue;
      if (D.ops[o].is_float != PG_ACCV_INT) ok = false;
      if (src >= 0 && D.ops[o].src != src) ok = false;
      src = D.ops[o].src;
    }
    if (ok && src >= 0 && D.srcs[src].col_kind == PG_COL_RAW32) { D.pipe_fit = 1; D.pipe_src = src; }
  }
  if (P.fast_filter != 4) D.pipe_fit = 0;   // pg_fast_i32range_p itself: dense index program AND one raw-INT range scan
  // The pipeline's other shapes (pg_pipe_*): the same aggregation behind no filter at all, a lone range scan, index leaves only, and any of
  // them (or the headline shape) followed by the upsert snapshot's bitmap
  D.pipe_general = 0;
  if (!D.pipe_fit && P.fast_agg && D.agg_mode == PG_AGG_LDS && D.n_group_cols >= 1 && D.n_group_cols <= 2 && !knobs().no_pipe_general) {
    bool ok = true;
    int src = -1;
    for (int o = 0; o < D.n_ops && ok; o++) {
      if (D.ops[o].src < 0) continue;
      if (D.ops[o].is_float != PG_ACCV_INT) ok = false;
      if (src >= 0 && D.ops[o].src != src) ok = false;
      src = D.ops[o].src;
    }
    ok = ok && src >= 0 && D.srcs[src].col_kind == PG_COL_RAW32;
    const int32_t n_chunks = (int32_t)(((int64_t)seg.total_docs + PG_CHUNK_DOCS - 1) / PG_CHUNK_DOCS);
    const bool index_ok = D.n_index_instr == 0 || D.dense_fused;   // no index program, or the fused dense form
    bool has_scan = false, has_tail = false;
    D.pipe_vscan = -1;
    if (P.fast_filter == 100 && D.n_fast_scans == 2 && D.tail_posting < 0 && src >= 0 && ok) {
      // [index] AND range(raw INT a) AND range(value column): the second scan is tested on the value quads
      const int i1 = em.instrs[(size_t)D.n_index_instr].arg, i2 = em.instrs[(size_t)D.n_index_instr + 1].arg;
      const PgScanLeaf& S1 = em.scans[(size_t)i1];
      const PgScanLeaf& S2 = em.scans[(size_t)i2];
      auto raw_i32_range = [](const PgScanLeaf& S) { return S.col_kind == PG_COL_RAW32 && S.val_type == PG_V_I32 && S.pred_kind == PG_P_RANGE; };
      has_scan = true;
      ok = index_ok && raw_i32_range(S1) && raw_i32_range(S2) && S2.data == D.srcs[src].data;
      if (ok) { D.fast_scan = i1; D.pipe_vscan = i2; }
    } else if (P.fast_filter == -1) {
      ok = ok && index_ok;
    } else if (P.fast_filter == 4) {
      has_scan = true;
      ok = ok && D.fast_scan_pushed && D.n_index_instr == 0;   // with a dense index program it is pg_fast_i32range_p; any other index: not here
    } else if (P.fast_filter == 100 && D.n_fast_scans == 1 && D.tail_posting >= 0) {
      const PgScanLeaf& SL = em.scans[(size_t)em.instrs[(size_t)D.n_index_instr].arg];
      const PgPostingLeaf& TL = em.postings[(size_t)D.tail_posting];
      has_scan = true;
      has_tail = true;
      ok = ok && index_ok && SL.col_kind == PG_COL_RAW32 && SL.val_type == PG_V_I32 && SL.pred_kind == PG_P_RANGE &&
           !TL.has_csr && TL.n_dense == 1 && TL.dense_chunks >= n_chunks && !TL.exclusive;
      if (ok) { D.fast_scan = em.instrs[(size_t)D.n_index_instr].arg; D.pipe_tail = TL.dense[0]; }
    } else {
      ok = false;
    }
    if (ok) {
      D.pipe_general = 1;
      D.pipe_src = src;
      D.pipe_has_index = D.n_index_instr > 0 ? 1 : 0;
      D.pipe_has_scan = has_scan ? 1 : 0;
      if (!has_tail) D.pipe_tail = nullptr;
    } else {
      D.pipe_vscan = -1;
    }
  }
  // The candidate rate of the index program is known at plan time — posting cardinalities are exact, an AND over columns multiplies them (the
  // reference orders an AND's children by the same numbers, AndDocIdSet.java:110) — so a plan's FIRST execution already takes the kernel its
  // filter calls for (pg_fast_i32range_s streams everything, _p skips quads without candidates: pg_exec.hip, spec_shape); later executions
  // replace the estimate with what the kernels counted.
  if ((D.pipe_fit || D.pipe_general) && seg.total_docs > 0) {
    std::function<double(const FilterOp&)> index_rate = [&](const FilterOp& op) -> double {
      if (op.kind == OpKind::Scan) return 1.0;
      if (op.kind == OpKind::And) { double s = 1.0; for (auto& ch : op.children) s *= index_rate(*ch); return s; }
      return estimate_selectivity(op, (double)seg.total_docs);
    };
    const double cand = std::min(1.0, std::max(0.0, index_rate(*root)));
    P.observed_candidate_permille.store((int)(cand * 1000.0 + 0.5), std::memory_order_relaxed);
    P.observed_match_permille.store((int)(std::min(1.0, estimate_selectivity(*root, (double)seg.total_docs)) * 1000.0 + 0.5), std::memory_order_relaxed);
  }
  // pg_fast_dictrange_s family (pg_kernels_specd.hip, round 6): the same shapes — [dense index program] [AND one range scan] [AND the upsert
  // snapshot], one or two <= 8-bit group columns, integer accumulators over ONE INT column — where the scan column and / or the value column is
  // dictionary-encoded (Pinot's default, DictionaryIndexConfig.java:32): the range is a dictId interval over the fixed-bit stream
  // (RangePredicateEvaluatorFactory.java:126-167), the value dictionary.get(dictId) (DataFetcher.java:335-386) — computed for an arithmetic
  // dictionary, gathered from the native-endian copy otherwise.  Raw scan AND raw value stay with pg_fast_i32range_* / pg_pipe_*.
  D.specd = 0;
  // (no GROUP BY — AggregationOperator's shapes, PG_AGG_SINGLE — runs the same kernels with zero group columns: the slot is the lane's replica)
  const bool specd_no_group = q->n_group_by == 0 && D.agg_mode == PG_AGG_SINGLE && D.n_group_cols == 0 && D.n_groups == 1 && D.n_aux == 0;
  if (!D.pipe_fit && !D.pipe_general && P.fast_agg && ((D.agg_mode == PG_AGG_LDS && D.n_group_cols >= 1 && D.n_group_cols <= 2) || specd_no_group) && !knobs().no_specd) {
    bool ok = true;
    int src = -1;
    for (int o = 0; o < D.n_ops && ok; o++) {
      if (D.ops[o].src < 0) continue;
      if (D.ops[o].is_float != PG_ACCV_INT) ok = false;
      if (src >= 0 && D.ops[o].src != src) ok = false;
      src = D.ops[o].src;
    }
    ok = ok && src >= 0;
    for (int g = 0; g < D.n_group_cols && ok; g++) ok = D.gcols[g].col_kind == PG_COL_FIXED_BIT && D.gcols[g].bits >= 1 && D.gcols[g].bits <= 8 && D.mv_gcol_offsets[g] == nullptr;
    int vkind = 0, vbits = 0;
    int64_t vbase = 0, vstep = 0;
    if (ok) {
      const Column* c = srcs[(size_t)src];
      if (D.mv_src_offsets[src] != nullptr || D.mv_src_len[src]) ok = false;
      else if (D.srcs[src].col_kind == PG_COL_RAW32 && c->val_type == PG_V_I32) { vkind = 1; vbits = 32; }
      else if (D.srcs[src].col_kind == PG_COL_FIXED_BIT && c->has_dictionary && c->data_type == PG_TYPE_INT && c->bits >= 1 && c->bits <= 24) {
        vbits = c->bits;
        if (c->dict_affine && c->dict_step > 0 && c->dict_step < ((int64_t)1 << 24) && !knobs().specd_no_affine) { vkind = 2; vbase = c->dict_base; vstep = c->dict_step; }
        else if (D.srcs[src].dict != nullptr) vkind = 3;
        else ok = false;
      } else ok = false;
    }
    const int32_t n_chunks = (int32_t)(((int64_t)seg.total_docs + PG_CHUNK_DOCS - 1) / PG_CHUNK_DOCS);
    const bool index_ok = D.n_index_instr == 0 || D.dense_fused;
    bool has_scan = false, has_tail = false;
    int scan_leaf = -1, sbits = 0;
    const uint8_t* tail = nullptr;
    if (!ok) {
    } else if (P.fast_filter == -1) {
      ok = index_ok;
    } else if ((P.fast_filter == 0 || P.fast_filter == 4 || (P.fast_filter == 100 && D.n_fast_scans == 1)) && (size_t)D.n_index_instr < em.instrs.size()) {
      scan_leaf = em.instrs[(size_t)D.n_index_instr].arg;
      const PgScanLeaf& SL = em.scans[(size_t)scan_leaf];
      has_scan = true;
      ok = index_ok && SL.pred_kind == PG_P_RANGE && !SL.mv &&
           ((SL.col_kind == PG_COL_RAW32 && SL.val_type == PG_V_I32) || (SL.col_kind == PG_COL_FIXED_BIT && SL.bits >= 1 && SL.bits <= 24));
      sbits = SL.col_kind == PG_COL_RAW32 ? 32 : SL.bits;
      if (ok && P.fast_filter == 100 && D.tail_posting >= 0) {   // the upsert snapshot behind index AND scan
        const PgPostingLeaf& TL = em.postings[(size_t)D.tail_posting];
        has_tail = true;
        ok = D.n_index_instr > 0 && !TL.has_csr && TL.n_dense == 1 && TL.dense_chunks >= n_chunks && !TL.exclusive;
        tail = TL.dense[0];
      }
    } else {
      ok = false;
    }
    if (ok && vkind == 1 && (!has_scan || sbits == 32)) ok = false;   // raw scan and raw value: the kernels of rounds 2-5
    // table + a trash slot per lane and accumulator + one strip per wavefront (the sub-tile's column bytes + the selection list) inside the
    // dynamic-LDS limit (device_init asks for 160 KB - 8 KB): fewer replicas where the default table would not leave room (the list walk keeps
    // every lane on a replica of its own down to R = 64; below, lanes share)
    // the shared-stage frame (pg_kernels_specw.hip, late round 6; PG_SPECW=1 only) where the table leaves room for two stage buffers: the workgroup
    // requests whole stages as long rows straight into LDS; same shapes, same results — measured SLOWER than the independent wavefronts (44-47 %
    // against 65 % of 8 TB/s at 2 x 10^8 docs, profiles/r06_specd_steps.txt step 5: with two buffers only one stage is ever pending), kept as a
    // parity-tested measurement variant
    bool stage_frame = false;
    if (ok && knobs().specw && D.n_group_cols > 0) {   // (the shared-stage measurement variant keeps its one or two group columns)
      int n_bm = 0;
      if (D.n_index_instr > 0) {
        n_bm = 8;
        while (n_bm > 1 && D.dense_ptr[n_bm - 1] == D.dense_ptr[0] && D.dense_group[n_bm - 1] == D.dense_group[0]) n_bm--;   // (the padding, as the kernel recognises it)
      }
      const size_t stage = (size_t)pg_specw_stage_bytes(has_scan ? sbits : 0, vbits, D.gcols[0].bits, D.n_group_cols > 1 ? D.gcols[1].bits : 0, n_bm + (has_tail ? 1 : 0));
      const size_t fixed = 256 + 2 * stage + (size_t)pg_specw_list_bytes() + 16 + 512 * (size_t)D.n_ops;
      const size_t wgs_per_cu = (size_t)std::max(1, 16 / pg_specw_waves_per_block);   // 16 wavefronts per CU, as 1 x 16, 2 x 8 or 4 x 4
      const size_t limit = wgs_per_cu == 1 ? (size_t)160 * 1024 - 8192 : (size_t)160 * 1024 / wgs_per_cu - 2048;
      const size_t per_replica = (size_t)G * (size_t)D.n_ops * 8;
      int reps = D.replicas;
      while (reps > 1 && per_replica * (size_t)reps + fixed > limit) reps /= 2;
      // fewer than four replicas only for key spaces that spread a wavefront's 64 atomics by themselves
      if (per_replica * (size_t)reps + fixed <= limit && (reps >= 4 || (size_t)G * (size_t)reps >= 1024 || reps == D.replicas)) {
        P.lds_bytes -= per_replica * (size_t)(D.replicas - reps);
        D.replicas = reps;
        D.replica_shift = 0;
        while ((1 << D.replica_shift) < D.replicas) D.replica_shift++;
        stage_frame = true;
      }
    }
    bool dma = false;
    if (ok && !stage_frame) {
      auto region = [](int bits) { return bits > 0 ? (size_t)((bits * 64 + 16 + 15) & ~15) : (size_t)0; };
      const size_t area = (has_scan ? region(sbits) : 0) + region(vbits) + (D.n_group_cols > 0 ? region(D.gcols[0].bits) : 0) + (D.n_group_cols > 1 ? region(D.gcols[1].bits) : 0);
      const size_t limit = (size_t)160 * 1024 - 8192;
      const size_t per_replica = (size_t)G * (size_t)D.n_ops * 8;
      auto fixed_of = [&](size_t areas) { return 256 + (size_t)pg_specd_waves_per_block * (areas * area + (512 + 64) * 2) + 16 + 512 * (size_t)D.n_ops; };
      // the headline shape (index AND scan, no snapshot) by LDS-DMA into two column areas per strip where the table keeps >= 8 replicas
      // (or all it had) beside the doubled strips: +3-4 % (profiles/r06_specd_steps.txt)
      if (has_scan && D.n_index_instr > 0 && !has_tail && !knobs().specd_no_dma) {
        int reps = D.replicas;
        while (reps > 8 && per_replica * (size_t)reps + fixed_of(2) > limit) reps /= 2;
        if (per_replica * (size_t)reps + fixed_of(2) <= limit) {
          P.lds_bytes -= per_replica * (size_t)(D.replicas - reps);
          D.replicas = reps;
          dma = true;
        }
      }
      const size_t fixed = fixed_of(dma ? 2 : 1);
      while (D.replicas > 1 && per_replica * (size_t)D.replicas + fixed > limit) {
        D.replicas /= 2;
        P.lds_bytes -= per_replica * (size_t)D.replicas;
      }
      D.replica_shift = 0;
      while ((1 << D.replica_shift) < D.replicas) D.replica_shift++;
      if (P.lds_bytes + fixed > limit) ok = false;
    }
    if (ok) {
      D.specd = stage_frame ? 2 : 1;
      D.specd_dma = dma ? 1 : 0;
      D.specd_vkind = vkind;
      D.specd_sbits = has_scan ? sbits : 0;
      D.specd_vbits = vbits;
      D.specd_base = (int32_t)vbase;
      D.specd_step = (int32_t)vstep;
      D.pipe_src = src;
      D.pipe_has_index = D.n_index_instr > 0 ? 1 : 0;
      D.pipe_has_scan = has_scan ? 1 : 0;
      D.pipe_tail = has_tail ? tail : nullptr;
      if (has_scan) D.fast_scan = scan_leaf;
    }
  }
  // pg_nogroup_d (pg_kernels_scan.hip, round 6): AggregationOperator over MatchAllFilterOperator with integer accumulators over ONE dictionary-encoded
  // INT column — the fixed-bit dictId stream summed / min-ed / max-ed in registers (a sorted dictionary: the extreme values sit at the extreme
  // dictIds), values through the arithmetic form of the dictionary or gathered for SUM (DataFetcher.java:335-386)
  D.nogroup_d = 0;
  D.nogroup_lds_card = 0;
  if (q->n_group_by == 0 && D.agg_mode == PG_AGG_SINGLE && P.fast_filter == -1 && D.n_index_instr == 0 && D.tail_posting < 0 && D.n_aux == 0 && D.n_ops > 0 &&
      !D.mv && !knobs().no_scan_pipe) {
    bool ok = true;
    int src = -1;
    for (int o = 0; o < D.n_ops && ok; o++) {
      if (D.ops[o].src < 0) { ok = D.ops[o].fn == PG_ACC_COUNT; continue; }
      if (D.ops[o].is_float != PG_ACCV_INT || D.ops[o].limb != 0) ok = false;
      if (src >= 0 && D.ops[o].src != src) ok = false;
      src = D.ops[o].src;
    }
    if (ok && src >= 0) {
      const Column* c = srcs[(size_t)src];
      if (D.mv_src_offsets[src] == nullptr && !D.mv_src_len[src] && D.srcs[src].col_kind == PG_COL_FIXED_BIT && c->has_dictionary && !c->is_mv &&
          c->data_type == PG_TYPE_INT && c->val_type == PG_V_I32 && c->bits >= 1 && c->bits <= 24) {
        if (c->dict_affine && c->dict_step > 0) { D.nogroup_d = 1; D.nogroup_base = c->dict_base; D.nogroup_step = c->dict_step; }
        else if (D.srcs[src].dict != nullptr) {
          D.nogroup_d = 2; D.nogroup_base = 0; D.nogroup_step = 0;
          D.nogroup_lds_card = c->cardinality <= 36 * 1024 ? c->cardinality : 0;   // 144 KB of LDS
        }
        D.nogroup_src = src;
        D.nogroup_bits = c->bits;
      }
    }
  }
  // LDS tables that miss the narrow shape only by column width (group columns > 8 bits, LONG / DOUBLE sources, 64-bit
  // dictionaries) keep the 1024-thread kernels and run the general aggregator there (pg_fast_none_w / pg_fast_multi_w)
  P.wide_agg = !P.fast_agg && (D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE) && D.n_aux == 0 && P.first_doc_op < 0;
  // range-partitioned tables: pg_fast_none_w / pg_fast_multi_w walk the partitioned tile order with 16 wavefronts per CU
  if (D.agg_mode == PG_AGG_LDS_PART && D.n_aux == 0 && P.fast_filter != -2 && P.first_doc_op < 0) P.wide_agg = true;
  // ... and of those the ones the wide pipeline takes (pg_pipe_w_*, pg_kernels_pipe.hip): integer accumulators over ONE raw INT / LONG column
  // (or COUNT alone), zero to two group columns of <= 16 bits, behind no filter, a fused dense index program, a lone raw-INT range scan, or both
  D.pipe_wide = 0;
  if (P.wide_agg && (D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE) && D.n_group_cols <= 2 &&
      (int64_t)G * D.replicas <= 65536 && !knobs().no_pipe_wide) {
    bool ok = true;
    int src = -1;
    for (int o = 0; o < D.n_ops && ok; o++) {
      if (D.ops[o].src < 0) continue;
      if (src >= 0 && D.ops[o].src != src) ok = false;
      src = D.ops[o].src;
    }
    for (int o = 0; o < D.n_ops && ok; o++) ok = D.ops[o].fn == PG_ACC_COUNT || D.ops[o].fn == PG_ACC_SUM || D.ops[o].fn == PG_ACC_MIN || D.ops[o].fn == PG_ACC_MAX;
    for (int g = 0; g < D.n_group_cols && ok; g++)
      ok = D.gcols[g].col_kind == PG_COL_FIXED_BIT && D.gcols[g].bits >= 1 && D.gcols[g].bits <= 16 && D.mv_gcol_offsets[g] == nullptr;
    int vw = 1;   // value kind of pg_pipe_w*: 1 raw INT, 2 raw LONG, 3 raw DOUBLE
    if (ok && src >= 0) {
      const Column* c = srcs[(size_t)src];
      if (D.srcs[src].col_kind == PG_COL_RAW32 && c->val_type == PG_V_I32) vw = 1;
      else if (D.srcs[src].col_kind == PG_COL_RAW64 && c->val_type == PG_V_I64) vw = 2;
      else if (D.srcs[src].col_kind == PG_COL_RAW64 && c->val_type == PG_V_F64 && !knobs().no_pipe_wide_double) vw = 3;
      else ok = false;
    }
    // accumulator kinds: integers in int64; DOUBLE sums as fixed-point digits (a column with NaN / Inf keeps IEEE additions: not here),
    // DOUBLE MIN / MAX through order keys
    for (int o = 0; o < D.n_ops && ok; o++) {
      if (D.ops[o].src < 0) continue;
      if (vw == 3) ok = D.ops[o].fn == PG_ACC_SUM ? (D.ops[o].is_float == PG_ACCV_FIXED_DIGIT && D.ops[o].limb >= 0 && D.ops[o].limb < 4) : D.ops[o].is_float == PG_ACCV_DOUBLE;
      else ok = D.ops[o].is_float == PG_ACCV_INT;
    }
    // ... and the four digit accumulators of a DOUBLE sum are consecutive rows, limb 0 first (the kernel addresses rows limb0 + j)
    for (int o = 0; o < D.n_ops && ok && vw == 3; o++)
      if (D.ops[o].src >= 0 && D.ops[o].fn == PG_ACC_SUM && D.ops[o].limb == 0)
        for (int j = 1; j < 4 && ok; j++)
          ok = o + j < D.n_ops && D.ops[o + j].fn == PG_ACC_SUM && D.ops[o + j].src == D.ops[o].src && D.ops[o + j].is_float == PG_ACCV_FIXED_DIGIT && D.ops[o + j].limb == j;
    if (ok && D.n_group_cols == 0 && src < 0) ok = false;   // COUNT alone without GROUP BY: nothing to pipeline
    const bool index_ok = D.n_index_instr == 0 || D.dense_fused;
    bool has_scan = false;
    if (P.fast_filter == -1) ok = ok && index_ok;
    else if (P.fast_filter == 4) { has_scan = true; ok = ok && ((D.n_index_instr == 0 && D.fast_scan_pushed) || D.dense_fused); }
    else ok = false;
    if (ok) {
      D.pipe_wide = vw;
      D.pipe_src = src;
      D.pipe_has_index = D.n_index_instr > 0 ? 1 : 0;
      D.pipe_has_scan = has_scan ? 1 : 0;
    }
  }
  D.mv_no_windows = knobs().mv_no_windows ? 1 : 0;
  D.p2_no_pack = knobs().p2_no_pack ? 1 : 0;
  if (D.mv) {   // none of the single-value specialisations reads a multi-value column
    P.fast_filter = -2;
    P.fast_agg = false;
    P.wide_agg = false;
    D.fast_agg_shape = 0;
    D.pipe_fit = 0;
    D.pipe_general = 0;
    D.pipe_wide = 0;
    D.specd = 0;
    D.dense_fused = 0;
    D.tile_split_shift = 0;
    // the commonest multi-value shape as a kernel of its own (pg_kernels_mvg.hip): GROUP BY ONE multi-value column, no filter, integer
    // accumulators over at most one raw INT single-value column, the table in LDS
    D.mvg = 0;
    int mv_cols = 0, mv_at = 0;
    for (int g = 0; g < D.n_group_cols; g++) if (D.mv_gcol_offsets[g] != nullptr) { mv_cols++; mv_at = g; }
    if (!knobs().no_mvg && P.match_all && D.agg_mode == PG_AGG_LDS && D.n_group_cols >= 1 && D.n_group_cols <= 2 && mv_cols == 1 && D.n_aux == 0 &&
        P.first_doc_op < 0 && (int64_t)G * D.replicas <= 65536 && !((q ? q->flags : 0) & PG_QUERY_FLAG_NULL_HANDLING)) {
      bool ok = true;
      for (int g = 0; g < D.n_group_cols; g++) ok = ok && D.gcols[g].col_kind == PG_COL_FIXED_BIT && D.gcols[g].bits >= 1 && D.gcols[g].bits <= 16;
      int src = -1;
      for (int o = 0; o < D.n_ops && ok; o++) {
        if (D.ops[o].is_float != PG_ACCV_INT) ok = false;
        if (D.ops[o].src < 0) continue;
        if (src >= 0 && D.ops[o].src != src) ok = false;
        src = D.ops[o].src;
      }
      if (ok && src >= 0)
        ok = D.srcs[src].col_kind == PG_COL_RAW32 && srcs[(size_t)src]->val_type == PG_V_I32 && D.mv_src_offsets[src] == nullptr && !D.mv_src_len[src];
      if (ok) {
        D.mvg = P.group_cols[(size_t)mv_at]->max_entries_per_doc <= 4 ? 4 : 8;
        D.pipe_src = src;
      }
    }
    // ... and the *MV functions over ONE multi-value INT column (its entries' values and / or their number), grouped by one or two single-value
    // dictionary columns: pg_mv_aggr_*
    D.mvg_has_entries = 0;
    D.mvg_dict_card = 0;
    if (!D.mvg && !knobs().no_mvg && P.match_all && D.agg_mode == PG_AGG_LDS && D.n_group_cols >= 1 && D.n_group_cols <= 2 && D.n_aux == 0 &&
        P.first_doc_op < 0 && (int64_t)G * D.replicas <= 65536 && !((q ? q->flags : 0) & PG_QUERY_FLAG_NULL_HANDLING)) {
      bool ok = true;
      for (int g = 0; g < D.n_group_cols && ok; g++)
        ok = D.mv_gcol_offsets[g] == nullptr && D.gcols[g].col_kind == PG_COL_FIXED_BIT && D.gcols[g].bits >= 1 && D.gcols[g].bits <= 16;
      const int32_t* offsets = nullptr;
      int ent_src = -1, len_src = -1, max_entries = 0;
      for (int o = 0; o < D.n_ops && ok; o++) {
        if (D.ops[o].is_float != PG_ACCV_INT) ok = false;
        const int sidx = D.ops[o].src;
        if (sidx < 0) continue;
        if (D.mv_src_offsets[sidx] == nullptr || (offsets && offsets != D.mv_src_offsets[sidx])) { ok = false; break; }   // one multi-value column, nothing single-valued next to it
        offsets = D.mv_src_offsets[sidx];
        if (D.mv_src_len[sidx]) { len_src = sidx; continue; }
        const Column* c = srcs[(size_t)sidx];
        if (ent_src >= 0 && ent_src != sidx) ok = false;
        ent_src = sidx;
        ok = ok && D.srcs[sidx].col_kind == PG_COL_FIXED_BIT && D.srcs[sidx].val_type == PG_V_I32 && D.srcs[sidx].dict != nullptr && D.srcs[sidx].bits >= 1 && D.srcs[sidx].bits <= 24;
        max_entries = c->max_entries_per_doc;
      }
      if (ok && offsets) {
        D.mvg_has_entries = ent_src >= 0 ? 1 : 0;
        D.mvg_dict_card = 0;
        if (ent_src >= 0) {   // the dictionary in LDS where it is small and the table leaves room
          const int32_t card = srcs[(size_t)ent_src]->cardinality;
          if (card <= 4096 && P.lds_bytes + 512 * (size_t)D.n_ops + (size_t)card * 4 + 1024 <= (size_t)160 * 1024 - 8192) D.mvg_dict_card = card;
        }
        D.mvg = 16 + (ent_src < 0 || max_entries <= 4 ? 4 : 8);
        D.pipe_src = ent_src >= 0 ? ent_src : len_src;
      }
    }
  }
  return plan;
}

}  // namespace pg
