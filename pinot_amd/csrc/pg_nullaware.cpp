// Query-level null handling (PG_QUERY_FLAG_NULL_HANDLING = QueryContext#isNullHandlingEnabled) above the executor.
//
// The kernels know no nulls.  What the reference does per doc — skip the docs whose aggregation argument is null
// (NullableSingleInputAggregationFunction#forEachNotNull / foldNotNull, e.g. SumAggregationFunction.java:100-131,180-215,
// CountAggregationFunction.java:88-98,118-131), give a null group key a group of its own (DefaultGroupByExecutor.java:106-120 switches to the
// value-based key generators, which hold a null key) — is a partition of the matching docs by "which of the columns are null here", and every
// part is an ordinary query whose filter carries IS NULL / IS NOT NULL leaves over the null value vectors (two-valued bitmap leaves:
// FilterPlanNode.java:298-312).  So:
//   * aggregations over a column WITH nulls run as one more query, filter AND (column IS NOT NULL), and are joined to the main result by
//     group key; a group the sub-query did not produce holds no value: SUM / MIN / MAX / AVG / MINMAXRANGE are NULL there, COUNT(col) is 0,
//     the distinct counts an empty set;
//   * k group-by columns with nulls run as 2^k queries (k <= 3), subset S of them IS NULL (and dropped from the GROUP BY), the others
//     IS NOT NULL; their group lists are disjoint and are concatenated, the keys of S flagged NULL.
// The filter itself is three-valued inside the planner (pg_plan.cpp, nh_trues / nh_falses).  The CPU oracle restates the same semantics doc
// at a time (oracle/po_query.c), which is what the parity tests compare this composition with.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

#include "pg_internal.hpp"

namespace pg {
namespace {

bool has_nulls(Segment& seg, const char* name) {
  if (!name || !strcmp(name, "*")) return false;
  auto it = seg.null_vectors.find(name);
  if (it == seg.null_vectors.end() || !it->second) return false;
  return !it->second->posting_card.empty() && it->second->posting_card[0] > 0;
}

// SUM / MIN / MAX / AVG / MINMAXRANGE over no value are NULL; COUNT is 0, DISTINCTCOUNT / DISTINCTCOUNTHLL an empty set
bool null_over_nothing(int32_t function) {
  return function == PG_AGG_SUM || function == PG_AGG_MIN || function == PG_AGG_MAX || function == PG_AGG_AVG || function == PG_AGG_MINMAXRANGE;
}

// a derived query: the filter tree AND (IS NULL / IS NOT NULL leaves), a subset of the group-by columns and of the aggregations
struct Derived {
  std::vector<pg_filter_node> children;   // [original filter?, null leaves...]
  pg_filter_node root{};
  std::vector<const char*> group_by;
  std::vector<pg_agg_spec> aggs;
  pg_query q{};
  void finish(const pg_query& base, bool part) {
    q = base;
    if (part) q.flags |= kQueryFlagNullPartition;
    if (children.size() == 1) q.filter = &children[0];
    else if (children.empty()) q.filter = nullptr;
    else {
      root = pg_filter_node{};
      root.type = PG_FILTER_AND;
      root.n_children = (int32_t)children.size();
      root.children = children.data();
      q.filter = &root;
    }
    q.n_group_by = (int32_t)group_by.size();
    q.group_by_columns = group_by.empty() ? nullptr : group_by.data();
    q.n_aggregations = (int32_t)aggs.size();
    q.aggregations = aggs.data();
  }
  void add_null_leaf(const char* column, bool is_null) {
    pg_filter_node n{};
    n.type = PG_FILTER_PREDICATE;
    n.predicate_type = is_null ? PG_PRED_IS_NULL : PG_PRED_IS_NOT_NULL;
    n.column = column;
    children.push_back(n);
  }
};

// group i's key over the result's group-by columns, as bytes (the results joined here come from the same segment and the same columns:
// equal keys are equal bytes)
std::string key_of(const Result& r, int32_t i) {
  std::string k;
  for (size_t j = 0; j < r.group_key_type.size(); j++) {
    const int32_t t = r.group_key_type[j];
    if (t == PG_GROUP_KEY_DICT_IDS) k.append(reinterpret_cast<const char*>(&r.group_dict_ids[j][(size_t)i]), 4);
    else if (t == PG_GROUP_KEY_BYTES_VALUES) {
      const int64_t a = r.group_bytes_off[j][(size_t)i], b = r.group_bytes_off[j][(size_t)i + 1];
      const int64_t n = b - a;
      k.append(reinterpret_cast<const char*>(&n), 8);
      k.append(reinterpret_cast<const char*>(r.group_bytes[j].data() + a), (size_t)n);
    } else k.append(reinterpret_cast<const char*>(&r.group_values[j][(size_t)i]), 8);
  }
  return k;
}

const uint8_t* hll_row(const AggResult& a, int32_t i, size_t m) {
  if (a.hll_regs) return a.hll_regs + (size_t)a.hll_gids[(size_t)i] * (size_t)a.hll_stride;
  return a.hll.data() + (size_t)i * m;
}

// start of every group's ids in a PG_RESULT_DICTID_SET result's concatenated set (empty for the other kinds)
std::vector<int64_t> set_offsets(const AggResult& a) {
  std::vector<int64_t> off;
  if (a.kind != PG_RESULT_DICTID_SET) return off;
  off.resize(a.set_sizes.size() + 1, 0);
  for (size_t g = 0; g < a.set_sizes.size(); g++) off[g + 1] = off[g] + a.set_sizes[g];
  return off;
}

// appends group `i` of `src` (or, i < 0, "no value") to `dst`; `set_off` = set_offsets(*src)
void append_agg(AggResult& dst, const AggResult* src, int32_t i, int32_t kind, int32_t log2m, const std::vector<int64_t>& set_off) {
  dst.kind = kind;
  dst.log2m = log2m;
  switch (kind) {
    case PG_RESULT_LONG: dst.l[0].push_back(src && i >= 0 ? src->l[0][(size_t)i] : 0); break;
    case PG_RESULT_DOUBLE: dst.d[0].push_back(src && i >= 0 ? src->d[0][(size_t)i] : 0.0); break;
    case PG_RESULT_AVG_PAIR:
      dst.d[0].push_back(src && i >= 0 ? src->d[0][(size_t)i] : 0.0);
      dst.l[0].push_back(src && i >= 0 ? src->l[0][(size_t)i] : 0);
      break;
    case PG_RESULT_MINMAX_PAIR:
      dst.d[0].push_back(src && i >= 0 ? src->d[0][(size_t)i] : 0.0);
      dst.d[1].push_back(src && i >= 0 ? src->d[1][(size_t)i] : 0.0);
      break;
    case PG_RESULT_DICTID_SET: {
      int32_t n = 0;
      if (src && i >= 0) {
        const int64_t off = set_off[(size_t)i];
        n = src->set_sizes[(size_t)i];
        dst.set_ids.insert(dst.set_ids.end(), src->set_ids.begin() + off, src->set_ids.begin() + off + n);
      }
      dst.set_sizes.push_back(n);
      break;
    }
    case PG_RESULT_HLL: {
      const size_t m = (size_t)1 << log2m;
      if (src && i >= 0) { const uint8_t* p = hll_row(*src, i, m); dst.hll.insert(dst.hll.end(), p, p + m); }
      else dst.hll.insert(dst.hll.end(), m, (uint8_t)0);
      break;
    }
    default: fail(PG_ERR_INTERNAL, "null handling: result kind %d", kind);
  }
}

// the result kind a function's intermediate has (what an executed query reports; needed for a function no sub-query produced a group of)
int32_t kind_of(const pg_query& q, int32_t function) {
  const bool final_distinct = (q.flags & PG_QUERY_FLAG_FINAL_DISTINCT) != 0;
  switch (function) {
    case PG_AGG_COUNT: return PG_RESULT_LONG;
    case PG_AGG_AVG: return PG_RESULT_AVG_PAIR;
    case PG_AGG_MINMAXRANGE: return PG_RESULT_MINMAX_PAIR;
    case PG_AGG_DISTINCTCOUNT: return final_distinct ? PG_RESULT_LONG : PG_RESULT_DICTID_SET;
    case PG_AGG_DISTINCTCOUNTHLL: return final_distinct ? PG_RESULT_LONG : PG_RESULT_HLL;
    default: return PG_RESULT_DOUBLE;
  }
}

// One query whose GROUP BY columns hold no null: the aggregations over columns with nulls run as sub-queries and are joined by group key.
// `q` carries the (already null-partitioned) filter.  An aggregation-only query yields one group — or none when `drop_empty` and no doc matches
// (the caller folded GROUP BY columns away: a group that no doc reaches does not exist).
std::unique_ptr<Result> run_joined(Segment& seg, const pg_query& q, const CancelToken* cancel, bool drop_empty) {
  std::vector<std::string> null_cols;   // distinct aggregation arguments with nulls
  std::vector<int> part((size_t)q.n_aggregations, -1);   // aggregation -> index into null_cols, -1: main query
  std::unique_lock<std::mutex> guard(seg.mu);
  for (int a = 0; a < q.n_aggregations; a++) {
    const char* c = q.aggregations[a].column;
    if (!has_nulls(seg, c)) continue;
    size_t k = 0;
    while (k < null_cols.size() && null_cols[k] != c) k++;
    if (k == null_cols.size()) null_cols.push_back(c);
    part[(size_t)a] = (int)k;
  }
  guard.unlock();
  // main query: the aggregations without nulls — and a COUNT(*) of its own when there is none (it defines the groups)
  Derived main;
  if (q.filter) main.children.push_back(*q.filter);
  for (int j = 0; j < q.n_group_by; j++) main.group_by.push_back(q.group_by_columns[j]);
  std::vector<int> main_pos((size_t)q.n_aggregations, -1);
  for (int a = 0; a < q.n_aggregations; a++)
    if (part[(size_t)a] < 0) { main_pos[(size_t)a] = (int)main.aggs.size(); main.aggs.push_back(q.aggregations[a]); }
  if (main.aggs.empty()) main.aggs.push_back(pg_agg_spec{PG_AGG_COUNT, 0, "*"});
  main.finish(q, !null_cols.empty());   // (the query itself when nothing is joined: the reference's plan, FastFilteredCountOperator included)
  std::unique_ptr<Result> R = execute_query_plain(seg, main.q, cancel);
  const bool agg_only = q.n_group_by == 0;
  const bool nothing = agg_only && R->stats.num_docs_scanned == 0;
  if (null_cols.empty() && !nothing) {   // nothing to join
    R->agg_nulls.assign((size_t)q.n_aggregations, {});
    R->key_nulls.assign((size_t)q.n_group_by, {});
    return R;
  }
  const int32_t ng = (nothing && drop_empty) ? 0 : R->num_groups;
  // sub-queries
  std::vector<std::unique_ptr<Result>> subs;
  std::vector<std::vector<int>> sub_pos(null_cols.size(), std::vector<int>((size_t)q.n_aggregations, -1));
  std::vector<std::unordered_map<std::string, int32_t>> sub_index(null_cols.size());
  for (size_t k = 0; k < null_cols.size(); k++) {
    Derived d;
    if (q.filter) d.children.push_back(*q.filter);
    d.add_null_leaf(null_cols[k].c_str(), false);
    for (int j = 0; j < q.n_group_by; j++) d.group_by.push_back(q.group_by_columns[j]);
    for (int a = 0; a < q.n_aggregations; a++)
      if (part[(size_t)a] == (int)k) {
        sub_pos[k][(size_t)a] = (int)d.aggs.size();
        pg_agg_spec s = q.aggregations[a];
        if (s.function == PG_AGG_COUNT) s.column = "*";   // COUNT(col) over the docs that hold a value
        d.aggs.push_back(s);
      }
    d.finish(q, true);
    // the main query decides which groups exist (numGroupsLimit admits keys in docId order over ALL the matching docs): a sub-query trims nothing
    d.q.num_groups_limit = INT32_MAX;
    subs.push_back(execute_query_plain(seg, d.q, cancel));
    const Result& S = *subs.back();
    if (!agg_only)
      for (int32_t i = 0; i < S.num_groups; i++) sub_index[k][key_of(S, i)] = i;
  }
  // the joined result: the main query's groups in its order
  auto out = std::make_unique<Result>();
  out->num_groups = ng;
  out->group_key_type = R->group_key_type;
  out->group_dict_ids = R->group_dict_ids;
  out->group_values = R->group_values;
  out->group_bytes = R->group_bytes;
  out->group_bytes_off = R->group_bytes_off;
  if (ng == 0 && R->num_groups > 0) {
    for (auto& v : out->group_dict_ids) v.clear();
    for (auto& v : out->group_values) v.clear();
  }
  out->stats = R->stats;
  out->aggs.resize((size_t)q.n_aggregations);
  out->agg_nulls.assign((size_t)q.n_aggregations, {});
  out->key_nulls.assign((size_t)q.n_group_by, {});
  for (int a = 0; a < q.n_aggregations; a++) {
    const int32_t fn = q.aggregations[a].function;
    AggResult& dst = out->aggs[(size_t)a];
    const int k = part[(size_t)a];
    const Result& src_r = k < 0 ? *R : *subs[(size_t)k];
    const AggResult& src = src_r.aggs[(size_t)(k < 0 ? main_pos[(size_t)a] : sub_pos[(size_t)k][(size_t)a])];
    const int32_t log2m = src.log2m ? src.log2m : (q.aggregations[a].log2m > 0 ? q.aggregations[a].log2m : 8);
    const int32_t kind = src_r.num_groups > 0 ? src.kind : kind_of(q, fn);
    std::vector<uint8_t> nulls;
    bool any_null = false;
    const std::vector<int64_t> set_off = set_offsets(src);
    for (int32_t i = 0; i < ng; i++) {
      int32_t si = i;
      bool empty = false;
      if (agg_only) empty = src_r.stats.num_docs_scanned == 0;
      else if (k >= 0) {
        auto it = sub_index[(size_t)k].find(key_of(*R, i));
        if (it == sub_index[(size_t)k].end()) empty = true;
        else si = it->second;
      }
      append_agg(dst, empty ? nullptr : &src, empty ? -1 : si, kind, log2m, set_off);
      const bool is_null = empty && null_over_nothing(fn);
      nulls.push_back(is_null ? 1 : 0);
      any_null |= is_null;
    }
    if (ng == 0) { dst.kind = kind; dst.log2m = log2m; }
    if (any_null) out->agg_nulls[(size_t)a] = std::move(nulls);
  }
  for (auto& s : subs) out->stats.host_ms_total += s->stats.host_ms_total;
  return out;
}

// appends the groups of `part` (GROUP BY over the columns `kept` of the original GROUP BY; the others are NULL in every group) to `out`
void append_groups(Result& out, const Result& part, const pg_query& q, const std::vector<int>& kept, bool first_partition) {
  const int32_t n = part.num_groups;
  const int32_t before = out.num_groups;
  for (int j = 0; j < q.n_group_by; j++) {
    int kj = -1;
    for (size_t x = 0; x < kept.size(); x++) if (kept[x] == j) kj = (int)x;
    const bool dropped = kj < 0;
    std::vector<uint8_t>& kn = out.key_nulls[(size_t)j];
    if (dropped && kn.empty()) kn.assign((size_t)before, 0);
    if (!kn.empty() || dropped) kn.insert(kn.end(), (size_t)n, dropped ? 1 : 0);
    // the key arrays: a dropped column holds a placeholder under the NULL flag — dictId 0, value 0, an empty byte string (the key types are
    // those of the first partition, which keeps every column)
    if (dropped) {
      const int32_t t = out.group_key_type[(size_t)j];
      if (t == PG_GROUP_KEY_DICT_IDS) out.group_dict_ids[(size_t)j].insert(out.group_dict_ids[(size_t)j].end(), (size_t)n, 0);
      else if (t == PG_GROUP_KEY_BYTES_VALUES) {
        auto& off = out.group_bytes_off[(size_t)j];
        if (off.empty()) off.push_back(0);
        off.insert(off.end(), (size_t)n, off.back());
      } else out.group_values[(size_t)j].insert(out.group_values[(size_t)j].end(), (size_t)n, 0);
      continue;
    }
    const int32_t t = part.group_key_type[(size_t)kj];
    if (first_partition) out.group_key_type[(size_t)j] = t;
    if (out.group_key_type[(size_t)j] != t) fail(PG_ERR_INTERNAL, "null handling: group key types differ between the null partitions");
    if (t == PG_GROUP_KEY_DICT_IDS) out.group_dict_ids[(size_t)j].insert(out.group_dict_ids[(size_t)j].end(), part.group_dict_ids[(size_t)kj].begin(), part.group_dict_ids[(size_t)kj].begin() + n);
    else if (t == PG_GROUP_KEY_BYTES_VALUES) {
      auto& off = out.group_bytes_off[(size_t)j];
      auto& bytes = out.group_bytes[(size_t)j];
      if (off.empty()) off.push_back(0);
      const int64_t base = off.back();
      for (int32_t i = 0; i < n; i++) off.push_back(base + part.group_bytes_off[(size_t)kj][(size_t)i + 1]);
      bytes.insert(bytes.end(), part.group_bytes[(size_t)kj].begin(), part.group_bytes[(size_t)kj].begin() + part.group_bytes_off[(size_t)kj][(size_t)n]);
    } else out.group_values[(size_t)j].insert(out.group_values[(size_t)j].end(), part.group_values[(size_t)kj].begin(), part.group_values[(size_t)kj].begin() + n);
  }
  for (int a = 0; a < q.n_aggregations; a++) {
    AggResult& dst = out.aggs[(size_t)a];
    const AggResult& src = part.aggs[(size_t)a];
    std::vector<uint8_t>& an = out.agg_nulls[(size_t)a];
    const std::vector<uint8_t>& pn = part.agg_nulls.empty() ? std::vector<uint8_t>() : part.agg_nulls[(size_t)a];
    if (!pn.empty() || !an.empty()) {   // (flags exist once any partition brought one: zeros for the groups before and the partitions without)
      an.resize((size_t)before, 0);
      if (pn.empty()) an.insert(an.end(), (size_t)n, 0); else an.insert(an.end(), pn.begin(), pn.begin() + n);
    }
    const int32_t log2m = src.log2m ? src.log2m : (q.aggregations[a].log2m > 0 ? q.aggregations[a].log2m : 8);
    const std::vector<int64_t> set_off = set_offsets(src);
    for (int32_t i = 0; i < n; i++) append_agg(dst, &src, i, src.kind, log2m, set_off);
    if (n == 0 && before == 0) { dst.kind = src.kind; dst.log2m = log2m; }
  }
  out.num_groups = before + n;
}

// ---- segment-level group trim over a joined result (GroupByOperator.java:120-133 -> TableResizer#trimInSegmentResults :327-351 with the
//      null-aware comparator of TableResizer.java:98-116: a null order-by value sorts by isNullsLast, whatever the expression's direction).
//      The parts ran untrimmed (which groups survive is decided over ALL of them); order-by values as pg_exec.hip's trim extracts them: a group
//      key's value (dictIds of sorted dictionaries order as the values do), an aggregation's final result.
struct OrderValue { int type = 0; int64_t l = 0; double d = 0; const uint8_t* b = nullptr; int64_t blen = 0; bool is_null = false; };   // 0 long, 1 double, 2 BYTES, 3 STRING
int double_compare(double a, double b) {   // Double.compare
  if (a < b) return -1;
  if (a > b) return 1;
  int64_t x, y;
  memcpy(&x, &a, 8); memcpy(&y, &b, 8);
  if (a != a) x = INT64_MAX;
  if (b != b) y = INT64_MAX;
  return x == y ? 0 : (x < y ? -1 : 1);
}
int utf16_unit_order(const uint8_t* a, int64_t alen, const uint8_t* b, int64_t blen) {   // String.compareTo over UTF-8 bytes (see pg_exec.hip)
  const int64_t m = std::min(alen, blen);
  for (int64_t i = 0; i < m; i++) {
    if (a[i] == b[i]) continue;
    const int x = a[i] == 0xEE || a[i] == 0xEF ? a[i] + 0x10 : a[i], y = b[i] == 0xEE || b[i] == 0xEF ? b[i] + 0x10 : b[i];
    return x < y ? -1 : 1;
  }
  return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
void trim_joined(Segment& seg, const pg_query& q, Result& r) {
  const int64_t by_limit = (int64_t)std::max(q.limit, 0) * 5;   // GroupByUtils.getTableCapacity
  const int32_t trim_size = by_limit > INT32_MAX ? INT32_MAX : std::max((int32_t)by_limit, q.min_segment_group_trim_size);
  const size_t n = (size_t)r.num_groups, n_ob = (size_t)q.n_order_by;
  for (size_t k = 0; k < n_ob; k++) {
    const pg_order_by& ob = q.order_by[k];
    if (ob.kind == PG_ORDER_BY_GROUP_KEY) { if (ob.index < 0 || ob.index >= q.n_group_by) fail(PG_ERR_INVALID_ARGUMENT, "ORDER BY group-by expression %d of %d", ob.index, q.n_group_by); }
    else if (ob.kind == PG_ORDER_BY_AGGREGATION) { if (ob.index < 0 || ob.index >= q.n_aggregations) fail(PG_ERR_INVALID_ARGUMENT, "ORDER BY aggregation %d of %d", ob.index, q.n_aggregations); }
    else fail(PG_ERR_INVALID_ARGUMENT, "ORDER BY expression kind %d", ob.kind);
  }
  if ((int64_t)n <= (int64_t)trim_size) return;
  std::vector<OrderValue> vals(n * n_ob);
  for (size_t k = 0; k < n_ob; k++) {
    const pg_order_by& ob = q.order_by[k];
    if (ob.kind == PG_ORDER_BY_AGGREGATION) {
      const AggResult& a = r.aggs[(size_t)ob.index];
      const std::vector<uint8_t>& nulls = r.agg_nulls[(size_t)ob.index];
      for (size_t i = 0; i < n; i++) {
        OrderValue& v = vals[i * n_ob + k];
        v.is_null = !nulls.empty() && nulls[i];
        switch (a.kind) {
          case PG_RESULT_LONG: v.type = 0; v.l = a.l[0][i]; break;
          case PG_RESULT_DOUBLE: v.type = 1; v.d = a.d[0][i]; break;
          case PG_RESULT_AVG_PAIR: v.type = 1; v.d = a.l[0][i] == 0 ? -INFINITY : a.d[0][i] / (double)a.l[0][i]; break;
          case PG_RESULT_MINMAX_PAIR: v.type = 1; v.d = a.d[1][i] - a.d[0][i]; break;
          case PG_RESULT_DICTID_SET: v.type = 0; v.l = a.set_sizes[i]; break;
          case PG_RESULT_HLL: v.type = 0; v.l = hll_cardinality(hll_row(a, (int32_t)i, (size_t)1 << a.log2m), a.log2m); break;
          default: fail(PG_ERR_UNSUPPORTED, "segment-level group trim under enableNullHandling ordered by a result of kind %d", a.kind);
        }
      }
      continue;
    }
    const size_t j = (size_t)ob.index;
    const std::vector<uint8_t>& nulls = r.key_nulls[j];
    const int32_t t = r.group_key_type[j];
    int32_t data_type = PG_TYPE_STRING;
    if (t == PG_GROUP_KEY_BYTES_VALUES) {
      std::lock_guard<std::mutex> g(seg.mu);
      Column* c = seg.find(q.group_by_columns[j]);
      if (c) data_type = c->data_type;
    }
    for (size_t i = 0; i < n; i++) {
      OrderValue& v = vals[i * n_ob + k];
      v.is_null = !nulls.empty() && nulls[i];
      if (t == PG_GROUP_KEY_DICT_IDS) v.l = r.group_dict_ids[j][i];
      else if (t == PG_GROUP_KEY_LONG_VALUES) v.l = r.group_values[j][i];
      else if (t == PG_GROUP_KEY_DOUBLE_VALUES) { v.type = 1; memcpy(&v.d, &r.group_values[j][i], 8); }
      else {
        v.type = data_type == PG_TYPE_STRING ? 3 : 2;
        v.b = r.group_bytes[j].data() + r.group_bytes_off[j][i];
        v.blen = r.group_bytes_off[j][i + 1] - r.group_bytes_off[j][i];
      }
    }
  }
  std::vector<int32_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
  auto before = [&](int32_t ia, int32_t ib) {
    for (size_t k = 0; k < n_ob; k++) {
      const OrderValue& a = vals[(size_t)ia * n_ob + k];
      const OrderValue& b = vals[(size_t)ib * n_ob + k];
      if (a.is_null || b.is_null) {
        if (a.is_null && b.is_null) continue;
        return a.is_null ? !q.order_by[k].nulls_last : (bool)q.order_by[k].nulls_last;
      }
      int c;
      if (a.type == 0) c = a.l < b.l ? -1 : (a.l > b.l ? 1 : 0);
      else if (a.type == 1) c = double_compare(a.d, b.d);
      else if (a.type == 3) c = utf16_unit_order(a.b, a.blen, b.b, b.blen);
      else {
        const int64_t m = std::min(a.blen, b.blen);
        c = m ? memcmp(a.b, b.b, (size_t)m) : 0;
        if (c == 0) c = a.blen < b.blen ? -1 : (a.blen > b.blen ? 1 : 0);
      }
      if (c != 0) return q.order_by[k].ascending ? c < 0 : c > 0;
    }
    return ia < ib;
  };
  std::nth_element(order.begin(), order.begin() + trim_size, order.end(), before);
  order.resize((size_t)trim_size);
  std::sort(order.begin(), order.end());
  // the survivors, in the joined result's order
  const size_t m = order.size();
  for (size_t j = 0; j < (size_t)q.n_group_by; j++) {
    const int32_t t = r.group_key_type[j];
    if (t == PG_GROUP_KEY_DICT_IDS) {
      std::vector<int32_t> ids(m);
      for (size_t i = 0; i < m; i++) ids[i] = r.group_dict_ids[j][(size_t)order[i]];
      r.group_dict_ids[j].swap(ids);
    } else if (t == PG_GROUP_KEY_BYTES_VALUES) {
      std::vector<uint8_t> bytes;
      std::vector<int64_t> off(1, 0);
      for (size_t i = 0; i < m; i++) {
        const int64_t a = r.group_bytes_off[j][(size_t)order[i]], b = r.group_bytes_off[j][(size_t)order[i] + 1];
        bytes.insert(bytes.end(), r.group_bytes[j].begin() + a, r.group_bytes[j].begin() + b);
        off.push_back((int64_t)bytes.size());
      }
      r.group_bytes[j].swap(bytes);
      r.group_bytes_off[j].swap(off);
    } else {
      std::vector<int64_t> vs(m);
      for (size_t i = 0; i < m; i++) vs[i] = r.group_values[j][(size_t)order[i]];
      r.group_values[j].swap(vs);
    }
    if (!r.key_nulls[j].empty()) {
      std::vector<uint8_t> kn(m);
      for (size_t i = 0; i < m; i++) kn[i] = r.key_nulls[j][(size_t)order[i]];
      r.key_nulls[j].swap(kn);
    }
  }
  for (size_t a = 0; a < (size_t)q.n_aggregations; a++) {
    AggResult dst;
    const AggResult& src = r.aggs[a];
    const std::vector<int64_t> set_off = set_offsets(src);
    for (size_t i = 0; i < m; i++) append_agg(dst, &src, order[i], src.kind, src.log2m, set_off);
    r.aggs[a] = std::move(dst);
    if (!r.agg_nulls[a].empty()) {
      std::vector<uint8_t> an(m);
      for (size_t i = 0; i < m; i++) an[i] = r.agg_nulls[a][(size_t)order[i]];
      r.agg_nulls[a].swap(an);
    }
  }
  r.num_groups = (int32_t)m;
}

}  // namespace

// PG_QUERY_FLAG_NULL_HANDLING: what stays with the Java plan
void check_null_handling(Segment& seg, const pg_query& q) {
  if (!(q.flags & PG_QUERY_FLAG_NULL_HANDLING)) return;
  std::lock_guard<std::mutex> g(seg.mu);
  int null_keys = 0;
  bool mv_keys = false, any_nulls = false;
  for (int j = 0; j < q.n_group_by; j++) {
    Column* c = seg.find(q.group_by_columns[j]);
    if (!c) continue;
    mv_keys |= c->is_mv;
    if (!has_nulls(seg, q.group_by_columns[j])) continue;
    null_keys++;
    any_nulls = true;
    if (c->is_mv) fail(PG_ERR_UNSUPPORTED, "enableNullHandling: nulls in the multi-value group-by column %s", c->name.c_str());
  }
  if (null_keys > 3) fail(PG_ERR_UNSUPPORTED, "enableNullHandling: %d group-by columns hold nulls (at most 3 are partitioned)", null_keys);
  if (null_keys && mv_keys) fail(PG_ERR_UNSUPPORTED, "enableNullHandling: null group keys next to a multi-value group-by column");
  for (int a = 0; a < q.n_aggregations; a++) {
    const char* name = q.aggregations[a].column;
    if (!has_nulls(seg, name)) continue;
    any_nulls = true;
    Column* c = seg.find(name);
    if (c && c->is_mv) fail(PG_ERR_UNSUPPORTED, "enableNullHandling: nulls in the multi-value column %s", c->name.c_str());
  }
  if (any_nulls && (q.flags & PG_QUERY_FLAG_KEEP_DEVICE_TABLE))
    fail(PG_ERR_UNSUPPORTED, "enableNullHandling over columns with nulls: the result is joined on the host, no device table to keep");
}

std::unique_ptr<Result> execute_query(Segment& seg, const pg_query& q_in, const CancelToken* cancel) {
  pg_query q = q_in;
  q.flags &= ~kQueryFlagNullPartition;   // internal
  if (!(q.flags & PG_QUERY_FLAG_NULL_HANDLING)) return execute_query_plain(seg, q, cancel);
  if (q.n_aggregations <= 0 || !q.aggregations) fail(PG_ERR_INVALID_ARGUMENT, "query has no aggregation");
  check_null_handling(seg, q);
  std::vector<int> null_keys;
  bool null_args = false;
  {
    std::lock_guard<std::mutex> g(seg.mu);
    for (int j = 0; j < q.n_group_by; j++) if (has_nulls(seg, q.group_by_columns[j])) null_keys.push_back(j);
    for (int a = 0; a < q.n_aggregations; a++) null_args |= has_nulls(seg, q.aggregations[a].column);
  }
  // segment-level group trim: over columns without nulls the query's own (no order-by value can be null); else the parts run untrimmed and the
  // joined result is trimmed with the null-aware comparator
  const bool trim = q.n_group_by > 0 && q.n_order_by > 0 && q.order_by && q.min_segment_group_trim_size > 0 && (null_args || !null_keys.empty());
  const pg_query q_trim = q;
  if (trim) { q.n_order_by = 0; q.order_by = nullptr; q.min_segment_group_trim_size = 0; }
  if (null_keys.empty()) {
    auto r = run_joined(seg, q, cancel, false);
    if (trim) trim_joined(seg, q_trim, *r);
    if (null_args || r->schema_aggs.empty()) fill_result_schema(seg, q_trim, *r);
    r->null_handling = true;
    return r;
  }
  // 2^k partitions of the matching docs by which of the nullable group-by columns are null
  auto out = std::make_unique<Result>();
  out->group_key_type.assign((size_t)q.n_group_by, PG_GROUP_KEY_DICT_IDS);
  out->group_dict_ids.assign((size_t)q.n_group_by, {});
  out->group_values.assign((size_t)q.n_group_by, {});
  out->group_bytes.assign((size_t)q.n_group_by, {});
  out->group_bytes_off.assign((size_t)q.n_group_by, {});
  out->aggs.resize((size_t)q.n_aggregations);
  out->agg_nulls.assign((size_t)q.n_aggregations, {});
  out->key_nulls.assign((size_t)q.n_group_by, {});
  const int32_t limit = q.num_groups_limit > 0 ? q.num_groups_limit : 100000;
  bool first = true;
  for (uint32_t s = 0; s < (1u << null_keys.size()); s++) {
    Derived d;
    if (q.filter) d.children.push_back(*q.filter);
    std::vector<int> kept;
    for (int j = 0; j < q.n_group_by; j++) {
      int x = -1;
      for (size_t t = 0; t < null_keys.size(); t++) if (null_keys[t] == j) x = (int)t;
      const bool is_null = x >= 0 && ((s >> x) & 1u);
      if (x >= 0) d.add_null_leaf(q.group_by_columns[j], is_null);
      if (!is_null) { kept.push_back(j); d.group_by.push_back(q.group_by_columns[j]); }
    }
    for (int a = 0; a < q.n_aggregations; a++) d.aggs.push_back(q.aggregations[a]);
    d.finish(q, true);
    auto part = run_joined(seg, d.q, cancel, true);
    if (first) out->stats = part->stats;
    else {
      out->stats.num_docs_scanned += part->stats.num_docs_scanned;
      out->stats.num_entries_scanned_post_filter += part->stats.num_entries_scanned_post_filter;
      out->stats.host_ms_total += part->stats.host_ms_total;
      out->stats.device_ms_total += part->stats.device_ms_total;
    }
    append_groups(*out, *part, q, kept, s == 0);
    first = false;
  }
  // the reference admits the first numGroupsLimit keys in docId order over ALL the docs; which ones that is across the partitions is not
  // restated: refuse rather than return another subset
  if (out->num_groups > limit)
    fail(PG_ERR_UNSUPPORTED, "enableNullHandling: %d groups over the null partitions, more than numGroupsLimit (%d)", out->num_groups, limit);
  out->stats.num_groups_limit_reached = out->num_groups >= limit ? 1 : 0;
  out->stats.stats_exact = 0;   // the filter ran once per partition: numEntriesScannedInFilter is the first partition's
  if (trim) trim_joined(seg, q_trim, *out);
  fill_result_schema(seg, q_trim, *out);
  out->null_handling = true;
  return out;
}

}  // namespace pg
