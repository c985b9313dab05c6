// The result of a query as the bytes of a DataTableImplV4 — what a Pinot server sends to the broker for this
// segment-level (or, after pg_result_merge / pg_result_all_reduce, server-level) result, with the INTERMEDIATE results the broker's
// reduce merges.  Restated from:
//   DataTableImplV4#toBytes / writeLeadingSections     pinot-common/.../common/datatable/DataTableImplV4.java:422-517 (13-int header:
//       version 4, rows, columns, then (start, size) of exceptions | string dictionary | data schema | fixed-size rows | variable-size
//       data; the sections in that order; then the metadata's length and bytes), serializeStringDictionary :375-391,
//       serializeMetadata :531-557, DataSchema#toBytes pinot-common/.../common/utils/DataSchema.java:118-143 (names, then type NAMES)
//   DataTableUtils#computeColumnOffsets                pinot-common/.../common/datatable/DataTableUtils.java:41-65 (INT / FLOAT / STRING 4
//       bytes — STRING is an id of the table's own string dictionary —, everything else 8: LONG / DOUBLE values or (position, length))
//   BaseDataTableBuilder / DataTableBuilderV4#setColumn pinot-core/.../core/common/datatable/BaseDataTableBuilder.java, DataTableBuilderV4.java
//       (big-endian; an OBJECT is (position, length of the serialized bytes) in the row and [int object type][bytes] in the variable
//       section; BYTES is (position, length) and the bytes; string ids in first-use order)
//   GroupByResultsBlock#getDataTable :186-236 (a row per group: the keys, then every function's intermediate result),
//   AggregationResultsBlock#getDataTable :104-155 (one row), column names AggregationFunction#getResultColumnName
//       (BaseSingleInputAggregationFunction.java:46-48: lower-case type name + "(" + expression + ")"; CountAggregationFunction.java:37,64:
//       "count(*)"), column types getIntermediateResultColumnType (COUNT LONG; SUM / MIN / MAX DOUBLE; AVG, MINMAXRANGE, DISTINCTCOUNT,
//       DISTINCTCOUNTHLL OBJECT)
//   ObjectSerDeUtils.ObjectType :116-167 and the serializers: AvgPair(4) = double sum, long count (AvgPair.java:57-62); MinMaxRangePair(5) =
//       double min, double max (MinMaxRangePair.java:69-74); HyperLogLog(6) = stream-lib getBytes: int log2m, int byte size, the RegisterSet's
//       ints (six 5-bit registers per int); IntSet(9) / LongSet(15) / FloatSet(16) / DoubleSet(17) = int size + the values; StringSet(18) /
//       BytesSet(19) = int size + (int length, bytes) per value (:893-1039)
// What cannot be byte-identical to a JVM's output, by construction: the ORDER of the rows (IndexedTable iteration order), of a value set's
// elements (fastutil open-addressing order) and of the metadata entries (HashMap order) — here rows come in the result's group order, set
// values ascending by dictId, and the metadata section is empty (the server fills ExecutionStatistics in after the block is built).  Every
// reader of the format rebuilds maps / sets from them.  No JVM exists in this image: the bytes are checked against an independent Python
// restatement of builder AND reader (oracle/po_datatable.py) — parity unpinned against the reference itself.
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "pg_internal.hpp"

namespace pg {
namespace {

struct Out {
  std::vector<uint8_t> b;
  void i32(int32_t v) { const uint32_t u = (uint32_t)v; b.push_back((uint8_t)(u >> 24)); b.push_back((uint8_t)(u >> 16)); b.push_back((uint8_t)(u >> 8)); b.push_back((uint8_t)u); }
  void i64(int64_t v) { i32((int32_t)((uint64_t)v >> 32)); i32((int32_t)(uint32_t)(uint64_t)v); }
  void f32(float v) { int32_t u; memcpy(&u, &v, 4); i32(u); }
  void f64(double v) { int64_t u; memcpy(&u, &v, 8); i64(u); }
  void bytes(const void* p, size_t n) { const uint8_t* s = static_cast<const uint8_t*>(p); b.insert(b.end(), s, s + n); }
  void str(const std::string& s) { i32((int32_t)s.size()); bytes(s.data(), s.size()); }
  size_t size() const { return b.size(); }
};

enum ColType { C_INT, C_LONG, C_FLOAT, C_DOUBLE, C_STRING, C_BYTES, C_OBJECT };
const char* type_name(ColType t) {
  switch (t) {
    case C_INT: return "INT";
    case C_LONG: return "LONG";
    case C_FLOAT: return "FLOAT";
    case C_DOUBLE: return "DOUBLE";
    case C_STRING: return "STRING";
    case C_BYTES: return "BYTES";
    default: return "OBJECT";
  }
}
int fixed_width(ColType t) { return (t == C_INT || t == C_FLOAT || t == C_STRING) ? 4 : 8; }

const char* function_name(int32_t fn) {   // AggregationFunctionType#getName().toLowerCase()
  switch (fn) {
    case PG_AGG_COUNT: return "count";
    case PG_AGG_SUM: return "sum";
    case PG_AGG_MIN: return "min";
    case PG_AGG_MAX: return "max";
    case PG_AGG_AVG: return "avg";
    case PG_AGG_DISTINCTCOUNT: return "distinctcount";
    case PG_AGG_DISTINCTCOUNTHLL: return "distinctcounthll";
    case PG_AGG_MINMAXRANGE: return "minmaxrange";
    case PG_AGG_COUNTMV: return "countmv";
    case PG_AGG_SUMMV: return "summv";
    case PG_AGG_MINMV: return "minmv";
    case PG_AGG_MAXMV: return "maxmv";
    case PG_AGG_AVGMV: return "avgmv";
    case PG_AGG_MINMAXRANGEMV: return "minmaxrangemv";
    case PG_AGG_DISTINCTCOUNTMV: return "distinctcountmv";
    case PG_AGG_DISTINCTCOUNTHLLMV: return "distinctcounthllmv";
    default: fail(PG_ERR_INTERNAL, "aggregation function %d has no name", fn);
  }
  return "";
}

ColType key_type(int32_t data_type) {
  switch (data_type) {
    case PG_TYPE_INT: return C_INT;
    case PG_TYPE_LONG: return C_LONG;
    case PG_TYPE_FLOAT: return C_FLOAT;
    case PG_TYPE_DOUBLE: return C_DOUBLE;
    case PG_TYPE_STRING: return C_STRING;
    case PG_TYPE_BYTES: return C_BYTES;
    default: fail(PG_ERR_UNSUPPORTED, "group-by column of data type %d in a data table", data_type);
  }
  return C_INT;
}

// dictionary value `id` of a column (big-endian fixed-width entries; STRING entries padded with zero bytes: BaseImmutableDictionary)
int64_t dict_i64(const ResultColumn& c, int32_t id) {
  const uint8_t* p = c.dict + (size_t)id * (size_t)c.dict_width;
  uint64_t v = 0;
  for (int i = 0; i < c.dict_width; i++) v = (v << 8) | p[i];
  if (c.dict_width == 4) return (int64_t)(int32_t)(uint32_t)v;
  return (int64_t)v;
}
std::string dict_bytes(const ResultColumn& c, int32_t id, bool strip_padding) {
  const uint8_t* p = c.dict + (size_t)id * (size_t)c.dict_width;
  size_t n = (size_t)c.dict_width;
  if (strip_padding) while (n > 0 && p[n - 1] == 0) n--;
  return std::string(reinterpret_cast<const char*>(p), n);
}

}  // namespace

std::vector<uint8_t> result_data_table_v4(const Result& r) {
  const int n_keys = (int)r.schema_keys.size(), n_aggs = (int)r.schema_aggs.size();
  if (n_aggs != (int)r.aggs.size()) fail(PG_ERR_INVALID_ARGUMENT, "result carries no schema (not produced by pg_query_exec)");
  // the schema's dictionaries live in the segment's columns: a result outlives its segment, its data table cannot
  if (r.schema_segment.expired()) fail(PG_ERR_INVALID_ARGUMENT, "pg_result_data_table_v4: the result's segment was destroyed (its dictionaries give the keys their values)");
  const int32_t n_rows = n_keys ? r.num_groups : 1;
  // ---- schema -------------------------------------------------------------------------------------------------------------------------
  std::vector<std::string> names;
  std::vector<ColType> types;
  for (const ResultColumn& k : r.schema_keys) { names.push_back(k.name); types.push_back(key_type(k.data_type)); }
  for (int a = 0; a < n_aggs; a++) {
    const ResultColumn& c = r.schema_aggs[(size_t)a];
    // CountAggregationFunction#getResultColumnName (:64-66): "count(*)" whatever the argument — unless null handling is on, where COUNT(col)
    // counts the non-null values of col and keeps its argument: "count(col)"
    const bool count_star = c.function == PG_AGG_COUNT && (!r.schema_null_handling || c.name == "*");
    names.push_back(count_star ? std::string("count(*)") : std::string(function_name(c.function)) + "(" + c.name + ")");
    switch (r.aggs[(size_t)a].kind) {
      case PG_RESULT_LONG:
        if (c.function != PG_AGG_COUNT && c.function != PG_AGG_COUNTMV)
          fail(PG_ERR_INVALID_ARGUMENT, "final DISTINCTCOUNT values (PG_QUERY_FLAG_FINAL_DISTINCT) are not intermediate results: no data table");
        types.push_back(C_LONG);
        break;
      case PG_RESULT_DOUBLE: types.push_back(C_DOUBLE); break;
      default: types.push_back(C_OBJECT); break;
    }
  }
  const int n_cols = n_keys + n_aggs;
  std::vector<int> col_off((size_t)n_cols);
  int row_size = 0;
  for (int c = 0; c < n_cols; c++) { col_off[(size_t)c] = row_size; row_size += fixed_width(types[(size_t)c]); }
  // ---- rows ---------------------------------------------------------------------------------------------------------------------------
  Out fixed, var;
  std::unordered_map<std::string, int32_t> string_ids;
  std::vector<std::string> strings;
  auto string_id = [&](const std::string& s) {
    auto it = string_ids.find(s);
    if (it != string_ids.end()) return it->second;
    const int32_t id = (int32_t)strings.size();
    string_ids.emplace(s, id);
    strings.push_back(s);
    return id;
  };
  auto var_bytes = [&](const void* p, size_t n) {   // BYTES: (position, length) + the bytes
    fixed.i32((int32_t)var.size());
    fixed.i32((int32_t)n);
    var.bytes(p, n);
  };
  fixed.b.reserve((size_t)n_rows * (size_t)row_size);
  std::vector<int64_t> set_pos((size_t)n_aggs, 0);   // running offsets into the concatenated dictId sets
  // enableNullHandling (GroupByResultsBlock.java:196-222, AggregationResultsBlock.java:118-140): a null value travels as its column type's
  // placeholder (NullValuePlaceHolder: 0 / "" / empty bytes) with its row id in the column's null bitmap, a null OBJECT as CustomObject's
  // NULL_TYPE_VALUE with no bytes; the bitmaps follow the rows (DataTableBuilderV4#setNullRowIds)
  std::vector<std::vector<int32_t>> null_rows((size_t)n_cols);
  auto key_is_null = [&](int j, int32_t i) { return (size_t)j < r.key_nulls.size() && !r.key_nulls[(size_t)j].empty() && r.key_nulls[(size_t)j][(size_t)i]; };
  auto agg_is_null = [&](int a, int32_t i) { return (size_t)a < r.agg_nulls.size() && !r.agg_nulls[(size_t)a].empty() && r.agg_nulls[(size_t)a][(size_t)i]; };
  for (int32_t i = 0; i < n_rows; i++) {
    for (int j = 0; j < n_keys; j++) {
      const ResultColumn& k = r.schema_keys[(size_t)j];
      const ColType t = types[(size_t)j];
      const int32_t kt = r.group_key_type.empty() ? PG_GROUP_KEY_DICT_IDS : r.group_key_type[(size_t)j];
      if (key_is_null(j, i)) {
        null_rows[(size_t)j].push_back(i);
        switch (t) {
          case C_INT: case C_FLOAT: fixed.i32(0); break;
          case C_LONG: case C_DOUBLE: fixed.i64(0); break;
          case C_STRING: fixed.i32(string_id(std::string())); break;
          default: var_bytes(nullptr, 0); break;
        }
        continue;
      }
      if (kt == PG_GROUP_KEY_DICT_IDS) {
        if (!k.dict) fail(PG_ERR_INTERNAL, "group-by column %s has no dictionary on the host", k.name.c_str());
        const int32_t id = r.group_dict_ids[(size_t)j][(size_t)i];
        switch (t) {
          case C_INT: fixed.i32((int32_t)dict_i64(k, id)); break;
          case C_LONG: fixed.i64(dict_i64(k, id)); break;
          case C_FLOAT: fixed.i32((int32_t)dict_i64(k, id)); break;       // the entry's IEEE bits
          case C_DOUBLE: fixed.i64(dict_i64(k, id)); break;
          case C_STRING: fixed.i32(string_id(dict_bytes(k, id, true))); break;
          default: { const std::string v = dict_bytes(k, id, false); var_bytes(v.data(), v.size()); break; }
        }
      } else if (kt == PG_GROUP_KEY_LONG_VALUES) {
        const int64_t v = r.group_values[(size_t)j][(size_t)i];
        if (t == C_INT) fixed.i32((int32_t)v); else fixed.i64(v);
      } else if (kt == PG_GROUP_KEY_DOUBLE_VALUES) {
        double d;
        memcpy(&d, &r.group_values[(size_t)j][(size_t)i], 8);
        if (t == C_FLOAT) fixed.f32((float)d); else fixed.f64(d);
      } else {
        const auto& off = r.group_bytes_off[(size_t)j];
        const uint8_t* p = r.group_bytes[(size_t)j].data() + off[(size_t)i];
        const size_t n = (size_t)(off[(size_t)i + 1] - off[(size_t)i]);
        if (t == C_STRING) fixed.i32(string_id(std::string(reinterpret_cast<const char*>(p), n))); else var_bytes(p, n);
      }
    }
    for (int a = 0; a < n_aggs; a++) {
      const AggResult& ar = r.aggs[(size_t)a];
      const ResultColumn& c = r.schema_aggs[(size_t)a];
      auto object = [&](int32_t type, const Out& payload) {   // DataTableBuilder#setColumn(int, Object)
        fixed.i32((int32_t)var.size());
        fixed.i32((int32_t)payload.size());
        var.i32(type);
        var.bytes(payload.b.data(), payload.size());
      };
      if (agg_is_null(a, i)) {
        if (ar.kind == PG_RESULT_DOUBLE) { null_rows[(size_t)(n_keys + a)].push_back(i); fixed.f64(0.0); }
        else {   // OBJECT: DataTableBuilder#setColumn(int, null)
          if (n_keys == 0) null_rows[(size_t)(n_keys + a)].push_back(i);   // (AggregationResultsBlock marks every null result; GroupByResultsBlock only the non-OBJECT ones)
          fixed.i32((int32_t)var.size());
          fixed.i32(0);
          var.i32(100);   // CustomObject.NULL_TYPE_VALUE
          if (ar.kind == PG_RESULT_DICTID_SET || ar.kind == PG_RESULT_VALUE_SET) set_pos[(size_t)a] += ar.set_sizes[(size_t)i];
        }
        continue;
      }
      switch (ar.kind) {
        case PG_RESULT_LONG: fixed.i64(ar.l[0][(size_t)i]); break;
        case PG_RESULT_DOUBLE: fixed.f64(ar.d[0][(size_t)i]); break;
        case PG_RESULT_AVG_PAIR: { Out p; p.f64(ar.d[0][(size_t)i]); p.i64(ar.l[0][(size_t)i]); object(4, p); break; }
        case PG_RESULT_MINMAX_PAIR: { Out p; p.f64(ar.d[0][(size_t)i]); p.f64(ar.d[1][(size_t)i]); object(5, p); break; }
        case PG_RESULT_HLL: {
          const int m = 1 << ar.log2m;
          const uint8_t* regs = ar.hll_regs ? ar.hll_regs + (size_t)ar.hll_gids[(size_t)i] * (size_t)ar.hll_stride : ar.hll.data() + (size_t)i * (size_t)m;
          int words = m / 6;   // RegisterSet.getSizeForCount
          words = words == 0 ? 1 : (words % 32 == 0 ? words : words + 1);
          Out p;
          p.i32(ar.log2m);
          p.i32(words * 4);
          for (int w = 0; w < words; w++) {
            uint32_t x = 0;
            for (int s = 0; s < 6; s++) { const int reg = w * 6 + s; if (reg < m) x |= (uint32_t)(regs[reg] & 0x1F) << (5 * s); }
            p.i32((int32_t)x);
          }
          object(6, p);
          break;
        }
        case PG_RESULT_DICTID_SET: {   // the set of VALUES of the group's dictIds, typed by the column (BaseDistinctAggregateAggregationFunction)
          if (!c.dict) fail(PG_ERR_INTERNAL, "DISTINCTCOUNT column %s has no dictionary on the host", c.name.c_str());
          const int32_t n = ar.set_sizes[(size_t)i];
          const int32_t* ids = ar.set_ids.data() + set_pos[(size_t)a];
          set_pos[(size_t)a] += n;
          Out p;
          p.i32(n);
          int32_t type = 9;
          switch (c.data_type) {
            case PG_TYPE_INT: type = 9; for (int32_t e = 0; e < n; e++) p.i32((int32_t)dict_i64(c, ids[e])); break;
            case PG_TYPE_LONG: type = 15; for (int32_t e = 0; e < n; e++) p.i64(dict_i64(c, ids[e])); break;
            case PG_TYPE_FLOAT: type = 16; for (int32_t e = 0; e < n; e++) p.i32((int32_t)dict_i64(c, ids[e])); break;
            case PG_TYPE_DOUBLE: type = 17; for (int32_t e = 0; e < n; e++) p.i64(dict_i64(c, ids[e])); break;
            case PG_TYPE_STRING: type = 18; for (int32_t e = 0; e < n; e++) p.str(dict_bytes(c, ids[e], true)); break;
            default: type = 19; for (int32_t e = 0; e < n; e++) p.str(dict_bytes(c, ids[e], false)); break;
          }
          object(type, p);
          break;
        }
        case PG_RESULT_VALUE_SET: {   // a raw column's typed value set (IntOpenHashSet / LongOpenHashSet / FloatOpenHashSet / DoubleOpenHashSet)
          const int32_t n = ar.set_sizes[(size_t)i];
          const size_t at = set_pos[(size_t)a];
          set_pos[(size_t)a] += n;
          Out p;
          p.i32(n);
          int32_t type = 9;
          switch (ar.set_value_kind) {
            case 0: type = 9; for (int32_t e = 0; e < n; e++) p.i32((int32_t)ar.l[0][at + (size_t)e]); break;
            case 1: type = 15; for (int32_t e = 0; e < n; e++) p.i64(ar.l[0][at + (size_t)e]); break;
            case 2: type = 16; for (int32_t e = 0; e < n; e++) { const float f = (float)ar.d[0][at + (size_t)e]; int32_t b; memcpy(&b, &f, 4); p.i32(b); } break;
            default: type = 17; for (int32_t e = 0; e < n; e++) p.i64(ar.l[0][at + (size_t)e]); break;
          }
          object(type, p);
          break;
        }
        default: fail(PG_ERR_INTERNAL, "result kind %d in a data table", ar.kind);
      }
    }
  }
  if (r.null_handling) {   // (the combined block's table exists even without rows: the trailer is written then as well)
    for (int c = 0; c < n_cols; c++) {
      fixed.i32((int32_t)var.size());
      const std::vector<int32_t>& rows = null_rows[(size_t)c];
      if (rows.empty()) { fixed.i32(0); continue; }
      // RoaringBitmap#serialize (portable format; bitmaps built with add() hold array containers up to 4 096 values, bitmap containers beyond,
      // never run containers): cookie 12346, container count, (key, cardinality - 1) pairs, offsets, containers — little-endian
      Out rb;
      auto le16 = [&](uint32_t v) { rb.b.push_back((uint8_t)v); rb.b.push_back((uint8_t)(v >> 8)); };
      auto le32 = [&](uint32_t v) { le16(v & 0xFFFFu); le16(v >> 16); };
      std::vector<std::pair<uint32_t, std::pair<size_t, size_t>>> conts;   // key, [first, last) into rows
      for (size_t x = 0; x < rows.size();) {
        size_t y = x;
        while (y < rows.size() && ((uint32_t)rows[y] >> 16) == ((uint32_t)rows[x] >> 16)) y++;
        conts.push_back({(uint32_t)rows[x] >> 16, {x, y}});
        x = y;
      }
      le32(12346u);
      le32((uint32_t)conts.size());
      for (auto& ct : conts) { le16(ct.first); le16((uint32_t)(ct.second.second - ct.second.first - 1)); }
      uint32_t pos = 8 + 8 * (uint32_t)conts.size();
      for (auto& ct : conts) { le32(pos); const size_t card = ct.second.second - ct.second.first; pos += card > 4096 ? 8192u : 2u * (uint32_t)card; }
      for (auto& ct : conts) {
        const size_t card = ct.second.second - ct.second.first;
        if (card <= 4096) for (size_t x = ct.second.first; x < ct.second.second; x++) le16((uint32_t)rows[x] & 0xFFFFu);
        else {
          std::vector<uint64_t> words(1024, 0);
          for (size_t x = ct.second.first; x < ct.second.second; x++) { const uint32_t v = (uint32_t)rows[x] & 0xFFFFu; words[v >> 6] |= 1ull << (v & 63); }
          for (uint64_t w : words) { le32((uint32_t)w); le32((uint32_t)(w >> 32)); }
        }
      }
      fixed.i32((int32_t)rb.size());
      var.bytes(rb.b.data(), rb.size());
    }
  }
  // ---- sections -----------------------------------------------------------------------------------------------------------------------
  Out exceptions, dictionary, schema, metadata;
  exceptions.i32(0);
  dictionary.i32((int32_t)strings.size());
  for (const std::string& s : strings) dictionary.str(s);
  schema.i32(n_cols);
  for (const std::string& s : names) schema.str(s);
  for (ColType t : types) schema.str(type_name(t));
  metadata.i32(0);
  // every offset and length below is an int32 (DataTableImplV4's header): sizes are checked BEFORE they are narrowed
  if ((uint64_t)13 * 4 + exceptions.size() + dictionary.size() + schema.size() + fixed.size() + var.size() + metadata.size() + 4 > 0x7FFFFFF0ull)
    fail(PG_ERR_UNSUPPORTED, "data table beyond 2 GB");
  Out out;
  const int32_t header = 13 * 4;
  int32_t off = header;
  out.i32(4);
  out.i32(n_rows);
  out.i32(n_cols);
  out.i32(off); out.i32((int32_t)exceptions.size()); off += (int32_t)exceptions.size();
  out.i32(off); out.i32((int32_t)dictionary.size()); off += (int32_t)dictionary.size();
  out.i32(off); out.i32((int32_t)schema.size()); off += (int32_t)schema.size();
  out.i32(off); out.i32((int32_t)fixed.size()); off += (int32_t)fixed.size();
  out.i32(off); out.i32((int32_t)var.size());
  out.bytes(exceptions.b.data(), exceptions.size());
  out.bytes(dictionary.b.data(), dictionary.size());
  out.bytes(schema.b.data(), schema.size());
  out.bytes(fixed.b.data(), fixed.size());
  out.bytes(var.b.data(), var.size());
  out.i32((int32_t)metadata.size());
  out.bytes(metadata.b.data(), metadata.size());
  return std::move(out.b);
}

}  // namespace pg
