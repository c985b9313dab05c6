// C ABI entry points of libpinot_gpu.so (include/pinot_gpu.h).  Every exception becomes a status code plus a
// thread-local message (pg_last_error), which the JNI shim rethrows as RuntimeException.
#include "pg_internal.hpp"

using namespace pg;

struct pg_segment_s { Segment seg; };
struct pg_result_s { std::unique_ptr<Result> r; };
struct pg_docidset_s { std::unique_ptr<DocIdSet> s; };

template <typename F>
static int32_t guarded(F&& f) {
  try {
    f();
    return PG_OK;
  } catch (const Error& e) {
    set_last_error(e.what());
    return e.status;
  } catch (const std::bad_alloc&) {
    set_last_error("host out of memory");
    return PG_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return PG_ERR_INTERNAL;
  }
}
#define REQUIRE(cond, msg) do { if (!(cond)) fail(PG_ERR_INVALID_ARGUMENT, "%s", msg); } while (0)

extern "C" {

int32_t pg_abi_version(void) { return PG_ABI_VERSION; }

int32_t pg_init(int32_t device_ordinal) {
  return guarded([&] { device_init(device_ordinal); });
}

int32_t pg_device_count(int32_t* out_count) {
  return guarded([&] {
    REQUIRE(out_count, "out_count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *out_count = e == hipSuccess ? n : 0;
  });
}

int32_t pg_last_error(char* buf, size_t cap) {
  const std::string& e = last_error();
  if (buf && cap) {
    size_t k = e.size() < cap - 1 ? e.size() : cap - 1;
    memcpy(buf, e.data(), k);
    buf[k] = 0;
  }
  return (int32_t)e.size();
}

int32_t pg_segment_create(const char* segment_name, int32_t total_docs, pg_segment_t* out_segment) {
  return guarded([&] {
    REQUIRE(out_segment, "out_segment is null");
    REQUIRE(total_docs >= 0, "total_docs < 0");
    auto* s = new pg_segment_s();
    s->seg.name = segment_name ? segment_name : "";
    s->seg.total_docs = total_docs;
    s->seg.n_tiles = (int32_t)(((int64_t)total_docs + PG_TILE_DOCS - 1) / PG_TILE_DOCS);
    if (s->seg.n_tiles == 0) s->seg.n_tiles = 1;
    *out_segment = s;
  });
}

int32_t pg_segment_add_column(pg_segment_t segment, const pg_column_desc* column) {
  return guarded([&] {
    REQUIRE(segment && column, "null argument");
    std::lock_guard<std::mutex> g(segment->seg.mu);
    segment_add_column(segment->seg, *column);
    segment->seg.plan_cache.clear();
  });
}

int32_t pg_segment_add_star_tree(pg_segment_t segment, const pg_star_tree_desc* star_tree) {
  return guarded([&] {
    REQUIRE(segment && star_tree, "null argument");
    std::lock_guard<std::mutex> g(segment->seg.mu);
    segment_add_star_tree(segment->seg, *star_tree);
    segment->seg.plan_cache.clear();
  });
}
int32_t pg_segment_set_null_vector(pg_segment_t segment, const char* column, const void* roaring, uint64_t size) {
  return guarded([&] {
    REQUIRE(segment && column, "null argument");
    segment_set_null_vector(segment->seg, column, roaring, size);
  });
}
int32_t pg_segment_set_queryable_doc_ids(pg_segment_t segment, const void* roaring, uint64_t size) {
  return guarded([&] {
    REQUIRE(segment, "null argument");
    segment_set_queryable_doc_ids(segment->seg, roaring, size);
  });
}

int32_t pg_segment_num_docs(pg_segment_t segment, int32_t* out_num_docs) {
  return guarded([&] { REQUIRE(segment && out_num_docs, "null argument"); *out_num_docs = segment->seg.total_docs; });
}

int32_t pg_segment_device_bytes(pg_segment_t segment, uint64_t* out_bytes) {
  return guarded([&] { REQUIRE(segment && out_bytes, "null argument"); *out_bytes = segment->seg.device_bytes; });
}

int32_t pg_segment_destroy(pg_segment_t segment) {
  return guarded([&] { delete segment; });
}

int32_t pg_filter_exec(pg_segment_t segment, const pg_filter_node* filter, pg_docidset_t* out_docidset) {
  return guarded([&] {
    REQUIRE(segment && out_docidset, "null argument");
    auto s = execute_filter(segment->seg, filter);
    auto* h = new pg_docidset_s();
    h->s = std::move(s);
    *out_docidset = h;
  });
}
int32_t pg_docidset_cardinality(pg_docidset_t set, int64_t* out) {
  return guarded([&] { REQUIRE(set && out, "null argument"); *out = set->s->cardinality; });
}
int32_t pg_docidset_num_words(pg_docidset_t set, int64_t* out) {
  return guarded([&] { REQUIRE(set && out, "null argument"); *out = ((int64_t)set->s->num_docs + 63) / 64; });
}
int32_t pg_docidset_copy_words(pg_docidset_t set, uint64_t* out_words, int64_t capacity_words) {
  return guarded([&] {
    REQUIRE(set && (out_words || capacity_words == 0), "null argument");
    int64_t n = ((int64_t)set->s->num_docs + 63) / 64;
    REQUIRE(capacity_words >= n, "capacity too small");
    if (n) PG_HIP(hipMemcpy(out_words, set->s->words.ptr, (size_t)n * 8, hipMemcpyDeviceToHost));
  });
}
int32_t pg_docidset_copy_docids(pg_docidset_t set, int32_t* out_docids, int64_t capacity) {
  return guarded([&] {
    REQUIRE(set && (out_docids || capacity == 0), "null argument");
    docidset_copy_docids(*set->s, out_docids, capacity);
  });
}
int32_t pg_docidset_stats(pg_docidset_t set, pg_exec_stats* out_stats) {
  return guarded([&] { REQUIRE(set && out_stats, "null argument"); *out_stats = set->s->stats; });
}
int32_t pg_docidset_free(pg_docidset_t set) {
  return guarded([&] { delete set; });
}

int32_t pg_query_supported(pg_segment_t segment, const pg_query* query) {
  return guarded([&] {
    REQUIRE(segment && query, "null argument");
    (void)compile_plan(segment->seg, query->filter, query);   // throws PG_ERR_UNSUPPORTED for shapes off the GPU path
  });
}

int32_t pg_query_exec(pg_segment_t segment, const pg_query* query, pg_result_t* out_result) {
  return guarded([&] {
    REQUIRE(segment && query && out_result, "null argument");
    auto r = execute_query(segment->seg, *query);
    auto* h = new pg_result_s();
    h->r = std::move(r);
    *out_result = h;
  });
}

int32_t pg_result_num_groups(pg_result_t result, int32_t* out) {
  return guarded([&] { REQUIRE(result && out, "null argument"); *out = result->r->num_groups; });
}
int32_t pg_result_group_dict_ids(pg_result_t result, int32_t col, int32_t* out, int32_t capacity) {
  return guarded([&] {
    REQUIRE(result, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_dict_ids.size(), "group-by column index out of range");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    REQUIRE(!result->r->raw_group_keys, "group-by column has raw values, not dictIds (pg_result_group_values_long)");
    const auto& v = result->r->group_dict_ids[col];
    if (!v.empty()) memcpy(out, v.data(), v.size() * 4);
  });
}
int32_t pg_result_group_key_type(pg_result_t result, int32_t col, int32_t* out_type) {
  return guarded([&] {
    REQUIRE(result && out_type, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_dict_ids.size(), "group-by column index out of range");
    *out_type = result->r->raw_group_keys ? PG_GROUP_KEY_LONG_VALUES : PG_GROUP_KEY_DICT_IDS;
  });
}
int32_t pg_result_group_values_long(pg_result_t result, int32_t col, int64_t* out_values, int32_t capacity) {
  return guarded([&] {
    REQUIRE(result, "null argument");
    REQUIRE(col == 0 && result->r->raw_group_keys, "group-by column has dictIds, not raw values");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    if (!result->r->group_values.empty()) memcpy(out_values, result->r->group_values.data(), result->r->group_values.size() * 8);
  });
}
static AggResult& agg_of(pg_result_t result, int32_t agg) {
  if (!result) fail(PG_ERR_INVALID_ARGUMENT, "null result");
  if (agg < 0 || agg >= (int32_t)result->r->aggs.size()) fail(PG_ERR_INVALID_ARGUMENT, "aggregation index out of range");
  return result->r->aggs[agg];
}
int32_t pg_result_kind_of(pg_result_t result, int32_t agg, int32_t* out_kind) {
  return guarded([&] { REQUIRE(out_kind, "null argument"); *out_kind = agg_of(result, agg).kind; });
}
int32_t pg_result_doubles(pg_result_t result, int32_t agg, int32_t component, double* out, int32_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(component >= 0 && component < 2, "component out of range");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    if (!a.d[component].empty()) memcpy(out, a.d[component].data(), a.d[component].size() * 8);
  });
}
int32_t pg_result_longs(pg_result_t result, int32_t agg, int32_t component, int64_t* out, int32_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(component >= 0 && component < 2, "component out of range");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    if (!a.l[component].empty()) memcpy(out, a.l[component].data(), a.l[component].size() * 8);
  });
}
int32_t pg_result_set_sizes(pg_result_t result, int32_t agg, int32_t* out_sizes, int32_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_DICTID_SET, "aggregation is not a DISTINCTCOUNT");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    if (!a.set_sizes.empty()) memcpy(out_sizes, a.set_sizes.data(), a.set_sizes.size() * 4);
  });
}
int32_t pg_result_set_dict_ids(pg_result_t result, int32_t agg, int32_t* out_dict_ids, int64_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_DICTID_SET, "aggregation is not a DISTINCTCOUNT");
    REQUIRE(capacity >= (int64_t)a.set_ids.size(), "capacity too small");
    if (!a.set_ids.empty()) memcpy(out_dict_ids, a.set_ids.data(), a.set_ids.size() * 4);
  });
}
int32_t pg_result_hll_registers(pg_result_t result, int32_t agg, uint8_t* out_registers, int64_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_HLL, "aggregation is not a DISTINCTCOUNTHLL");
    REQUIRE(capacity >= (int64_t)a.hll.size(), "capacity too small");
    if (!a.hll.empty()) memcpy(out_registers, a.hll.data(), a.hll.size());
  });
}
int32_t pg_result_stats(pg_result_t result, pg_exec_stats* out_stats) {
  return guarded([&] { REQUIRE(result && out_stats, "null argument"); *out_stats = result->r->stats; });
}
int32_t pg_result_free(pg_result_t result) {
  return guarded([&] { delete result; });
}

}  // extern "C"
