// C ABI entry points of libpinot_gpu.so (include/pinot_gpu.h).  Every exception becomes a status code plus a
// thread-local message (pg_last_error), which the JNI shim rethrows as RuntimeException.
#include "pg_internal.hpp"
#include <atomic>
#include <deque>
#include <memory>
#include <mutex>

namespace pg {
// The current knobs are an immutable record behind an atomic pointer; pg_options_reload publishes a NEW record and leaves the old ones alive
// (a few hundred bytes each, reloads are rare): a query thread that read knobs() a moment ago keeps a valid reference — no lock on the
// query path, no use-after-free of the std::string members.
static std::mutex g_knobs_mu;
static std::deque<std::unique_ptr<Knobs>> g_knobs_all;
static std::atomic<const Knobs*> g_knobs_cur{nullptr};
static void knobs_read(Knobs& k) {
  k = Knobs();
  auto flag = [](const char* name) { return getenv(name) != nullptr; };
  auto num = [](const char* name, int64_t dflt) { const char* e = getenv(name); return e ? (int64_t)atoll(e) : dflt; };
  auto str = [](const char* name) { const char* e = getenv(name); return std::string(e ? e : ""); };
  k.no_oct = flag("PG_NO_OCT"); k.oct_no_affine = flag("PG_OCT_NO_AFFINE"); k.oct_byte_regs = flag("PG_OCT_BYTE_REGS");
  k.oct_any_cardinality = flag("PG_OCT_ANY_CARDINALITY"); k.no_oct_prune = flag("PG_NO_OCT_PRUNE"); k.oct_dword_regs = flag("PG_OCT_DWORD_REGS");
  k.no_p2 = flag("PG_NO_P2"); k.p2_no_fast_a = flag("PG_P2_NO_FAST_A"); k.no_p2_oct = flag("PG_NO_P2_OCT"); k.no_radix = flag("PG_NO_RADIX");
  k.no_radix_aux = flag("PG_NO_RADIX_AUX"); k.no_part = flag("PG_NO_PART"); k.no_radix_packed = flag("PG_NO_RADIX_PACKED");
  k.no_pipe_general = flag("PG_NO_PIPE_GENERAL"); k.no_pipe_wide = flag("PG_NO_PIPE_WIDE"); k.no_pipe_wide_double = flag("PG_NO_PIPE_WIDE_DOUBLE");
  k.mv_no_windows = flag("PG_MV_NO_WINDOWS");
  k.no_specd = flag("PG_NO_SPECD"); k.specw = flag("PG_SPECW"); k.no_mvg = flag("PG_NO_MVG"); k.p2_no_pack = flag("PG_P2_NO_PACK"); k.specd_no_dma = flag("PG_SPECD_NO_DMA"); k.specd_no_affine = flag("PG_SPECD_NO_AFFINE"); k.specd_wgs_per_cu = (int)num("PG_SPECD_WGS_PER_CU", 1);
  k.no_oct_count = flag("PG_NO_OCT_COUNT"); k.no_oct_count_kernel = flag("PG_NO_OCT_COUNT_KERNEL"); k.oct_count_min_docs = num("PG_OCT_COUNT_MIN_DOCS", (int64_t)1 << 22);
  k.oct_min_docs = num("PG_OCT_MIN_DOCS", -1); k.part_min = (int)num("PG_PART_MIN", -1);
  k.force_interpreter = flag("PG_FORCE_INTERPRETER"); k.no_scan_pipe = flag("PG_NO_SCAN_PIPE"); k.no_pipe = flag("PG_NO_PIPE");
  k.no_dense_fused = flag("PG_NO_DENSE_FUSED"); k.no_part_grid_clamp = flag("PG_NO_PART_GRID_CLAMP"); k.no_spin_wait = flag("PG_NO_SPIN_WAIT");
  k.trace_oct = flag("PG_TRACE_OCT"); k.no_tile_split = flag("PG_NO_TILE_SPLIT"); k.no_oct_exec = flag("PG_NO_OCT_EXEC");
  k.no_p2_simple = flag("PG_NO_P2_SIMPLE"); k.no_dense_count = flag("PG_NO_DENSE_COUNT"); k.no_direct_result = flag("PG_NO_DIRECT_RESULT");
  k.trace_host = flag("PG_TRACE_HOST"); k.no_limit_prefix = flag("PG_NO_LIMIT_PREFIX"); k.no_device_trim = flag("PG_NO_DEVICE_TRIM"); k.no_fused_finish = flag("PG_NO_FUSED_FINISH");
  k.scan_wgs_per_cu = (int)num("PG_SCAN_WGS_PER_CU", 1); k.pipe_wgs_per_cu = (int)num("PG_PIPE_WGS_PER_CU", 1); k.wgs_per_cu = (int)num("PG_WGS_PER_CU", 1);
  k.p2_wgs_per_cu = (int)num("PG_P2_WGS_PER_CU", 4); k.dense_count_wgs = (int)num("PG_DENSE_COUNT_WGS", 1); k.tile_split_max = (int)num("PG_TILE_SPLIT_MAX", -1);
  k.hash_first_buckets = (int)num("PG_HASH_FIRST_BUCKETS", -1);
  k.max_inflight = (int)num("PG_MAX_INFLIGHT", 16); k.wave_specialised = flag("PG_WAVE_SPECIALISED"); k.no_wave_specialised = flag("PG_NO_WAVE_SPECIALISED");
  k.wave_specialised_min_permille = (int)num("PG_WAVE_SPECIALISED_MIN_PERMILLE", 150);
  k.exact_stats_max_docs = num("PG_EXACT_STATS_MAX_DOCS", (int64_t)1 << 22);
  k.filter_stats_host = flag("PG_FILTER_STATS_HOST"); k.exact_stats_device_max_docs = num("PG_EXACT_STATS_DEVICE_MAX_DOCS", (int64_t)1 << 27);
  k.limit_prefix_min_docs = std::max<int64_t>(PG_WAVE_DOCS, num("PG_LIMIT_PREFIX_MIN_DOCS", (int64_t)1 << 20));
  k.oct_passes = str("PG_OCT_PASSES"); k.rccl_library = str("PG_RCCL_LIBRARY");
}
static const Knobs* knobs_publish() {   // g_knobs_mu held
  auto k = std::make_unique<Knobs>();
  knobs_read(*k);
  const Knobs* p = k.get();
  g_knobs_all.push_back(std::move(k));
  g_knobs_cur.store(p, std::memory_order_release);
  return p;
}
const Knobs& knobs() {
  const Knobs* k = g_knobs_cur.load(std::memory_order_acquire);
  if (!k) {
    std::lock_guard<std::mutex> g(g_knobs_mu);
    k = g_knobs_cur.load(std::memory_order_acquire);
    if (!k) k = knobs_publish();
  }
  return *k;
}
void knobs_reload() {
  std::lock_guard<std::mutex> g(g_knobs_mu);
  (void)knobs_publish();
}
}  // namespace pg


using namespace pg;

struct pg_segment_s { Segment seg; };
struct pg_result_s { std::unique_ptr<Result> r; };
struct pg_docidset_s { std::unique_ptr<DocIdSet> s; };
struct pg_cancel_s { CancelToken token; };
struct pg_comm_s { Comm* c = nullptr; };

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
  return guarded([&] {
    (void)knobs();   // the PG_* environment is read here, once
    device_init(device_ordinal);
  });
}

int32_t pg_options_reload(void) {
  return guarded([&] { knobs_reload(); });
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

static void segment_create_on(const char* segment_name, int32_t total_docs, int32_t device, pg_segment_t* out_segment) {
  REQUIRE(out_segment, "out_segment is null");
  REQUIRE(total_docs >= 0, "total_docs < 0");
  use_device(device);   // fails loudly without a HIP device / for an ordinal out of range
  auto s = std::make_unique<pg_segment_s>();
  s->seg.name = segment_name ? segment_name : "";
  s->seg.device = device;
  s->seg.total_docs = total_docs;
  s->seg.n_tiles = (int32_t)(((int64_t)total_docs + PG_TILE_DOCS - 1) / PG_TILE_DOCS);
  if (s->seg.n_tiles == 0) s->seg.n_tiles = 1;
  *out_segment = s.release();
}
int32_t pg_segment_create(const char* segment_name, int32_t total_docs, pg_segment_t* out_segment) {
  return guarded([&] { segment_create_on(segment_name, total_docs, default_device(), out_segment); });
}
int32_t pg_segment_create_on_device(const char* segment_name, int32_t total_docs, int32_t device_ordinal, pg_segment_t* out_segment) {
  return guarded([&] { segment_create_on(segment_name, total_docs, device_ordinal, out_segment); });
}
int32_t pg_segment_device(pg_segment_t segment, int32_t* out_device_ordinal) {
  return guarded([&] { REQUIRE(segment && out_device_ordinal, "null argument"); *out_device_ordinal = segment->seg.device; });
}

int32_t pg_segment_add_column(pg_segment_t segment, const pg_column_desc* column) {
  return guarded([&] {
    REQUIRE(segment && column, "null argument");
    use_device(segment->seg.device);
    std::lock_guard<std::mutex> g(segment->seg.mu);
    segment_add_column(segment->seg, *column);
    segment->seg.plan_cache.clear();
  });
}

int32_t pg_segment_add_star_tree(pg_segment_t segment, const pg_star_tree_desc* star_tree) {
  return guarded([&] {
    REQUIRE(segment && star_tree, "null argument");
    use_device(segment->seg.device);
    std::lock_guard<std::mutex> g(segment->seg.mu);
    segment_add_star_tree(segment->seg, *star_tree);
    segment->seg.plan_cache.clear();
  });
}
int32_t pg_segment_set_null_vector(pg_segment_t segment, const char* column, const void* roaring, uint64_t size) {
  return guarded([&] {
    REQUIRE(segment && column, "null argument");
    use_device(segment->seg.device);
    segment_set_null_vector(segment->seg, column, roaring, size);
  });
}
int32_t pg_segment_set_range_index(pg_segment_t segment, const char* column, const void* range_index, uint64_t size) {
  return guarded([&] {
    REQUIRE(segment && column, "null argument");
    use_device(segment->seg.device);
    segment_set_range_index(segment->seg, column, range_index, size);
  });
}
int32_t pg_segment_set_queryable_doc_ids(pg_segment_t segment, const void* roaring, uint64_t size) {
  return guarded([&] {
    REQUIRE(segment, "null argument");
    use_device(segment->seg.device);
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
  return guarded([&] {
    if (segment) use_device(segment->seg.device);
    delete segment;
  });
}

int32_t pg_filter_exec_flags(pg_segment_t segment, const pg_filter_node* filter, int32_t flags, pg_docidset_t* out_docidset) {
  return guarded([&] {
    REQUIRE(segment && out_docidset, "null argument");
    auto s = execute_filter(segment->seg, filter, flags);
    auto* h = new pg_docidset_s();
    h->s = std::move(s);
    *out_docidset = h;
  });
}
int32_t pg_filter_exec(pg_segment_t segment, const pg_filter_node* filter, pg_docidset_t* out_docidset) {
  return pg_filter_exec_flags(segment, filter, 0, out_docidset);
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
    use_device(set->s->device);
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
  return guarded([&] {
    if (set && set->s) use_device(set->s->device);
    delete set;
  });
}

int32_t pg_query_supported(pg_segment_t segment, const pg_query* query) {
  return guarded([&] {
    REQUIRE(segment && query, "null argument");
    use_device(segment->seg.device);
    // compiled under the segment's lock and cached: the pg_query_exec that follows finds the plan (PlanMaker calls supported()
    // then exec() from many worker threads); throws PG_ERR_UNSUPPORTED for shapes off the GPU path
    check_null_handling(segment->seg, *query);
    (void)get_plan(segment->seg, query->filter, query);
  });
}

int32_t pg_query_exec(pg_segment_t segment, const pg_query* query, pg_result_t* out_result) {
  return guarded([&] {
    REQUIRE(segment && query && out_result, "null argument");
    auto r = execute_query(segment->seg, *query, nullptr);
    auto* h = new pg_result_s();
    h->r = std::move(r);
    *out_result = h;
  });
}

int32_t pg_cancel_create(pg_cancel_t* out_cancel) {
  return guarded([&] { REQUIRE(out_cancel, "null argument"); *out_cancel = new pg_cancel_s(); });
}
int32_t pg_cancel_request(pg_cancel_t cancel) {
  return guarded([&] { REQUIRE(cancel, "null argument"); cancel->token.requested.store(1, std::memory_order_release); });
}
int32_t pg_cancel_reset(pg_cancel_t cancel) {
  return guarded([&] { REQUIRE(cancel, "null argument"); cancel->token.requested.store(0, std::memory_order_release); });
}
int32_t pg_cancel_destroy(pg_cancel_t cancel) {
  return guarded([&] { delete cancel; });
}
int32_t pg_query_exec_cancellable(pg_segment_t segment, const pg_query* query, pg_cancel_t cancel, pg_result_t* out_result) {
  return guarded([&] {
    REQUIRE(segment && query && out_result, "null argument");
    auto r = execute_query(segment->seg, *query, cancel ? &cancel->token : nullptr);
    auto* h = new pg_result_s();
    h->r = std::move(r);
    *out_result = h;
  });
}

int32_t pg_result_merge(pg_result_t dst, pg_result_t src) {
  return guarded([&] {
    REQUIRE(dst && src && dst != src, "null or identical results");
    result_merge(*dst->r, *src->r);
  });
}
int32_t pg_result_all_reduce(pg_result_t result, pg_comm_t comm) {
  return guarded([&] {
    REQUIRE(result && comm && comm->c, "null argument");
    result_all_reduce(*result->r, *comm->c);
  });
}
int32_t pg_comm_get_unique_id(void* out_unique_id) {
  return guarded([&] { REQUIRE(out_unique_id, "null argument"); comm_unique_id(out_unique_id); });
}
int32_t pg_comm_init_rank(int32_t device_ordinal, int32_t world_size, int32_t rank, const void* unique_id, pg_comm_t* out_comm) {
  return guarded([&] {
    REQUIRE(unique_id && out_comm, "null argument");
    auto h = std::make_unique<pg_comm_s>();
    h->c = comm_init_rank(device_ordinal, world_size, rank, unique_id);
    *out_comm = h.release();
  });
}
int32_t pg_comm_init_all(int32_t n_devices, const int32_t* device_ordinals, pg_comm_t* out_comms) {
  return guarded([&] {
    REQUIRE(device_ordinals && out_comms && n_devices > 0, "null argument");
    std::vector<Comm*> cs((size_t)n_devices, nullptr);
    comm_init_all(n_devices, device_ordinals, cs.data());
    for (int i = 0; i < n_devices; i++) {
      auto* h = new pg_comm_s();
      h->c = cs[(size_t)i];
      out_comms[i] = h;
    }
  });
}
int32_t pg_comm_world_size(pg_comm_t comm, int32_t* out_world_size) {
  return guarded([&] { REQUIRE(comm && comm->c && out_world_size, "null argument"); *out_world_size = comm_world(*comm->c); });
}
int32_t pg_comm_destroy(pg_comm_t comm) {
  return guarded([&] {
    if (comm) comm_destroy(comm->c);
    delete comm;
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
    REQUIRE(result->r->group_key_type[(size_t)col] == PG_GROUP_KEY_DICT_IDS, "group-by column has raw values, not dictIds (pg_result_group_values_long / _double)");
    const auto& v = result->r->group_dict_ids[col];
    if (!v.empty()) memcpy(out, v.data(), v.size() * 4);
  });
}
int32_t pg_result_group_key_type(pg_result_t result, int32_t col, int32_t* out_type) {
  return guarded([&] {
    REQUIRE(result && out_type, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_key_type.size(), "group-by column index out of range");
    *out_type = result->r->group_key_type[(size_t)col];
  });
}
int32_t pg_result_group_values_long(pg_result_t result, int32_t col, int64_t* out_values, int32_t capacity) {
  return guarded([&] {
    REQUIRE(result, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_key_type.size(), "group-by column index out of range");
    REQUIRE(result->r->group_key_type[(size_t)col] == PG_GROUP_KEY_LONG_VALUES, "group-by column does not have LONG values");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    const auto& v = result->r->group_values[(size_t)col];
    if (!v.empty()) memcpy(out_values, v.data(), v.size() * 8);
  });
}
int32_t pg_result_group_values_double(pg_result_t result, int32_t col, double* out_values, int32_t capacity) {
  return guarded([&] {
    REQUIRE(result, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_key_type.size(), "group-by column index out of range");
    REQUIRE(result->r->group_key_type[(size_t)col] == PG_GROUP_KEY_DOUBLE_VALUES, "group-by column does not have DOUBLE values");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    const auto& v = result->r->group_values[(size_t)col];
    if (!v.empty()) memcpy(out_values, v.data(), v.size() * 8);
  });
}
int32_t pg_result_group_values_bytes_size(pg_result_t result, int32_t col, uint64_t* out_total_bytes) {
  return guarded([&] {
    REQUIRE(result && out_total_bytes, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_key_type.size(), "group-by column index out of range");
    REQUIRE(result->r->group_key_type[(size_t)col] == PG_GROUP_KEY_BYTES_VALUES, "group-by column does not have BYTES values");
    *out_total_bytes = (uint64_t)result->r->group_bytes[(size_t)col].size();
  });
}
int32_t pg_result_group_values_bytes(pg_result_t result, int32_t col, int64_t* out_offsets, int32_t offsets_capacity, uint8_t* out_bytes,
                                     uint64_t bytes_capacity) {
  return guarded([&] {
    REQUIRE(result && out_offsets, "null argument");
    REQUIRE(col >= 0 && col < (int32_t)result->r->group_key_type.size(), "group-by column index out of range");
    REQUIRE(result->r->group_key_type[(size_t)col] == PG_GROUP_KEY_BYTES_VALUES, "group-by column does not have BYTES values");
    const auto& b = result->r->group_bytes[(size_t)col];
    const auto& o = result->r->group_bytes_off[(size_t)col];
    REQUIRE(offsets_capacity >= result->r->num_groups + 1 && bytes_capacity >= b.size(), "capacity too small");
    REQUIRE(out_bytes || b.empty(), "null argument");
    if (o.empty()) out_offsets[0] = 0;   // no group
    else memcpy(out_offsets, o.data(), o.size() * 8);
    if (!b.empty()) memcpy(out_bytes, b.data(), b.size());
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
    else if (result->r->num_groups > 0) memset(out, 0, (size_t)result->r->num_groups * 8);   // a component the result kind does not have
  });
}
int32_t pg_result_longs(pg_result_t result, int32_t agg, int32_t component, int64_t* out, int32_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(component >= 0 && component < 2, "component out of range");
    REQUIRE(capacity >= result->r->num_groups, "capacity too small");
    if (!a.l[component].empty()) memcpy(out, a.l[component].data(), a.l[component].size() * 8);
    else if (result->r->num_groups > 0) memset(out, 0, (size_t)result->r->num_groups * 8);
  });
}
int32_t pg_result_set_sizes(pg_result_t result, int32_t agg, int32_t* out_sizes, int32_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_DICTID_SET || a.kind == PG_RESULT_VALUE_SET, "aggregation is not a DISTINCTCOUNT");
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
int32_t pg_result_set_values_long(pg_result_t result, int32_t agg, int64_t* out_values, int64_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_VALUE_SET && a.set_value_kind <= 1, "aggregation is not a DISTINCTCOUNT over a raw INT / LONG column");
    REQUIRE(capacity >= (int64_t)a.l[0].size(), "capacity too small");
    if (!a.l[0].empty()) memcpy(out_values, a.l[0].data(), a.l[0].size() * 8);
  });
}
int32_t pg_result_set_values_double(pg_result_t result, int32_t agg, double* out_values, int64_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_VALUE_SET && a.set_value_kind >= 2, "aggregation is not a DISTINCTCOUNT over a raw FLOAT / DOUBLE column");
    REQUIRE(capacity >= (int64_t)a.d[0].size(), "capacity too small");
    if (!a.d[0].empty()) memcpy(out_values, a.d[0].data(), a.d[0].size() * 8);
  });
}
int32_t pg_result_hll_registers(pg_result_t result, int32_t agg, uint8_t* out_registers, int64_t capacity) {
  return guarded([&] {
    AggResult& a = agg_of(result, agg);
    REQUIRE(a.kind == PG_RESULT_HLL, "aggregation is not a DISTINCTCOUNTHLL");
    if (a.hll_regs) {   // registers still in the page-locked block the device wrote: gather the groups (runs of ids in one copy)
      const size_t ng = a.hll_gids.size(), stride = (size_t)a.hll_stride;
      REQUIRE(capacity >= (int64_t)(ng * stride), "capacity too small");
      for (size_t i = 0; i < ng;) {
        size_t j = i + 1;
        while (j < ng && a.hll_gids[j] == a.hll_gids[j - 1] + 1) j++;
        memcpy(out_registers + i * stride, a.hll_regs + (size_t)a.hll_gids[i] * stride, (j - i) * stride);
        i = j;
      }
      return;
    }
    REQUIRE(capacity >= (int64_t)a.hll.size(), "capacity too small");
    if (!a.hll.empty()) memcpy(out_registers, a.hll.data(), a.hll.size());
  });
}
int32_t pg_result_data_table_v4(pg_result_t result, uint8_t* out, int64_t capacity, int64_t* out_size) {
  return guarded([&] {
    REQUIRE(result && out_size, "null argument");
    const std::vector<uint8_t> b = result_data_table_v4(*result->r);
    *out_size = (int64_t)b.size();
    if (!out) return;                              // size query
    REQUIRE(capacity >= (int64_t)b.size(), "capacity too small");
    memcpy(out, b.data(), b.size());
  });
}
int32_t pg_result_agg_nulls(pg_result_t result, int32_t agg, uint8_t* out, int32_t capacity) {
  return guarded([&] {
    REQUIRE(result && out, "null argument");
    const Result& r = *result->r;
    REQUIRE(agg >= 0 && agg < (int32_t)r.aggs.size() && capacity >= r.num_groups, "bad aggregation / capacity");
    if ((size_t)agg < r.agg_nulls.size() && !r.agg_nulls[(size_t)agg].empty()) memcpy(out, r.agg_nulls[(size_t)agg].data(), (size_t)r.num_groups);
    else memset(out, 0, (size_t)r.num_groups);
  });
}
int32_t pg_result_group_key_nulls(pg_result_t result, int32_t col, uint8_t* out, int32_t capacity) {
  return guarded([&] {
    REQUIRE(result && out, "null argument");
    const Result& r = *result->r;
    REQUIRE(col >= 0 && col < (int32_t)r.group_key_type.size() && capacity >= r.num_groups, "bad column / capacity");
    if ((size_t)col < r.key_nulls.size() && !r.key_nulls[(size_t)col].empty()) memcpy(out, r.key_nulls[(size_t)col].data(), (size_t)r.num_groups);
    else memset(out, 0, (size_t)r.num_groups);
  });
}
int32_t pg_result_stats(pg_result_t result, pg_exec_stats* out_stats) {
  return guarded([&] { REQUIRE(result && out_stats, "null argument"); *out_stats = result->r->stats; });
}
int32_t pg_result_free(pg_result_t result) {
  return guarded([&] {
    if (result && result->r && result->r->dev) use_device(result->r->dev->device);
    delete result;
  });
}

}  // extern "C"
