// Cross-GPU merge of dense group tables over RCCL (pg_comm_*, pg_result_all_reduce): the one exchange step of the path
//  — what GroupByCombineOperator.mergeResults does across worker threads
// (pinot-core/.../operator/combine/GroupByCombineOperator.java:102-165,191-222), for segments pinned one per GPU that share
// their key space.  Payloads are KBs..MBs (groups x accumulators x 8 B, + 2^log2m register bytes per group), i.e. the
// collective is latency-bound: every accumulator row goes into ONE grouped RCCL launch (ncclGroupStart/End), in place on
// the table the query kernels left in HBM; no host round trip, no pickling.
//
// librccl is opened on first use (dlopen by soname: inside a process that already loaded RCCL — PyTorch's bundled copy under
// bench.py — this resolves to that same copy, so there is one RCCL and one HIP runtime per process).  Single-GPU callers never
// touch it and the library has no link-time dependency on RCCL.
#include <dlfcn.h>

#include <atomic>

#include "pg_internal.hpp"

namespace pg {

// ---- the slice of rccl.h this file uses (stable NCCL 2.x ABI) -------------------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // 0 = ncclSuccess
enum { kNcclInt8 = 0, kNcclUint8 = 1, kNcclInt32 = 2, kNcclInt64 = 4 };   // ncclDataType_t
enum { kNcclSum = 0, kNcclProd = 1, kNcclMax = 2, kNcclMin = 3 };          // ncclRedOp_t

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static Rccl& rccl() {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_rccl.handle) return g_rccl;
  // PG_RCCL_LIBRARY names the collective library to bind instead of the system's RCCL: any shared object with the NCCL 2.x C ABI
  // (a site's own RCCL build; tests/fake_rccl — N ranks of one process on ONE device — on one-GPU boxes).
  const char* override_path = knobs().rccl_library.empty() ? nullptr : knobs().rccl_library.c_str();
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  if (override_path && override_path[0]) {
    h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) fail(PG_ERR_DEVICE, "cannot load PG_RCCL_LIBRARY=%s (%s)", override_path, dlerror());
  }
  for (const char* n : names)
    if (!h && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) fail(PG_ERR_DEVICE, "cannot load librccl (%s): the cross-GPU merge needs RCCL", dlerror());
  Rccl r;
  r.handle = h;
  auto sym = [&](const char* name) {
    void* p = dlsym(h, name);
    if (!p) fail(PG_ERR_DEVICE, "librccl lacks %s", name);
    return p;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  g_rccl = r;
  return g_rccl;
}
#define PG_NCCL(expr)                                                                                         \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != 0) ::pg::fail(PG_ERR_DEVICE, "RCCL error %d (%s) at %s:%d: %s", _r, rccl().GetErrorString(_r), __FILE__, __LINE__, #expr); \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int device = 0;
  int world = 1;
  int rank = 0;
  DeviceBuffer scratch;   // merged image of a table (signature probe, accumulators, states) + gathered dictId sets
  // the probe, out and back: ncclMax over {sig, -sig, largest per-doc |value| of an int64 SUM, digit sums present, a rank refuses
  // (IEEE-double SUM)}; ncclSum over {full-scan entries, total docs}
  int64_t probe[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [7 .. 9]: {key-space-free signature, its negation, a rank cannot re-key} (ncclMax)
  // One pg_result_all_reduce at a time per communicator: `probe` and `scratch` are per-communicator state, and two merges entering at
  // once would enqueue their collectives in different orders on different ranks (a hang in RCCL).  The CALLER serialises whole merges
  // across all ranks (GpuGroupByCombineOperator's COLLECTIVE lock); a per-communicator lock here could only deadlock two of them
  // against each other.  What the library does is refuse loudly instead of corrupting silently.
  std::atomic<bool> busy{false};
};
struct CommBusy {
  Comm& c;
  explicit CommBusy(Comm& comm) : c(comm) {
    if (c.busy.exchange(true, std::memory_order_acq_rel))
      fail(PG_ERR_INVALID_ARGUMENT, "pg_result_all_reduce: the communicator of device %d is inside another merge (serialise the merges of one communicator set)", c.device);
  }
  ~CommBusy() { c.busy.store(false, std::memory_order_release); }
};

void comm_unique_id(void* out128) {
  ncclUniqueId id;
  memset(&id, 0, sizeof(id));
  PG_NCCL(rccl().GetUniqueId(&id));
  memcpy(out128, id.internal, sizeof(id.internal));
}

Comm* comm_init_rank(int device, int world, int rank, const void* id128) {
  if (world < 1 || rank < 0 || rank >= world) fail(PG_ERR_INVALID_ARGUMENT, "pg_comm_init_rank: rank %d of %d", rank, world);
  use_device(device);
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  auto c = std::make_unique<Comm>();
  c->device = device;
  c->world = world;
  c->rank = rank;
  PG_NCCL(rccl().CommInitRank(&c->comm, world, id, rank));
  return c.release();
}

void comm_init_all(int n, const int32_t* devices, Comm** out) {
  if (n < 1 || n > 64) fail(PG_ERR_INVALID_ARGUMENT, "pg_comm_init_all: %d devices", n);
  std::vector<int> devs(devices, devices + n);
  for (int d : devs) use_device(d);   // validates the ordinals, initialises each device
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  PG_NCCL(rccl().CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; i++) {
    auto c = std::make_unique<Comm>();
    c->comm = comms[(size_t)i];
    c->device = devs[(size_t)i];
    c->world = n;
    c->rank = i;
    out[i] = c.release();
  }
}

int comm_world(const Comm& c) { return c.world; }

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)rccl().CommDestroy(c->comm);
  }
  delete c;
}

// ---- value-keyed merge: segments with DIFFERENT group-by dictionaries (every real Pinot table: one dictionary per segment and column) -----
// GroupByCombineOperator merges by VALUE for exactly that reason (GroupByCombineOperator.java:135-144 keys its IndexedTable by the groups'
// values, DictionaryBasedGroupKeyGenerator.java:578-606 turns dictIds back into values before the merge).  Here: the ranks all-gather the
// group-by columns' dictionaries (KBs), every rank builds the SAME sorted union per column, re-keys its dense table into the union's key space
// on the device (pg_remap_table_kernel: one scatter) and the ordinary grouped all-reduce runs over the re-keyed tables.  The merged result
// presents its groups by value (PG_GROUP_KEY_*_VALUES), as raw group-by columns do.
//
// The signature WITHOUT the key space (cardinalities, dictionary contents, table size): equal on all ranks <=> the tables differ at most in
// their group-by dictionaries.
static int64_t layout_signature(const DeviceTable& T) {
  const PgQueryPlan& D = T.plan->dev;
  uint64_t h = 1469598103934665603ULL;
  auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ULL; };
  mix(0x756e696f6eULL); mix((uint64_t)D.n_ops); mix((uint64_t)T.n_group_by); mix((uint64_t)T.n_aggregations);
  for (int o = 0; o < D.n_ops; o++) { mix((uint64_t)D.ops[o].fn); mix((uint64_t)D.ops[o].is_float); mix((uint64_t)(uint32_t)D.ops[o].limb); if (D.ops[o].src >= 0) mix((uint64_t)(uint32_t)D.srcs[D.ops[o].src].fx_q); }
  for (size_t a = 0; a < T.plan->aggs.size(); a++) { mix((uint64_t)T.plan->aggs[a].function); mix((uint64_t)(uint32_t)T.plan->aggs[a].op_a); }
  for (const Column* c : T.plan->group_cols) mix(c ? (uint64_t)c->data_type : 99u);
  return (int64_t)(h >> 2);
}
// Can THIS rank's table be re-keyed: a dense key space over dictionary-encoded group-by columns of a type whose values order as bytes / numbers,
// no auxiliary state indexed by group (HyperLogLog registers could follow, dictId sets of ANOTHER column's dictionary could not).
static bool union_eligible(const DeviceTable& T) {
  const CompiledPlan& P = *T.plan;
  const PgQueryPlan& D = P.dev;
  if (T.keys || T.n_group_by < 1 || T.n_group_by > PG_MAX_GROUP_COLS || D.n_aux != 0 || P.raw_group || D.mv) return false;
  if (D.agg_mode == PG_AGG_RADIX_HASH || (int64_t)D.n_ops * (int64_t)std::max(D.n_groups, 1) != T.n_out) return false;
  if ((int)P.group_cols.size() != T.n_group_by || (int)P.group_cards.size() != T.n_group_by) return false;
  for (int j = 0; j < T.n_group_by; j++) {
    const Column* c = P.group_cols[(size_t)j];
    if (!c || !c->has_dictionary || c->is_mv || ((size_t)j < P.group_vdict.size() && P.group_vdict[(size_t)j])) return false;
    if (c->data_type > PG_TYPE_BYTES || c->dict_bytes_per_value <= 0 || c->dict_host.size() != (size_t)c->cardinality * (size_t)c->dict_bytes_per_value) return false;
    if (c->cardinality != P.group_cards[(size_t)j]) return false;
  }
  return true;
}
static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
static inline uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
// one dictionary value as the key the union sorts by: numbers as order-preserving 64-bit keys (pg_vdict.hip's), strings as their bytes
static uint64_t union_number_key(const uint8_t* p, int32_t data_type) {
  if (data_type == PG_TYPE_INT) return (uint64_t)(int64_t)(int32_t)be32(p) ^ (1ULL << 63);
  if (data_type == PG_TYPE_LONG) return be64(p) ^ (1ULL << 63);
  if (data_type == PG_TYPE_FLOAT) { const uint32_t f = be32(p); return (uint64_t)((f >> 31) ? ~f : (f ^ 0x80000000u)); }
  const uint64_t d = be64(p);
  return (d >> 63) ? ~d : (d ^ (1ULL << 63));
}
// Steps 1-5 of the value-keyed merge; leaves T re-keyed (T.keys set, T.table / T.n_out in the union's key space).  Every decision is taken
// from gathered data, so every rank takes it alike.
static void union_key_space(DeviceTable& T, Comm& c, Rccl& R, hipStream_t stream) {
  const CompiledPlan& P = *T.plan;
  const PgQueryPlan& D = P.dev;
  const int nc = T.n_group_by, W = c.world;
  // ---- 1. cardinalities and value widths of every rank's dictionaries -----------------------------------------------------------------
  const size_t meta_bytes = 64;   // 8 x {cardinality, bytes per value}
  std::vector<int32_t> meta(16, 0);
  for (int j = 0; j < nc; j++) { meta[(size_t)(2 * j)] = P.group_cols[(size_t)j]->cardinality; meta[(size_t)(2 * j + 1)] = P.group_cols[(size_t)j]->dict_bytes_per_value; }
  if (c.scratch.size < meta_bytes * (size_t)(W + 1)) c.scratch.alloc(meta_bytes * (size_t)(W + 1) + 4096);
  uint8_t* S = c.scratch.as<uint8_t>();
  PG_HIP(hipMemcpyAsync(S, meta.data(), meta_bytes, hipMemcpyHostToDevice, stream));
  PG_NCCL(R.AllGather(S, S + meta_bytes, meta_bytes, kNcclUint8, c.comm, stream));
  std::vector<int32_t> all_meta((size_t)W * 16);
  PG_HIP(hipMemcpyAsync(all_meta.data(), S + meta_bytes, meta_bytes * (size_t)W, hipMemcpyDeviceToHost, stream));
  PG_HIP(hipStreamSynchronize(stream));
  auto card_of = [&](int r, int j) { return (int64_t)all_meta[(size_t)r * 16 + (size_t)(2 * j)]; };
  auto width_of = [&](int r, int j) { return (int64_t)all_meta[(size_t)r * 16 + (size_t)(2 * j + 1)]; };
  // ---- 2. the dictionaries themselves, padded to the largest of each column ----------------------------------------------------------------
  std::vector<size_t> col_off((size_t)nc + 1, 0);
  for (int j = 0; j < nc; j++) {
    int64_t mx = 0;
    for (int r = 0; r < W; r++) {
      if (card_of(r, j) <= 0 || width_of(r, j) <= 0) fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: rank %d holds no dictionary for group-by column %d: merge on the host by values", r, j);
      mx = std::max(mx, card_of(r, j) * width_of(r, j));
    }
    col_off[(size_t)j + 1] = col_off[(size_t)j] + (((size_t)mx + 15) & ~(size_t)15);
  }
  const size_t per_rank = col_off[(size_t)nc];
  if (per_rank * (size_t)(W + 1) > ((size_t)1 << 30)) fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: %zu bytes of group-by dictionaries per rank: merge on the host by values", per_rank);
  if (c.scratch.size < per_rank * (size_t)(W + 1) + 64) c.scratch.alloc(per_rank * (size_t)(W + 1) + 4096);
  S = c.scratch.as<uint8_t>();
  std::vector<uint8_t> mine(per_rank, 0);
  for (int j = 0; j < nc; j++) memcpy(mine.data() + col_off[(size_t)j], P.group_cols[(size_t)j]->dict_host.data(), P.group_cols[(size_t)j]->dict_host.size());
  PG_HIP(hipMemcpyAsync(S, mine.data(), per_rank, hipMemcpyHostToDevice, stream));
  PG_NCCL(R.AllGather(S, S + per_rank, per_rank, kNcclUint8, c.comm, stream));
  std::vector<uint8_t> all(per_rank * (size_t)W);
  PG_HIP(hipMemcpyAsync(all.data(), S + per_rank, per_rank * (size_t)W, hipMemcpyDeviceToHost, stream));
  PG_HIP(hipStreamSynchronize(stream));
  // ---- 3. the union per column (identical on every rank) and this rank's dictId -> union id maps ----------------------------------------------
  auto keys = std::make_unique<UnionKeys>();
  std::vector<int32_t> maps;
  std::vector<int64_t> geo((size_t)nc * 3, 0);
  int64_t G2 = 1;
  for (int j = 0; j < nc; j++) {
    const Column* mc = P.group_cols[(size_t)j];
    const int32_t dt = mc->data_type;
    auto u = std::make_unique<Column>();
    u->name = mc->name;
    u->data_type = dt;
    const size_t map_at = maps.size();
    maps.resize(map_at + (size_t)mc->cardinality);
    if (dt <= PG_TYPE_DOUBLE) {
      std::vector<uint64_t> ks;
      for (int r = 0; r < W; r++) {
        if (width_of(r, j) != (dt == PG_TYPE_INT || dt == PG_TYPE_FLOAT ? 4 : 8)) fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: rank %d stores column %s %lld bytes wide", r, mc->name.c_str(), (long long)width_of(r, j));
        const uint8_t* d = all.data() + (size_t)r * per_rank + col_off[(size_t)j];
        for (int64_t i = 0; i < card_of(r, j); i++) ks.push_back(union_number_key(d + (size_t)i * (size_t)width_of(r, j), dt));
      }
      std::sort(ks.begin(), ks.end());
      ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
      u->vdict_kind = dt == PG_TYPE_INT ? 0 : (dt == PG_TYPE_LONG ? 1 : (dt == PG_TYPE_FLOAT ? 2 : 3));
      for (int32_t i = 0; i < mc->cardinality; i++) {
        const uint64_t k = union_number_key(mc->dict_host.data() + (size_t)i * (size_t)mc->dict_bytes_per_value, dt);
        maps[map_at + (size_t)i] = (int32_t)(std::lower_bound(ks.begin(), ks.end(), k) - ks.begin());
      }
      u->cardinality = (int32_t)ks.size();
      u->vdict_keys = std::move(ks);
    } else {   // STRING (entries padded with zero bytes: BaseImmutableDictionary) / BYTES: the values' bytes, ordered as unsigned bytes
      std::vector<std::string> vs;
      auto value_at = [&](const uint8_t* d, int64_t i, int64_t w) {
        size_t n = (size_t)w;
        if (dt == PG_TYPE_STRING) while (n > 0 && d[(size_t)i * (size_t)w + n - 1] == 0) n--;
        return std::string(reinterpret_cast<const char*>(d + (size_t)i * (size_t)w), n);
      };
      for (int r = 0; r < W; r++) {
        const uint8_t* d = all.data() + (size_t)r * per_rank + col_off[(size_t)j];
        for (int64_t i = 0; i < card_of(r, j); i++) vs.push_back(value_at(d, i, width_of(r, j)));
      }
      std::sort(vs.begin(), vs.end());
      vs.erase(std::unique(vs.begin(), vs.end()), vs.end());
      u->vdict_kind = 4;
      u->vdict_bytes_off.assign(1, 0);
      for (const std::string& v : vs) { u->vdict_bytes.insert(u->vdict_bytes.end(), v.begin(), v.end()); u->vdict_bytes_off.push_back((int64_t)u->vdict_bytes.size()); }
      for (int32_t i = 0; i < mc->cardinality; i++)
        maps[map_at + (size_t)i] = (int32_t)(std::lower_bound(vs.begin(), vs.end(), value_at(mc->dict_host.data(), i, mc->dict_bytes_per_value)) - vs.begin());
      u->cardinality = (int32_t)vs.size();
    }
    geo[(size_t)(3 * j)] = mc->cardinality;
    geo[(size_t)(3 * j + 1)] = G2;
    geo[(size_t)(3 * j + 2)] = (int64_t)map_at;
    keys->cards.push_back(u->cardinality);
    keys->mults.push_back(G2);
    if (G2 > ((int64_t)1 << 40) / std::max(u->cardinality, 1)) G2 = (int64_t)1 << 40; else G2 *= u->cardinality;
    keys->dicts.push_back(std::move(u));
  }
  // ---- 4. the union's dense table must stay a table ------------------------------------------------------------------------------------
  if (G2 > ((int64_t)1 << 26) || (int64_t)D.n_ops * G2 > ((int64_t)1 << 27))
    fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: the union of the ranks' dictionaries spans %lld keys x %d accumulators: merge on the host by values", (long long)G2, D.n_ops);
  keys->n_groups = G2;
  // ---- 5. re-key this rank's table on the device ------------------------------------------------------------------------------------------
  const int64_t G = std::max(D.n_groups, 1), n_out2 = (int64_t)D.n_ops * G2;
  DeviceBuffer fresh;
  fresh.alloc(((size_t)n_out2 + PG_MAX_STATS + 2) * 8 + 2 * 8);
  DeviceBuffer dmaps = upload_vector(maps), dgeo = upload_vector(geo);
  remap_table_on_stream(T.table.as<int64_t>(), fresh.as<int64_t>(), G, G2, D.n_ops, nc, dmaps.as<int32_t>(), dgeo.as<int64_t>(), P.ops_dev.as<PgAccOp>(), stream);
  PG_HIP(hipMemcpyAsync(fresh.as<int64_t>() + n_out2, T.table.as<int64_t>() + T.n_out, (size_t)(PG_MAX_STATS + 2) * 8, hipMemcpyDeviceToDevice, stream));
  PG_HIP(hipStreamSynchronize(stream));   // (the maps' buffers die with this scope)
  T.table = std::move(fresh);
  T.n_out = n_out2;
  T.keys = std::move(keys);
}

// Every rank calls this with its own result of the same query.  Two launches:
//   1. the probe — its shape does not depend on the table, so it is safe whatever the ranks hold: ncclMax on {sig, -sig} (the layout's
//      signature, dictionary contents included), on the overflow guards of the summed accumulators {largest per-doc |value|, digit sums}
//      and on a "some rank must refuse" flag; ncclSum on {full-scan entries, total docs}; one stream synchronisation, then the host
//      checks that every rank holds the same layout and that the SUM rows cannot overflow for the merged doc count.  EVERY refusal
//      (PG_ERR_UNSUPPORTED -> merge on the host by values) is decided HERE, after the probe, from REDUCED values only — so every rank
//      decides alike and none enters the table launch alone.  (Round 3 had two rank-local decisions in this function — the IEEE-double
//      SUM refusal in front of the probe and the overflow bound evaluated on the rank's own value range — either of which leaves the
//      other ranks waiting in a collective: ADVICE r3.)
//      Earlier in round 3 the probe rode inside the data launch to save a synchronisation.  Reading the code again: ranks that disagree
//      on the layout would then enqueue collectives of different counts and sizes inside one group, which NCCL / RCCL leaves undefined
//      (in practice a hang) — the very case the probe exists for.
//   2. ONE grouped RCCL launch over the table, out of place into the communicator's scratch, then committed on the stream:
//   row o of the accumulator table   ncclSum (COUNT, SUM limbs) / ncclMin / ncclMax on int64 (float MIN / MAX are order-preserving
//                                    int64 keys, float SUMs are fixed-point int64 limbs: every merge is exact and order-free)
//   statistics counters + tail       ncclSum on int64
//   HyperLogLog registers            ncclMax on uint8
//   dictId sets                      all-gather, then OR (RCCL has no bitwise reduction)
//      followed by the copy of the merged image over the table the query left in HBM and the reassembly's synchronisation.
void result_all_reduce(Result& r, Comm& c) {
  if (!r.dev) fail(PG_ERR_INVALID_ARGUMENT, "pg_result_all_reduce needs a result executed with PG_QUERY_FLAG_KEEP_DEVICE_TABLE");
  DeviceTable& T = *r.dev;
  if (T.device != c.device) fail(PG_ERR_INVALID_ARGUMENT, "pg_result_all_reduce: result on device %d, communicator on device %d", T.device, c.device);
  CommBusy in_use(c);
  use_device(T.device);
  hipStream_t stream = thread_stream(T.device);
  Rccl& R = rccl();
  const PgQueryPlan& D = T.plan->dev;
  int64_t refuse_local = 0;   // decided rank-locally, acted upon only after the probe (on the reduced flag)
  for (int o = 0; o < D.n_ops; o++)
    if (D.ops[o].fn == PG_ACC_SUM && D.ops[o].is_float == 1) refuse_local = 1;
  if (T.keys) refuse_local = 1;   // already re-keyed by an earlier merge: further merges go by values on the host
  device_table_tail_store(T, stream);
  int64_t G = std::max(D.n_groups, 1);
  size_t n_table = (size_t)T.n_out + PG_MAX_STATS + 2;   // accumulators, statistics counters, {full-scan entries, total docs}
  size_t set_bytes = 0;
  for (int x = 0; x < D.n_aux; x++) if (D.aux[x].kind == PG_AUX_DICT_SET) set_bytes += T.plan->aux_bytes[(size_t)x];
  // scratch: [probe 7 x int64 | pad to 64][table image][aux image][gathered sets x world]
  const size_t off_table = 128;
  size_t off_aux = off_table + n_table * 8, off_gather = (off_aux + T.aux_total + 63) & ~(size_t)63;
  size_t need = off_gather + set_bytes * (size_t)c.world + 64;
  if (c.scratch.size < need) c.scratch.alloc(need + need / 4);
  uint8_t* S = c.scratch.as<uint8_t>();
  int64_t* table = T.table.as<int64_t>();
  int64_t* image = reinterpret_cast<int64_t*>(S + off_table);
  // ---- 1. the probe ---------------------------------------------------------------------------------------------------------------
  const int64_t sig = table_signature(T);
  c.probe[0] = sig; c.probe[1] = -sig;
  c.probe[2] = (int64_t)std::min<uint64_t>(T.sum_max_abs, (uint64_t)INT64_MAX);
  c.probe[3] = T.has_digit_sums ? 1 : 0;
  c.probe[4] = refuse_local;
  const int64_t lsig = layout_signature(T);
  c.probe[7] = lsig; c.probe[8] = -lsig;
  c.probe[9] = union_eligible(T) ? 0 : 1;
  PG_HIP(hipMemcpyAsync(S, c.probe, 40, hipMemcpyHostToDevice, stream));
  PG_HIP(hipMemcpyAsync(S + 40, table + T.n_out + PG_MAX_STATS, 16, hipMemcpyDeviceToDevice, stream));   // {full-scan entries, total docs}
  PG_HIP(hipMemcpyAsync(S + 56, c.probe + 7, 24, hipMemcpyHostToDevice, stream));
  PG_NCCL(R.GroupStart());
  PG_NCCL(R.AllReduce(S, S, 5, kNcclInt64, kNcclMax, c.comm, stream));
  PG_NCCL(R.AllReduce(S + 40, S + 40, 2, kNcclInt64, kNcclSum, c.comm, stream));
  PG_NCCL(R.AllReduce(S + 56, S + 56, 3, kNcclInt64, kNcclMax, c.comm, stream));
  PG_NCCL(R.GroupEnd());
  PG_HIP(hipMemcpyAsync(c.probe, S, 80, hipMemcpyDeviceToHost, stream));
  PG_HIP(hipStreamSynchronize(stream));
  // from here on every value is the same on every rank: all of them refuse, all of them re-key, or none does
  const bool same_layout = c.probe[0] == sig && c.probe[1] == -sig;
  const bool same_but_keys = c.probe[7] == lsig && c.probe[8] == -lsig && c.probe[9] == 0;
  if (!same_layout && !same_but_keys)
    fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: the ranks' results do not share their table layout (different aggregations, fixed-point scale, or a "
                             "key space that does not re-key by value: hashed / raw / multi-value keys, distinct-count states): merge on the host by values");
  if (c.probe[4])
    fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: a floating-point SUM accumulated in double (a column holding NaN / Inf) does not merge exactly");
  check_merge_bounds((uint64_t)c.probe[2], c.probe[3] != 0, c.probe[6]);
  T.sum_max_abs = (uint64_t)c.probe[2];   // the merged table's bound, for later merges
  T.has_digit_sums = c.probe[3] != 0;
  if (!same_layout) {
    // different group-by dictionaries: into the union's key space first (collective: two all-gathers), then the launch below as ever
    union_key_space(T, c, R, stream);
    G = T.keys->n_groups;
    n_table = (size_t)T.n_out + PG_MAX_STATS + 2;
    off_aux = off_table + n_table * 8;
    off_gather = (off_aux + T.aux_total + 63) & ~(size_t)63;
    need = off_gather + 64;
    if (c.scratch.size < need) c.scratch.alloc(need + need / 4);
    S = c.scratch.as<uint8_t>();
    table = T.table.as<int64_t>();
    image = reinterpret_cast<int64_t*>(S + off_table);
  }
  // ---- 2. the table (every rank now known to hold the same layout) ------------------------------------------------------------------
  PG_NCCL(R.GroupStart());
  for (int o = 0; o < D.n_ops && T.n_out > 0; o++) {
    const int red = D.ops[o].fn == PG_ACC_MIN ? kNcclMin : (D.ops[o].fn == PG_ACC_MAX ? kNcclMax : kNcclSum);
    PG_NCCL(R.AllReduce(table + (int64_t)o * G, image + (int64_t)o * G, (size_t)G, kNcclInt64, red, c.comm, stream));
  }
  PG_NCCL(R.AllReduce(table + T.n_out, image + T.n_out, PG_MAX_STATS + 2, kNcclInt64, kNcclSum, c.comm, stream));
  {
    size_t off = 0, goff = 0;
    for (int x = 0; x < D.n_aux; x++) {
      const size_t bytes = T.plan->aux_bytes[(size_t)x];
      uint8_t* region = T.aux.as<uint8_t>() + off;
      if (D.aux[x].kind == PG_AUX_DICT_SET) {
        PG_NCCL(R.AllGather(region, S + off_gather + goff, bytes, kNcclUint8, c.comm, stream));
        goff += bytes * (size_t)c.world;
      } else {
        PG_NCCL(R.AllReduce(region, S + off_aux + off, bytes, kNcclUint8, kNcclMax, c.comm, stream));
      }
      off += bytes;
    }
  }
  PG_NCCL(R.GroupEnd());
  {   // dictId sets: OR of the gathered copies into the aux image
    size_t off = 0, goff = 0;
    for (int x = 0; x < D.n_aux; x++) {
      const size_t bytes = T.plan->aux_bytes[(size_t)x];
      if (D.aux[x].kind == PG_AUX_DICT_SET) {
        PG_HIP(hipMemsetAsync(S + off_aux + off, 0, bytes, stream));
        merge_sets_on_stream(reinterpret_cast<uint32_t*>(S + off_aux + off), reinterpret_cast<const uint32_t*>(S + off_gather + goff), (int64_t)(bytes / 4),
                             c.world, stream);
        goff += bytes * (size_t)c.world;
      }
      off += bytes;
    }
  }
  PG_HIP(hipMemcpyAsync(table, image, n_table * 8, hipMemcpyDeviceToDevice, stream));
  if (T.aux_total) PG_HIP(hipMemcpyAsync(T.aux.ptr, S + off_aux, T.aux_total, hipMemcpyDeviceToDevice, stream));
  result_reassemble(r);   // copies the merged table back (its one synchronisation) and rebuilds the groups
}

}  // namespace pg
