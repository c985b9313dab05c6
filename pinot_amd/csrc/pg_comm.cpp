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
  int64_t probe[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
  device_table_tail_store(T, stream);
  const int64_t G = std::max(D.n_groups, 1);
  const size_t n_table = (size_t)T.n_out + PG_MAX_STATS + 2;   // accumulators, statistics counters, {full-scan entries, total docs}
  size_t set_bytes = 0;
  for (int x = 0; x < D.n_aux; x++) if (D.aux[x].kind == PG_AUX_DICT_SET) set_bytes += T.plan->aux_bytes[(size_t)x];
  // scratch: [probe 7 x int64 | pad to 64][table image][aux image][gathered sets x world]
  const size_t off_table = 64, off_aux = off_table + n_table * 8, off_gather = (off_aux + T.aux_total + 63) & ~(size_t)63;
  const size_t need = off_gather + set_bytes * (size_t)c.world + 64;
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
  PG_HIP(hipMemcpyAsync(S, c.probe, 40, hipMemcpyHostToDevice, stream));
  PG_HIP(hipMemcpyAsync(S + 40, table + T.n_out + PG_MAX_STATS, 16, hipMemcpyDeviceToDevice, stream));   // {full-scan entries, total docs}
  PG_NCCL(R.GroupStart());
  PG_NCCL(R.AllReduce(S, S, 5, kNcclInt64, kNcclMax, c.comm, stream));
  PG_NCCL(R.AllReduce(S + 40, S + 40, 2, kNcclInt64, kNcclSum, c.comm, stream));
  PG_NCCL(R.GroupEnd());
  PG_HIP(hipMemcpyAsync(c.probe, S, 56, hipMemcpyDeviceToHost, stream));
  PG_HIP(hipStreamSynchronize(stream));
  // from here on every value is the same on every rank: all of them refuse, or none does
  if (c.probe[0] != sig || c.probe[1] != -sig)
    fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: the ranks' results do not share their table layout (different key space, dictionaries, "
                             "aggregations or fixed-point scale): merge on the host by values");
  if (c.probe[4])
    fail(PG_ERR_UNSUPPORTED, "pg_result_all_reduce: a floating-point SUM accumulated in double (a column holding NaN / Inf) does not merge exactly");
  check_merge_bounds((uint64_t)c.probe[2], c.probe[3] != 0, c.probe[6]);
  T.sum_max_abs = (uint64_t)c.probe[2];   // the merged table's bound, for later merges
  T.has_digit_sums = c.probe[3] != 0;
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
