// TEST INFRASTRUCTURE — a collective test double with the slice of the NCCL 2.x / RCCL C ABI that pinot_amd/csrc/pg_comm.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommDestroy, ncclGroupStart/End, ncclAllReduce, ncclAllGather,
// ncclGetErrorString).  It lets N host threads of ONE process, each holding its own "rank", run pg_result_all_reduce against
// tables that all live on the SAME GPU — so the multi-rank control flow of the cross-GPU merge (the probe, the refusals decided on
// reduced values, the grouped table launch, all-gather + OR of dictId sets) executes on a one-GPU box.  It is NOT a transport:
// every collective stages the ranks' send buffers on the host, rendezvouses the ranks on a condition variable, reduces on the
// host and copies the result into each rank's receive buffer.  Selected by PG_RCCL_LIBRARY=<this .so> (read when the library
// first opens RCCL); nothing under pinot_amd/ knows it exists.
//
// Where real RCCL would HANG, the double reports instead:
//   * a rank alone in a collective (the others never arrive) -> after FAKE_RCCL_TIMEOUT_MS (default 20 s) every waiter returns
//     ncclSystemError and the world is poisoned; the event is counted in fake_rccl_lonely_ranks();
//   * ranks that enqueue different collectives (kind / count / type / reduction, or a different number of them inside one group)
//     -> every rank returns ncclInvalidArgument, counted in fake_rccl_mismatched_collectives().
// Tests assert both counters stay 0 on the product path.
//
// FAKE_RCCL_HOST_BUFFERS=1 treats the buffers as host memory (plain memcpy): the double's own rendezvous / reduction logic is then
// testable without a GPU (tests/test_fake_rccl.py).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
}

namespace {

enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
enum { kInt8 = 0, kUint8 = 1, kInt32 = 2, kUint32 = 3, kInt64 = 4, kUint64 = 5, kFloat16 = 6, kFloat32 = 7, kFloat64 = 8 };
enum { kSum = 0, kProd = 1, kMax = 2, kMin = 3 };
enum { kAllReduce = 1, kAllGather = 2 };

std::atomic<int64_t> g_lonely{0}, g_mismatched{0}, g_collectives{0};

size_t type_size(int t) {
  switch (t) {
    case kInt8: case kUint8: return 1;
    case kInt32: case kUint32: case kFloat32: return 4;
    case kInt64: case kUint64: case kFloat64: return 8;
    default: return 0;
  }
}

bool host_buffers() {
  static const bool v = [] { const char* e = getenv("FAKE_RCCL_HOST_BUFFERS"); return e && e[0] == '1'; }();
  return v;
}
int timeout_ms() {
  static const int v = [] { const char* e = getenv("FAKE_RCCL_TIMEOUT_MS"); return e ? atoi(e) : 20000; }();
  return v;
}

struct Op {
  int kind = 0, dtype = 0, red = 0;
  size_t count = 0;
  const void* send = nullptr;
  void* recv = nullptr;
  hipStream_t stream = nullptr;
  std::vector<uint8_t> staged;   // this rank's send buffer, on the host
};

struct World {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool poisoned = false;
  int joined = 0, alive = 0;
  std::vector<std::vector<Op>> ops;   // per rank: the collectives of the launch in flight

  // sense-reversing barrier with a deadline; false = some rank never came (the world is poisoned for everybody)
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (poisoned) return false;
    const uint64_t gen = generation;
    if (++arrived == n) {
      arrived = 0;
      generation++;
      cv.notify_all();
      return true;
    }
    const bool ok = cv.wait_for(lk, std::chrono::milliseconds(timeout_ms()), [&] { return generation != gen || poisoned; });
    if (!ok) {
      poisoned = true;
      g_lonely++;
      fprintf(stderr, "fake_rccl: %d of %d ranks waited %d ms in a collective the others never entered (real RCCL would hang)\n", arrived, n,
              timeout_ms());
      cv.notify_all();
      return false;
    }
    return generation != gen;   // woken by the last arrival (true) or by a waiter that gave up (false)
  }
};

std::mutex g_worlds_mu;
std::map<uint64_t, std::shared_ptr<World>> g_worlds;   // by unique id, until every rank joined
std::atomic<uint64_t> g_next_id{1};

}  // namespace

struct ncclComm {
  std::shared_ptr<World> world;
  int rank = 0;
};

namespace {

thread_local int t_depth = 0;
thread_local std::vector<Op> t_pending;
thread_local ncclComm* t_comm = nullptr;
thread_local bool t_mixed_comms = false;

template <typename T>
void reduce_typed(T* acc, const T* in, size_t n, int red) {
  for (size_t i = 0; i < n; i++) {
    switch (red) {
      case kSum: acc[i] = (T)(acc[i] + in[i]); break;
      case kProd: acc[i] = (T)(acc[i] * in[i]); break;
      case kMax: acc[i] = in[i] > acc[i] ? in[i] : acc[i]; break;
      default: acc[i] = in[i] < acc[i] ? in[i] : acc[i]; break;
    }
  }
}

// integer sums wrap like the hardware's (two's complement): accumulate unsigned
void reduce_into(std::vector<uint8_t>& acc, const std::vector<uint8_t>& in, size_t n, int dtype, int red) {
  switch (dtype) {
    case kInt8:
      if (red == kSum || red == kProd) reduce_typed(reinterpret_cast<uint8_t*>(acc.data()), reinterpret_cast<const uint8_t*>(in.data()), n, red);
      else reduce_typed(reinterpret_cast<int8_t*>(acc.data()), reinterpret_cast<const int8_t*>(in.data()), n, red);
      break;
    case kUint8: reduce_typed(reinterpret_cast<uint8_t*>(acc.data()), reinterpret_cast<const uint8_t*>(in.data()), n, red); break;
    case kInt32:
      if (red == kSum || red == kProd) reduce_typed(reinterpret_cast<uint32_t*>(acc.data()), reinterpret_cast<const uint32_t*>(in.data()), n, red);
      else reduce_typed(reinterpret_cast<int32_t*>(acc.data()), reinterpret_cast<const int32_t*>(in.data()), n, red);
      break;
    case kUint32: reduce_typed(reinterpret_cast<uint32_t*>(acc.data()), reinterpret_cast<const uint32_t*>(in.data()), n, red); break;
    case kInt64:
      if (red == kSum || red == kProd) reduce_typed(reinterpret_cast<uint64_t*>(acc.data()), reinterpret_cast<const uint64_t*>(in.data()), n, red);
      else reduce_typed(reinterpret_cast<int64_t*>(acc.data()), reinterpret_cast<const int64_t*>(in.data()), n, red);
      break;
    case kUint64: reduce_typed(reinterpret_cast<uint64_t*>(acc.data()), reinterpret_cast<const uint64_t*>(in.data()), n, red); break;
    case kFloat32: reduce_typed(reinterpret_cast<float*>(acc.data()), reinterpret_cast<const float*>(in.data()), n, red); break;
    case kFloat64: reduce_typed(reinterpret_cast<double*>(acc.data()), reinterpret_cast<const double*>(in.data()), n, red); break;
    default: break;
  }
}

int copy_in(void* host, const void* buf, size_t bytes, hipStream_t stream) {
  if (!bytes) return kSuccess;
  if (host_buffers()) { memcpy(host, buf, bytes); return kSuccess; }
  // the caller enqueued work on `stream` in front of the collective: order after it, as a real collective on that stream would
  if (hipStreamSynchronize(stream) != hipSuccess) return kUnhandledCudaError;
  return hipMemcpy(host, buf, bytes, hipMemcpyDeviceToHost) == hipSuccess ? kSuccess : kUnhandledCudaError;
}
int copy_out(void* buf, const void* host, size_t bytes, hipStream_t stream) {
  if (!bytes) return kSuccess;
  if (host_buffers()) { memcpy(buf, host, bytes); return kSuccess; }
  if (hipMemcpyAsync(buf, host, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return kUnhandledCudaError;
  return hipStreamSynchronize(stream) == hipSuccess ? kSuccess : kUnhandledCudaError;   // `host` dies with this launch
}

// One launch (a single collective, or everything between ncclGroupStart and the matching ncclGroupEnd) of one rank.
int launch(ncclComm* c, std::vector<Op>& mine) {
  World& w = *c->world;
  if (mine.empty()) return kSuccess;
  for (Op& op : mine) {
    const size_t bytes = op.count * type_size(op.dtype);
    op.staged.resize(bytes);
    const int rc = copy_in(op.staged.data(), op.send, bytes, op.stream);
    if (rc != kSuccess) return rc;
  }
  {
    std::lock_guard<std::mutex> lk(w.mu);
    w.ops[(size_t)c->rank] = std::move(mine);
  }
  mine.clear();
  if (!w.barrier()) return kSystemError;                 // everybody published
  int rc = kSuccess;
  const std::vector<Op>& own = w.ops[(size_t)c->rank];
  for (int r = 0; r < w.n && rc == kSuccess; r++) {      // every rank sees the same lists, so every rank decides alike
    const std::vector<Op>& other = w.ops[(size_t)r];
    if (other.size() != own.size()) { rc = kInvalidArgument; break; }
    for (size_t i = 0; i < own.size(); i++)
      if (other[i].kind != own[i].kind || other[i].count != own[i].count || other[i].dtype != own[i].dtype ||
          (own[i].kind == kAllReduce && other[i].red != own[i].red)) { rc = kInvalidArgument; break; }
  }
  if (rc == kInvalidArgument) {
    if (c->rank == 0) {
      g_mismatched++;
      fprintf(stderr, "fake_rccl: the ranks enqueued different collectives in one launch (real RCCL: undefined, in practice a hang)\n");
    }
  } else {
    for (size_t i = 0; i < own.size() && rc == kSuccess; i++) {
      const Op& op = own[i];
      const size_t bytes = op.count * type_size(op.dtype);
      std::vector<uint8_t> out;
      if (op.kind == kAllReduce) {
        out = w.ops[0][i].staged;
        for (int r = 1; r < w.n; r++) reduce_into(out, w.ops[(size_t)r][i].staged, op.count, op.dtype, op.red);
      } else {
        out.resize(bytes * (size_t)w.n);
        for (int r = 0; r < w.n; r++)
          if (bytes) memcpy(out.data() + bytes * (size_t)r, w.ops[(size_t)r][i].staged.data(), bytes);
      }
      rc = copy_out(op.recv, out.data(), out.size(), op.stream);
      g_collectives++;
    }
  }
  if (!w.barrier()) return kSystemError;                 // everybody has read everybody's staging: the slots may be reused
  return rc;
}

int enqueue(ncclComm* c, Op&& op) {
  if (!c || !c->world) return kInvalidArgument;
  if (type_size(op.dtype) == 0) return kInvalidArgument;
  if (t_depth > 0) {
    if (t_comm && t_comm != c) t_mixed_comms = true;   // one thread driving several ranks inside a group: not what the product does
    t_comm = c;
    t_pending.push_back(std::move(op));
    return kSuccess;
  }
  std::vector<Op> one;
  one.push_back(std::move(op));
  return launch(c, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return kInvalidArgument;
  memset(id, 0, sizeof(*id));
  const uint64_t v = g_next_id++;
  memcpy(id->internal, "FAKERCCL", 8);
  memcpy(id->internal + 8, &v, 8);
  return kSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks || memcmp(id.internal, "FAKERCCL", 8) != 0) return kInvalidArgument;
  uint64_t key;
  memcpy(&key, id.internal + 8, 8);
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_worlds_mu);
    auto it = g_worlds.find(key);
    if (it == g_worlds.end()) {
      w = std::make_shared<World>();
      w->n = nranks;
      w->ops.resize((size_t)nranks);
      g_worlds[key] = w;
    } else {
      w = it->second;
    }
    if (w->n != nranks) return kInvalidArgument;
    if (++w->joined == nranks) g_worlds.erase(key);
    w->alive++;
  }
  if (!w->barrier()) return kSystemError;   // ncclCommInitRank is collective
  auto* c = new ncclComm;
  c->world = w;
  c->rank = rank;
  *comm = c;
  return kSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  (void)devlist;   // real RCCL rejects a device listed twice; the double exists to allow exactly that
  if (!comms || ndev < 1) return kInvalidArgument;
  auto w = std::make_shared<World>();
  w->n = ndev;
  w->ops.resize((size_t)ndev);
  w->joined = w->alive = ndev;
  for (int i = 0; i < ndev; i++) {
    auto* c = new ncclComm;
    c->world = w;
    c->rank = i;
    comms[i] = c;
  }
  return kSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return kSuccess;
}

ncclResult_t ncclGroupStart() {
  t_depth++;
  return kSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return kInvalidUsage;
  if (--t_depth > 0) return kSuccess;
  ncclComm* c = t_comm;
  const bool mixed = t_mixed_comms;
  t_comm = nullptr;
  t_mixed_comms = false;
  if (mixed) { t_pending.clear(); return kInvalidUsage; }
  if (!c) return kSuccess;
  std::vector<Op> ops;
  ops.swap(t_pending);
  return launch(c, ops);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, ncclComm_t comm, hipStream_t stream) {
  if (op < kSum || op > kMin) return kInvalidArgument;
  Op o;
  o.kind = kAllReduce; o.dtype = datatype; o.red = op; o.count = count; o.send = sendbuff; o.recv = recvbuff; o.stream = stream;
  return enqueue(comm, std::move(o));
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, ncclComm_t comm, hipStream_t stream) {
  Op o;
  o.kind = kAllGather; o.dtype = datatype; o.count = sendcount; o.send = sendbuff; o.recv = recvbuff; o.stream = stream;
  return enqueue(comm, std::move(o));
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case kSuccess: return "no error";
    case kUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case kSystemError: return "a rank waited alone in a collective (fake_rccl timeout)";
    case kInternalError: return "internal error (fake_rccl)";
    case kInvalidArgument: return "invalid argument / the ranks enqueued different collectives (fake_rccl)";
    case kInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "unknown result (fake_rccl)";
  }
}

// ---- what the tests read -----------------------------------------------------------------------------------------------------------
int64_t fake_rccl_lonely_ranks() { return g_lonely.load(); }
int64_t fake_rccl_mismatched_collectives() { return g_mismatched.load(); }
int64_t fake_rccl_collectives() { return g_collectives.load(); }

}  // extern "C"
