// Segment-level group trim on the device (round 5): GROUP BY high-cardinality ORDER BY <key | aggregation> LIMIT n is Pinot's everyday
// query, and GroupByOperator.getNextBlock trims the segment's groups to trimSize = max(5 x limit, minSegmentGroupTrimSize) before they
// leave the operator (core/operator/query/GroupByOperator.java:120-133 -> TableResizer#trimInSegmentResults, core/data/table/
// TableResizer.java:327-351).  Without this file every group of the dense accumulator table crosses PCIe and is boxed on the host (1 M
// groups x accumulators x 8 B) only to be dropped by the resizer.  Here the table never leaves HBM: one 64-bit order-preserving key per
// group from the FIRST order-by expression, an 8 x 8-bit radix select of the trimSize-th key among the groups that exist, then the
// survivors' group ids and accumulator rows are gathered into a compact block — only that block is copied.
//
// Ties.  Groups whose first-expression key equals the threshold key form the tie class.  With one ORDER BY expression any of them may
// fill the remaining places (the reference's heap does not define which).  With more expressions the whole class is handed over (up to the
// block's capacity; beyond it the host falls back to the full table) and the host finishes the selection with the full comparator
// (pg_exec.hip, assemble_result).
//
// Keys (smaller sorts first): a dictionary group column's dictId digit of the raw key (sorted dictionaries: dictIds order as the values do;
// descending: cardinality - 1 - dictId); an int64 accumulator row (COUNT, integer SUM / MIN / MAX, and the
// order-preserving keys of floating MIN / MAX) biased by 2^63.  An int64 order REFINES the order of the doubles the reference compares
// ((double) v is monotone): it only decides what the reference leaves tied.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_device.h"

extern "C" __global__ void __launch_bounds__(256) pg_trim_keys_kernel(const PgTrimArgs a) {
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_exist;
  const int t = threadIdx.x;
  s_hist[t] = 0;
  if (t == 0) s_exist = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (int64_t g = (int64_t)blockIdx.x * 256 + t; g < a.G; g += (int64_t)gridDim.x * 256) {
    const bool ex = a.table[(int64_t)a.exist_op * a.G + g] != a.exist_ident;
    uint64_t key = ~0ULL;
    if (ex) {
      if (a.key_op >= 0) {
        key = (uint64_t)a.table[(int64_t)a.key_op * a.G + g] ^ (1ULL << 63);
        if (a.descending) key = ~key;
        // the all-ones key marks "no such group".  A real row that maps onto it (Long.MAX_VALUE ascending, Long.MIN_VALUE descending) cannot be
        // told from the next key without merging two tie classes: the overflow flag sends the query to the whole-table path on the host
        if (key == ~0ULL) { a.ctrl[3] = 1u; key = ~0ULL - 1; }
      } else {
        // a group column's dictId digit: descending = counted from the top of the dictionary (complementing the 64-bit key would turn dictId 0
        // into the all-ones marker — and clamping that made dictId 0 and dictId 1 one tie class)
        const uint64_t digit = (uint64_t)((g / a.key_mult) % a.key_card);
        key = a.descending ? (uint64_t)a.key_card - 1u - digit : digit;
      }
      mine++;
      atomicAdd(&s_hist[key >> 56], 1u);
    }
    a.keys[g] = key;
  }
  if (mine) atomicAdd(&s_exist, mine);
  __syncthreads();
  if (s_hist[t]) atomicAdd(&a.ctrl[PG_TRIM_CTRL_HIST + t], s_hist[t]);
  if (t == 0 && s_exist) atomicAdd(&a.ctrl[0], s_exist);
}

// the digit of pass p - 1 that holds the `remaining`-th key, from that pass's histogram; returns the state of pass p
struct PgTrimState { uint64_t prefix; uint32_t remaining; };
__device__ __forceinline__ PgTrimState pg_trim_resolve(const PgTrimArgs& a, int p, uint32_t* s_scratch) {
  // (every workgroup resolves for itself; workgroup 0 also records the state for the next launch)
  const int t = threadIdx.x;
  __shared__ PgTrimState s_state;
  s_scratch[t] = a.ctrl[PG_TRIM_CTRL_HIST + 256 * (p - 1) + t];
  __syncthreads();
  if (t == 0) {
    uint64_t prefix = 0;
    uint32_t remaining;
    if (p == 1) {
      const uint32_t n_exist = a.ctrl[0];
      remaining = (uint32_t)a.k < n_exist ? (uint32_t)a.k : n_exist;
    } else {
      prefix = (uint64_t)a.ctrl[PG_TRIM_CTRL_STATE + 4 * (p - 1)] | ((uint64_t)a.ctrl[PG_TRIM_CTRL_STATE + 4 * (p - 1) + 1] << 32);
      remaining = a.ctrl[PG_TRIM_CTRL_STATE + 4 * (p - 1) + 2];
    }
    uint32_t cum = 0, d = 0;
    for (; d < 255; d++) {
      if (cum + s_scratch[d] >= remaining && remaining > 0) break;
      cum += s_scratch[d];
    }
    s_state.prefix = (prefix << 8) | d;
    s_state.remaining = remaining - (remaining > 0 ? cum : 0);
    if (blockIdx.x == 0) {
      a.ctrl[PG_TRIM_CTRL_STATE + 4 * p] = (uint32_t)s_state.prefix;
      a.ctrl[PG_TRIM_CTRL_STATE + 4 * p + 1] = (uint32_t)(s_state.prefix >> 32);
      a.ctrl[PG_TRIM_CTRL_STATE + 4 * p + 2] = s_state.remaining;
    }
  }
  __syncthreads();
  return s_state;
}

// pass p = 1 .. 7: histogram of byte 7 - p of the keys whose top p bytes equal the prefix
extern "C" __global__ void __launch_bounds__(256) pg_trim_pass_kernel(const PgTrimArgs a, int p) {
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_scratch[256];
  const int t = threadIdx.x;
  const PgTrimState st = pg_trim_resolve(a, p, s_scratch);
  s_hist[t] = 0;
  __syncthreads();
  const int shift = 8 * (7 - p);
  for (int64_t g = (int64_t)blockIdx.x * 256 + t; g < a.G; g += (int64_t)gridDim.x * 256) {
    const uint64_t key = a.keys[g];
    if (key != ~0ULL && (key >> (shift + 8)) == st.prefix) atomicAdd(&s_hist[(key >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  if (s_hist[t]) atomicAdd(&a.ctrl[PG_TRIM_CTRL_HIST + 256 * p + t], s_hist[t]);
}

// the threshold is complete after pass 7: survivors below it, then as much of the tie class as wanted, into the compact block
extern "C" __global__ void __launch_bounds__(256) pg_trim_select_kernel(const PgTrimArgs a) {
  __shared__ uint32_t s_scratch[256];
  const int t = threadIdx.x;
  const PgTrimState st = pg_trim_resolve(a, 8, s_scratch);   // prefix = the whole threshold key, remaining = places left for its tie class
  const uint64_t thr = st.prefix;
  const uint32_t n_exist = a.ctrl[0];
  const uint32_t k_eff = (uint32_t)a.k < n_exist ? (uint32_t)a.k : n_exist;
  const uint32_t n_lt = k_eff - st.remaining;
  if (blockIdx.x == 0 && t == 0) { a.ctrl[4] = st.remaining; a.ctrl[5] = n_lt; }
  for (int64_t g = (int64_t)blockIdx.x * 256 + t; g < a.G; g += (int64_t)gridDim.x * 256) {
    const uint64_t key = a.keys[g];
    if (key == ~0ULL || key > thr) continue;
    uint32_t slot;
    if (key < thr) {
      slot = atomicAdd(&a.ctrl[1], 1u);
    } else {
      const uint32_t e = atomicAdd(&a.ctrl[2], 1u);
      if (!a.take_whole_tie_class && e >= st.remaining) continue;
      slot = n_lt + e;
    }
    if (slot >= (uint32_t)a.cap) { a.ctrl[3] = 1u; continue; }
    a.out_gids[slot] = g;
    for (int o = 0; o < a.n_ops; o++) a.out_table[(int64_t)o * a.cap + slot] = a.table[(int64_t)o * a.G + g];
  }
}

// keys: one key per group + the first histogram; select: the seven remaining radix passes and the gather.  Two calls because the keys may come
// from ANOTHER table than the rows (numGroupsLimit by a prefix pass, pg_exec.hip: the keys are the prefix's first docIds, the rows the segment's).
extern "C" void pg_trim_launch_keys(const PgTrimArgs* args, int grid, hipStream_t stream) {
  const PgTrimArgs a = *args;
  hipLaunchKernelGGL(pg_trim_keys_kernel, dim3(grid), dim3(256), 0, stream, a);
}
extern "C" void pg_trim_launch_select(const PgTrimArgs* args, int grid, hipStream_t stream) {
  const PgTrimArgs a = *args;
  for (int p = 1; p <= 7; p++) hipLaunchKernelGGL(pg_trim_pass_kernel, dim3(grid), dim3(256), 0, stream, a, p);
  hipLaunchKernelGGL(pg_trim_select_kernel, dim3(grid), dim3(256), 0, stream, a);
}
extern "C" void pg_trim_launch(const PgTrimArgs* args, int grid, hipStream_t stream) {
  pg_trim_launch_keys(args, grid, stream);
  pg_trim_launch_select(args, grid, stream);
}
