// Execution: launches the segment query kernel(s) on the calling thread's stream and assembles results.
//   GroupByOperator.getNextBlock / AggregationOperator.getNextBlock  (core/operator/query/GroupByOperator.java:100-140)
//   ExecutionStatistics synthesis                                    (core/operator/query/GroupByOperator.java:148-153,
//                                                                     core/operator/DocIdSetOperator.java:105-108)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cmath>

#include "pg_internal.hpp"

#define PG_DECL_FAST(NAME) extern "C" __global__ void NAME(const PgQueryPlan p);
#define PG_DECL_GENERIC PG_DECL_FAST(pg_generic_query_f) PG_DECL_FAST(pg_generic_query_l) PG_DECL_FAST(pg_generic_query_g)
PG_DECL_GENERIC
PG_DECL_FAST(pg_fast_none_f) PG_DECL_FAST(pg_fast_none_a) PG_DECL_FAST(pg_fast_i32range_f) PG_DECL_FAST(pg_fast_i32range_a)
PG_DECL_FAST(pg_fast_dictrange_f) PG_DECL_FAST(pg_fast_dictrange_a) PG_DECL_FAST(pg_fast_dictlut_f) PG_DECL_FAST(pg_fast_dictlut_a)
PG_DECL_FAST(pg_fast_multi_f) PG_DECL_FAST(pg_fast_multi_a) PG_DECL_FAST(pg_fast_multi_w) PG_DECL_FAST(pg_fast_none_w)
PG_DECL_FAST(pg_nogroup_s1) PG_DECL_FAST(pg_nogroup_s2) PG_DECL_FAST(pg_nogroup_da) PG_DECL_FAST(pg_nogroup_dg) PG_DECL_FAST(pg_nogroup_dl) PG_DECL_FAST(pg_dictrange_fo) PG_DECL_FAST(pg_fast_i32range_d) PG_DECL_FAST(pg_fast_i32range_p) PG_DECL_FAST(pg_fast_i32range_fp) PG_DECL_FAST(pg_fast_i32range_s) PG_DECL_FAST(pg_spec_none) PG_DECL_FAST(pg_spec_scan) PG_DECL_FAST(pg_spec_index) PG_DECL_FAST(pg_fast_i32range_st)
// pg_kernels_specd.hip: the loader / consumer frame over dictionary-encoded scan / value columns (_r raw INT values, _a arithmetic dictionary, _g gathered)
PG_DECL_FAST(pg_fast_dictrange_s_r_dma) PG_DECL_FAST(pg_fast_dictrange_s_a_dma) PG_DECL_FAST(pg_fast_dictrange_s_g_dma) PG_DECL_FAST(pg_fast_dictrange_s_r) PG_DECL_FAST(pg_fast_dictrange_st_r) PG_DECL_FAST(pg_specd_none_r) PG_DECL_FAST(pg_specd_scan_r) PG_DECL_FAST(pg_specd_index_r) PG_DECL_FAST(pg_fast_dictrange_s_a) PG_DECL_FAST(pg_fast_dictrange_st_a) PG_DECL_FAST(pg_specd_none_a) PG_DECL_FAST(pg_specd_scan_a) PG_DECL_FAST(pg_specd_index_a) PG_DECL_FAST(pg_fast_dictrange_s_g) PG_DECL_FAST(pg_fast_dictrange_st_g) PG_DECL_FAST(pg_specd_none_g) PG_DECL_FAST(pg_specd_scan_g) PG_DECL_FAST(pg_specd_index_g)
// pg_kernels_specw.hip: the same shapes with a shared stage per workgroup (whole stages requested as long rows straight into LDS)
PG_DECL_FAST(pg_fast_dictrange_w_r) PG_DECL_FAST(pg_fast_dictrange_wt_r) PG_DECL_FAST(pg_specw_none_r) PG_DECL_FAST(pg_specw_scan_r) PG_DECL_FAST(pg_specw_index_r) PG_DECL_FAST(pg_fast_dictrange_w_a) PG_DECL_FAST(pg_fast_dictrange_wt_a) PG_DECL_FAST(pg_specw_none_a) PG_DECL_FAST(pg_specw_scan_a) PG_DECL_FAST(pg_specw_index_a) PG_DECL_FAST(pg_fast_dictrange_w_g) PG_DECL_FAST(pg_fast_dictrange_wt_g) PG_DECL_FAST(pg_specw_none_g) PG_DECL_FAST(pg_specw_scan_g) PG_DECL_FAST(pg_specw_index_g)
PG_DECL_FAST(pg_dense_count_1) PG_DECL_FAST(pg_dense_count_2) PG_DECL_FAST(pg_dense_count_3) PG_DECL_FAST(pg_dense_count_4)
PG_DECL_FAST(pg_dense_count_5) PG_DECL_FAST(pg_dense_count_6) PG_DECL_FAST(pg_dense_count_7) PG_DECL_FAST(pg_dense_count_8)
PG_DECL_FAST(pg_dict_count_1) PG_DECL_FAST(pg_dict_count_2) PG_DECL_FAST(pg_dict_count_3) PG_DECL_FAST(pg_dict_count_4)
PG_DECL_FAST(pg_dict_count_5) PG_DECL_FAST(pg_dict_count_6) PG_DECL_FAST(pg_dict_count_7) PG_DECL_FAST(pg_dict_count_8)
PG_DECL_FAST(pg_pipe_w0_none) PG_DECL_FAST(pg_pipe_w0_index) PG_DECL_FAST(pg_pipe_w0_scan) PG_DECL_FAST(pg_pipe_w0_index_scan)
PG_DECL_FAST(pg_pipe_w32_none) PG_DECL_FAST(pg_pipe_w32_index) PG_DECL_FAST(pg_pipe_w32_scan) PG_DECL_FAST(pg_pipe_w32_index_scan)
PG_DECL_FAST(pg_pipe_w64_none) PG_DECL_FAST(pg_pipe_w64_index) PG_DECL_FAST(pg_pipe_w64_scan) PG_DECL_FAST(pg_pipe_w64_index_scan)
PG_DECL_FAST(pg_pipe_wd_none) PG_DECL_FAST(pg_pipe_wd_index) PG_DECL_FAST(pg_pipe_wd_scan) PG_DECL_FAST(pg_pipe_wd_index_scan)
PG_DECL_FAST(pg_pipe_scan) PG_DECL_FAST(pg_pipe_scan_tail) PG_DECL_FAST(pg_pipe_index_scan_tail) PG_DECL_FAST(pg_pipe_none) PG_DECL_FAST(pg_pipe_tail)
PG_DECL_FAST(pg_pipe_index) PG_DECL_FAST(pg_pipe_index_tail) PG_DECL_FAST(pg_pipe_index2) PG_DECL_FAST(pg_pipe_index2_tail)
PG_DECL_FAST(pg_pipe_scan_vscan) PG_DECL_FAST(pg_pipe_index_scan_vscan)
PG_DECL_FAST(pg_mv_query_f) PG_DECL_FAST(pg_mv_query_l) PG_DECL_FAST(pg_mv_query_g)   // pg_kernels_mv.hip
PG_DECL_FAST(pg_mv_group_4) PG_DECL_FAST(pg_mv_group_8) PG_DECL_FAST(pg_mv_aggr_4) PG_DECL_FAST(pg_mv_aggr_8)   // pg_kernels_mvg.hip: GROUP BY one multi-value column; the *MV functions over one
extern "C" const int pg_scan_waves_per_block;   // pg_kernels_scan.hip: wavefronts per workgroup of pg_fast_i32range_fp
extern "C" const int pg_pipe_waves_per_block;   // pg_kernels_pipe.hip: wavefronts per workgroup of pg_fast_i32range_p
extern "C" const int pg_spec_waves_per_block;   // pg_kernels_spec.hip: pg_fast_i32range_s (4 loader + 8 consumer wavefronts)
extern "C" int pg_spec_stage_bytes(int bits0, int bits1);
extern "C" const int pg_specd_waves_per_block;   // pg_kernels_specd.hip: pg_fast_dictrange_s family (dictionary-encoded scan / value columns)
extern "C" int pg_specd_stage_bytes(int scan_bits, int value_bits, int bits0, int bits1, int areas);
extern "C" const int pg_specw_waves_per_block;   // pg_kernels_specw.hip: pg_fast_dictrange_w family (the shared-stage frame of the same plans)
extern "C" int pg_specw_stage_bytes(int scan_bits, int value_bits, int bits0, int bits1, int n_bitmaps);
extern "C" int pg_specw_list_bytes();
PG_DECL_FAST(pg_fast_multi_wd) PG_DECL_FAST(pg_fast_none_wd) PG_DECL_FAST(pg_generic_query_ld) PG_DECL_FAST(pg_generic_query_gd)
extern "C" __global__ void pg_reduce_partials_kernel(const int64_t* partials, int64_t* out, int n_wg, int n_ops,
                                                     int n_groups, const PgAccOp* ops, unsigned long long* stats, int reduce);
extern "C" __global__ void pg_reduce_parts_kernel(const int64_t* partials, int64_t* out, int n_wg, int n_ops, int n_groups,
                                                   int n_parts, int part_groups, const PgAccOp* ops);
PG_DECL_FAST(pg_radix_count_kernel) PG_DECL_FAST(pg_radix_scatter_kernel) PG_DECL_FAST(pg_radix_aggregate_kernel)
PG_DECL_FAST(pg_radix_scatter_packed_kernel) PG_DECL_FAST(pg_radix_aggregate_packed_kernel)
PG_DECL_FAST(pg_hash_count_kernel) PG_DECL_FAST(pg_hash_scatter_kernel) PG_DECL_FAST(pg_hash_aggregate_kernel)
PG_DECL_FAST(pg_p2_scatter_1) PG_DECL_FAST(pg_p2_scatter_2) PG_DECL_FAST(pg_p2_scatter_3) PG_DECL_FAST(pg_p2_scatter_4)
PG_DECL_FAST(pg_p2_scatter_1f) PG_DECL_FAST(pg_p2_scatter_2f) PG_DECL_FAST(pg_p2_scatter_1f_key) PG_DECL_FAST(pg_p2_scatter_1f_hll)
PG_DECL_FAST(pg_p2_scatter_o_key) PG_DECL_FAST(pg_p2_scatter_o_raw) PG_DECL_FAST(pg_p2_scatter_o_dict)
PG_DECL_FAST(pg_p2_scatter_ow_key) PG_DECL_FAST(pg_p2_scatter_ow_raw) PG_DECL_FAST(pg_p2_scatter_ow_dict)
PG_DECL_FAST(pg_p2_aggregate_1) PG_DECL_FAST(pg_p2_aggregate_2) PG_DECL_FAST(pg_p2_aggregate_3) PG_DECL_FAST(pg_p2_aggregate_4)
PG_DECL_FAST(pg_p2_aggregate_1n) PG_DECL_FAST(pg_p2_aggregate_2n) PG_DECL_FAST(pg_p2_aggregate_3n) PG_DECL_FAST(pg_p2_aggregate_4n)
PG_DECL_FAST(pg_p2_index_count_kernel) PG_DECL_FAST(pg_p2_index_scan_kernel) PG_DECL_FAST(pg_p2_index_fill_kernel)
PG_DECL_FAST(pg_p2_scatter_stream) PG_DECL_FAST(pg_p2_aggregate_1b) PG_DECL_FAST(pg_p2_aggregate_1set) PG_DECL_FAST(pg_p2_aggregate_1s) PG_DECL_FAST(pg_p2_aggregate_2s) PG_DECL_FAST(pg_p2_aggregate_1sg) PG_DECL_FAST(pg_p2_aggregate_2sg)
// pg_kernels_oct.hip: oct-layout DISTINCTCOUNTHLL / DISTINCTCOUNT kernels (LDS-resident states; pruned offers) and their small helpers
PG_DECL_FAST(pg_oct_l) PG_DECL_FAST(pg_oct_lm) PG_DECL_FAST(pg_oct_c) PG_DECL_FAST(pg_oct_p) PG_DECL_FAST(pg_oct_pm)
extern "C" __global__ void pg_oct_merge_floor_kernel(const uint32_t* partials, uint32_t* regs, uint8_t* floors, int n_groups, int log2m, int radix_shift,
                                                      int slices);
extern "C" __global__ void pg_oct_pass_reset_kernel(uint32_t* p2_meta, int64_t n_meta, uint32_t* p2_ctrl, uint32_t* cursor);
extern "C" __global__ void pg_oct_stream_index_kernel(uint32_t* ctrl, int n_regions);
extern "C" __global__ void pg_oct_reduce_counts_kernel(const uint32_t* counts, int64_t* out, int n_parts, int n_groups);
extern "C" const int pg_p2_round_quads[5];   // pg_kernels_part.hip: quads per lane and round of the scatter kernel, by plane count
extern "C" __global__ void pg_radix_offsets_kernel(uint32_t* hist, uint32_t* bucket_total, int n_wg, int n_buckets, int stage, int stage_waves);
extern "C" __global__ void pg_radix_bucket_scan_kernel(const uint32_t* bucket_total, uint32_t* bucket_start, int n_buckets);
extern "C" __global__ void pg_radix_reduce_kernel(const int64_t* partials, int64_t* out, int n_ops, int n_groups, int radix_shift, int slices,
                                                   const PgAccOp* ops);
extern "C" __global__ void pg_reduce_aux_kernel(const uint32_t* partials, uint32_t* out, int n_wg, int64_t n_words, int bytewise_max);
extern "C" __global__ void pg_radix_reduce_aux_kernel(const uint32_t* partials, uint32_t* out, int slices, int64_t bucket_words, int64_t n_words, int bitwise_or);
extern "C" __global__ void pg_fill_i64_kernel(int64_t* dst, int64_t n_per_op, int n_ops, const PgAccOp* ops);
extern "C" __global__ void pg_expand_docids_kernel(const uint64_t* words, const int64_t* tile_offsets, int32_t* out,
                                                   int n_tiles);

// DISTINCTCOUNT / DISTINCTCOUNTHLL final values without moving the states (PG_QUERY_FLAG_FINAL_DISTINCT): one wavefront per group folds
// the group's state into its final value — a dictId set into its size; 2^log2m HyperLogLog registers through stream-lib's
// HyperLogLog#cardinality: registerSum = Σ 1.0 / (1 << register) is accumulated as the INTEGER Σ 2^(40 - register) (every
// term and every partial sum of the Java loop is a multiple of 2^-40 far below 2^53, so the integer sum IS that double sum, whatever the
// order), estimate = alphaMM * (1 / registerSum) in IEEE double (division and multiplication are correctly rounded here as in Java),
// and the small-range branch round(m * ln(m / zeros)) — one of m + 1 values — comes from a table the host filled with its libm.
// A star-tree answer of BASELINE config 5 then copies 100 KB back instead of 3.3 MB of registers.
extern "C" __global__ void __launch_bounds__(256) pg_aux_finish_kernel(const uint32_t* __restrict__ region, int kind, int words_per_group,
                                                                       int n_groups, double alpha_mm, int m,
                                                                       const long long* __restrict__ small_range, long long* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int g = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (g >= n_groups) return;
  const uint32_t* w = region + (int64_t)g * words_per_group;
  unsigned long long a = 0, z = 0;
  for (int i = lane; i < words_per_group; i += 64) {
    const uint32_t x = w[i];
    if (kind == PG_AUX_DICT_SET) {
      a += (unsigned long long)__popc(x);
    } else {
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint32_t r = (x >> (8 * b)) & 0xFFu;
        a += 1ULL << (40u - (r > 40u ? 40u : r));
        z += r == 0u ? 1ULL : 0ULL;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    z += __shfl_xor(z, off, 64);
  }
  if (lane != 0) return;
  if (kind == PG_AUX_DICT_SET) { out[g] = (long long)a; return; }
  const double register_sum = ldexp((double)a, -40);
  const double estimate = alpha_mm * (1.0 / register_sum);
  if (estimate <= (5.0 / 2.0) * (double)m) out[g] = small_range[z < (unsigned long long)m ? z : (unsigned long long)m];
  else out[g] = (long long)floor(estimate + 0.5);
}

// The tail of a PG_QUERY_FLAG_FINAL_DISTINCT query in ONE launch (round 5): the reduction of the workgroups' partial tables
// (pg_reduce_parts_kernel's or pg_reduce_partials_kernel's work), the statistics counters, and the final values of every DISTINCTCOUNT /
// HyperLogLog state (pg_aux_finish_kernel's work) are independent of one another — they only wait for the query kernel — so their workgroups
// share a grid: [0, b_table) table slots, [b_table] statistics, then (n_groups + 3) / 4 workgroups per auxiliary state.  Results go straight
// into the page-locked result block (mapped into the device's address space): no copy command behind the kernel.  The star-tree route of
// BASELINE config 5 was five dependent device operations for 50 us of work (fill, query, reduce_parts, reduce_partials + copy,
// aux_finish + copy: p50 0.19 ms, profiles/r04_aa_star_tree_latency.txt); it is two now.  With `rezero` a state is zeroed once it has
// been folded, so the next query on this stream finds its state area clean and skips the fill.
struct PgFinishAux {
  uint32_t* region;
  const long long* small_range;
  long long* out;
  double alpha_mm;
  int32_t kind, words_per_group, m, pad;
};
struct PgFinishArgs {
  const int64_t* partials;
  int64_t* out;                 // [n_ops][n_groups] + PG_MAX_STATS counters
  const PgAccOp* ops;
  unsigned long long* stats;
  int32_t n_wg, n_ops, n_groups, n_parts, part_groups;
  int32_t mode;                 // 0: the table is final already; 1: per-workgroup tables (one wavefront per slot); 2: range-partitioned tables
  int32_t b_table, b_aux, n_aux, rezero;
  PgFinishAux aux[PG_MAX_AUX];
};
__device__ __forceinline__ int64_t pg_finish_combine(int kind, int64_t a, int64_t b) {
  if (kind == 0) return __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b));
  if (kind == 1) return a + b;
  if (kind == 2) return b < a ? b : a;
  return b > a ? b : a;
}
extern "C" __global__ void __launch_bounds__(256) pg_finish_fused_kernel(const PgFinishArgs a) {
  __shared__ int64_t s_acc[4][64];
  const int b = (int)blockIdx.x, lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
  const int64_t n_out = (int64_t)a.n_ops * a.n_groups;
  if (b < a.b_table) {
    if (a.mode == 1) {   // pg_reduce_partials_kernel: one wavefront per slot, lanes stride over the workgroups, fixed butterfly order
      const int64_t i = (int64_t)b * 4 + quarter;
      if (i >= n_out) return;
      const PgAccOp op = a.ops[i / a.n_groups];
      const int kind = (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) ? 0 : ((op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) ? 1 : (op.fn == PG_ACC_MIN ? 2 : 3));
      int64_t acc = pg_acc_identity(op.fn, op.is_float);
      for (int w = lane; w < a.n_wg; w += 64) acc = pg_finish_combine(kind, acc, a.partials[(int64_t)w * n_out + i]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc = pg_finish_combine(kind, acc, __shfl_xor((long long)acc, off, 64));
      if (lane == 0) a.out[i] = acc;
      return;
    }
    // pg_reduce_parts_kernel: 64 slots per workgroup, its four wavefronts split the workgroups that own the slots' key range
    const int64_t i = (int64_t)b * 64 + lane;
    const bool live = i < n_out;
    const int o = live ? (int)(i / a.n_groups) : 0, g = live ? (int)(i % a.n_groups) : 0;
    const int range = g / a.part_groups, l = g % a.part_groups;
    const PgAccOp op = a.ops[o];
    const int kind = (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) ? 0 : ((op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) ? 1 : (op.fn == PG_ACC_MIN ? 2 : 3));
    int64_t acc = pg_acc_identity(op.fn, op.is_float);
    const int64_t wg_stride = (int64_t)a.n_ops * a.part_groups;
    const int64_t* src = a.partials + (int64_t)o * a.part_groups + l;
    const int per_range = (a.n_wg >> 3) / a.n_parts;
    for (int j = quarter; j < per_range; j += 4) {
      const int64_t w0 = 8 * ((int64_t)range + (int64_t)a.n_parts * j);
      int64_t v[8];
#pragma unroll
      for (int x = 0; x < 8; x++) v[x] = src[(w0 + x) * wg_stride];
#pragma unroll
      for (int x = 0; x < 8; x++) acc = pg_finish_combine(kind, acc, v[x]);
    }
    s_acc[quarter][lane] = acc;
    __syncthreads();
    if (quarter == 0 && live)
      a.out[i] = pg_finish_combine(kind, pg_finish_combine(kind, s_acc[0][lane], s_acc[1][lane]), pg_finish_combine(kind, s_acc[2][lane], s_acc[3][lane]));
    return;
  }
  if (b == a.b_table) {   // the statistics counters behind the table, re-zeroed for the next query on this stream
    if (threadIdx.x < PG_MAX_STATS) {
      a.out[n_out + threadIdx.x] = (int64_t)a.stats[threadIdx.x];
      a.stats[threadIdx.x] = 0;
    }
    return;
  }
  const int rel = b - a.b_table - 1;
  const int x = rel / a.b_aux;
  if (x >= a.n_aux) return;
  const PgFinishAux A = a.aux[x];
  const int g = (rel % a.b_aux) * 4 + quarter;
  const int G1 = a.n_groups > 0 ? a.n_groups : 1;
  if (g >= G1) return;
  uint32_t* w = A.region + (int64_t)g * A.words_per_group;
  unsigned long long acc = 0, z = 0;
  for (int i = lane; i < A.words_per_group; i += 64) {
    const uint32_t v = w[i];
    if (a.rezero && v) w[i] = 0u;
    if (A.kind == PG_AUX_DICT_SET) {
      acc += (unsigned long long)__popc(v);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t r = (v >> (8 * k)) & 0xFFu;
        acc += 1ULL << (40u - (r > 40u ? 40u : r));
        z += r == 0u ? 1ULL : 0ULL;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_xor(acc, off, 64);
    z += __shfl_xor(z, off, 64);
  }
  if (lane != 0) return;
  if (A.kind == PG_AUX_DICT_SET) { A.out[g] = (long long)acc; return; }
  const double register_sum = ldexp((double)acc, -40);
  const double estimate = A.alpha_mm * (1.0 / register_sum);
  if (estimate <= (5.0 / 2.0) * (double)A.m) A.out[g] = A.small_range[z < (unsigned long long)A.m ? z : (unsigned long long)A.m];
  else A.out[g] = (long long)floor(estimate + 0.5);
}

namespace pg {

// ---- errors / device buffers ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error() { return g_last_error; }
void fail(int32_t status, const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(status, buf);
}

void DeviceBuffer::alloc(size_t n, bool zero) {
  release();
  size = (n + 255) & ~(size_t)255;
  PG_HIP(hipGetDevice(&device));
  PG_HIP(hipMalloc(&ptr, size));
  if (zero) {
    // hipMemset on device memory is asynchronous (legacy default stream) and the per-thread streams are non-blocking: without the wait a
    // kernel launched next on such a stream could run BEFORE the fill and have its output zeroed afterwards (seen once in ~10^3 filter
    // calls with four processes sharing the GPU: pg_docidset_copy_docids returned zeros for a set of cardinality 2)
    PG_HIP(hipMemset(ptr, 0, size));
    PG_HIP(hipStreamSynchronize(nullptr));
  }
}
void DeviceBuffer::release() {
  if (ptr) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && device >= 0 && cur != device) {
      (void)hipSetDevice(device);
      (void)hipFree(ptr);
      (void)hipSetDevice(cur);
    } else {
      (void)hipFree(ptr);
    }
  }
  ptr = nullptr;
  size = 0;
}
void DeviceBuffer::upload(const void* src, size_t n, size_t off) {
  if (n == 0) return;
  if (off + n > size) fail(PG_ERR_INTERNAL, "upload past the end of a device buffer");
  PG_HIP(hipMemcpy(static_cast<uint8_t*>(ptr) + off, src, n, hipMemcpyHostToDevice));
}

// ---- devices / per-thread contexts ------------------------------------------------------------------------------------------
// One process may hold segments on several GPUs (the reference runs every segment of a server in one JVM, one worker task per
// segment — BaseCombineOperator.java:81-142): a segment names its device, every entry point makes that device current on the
// calling thread, and each (thread, device) pair owns a stream, events and work areas.
#define PG_MAX_DEVICES 32
struct DeviceInfo {
  std::atomic<int> ready{0};
  int num_cus = 256;
  size_t lds_per_cu = 160 * 1024;
};
static DeviceInfo g_devices[PG_MAX_DEVICES];
static std::mutex g_devices_mu;
static std::atomic<int> g_default_device{-1};
static thread_local int t_device = 0;   // the device use_device() made current on this thread

static int device_count_or_fail() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    fail(PG_ERR_DEVICE, "no HIP device available (%s): libpinot_gpu has no CPU fallback", hipGetErrorName(e));
  return n;
}

void use_device(int ordinal) {
  if (ordinal < 0 || ordinal >= PG_MAX_DEVICES) fail(PG_ERR_INVALID_ARGUMENT, "device ordinal %d out of range", ordinal);
  DeviceInfo& di = g_devices[ordinal];
  if (!di.ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> g(g_devices_mu);
    if (!di.ready.load(std::memory_order_relaxed)) {
      const int n = device_count_or_fail();
      if (ordinal >= n) fail(PG_ERR_INVALID_ARGUMENT, "device ordinal %d out of range (0..%d)", ordinal, n - 1);
      PG_HIP(hipSetDevice(ordinal));
      hipDeviceProp_t prop;
      PG_HIP(hipGetDeviceProperties(&prop, ordinal));
      di.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      di.lds_per_cu = prop.maxSharedMemoryPerMultiProcessor > 0 ? (size_t)prop.maxSharedMemoryPerMultiProcessor : 160 * 1024;
      // opt in to large dynamic LDS for the query kernels (function attributes are per device)
      typedef void (*QueryKernel)(const PgQueryPlan);
      const QueryKernel all[] = {pg_generic_query_f, pg_generic_query_l, pg_generic_query_g, pg_fast_none_f, pg_fast_none_a, pg_fast_i32range_f, pg_fast_i32range_a,
                                 pg_fast_dictrange_f, pg_fast_dictrange_a, pg_fast_dictlut_f, pg_fast_dictlut_a, pg_fast_multi_f, pg_fast_multi_a, pg_fast_multi_w, pg_fast_none_w, pg_fast_multi_wd, pg_fast_none_wd, pg_generic_query_ld, pg_generic_query_gd, pg_fast_i32range_d, pg_fast_i32range_p, pg_pipe_scan, pg_pipe_scan_tail, pg_pipe_index_scan_tail, pg_pipe_none, pg_pipe_tail, pg_pipe_index, pg_pipe_index_tail, pg_pipe_index2, pg_pipe_index2_tail, pg_pipe_scan_vscan, pg_pipe_index_scan_vscan, pg_pipe_w0_none, pg_pipe_w0_index, pg_pipe_w0_scan, pg_pipe_w0_index_scan, pg_pipe_w32_none, pg_pipe_w32_index, pg_pipe_w32_scan, pg_pipe_w32_index_scan, pg_pipe_w64_none, pg_pipe_w64_index, pg_pipe_w64_scan, pg_pipe_w64_index_scan, pg_pipe_wd_none, pg_pipe_wd_index, pg_pipe_wd_scan, pg_pipe_wd_index_scan, pg_mv_query_f, pg_mv_query_l, pg_mv_query_g,
                                 pg_radix_aggregate_kernel, pg_hash_aggregate_kernel,
                                 pg_p2_scatter_1, pg_p2_scatter_2, pg_p2_scatter_3, pg_p2_scatter_4, pg_p2_scatter_1f, pg_p2_scatter_2f, pg_p2_scatter_1f_key, pg_p2_scatter_1f_hll,
                                 pg_p2_scatter_o_key, pg_p2_scatter_o_raw, pg_p2_scatter_o_dict, pg_p2_scatter_ow_key, pg_p2_scatter_ow_raw, pg_p2_scatter_ow_dict,
                                 pg_p2_aggregate_1, pg_p2_aggregate_2, pg_p2_aggregate_3, pg_p2_aggregate_4,
                                 pg_p2_aggregate_1n, pg_p2_aggregate_2n, pg_p2_aggregate_3n, pg_p2_aggregate_4n, pg_p2_aggregate_1s, pg_p2_aggregate_2s, pg_p2_aggregate_1sg, pg_p2_aggregate_2sg, pg_p2_aggregate_1set, pg_p2_aggregate_1b};
      for (QueryKernel k : all)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      for (QueryKernel k : {pg_fast_i32range_s, pg_fast_i32range_st, pg_spec_none, pg_spec_scan, pg_spec_index})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      for (QueryKernel k : {pg_fast_dictrange_s_r_dma, pg_fast_dictrange_s_a_dma, pg_fast_dictrange_s_g_dma, pg_fast_dictrange_s_r, pg_fast_dictrange_st_r, pg_specd_none_r, pg_specd_scan_r, pg_specd_index_r, pg_fast_dictrange_s_a, pg_fast_dictrange_st_a, pg_specd_none_a, pg_specd_scan_a, pg_specd_index_a, pg_fast_dictrange_s_g, pg_fast_dictrange_st_g, pg_specd_none_g, pg_specd_scan_g, pg_specd_index_g})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      for (QueryKernel k : {pg_fast_dictrange_w_r, pg_fast_dictrange_wt_r, pg_specw_none_r, pg_specw_scan_r, pg_specw_index_r, pg_fast_dictrange_w_a, pg_fast_dictrange_wt_a, pg_specw_none_a, pg_specw_scan_a, pg_specw_index_a, pg_fast_dictrange_w_g, pg_fast_dictrange_wt_g, pg_specw_none_g, pg_specw_scan_g, pg_specw_index_g})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pg_nogroup_dl), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      for (QueryKernel k : {pg_mv_group_4, pg_mv_group_8, pg_mv_aggr_4, pg_mv_aggr_8})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      for (QueryKernel k : {pg_mv_query_f, pg_mv_query_l, pg_mv_query_g})   // 10.5 KB of static LDS (per-wavefront entry bitmaps): the planner's 144 KB still fit
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 12288);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pg_radix_scatter_packed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pg_radix_aggregate_packed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
      (void)hipGetLastError();   // a refused attribute must not surface as the "last error" of a later launch
      di.ready.store(1, std::memory_order_release);
    }
  }
  PG_HIP(hipSetDevice(ordinal));
  t_device = ordinal;
}

void device_init(int ordinal) {
  const int n = device_count_or_fail();
  if (ordinal < 0 || ordinal >= n) fail(PG_ERR_INVALID_ARGUMENT, "device ordinal %d out of range (0..%d)", ordinal, n - 1);
  use_device(ordinal);
  g_default_device.store(ordinal);
}
int default_device() {
  const int d = g_default_device.load();
  return d < 0 ? 0 : d;
}
static int num_cus() { return g_devices[t_device].num_cus; }
static size_t lds_per_cu() { return g_devices[t_device].lds_per_cu; }

static bool uses_fast_kernel(const CompiledPlan& P, int agg_mode) {
  const bool force_interpreter = knobs().force_interpreter;   // measurement knob
  if (force_interpreter || P.dev.mv) return false;   // multi-value plans: pg_mv_query_* (the interpreter's frame)
  const bool agg = agg_mode != PG_AGG_NONE;
  return P.fast_filter != -2 && (!agg || ((P.fast_agg || P.wide_agg) && agg_mode != PG_AGG_GLOBAL));
}

// pg_mv_group_* (pg_kernels_mvg.hip): GROUP BY one multi-value column, decided at plan time (PgQueryPlan::mvg)
static bool uses_mvg(const CompiledPlan& P, int agg_mode) { return P.dev.mv && P.dev.mvg && agg_mode == PG_AGG_LDS && !knobs().no_mvg; }
// pg_fast_i32range_fp (double-buffered raw-INT range scan over the whole segment, filter only: pg_kernels_scan.hip)
static bool uses_scan_kernel(const CompiledPlan& P, int agg_mode) {
  const bool no_scan = knobs().no_scan_pipe;   // measurement knob
  return !no_scan && agg_mode == PG_AGG_NONE && uses_fast_kernel(P, agg_mode) && P.fast_filter == 4 && P.dev.n_index_instr == 0 &&
         P.dev.fast_scan_pushed;
}
// pg_nogroup_s1 / _s2 (pg_kernels_scan.hip): no GROUP BY, no filter, integer accumulators over one or two raw INT columns — streamed with the
// accumulators in registers; returns the number of columns (0: another kernel)
static int uses_nogroup_stream(const CompiledPlan& P, int agg_mode) {
  const PgQueryPlan& D = P.dev;
  if (knobs().no_scan_pipe || agg_mode != PG_AGG_SINGLE || !uses_fast_kernel(P, agg_mode) || P.fast_filter != -1 || D.n_index_instr != 0 || D.tail_posting >= 0 ||
      D.n_group_cols != 0 || D.n_groups != 1 || D.n_aux != 0 || D.n_ops <= 0 || D.n_ops > PG_MAX_OPS)
    return 0;
  int a = -1, b = -1;
  for (int o = 0; o < D.n_ops; o++) {
    const PgAccOp& op = D.ops[o];
    if (op.fn == PG_ACC_COUNT && op.src < 0) continue;
    if (op.src < 0 || op.is_float != PG_ACCV_INT || op.limb != 0) return 0;
    const PgValueSrc& S = D.srcs[op.src];
    if (S.col_kind != PG_COL_RAW32 || S.val_type != PG_V_I32) return 0;
    if (op.src == a || op.src == b) continue;
    if (a < 0) a = op.src; else if (b < 0) b = op.src; else return 0;
  }
  return a < 0 ? 0 : (b < 0 ? 1 : 2);
}
// pg_nogroup_da / _dg (pg_kernels_scan.hip): the same over one dictionary-encoded INT column — decided at plan time (PgQueryPlan::nogroup_d)
static bool uses_nogroup_dict(const CompiledPlan& P, int agg_mode) {
  return P.dev.nogroup_d != 0 && agg_mode == PG_AGG_SINGLE && uses_fast_kernel(P, agg_mode) && !knobs().no_scan_pipe;
}
// pg_dictrange_fo (pg_kernels_scan.hip): filter only, [index program AND] one dictId-interval scan — decided at plan time (PgQueryPlan::dict_filter_only)
static bool uses_dict_filter_only(const CompiledPlan& P, int agg_mode) {
  return P.dev.dict_filter_only != 0 && agg_mode == PG_AGG_NONE && uses_fast_kernel(P, agg_mode) && !knobs().no_scan_pipe;
}
// pg_fast_dictrange_s family (pg_kernels_specd.hip): the loader / consumer frame over dictionary-encoded scan / value columns — decided at plan
// time (PgQueryPlan::specd), whatever the filter lets through
static bool uses_specd(const CompiledPlan& P, int agg_mode) {
  return P.dev.specd && (agg_mode == PG_AGG_LDS || (agg_mode == PG_AGG_SINGLE && P.dev.n_group_cols == 0)) && uses_fast_kernel(P, agg_mode) && P.fast_agg && !P.wide_agg && !knobs().no_specd;
}
static size_t specd_stage_bytes(const CompiledPlan& P) {
  return ((size_t)pg_specd_stage_bytes(P.dev.specd_sbits, P.dev.specd_vbits, P.dev.n_group_cols > 0 ? P.dev.gcols[0].bits : 0, P.dev.n_group_cols > 1 ? P.dev.gcols[1].bits : 0, P.dev.specd_dma ? 2 : 1) + 15) & ~(size_t)15;
}
// ... in the shared-stage frame (PgQueryPlan::specd == 2, pg_kernels_specw.hip): two stage buffers + one selection list per wavefront
static bool uses_specw(const CompiledPlan& P, int agg_mode) { return uses_specd(P, agg_mode) && P.dev.specd == 2; }
static size_t specw_stage_bytes(const CompiledPlan& P) {
  int n_bm = 0;
  if (P.dev.pipe_has_index) {
    n_bm = 8;
    while (n_bm > 1 && P.dev.dense_ptr[n_bm - 1] == P.dev.dense_ptr[0] && P.dev.dense_group[n_bm - 1] == P.dev.dense_group[0]) n_bm--;
  }
  if (P.dev.pipe_tail != nullptr) n_bm++;
  return 2 * (size_t)pg_specw_stage_bytes(P.dev.specd_sbits, P.dev.specd_vbits, P.dev.gcols[0].bits, P.dev.n_group_cols > 1 ? P.dev.gcols[1].bits : 0, n_bm) + (size_t)pg_specw_list_bytes() + 16;
}
// pg_fast_i32range_p (software-pipelined headline shape, pg_kernels_pipe.hip): its own workgroup size
static bool uses_pipe_general(const CompiledPlan& P, int agg_mode) {   // pg_pipe_*: the pipeline's other filter shapes
  const bool no_pipe = knobs().no_pipe;   // measurement knob
  return !no_pipe && uses_fast_kernel(P, agg_mode) && agg_mode == PG_AGG_LDS && !P.wide_agg && P.fast_agg && P.dev.pipe_general;
}
static bool uses_pipe_wide(const CompiledPlan& P, int agg_mode) {   // pg_pipe_w_*: raw LONG values / group columns of 9 .. 16 bits
  const bool no_pipe = knobs().no_pipe;   // measurement knob
  return !no_pipe && uses_fast_kernel(P, agg_mode) && (agg_mode == PG_AGG_LDS || agg_mode == PG_AGG_SINGLE) && P.wide_agg && P.dev.pipe_wide;
}
static bool uses_pipe_kernel(const CompiledPlan& P, int agg_mode) {
  const bool no_pipe = knobs().no_pipe || knobs().no_dense_fused;   // measurement knobs
  if (uses_pipe_general(P, agg_mode) || uses_pipe_wide(P, agg_mode)) return true;
  return !no_pipe && uses_fast_kernel(P, agg_mode) && agg_mode == PG_AGG_LDS && !P.wide_agg && P.fast_agg && P.fast_filter == 4 &&
         P.dev.dense_fused && P.dev.pipe_fit;
}

// pg_fast_i32range_s (pg_kernels_spec.hip): the same plans as pg_fast_i32range_p with loader / consumer wavefronts — PG_WAVE_SPECIALISED only
static size_t spec_stage_bytes(const CompiledPlan& P) {
  return ((size_t)pg_spec_stage_bytes(P.dev.gcols[0].bits, P.dev.n_group_cols > 1 ? P.dev.gcols[1].bits : 0) + 15) & ~(size_t)15;
}
// Which plans: pg_fast_i32range_p's (index AND scan) and the pipeline's general shapes without a tail bitmap or a second scan (none / scan /
// index).  It streams every column whole (84-92 % of 8 TB/s whatever the filter); the pipelined kernels request only the quads that hold a
// candidate / a match and win where the filter is selective — so the candidate rate the plan's LAST execution counted decides (a plan's first
// execution takes the pipelined kernel): >= 15 % candidates behind the index.  The general shapes: only when forced (below).
static int spec_shape(const CompiledPlan& P, int agg_mode) {   // 0: no; 1 index + scan; 2 none; 3 scan; 4 index; 5 index + scan + upsert snapshot
  if (knobs().no_wave_specialised || !uses_pipe_kernel(P, agg_mode) || uses_pipe_wide(P, agg_mode)) return 0;
  if (P.dev.n_group_cols < 1 || P.dev.n_group_cols > 2 || P.lds_bytes + 256 + 2 * spec_stage_bytes(P) + 512 * (size_t)P.dev.n_ops > (size_t)160 * 1024 - 8192) return 0;   // (the dynamic-LDS limit device_init asks for)
  const bool force = knobs().wave_specialised;
  const int cand = P.observed_candidate_permille.load(std::memory_order_relaxed), match = P.observed_match_permille.load(std::memory_order_relaxed);
  const int min_cand = knobs().wave_specialised_min_permille;
  if (uses_pipe_general(P, agg_mode)) {
    const bool idx = P.dev.pipe_has_index != 0, scan = P.dev.pipe_has_scan != 0;
    if (P.dev.pipe_vscan >= 0) return 0;
    if (idx && scan && P.dev.pipe_tail != nullptr) return force || cand >= min_cand ? 5 : 0;   // the headline shape behind an upsert snapshot (pg_pipe_index_scan_tail)
    if (P.dev.pipe_tail != nullptr || (idx && scan)) return 0;
    // Measured over 10^9 docs (profiles/r05_wave_specialised.txt): no filter 1.096 ms against pg_pipe_none's 0.926, a lone scan 1.394 = 1.394, index
    // only 1.017 against 0.918 — these shapes stream 5 - 9 bytes per doc and EVERY candidate matches, so the eight consumers' LDS atomics, not
    // the stream, are the long path.  Only when forced (tests, measurements).
    (void)match;
    if (!force) return 0;
    return !idx && !scan ? 2 : (scan ? 3 : 4);
  }
  return force || cand >= min_cand ? 1 : 0;
}
// One decision per execution: the rates move under concurrent executions of the same plan, and the launch shape (12 wavefronts, stage buffers)
// and the kernel must agree.  execute_query_plain pins it before it sizes the launch.
struct SpecPin { const CompiledPlan* plan = nullptr; int agg_mode = 0, shape = 0; };
static thread_local SpecPin t_spec_pin;
static int pinned_spec_shape(const CompiledPlan& P, int agg_mode) {
  if (t_spec_pin.plan == &P && t_spec_pin.agg_mode == agg_mode) return t_spec_pin.shape;
  return 0;   // not pinned: the pipelined kernels
}
static bool uses_spec_kernel(const CompiledPlan& P, int agg_mode) { return pinned_spec_shape(P, agg_mode) != 0; }

extern "C" void pg_trim_launch(const PgTrimArgs* args, int grid, hipStream_t stream);
extern "C" void pg_trim_launch_keys(const PgTrimArgs* args, int grid, hipStream_t stream);
extern "C" void pg_trim_launch_select(const PgTrimArgs* args, int grid, hipStream_t stream);
typedef void (*QueryKernel)(const PgQueryPlan);
static QueryKernel select_kernel(const CompiledPlan& P, int agg_mode, const char** name) {
  const bool agg = agg_mode != PG_AGG_NONE;
  if (uses_dict_filter_only(P, agg_mode)) { *name = "pg_dictrange_fo"; return pg_dictrange_fo; }
  if (uses_nogroup_dict(P, agg_mode)) {
    if (P.dev.nogroup_d == 1) { *name = "pg_nogroup_da"; return pg_nogroup_da; }
    *name = P.dev.nogroup_lds_card > 0 ? "pg_nogroup_dl" : "pg_nogroup_dg";
    return P.dev.nogroup_lds_card > 0 ? pg_nogroup_dl : pg_nogroup_dg;
  }
  if (const int ns = uses_nogroup_stream(P, agg_mode)) { *name = ns == 1 ? "pg_nogroup_s1" : "pg_nogroup_s2"; return ns == 1 ? pg_nogroup_s1 : pg_nogroup_s2; }
  if (uses_fast_kernel(P, agg_mode)) {
    if (agg && uses_pipe_wide(P, agg_mode)) {
      // one kernel per value width (no value column / raw INT / raw LONG) and filter shape
      static const struct { const char* name; QueryKernel fn; } kWide[4][4] = {
          {{"pg_pipe_w0_none", pg_pipe_w0_none}, {"pg_pipe_w0_index", pg_pipe_w0_index}, {"pg_pipe_w0_scan", pg_pipe_w0_scan}, {"pg_pipe_w0_index_scan", pg_pipe_w0_index_scan}},
          {{"pg_pipe_w32_none", pg_pipe_w32_none}, {"pg_pipe_w32_index", pg_pipe_w32_index}, {"pg_pipe_w32_scan", pg_pipe_w32_scan}, {"pg_pipe_w32_index_scan", pg_pipe_w32_index_scan}},
          {{"pg_pipe_w64_none", pg_pipe_w64_none}, {"pg_pipe_w64_index", pg_pipe_w64_index}, {"pg_pipe_w64_scan", pg_pipe_w64_scan}, {"pg_pipe_w64_index_scan", pg_pipe_w64_index_scan}},
          {{"pg_pipe_wd_none", pg_pipe_wd_none}, {"pg_pipe_wd_index", pg_pipe_wd_index}, {"pg_pipe_wd_scan", pg_pipe_wd_scan}, {"pg_pipe_wd_index_scan", pg_pipe_wd_index_scan}}};
      const int vw = P.dev.pipe_src >= 0 ? P.dev.pipe_wide : 0;
      const int shape = (P.dev.pipe_has_scan ? 2 : 0) + (P.dev.pipe_has_index ? 1 : 0);
      *name = kWide[vw][shape].name;
      return kWide[vw][shape].fn;
    }
    if (agg && P.wide_agg) {
      if (P.fast_filter == -1) { *name = P.digit_ops ? "pg_fast_none_wd" : "pg_fast_none_w"; return P.digit_ops ? pg_fast_none_wd : pg_fast_none_w; }
      *name = P.digit_ops ? "pg_fast_multi_wd" : "pg_fast_multi_w";
      return P.digit_ops ? pg_fast_multi_wd : pg_fast_multi_w;
    }
    if (uses_specd(P, agg_mode)) {
      // one kernel per value kind (raw INT / arithmetic dictionary / gathered dictionary) and filter shape
      static const struct { const char* name; QueryKernel fn; } kSpecd[3][5] = {
          {{"pg_specd_none_r", pg_specd_none_r}, {"pg_specd_index_r", pg_specd_index_r}, {"pg_specd_scan_r", pg_specd_scan_r}, {"pg_fast_dictrange_s_r", pg_fast_dictrange_s_r}, {"pg_fast_dictrange_st_r", pg_fast_dictrange_st_r}},
          {{"pg_specd_none_a", pg_specd_none_a}, {"pg_specd_index_a", pg_specd_index_a}, {"pg_specd_scan_a", pg_specd_scan_a}, {"pg_fast_dictrange_s_a", pg_fast_dictrange_s_a}, {"pg_fast_dictrange_st_a", pg_fast_dictrange_st_a}},
          {{"pg_specd_none_g", pg_specd_none_g}, {"pg_specd_index_g", pg_specd_index_g}, {"pg_specd_scan_g", pg_specd_scan_g}, {"pg_fast_dictrange_s_g", pg_fast_dictrange_s_g}, {"pg_fast_dictrange_st_g", pg_fast_dictrange_st_g}}};
      const int shape = P.dev.pipe_tail != nullptr ? 4 : (P.dev.pipe_has_scan ? 2 : 0) + (P.dev.pipe_has_index ? 1 : 0);
      if (P.dev.specd == 2) {
        static const struct { const char* name; QueryKernel fn; } kSpecw[3][5] = {
            {{"pg_specw_none_r", pg_specw_none_r}, {"pg_specw_index_r", pg_specw_index_r}, {"pg_specw_scan_r", pg_specw_scan_r}, {"pg_fast_dictrange_w_r", pg_fast_dictrange_w_r}, {"pg_fast_dictrange_wt_r", pg_fast_dictrange_wt_r}},
            {{"pg_specw_none_a", pg_specw_none_a}, {"pg_specw_index_a", pg_specw_index_a}, {"pg_specw_scan_a", pg_specw_scan_a}, {"pg_fast_dictrange_w_a", pg_fast_dictrange_w_a}, {"pg_fast_dictrange_wt_a", pg_fast_dictrange_wt_a}},
            {{"pg_specw_none_g", pg_specw_none_g}, {"pg_specw_index_g", pg_specw_index_g}, {"pg_specw_scan_g", pg_specw_scan_g}, {"pg_fast_dictrange_w_g", pg_fast_dictrange_w_g}, {"pg_fast_dictrange_wt_g", pg_fast_dictrange_wt_g}}};
        *name = kSpecw[P.dev.specd_vkind - 1][shape].name;
        return kSpecw[P.dev.specd_vkind - 1][shape].fn;
      }
      if (P.dev.specd_dma && shape == 3) {   // the headline shape's columns by LDS-DMA (two column areas per strip: the planner found room)
        static const struct { const char* name; QueryKernel fn; } kDma[3] = {
            {"pg_fast_dictrange_s_r_dma", pg_fast_dictrange_s_r_dma}, {"pg_fast_dictrange_s_a_dma", pg_fast_dictrange_s_a_dma}, {"pg_fast_dictrange_s_g_dma", pg_fast_dictrange_s_g_dma}};
        *name = kDma[P.dev.specd_vkind - 1].name;
        return kDma[P.dev.specd_vkind - 1].fn;
      }
      *name = kSpecd[P.dev.specd_vkind - 1][shape].name;
      return kSpecd[P.dev.specd_vkind - 1][shape].fn;
    }
    const bool no_dense = knobs().no_dense_fused;   // measurement knob
    switch (pinned_spec_shape(P, agg_mode)) {
      case 1: *name = "pg_fast_i32range_s"; return pg_fast_i32range_s;
      case 2: *name = "pg_spec_none"; return pg_spec_none;
      case 3: *name = "pg_spec_scan"; return pg_spec_scan;
      case 4: *name = "pg_spec_index"; return pg_spec_index;
      case 5: *name = "pg_fast_i32range_st"; return pg_fast_i32range_st;
      default: break;
    }
    if (uses_pipe_general(P, agg_mode)) {
      const bool idx = P.dev.pipe_has_index != 0, scan = P.dev.pipe_has_scan != 0, tail = P.dev.pipe_tail != nullptr;
      if (scan && P.dev.pipe_vscan >= 0) {
        *name = idx ? "pg_pipe_index_scan_vscan" : "pg_pipe_scan_vscan";
        return idx ? pg_pipe_index_scan_vscan : pg_pipe_scan_vscan;
      }
      if (scan) {
        if (idx) { *name = "pg_pipe_index_scan_tail"; return pg_pipe_index_scan_tail; }   // (index + scan without a tail is pg_fast_i32range_p)
        *name = tail ? "pg_pipe_scan_tail" : "pg_pipe_scan";
        return tail ? pg_pipe_scan_tail : pg_pipe_scan;
      }
      if (idx) {
        int n_ptr = 0;   // distinct dense posting pointers (the planner pads the eight slots with repeats of slot 0)
        for (int j = 0; j < 8; j++) if (j == 0 || P.dev.dense_ptr[j] != P.dev.dense_ptr[0] || P.dev.dense_group[j] != P.dev.dense_group[0]) n_ptr = j + 1;
        if (n_ptr <= 2) { *name = tail ? "pg_pipe_index2_tail" : "pg_pipe_index2"; return tail ? pg_pipe_index2_tail : pg_pipe_index2; }
        *name = tail ? "pg_pipe_index_tail" : "pg_pipe_index";
        return tail ? pg_pipe_index_tail : pg_pipe_index;
      }
      *name = tail ? "pg_pipe_tail" : "pg_pipe_none";
      return tail ? pg_pipe_tail : pg_pipe_none;
    }
    if (uses_pipe_kernel(P, agg_mode)) { *name = "pg_fast_i32range_p"; return pg_fast_i32range_p; }
    if (uses_scan_kernel(P, agg_mode)) { *name = "pg_fast_i32range_fp"; return pg_fast_i32range_fp; }
    if (agg && P.fast_filter == 4 && P.dev.dense_fused && !no_dense && P.fast_agg && agg_mode == PG_AGG_LDS) { *name = "pg_fast_i32range_d"; return pg_fast_i32range_d; }
    switch (P.fast_filter) {
      case -1: *name = agg ? "pg_fast_none_a" : "pg_fast_none_f"; return agg ? pg_fast_none_a : pg_fast_none_f;
      case 4: *name = agg ? "pg_fast_i32range_a" : "pg_fast_i32range_f"; return agg ? pg_fast_i32range_a : pg_fast_i32range_f;
      case 0: *name = agg ? "pg_fast_dictrange_a" : "pg_fast_dictrange_f"; return agg ? pg_fast_dictrange_a : pg_fast_dictrange_f;
      case 2: *name = agg ? "pg_fast_dictlut_a" : "pg_fast_dictlut_f"; return agg ? pg_fast_dictlut_a : pg_fast_dictlut_f;
      case 100: *name = agg ? "pg_fast_multi_a" : "pg_fast_multi_f"; return agg ? pg_fast_multi_a : pg_fast_multi_f;
      default: break;
    }
  }
  if (P.dev.mv) {   // a multi-value column in the filter, the group key or an aggregation (pg_kernels_mv.hip)
    if (uses_mvg(P, agg_mode)) {   // ... GROUP BY one multi-value column, no filter: its own kernel (pg_kernels_mvg.hip)
      if (P.dev.mvg >= 16) {   // the *MV functions over one multi-value column, single-value keys
        *name = P.dev.mvg == 20 ? "pg_mv_aggr_4" : "pg_mv_aggr_8";
        return P.dev.mvg == 20 ? pg_mv_aggr_4 : pg_mv_aggr_8;
      }
      *name = P.dev.mvg == 4 ? "pg_mv_group_4" : "pg_mv_group_8";
      return P.dev.mvg == 4 ? pg_mv_group_4 : pg_mv_group_8;
    }
    if (agg_mode == PG_AGG_NONE) { *name = "pg_mv_query_f"; return pg_mv_query_f; }
    if (agg_mode == PG_AGG_GLOBAL) { *name = "pg_mv_query_g"; return pg_mv_query_g; }
    *name = "pg_mv_query_l";
    return pg_mv_query_l;
  }
  if (agg_mode == PG_AGG_NONE) { *name = "pg_generic_query_f"; return pg_generic_query_f; }
  if (agg_mode == PG_AGG_GLOBAL) { *name = P.digit_ops ? "pg_generic_query_gd" : "pg_generic_query_g"; return P.digit_ops ? pg_generic_query_gd : pg_generic_query_g; }
  *name = P.digit_ops ? "pg_generic_query_ld" : "pg_generic_query_l";
  return P.digit_ops ? pg_generic_query_ld : pg_generic_query_l;
}

struct ThreadCtx {
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  DeviceBuffer stats, partials, final_table, tile_counts, aux;
  DeviceBuffer words, radix_hist, radix_start, radix_tuples, hash_count, hash_keys, hash_acc;   // PG_AGG_RADIX work areas
  DeviceBuffer aux_summary;   // PG_QUERY_FLAG_FINAL_DISTINCT: [n_aux][G] final values
  DeviceBuffer fs_leaves, fs_arena;   // exact numEntriesScannedInFilter on the device: the leaves' match bitmaps, scratch (pg_filter_stats.cpp)
  DeviceBuffer trim_keys, trim_ctrl, trim_out;   // segment-level group trim on the device: [G] keys, counters, the compact block
  size_t aux_clean_bytes = 0;    // the first bytes of `aux` are zero (pg_finish_fused_kernel re-zeroes the states it folds): the next query's fill is skipped
  const void* aux_clean_ptr = nullptr;
  DeviceBuffer hll_small[17];  // per log2m: round(m * ln(m / zeros)), zeros = 0 .. m
  double hll_alpha_mm[17] = {0};
  DeviceBuffer p2_meta, p2_list, p2_ctrl;   // partition pipeline v2: chunk records, the same grouped by bucket, counters (PG_P2_CTRL_*)
  DeviceBuffer oct_floor, oct_counts, oct_stream, oct_cursor;   // pruned-offer passes (pg_kernels_oct.hip)
  uint32_t* p2_ctrl_host = nullptr;   // page-locked copy of p2_ctrl
  bool stats_dirty = true;      // the stats counters may be non-zero (first use, or a query that failed midway)
  void* pinned = nullptr;       // page-locked staging for the result copy (pageable copies are staged synchronously)
  size_t pinned_size = 0;
  void* pin(size_t n) {
    if (pinned_size < n) {
      if (pinned) (void)hipHostFree(pinned);
      pinned = nullptr;
      pinned_size = 0;
      PG_HIP(hipHostMalloc(&pinned, n + n / 4 + 4096, hipHostMallocDefault));
      pinned_size = n + n / 4 + 4096;
    }
    return pinned;
  }
  ~ThreadCtx() {
    if (pinned) (void)hipHostFree(pinned);
    if (p2_ctrl_host) (void)hipHostFree(p2_ctrl_host);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
  void ensure() {
    if (!stream) {
      PG_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      for (auto& e : ev) PG_HIP(hipEventCreate(&e));
      stats.alloc(PG_MAX_STATS * 8, true);
    }
  }
  static void grow(DeviceBuffer& b, size_t n) { if (b.size < n) b.alloc(n + n / 4); }
};
struct ThreadCtxSet {
  std::unique_ptr<ThreadCtx> per_device[PG_MAX_DEVICES];
  ~ThreadCtxSet() {
    for (int d = 0; d < PG_MAX_DEVICES; d++)
      if (per_device[d]) { (void)hipSetDevice(d); per_device[d].reset(); }
  }
};
static thread_local ThreadCtxSet t_ctxs;
// the calling thread's context on `device`, with that device made current
static ThreadCtx& ctx_on(int device) {
  use_device(device);
  auto& slot = t_ctxs.per_device[device];
  if (!slot) slot = std::make_unique<ThreadCtx>();
  slot->ensure();
  return *slot;
}
hipStream_t thread_stream(int device) { return ctx_on(device).stream; }

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

std::shared_ptr<CompiledPlan> get_plan(Segment& seg, const pg_filter_node* filter, const pg_query* q, int32_t flags) {
  std::string sig = query_signature(filter, q, flags);
  // compilation stays under the segment's lock: it fills per-column caches (HyperLogLog look-up tables) and uploads leaves
  std::lock_guard<std::mutex> g(seg.mu);
  static std::atomic<uint64_t> clock{0};
  auto it = seg.plan_cache.find(sig);
  if (it != seg.plan_cache.end()) {
    it->second->last_used.store(clock.fetch_add(1, std::memory_order_relaxed) + 1, std::memory_order_relaxed);
    return it->second;
  }
  auto plan = compile_plan(seg, filter, q, flags);
  if (seg.plan_cache.size() >= 256) {
    // the 64 plans that were used longest ago go (a wholesale clear() also forgot the candidate rates the kernels counted: VERDICT r5 #10)
    std::vector<std::pair<uint64_t, const std::string*>> age;
    age.reserve(seg.plan_cache.size());
    for (auto& kv : seg.plan_cache) age.emplace_back(kv.second->last_used.load(std::memory_order_relaxed), &kv.first);
    std::nth_element(age.begin(), age.begin() + 64, age.end());
    std::vector<std::string> victims;
    for (size_t i = 0; i < 64; i++) victims.push_back(*age[i].second);
    for (auto& v : victims) seg.plan_cache.erase(v);
  }
  plan->last_used.store(clock.fetch_add(1, std::memory_order_relaxed) + 1, std::memory_order_relaxed);
  seg.plan_cache[sig] = plan;
  return plan;
}

struct LaunchShape { int grid; int block; size_t lds; };
static LaunchShape launch_shape(const CompiledPlan& P, int n_wtiles, int agg_mode) {
  size_t lds = P.lds_bytes + 64;
  if (P.dev.agg_mode == PG_AGG_LDS_PART && agg_mode == PG_AGG_LDS_PART) {
    // range-partitioned aggregation: one workgroup per CU, 8 x per_xcd of them with per_xcd a multiple of the range count
    int per_xcd = std::max(num_cus() / 8, 1);
    const int block = uses_fast_kernel(P, agg_mode) ? PG_BLOCK : PG_GENERIC_BLOCK;
    // (a small doc space — a star-tree's pre-aggregated docs — does not need every CU: a workgroup without tiles still fills and flushes its table)
    const int chunks = (n_wtiles + block / 64 - 1) / (block / 64);
    const bool no_clamp = knobs().no_part_grid_clamp;   // measurement knob
    int per_range = std::max(per_xcd / P.dev.n_parts, 1);
    if (!no_clamp) per_range = std::max(1, std::min(per_range, (chunks + 7) / 8));
    return {8 * per_range * P.dev.n_parts, block, lds};
  }
  if (uses_scan_kernel(P, agg_mode) || uses_nogroup_stream(P, agg_mode) || uses_nogroup_dict(P, agg_mode) || uses_dict_filter_only(P, agg_mode)) {
    // tuning knob; the dictId stream of pg_nogroup_d* is 1-3 bytes per doc — 5 KB per tile at 20 bits: two workgroups per CU keep as many bytes in
    // flight as one does over a raw column (0.143 -> 0.102 ms per 2 x 10^8 docs; three to five: 0.105-0.112, profiles/r06_nogroup_stream.txt)
    const size_t dict_lds = uses_nogroup_dict(P, agg_mode) && P.dev.nogroup_d == 2 ? (size_t)P.dev.nogroup_lds_card * 4 : 0;   // pg_nogroup_dl: the dictionary's copy
    // pg_dictrange_fo waits for its posting dwords once per tile (the index program is not prefetched): four workgroups per CU hide that
    // (two: 0.181 ms per 2 x 10^8 docs, four: 0.139, six: 0.151)
    const int wgs_per_cu = knobs().scan_wgs_per_cu * (uses_dict_filter_only(P, agg_mode) ? 4 : ((uses_nogroup_dict(P, agg_mode) && 2 * (dict_lds + 4096) <= lds_per_cu()) ? 2 : 1));
    const int waves = pg_scan_waves_per_block;
    int grid = std::min((n_wtiles + waves - 1) / waves, num_cus() * std::max(wgs_per_cu, 1));
    return {std::max(grid, 1), waves * 64, dict_lds};
  }
  if (uses_mvg(P, agg_mode))   // one 16-wavefront workgroup per CU; the table and a trash slot per lane and accumulator
    return {std::max(1, std::min((n_wtiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus())), PG_BLOCK, lds + 64 + 512 * (size_t)P.dev.n_ops + (size_t)P.dev.mvg_dict_card * 4};
  if (uses_specw(P, agg_mode)) {   // one workgroup per CU walking stages (waves x 512 docs) b, b + grid, ...; table + two stage buffers + the selection lists in LDS
    const int waves = pg_specw_waves_per_block;
    const int64_t stage_docs = (int64_t)waves * 512;
    const int n_stages = (int)(((int64_t)P.dev.num_docs + stage_docs - 1) / stage_docs);
    return {std::max(1, std::min(n_stages, num_cus() * std::max(1, 16 / waves))), waves * 64, lds + 64 + specw_stage_bytes(P) + 512 * (size_t)P.dev.n_ops};
  }
  if (uses_specd(P, agg_mode)) {   // behind the table and its trash slots a private strip of LDS per wavefront; two workgroups per CU where their LDS fits
    const int waves = pg_specd_waves_per_block;
    const size_t need = lds + 64 + specd_stage_bytes(P) + 512 * (size_t)P.dev.n_ops;
    const int per_cu = knobs().specd_wgs_per_cu > 0 ? ((size_t)knobs().specd_wgs_per_cu * (need + 1024) <= lds_per_cu() ? knobs().specd_wgs_per_cu : 1) : 1;
    return {std::max(1, std::min((n_wtiles + waves - 1) / waves, num_cus() * per_cu)), waves * 64, need};
  }
  if (uses_spec_kernel(P, agg_mode))   // one 12-wavefront workgroup per CU walking tiles b, b + grid, ...; table + two stage buffers in LDS
    return {std::max(1, std::min(n_wtiles, num_cus())), pg_spec_waves_per_block * 64, lds + 64 + 2 * spec_stage_bytes(P) + 512 * (size_t)P.dev.n_ops};   // (+ 64 trash slots per accumulator)
  if (uses_pipe_kernel(P, agg_mode)) {
    const int wgs_per_cu = knobs().pipe_wgs_per_cu;   // tuning knob
    const int per_cu = ((size_t)wgs_per_cu * (lds + 4096) <= lds_per_cu()) ? wgs_per_cu : 1;
    const int waves = pg_pipe_waves_per_block;
    int grid = std::min((n_wtiles + waves - 1) / waves, num_cus() * per_cu);
    return {std::max(grid, 1), waves * 64, lds};
  }
  if (uses_fast_kernel(P, agg_mode)) {
    // one 16-wave workgroup per CU; fewer when the segment has fewer wave tiles than that
    const int wgs_per_cu = knobs().wgs_per_cu;   // tuning knob
    const int per_cu = ((size_t)wgs_per_cu * (lds + 4096) <= lds_per_cu()) ? wgs_per_cu : 1;
    int grid = std::min((n_wtiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus() * per_cu);
    return {std::max(grid, 1), PG_BLOCK, lds};
  }
  // interpreter kernel: 8-wave workgroups, two per CU when both LDS tables fit
  const int waves = PG_GENERIC_BLOCK / 64;
  const int per_cu = (2 * (lds + 4096) <= lds_per_cu()) ? 2 : 1;
  int grid = std::min((n_wtiles + waves - 1) / waves, num_cus() * per_cu);
  return {std::max(grid, 1), PG_GENERIC_BLOCK, lds};
}

static void fill_stats(pg_exec_stats& st, const CompiledPlan& P, int64_t full_scan_entries, int64_t total_docs, const uint64_t* stats_host) {
  st.num_docs_scanned = (int64_t)stats_host[0];
  int64_t in_filter = full_scan_entries;
  for (int i = 1; i < P.n_stat_slots; i++) in_filter += (int64_t)stats_host[i];
  st.num_entries_scanned_in_filter = in_filter;
  st.num_entries_scanned_post_filter = st.num_docs_scanned * P.n_projected_columns;
  st.num_total_docs = total_docs;
  st.stats_exact = P.stats_exact ? 1 : 0;
  st.algorithmic_bytes = P.algorithmic_bytes;
}

static double order_key_to_double(int64_t k) {
  int64_t b = k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFLL);
  double d;
  memcpy(&d, &b, 8);
  return d;
}

// numEntriesScannedInFilter for plans whose count the kernels' counters do not give (CompiledPlan::stats_exact == false): the match
// bitmap of every Scan / Inverted leaf comes from a filter launch of its own, the reference's iterator automaton runs over the
// bitmaps on the host (pg_filter_stats.cpp).
// shapes whose automaton decomposes into tiles are counted where the bitmaps are (pg_filter_stats_tiles.h); the others — NOT or a compound
// child under an AND, multi-value scans — take the host walk
static bool stats_counted_on_device(const CompiledPlan& P) {
  if (knobs().filter_stats_host || P.space_docs <= 0 || !P.root_op) return false;
  int v = P.stats_on_device.load(std::memory_order_relaxed);
  if (v < 0) {
    v = filter_stats_on_device(*P.root_op, P.space_docs) ? 1 : 0;
    P.stats_on_device.store(v, std::memory_order_relaxed);
  }
  return v == 1;
}
// do this plan's statistics get the exact count by default (PG_QUERY_FLAG_EXACT_FILTER_STATS asks for it at any size)
static bool exact_stats_by_default(const CompiledPlan& P) {
  return (int64_t)P.space_docs <= knobs().exact_stats_max_docs || ((int64_t)P.space_docs <= knobs().exact_stats_device_max_docs && stats_counted_on_device(P));
}

static int64_t exact_entries_scanned(CompiledPlan& P, ThreadCtx& ctx, const CancelToken* cancel, int32_t* path_out = nullptr) {
  StatLeafBits bits;
  const int32_t n_docs = P.space_docs;
  const bool on_device = stats_counted_on_device(P);
  StatLeafWords dev_bits;
  size_t slot_words = 0;
  if (on_device) {
    for (auto& lf : P.stat_leaves) slot_words = std::max(slot_words, (size_t)std::max(lf.second->dev.n_tiles, 1) * PG_TILE_WORDS);
    ThreadCtx::grow(ctx.fs_leaves, std::max<size_t>(P.stat_leaves.size(), 1) * slot_words * 8);
  }
  size_t slot = 0;
  for (auto& lf : P.stat_leaves) {
    if (cancel && cancel->requested.load(std::memory_order_acquire)) fail(PG_ERR_CANCELLED, "query cancelled (EarlyTerminationException)");
    CompiledPlan& L = *lf.second;
    PgQueryPlan D = L.dev;
    const size_t dev_words = (size_t)std::max(D.n_tiles, 1) * PG_TILE_WORDS;
    if (!on_device) ThreadCtx::grow(ctx.words, dev_words * 8);
    PG_HIP(hipMemsetAsync(ctx.stats.ptr, 0, PG_MAX_STATS * 8, ctx.stream));
    ctx.stats_dirty = true;
    D.stats = ctx.stats.as<unsigned long long>();
    D.out_words = on_device ? ctx.fs_leaves.as<uint64_t>() + slot++ * slot_words : ctx.words.as<uint64_t>();
    D.agg_mode = PG_AGG_NONE;
    HostBits hb;
    if (!on_device) hb.resize_for(n_docs);
    if (n_docs > 0) {
      const LaunchShape shape = launch_shape(L, D.n_wtiles, PG_AGG_NONE);
      const char* kname = "";
      hipLaunchKernelGGL(select_kernel(L, PG_AGG_NONE, &kname), dim3(shape.grid), dim3(shape.block), shape.lds, ctx.stream, D);
      PG_HIP(hipGetLastError());
      if (on_device) { dev_bits.emplace(lf.first, D.out_words); continue; }
      PG_HIP(hipMemcpyAsync(hb.w.data(), ctx.words.ptr, (size_t)(((int64_t)n_docs + 63) / 64) * 8, hipMemcpyDeviceToHost, ctx.stream));
      PG_HIP(hipStreamSynchronize(ctx.stream));
    }
    bits.emplace(lf.first, std::move(hb));
  }
  if (on_device) {
    const int64_t n = entries_scanned_on_device(*P.root_op, dev_bits, n_docs, ctx.fs_arena, ctx.stream);
    if (path_out) *path_out = 2;
    return n;
  }
  if (path_out) *path_out = 1;
  return emulate_entries_scanned_in_filter(*P.root_op, bits, n_docs);
}

// ---- result assembly: dense accumulator table (+ statistics, + DISTINCTCOUNT / HLL regions) -> groups and intermediates --------
struct HostTable {
  std::vector<int64_t> table;            // [n_ops][G] (G = groups of the compact table for hashed key spaces)
  uint64_t stats[PG_MAX_STATS] = {0};
  uint8_t* aux = nullptr;                // merged auxiliary regions (replicas are folded in place)
  const uint64_t* aux_summary = nullptr; // PG_QUERY_FLAG_FINAL_DISTINCT: [n_aux][G] final values instead of `aux`
  std::shared_ptr<PinnedBlock> aux_block; // set when `aux` lies in a block the result may keep (see AggResult::hll_block)
  bool hashed = false;
  int64_t hash_groups = 0;
  std::vector<int64_t> hash_keys;        // PG_AGG_RADIX_HASH: raw key of every group of the compact table
  int64_t full_scan_entries = 0;
  int64_t total_docs = 0;
  // numGroupsLimit decided by a prefix pass (execute_limit_by_prefix): the first matching docId of every group that occurs in the prefix
  // (INT64_MAX elsewhere) and the limit itself — the plan that filled `table` carries neither
  const int64_t* admit_first = nullptr;
  int32_t admit_limit = 0;
  int64_t groups_found = -1;             // >= 0: the table was trimmed on the device; the groups the segment held before that
  const UnionKeys* keys = nullptr;       // the table's key space is a union of several segments' dictionaries (DeviceTable::keys)
};
// What execute_query_impl does beside the plain query: stop after `doc_limit` docs of the doc space; hand the raw table over instead of
// assembling groups; trim to numGroupsLimit by another pass's first docIds.
struct LimitAdmission { const int64_t* first; int32_t limit; };
struct ExecOptions {
  int64_t doc_limit = 0;
  HostTable* raw_out = nullptr;
  const LimitAdmission* admit = nullptr;
  // the same admission without the tables leaving HBM (dense tables without auxiliary state): the prefix pass turns its first-docId row into
  // selection keys on the device and reports only how many groups it found; the pass over the segment then selects the `limit` groups with
  // the smallest keys and copies their rows alone (pg_kernels_trim.hip; 10^6 groups: 2 x 8 MB + 8 MB of tables -> 2.4 MB)
  int64_t* prefix_groups_out = nullptr;
  int32_t admit_on_device = 0;   // numGroupsLimit to apply with the keys the prefix pass left in the thread's context
};
static void assemble_result(Result& res_out, const CompiledPlan& P, int32_t n_group_by, int32_t n_aggregations, HostTable& H);
static void hll_small_range_table(int log2m, std::vector<long long>& t, double& alpha_mm);

static void check_cancel(const CancelToken* c, ThreadCtx* ctx) {
  if (c && c->requested.load(std::memory_order_acquire)) {
    if (ctx && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); ctx->stats_dirty = true; }   // let what was launched finish
    fail(PG_ERR_CANCELLED, "query cancelled (EarlyTerminationException)");
  }
}
// Waits for the stream; with a cancellation token the wait polls it (a running kernel is left to finish: milliseconds).
static void stream_wait(ThreadCtx& ctx, const CancelToken* c) {
  if (!c) {
    // short queries: poll for a while before blocking — a blocking wait is woken by an interrupt, microseconds after the stream drained
    // (config 2 is a 62 us kernel; PG_NO_SPIN_WAIT is the A/B knob)
    // At most four callers poll at a time: with 64 callers every one of them polling (hipStreamQuery takes the runtime's locks, and the
    // pollers outnumber the cores) 16.9 k queries/s at 16 callers fell to 5.4 k at 64 with a p99 of 80 ms (profiles/r05_concurrency.txt);
    // the others block at once and are woken by the interrupt.
    static std::atomic<int> pollers{0};
    const bool no_spin = knobs().no_spin_wait;
    if (!no_spin && pollers.fetch_add(1, std::memory_order_acq_rel) < 4) {
      const double until = now_ms() + 0.3;
      do {
        const hipError_t e = hipStreamQuery(ctx.stream);
        if (e == hipSuccess) { pollers.fetch_sub(1, std::memory_order_acq_rel); return; }
        if (e != hipErrorNotReady) { pollers.fetch_sub(1, std::memory_order_acq_rel); PG_HIP(e); }
      } while (now_ms() < until);
      pollers.fetch_sub(1, std::memory_order_acq_rel);
    } else if (!no_spin) {
      pollers.fetch_sub(1, std::memory_order_acq_rel);
    }
    PG_HIP(hipStreamSynchronize(ctx.stream));
    return;
  }
  for (;;) {
    const hipError_t e = hipStreamQuery(ctx.stream);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) PG_HIP(e);
    if (c->requested.load(std::memory_order_acquire)) check_cancel(c, &ctx);
  }
  check_cancel(c, &ctx);
}

namespace {
std::mutex g_pinned_mu;
std::vector<std::pair<void*, size_t>> g_pinned_pool;   // parked blocks (at most 8)
}
PinnedBlock::~PinnedBlock() {
  if (!ptr) return;
  {
    std::lock_guard<std::mutex> g(g_pinned_mu);
    if (g_pinned_pool.size() < 8 && size <= ((size_t)256 << 20)) { g_pinned_pool.emplace_back(ptr, size); return; }
  }
  (void)hipHostFree(ptr);
}
std::shared_ptr<PinnedBlock> acquire_pinned(size_t bytes) {
  auto b = std::make_shared<PinnedBlock>();
  {
    std::lock_guard<std::mutex> g(g_pinned_mu);
    size_t best = g_pinned_pool.size();
    for (size_t i = 0; i < g_pinned_pool.size(); i++)
      if (g_pinned_pool[i].second >= bytes && (best == g_pinned_pool.size() || g_pinned_pool[i].second < g_pinned_pool[best].second)) best = i;
    if (best < g_pinned_pool.size()) {
      b->ptr = g_pinned_pool[best].first;
      b->size = g_pinned_pool[best].second;
      g_pinned_pool.erase(g_pinned_pool.begin() + (long)best);
      return b;
    }
  }
  const size_t want = bytes + bytes / 4 + 4096;
  PG_HIP(hipHostMalloc(&b->ptr, want, hipHostMallocPortable));
  b->size = want;
  return b;
}

// Chunk capacity of the partition pipeline's tuple area under the striped allocator: scatter workgroup b claims ids from stripe
// b % PG_P2_STRIPES, so every stripe must hold what ITS workgroups can need — their share of the tuples (+ 25 %: matches are spread
// over the interleaved quartets, not perfectly), a partly filled chunk and a padded line per bucket, the ids claimed ahead.
static size_t p2_stripe_capacity(size_t tuples, int sgrid, int nb, size_t round_tuples, size_t min_wg_tuples) {
  const size_t stripes_used = (size_t)std::min(sgrid, PG_P2_STRIPES);
  const size_t wgs_per_stripe = ((size_t)sgrid + stripes_used - 1) / stripes_used;
  size_t per_wg_tuples = (tuples + (size_t)sgrid - 1) / (size_t)sgrid;
  per_wg_tuples = std::max(per_wg_tuples + per_wg_tuples / 4 + round_tuples, min_wg_tuples);
  const size_t per_wg_chunks = per_wg_tuples / PG_P2_CHUNK + 2 * (size_t)nb + PG_P2_BATCH + 8;
  return wgs_per_stripe * per_wg_chunks;
}

// ---- pruned-offer passes (pg_kernels_oct.hip, PgQueryPlan::oct == 2) ---------------------------------------------------------------------
// GROUP BY over a key space whose 32-bit COUNTs fit LDS, with ONE DISTINCTCOUNTHLL whose registers do not (config 5 flat).  The doc space is
// walked in passes of growing size; per pass: pg_oct_p (COUNT in LDS, offers that cannot raise a register of their group dropped, survivors
// -> tuple stream) -> pg_p2_scatter_stream -> chunk index -> pg_p2_aggregate_1n (registers of a bucket of groups in LDS) ->
// pg_oct_merge_aux_kernel (max into the registers of the passes before) -> pg_oct_floor_kernel (the groups' smallest registers: the next
// pass's floors).  Everything is queued on the stream without a host round trip; areas are sized for "every offer of the pass survives".
static long long tiles_docs(int t0, int t1) { return (long long)(t1 - t0) * PG_WAVE_DOCS; }
// Pass boundaries.  What makes a floor rise is the number of offers a REGISTER has seen: after ~6 offers per register hardly any register of
// a group is empty (floor >= 1: half of the offers are dropped), after ~24 the floors sit at 2-3, after ~96 at 4-5.  So the passes end
// where the docs seen so far amount to 6 / 24 / 96 offers per register — 20 M / 79 M / 315 M docs for config 5's 12 800 x 256 registers,
// i.e. 2 % / 8 % / 30 % of 10^9 docs (13.7 % of all offers survive) but 10 % / 40 % of 2 x 10^8 (41 %: profiles/r04_h_pruned_passes.txt).
static std::vector<int> oct_pass_bounds(int n_wtiles, int64_t registers) {
  std::vector<double> frac;
  if (const char* e = knobs().oct_passes.empty() ? nullptr : knobs().oct_passes.c_str()) {   // test / measurement knob: cumulative fractions, e.g. "0.02,0.08,0.3,1"
    for (const char* c = e; *c;) {
      char* end = nullptr;
      const double v = strtod(c, &end);
      if (end == c) break;
      frac.push_back(v);
      c = *end == ',' ? end + 1 : end;
    }
  }
  if (frac.empty()) {
    const double docs = (double)n_wtiles * PG_WAVE_DOCS;
    for (double per_register : {6.0, 24.0, 96.0}) {
      const double f = per_register * (double)registers / docs;
      if (f < 0.6) frac.push_back(f);   // a pass that would leave less than 40 % of the docs to the next one is folded into the last
    }
    frac.push_back(1.0);
  }
  std::vector<int> b;
  int prev = 0;
  for (double f : frac) {
    int t = (int)std::llround(f * n_wtiles);
    t = std::max(prev + 1, std::min(t, n_wtiles));
    if (t > n_wtiles) break;
    b.push_back(t);
    prev = t;
    if (t == n_wtiles) break;
  }
  if (b.empty() || b.back() != n_wtiles) b.push_back(n_wtiles);
  return b;
}

static void run_oct_pruned(CompiledPlan& P, PgQueryPlan& D, ThreadCtx& ctx, const CancelToken* cancel, const std::vector<uint32_t*>& aux_final, int64_t n_out) {
  const size_t G = (size_t)D.n_groups, G_pad = (G + 3) & ~(size_t)3;
  const int NB = D.radix_buckets, Q = pg_p2_round_quads[1];
  D.match_words = nullptr;
  if (!P.match_all) {   // the filter's match words first (one launch, no host round trip: nothing here is sized by the match count)
    ThreadCtx::grow(ctx.words, (size_t)D.n_wtiles * 64 * 4);
    PgQueryPlan F = D;
    F.agg_mode = PG_AGG_NONE;
    F.out_words = ctx.words.as<uint64_t>();
    const char* fname = "";
    const LaunchShape fshape = launch_shape(P, D.n_wtiles, PG_AGG_NONE);
    hipLaunchKernelGGL(select_kernel(P, PG_AGG_NONE, &fname), dim3(fshape.grid), dim3(fshape.block), fshape.lds, ctx.stream, F);
    PG_HIP(hipGetLastError());
    D.match_words = ctx.words.as<uint32_t>();
  }
  const std::vector<int> bounds = oct_pass_bounds(D.n_wtiles, (int64_t)G << D.aux[0].log2m);
  const int n_pass = (int)bounds.size();
  int max_tiles = 0;
  for (int i = 0, prev = 0; i < n_pass; prev = bounds[(size_t)i], i++) max_tiles = std::max(max_tiles, bounds[(size_t)i] - prev);
  const int ogrid_max = std::max(1, std::min((max_tiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus()));
  // the survivor stream: one region per pg_oct_p workgroup, sized for "every offer of the workgroup's docs survives" + a padded block per wavefront
  auto region_of = [&](int tiles, int ogrid) {
    const size_t tiles_per_wg = (size_t)((tiles + ogrid * PG_WAVES_PER_BLOCK - 1) / (ogrid * PG_WAVES_PER_BLOCK)) * PG_WAVES_PER_BLOCK;
    return tiles_per_wg * PG_WAVE_DOCS + (size_t)PG_WAVES_PER_BLOCK * 256;   // a multiple of 2 048
  };
  if (ogrid_max > PG_OCT_MAX_REGIONS) fail(PG_ERR_INTERNAL, "pruned-offer passes: %d workgroups", ogrid_max);
  size_t stream_cap = 0;
  for (int i = 0, prev = 0; i < n_pass; prev = bounds[(size_t)i], i++) {
    const int tiles = bounds[(size_t)i] - prev;
    const int og = std::max(1, std::min((tiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus()));
    stream_cap = std::max(stream_cap, region_of(tiles, og) * (size_t)og);
  }
  if (stream_cap >= ((size_t)1 << 32)) fail(PG_ERR_UNSUPPORTED, "pruned-offer pass of %zu entries", stream_cap);
  ThreadCtx::grow(ctx.oct_stream, stream_cap * 4 + 256);
  ThreadCtx::grow(ctx.oct_floor, G_pad + 256);
  ThreadCtx::grow(ctx.oct_counts, (size_t)ogrid_max * G * 4 + 256);
  PG_HIP(hipMemsetAsync(ctx.oct_counts.ptr, 0, (size_t)ogrid_max * G * 4, ctx.stream));   // the passes add their COUNT partials row by row
  if (!ctx.oct_cursor.ptr) ctx.oct_cursor.alloc((size_t)PG_OCT_CTRL_DWORDS * 4, true);
  PG_HIP(hipMemsetAsync(ctx.oct_floor.ptr, 0, G_pad, ctx.stream));
  // the registers accumulate over the passes (max): they start from zero
  const size_t aux_bytes = P.aux_bytes[0];
  PG_HIP(hipMemsetAsync(aux_final[0], 0, aux_bytes, ctx.stream));
  // partition pipeline areas for the largest pass
  const size_t round_tuples = (size_t)PG_P2_WAVES * (size_t)Q * 256;
  const size_t s_lds = ((size_t)6 * PG_P2_MAX_BUCKETS + PG_P2_POOL + 8 + 2 * (round_tuples / PG_P2_LINE + (size_t)NB + 1) + round_tuples + (size_t)NB * PG_P2_LINE) * 4;
  const int p2_wgs = knobs().p2_wgs_per_cu;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(p2_wgs, 1), (lds_per_cu() - 1024) / (s_lds + 512)));
  const int sgrid_max = num_cus() * per_cu;
  // (a stream scatter workgroup takes at least PG_P2_STREAM_MIN_QUARTETS = 2 quartets of 8 192 entries when the stream is short)
  const size_t min_wg_tuples = (size_t)2 * PG_P2_WAVES * PG_WAVE_DOCS + round_tuples;
  const size_t cap = p2_stripe_capacity(stream_cap, sgrid_max, NB, round_tuples, min_wg_tuples) * PG_P2_STRIPES;
  if (cap >= ((size_t)1 << 27)) fail(PG_ERR_UNSUPPORTED, "partition pipeline: %zu chunks", cap);
  D.p2_capacity = (int32_t)cap;
  D.p2_plane_stride = (int64_t)(cap + 1) * PG_P2_CHUNK;
  ThreadCtx::grow(ctx.radix_tuples, (size_t)D.p2_plane_stride * 4 + 256);
  ThreadCtx::grow(ctx.p2_meta, cap * 4 + 64);
  ThreadCtx::grow(ctx.p2_list, cap * 4 + 64);
  if (!ctx.p2_ctrl.ptr) ctx.p2_ctrl.alloc((size_t)PG_P2_CTRL_DWORDS * 4, true);
  if (!ctx.p2_ctrl_host) PG_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx.p2_ctrl_host), 256, hipHostMallocDefault));
  memset(ctx.p2_ctrl_host, 0, 256);
  PG_HIP(hipMemsetAsync(ctx.p2_ctrl.ptr, 0, (size_t)PG_P2_CTRL_DWORDS * 4, ctx.stream));   // incl. the error flag: sticky over the passes
  PG_HIP(hipMemsetAsync(ctx.oct_cursor.ptr, 0, 8, ctx.stream));
  D.p2_tuples = ctx.radix_tuples.as<uint32_t>();
  D.p2_meta = ctx.p2_meta.as<uint32_t>();
  D.p2_list = ctx.p2_list.as<uint32_t>();
  D.p2_ctrl = ctx.p2_ctrl.as<uint32_t>();
  D.oct_floor = ctx.oct_floor.as<uint8_t>();
  D.oct_stream = ctx.oct_stream.as<uint32_t>();
  D.oct_cursor = ctx.oct_cursor.as<uint32_t>();
  D.oct_stream_cap = (int64_t)stream_cap;
  const int slices_max = std::max(D.radix_slices, 1);   // what the partial register areas were sized for
  const size_t o_lds = ((G * 4 + G_pad + 15) & ~(size_t)15) + (size_t)PG_WAVES_PER_BLOCK * 1024 * 4 + 64;   // counts | floors | a 1 024-entry ring per wavefront
  int parts = 0;   // per-workgroup COUNT partials written so far
  for (int pass = 0, t0 = 0; pass < n_pass; t0 = bounds[(size_t)pass], pass++) {
    check_cancel(cancel, &ctx);
    const int t1 = bounds[(size_t)pass], tiles = t1 - t0;
    const int ogrid = std::max(1, std::min((tiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus()));
    PgQueryPlan O = D;
    O.oct_t0 = t0;
    O.oct_t1 = t1;
    O.oct_counts = ctx.oct_counts.as<uint32_t>();
    O.oct_region = (int32_t)region_of(tiles, ogrid);
    O.oct_n_regions = ogrid;
    // the survivors go through the partition pipeline (sized for all of the pass's docs; the regions' true fills are read on the device)
    const size_t pass_entries = (size_t)O.oct_region * (size_t)ogrid;
    const int quartets = (int)((pass_entries / PG_WAVE_DOCS + PG_P2_WAVES) / PG_P2_WAVES);
    const int sgrid = std::max(1, std::min(quartets, sgrid_max));
    const size_t pass_stripe_cap = std::min(cap / PG_P2_STRIPES, p2_stripe_capacity(pass_entries, sgrid, NB, round_tuples, min_wg_tuples));
    const size_t pass_cap = pass_stripe_cap * PG_P2_STRIPES;
    hipLaunchKernelGGL(pg_oct_pass_reset_kernel, dim3((unsigned)std::min<size_t>(1024, (pass_cap + 255) / 256 + 2)), dim3(256), 0, ctx.stream,
                       ctx.p2_meta.as<uint32_t>(), (int64_t)pass_cap, ctx.p2_ctrl.as<uint32_t>(), ctx.oct_cursor.as<uint32_t>());
    hipLaunchKernelGGL(D.match_words ? pg_oct_pm : pg_oct_p, dim3(ogrid), dim3(PG_BLOCK), o_lds, ctx.stream, O);
    PG_HIP(hipGetLastError());
    parts = std::max(parts, ogrid);
    hipLaunchKernelGGL(pg_oct_stream_index_kernel, dim3(1), dim3(256), 0, ctx.stream, ctx.oct_cursor.as<uint32_t>(), ogrid);
    PgQueryPlan S = O;
    S.match_words = nullptr;
    S.n_ops = 0;   // COUNT is pg_oct_p's: the aggregation pass sees HyperLogLog offers only
    S.p2_capacity = (int32_t)pass_cap;
    S.p2_stripe_cap = (int32_t)pass_stripe_cap;
    // later passes keep a fraction of their offers (floors): fewer aggregation slices for them (the scatter sizes itself on the device)
    const double keep = pass == 0 ? 1.0 : (pass == 1 ? 0.75 : 0.25);
    const size_t est = (size_t)((double)tiles * PG_WAVE_DOCS * keep) + 1;
    // every pass takes all the slices: a short stream still leaves a chunk or two per (scatter workgroup, bucket), and a work item walks
    // its chunks one wavefront each — 157 work items over 490 half-empty chunks took 157 us (profiles/r04_d_kernels__cfg5_.txt)
    (void)est;
    S.radix_slices = slices_max;
    hipLaunchKernelGGL(pg_p2_scatter_stream, dim3(sgrid), dim3(PG_P2_WAVES * 64), s_lds, ctx.stream, S);
    PG_HIP(hipGetLastError());
    const int igrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)num_cus(), (pass_cap + 4095) / 4096));
    hipLaunchKernelGGL(pg_p2_index_count_kernel, dim3(igrid), dim3(1024), 0, ctx.stream, S);
    hipLaunchKernelGGL(pg_p2_index_scan_kernel, dim3(1), dim3(PG_P2_MAX_BUCKETS), 0, ctx.stream, S);
    hipLaunchKernelGGL(pg_p2_index_fill_kernel, dim3(igrid), dim3(1024), 0, ctx.stream, S);
    PG_HIP(hipGetLastError());
    const int agrid = std::min(NB * S.radix_slices, num_cus());
    hipLaunchKernelGGL(D.p2_byte_regs ? pg_p2_aggregate_1b : pg_p2_aggregate_1n, dim3(agrid), dim3(PG_P2_AGG_THREADS), P.lds_bytes + 64, ctx.stream, S);
    PG_HIP(hipGetLastError());
    hipLaunchKernelGGL(pg_oct_merge_floor_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, ctx.stream, S.aux[0].base, aux_final[0],
                       ctx.oct_floor.as<uint8_t>(), (int)G, D.aux[0].log2m, D.radix_shift, S.radix_slices);
    PG_HIP(hipGetLastError());
    {
      const bool trace = knobs().trace_oct;   // debugging knob (synchronises): what each pass left in the stream
      if (trace) {
        uint32_t tiles = 0;
        std::vector<uint8_t> fl(G);
        PG_HIP(hipStreamSynchronize(ctx.stream));
        PG_HIP(hipMemcpy(&tiles, ctx.oct_cursor.ptr, 4, hipMemcpyDeviceToHost));
        PG_HIP(hipMemcpy(fl.data(), ctx.oct_floor.ptr, G, hipMemcpyDeviceToHost));
        uint64_t hist[8] = {0};
        for (uint8_t f : fl) hist[f < 7 ? f : 7]++;
        fprintf(stderr, "[pg] pruned pass %d: tiles [%d, %d) = %lld docs, %d workgroups, region %d, stream tiles %u (<= %lld entries); floors after the pass:"
                        " %llu %llu %llu %llu %llu %llu %llu %llu+\n", pass, t0, t1, (long long)tiles_docs(t0, t1), ogrid, O.oct_region, tiles, (long long)tiles * PG_WAVE_DOCS,
                (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4],
                (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7]);
      }
    }
  }
  // the error flags of all the passes: p2_ctrl {chunks claimed by the last pass, out of chunks}, cursor {entries of the last pass, overflow}
  PG_HIP(hipMemcpyAsync(ctx.p2_ctrl_host + 4, ctx.p2_ctrl.ptr, 8, hipMemcpyDeviceToHost, ctx.stream));
  PG_HIP(hipMemcpyAsync(ctx.p2_ctrl_host + 6, ctx.oct_cursor.ptr, 8, hipMemcpyDeviceToHost, ctx.stream));
  // COUNT row of the table = the passes' per-workgroup counters (the plan has at most this one accumulator)
  if (D.n_ops == 1) {
    hipLaunchKernelGGL(pg_oct_reduce_counts_kernel, dim3((unsigned)((G + 63) / 64)), dim3(256), 0, ctx.stream, ctx.oct_counts.as<uint32_t>(),
                       ctx.final_table.as<int64_t>(), parts, (int)G);
    PG_HIP(hipGetLastError());
  }
  (void)n_out;
}

// the columns' names, types and host dictionaries, for pg_result_data_table_v4
void fill_result_schema(Segment& seg, const pg_query& q, Result& r) {
  Result* res = &r;
    auto column_of = [&](const char* name, ResultColumn& rc) {
      rc.name = name ? name : "*";
      Column* c = name ? seg.find(name) : nullptr;
      if (c) {
        rc.data_type = c->data_type;
        if (c->has_dictionary && !c->dict_host.empty()) { rc.dict = c->dict_host.data(); rc.dict_width = c->dict_bytes_per_value; }
      }
    };
    res->schema_keys.resize((size_t)q.n_group_by);
    for (int j = 0; j < q.n_group_by; j++) column_of(q.group_by_columns[j], res->schema_keys[(size_t)j]);
    res->schema_aggs.resize((size_t)q.n_aggregations);
    for (int a = 0; a < q.n_aggregations; a++) {
      const char* col = q.aggregations[a].column;
      column_of(col && strcmp(col, "*") != 0 ? col : nullptr, res->schema_aggs[(size_t)a]);
      res->schema_aggs[(size_t)a].function = q.aggregations[a].function;
    }
    res->schema_segment = seg.alive;
    res->schema_null_handling = (q.flags & PG_QUERY_FLAG_NULL_HANDLING) != 0;
  }

static std::unique_ptr<Result> execute_query_impl(Segment& seg, const pg_query& q, const CancelToken* cancel, const ExecOptions& opt);

// numGroupsLimit without a docId per tuple.  The reference admits groups in docId order until `limit` exist
// (DictionaryBasedGroupKeyGenerator.java:416-446); equivalent: keep the `limit` groups whose FIRST matching docId is smallest.  Plans
// whose key space exceeds the limit therefore carry a MIN(docId) accumulator — in the partition pipeline a second plane that doubles
// every tuple (COUNT over 10^6 groups: 1.50 ms per 2 x 10^8 docs against 0.80 ms with the limit raised, profiles/r04_ab_*).  But only the
// groups' ORDER of first appearance matters, and the first `limit` groups all appear early: run the MIN(docId) plan (COUNT(*) only) over
// a doc PREFIX; if it holds >= limit groups, every group outside it starts later than all of them, so the prefix's first docIds decide
// the admission exactly — and the whole segment is then aggregated by the plan WITHOUT the accumulator (one-plane tuples), its table
// trimmed by the prefix's order.  A prefix with fewer groups is grown 8 x; past a quarter of the segment the one-pass plan runs.
static std::unique_ptr<Result> execute_limit_by_prefix(Segment& seg, const pg_query& q, const CompiledPlan& P, const CancelToken* cancel) {
  const int32_t limit = P.num_groups_limit;
  const int64_t G = P.dev.n_groups;
  pg_query qm = q;
  qm.num_groups_limit = INT32_MAX;
  auto pm = get_plan(seg, qm.filter, &qm);
  const bool trace = knobs().trace_host;
  if (pm->first_doc_op >= 0 || pm->dev.agg_mode != PG_AGG_RADIX || !pm->dev.p2 || (int64_t)pm->dev.n_groups != G || pm->non_scan_based) {
    if (trace) fprintf(stderr, "[pg] limit by prefix: the plan without the limit is not a partition-pipeline plan (mode %d, p2 %d)\n", pm->dev.agg_mode, pm->dev.p2);
    return nullptr;
  }
  pg_agg_spec count_star;
  memset(&count_star, 0, sizeof(count_star));
  count_star.function = PG_AGG_COUNT;
  count_star.column = "*";
  pg_query qp = q;
  qp.aggregations = &count_star;
  qp.n_aggregations = 1;
  qp.flags &= PG_QUERY_FLAG_PROFILE | PG_QUERY_FLAG_SKIP_STAR_TREE | PG_QUERY_FLAG_APPROX_FILTER_STATS;
  qp.flags |= PG_QUERY_FLAG_APPROX_FILTER_STATS;   // the prefix's statistics are not the query's
  auto pp = get_plan(seg, qp.filter, &qp);
  if (pp->first_doc_op < 0 || pp->dev.agg_mode != PG_AGG_RADIX || !pp->dev.p2 || (int64_t)pp->dev.n_groups != G || pp->space_docs != P.space_docs) {
    if (trace) fprintf(stderr, "[pg] limit by prefix: the prefix plan is not a partition-pipeline plan (mode %d, p2 %d, first-doc op %d)\n", pp->dev.agg_mode, pp->dev.p2, pp->first_doc_op);
    return nullptr;
  }
  const int64_t ident = pg_acc_identity(PG_ACC_MIN, 0);
  HostTable raw;
  float prefix_ms = 0;
  bool decided = false;
  // dense tables without auxiliary state: nothing but the admitted groups' rows leaves HBM (ExecOptions::admit_on_device)
  const bool on_device = pm->dev.n_aux == 0 && pp->dev.n_aux == 0 && !knobs().no_device_trim && (int64_t)limit * 2 + 64 <= G;
  for (int64_t np = std::max<int64_t>(knobs().limit_prefix_min_docs, 16 * (int64_t)limit); np * 4 <= (int64_t)P.space_docs; np *= 8) {
    np = (np + PG_WAVE_DOCS - 1) / PG_WAVE_DOCS * PG_WAVE_DOCS;
    ExecOptions o;
    o.doc_limit = np;
    int64_t found = 0;
    if (on_device) o.prefix_groups_out = &found;
    else o.raw_out = &raw;
    auto r = execute_query_impl(seg, qp, cancel, o);
    prefix_ms += r->stats.device_ms_total;
    if (!on_device) {
      const int64_t* first = raw.table.data() + (size_t)pp->first_doc_op * (size_t)G;
      for (int64_t g = 0; g < G && found < limit; g++) found += first[g] != ident;
    }
    if (trace) fprintf(stderr, "[pg] limit by prefix: %lld docs hold %s%lld groups (limit %d)\n", (long long)np, found >= limit ? ">= " : "", (long long)found, limit);
    if (found >= limit) { decided = true; break; }
  }
  if (!decided) return nullptr;
  LimitAdmission adm{on_device ? nullptr : raw.table.data() + (size_t)pp->first_doc_op * (size_t)G, limit};
  ExecOptions o;
  if (on_device) o.admit_on_device = limit;
  else o.admit = &adm;
  auto res = execute_query_impl(seg, qm, cancel, o);
  res->stats.device_ms_aggregate += prefix_ms;   // the prefix pass is part of the query's device time
  res->stats.device_ms_total += prefix_ms;
  snprintf(res->stats.kernel, sizeof(res->stats.kernel), "pg_part_group_by_prefix");
  return res;
}

std::unique_ptr<Result> execute_query_plain(Segment& seg, const pg_query& q, const CancelToken* cancel) {
  if (q.n_group_by > 0 && q.n_aggregations > 0 && q.aggregations && !(q.flags & PG_QUERY_FLAG_KEEP_DEVICE_TABLE) && !knobs().no_limit_prefix) {
    auto plan = get_plan(seg, q.filter, &q);
    const CompiledPlan& P = *plan;
    if (P.first_doc_op >= 0 && P.dev.agg_mode == PG_AGG_RADIX && P.dev.p2 && !P.dev.mv && P.star_tree_index < 0 && !P.non_scan_based)
      if (auto r = execute_limit_by_prefix(seg, q, P, cancel)) return r;
  }
  return execute_query_impl(seg, q, cancel, ExecOptions());
}

// Per-device admission: at most `max_inflight` queries between submission and result on one GPU, the others wait on a condition
// variable (no polling).  A query of this path fills the whole GPU, so throughput peaks at ~4 callers (profiles/r05_concurrency.txt:
// config 2 17.3 k queries/s at 4 callers, 16.7 k at 16); at 64 callers — more than the host has cores for the runtime's waits — it fell to
// 5.3 k queries/s with a p99 of 78 ms.  The reference bounds its workers the same way (a fixed pool of query worker threads,
// BaseCombineOperator.java:97-142 runs at most maxExecutionThreads tasks per query).  PG_MAX_INFLIGHT (default 16; <= 0: no bound).
namespace {
struct Admission {   // first come, first served, ONE wake-up per finished query: a plain counter + notify_one starved callers (p99 84 ms
  std::mutex mu;     // next to a p50 of 0.8 ms at 64 callers), tickets + notify_all woke every sleeper per query (16.6 k -> 7.3 k queries/s)
  int inflight = 0;
  struct Waiter { std::condition_variable cv; bool go = false; };
  std::deque<Waiter*> queue;
};
Admission g_admission[PG_MAX_DEVICES];
struct AdmissionGuard {
  Admission* a = nullptr;
  AdmissionGuard(int device, int limit) {
    if (limit <= 0 || device < 0 || device >= PG_MAX_DEVICES) return;
    a = &g_admission[device];
    std::unique_lock<std::mutex> lk(a->mu);
    if (a->inflight < limit && a->queue.empty()) { a->inflight++; return; }
    Admission::Waiter w;
    a->queue.push_back(&w);
    w.cv.wait(lk, [&] { return w.go; });   // the finishing query handed its slot over (inflight stays as it was)
  }
  ~AdmissionGuard() {
    if (!a) return;
    std::lock_guard<std::mutex> lk(a->mu);
    if (a->queue.empty()) { a->inflight--; return; }
    Admission::Waiter* w = a->queue.front();
    a->queue.pop_front();
    w->go = true;
    w->cv.notify_one();
  }
};
}  // namespace

static std::unique_ptr<Result> execute_query_impl(Segment& seg, const pg_query& q, const CancelToken* cancel, const ExecOptions& opt) {
  AdmissionGuard admitted(seg.device, knobs().max_inflight);
  const double t0 = now_ms();
  if (q.n_aggregations <= 0 || !q.aggregations) fail(PG_ERR_INVALID_ARGUMENT, "query has no aggregation");
  check_cancel(cancel, nullptr);
  ThreadCtx& ctx = ctx_on(seg.device);
  auto plan = get_plan(seg, q.filter, &q);
  CompiledPlan& P = *plan;
  const double t_plan = now_ms();
  check_cancel(cancel, nullptr);
  const bool profile = (q.flags & PG_QUERY_FLAG_PROFILE) != 0;

  if (P.non_scan_based) {
    // NonScanBasedAggregationOperator#getNextBlock (core/operator/query/NonScanBasedAggregationOperator.java:83-150): answered
    // from the dictionaries on the host — no kernel, no doc is read
    auto res = std::make_unique<Result>();
    res->num_groups = 1;
    res->aggs.resize(P.aggs.size());
    for (size_t a = 0; a < P.aggs.size(); a++) {
      const AggOut& ao = P.aggs[a];
      AggResult& r = res->aggs[a];
      for (int k = 0; k < 2; k++) { r.d[k].assign(1, 0.0); r.l[k].assign(1, 0); }
      Column* c = ao.aux_col;
      switch (ao.function) {
        case PG_AGG_COUNT: r.kind = PG_RESULT_LONG; r.l[0][0] = seg.total_docs; break;
        case PG_AGG_MIN: r.kind = PG_RESULT_DOUBLE; r.d[0][0] = dictionary_value_as_double(*c, 0); break;
        case PG_AGG_MAX: r.kind = PG_RESULT_DOUBLE; r.d[0][0] = dictionary_value_as_double(*c, c->cardinality - 1); break;
        case PG_AGG_MINMAXRANGE:
          r.kind = PG_RESULT_MINMAX_PAIR;
          r.d[0][0] = dictionary_value_as_double(*c, 0);
          r.d[1][0] = dictionary_value_as_double(*c, c->cardinality - 1);
          break;
        case PG_AGG_DISTINCTCOUNT:
          r.kind = PG_RESULT_DICTID_SET;
          r.set_sizes.assign(1, c->cardinality);
          r.set_ids.resize((size_t)c->cardinality);
          for (int32_t d = 0; d < c->cardinality; d++) r.set_ids[(size_t)d] = d;
          break;
        default:
          r.kind = PG_RESULT_HLL;
          r.log2m = ao.log2m;
          r.hll.assign((size_t)1 << ao.log2m, 0);
          hll_registers_of_dictionary(*c, ao.log2m, r.hll.data());
          break;
      }
    }
    res->stats.num_docs_scanned = seg.total_docs;   // "Set numDocsScanned to numTotalDocs for backward compatibility" (:300-303)
    res->stats.num_total_docs = seg.total_docs;
    res->stats.stats_exact = 1;
    res->stats.star_tree_index = -1;
    res->stats.host_ms_plan = (float)(t_plan - t0);
    res->stats.host_ms_total = (float)(now_ms() - t0);
    return res;
  }

  PgQueryPlan D = P.dev;
  int64_t space_docs = P.space_docs;
  if (opt.doc_limit > 0 && opt.doc_limit < space_docs) {   // a doc prefix (the partition pipeline honours D.num_docs / D.n_wtiles everywhere)
    space_docs = opt.doc_limit;
    D.num_docs = (decltype(D.num_docs))opt.doc_limit;
    D.n_wtiles = (int32_t)((opt.doc_limit + PG_WAVE_DOCS - 1) / PG_WAVE_DOCS);
  }
  // small doc spaces with per-doc state merges (PgQueryPlan::tile_split_shift): up to 128 wavefronts share a wave tile
  int split_shift = 0;
  {
    const bool no_split = knobs().no_tile_split;   // measurement knob
    const bool table_mode = D.agg_mode == PG_AGG_LDS || D.agg_mode == PG_AGG_SINGLE || D.agg_mode == PG_AGG_LDS_PART || D.agg_mode == PG_AGG_GLOBAL;
    if (!no_split && table_mode && D.n_aux > 0 && !uses_fast_kernel(P, D.agg_mode))
    {
      // (serialized-HyperLogLog merges — a star-tree's pair column — are a serial chain of loads and compare-and-swaps per wavefront: 16 docs
      // each; the other states take one atomic per doc and stop at 64 docs per wavefront)
      const int knob = knobs().tile_split_max;   // tuning knob (≤ 9: 8 quad slots x 64 lanes)
      bool merges = false;
      for (int x = 0; x < D.n_aux; x++) merges |= D.aux[x].kind == PG_AUX_HLL_BYTES;
      const int max_split = knob >= 0 ? knob : (merges ? 7 : 5);
      while (split_shift < max_split && ((int64_t)std::max(D.n_wtiles, 1) << (split_shift + 1)) <= 4096) split_shift++;
    }
  }
  // oct-layout kernels (pg_kernels_oct.hip): one 16-wavefront workgroup per CU, no tile splitting
  const bool no_oct = knobs().no_oct_exec;   // measurement knob: plans keep D.oct, the round-3 kernels run them
  const bool oct_lds = D.oct == 1 && !no_oct, oct_pruned = D.oct == 2 && (!no_oct || D.p2_byte_regs) && D.agg_mode == PG_AGG_RADIX && D.p2;
  if (oct_lds) split_shift = 0;
  D.tile_split_shift = split_shift;
  t_spec_pin = {&P, D.agg_mode, spec_shape(P, D.agg_mode)};   // (cleared when the query has been submitted: the plan may die before this thread's next query)
  LaunchShape shape = launch_shape(P, D.n_wtiles << split_shift, D.agg_mode);
  if (oct_lds) shape = {std::max(1, std::min((D.n_wtiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus())), PG_BLOCK, (D.oct_dword ? (size_t)D.aux[0].lds_offset + (size_t)D.aux[0].rep_bytes * 4 : P.lds_bytes) + 64};
  const int64_t n_out = (int64_t)D.n_ops * D.n_groups;
  // The stats counters are zero on entry: the reduce kernel of the previous query on this stream re-zeroes them after
  // moving them behind the result table (one device→host copy per query).
  if (ctx.stats_dirty) PG_HIP(hipMemsetAsync(ctx.stats.ptr, 0, PG_MAX_STATS * 8, ctx.stream));
  ctx.stats_dirty = true;
  D.stats = ctx.stats.as<unsigned long long>();
  const size_t out_bytes = ((size_t)n_out + PG_MAX_STATS) * 8;
  ThreadCtx::grow(ctx.final_table, out_bytes);
  if (D.agg_mode == PG_AGG_GLOBAL) {
    D.partials = ctx.final_table.as<int64_t>();
    int blocks = (int)((n_out + 255) / 256);
    hipLaunchKernelGGL(pg_fill_i64_kernel, dim3(blocks), dim3(256), 0, ctx.stream, D.partials, (int64_t)D.n_groups,
                       D.n_ops, P.ops_dev.as<PgAccOp>());
  } else if (D.agg_mode == PG_AGG_LDS_PART) {
    ThreadCtx::grow(ctx.partials, (size_t)D.n_ops * (size_t)D.part_groups * 8 * (size_t)shape.grid + 8);
    D.partials = ctx.partials.as<int64_t>();
  } else if (D.agg_mode == PG_AGG_RADIX) {
    // sized where the passes are launched
  } else {
    ThreadCtx::grow(ctx.partials, (size_t)n_out * 8 * (size_t)shape.grid + 8);
    D.partials = ctx.partials.as<int64_t>();
  }
  // auxiliary regions (DISTINCTCOUNT sets / HLL registers): one zeroed HBM region per op
  size_t aux_total = 0;
  for (size_t b : P.aux_bytes) aux_total += b;
  std::vector<uint32_t*> aux_final((size_t)D.n_aux, nullptr);
  const bool radix_aux = D.agg_mode == PG_AGG_RADIX && D.n_aux > 0;   // HLL registers of a bucket in LDS, one partial per work item
  if (radix_aux) D.radix_slices = std::max(1, num_cus() / D.radix_buckets);
  // partition pipeline v2: up to two work items per CU; fewer when the matches are few (set before the aggregation launch)
  if (D.agg_mode == PG_AGG_RADIX && D.p2) D.radix_slices = std::max(1, 2 * num_cus() / D.radix_buckets);
  if (aux_total) {
    // LDS-resident states: the kernel writes one partial per workgroup behind the merged regions
    size_t partial_total = P.aux_in_lds ? aux_total * (size_t)shape.grid : 0;
    const size_t radix_items = (size_t)D.radix_buckets * (size_t)std::max(D.radix_slices, 1);
    if (radix_aux)
      for (int x = 0; x < D.n_aux; x++) partial_total += radix_items * (((size_t)D.aux[x].stride << D.radix_shift) * (D.aux[x].kind == PG_AUX_DICT_SET ? 4 : 1));   // (a set's stride counts words)
    ThreadCtx::grow(ctx.aux, aux_total + partial_total);
    {
      const bool clean = ctx.aux_clean_ptr == ctx.aux.ptr && ctx.aux_clean_bytes >= aux_total;
      ctx.aux_clean_bytes = 0;   // whatever runs next writes into it
      if (!P.aux_in_lds && !radix_aux && !clean) PG_HIP(hipMemsetAsync(ctx.aux.ptr, 0, aux_total, ctx.stream));
    }
    size_t off = 0, poff = aux_total;
    for (int x = 0; x < D.n_aux; x++) {
      aux_final[(size_t)x] = reinterpret_cast<uint32_t*>(ctx.aux.as<uint8_t>() + off);
      if (P.aux_in_lds) { D.aux[x].base = reinterpret_cast<uint32_t*>(ctx.aux.as<uint8_t>() + poff); poff += P.aux_bytes[x] * (size_t)shape.grid; }
      else if (radix_aux) { D.aux[x].base = reinterpret_cast<uint32_t*>(ctx.aux.as<uint8_t>() + poff); poff += radix_items * (((size_t)D.aux[x].stride << D.radix_shift) * (D.aux[x].kind == PG_AUX_DICT_SET ? 4 : 1)); }
      else D.aux[x].base = aux_final[(size_t)x];
      off += P.aux_bytes[x];
    }
  }
  // MBs of auxiliary state (HyperLogLog registers of thousands of groups) land in a pooled page-locked block the result keeps
  // PG_QUERY_FLAG_FINAL_DISTINCT: the states stay in HBM, one final value per group and aggregation comes back (pg_aux_finish_kernel)
  bool final_distinct = (q.flags & PG_QUERY_FLAG_FINAL_DISTINCT) != 0 && D.n_aux > 0 && !(q.flags & PG_QUERY_FLAG_KEEP_DEVICE_TABLE) &&
                        D.agg_mode != PG_AGG_RADIX_HASH && space_docs > 0;
  for (int x = 0; x < D.n_aux; x++) final_distinct = final_distinct && D.aux[x].n_rep == 1;
  const size_t summary_bytes = final_distinct ? (size_t)D.n_aux * (size_t)std::max(D.n_groups, 1) * 8 : 0;
  const size_t host_aux_bytes = final_distinct ? summary_bytes : aux_total;
  std::shared_ptr<PinnedBlock> out_block;
  if (host_aux_bytes >= ((size_t)1 << 20)) out_block = acquire_pinned(out_bytes + host_aux_bytes);
  int64_t* host_out = static_cast<int64_t*>(out_block ? out_block->ptr : ctx.pin(out_bytes + host_aux_bytes));
  uint8_t* aux_host = reinterpret_cast<uint8_t*>(host_out) + out_bytes;
  if (profile) PG_HIP(hipEventRecord(ctx.ev[0], ctx.stream));
  const char* kname = "";
  const bool has_docs = space_docs > 0;   // docs of the doc space the plan runs on (the segment's or a star-tree's)
  const bool hashed = D.agg_mode == PG_AGG_RADIX_HASH;
  const bool keep_table = (q.flags & PG_QUERY_FLAG_KEEP_DEVICE_TABLE) != 0;
  // (a multi-value plan carries no first-docId accumulator: a key space that CAN exceed numGroupsLimit would be trimmed per segment by
  // the reference and only after the merge here — so such tables never enter the element-wise merges: ADVICE r3)
  const bool mv_beyond_limit = D.mv && q.n_group_by > 0 && (int64_t)D.n_groups > (int64_t)P.num_groups_limit;
  if (keep_table && (hashed || P.first_doc_op >= 0 || mv_beyond_limit))
    fail(PG_ERR_UNSUPPORTED, "PG_QUERY_FLAG_KEEP_DEVICE_TABLE: %s has no dense table that merges element-wise (merge on the host by values)",
         hashed ? "a hashed key space" : "a key space beyond numGroupsLimit");
  std::unique_ptr<DeviceTable> kept;
  const bool radix = D.agg_mode == PG_AGG_RADIX || hashed;
  int64_t hash_groups = 0;
  bool p2_ran = false;
  if (has_docs && oct_pruned) {
    kname = "pg_oct_pruned_group_by";
    p2_ran = true;
    run_oct_pruned(P, D, ctx, cancel, aux_final, n_out);
  } else if (has_docs && radix && D.p2) {
    // ---- partition pipeline v2 (pg_kernels_part.hip): [filter → match words;] ONE scatter pass into chunked per-bucket streams of
    //      bit-packed tuples; per-bucket LDS aggregation; merge of the slices ------------------------------------------------------------
    kname = "pg_part_group_by";
    p2_ran = true;
    unsigned long long matched_now = (unsigned long long)space_docs;
    D.match_words = nullptr;
    if (!P.match_all) {
      const size_t n_words = (size_t)D.n_wtiles * 64;
      ThreadCtx::grow(ctx.words, n_words * 4);
      PgQueryPlan F = D;
      F.agg_mode = PG_AGG_NONE;
      F.out_words = ctx.words.as<uint64_t>();
      const char* fname = "";
      const LaunchShape fshape = launch_shape(P, D.n_wtiles, PG_AGG_NONE);
      hipLaunchKernelGGL(select_kernel(P, PG_AGG_NONE, &fname), dim3(fshape.grid), dim3(fshape.block), fshape.lds, ctx.stream, F);
      PG_HIP(hipGetLastError());
      // the tuple area is sized by the docs that passed the filter
      PG_HIP(hipMemcpyAsync(&matched_now, ctx.stats.ptr, 8, hipMemcpyDeviceToHost, ctx.stream));
      stream_wait(ctx, cancel);
      D.match_words = ctx.words.as<uint32_t>();
    }
    const int T = D.p2_planes, NB = D.radix_buckets, Q = pg_p2_round_quads[T];
    const size_t nbp = (size_t)((NB + 63) & ~63);
    (void)nbp;
    const size_t round_tuples = (size_t)PG_P2_WAVES * (size_t)Q * 256;
    const size_t s_lds = ((size_t)6 * PG_P2_MAX_BUCKETS + PG_P2_POOL + 8 + 2 * (round_tuples / PG_P2_LINE + (size_t)NB + 1) + (size_t)T * round_tuples +
                          (size_t)T * (size_t)NB * PG_P2_LINE) * 4;
    const int p2_wgs = knobs().p2_wgs_per_cu;   // tuning knob
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(p2_wgs, 1), (lds_per_cu() - 1024) / (s_lds + 512)));
    const int quartets = (D.n_wtiles + PG_P2_WAVES - 1) / PG_P2_WAVES;
    const int sgrid = std::max(1, std::min(quartets, num_cus() * per_cu));
    // chunks: the tuples themselves, one partly filled chunk + one padded line per (workgroup, bucket), the ids a workgroup claimed
    // ahead and did not use
    const size_t stripe_cap = p2_stripe_capacity((size_t)matched_now, sgrid, NB, round_tuples, 0);
    const size_t cap = stripe_cap * PG_P2_STRIPES;
    if (cap >= ((size_t)1 << 27)) fail(PG_ERR_UNSUPPORTED, "partition pipeline: %zu chunks", cap);
    D.p2_capacity = (int32_t)cap;
    D.p2_stripe_cap = (int32_t)stripe_cap;
    D.p2_plane_stride = (int64_t)(cap + 1) * PG_P2_CHUNK;
    ThreadCtx::grow(ctx.radix_tuples, (size_t)D.p2_plane_stride * 4 * (size_t)T + 256);
    ThreadCtx::grow(ctx.p2_meta, cap * 4 + 64);
    ThreadCtx::grow(ctx.p2_list, cap * 4 + 64);
    if (!ctx.p2_ctrl.ptr) ctx.p2_ctrl.alloc((size_t)PG_P2_CTRL_DWORDS * 4, true);
    if (!ctx.p2_ctrl_host) PG_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx.p2_ctrl_host), 256, hipHostMallocDefault));
    PG_HIP(hipMemsetAsync(ctx.p2_meta.ptr, 0xFF, cap * 4, ctx.stream));
    PG_HIP(hipMemsetAsync(ctx.p2_ctrl.ptr, 0, (size_t)PG_P2_CTRL_DWORDS * 4, ctx.stream));
    D.p2_tuples = ctx.radix_tuples.as<uint32_t>();
    D.p2_meta = ctx.p2_meta.as<uint32_t>();
    D.p2_list = ctx.p2_list.as<uint32_t>();
    D.p2_ctrl = ctx.p2_ctrl.as<uint32_t>();
    static const QueryKernel scatter_k[5] = {nullptr, pg_p2_scatter_1, pg_p2_scatter_2, pg_p2_scatter_3, pg_p2_scatter_4};
    static const QueryKernel scatter_fast_k[3] = {nullptr, pg_p2_scatter_1f, pg_p2_scatter_2f};
    static const QueryKernel aggregate_k[5] = {nullptr, pg_p2_aggregate_1, pg_p2_aggregate_2, pg_p2_aggregate_3, pg_p2_aggregate_4};
    static const QueryKernel aggregate_nogather_k[5] = {nullptr, pg_p2_aggregate_1n, pg_p2_aggregate_2n, pg_p2_aggregate_3n, pg_p2_aggregate_4n};
    bool gathers = false;   // a source travelling as a dictId is looked up in the aggregation pass
    bool gathers_lean = false;   // ... in the lean consumer (pg_p2_aggregate_*s), which computes the values of arithmetic INT dictionaries (pk_affine 3)
    for (int si = 0; si < D.n_srcs; si++) { gathers |= D.p2_fkind[si] == PG_P2_F_DICTID; gathers_lean |= D.p2_fkind[si] == PG_P2_F_DICTID && D.pk_affine[si] != 3; }
    // every work item should see >= 64 K tuples (its table is zeroed, flushed and merged whatever it aggregates)
    const int slices_max = D.radix_slices;
    D.radix_slices = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)slices_max, matched_now / ((unsigned long long)NB * 65536ULL)));
    const size_t slots = (size_t)1 << D.radix_shift;
    ThreadCtx::grow(ctx.partials, (size_t)NB * D.radix_slices * D.n_ops * slots * 8 + 8);
    D.partials = ctx.partials.as<int64_t>();
    QueryKernel sk = D.p2_fast_a && T <= 2 ? scatter_fast_k[T] : scatter_k[T];
    if (D.p2_fast_a && T == 1 && D.n_srcs == 0) sk = pg_p2_scatter_1f_key;
    if (D.p2_fast_a && T == 1 && D.n_srcs == 1 && D.p2_fkind[0] == PG_P2_F_HLL && D.pk_affine[0] == 2 && D.srcs[0].col_kind == PG_COL_FIXED_BIT) sk = pg_p2_scatter_1f_hll;
    if (D.p2_oct_a && T == 1 && P.match_all) {   // no filter pass in front: the oct-layout phase A
      static const QueryKernel oct_k[2][3] = {{pg_p2_scatter_o_key, pg_p2_scatter_o_raw, pg_p2_scatter_o_dict},
                                              {pg_p2_scatter_ow_key, pg_p2_scatter_ow_raw, pg_p2_scatter_ow_dict}};
      const int osk = D.n_srcs == 0 ? 0 : (D.p2_fkind[0] == PG_P2_F_RAW32 ? 1 : 2);
      sk = oct_k[D.gcols[0].bits > 8 ? 1 : 0][osk];
    }
    hipLaunchKernelGGL(sk, dim3(sgrid), dim3(PG_P2_WAVES * 64), s_lds, ctx.stream, D);
    PG_HIP(hipGetLastError());
#ifdef PG_P2_TIMING
    {   // measurement variant: cycles per phase of the scatter's round, summed over the wavefronts
      unsigned long long tm[13];
      PG_HIP(hipMemcpyAsync(tm, D.p2_ctrl + PG_P2_CTRL_TIMING, sizeof(tm), hipMemcpyDeviceToHost, ctx.stream));
      PG_HIP(hipStreamSynchronize(ctx.stream));
      static const char* names[13] = {"A(rest)", "wait1", "B", "wait2", "C", "wait3", "D1", "wait4", "D2", "A:loads+decode0", "A:ranks0", "A:decode1", "A:ranks1"};
      unsigned long long sum = 0;
      for (int i = 0; i < 13; i++) sum += tm[i];
      fprintf(stderr, "p2 scatter phases (%% of wavefront time, grid %d):", sgrid);
      for (int i = 0; i < 13; i++) fprintf(stderr, " %s %.1f", names[i], 100.0 * (double)tm[i] / (double)std::max<unsigned long long>(sum, 1));
      fprintf(stderr, "  | ticks per wavefront %.0f\n", (double)sum / ((double)sgrid * PG_P2_WAVES));
    }
#endif
    {   // the chunk records grouped by bucket (counting sort): count, scan, fill
      const int igrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)num_cus(), (cap + 4095) / 4096));
      hipLaunchKernelGGL(pg_p2_index_count_kernel, dim3(igrid), dim3(1024), 0, ctx.stream, D);
      hipLaunchKernelGGL(pg_p2_index_scan_kernel, dim3(1), dim3(PG_P2_MAX_BUCKETS), 0, ctx.stream, D);
      hipLaunchKernelGGL(pg_p2_index_fill_kernel, dim3(igrid), dim3(1024), 0, ctx.stream, D);
      PG_HIP(hipGetLastError());
    }
    const int agrid = std::min(NB * D.radix_slices, num_cus());
    // COUNT(*) and SUM / MIN / MAX of raw INT fields only: the consumer with the ops' descriptors in scalar registers (pg_p2_aggregate_*s)
    const bool no_simple = knobs().no_p2_simple;   // A/B knob
    bool simple = !no_simple && D.n_aux == 0 && T <= 2 && D.n_ops <= 4;
    for (int o = 0; o < D.n_ops && simple; o++) {
      const PgAccOp& op = D.ops[o];
      if (op.src < 0) simple = op.fn == PG_ACC_COUNT || (op.fn == PG_ACC_MIN && D.p2_docid_plane >= 0);
      else {
        const int kind = D.p2_fkind[op.src], vt = D.srcs[op.src].val_type;
        simple = op.is_float == PG_ACCV_INT && op.fn != PG_ACC_COUNT &&
                 ((kind == PG_P2_F_RAW32 && vt == PG_V_I32) || (kind == PG_P2_F_DICTID && (vt == PG_V_I32 || vt == PG_V_I64)));
      }
    }
    static const QueryKernel aggregate_simple_k[2][3] = {{nullptr, pg_p2_aggregate_1s, pg_p2_aggregate_2s}, {nullptr, pg_p2_aggregate_1sg, pg_p2_aggregate_2sg}};
    const bool sets = D.n_aux == 1 && D.aux[0].kind == PG_AUX_DICT_SET;   // DISTINCTCOUNT: the bucket's dictId sets in LDS (pg_p2_aggregate_1set)
    if (sets && T != 1) fail(PG_ERR_INTERNAL, "partition pipeline: dictId sets travel in one-plane tuples");
    const QueryKernel ak = sets ? pg_p2_aggregate_1set : (simple ? aggregate_simple_k[gathers_lean ? 1 : 0][T] : (gathers ? aggregate_k[T] : aggregate_nogather_k[T]));
    hipLaunchKernelGGL(ak, dim3(agrid), dim3(PG_P2_AGG_THREADS), P.lds_bytes + 64, ctx.stream, D);
    PG_HIP(hipGetLastError());
    for (int x = 0; x < D.n_aux; x++) {
      const bool set = D.aux[x].kind == PG_AUX_DICT_SET;   // (stride: words per group for a set, bytes for HyperLogLog registers)
      const int64_t n_words = set ? (int64_t)D.n_groups * D.aux[x].stride : (int64_t)D.n_groups * D.aux[x].stride / 4;
      const int64_t bucket_words = set ? (int64_t)(slots * (size_t)D.aux[x].stride) : (int64_t)(slots * (size_t)D.aux[x].stride / 4);
      hipLaunchKernelGGL(pg_radix_reduce_aux_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx.stream, D.aux[x].base,
                         aux_final[(size_t)x], D.radix_slices, bucket_words, n_words, set ? 1 : 0);
    }
    hipLaunchKernelGGL(pg_radix_reduce_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx.stream, ctx.partials.as<int64_t>(),
                       ctx.final_table.as<int64_t>(), D.n_ops, D.n_groups, D.radix_shift, D.radix_slices, P.ops_dev.as<PgAccOp>());
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(ctx.p2_ctrl_host, ctx.p2_ctrl.ptr, 8, hipMemcpyDeviceToHost, ctx.stream));
  } else if (has_docs && radix) {
    // ---- radix-partitioned group-by: filter → match words; count; offsets; scatter; per-bucket LDS aggregation; merge ---------
    kname = "pg_radix_group_by";
    const size_t n_words = (size_t)D.n_wtiles * 64;
    ThreadCtx::grow(ctx.words, n_words * 4);
    PgQueryPlan F = D;
    F.agg_mode = PG_AGG_NONE;
    F.out_words = ctx.words.as<uint64_t>();
    const char* fname = "";
    const LaunchShape fshape = launch_shape(P, D.n_wtiles, PG_AGG_NONE);
    hipLaunchKernelGGL(select_kernel(P, PG_AGG_NONE, &fname), dim3(fshape.grid), dim3(fshape.block), fshape.lds, ctx.stream, F);
    PG_HIP(hipGetLastError());
    // the tuple area is sized by the docs that passed the filter, not by the segment (48 B x 2^31 docs would not fit)
    unsigned long long matched_now = 0;
    PG_HIP(hipMemcpyAsync(&matched_now, ctx.stats.ptr, 8, hipMemcpyDeviceToHost, ctx.stream));
    stream_wait(ctx, cancel);
    if (hashed) {   // buckets sized so that even all-distinct keys half-fill a bucket's table, within [16, 2048]
      int nb = 16;
      while (nb < PG_MAX_RADIX_BUCKETS && (unsigned long long)nb * (unsigned long long)(D.hash_cap / 2) < matched_now) nb *= 2;
      if (knobs().hash_first_buckets >= 0) nb = std::max(16, std::min(PG_MAX_RADIX_BUCKETS, knobs().hash_first_buckets));   // test knob: start too low
      D.radix_buckets = nb;
    }
    const int rgrid = std::max(1, std::min((D.n_wtiles + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK, num_cus()));
    ThreadCtx::grow(ctx.radix_hist, (size_t)rgrid * D.radix_buckets * 4);
    ThreadCtx::grow(ctx.radix_start, ((size_t)D.radix_buckets + 1) * 8);
    D.radix_stride = hashed ? ((16 + 8 * (int64_t)D.n_srcs + 15) & ~(int64_t)15)
                            : (D.n_srcs == 0 ? 8 : ((8 + 8 * (int64_t)D.n_srcs + 15) & ~(int64_t)15));
    // Packed tuples (one dword per doc, staged per wavefront and bucket in LDS, written as whole 128-byte lines of 32): every
    // (workgroup, bucket) range then holds whole lines — up to one padded line per wavefront.
    D.radix_stage = 0;
    size_t stage_lds = 0;
    const int stage_waves = D.radix_buckets <= 32 ? PG_WAVES_PER_BLOCK : PG_WAVES_PER_BLOCK / 2;   // the rings of 64 buckets fit for 8 wavefronts
    if (!hashed && D.radix_packed) {
      D.radix_stage = 32;
      D.radix_stride = 4;
      stage_lds = ((size_t)2 * stage_waves * D.radix_buckets + (size_t)stage_waves * 160 /* PG_PK_WORK */ + (size_t)stage_waves * D.radix_buckets * 64) * 4;
    }
    const size_t pad_tuples = D.radix_stage ? (size_t)rgrid * (size_t)D.radix_buckets * (size_t)stage_waves * (size_t)D.radix_stage : 0;
    ThreadCtx::grow(ctx.radix_tuples, ((size_t)matched_now + pad_tuples) * (size_t)D.radix_stride + 256);
    D.match_words = ctx.words.as<uint32_t>();
    D.radix_hist = ctx.radix_hist.as<uint32_t>();
    D.radix_bucket_start = ctx.radix_start.as<uint32_t>();
    D.radix_tuples = ctx.radix_tuples.as<uint8_t>();
    uint32_t* bucket_total = D.radix_bucket_start + D.radix_buckets + 1;   // second half of the same buffer
    if (hashed) {
      D.hash_out_cap = (int64_t)std::min<unsigned long long>(matched_now, (unsigned long long)D.radix_buckets * (unsigned long long)D.hash_cap) + 1;
      ThreadCtx::grow(ctx.hash_count, 64);
      ThreadCtx::grow(ctx.hash_keys, (size_t)D.hash_out_cap * 8);
      ThreadCtx::grow(ctx.hash_acc, (size_t)D.hash_out_cap * 8 * (size_t)std::max(D.n_ops, 1));
      PG_HIP(hipMemsetAsync(ctx.hash_count.ptr, 0, 16, ctx.stream));
      D.hash_out_count = ctx.hash_count.as<unsigned long long>();
      D.hash_out_keys = ctx.hash_keys.as<int64_t>();
      D.hash_out_acc = ctx.hash_acc.as<int64_t>();
      // A bucket whose distinct keys overflow its LDS table (skewed hash ranges: the bucket count above assumes an even spread)
      // is met with more buckets — four times as many per attempt, up to PG_MAX_RADIX_BUCKETS — instead of a refusal after the work.
      for (;;) {
        hipLaunchKernelGGL(pg_hash_count_kernel, dim3(rgrid), dim3(PG_BLOCK), 0, ctx.stream, D);
        hipLaunchKernelGGL(pg_radix_offsets_kernel, dim3((unsigned)((D.radix_buckets + 15) / 16)), dim3(1024), 0, ctx.stream, D.radix_hist,
                           bucket_total, rgrid, D.radix_buckets, 0, 0);
        hipLaunchKernelGGL(pg_radix_bucket_scan_kernel, dim3(1), dim3(64), 0, ctx.stream, bucket_total, D.radix_bucket_start, D.radix_buckets);
        hipLaunchKernelGGL(pg_hash_scatter_kernel, dim3(rgrid), dim3(PG_BLOCK), 0, ctx.stream, D);
        hipLaunchKernelGGL(pg_hash_aggregate_kernel, dim3(std::min(D.radix_buckets, num_cus())), dim3(PG_BLOCK),
                           (size_t)D.hash_cap * (8 + 8 * (size_t)D.n_ops) + 64, ctx.stream, D);
        PG_HIP(hipGetLastError());
        if (D.radix_buckets >= PG_MAX_RADIX_BUCKETS) break;
        unsigned long long flag[2] = {0, 0};
        PG_HIP(hipMemcpyAsync(flag, ctx.hash_count.ptr, 16, hipMemcpyDeviceToHost, ctx.stream));
        stream_wait(ctx, cancel);
        if (flag[1] != 1) break;   // 0: fine; 2: Long.MAX_VALUE key (reported below)
        D.radix_buckets = std::min(D.radix_buckets * 4, PG_MAX_RADIX_BUCKETS);
        ThreadCtx::grow(ctx.radix_hist, (size_t)rgrid * D.radix_buckets * 4);
        ThreadCtx::grow(ctx.radix_start, ((size_t)D.radix_buckets + 1) * 8);
        D.radix_hist = ctx.radix_hist.as<uint32_t>();
        D.radix_bucket_start = ctx.radix_start.as<uint32_t>();
        bucket_total = D.radix_bucket_start + D.radix_buckets + 1;
        D.hash_out_cap = (int64_t)std::min<unsigned long long>(matched_now, (unsigned long long)D.radix_buckets * (unsigned long long)D.hash_cap) + 1;
        ThreadCtx::grow(ctx.hash_keys, (size_t)D.hash_out_cap * 8);
        ThreadCtx::grow(ctx.hash_acc, (size_t)D.hash_out_cap * 8 * (size_t)std::max(D.n_ops, 1));
        D.hash_out_keys = ctx.hash_keys.as<int64_t>();
        D.hash_out_acc = ctx.hash_acc.as<int64_t>();
        PG_HIP(hipMemsetAsync(ctx.hash_count.ptr, 0, 16, ctx.stream));
      }
      kname = "pg_hash_group_by";
    } else {
      D.radix_slices = std::max(1, num_cus() / D.radix_buckets);
      const size_t slots = (size_t)1 << D.radix_shift;
      ThreadCtx::grow(ctx.partials, (size_t)D.radix_buckets * D.radix_slices * D.n_ops * slots * 8 + 8);
      D.partials = ctx.partials.as<int64_t>();
      hipLaunchKernelGGL(pg_radix_count_kernel, dim3(rgrid), dim3(PG_BLOCK), 0, ctx.stream, D);
      hipLaunchKernelGGL(pg_radix_offsets_kernel, dim3((unsigned)((D.radix_buckets + 15) / 16)), dim3(1024), 0, ctx.stream, D.radix_hist,
                         bucket_total, rgrid, D.radix_buckets, D.radix_stage, stage_waves);
      hipLaunchKernelGGL(pg_radix_bucket_scan_kernel, dim3(1), dim3(64), 0, ctx.stream, bucket_total, D.radix_bucket_start, D.radix_buckets);
      if (D.radix_packed) hipLaunchKernelGGL(pg_radix_scatter_packed_kernel, dim3(rgrid), dim3(stage_waves * 64), stage_lds, ctx.stream, D);
      else hipLaunchKernelGGL(pg_radix_scatter_kernel, dim3(rgrid), dim3(PG_BLOCK), 0, ctx.stream, D);
      const int agrid = std::min(D.radix_buckets * D.radix_slices, num_cus());
      size_t agg_lds = (size_t)D.n_ops * slots * 8 + 64;
      for (int x = 0; x < D.n_aux; x++) agg_lds += slots * (size_t)D.aux[x].stride;
      if (D.radix_packed) hipLaunchKernelGGL(pg_radix_aggregate_packed_kernel, dim3(agrid), dim3(PG_BLOCK), agg_lds, ctx.stream, D);
      else hipLaunchKernelGGL(pg_radix_aggregate_kernel, dim3(agrid), dim3(PG_BLOCK), agg_lds, ctx.stream, D);
      for (int x = 0; x < D.n_aux; x++) {
        const int64_t n_words = (int64_t)D.n_groups * D.aux[x].stride / 4, bucket_words = (int64_t)(slots * (size_t)D.aux[x].stride / 4);
        hipLaunchKernelGGL(pg_radix_reduce_aux_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx.stream, D.aux[x].base,
                           aux_final[(size_t)x], D.radix_slices, bucket_words, n_words, 0);
      }
      hipLaunchKernelGGL(pg_radix_reduce_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx.stream, ctx.partials.as<int64_t>(),
                         ctx.final_table.as<int64_t>(), D.n_ops, D.n_groups, D.radix_shift, D.radix_slices, P.ops_dev.as<PgAccOp>());
      PG_HIP(hipGetLastError());
    }
  } else if (has_docs && oct_lds) {
    // ---- LDS-resident DISTINCTCOUNTHLL / DISTINCTCOUNT next to a small key (pg_kernels_oct.hip): [filter -> match words;] one pass ----
    D.match_words = nullptr;
    if (!P.match_all) {
      ThreadCtx::grow(ctx.words, (size_t)D.n_wtiles * 64 * 4);
      PgQueryPlan F = D;
      F.agg_mode = PG_AGG_NONE;
      F.out_words = ctx.words.as<uint64_t>();
      const char* fname = "";
      const LaunchShape fshape = launch_shape(P, D.n_wtiles, PG_AGG_NONE);
      hipLaunchKernelGGL(select_kernel(P, PG_AGG_NONE, &fname), dim3(fshape.grid), dim3(fshape.block), fshape.lds, ctx.stream, F);
      PG_HIP(hipGetLastError());
      D.match_words = ctx.words.as<uint32_t>();
    }
    const bool count_only = D.oct_src_kind == 0 && D.n_aux == 0 && !D.match_words && D.n_group_cols >= 1 && !knobs().no_oct_count_kernel;   // plan_oct: COUNT(*) alone
    kname = count_only ? "pg_oct_c" : (D.match_words ? "pg_oct_lm" : "pg_oct_l");
    hipLaunchKernelGGL(count_only ? pg_oct_c : (D.match_words ? pg_oct_lm : pg_oct_l), dim3(shape.grid), dim3(shape.block), shape.lds, ctx.stream, D);
    PG_HIP(hipGetLastError());
  } else if (has_docs) {
    // COUNT(*) behind an index-only filter of dense postings: the bitmap stream (pg_dense_count_*), not the tile walk
    const bool no_dense_count = knobs().no_dense_count;   // A/B knob
    int n_ptr = 0;   // distinct dense posting pointers (the planner pads the eight slots with repeats of slot 0)
    for (int j = 0; j < 8; j++) if (j == 0 || D.dense_ptr[j] != D.dense_ptr[0] || D.dense_group[j] != D.dense_group[0]) n_ptr = j + 1;
    if (!no_dense_count && D.agg_mode == PG_AGG_NONE && !D.out_words && !D.out_tile_counts && uses_fast_kernel(P, PG_AGG_NONE) && P.fast_filter == -1 &&
        D.dense_fused && D.n_index_instr > 0 && !D.mv) {
      static const QueryKernel kCount[8] = {pg_dense_count_1, pg_dense_count_2, pg_dense_count_3, pg_dense_count_4,
                                            pg_dense_count_5, pg_dense_count_6, pg_dense_count_7, pg_dense_count_8};
      const int64_t n_q = ((int64_t)D.num_docs + 127) >> 7;
      const int wgs_per_cu = knobs().dense_count_wgs;   // tuning knob
      const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_q + 2047) / 2048, (int64_t)num_cus() * std::max(wgs_per_cu, 1)));
      kname = "pg_dense_count";
      hipLaunchKernelGGL(kCount[n_ptr - 1], dim3(grid), dim3(1024), 0, ctx.stream, D);
    } else if (!no_dense_count && D.agg_mode == PG_AGG_NONE && !D.out_words && !D.out_tile_counts && uses_fast_kernel(P, PG_AGG_NONE) &&
               (P.fast_filter == 0 || P.fast_filter == 2) && D.n_index_instr == 0 && D.fast_scan_pushed && !D.mv) {
      // COUNT(*) behind one scan of a <= 8-bit dictionary column over the whole segment: the bit stream, 32 docs per thread (pg_dict_count_*)
      static const QueryKernel kDict[8] = {pg_dict_count_1, pg_dict_count_2, pg_dict_count_3, pg_dict_count_4,
                                           pg_dict_count_5, pg_dict_count_6, pg_dict_count_7, pg_dict_count_8};
      const int bits = std::min(std::max(P.fast_scan_bits, 1), 8);
      const int64_t n_g = ((int64_t)D.num_docs + 31) >> 5;
      const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_g + 2047) / 2048, (int64_t)num_cus()));
      kname = "pg_dict_count";
      hipLaunchKernelGGL(kDict[bits - 1], dim3(grid), dim3(1024), 0, ctx.stream, D);
    } else {
      QueryKernel kern = select_kernel(P, D.agg_mode, &kname);
      hipLaunchKernelGGL(kern, dim3(shape.grid), dim3(shape.block), shape.lds, ctx.stream, D);
    }
    PG_HIP(hipGetLastError());
  }
  if (profile) PG_HIP(hipEventRecord(ctx.ev[1], ctx.stream));
  double t_queued_at = 0, t_synced = 0;   // PG_TRACE_HOST: host-side timeline of one query
  std::vector<int64_t> table((size_t)n_out);
  std::vector<int64_t> hash_keys_host;   // PG_AGG_RADIX_HASH: raw key of every group of the compact table
  int64_t compact_groups = -1, groups_found = -1;   // the device trimmed the dense table: rows of the compact table (keys in hash_keys_host), groups that existed
  uint64_t stats_host[PG_MAX_STATS] = {0};
  if (has_docs) {
    if (P.aux_in_lds)
      for (int x = 0; x < D.n_aux; x++) {
        const int64_t n_words = (int64_t)P.aux_bytes[(size_t)x] / 4;
        hipLaunchKernelGGL(pg_reduce_aux_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx.stream, D.aux[x].base,
                           aux_final[(size_t)x], shape.grid, n_words, D.aux[x].kind == PG_AUX_DICT_SET ? 0 : 1);
        PG_HIP(hipGetLastError());
      }
    // ---- PG_QUERY_FLAG_FINAL_DISTINCT tails in one launch, results straight into the page-locked block (pg_finish_fused_kernel) -----------------
    const int reduce_mode = (D.agg_mode == PG_AGG_LDS_PART && n_out > 0) ? 2 : ((D.agg_mode != PG_AGG_GLOBAL && D.agg_mode != PG_AGG_LDS_PART && !radix && n_out > 0) ? 1 : 0);
    const bool fused_finish = final_distinct && !keep_table && !P.aux_in_lds && !radix && P.trim_size == 0 && !opt.raw_out && !knobs().no_fused_finish &&
                              out_bytes + summary_bytes <= ((size_t)1 << 20);
    if (fused_finish) {
      const int G1 = std::max(D.n_groups, 1);
      PgFinishArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.partials = ctx.partials.as<int64_t>();
      fa.out = host_out;
      fa.ops = P.ops_dev.as<PgAccOp>();
      fa.stats = ctx.stats.as<unsigned long long>();
      fa.n_wg = shape.grid; fa.n_ops = D.n_ops; fa.n_groups = D.n_groups; fa.n_parts = D.n_parts; fa.part_groups = D.part_groups;
      fa.mode = reduce_mode;
      fa.b_table = reduce_mode == 2 ? (int)((n_out + 63) / 64) : (reduce_mode == 1 ? (int)((n_out + 3) / 4) : 0);
      fa.b_aux = (G1 + 3) / 4;
      fa.n_aux = D.n_aux;
      fa.rezero = 1;
      for (int x = 0; x < D.n_aux; x++) {
        const bool set = D.aux[x].kind == PG_AUX_DICT_SET;
        const int lm = set ? 0 : D.aux[x].log2m;
        if (!set && !ctx.hll_small[lm].ptr) {
          std::vector<long long> t;
          hll_small_range_table(lm, t, ctx.hll_alpha_mm[lm]);
          ctx.hll_small[lm] = upload_vector(t);
        }
        fa.aux[x].region = aux_final[(size_t)x];
        fa.aux[x].small_range = ctx.hll_small[lm].as<long long>();
        fa.aux[x].out = reinterpret_cast<long long*>(aux_host) + (size_t)x * (size_t)G1;
        fa.aux[x].alpha_mm = ctx.hll_alpha_mm[lm];
        fa.aux[x].kind = D.aux[x].kind;
        fa.aux[x].words_per_group = set ? D.aux[x].stride : D.aux[x].stride / 4;
        fa.aux[x].m = 1 << lm;
      }
      if (reduce_mode == 0 && n_out > 0)   // the table is final in HBM already (dense HBM table): it still has to reach the block
        PG_HIP(hipMemcpyAsync(host_out, ctx.final_table.ptr, (size_t)n_out * 8, hipMemcpyDeviceToHost, ctx.stream));
      hipLaunchKernelGGL(pg_finish_fused_kernel, dim3((unsigned)(fa.b_table + 1 + fa.b_aux * D.n_aux)), dim3(256), 0, ctx.stream, fa);
      PG_HIP(hipGetLastError());
      if (profile) PG_HIP(hipEventRecord(ctx.ev[2], ctx.stream));
      ctx.aux_clean_ptr = ctx.aux.ptr;        // every state was folded and zeroed: the next query on this stream needs no fill
      ctx.aux_clean_bytes = aux_total;
    }
    if (!fused_finish && D.agg_mode == PG_AGG_LDS_PART && n_out > 0) {
      hipLaunchKernelGGL(pg_reduce_parts_kernel, dim3((unsigned)((n_out + 63) / 64)), dim3(256), 0, ctx.stream, ctx.partials.as<int64_t>(),
                         ctx.final_table.as<int64_t>(), shape.grid, D.n_ops, D.n_groups, D.n_parts, D.part_groups, P.ops_dev.as<PgAccOp>());
      PG_HIP(hipGetLastError());
    }
    const int reduce = (D.agg_mode != PG_AGG_GLOBAL && D.agg_mode != PG_AGG_LDS_PART && !radix && n_out > 0) ? 1 : 0;
    const int blocks = (reduce ? (int)((n_out + 3) / 4) : 0) + 1;   // one wavefront per output slot + the stats block
    // Small tables whose every slot this kernel writes go straight into the pinned result block (mapped into the device's address space:
    // the stores cross the bus as they retire) — no copy command behind the kernel, one launch less on the query's critical path
    // (config 2: profiles/r04_j_small_query_latency.txt).
    const bool no_direct = knobs().no_direct_result;   // A/B knob
    const bool direct_out = fused_finish || (!no_direct && !keep_table && (reduce || n_out == 0) && out_bytes <= ((size_t)64 << 10));
    if (!fused_finish) {
      hipLaunchKernelGGL(pg_reduce_partials_kernel, dim3(blocks), dim3(256), 0, ctx.stream, ctx.partials.as<int64_t>(),
                         direct_out ? host_out : ctx.final_table.as<int64_t>(), shape.grid, D.n_ops, D.n_groups, P.ops_dev.as<PgAccOp>(),
                         ctx.stats.as<unsigned long long>(), reduce);
      PG_HIP(hipGetLastError());
      if (profile) PG_HIP(hipEventRecord(ctx.ev[2], ctx.stream));
    }
    // Segment-level group trim on the device (pg_kernels_trim.hip): dense tables without auxiliary state whose first ORDER BY expression is
    // a dictionary group column or an int64 accumulator row, and far more slots than trimSize: the survivors are selected in HBM and only
    // their rows are copied.  Everything else copies the table and trims at assembly.
    int trim_cap = 0, trim_key_op = -1;
    bool trim_whole_class = false;
    const int tgrid_adm = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)num_cus() * 4, ((int64_t)D.n_groups + 1023) / 1024));
    if (opt.prefix_groups_out) {   // prefix pass of the admission: keys of its first-docId row stay in the context, the count of its groups comes back
      ThreadCtx::grow(ctx.trim_keys, (size_t)D.n_groups * 8);
      if (!ctx.trim_ctrl.ptr) ctx.trim_ctrl.alloc((size_t)PG_TRIM_CTRL_WORDS * 4, true);
      PG_HIP(hipMemsetAsync(ctx.trim_ctrl.ptr, 0, (size_t)PG_TRIM_CTRL_WORDS * 4, ctx.stream));
      PgTrimArgs ta;
      memset(&ta, 0, sizeof(ta));
      ta.table = ctx.final_table.as<int64_t>();
      ta.G = D.n_groups;
      ta.n_ops = D.n_ops;
      ta.exist_op = P.first_doc_op;
      ta.exist_ident = pg_acc_identity(PG_ACC_MIN, 0);
      ta.key_op = P.first_doc_op;
      ta.key_mult = 1; ta.key_card = 1;
      ta.keys = ctx.trim_keys.as<uint64_t>();
      ta.ctrl = ctx.trim_ctrl.as<uint32_t>();
      pg_trim_launch_keys(&ta, tgrid_adm, ctx.stream);
      PG_HIP(hipGetLastError());
      if (profile) PG_HIP(hipEventRecord(ctx.ev[2], ctx.stream));
      PG_HIP(hipMemcpyAsync(host_out, ctx.trim_ctrl.ptr, 8, hipMemcpyDeviceToHost, ctx.stream));
      PG_HIP(hipMemcpyAsync(host_out + 1, ctx.final_table.as<int64_t>() + n_out, (size_t)PG_MAX_STATS * 8, hipMemcpyDeviceToHost, ctx.stream));
    }
    bool admission_select = false, admission_own = false;
    if (opt.admit_on_device > 0 && !hashed && D.n_aux == 0 && !keep_table) {   // the pass over the segment: rows of the `limit` smallest keys
      admission_select = true;
      trim_cap = opt.admit_on_device;
    }
    // a one-pass plan that carries its own MIN(docId) row (no prefix pass applied): the same selection over that row — the host's nth_element
    // over 10^6 first docIds was 8.6 ms of a 13.8 ms query, behind 24 MB of table copies (profiles/r05_num_groups_limit_latency.txt)
    if (!admission_select && !opt.prefix_groups_out && !opt.admit && !opt.raw_out && P.first_doc_op >= 0 && !hashed && D.n_aux == 0 && !keep_table && !D.mv &&
        !direct_out && !knobs().no_device_trim && (int64_t)P.num_groups_limit * 2 + 64 <= (int64_t)D.n_groups) {
      admission_select = admission_own = true;
      trim_cap = P.num_groups_limit;
    }
    if (!admission_select && !opt.prefix_groups_out && P.trim_size > 0 && !hashed && D.n_aux == 0 && P.first_doc_op < 0 && !opt.admit && !opt.raw_out && !keep_table && !direct_out &&
        P.exist_op >= 0 && !D.mv && !knobs().no_device_trim && (int64_t)D.n_groups >= 8 * ((int64_t)P.trim_size + 4096)) {
      const pg_order_by& ob = P.order_by[0];
      bool ok = true;
      if (ob.kind == PG_ORDER_BY_GROUP_KEY) {
        ok = !P.raw_group && !((size_t)ob.index < P.group_vdict.size() && P.group_vdict[(size_t)ob.index]);
      } else {
        const AggOut& ao = P.aggs[(size_t)ob.index];
        const bool plain_row = ao.op_a >= 0 && ao.sum_limbs <= 0 &&
                               (ao.function == PG_AGG_COUNT || D.ops[ao.op_a].is_float == PG_ACCV_INT ||
                                ((ao.function == PG_AGG_MIN || ao.function == PG_AGG_MAX) && D.ops[ao.op_a].is_float == PG_ACCV_DOUBLE));
        ok = plain_row && (ao.function == PG_AGG_COUNT || ao.function == PG_AGG_SUM || ao.function == PG_AGG_MIN || ao.function == PG_AGG_MAX);
        trim_key_op = ok ? ao.op_a : -1;
      }
      if (ok) {
        trim_whole_class = P.order_by.size() > 1;
        trim_cap = P.trim_size + 4096;
      }
    }
    if (opt.prefix_groups_out) {
      // (nothing of the table is copied: the caller only wants the count)
    } else if (trim_cap > 0) {
      ThreadCtx::grow(ctx.trim_keys, (size_t)D.n_groups * 8);
      ThreadCtx::grow(ctx.trim_out, (size_t)(D.n_ops + 1) * (size_t)trim_cap * 8);
      if (!ctx.trim_ctrl.ptr) ctx.trim_ctrl.alloc((size_t)PG_TRIM_CTRL_WORDS * 4, true);
      PgTrimArgs ta;
      memset(&ta, 0, sizeof(ta));
      ta.table = ctx.final_table.as<int64_t>();
      ta.G = D.n_groups;
      ta.n_ops = D.n_ops;
      ta.keys = ctx.trim_keys.as<uint64_t>();
      ta.ctrl = ctx.trim_ctrl.as<uint32_t>();
      ta.cap = trim_cap;
      ta.out_table = ctx.trim_out.as<int64_t>();
      ta.out_gids = ctx.trim_out.as<int64_t>() + (size_t)D.n_ops * (size_t)trim_cap;
      const int tgrid = tgrid_adm;
      if (admission_own) {      // keys = this table's own first-docId row
        PG_HIP(hipMemsetAsync(ctx.trim_ctrl.ptr, 0, (size_t)PG_TRIM_CTRL_WORDS * 4, ctx.stream));
        ta.exist_op = ta.key_op = P.first_doc_op;
        ta.exist_ident = pg_acc_identity(PG_ACC_MIN, 0);
        ta.key_mult = 1; ta.key_card = 1;
        ta.k = trim_cap;
        pg_trim_launch(&ta, tgrid, ctx.stream);
      } else if (admission_select) {   // keys, histogram of their first byte and the group count are the prefix pass's (same thread, same stream)
        ta.k = opt.admit_on_device;
        ta.key_mult = 1; ta.key_card = 1;
        pg_trim_launch_select(&ta, tgrid, ctx.stream);
      } else {
        const pg_order_by& ob = P.order_by[0];
        PG_HIP(hipMemsetAsync(ctx.trim_ctrl.ptr, 0, (size_t)PG_TRIM_CTRL_WORDS * 4, ctx.stream));
        ta.exist_op = P.exist_op;
        ta.exist_ident = pg_acc_identity(D.ops[P.exist_op].fn, 0);
        ta.key_op = trim_key_op;
        ta.descending = ob.ascending ? 0 : 1;
        ta.key_mult = ob.kind == PG_ORDER_BY_GROUP_KEY ? D.gcols[ob.index].mult : 1;
        ta.key_card = ob.kind == PG_ORDER_BY_GROUP_KEY ? P.group_cards[ob.index] : 1;
        ta.k = P.trim_size;
        ta.take_whole_tie_class = trim_whole_class ? 1 : 0;
        pg_trim_launch(&ta, tgrid, ctx.stream);
      }
      PG_HIP(hipGetLastError());
      if (profile) PG_HIP(hipEventRecord(ctx.ev[2], ctx.stream));   // the selection is part of the query's device time
      // compact block | statistics | counters, where the whole table would have gone
      const size_t block_words = (size_t)(D.n_ops + 1) * (size_t)trim_cap;
      PG_HIP(hipMemcpyAsync(host_out, ctx.trim_out.ptr, block_words * 8, hipMemcpyDeviceToHost, ctx.stream));
      PG_HIP(hipMemcpyAsync(host_out + block_words, ctx.final_table.as<int64_t>() + n_out, (size_t)PG_MAX_STATS * 8, hipMemcpyDeviceToHost, ctx.stream));
      PG_HIP(hipMemcpyAsync(host_out + block_words + PG_MAX_STATS, ctx.trim_ctrl.ptr, 32, hipMemcpyDeviceToHost, ctx.stream));
    } else if (!direct_out) {
      PG_HIP(hipMemcpyAsync(host_out, ctx.final_table.ptr, out_bytes, hipMemcpyDeviceToHost, ctx.stream));
    }
    if (fused_finish) {
      // (pg_finish_fused_kernel wrote the final values into the block)
    } else if (final_distinct) {
      const int G1 = std::max(D.n_groups, 1);
      ThreadCtx::grow(ctx.aux_summary, summary_bytes);
      for (int x = 0; x < D.n_aux; x++) {
        const bool set = D.aux[x].kind == PG_AUX_DICT_SET;
        const int words = set ? D.aux[x].stride : D.aux[x].stride / 4, lm = set ? 0 : D.aux[x].log2m;
        if (!set && !ctx.hll_small[lm].ptr) {
          std::vector<long long> t;
          hll_small_range_table(lm, t, ctx.hll_alpha_mm[lm]);
          ctx.hll_small[lm] = upload_vector(t);
        }
        hipLaunchKernelGGL(pg_aux_finish_kernel, dim3((unsigned)((G1 + 3) / 4)), dim3(256), 0, ctx.stream, aux_final[(size_t)x], D.aux[x].kind, words, G1,
                           ctx.hll_alpha_mm[lm], 1 << lm, ctx.hll_small[lm].as<long long>(), ctx.aux_summary.as<long long>() + (size_t)x * (size_t)G1);
      }
      PG_HIP(hipGetLastError());
      PG_HIP(hipMemcpyAsync(aux_host, ctx.aux_summary.ptr, summary_bytes, hipMemcpyDeviceToHost, ctx.stream));
    } else if (aux_total) {
      PG_HIP(hipMemcpyAsync(aux_host, ctx.aux.ptr, aux_total, hipMemcpyDeviceToHost, ctx.stream));
    }
    if (keep_table) {   // PG_QUERY_FLAG_KEEP_DEVICE_TABLE: the dense table, its counters and states stay in HBM with the result
      kept = std::make_unique<DeviceTable>();
      kept->table.alloc(out_bytes + 2 * 8);
      PG_HIP(hipMemcpyAsync(kept->table.ptr, ctx.final_table.ptr, out_bytes, hipMemcpyDeviceToDevice, ctx.stream));
      if (aux_total) {
        kept->aux.alloc(aux_total);
        PG_HIP(hipMemcpyAsync(kept->aux.ptr, ctx.aux.ptr, aux_total, hipMemcpyDeviceToDevice, ctx.stream));
      }
    }
    const double t_queued = now_ms();
    stream_wait(ctx, cancel);
    t_synced = now_ms();
    t_queued_at = t_queued;
    if (opt.prefix_groups_out) {
      *opt.prefix_groups_out = (int64_t)reinterpret_cast<const uint32_t*>(host_out)[0];
      memcpy(stats_host, host_out + 1, sizeof(stats_host));
    } else if (trim_cap > 0) {
      const size_t block_words = (size_t)(D.n_ops + 1) * (size_t)trim_cap;
      const uint32_t* tc = reinterpret_cast<const uint32_t*>(host_out + block_words + PG_MAX_STATS);
      memcpy(stats_host, host_out + block_words, sizeof(stats_host));
      if (tc[3]) {   // the tie class of several ORDER BY expressions did not fit the block: the whole table after all
        PG_HIP(hipMemcpyAsync(host_out, ctx.final_table.ptr, (size_t)n_out * 8, hipMemcpyDeviceToHost, ctx.stream));
        stream_wait(ctx, cancel);
        if (n_out) memcpy(table.data(), host_out, (size_t)n_out * 8);
      } else {
        const int64_t n_exist = tc[0];
        const int64_t n_sel = trim_whole_class ? (int64_t)tc[5] + (int64_t)tc[2] : std::min<int64_t>(admission_select ? trim_cap : P.trim_size, n_exist);
        // survivors in group-id order (the device appends them as its wavefronts come by): a two-pass 16-bit radix sort of their positions —
        // std::sort with an indirect comparison took several milliseconds for the 100 000 groups of a default numGroupsLimit
        const int64_t* gsel = host_out + (size_t)D.n_ops * (size_t)trim_cap;
        std::vector<int32_t> perm((size_t)n_sel), tmp((size_t)n_sel);
        for (int64_t i = 0; i < n_sel; i++) tmp[(size_t)i] = (int32_t)i;
        for (int pass = 0; pass < 2; pass++) {
          std::vector<uint32_t> count((size_t)65537, 0u);
          const std::vector<int32_t>& src = pass == 0 ? tmp : perm;
          std::vector<int32_t>& dst = pass == 0 ? perm : tmp;
          const int shift = 16 * pass;
          for (int64_t i = 0; i < n_sel; i++) count[(size_t)(((uint64_t)gsel[src[(size_t)i]] >> shift) & 0xFFFFu) + 1]++;
          for (size_t b = 1; b <= 65536; b++) count[b] += count[b - 1];
          for (int64_t i = 0; i < n_sel; i++) dst[count[(size_t)(((uint64_t)gsel[src[(size_t)i]] >> shift) & 0xFFFFu)]++] = src[(size_t)i];
        }
        perm.swap(tmp);   // (dense key spaces are below 2^31 groups: 32 bits of the group id order them)
        hash_keys_host.resize((size_t)n_sel);
        table.assign((size_t)n_sel * (size_t)D.n_ops, 0);
        for (int64_t i = 0; i < n_sel; i++) {
          hash_keys_host[(size_t)i] = gsel[perm[(size_t)i]];
          for (int o = 0; o < D.n_ops; o++) table[(size_t)o * (size_t)n_sel + (size_t)i] = host_out[(size_t)o * (size_t)trim_cap + (size_t)perm[(size_t)i]];
        }
        compact_groups = n_sel;
        groups_found = n_exist;
      }
    } else {
      if (n_out) memcpy(table.data(), host_out, (size_t)n_out * 8);
      memcpy(stats_host, host_out + n_out, sizeof(stats_host));
    }
    ctx.stats_dirty = false;    // the reduce kernel left them zero
    if (p2_ran && ctx.p2_ctrl_host[1])
      fail(PG_ERR_INTERNAL, "partition pipeline ran out of chunks (%u claimed, %d sized)", ctx.p2_ctrl_host[0], D.p2_capacity);
    if (oct_pruned) {
      const uint32_t* f = ctx.p2_ctrl_host + 4;
      if (f[1] || f[3]) fail(PG_ERR_INTERNAL, "pruned-offer passes: %s (last pass: %u chunks, %u stream entries)", f[1] ? "out of chunks" : "survivor stream overflow", f[0], f[2]);
    }
    if (hashed) {
      // the occupied slots of every bucket's hash table: raw keys + [n_ops][groups] accumulators → a compact table
      unsigned long long cnt[2] = {0, 0};
      PG_HIP(hipMemcpyAsync(cnt, ctx.hash_count.ptr, 16, hipMemcpyDeviceToHost, ctx.stream));
      PG_HIP(hipStreamSynchronize(ctx.stream));
      if (cnt[1] == 2) fail(PG_ERR_UNSUPPORTED, "raw group key Long.MAX_VALUE is outside the GPU path");
      if (cnt[1]) fail(PG_ERR_UNSUPPORTED, "more distinct group keys in one hash bucket than its LDS table holds (%d slots, %d buckets)", D.hash_cap, D.radix_buckets);
      hash_groups = (int64_t)cnt[0];
      hash_keys_host.resize((size_t)hash_groups);
      table.assign((size_t)hash_groups * (size_t)D.n_ops, 0);
      if (hash_groups) {
        PG_HIP(hipMemcpyAsync(hash_keys_host.data(), ctx.hash_keys.ptr, (size_t)hash_groups * 8, hipMemcpyDeviceToHost, ctx.stream));
        for (int o = 0; o < D.n_ops; o++)
          PG_HIP(hipMemcpyAsync(table.data() + (size_t)o * (size_t)hash_groups, ctx.hash_acc.as<int64_t>() + (size_t)o * (size_t)D.hash_out_cap,
                                (size_t)hash_groups * 8, hipMemcpyDeviceToHost, ctx.stream));
        PG_HIP(hipStreamSynchronize(ctx.stream));
      }
    }
  } else {
    if (profile) PG_HIP(hipEventRecord(ctx.ev[2], ctx.stream));
    PG_HIP(hipStreamSynchronize(ctx.stream));
    for (int o = 0; o < D.n_ops; o++)
      for (int64_t g = 0; g < D.n_groups; g++) table[(size_t)(o * (int64_t)D.n_groups + g)] = pg_acc_identity(D.ops[o].fn, D.ops[o].is_float);
    if (host_aux_bytes) memset(aux_host, 0, host_aux_bytes);
    // (final_distinct: an empty HyperLogLog has cardinality round(m * ln(m / m)) = 0, an empty set size 0: the zeroes stand)
  }

  if (keep_table && !has_docs) {   // an empty doc space still yields a (neutral) table to merge with
    kept = std::make_unique<DeviceTable>();
    kept->table.alloc(out_bytes + 2 * 8, true);
    if (n_out) kept->table.upload(table.data(), (size_t)n_out * 8);
    if (aux_total) kept->aux.alloc(aux_total, true);
  }
  auto res = std::make_unique<Result>();
  HostTable H;
  H.table = std::move(table);
  memcpy(H.stats, stats_host, sizeof(stats_host));
  H.aux = final_distinct ? nullptr : aux_host;
  H.aux_summary = final_distinct ? reinterpret_cast<const uint64_t*>(aux_host) : nullptr;
  H.aux_block = final_distinct ? nullptr : out_block;
  H.hashed = hashed || compact_groups >= 0;   // a table trimmed on the device reads like a hashed one: compact rows + the raw key of each
  H.hash_groups = compact_groups >= 0 ? compact_groups : hash_groups;
  H.groups_found = groups_found;
  H.hash_keys = std::move(hash_keys_host);
  H.full_scan_entries = P.full_scan_entries;
  H.total_docs = seg.total_docs;
  if (opt.admit) { H.admit_first = opt.admit->first; H.admit_limit = opt.admit->limit; }
  if (opt.admit_on_device > 0) H.admit_limit = opt.admit_on_device;   // (the rows that came back ARE the admitted groups)
  if (opt.prefix_groups_out) {   // the caller wanted the count only
    if (profile) {
      float a = 0, b = 0;
      PG_HIP(hipEventElapsedTime(&a, ctx.ev[0], ctx.ev[1]));
      PG_HIP(hipEventElapsedTime(&b, ctx.ev[1], ctx.ev[2]));
      res->stats.device_ms_aggregate = a;
      res->stats.device_ms_reduce = b;
      res->stats.device_ms_total = a + b;
    }
    return res;
  }
  if (opt.raw_out) {   // the caller wants the table, not groups (execute_limit_by_prefix)
    if (profile) {
      float a = 0, b = 0;
      PG_HIP(hipEventElapsedTime(&a, ctx.ev[0], ctx.ev[1]));
      PG_HIP(hipEventElapsedTime(&b, ctx.ev[1], ctx.ev[2]));
      res->stats.device_ms_aggregate = a;
      res->stats.device_ms_reduce = b;
      res->stats.device_ms_total = a + b;
    }
    *opt.raw_out = std::move(H);
    return res;
  }
  int64_t exact_entries = -1;
  int32_t stats_path = 0;
  // exact numEntriesScannedInFilter of leapfrogged shapes: by default up to 2^27 docs where the device counts it (one filter launch per leaf + the
  // tile automaton), up to 2^22 docs where the host walks the bitmaps (a multiple of the query beyond that), on request at any size
  const bool want_exact = !P.stats_exact && !(q.flags & PG_QUERY_FLAG_APPROX_FILTER_STATS) && ((q.flags & PG_QUERY_FLAG_EXACT_FILTER_STATS) || exact_stats_by_default(P));
  if (!P.stats_exact && want_exact) {
    check_cancel(cancel, &ctx);
    exact_entries = exact_entries_scanned(P, ctx, cancel, &stats_path);
    // the merged tables carry the count in the statistics tail: fold the exact value in as "full scan entries" of this segment
    H.full_scan_entries = exact_entries;
    for (int i = 1; i < PG_MAX_STATS; i++) H.stats[i] = 0;
  }
  const double t_before_assembly = now_ms();
  assemble_result(*res, P, q.n_group_by, q.n_aggregations, H);
  fill_result_schema(seg, q, *res);
  {
    const bool trace = knobs().trace_host;   // debugging knob: where the host time of a query goes
    if (trace)
      fprintf(stderr, "[pg] %s: plan %.3f ms, queue %.3f, wait %.3f, unpack %.3f, assembly %.3f\n", kname, t_plan - t0, t_queued_at - t_plan,
              t_synced - t_queued_at, t_before_assembly - t_synced, now_ms() - t_before_assembly);
  }
  if (exact_entries >= 0) res->stats.stats_exact = 1;
  res->stats.filter_stats_path = stats_path;
  snprintf(res->stats.kernel, sizeof(res->stats.kernel), "%s", kname);
  if (profile) {
    float a = 0, b = 0;
    PG_HIP(hipEventElapsedTime(&a, ctx.ev[0], ctx.ev[1]));
    PG_HIP(hipEventElapsedTime(&b, ctx.ev[1], ctx.ev[2]));
    res->stats.device_ms_aggregate = a;
    res->stats.device_ms_filter = 0;
    res->stats.device_ms_reduce = b;
    res->stats.device_ms_total = a + b;
  }
  if (kept) {
    kept->plan = plan;
    kept->device = seg.device;
    kept->n_group_by = q.n_group_by;
    kept->n_aggregations = q.n_aggregations;
    kept->n_out = n_out;
    kept->aux_total = aux_total;
    kept->full_scan_entries = exact_entries >= 0 ? exact_entries : P.full_scan_entries;
    if (exact_entries >= 0) {   // the per-scan candidate counters of the kept table are superseded by the exact count
      PG_HIP(hipMemsetAsync(kept->table.as<int64_t>() + n_out + 1, 0, (PG_MAX_STATS - 1) * 8, ctx.stream));   // (stream-ordered behind the copy into the kept table)
      PG_HIP(hipStreamSynchronize(ctx.stream));
    }
    kept->num_total_docs = seg.total_docs;
    kept->sum_max_abs = P.sum_max_abs;
    kept->has_digit_sums = P.has_digit_sums;
    res->dev = std::move(kept);
  }
  res->stats.host_ms_plan = (float)(t_plan - t0);
  res->stats.host_ms_total = (float)(now_ms() - t0);
  return res;
}

// The small-range branch of HyperLogLog#cardinality, round(m * ln(m / zeros)), for zeros = 0 .. m (host libm; Math.round = floor(x + 0.5)),
// and alphaMM — what pg_aux_finish_kernel needs besides the registers
static void hll_small_range_table(int log2m, std::vector<long long>& t, double& alpha_mm) {
  const int m = 1 << log2m;
  if (m == 16) alpha_mm = 0.673 * m * m;
  else if (m == 32) alpha_mm = 0.697 * m * m;
  else if (m == 64) alpha_mm = 0.709 * m * m;
  else alpha_mm = (0.7213 / (1 + 1.079 / m)) * m * m;
  t.assign((size_t)m + 1, 0);
  t[0] = INT64_MAX;   // no empty register: linearCounting's m * log(m / 0.0) = Infinity, Math.round(Infinity) = Long.MAX_VALUE
  for (int v = 1; v <= m; v++) t[(size_t)v] = (long long)std::floor(m * std::log(m / (double)v) + 0.5);
  t[0] = INT64_MAX;   // Math.round(+Infinity) = Long.MAX_VALUE (cannot occur: estimate <= 2.5 m implies zero registers exist)
}

// HyperLogLog#cardinality of one register row (host; the order-by value of a DISTINCTCOUNTHLL state)
int64_t hll_cardinality(const uint8_t* regs, int log2m) {
  const int m = 1 << log2m;
  std::vector<long long> small;
  double alpha_mm = 0;
  hll_small_range_table(log2m, small, alpha_mm);
  double register_sum = 0;
  int zeros = 0;
  for (int j = 0; j < m; j++) { register_sum += 1.0 / (double)(1ULL << regs[j]); zeros += regs[j] == 0; }
  const double estimate = alpha_mm * (1 / register_sum);
  return estimate <= (5.0 / 2.0) * m ? (int64_t)small[(size_t)zeros] : (int64_t)std::floor(estimate + 0.5);
}

static void assemble_result(Result& res, const CompiledPlan& P, int32_t n_group_by, int32_t n_aggregations, HostTable& H) {
  const PgQueryPlan& D = P.dev;
  const bool hashed = H.hashed;
  std::vector<int64_t>& table = H.table;
  const bool trace = knobs().trace_host;
  const double ta0 = trace ? now_ms() : 0.0;
  double ta1 = 0, ta2 = 0;
  const pg_exec_stats keep = res.stats;   // timings / kernel name survive a re-assembly after a merge
  fill_stats(res.stats, P, H.full_scan_entries, H.total_docs, H.stats);
  if ((P.dev.pipe_fit || P.dev.pipe_general) && H.total_docs > 0) {   // (see spec_shape)
    int64_t cand = 0;
    for (int i = 1; i < P.n_stat_slots; i++) cand += (int64_t)H.stats[i];
    P.observed_candidate_permille.store((int)std::min<int64_t>(1000, cand * 1000 / H.total_docs), std::memory_order_relaxed);
    P.observed_match_permille.store((int)std::min<int64_t>(1000, (int64_t)H.stats[0] * 1000 / H.total_docs), std::memory_order_relaxed);
  }
  res.stats.star_tree_index = P.star_tree_index;
  res.stats.device_ms_total = keep.device_ms_total; res.stats.device_ms_filter = keep.device_ms_filter;
  res.stats.device_ms_aggregate = keep.device_ms_aggregate; res.stats.device_ms_reduce = keep.device_ms_reduce;
  res.stats.host_ms_plan = keep.host_ms_plan; res.stats.host_ms_total = keep.host_ms_total;
  memcpy(res.stats.kernel, keep.kernel, sizeof(keep.kernel));
  res.group_dict_ids.clear();
  res.group_values.clear();
  res.aggs.clear();
  // ---- assemble groups: a group exists iff its hidden COUNT is > 0 (ArrayBasedHolder flags / map entries) ----------------
  const int64_t G = hashed ? H.hash_groups : (H.keys ? H.keys->n_groups : D.n_groups);   // hashed key space: the compact table of the groups found
  // key space of the table: the plan's own (the segment's dictionaries), or the union a value-keyed merge built (pg_comm.cpp)
  auto key_mult = [&](int j) -> int64_t { return H.keys ? H.keys->mults[(size_t)j] : D.gcols[j].mult; };
  auto key_card = [&](int j) -> int32_t { return H.keys ? H.keys->cards[(size_t)j] : P.group_cards[(size_t)j]; };
  auto key_vdict = [&](int j) -> const Column* {
    if (H.keys) return H.keys->dicts[(size_t)j].get();
    return (size_t)j < P.group_vdict.size() ? P.group_vdict[(size_t)j] : nullptr;
  };
  const int64_t matched = (int64_t)H.stats[0];
  const bool ex_stats = P.exist_op == kCountFromStats;
  const int64_t* ex = ex_stats ? nullptr : table.data() + (size_t)P.exist_op * G;
  const int64_t ex_ident = ex_stats ? 0 : pg_acc_identity(D.ops[P.exist_op].fn, 0);
  auto exists = [&](int64_t g) { return ex_stats ? matched > 0 : ex[g] != ex_ident; };   // COUNT: != 0; MIN/MAX over INT: left its identity
  auto count_of = [&](int32_t op, int64_t g) { return op == kCountFromStats ? matched : table[(size_t)op * G + g]; };
  std::vector<int64_t> gids;
  if (n_group_by == 0) gids.push_back(0);
  else if (hashed) { gids.resize((size_t)G); for (int64_t g = 0; g < G; g++) gids[(size_t)g] = g; }
  else {
    gids.reserve((size_t)std::min<int64_t>(G, 1 << 20));
    for (int64_t g = 0; g < G; g++) if (exists(g)) gids.push_back(g);
  }
  const int64_t groups_limit = H.admit_limit > 0 ? (int64_t)H.admit_limit : (int64_t)P.num_groups_limit;
  bool limit_reached = n_group_by > 0 && (H.groups_found >= 0 ? H.groups_found : (int64_t)gids.size()) >= groups_limit;
  if (n_group_by > 0 && (int64_t)gids.size() > groups_limit) {
    // keep the numGroupsLimit groups whose first matching docId is smallest (= the keys the reference admits in docId order)
    if (P.dev.mv)   // which keys the reference admits depends on the entry order inside the docs: left to the Java plan
      fail(PG_ERR_UNSUPPORTED, "multi-value group-by found %zu groups, more than numGroupsLimit (%d)", gids.size(), P.num_groups_limit);
    if (P.first_doc_op < 0 && !H.admit_first) fail(PG_ERR_INTERNAL, "plan lacks the first-docId accumulator");
    const int64_t* first = H.admit_first ? H.admit_first : table.data() + (size_t)P.first_doc_op * G;
    std::nth_element(gids.begin(), gids.begin() + groups_limit, gids.end(), [&](int64_t a, int64_t b) { return first[a] < first[b]; });
    gids.resize((size_t)groups_limit);
    std::sort(gids.begin(), gids.end());
  }
  auto op_double = [&](int o, int64_t g) -> double {
    const PgAccOp& op = D.ops[o];
    int64_t v = table[(size_t)o * G + g];
    const bool empty = !exists(g);
    switch (op.fn) {
      case PG_ACC_COUNT: return (double)v;
      case PG_ACC_SUM:
        if (op.is_float == PG_ACCV_DOUBLE) { double d; memcpy(&d, &v, 8); return d; }
        return (double)v;
      case PG_ACC_MIN:
        if (empty) return INFINITY;    // MinAggregationFunction default holder value
        return op.is_float ? order_key_to_double(v) : (double)v;
      default:
        if (empty) return -INFINITY;   // MaxAggregationFunction.java:37
        return op.is_float ? order_key_to_double(v) : (double)v;
    }
  };
  // SUM kept in fixed-point / two-digit limbs: combined and rounded to double once (exact sum, correctly rounded)
  auto sum_double = [&](const AggOut& ao, int64_t g) -> double {
    if (ao.sum_limbs <= 0) return op_double(ao.op_a, g);
    int64_t limbs[4] = {0, 0, 0, 0};
    for (int j = 0; j < ao.sum_limbs; j++) limbs[j] = table[(size_t)(ao.op_a + j) * G + g];
    return limbs_to_double(limbs, ao.sum_limbs, ao.fx_q);
  };
  // DISTINCTCOUNT / DISTINCTCOUNTHLL states: the region of aggregation state `x` with its replicas merged into replica 0 (set union / register
  // maximum), once — the trim below orders by the states' final values, the assembly extracts them
  std::vector<char> aux_merged((size_t)std::max(D.n_aux, 1), 0);
  auto aux_region = [&](int x) -> uint8_t* {
    size_t off = 0;
    for (int y = 0; y < x; y++) off += P.aux_bytes[y];
    const PgAuxOp& A = D.aux[x];
    if (!aux_merged[(size_t)x]) {
      for (int rr = 1; rr < A.n_rep; rr++) {
        uint8_t* dst = H.aux + off;
        const uint8_t* src = H.aux + off + (size_t)rr * (size_t)A.rep_bytes;
        if (A.kind == PG_AUX_DICT_SET) for (int64_t b = 0; b < A.rep_bytes; b++) dst[b] |= src[b];
        else for (int64_t b = 0; b < A.rep_bytes; b++) dst[b] = src[b] > dst[b] ? src[b] : dst[b];
      }
      aux_merged[(size_t)x] = 1;
    }
    return H.aux + off;
  };
  // AggregationFunction#extractFinalResult of such a state: the set's size (DistinctCountAggregationFunction), HyperLogLog#cardinality
  auto aux_final_value = [&](const AggOut& ao, int64_t g) -> int64_t {
    if (H.aux_summary) return (int64_t)H.aux_summary[(size_t)ao.aux * (size_t)std::max(D.n_groups, 1) + (size_t)g];
    const PgAuxOp& A = D.aux[ao.aux];
    const uint8_t* region = aux_region(ao.aux);
    if (A.kind == PG_AUX_DICT_SET) {
      const uint32_t* w = reinterpret_cast<const uint32_t*>(region) + (size_t)g * A.stride;
      int64_t n = 0;
      for (int32_t k = 0; k < A.stride; k++) n += __builtin_popcount(w[k]);
      return n;
    }
    return hll_cardinality(region + (size_t)g * A.stride, ao.log2m);
  };
  // ---- segment-level group trim (GroupByOperator.java:120-133 -> TableResizer#trimInSegmentResults :327-351): more groups than trimSize and
  //      an ORDER BY: keep the trimSize groups that sort first.  Order-by values as the extractors of TableResizer.java:406-445 give them:
  //      a group key's value, an aggregation's final result (COUNT long; SUM / MIN / MAX double; AVG sum / count; MINMAXRANGE max - min).
  //      (Tables trimmed on the device arrive here already compact — device_trim below — and pass through: ng <= trimSize.)
  if (n_group_by > 0 && P.trim_size > 0 && (int64_t)gids.size() > (int64_t)P.trim_size) {
    struct OV { int type; int64_t l; double d; const uint8_t* b; int64_t blen; };   // type 0 long, 1 double, 2 BYTES (unsigned bytes), 3 STRING
    // String.compareTo orders UTF-16 code units (TableResizer's comparators on STRING keys); the UTF-8 byte order differs only where a
    // supplementary character (lead byte F0..F4: surrogates D800..DFFF in UTF-16) meets U+E000..U+FFFF (lead byte EE / EF): those two
    // lead bytes sort behind F0..F4
    auto utf16_unit_order = [](const uint8_t* a, int64_t alen, const uint8_t* b, int64_t blen) {
      const int64_t m = std::min(alen, blen);
      for (int64_t i = 0; i < m; i++) {
        if (a[i] == b[i]) continue;
        const int x = a[i] == 0xEE || a[i] == 0xEF ? a[i] + 0x10 : a[i], y = b[i] == 0xEE || b[i] == 0xEF ? b[i] + 0x10 : b[i];
        return x < y ? -1 : 1;
      }
      return alen < blen ? -1 : (alen > blen ? 1 : 0);
    };
    const size_t n = gids.size(), n_ob = P.order_by.size();
    std::vector<OV> vals(n * n_ob);
    for (size_t k = 0; k < n_ob; k++) {
      const pg_order_by& ob = P.order_by[k];
      if (ob.kind == PG_ORDER_BY_AGGREGATION) {
        const AggOut& ao = P.aggs[(size_t)ob.index];
        for (size_t i = 0; i < n; i++) {
          OV& v = vals[i * n_ob + k];
          const int64_t g = gids[i];
          v = OV{1, 0, 0.0, nullptr, 0};
          if (ao.aux >= 0) { v.type = 0; v.l = aux_final_value(ao, g); continue; }   // DISTINCTCOUNT Integer / DISTINCTCOUNTHLL Long
          switch (ao.function) {
            case PG_AGG_COUNT: v.type = 0; v.l = count_of(ao.op_a, g); break;
            case PG_AGG_SUM: v.d = sum_double(ao, g); break;
            case PG_AGG_AVG: { const int64_t c = count_of(ao.op_b, g); v.d = c == 0 ? -INFINITY : sum_double(ao, g) / (double)c; break; }
            case PG_AGG_MINMAXRANGE: v.d = op_double(ao.op_b, g) - op_double(ao.op_a, g); break;
            default: v.d = op_double(ao.op_a, g); break;   // MIN / MAX
          }
        }
        continue;
      }
      const int j = ob.index;
      const Column* vd = key_vdict(j);
      for (size_t i = 0; i < n; i++) {
        OV& v = vals[i * n_ob + k];
        v = OV{0, 0, 0.0, nullptr, 0};
        if (P.raw_group) { v.l = (int64_t)((uint64_t)H.hash_keys[(size_t)gids[i]] ^ (1ULL << 63)); continue; }
        const int64_t raw = hashed ? H.hash_keys[(size_t)gids[i]] : gids[i];
        const int64_t id = (raw / key_mult(j)) % key_card(j);
        if (!vd) { v.l = id; continue; }   // a sorted dictionary: dictIds order as the values do
        if (vd->vdict_kind == 4) {
          v.type = vd->data_type == PG_TYPE_STRING ? 3 : 2;
          v.b = vd->vdict_bytes.data() + vd->vdict_bytes_off[(size_t)id];
          v.blen = vd->vdict_bytes_off[(size_t)id + 1] - vd->vdict_bytes_off[(size_t)id];
        } else if (vd->vdict_kind <= 1) {
          v.l = vdict_value_of_key(vd->vdict_keys[(size_t)id], vd->vdict_kind, nullptr);
        } else {
          v.type = 1;
          (void)vdict_value_of_key(vd->vdict_keys[(size_t)id], vd->vdict_kind, &v.d);
        }
      }
    }
    auto dcmp = [](double a, double b) {   // Double.compare
      if (a < b) return -1;
      if (a > b) return 1;
      int64_t x, y;
      memcpy(&x, &a, 8); memcpy(&y, &b, 8);
      if (a != a) x = INT64_MAX;
      if (b != b) y = INT64_MAX;
      return x == y ? 0 : (x < y ? -1 : 1);
    };
    std::vector<int32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
    auto before = [&](int32_t ia, int32_t ib) {
      for (size_t k = 0; k < n_ob; k++) {
        const OV& a = vals[(size_t)ia * n_ob + k];
        const OV& b = vals[(size_t)ib * n_ob + k];
        int r;
        if (a.type == 0) r = a.l < b.l ? -1 : (a.l > b.l ? 1 : 0);
        else if (a.type == 1) r = dcmp(a.d, b.d);
        else if (a.type == 3) r = utf16_unit_order(a.b, a.blen, b.b, b.blen);
        else {
          const int64_t m = std::min(a.blen, b.blen);
          r = m ? memcmp(a.b, b.b, (size_t)m) : 0;
          if (r == 0) r = a.blen < b.blen ? -1 : (a.blen > b.blen ? 1 : 0);
        }
        if (r != 0) return P.order_by[k].ascending ? r < 0 : r > 0;
      }
      return ia < ib;
    };
    std::nth_element(order.begin(), order.begin() + P.trim_size, order.end(), before);
    order.resize((size_t)P.trim_size);
    std::sort(order.begin(), order.end());
    std::vector<int64_t> kept_gids(order.size());
    for (size_t i = 0; i < order.size(); i++) kept_gids[i] = gids[(size_t)order[i]];
    gids.swap(kept_gids);
  }
  const int32_t ng = (int32_t)gids.size();
  if (trace) ta1 = now_ms();
  res.num_groups = ng;
  res.group_key_type.assign((size_t)n_group_by, PG_GROUP_KEY_DICT_IDS);
  res.group_values.assign((size_t)n_group_by, {});
  res.group_bytes.assign((size_t)n_group_by, {});
  res.group_bytes_off.assign((size_t)n_group_by, {});
  res.group_dict_ids.assign((size_t)n_group_by, {});
  if (P.raw_group) {   // one raw INT / LONG column hashed by value: key = value ^ 2^63
    res.group_key_type[0] = PG_GROUP_KEY_LONG_VALUES;
    res.group_values[0].resize((size_t)ng);
    for (int32_t i = 0; i < ng; i++) res.group_values[0][(size_t)i] = (int64_t)((uint64_t)H.hash_keys[(size_t)gids[i]] ^ (1ULL << 63));
  }
  for (int j = 0; j < n_group_by && !P.raw_group; j++) {
    int64_t mult = key_mult(j);
    int32_t card = key_card(j);
    const Column* vd = key_vdict(j);
    if (vd && vd->vdict_kind == 4) {   // raw STRING / BYTES: the groups' byte strings from the virtual dictionary's values
      res.group_key_type[(size_t)j] = PG_GROUP_KEY_BYTES_VALUES;
      auto& bytes = res.group_bytes[(size_t)j];
      auto& boff = res.group_bytes_off[(size_t)j];
      boff.resize((size_t)ng + 1);
      int64_t total = 0;
      for (int32_t i = 0; i < ng; i++) {
        const int64_t raw = hashed ? H.hash_keys[(size_t)gids[i]] : gids[i];
        const size_t id = (size_t)((raw / mult) % card);
        boff[(size_t)i] = total;
        total += vd->vdict_bytes_off[id + 1] - vd->vdict_bytes_off[id];
      }
      boff[(size_t)ng] = total;
      bytes.resize((size_t)total);
      for (int32_t i = 0; i < ng; i++) {
        const int64_t raw = hashed ? H.hash_keys[(size_t)gids[i]] : gids[i];
        const size_t id = (size_t)((raw / mult) % card);
        if (boff[(size_t)i + 1] > boff[(size_t)i])
          memcpy(bytes.data() + boff[(size_t)i], vd->vdict_bytes.data() + vd->vdict_bytes_off[id], (size_t)(boff[(size_t)i + 1] - boff[(size_t)i]));
      }
      continue;
    }
    if (vd) {   // ids of a virtual dictionary: hand the values over (the caller holds no dictionary for this column)
      res.group_key_type[(size_t)j] = vd->vdict_kind <= 1 ? PG_GROUP_KEY_LONG_VALUES : PG_GROUP_KEY_DOUBLE_VALUES;
      auto& v = res.group_values[(size_t)j];
      v.resize((size_t)ng);
      for (int32_t i = 0; i < ng; i++) {
        const int64_t raw = hashed ? H.hash_keys[(size_t)gids[i]] : gids[i];
        v[(size_t)i] = vdict_value_of_key(vd->vdict_keys[(size_t)((raw / mult) % card)], vd->vdict_kind, nullptr);
      }
      continue;
    }
    auto& v = res.group_dict_ids[j];
    if (!hashed && !H.keys && (int64_t)ng == G && G <= (int64_t)1 << 22) {   // every group of the key space exists: the columns' dictIds come from the plan's cache
      std::call_once(P.full_keys_once, [&] {
        P.full_keys.assign((size_t)n_group_by, {});
        for (int jj = 0; jj < n_group_by; jj++) {
          const uint32_t m32 = (uint32_t)D.gcols[jj].mult, c32 = (uint32_t)P.group_cards[jj];
          auto& f = P.full_keys[(size_t)jj];
          f.resize((size_t)G);
          uint32_t low = 0, digit = 0;
          for (int64_t g = 0; g < G; g++) {
            f[(size_t)g] = (int32_t)digit;
            if (++low == m32) { low = 0; if (++digit == c32) digit = 0; }
          }
        }
      });
      v = P.full_keys[(size_t)j];
      continue;
    }
    v.resize((size_t)ng);
    if (!hashed && G <= (int64_t)0x7FFFFFFF) {   // dense key space: 32-bit arithmetic (a 64-bit division costs several times as much)
      // ids ascend: digit j of g + 1 follows from digit j of g without a division (an odometer: (low part, digit) advance together);
      // gaps between existing groups fall back to the division.  12 800 groups x 4 columns were 100 k divisions, ~0.1 ms
      const uint32_t m32 = (uint32_t)mult, c32 = (uint32_t)card;
      uint32_t prev = 0xFFFFFFFFu, low = 0, digit = 0;   // low = g % mult, digit = (g / mult) % card
      for (int32_t i = 0; i < ng; i++) {
        const uint32_t g = (uint32_t)gids[i];
        if (g == prev + 1u && prev != 0xFFFFFFFFu) {
          if (++low == m32) { low = 0; if (++digit == c32) digit = 0; }
        } else {
          low = g % m32;
          digit = (g / m32) % c32;
        }
        prev = g;
        v[i] = (int32_t)digit;
      }
      continue;
    }
    for (int32_t i = 0; i < ng; i++) {   // getKeys: col 0 least significant
      const int64_t raw = hashed ? H.hash_keys[(size_t)gids[i]] : gids[i];
      v[i] = (int32_t)((raw / mult) % card);
    }
  }
  if (n_group_by > 0) res.stats.num_groups_limit_reached = limit_reached ? 1 : 0;
  if (trace) ta2 = now_ms();

  res.aggs.resize((size_t)n_aggregations);
  for (int a = 0; a < n_aggregations; a++) {
    const AggOut& ao = P.aggs[a];
    AggResult& r = res.aggs[a];
    // only the components the result kind has are sized (all four were filled with zeros for every aggregation: 0.8 MB per query for
    // config 5's 12 800 groups, a third of the star-tree route's assembly time)
    for (int k = 0; k < 2; k++) { r.d[k].clear(); r.l[k].clear(); }
    if (ao.aux >= 0 && H.aux_summary) {   // PG_QUERY_FLAG_FINAL_DISTINCT: the final value from the device's two integers per group
      const uint64_t* sm = H.aux_summary + (size_t)ao.aux * (size_t)std::max(D.n_groups, 1);
      r.kind = PG_RESULT_LONG;
      r.l[0].resize((size_t)ng);
      for (int32_t i = 0; i < ng; i++) r.l[0][i] = (int64_t)sm[(size_t)gids[i]];
      continue;
    }
    if (ao.aux >= 0) {   // DISTINCTCOUNT / DISTINCTCOUNTHLL: extract the groups' regions
      const PgAuxOp& A = D.aux[ao.aux];
      const size_t off = (size_t)(aux_region(ao.aux) - H.aux);   // (replicas merged into replica 0)
      if (A.kind == PG_AUX_DICT_SET) {
        r.kind = PG_RESULT_DICTID_SET;
        r.set_sizes.assign((size_t)ng, 0);
        const uint32_t* words = reinterpret_cast<const uint32_t*>(H.aux + off);
        for (int32_t i = 0; i < ng; i++) {
          const uint32_t* w = words + (size_t)gids[i] * A.stride;
          for (int32_t k = 0; k < A.stride; k++) {
            uint32_t bitsw = w[k];
            while (bitsw) {
              const int b = __builtin_ctz(bitsw);
              r.set_ids.push_back(k * 32 + b);
              r.set_sizes[i]++;
              bitsw &= bitsw - 1;
            }
          }
        }
        if (ao.aux_col && ao.aux_col->vdict_kind >= 0 && ao.aux_col->vdict_kind <= 3) {   // a raw column through its virtual dictionary: the caller gets VALUES
          r.kind = PG_RESULT_VALUE_SET;
          r.set_value_kind = ao.aux_col->vdict_kind;
          r.l[0].resize(r.set_ids.size());
          r.d[0].resize(r.set_ids.size());
          for (size_t e = 0; e < r.set_ids.size(); e++) r.l[0][e] = vdict_value_of_key(ao.aux_col->vdict_keys[(size_t)r.set_ids[e]], ao.aux_col->vdict_kind, &r.d[0][e]);
        }
      } else {
        r.kind = PG_RESULT_HLL;
        r.log2m = ao.log2m;
        // runs of consecutive group ids are appended with one copy each (a full key space is one copy; no zero-fill first)
        if (H.aux_block) {   // big states stay in the page-locked block the device wrote them to
          r.hll.clear();
          r.hll_block = H.aux_block;
          r.hll_regs = H.aux + off;
          r.hll_stride = A.stride;
          r.hll_gids.assign(gids.begin(), gids.end());
          continue;
        }
        r.hll.clear();
        r.hll.reserve((size_t)ng * A.stride);
        const uint8_t* regs = H.aux + off;
        for (int32_t i = 0; i < ng;) {
          int32_t j = i + 1;
          while (j < ng && gids[j] == gids[j - 1] + 1) j++;
          r.hll.insert(r.hll.end(), regs + (size_t)gids[i] * A.stride, regs + ((size_t)gids[j - 1] + 1) * A.stride);
          i = j;
        }
      }
      continue;
    }
    switch (ao.function) {
      case PG_AGG_COUNT:
        r.kind = PG_RESULT_LONG;
        r.l[0].resize((size_t)ng);
        for (int32_t i = 0; i < ng; i++) r.l[0][i] = count_of(ao.op_a, gids[i]);
        break;
      case PG_AGG_AVG:
        r.kind = PG_RESULT_AVG_PAIR;
        r.d[0].resize((size_t)ng);
        r.l[0].resize((size_t)ng);
        for (int32_t i = 0; i < ng; i++) { r.d[0][i] = sum_double(ao, gids[i]); r.l[0][i] = count_of(ao.op_b, gids[i]); }
        break;
      case PG_AGG_MINMAXRANGE:
        r.kind = PG_RESULT_MINMAX_PAIR;
        r.d[0].resize((size_t)ng);
        r.d[1].resize((size_t)ng);
        for (int32_t i = 0; i < ng; i++) { r.d[0][i] = op_double(ao.op_a, gids[i]); r.d[1][i] = op_double(ao.op_b, gids[i]); }
        break;
      default:
        r.kind = PG_RESULT_DOUBLE;
        r.d[0].resize((size_t)ng);
        if (ao.function == PG_AGG_SUM) for (int32_t i = 0; i < ng; i++) r.d[0][i] = sum_double(ao, gids[i]);
        else for (int32_t i = 0; i < ng; i++) r.d[0][i] = op_double(ao.op_a, gids[i]);
        break;
    }
  }
  if (trace) fprintf(stderr, "[pg] assembly: groups %.3f ms, keys %.3f, values %.3f (%d groups)\n", ta1 - ta0, ta2 - ta1, now_ms() - ta2, ng);
}

// Rebuilds the host view (groups, intermediates, statistics) of a result from its device table — after a merge.
void result_reassemble(Result& r) {
  if (!r.dev) fail(PG_ERR_INVALID_ARGUMENT, "result has no device table (PG_QUERY_FLAG_KEEP_DEVICE_TABLE)");
  DeviceTable& T = *r.dev;
  ThreadCtx& ctx = ctx_on(T.device);
  const size_t out_bytes = ((size_t)T.n_out + PG_MAX_STATS + 2) * 8;
  int64_t* host_out = static_cast<int64_t*>(ctx.pin(out_bytes + T.aux_total));
  uint8_t* aux_host = reinterpret_cast<uint8_t*>(host_out) + out_bytes;
  PG_HIP(hipMemcpyAsync(host_out, T.table.ptr, out_bytes, hipMemcpyDeviceToHost, ctx.stream));
  if (T.aux_total) PG_HIP(hipMemcpyAsync(aux_host, T.aux.ptr, T.aux_total, hipMemcpyDeviceToHost, ctx.stream));
  PG_HIP(hipStreamSynchronize(ctx.stream));
  HostTable H;
  H.table.assign(host_out, host_out + T.n_out);
  memcpy(H.stats, host_out + T.n_out, sizeof(H.stats));
  H.aux = aux_host;
  H.full_scan_entries = host_out[T.n_out + PG_MAX_STATS];
  H.total_docs = host_out[T.n_out + PG_MAX_STATS + 1];
  T.full_scan_entries = H.full_scan_entries;
  T.num_total_docs = H.total_docs;
  H.keys = T.keys.get();
  assemble_result(r, *T.plan, T.n_group_by, T.n_aggregations, H);
}

// ---- element-wise merge of two dense tables of the same query (GroupByCombineOperator over segments sharing their key space) ----
extern "C" __global__ void __launch_bounds__(256) pg_merge_tables_kernel(int64_t* __restrict__ dst, const int64_t* __restrict__ src,
                                                                          int64_t n_out, int64_t n_groups, int64_t n_tail, const PgAccOp* __restrict__ ops) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out + n_tail) return;
  if (i >= n_out) { dst[i] += src[i]; return; }   // statistics counters, full-scan entries, total docs
  const PgAccOp op = ops[i / n_groups];
  const int64_t a = dst[i], b = src[i];
  if (op.fn == PG_ACC_MIN) dst[i] = b < a ? b : a;
  else if (op.fn == PG_ACC_MAX) dst[i] = b > a ? b : a;
  else if (op.fn == PG_ACC_SUM && op.is_float == 1) dst[i] = __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b));
  else dst[i] = a + b;
}
extern "C" __global__ void __launch_bounds__(256) pg_merge_aux_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int64_t n_words,
                                                                       int n_src, int bytewise_max) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint32_t acc = dst[w];
  for (int k = 0; k < n_src; k++) {
    const uint32_t v = src[(int64_t)k * n_words + w];
    if (bytewise_max) {
      uint32_t r = 0;
      for (int b = 0; b < 4; b++) {
        const uint32_t x = (acc >> (8 * b)) & 0xFFu, y = (v >> (8 * b)) & 0xFFu;
        r |= (x > y ? x : y) << (8 * b);
      }
      acc = r;
    } else {
      acc |= v;
    }
  }
  dst[w] = acc;
}

// Layout signature two tables must share to merge element-wise: accumulators, key space, auxiliary states.
int64_t table_signature(const DeviceTable& T) {
  const PgQueryPlan& D = T.plan->dev;
  uint64_t h = 1469598103934665603ULL;
  auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ULL; };
  mix((uint64_t)T.n_out); mix((uint64_t)D.n_groups); mix((uint64_t)D.n_ops); mix((uint64_t)T.aux_total); mix((uint64_t)T.n_group_by);
  for (int o = 0; o < D.n_ops; o++) { mix((uint64_t)D.ops[o].fn); mix((uint64_t)D.ops[o].is_float); mix((uint64_t)(uint32_t)D.ops[o].limb); if (D.ops[o].src >= 0) mix((uint64_t)(uint32_t)D.srcs[D.ops[o].src].fx_q); }
  for (int x = 0; x < D.n_aux; x++) { mix((uint64_t)D.aux[x].kind); mix((uint64_t)D.aux[x].stride); mix((uint64_t)D.aux[x].n_rep); mix((uint64_t)D.aux[x].rep_bytes); }
  for (size_t a = 0; a < T.plan->aggs.size(); a++) { mix((uint64_t)T.plan->aggs[a].function); mix((uint64_t)(uint32_t)T.plan->aggs[a].op_a); }
  for (int32_t c : T.plan->group_cards) mix((uint64_t)c);
  for (const Column* vd : T.plan->group_vdict) if (vd) mix(vd->vdict_hash);   // segment-local ids: only equal dictionaries merge
  // dictIds index the table (group-by columns) and the DISTINCTCOUNT sets: different dictionaries of equal cardinality must not merge
  for (uint64_t dh : T.plan->dict_hashes) mix(dh);
  if (T.keys) {   // re-keyed by a merge by value: the union's values ARE the key space (two tables over the same union merge element-wise)
    mix(0x6b657973ULL);
    for (const auto& u : T.keys->dicts) {
      mix((uint64_t)u->cardinality);
      for (uint64_t k : u->vdict_keys) mix(k);
      for (uint8_t b : u->vdict_bytes) mix(b);
      for (int64_t o : u->vdict_bytes_off) mix((uint64_t)o);
    }
  }
  return (int64_t)(h >> 2);   // 62 bits: survives ncclMax / negation
}
// The plan-time guarantees of the accumulators hold per segment (docs x largest |value| < 2^63 for an int64 SUM; < 2^31 docs for
// 32-bit digits summed in int64): a merged table must still satisfy them over the docs of all the segments it folds.
// The bound is a property of the merged TABLE (largest per-doc magnitude over everything folded into it), not of one plan.
void check_merge_bounds(uint64_t sum_max_abs, bool has_digit_sums, int64_t total_docs) {
  if (has_digit_sums && total_docs >= ((int64_t)1 << 31))
    fail(PG_ERR_UNSUPPORTED, "merge: %lld docs in total overflow the digit accumulators of an exact SUM (merge on the host by values)", (long long)total_docs);
  if (sum_max_abs && (unsigned __int128)sum_max_abs * (unsigned __int128)std::max<int64_t>(total_docs, 1) >= ((unsigned __int128)1 << 63))
    fail(PG_ERR_UNSUPPORTED, "merge: a SUM over %lld docs in total may leave int64 (merge on the host by values)", (long long)total_docs);
}
void device_table_tail_store(DeviceTable& T, hipStream_t stream) {   // full-scan entries + total docs behind the statistics counters
  // asynchronous: the source lives in the table object (round 2 synchronised the stream here, once per merge)
  T.tail_host[0] = T.full_scan_entries;
  T.tail_host[1] = T.num_total_docs;
  PG_HIP(hipMemcpyAsync(T.table.as<int64_t>() + T.n_out + PG_MAX_STATS, T.tail_host, sizeof(T.tail_host), hipMemcpyHostToDevice, stream));
}

extern "C" __global__ void __launch_bounds__(256) pg_remap_fill_kernel(int64_t* __restrict__ dst, int64_t n_groups, int n_ops, const PgAccOp* __restrict__ ops) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups * n_ops) return;
  const PgAccOp op = ops[i / n_groups];
  dst[i] = pg_acc_identity(op.fn, op.is_float);
}
// group g of the plan's key space -> group of the union's: digit j of g through column j's dictId map (maps back to back, column 0 first)
extern "C" __global__ void __launch_bounds__(256) pg_remap_table_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst, int64_t G, int64_t G2,
                                                                         int n_ops, int n_cols, const int32_t* __restrict__ maps, const int64_t* __restrict__ geo) {
  // geo: per column {old cardinality, new multiplier, offset of its map}
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  int64_t rest = g, g2 = 0;
  for (int j = 0; j < n_cols; j++) {
    const int64_t card = geo[3 * j], digit = rest % card;
    rest /= card;
    g2 += (int64_t)maps[geo[3 * j + 2] + digit] * geo[3 * j + 1];
  }
  for (int o = 0; o < n_ops; o++) dst[(int64_t)o * G2 + g2] = src[(int64_t)o * G + g];   // (a group that does not exist carries the identities)
}
// the value-keyed merge's re-keying (pg_comm.cpp, union_key_space): identities everywhere, then every group of the plan's key space to its place
void remap_table_on_stream(const int64_t* src, int64_t* dst, int64_t G, int64_t G2, int n_ops, int n_cols, const int32_t* maps, const int64_t* geo,
                           const PgAccOp* ops, hipStream_t stream) {
  const int64_t n_out2 = (int64_t)n_ops * G2;
  hipLaunchKernelGGL(pg_remap_fill_kernel, dim3((unsigned)((n_out2 + 255) / 256)), dim3(256), 0, stream, dst, G2, n_ops, ops);
  hipLaunchKernelGGL(pg_remap_table_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, stream, src, dst, G, G2, n_ops, n_cols, maps, geo);
  PG_HIP(hipGetLastError());
}
void merge_sets_on_stream(uint32_t* dst, const uint32_t* gathered, int64_t n_words, int n_src, hipStream_t stream) {
  hipLaunchKernelGGL(pg_merge_aux_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, dst, gathered, n_words, n_src, 0);
  PG_HIP(hipGetLastError());
}

void result_merge(Result& dst, Result& src) {
  if (!dst.dev || !src.dev) fail(PG_ERR_INVALID_ARGUMENT, "pg_result_merge needs results executed with PG_QUERY_FLAG_KEEP_DEVICE_TABLE");
  DeviceTable& A = *dst.dev;
  DeviceTable& B = *src.dev;
  if (A.device != B.device) fail(PG_ERR_INVALID_ARGUMENT, "pg_result_merge: results live on devices %d and %d (use pg_result_all_reduce across devices)", A.device, B.device);
  // segments of one table carry dictionaries of their own: tables that differ ONLY in their group-by dictionaries are re-keyed into the union
  // of the dictionaries first (pg_comm.cpp, merge by value: GroupByCombineOperator.java:135-144), then merge element-wise like any others
  if (table_signature(A) != table_signature(B) && !merge_rekey_by_value(A, B, ctx_on(A.device).stream))
    fail(PG_ERR_UNSUPPORTED, "pg_result_merge: the two results do not share their table layout (different aggregations, or a key space that does not "
                             "re-key by value: hashed / raw / multi-value keys, distinct-count states): merge on the host by values");
  // the larger of the two tables' per-doc magnitudes bounds the merged sums; the merged table keeps it for later merges
  const uint64_t merged_max_abs = std::max(A.sum_max_abs, B.sum_max_abs);
  const bool merged_digits = A.has_digit_sums || B.has_digit_sums;
  check_merge_bounds(merged_max_abs, merged_digits, A.num_total_docs + B.num_total_docs);
  A.sum_max_abs = merged_max_abs;
  A.has_digit_sums = merged_digits;
  ThreadCtx& ctx = ctx_on(A.device);
  device_table_tail_store(A, ctx.stream);
  device_table_tail_store(B, ctx.stream);
  const int64_t n_tail = PG_MAX_STATS + 2, n = A.n_out + n_tail;
  hipLaunchKernelGGL(pg_merge_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx.stream, A.table.as<int64_t>(), B.table.as<int64_t>(),
                     A.n_out, A.keys ? A.keys->n_groups : (int64_t)std::max(A.plan->dev.n_groups, 1), n_tail, A.plan->ops_dev.as<PgAccOp>());
  PG_HIP(hipGetLastError());
  size_t off = 0;
  for (int x = 0; x < A.plan->dev.n_aux; x++) {
    const int64_t n_words = (int64_t)A.plan->aux_bytes[(size_t)x] / 4;
    hipLaunchKernelGGL(pg_merge_aux_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx.stream,
                       reinterpret_cast<uint32_t*>(A.aux.as<uint8_t>() + off), reinterpret_cast<const uint32_t*>(B.aux.as<uint8_t>() + off), n_words, 1,
                       A.plan->dev.aux[x].kind == PG_AUX_DICT_SET ? 0 : 1);
    PG_HIP(hipGetLastError());
    off += A.plan->aux_bytes[(size_t)x];
  }
  PG_HIP(hipStreamSynchronize(ctx.stream));
  result_reassemble(dst);
}

std::unique_ptr<DocIdSet> execute_filter(Segment& seg, const pg_filter_node* filter, int32_t flags) {
  const double t0 = now_ms();
  ThreadCtx& ctx = ctx_on(seg.device);
  auto plan = get_plan(seg, filter, nullptr, flags);
  CompiledPlan& P = *plan;
  auto out = std::make_unique<DocIdSet>();
  out->device = seg.device;
  out->num_docs = seg.total_docs;
  const size_t n_words = (size_t)std::max(seg.n_tiles, 1) * PG_TILE_WORDS;
  out->words.alloc(n_words * 8, true);
  out->tile_counts.assign((size_t)std::max(seg.n_tiles, 1), 0);
  ThreadCtx::grow(ctx.tile_counts, out->tile_counts.size() * 4);
  PG_HIP(hipMemsetAsync(ctx.tile_counts.ptr, 0, out->tile_counts.size() * 4, ctx.stream));
  PG_HIP(hipMemsetAsync(ctx.stats.ptr, 0, PG_MAX_STATS * 8, ctx.stream));
  ctx.stats_dirty = true;
  PgQueryPlan D = P.dev;
  D.stats = ctx.stats.as<unsigned long long>();
  D.out_words = out->words.as<uint64_t>();
  D.out_tile_counts = ctx.tile_counts.as<uint32_t>();
  D.agg_mode = PG_AGG_NONE;
  const LaunchShape shape = launch_shape(P, P.dev.n_wtiles, PG_AGG_NONE);
  PG_HIP(hipEventRecord(ctx.ev[0], ctx.stream));
  const char* kname = "";
  if (seg.total_docs > 0) {
    QueryKernel kern = select_kernel(P, PG_AGG_NONE, &kname);
    hipLaunchKernelGGL(kern, dim3(shape.grid), dim3(shape.block), shape.lds, ctx.stream, D);
    PG_HIP(hipGetLastError());
  }
  PG_HIP(hipEventRecord(ctx.ev[1], ctx.stream));
  uint64_t stats_host[PG_MAX_STATS];
  PG_HIP(hipMemcpyAsync(stats_host, ctx.stats.ptr, sizeof(stats_host), hipMemcpyDeviceToHost, ctx.stream));
  PG_HIP(hipMemcpyAsync(out->tile_counts.data(), ctx.tile_counts.ptr, out->tile_counts.size() * 4, hipMemcpyDeviceToHost, ctx.stream));
  PG_HIP(hipStreamSynchronize(ctx.stream));
  fill_stats(out->stats, P, P.full_scan_entries, seg.total_docs, stats_host);
  {   // pg_filter_exec has no flags: the exact count of leapfrogged shapes up to the default size (see execute_query)
    if (!P.stats_exact && exact_stats_by_default(P)) {
      out->stats.num_entries_scanned_in_filter = exact_entries_scanned(P, ctx, nullptr, &out->stats.filter_stats_path);
      out->stats.stats_exact = 1;
    }
  }
  out->stats.star_tree_index = -1;
  snprintf(out->stats.kernel, sizeof(out->stats.kernel), "%s", kname);
  out->stats.num_entries_scanned_post_filter = 0;
  out->cardinality = (int64_t)stats_host[0];
  float ms = 0;
  PG_HIP(hipEventElapsedTime(&ms, ctx.ev[0], ctx.ev[1]));
  out->stats.device_ms_filter = ms;
  out->stats.device_ms_total = ms;
  out->stats.host_ms_total = (float)(now_ms() - t0);
  return out;
}

void docidset_copy_docids(DocIdSet& s, int32_t* out, int64_t cap) {
  if (cap < s.cardinality) fail(PG_ERR_INVALID_ARGUMENT, "capacity %lld < cardinality %lld", (long long)cap, (long long)s.cardinality);
  if (s.cardinality == 0) return;
  ThreadCtx& ctx = ctx_on(s.device);
  const int n_tiles = (int)s.tile_counts.size();
  std::vector<int64_t> offs((size_t)n_tiles + 1, 0);
  for (int i = 0; i < n_tiles; i++) offs[(size_t)i + 1] = offs[i] + s.tile_counts[i];
  DeviceBuffer d_offs = upload_vector(offs);
  DeviceBuffer d_out((size_t)s.cardinality * 4);
  int grid = std::min(n_tiles, num_cus() * 8);
  hipLaunchKernelGGL(pg_expand_docids_kernel, dim3(grid), dim3(PG_TILE_WORDS), 0, ctx.stream, s.words.as<uint64_t>(),
                     d_offs.as<int64_t>(), d_out.as<int32_t>(), n_tiles);
  PG_HIP(hipGetLastError());
  PG_HIP(hipMemcpyAsync(out, d_out.ptr, (size_t)s.cardinality * 4, hipMemcpyDeviceToHost, ctx.stream));
  PG_HIP(hipStreamSynchronize(ctx.stream));
}

}  // namespace pg
