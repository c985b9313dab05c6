// pg_fast_i32range_d in a translation unit of its own (see PG_KERNEL in pg_kernels.hip): the device code is shared by inclusion, the
// other kernels become unreferenced static functions here and are dropped.
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"
#include <type_traits>

// Headline shape with the index program fused (PgQueryPlan::dense_fused): [AND of <= 4 OR-groups of dense postings] AND raw-INT range
// → LDS-table aggregation.  Two differences from pg_fast_i32range_a: (1) the index program is 8 loads and a few ANDs / ORs instead
// of the interpreted leaves (2 x 8 loads, half of them repeats); (2) those 8 loads are issued ONE TILE AHEAD — while the current
// tile's scan and aggregation run — which takes the postings round trip out of the tile's dependent chain.
template <int UNUSED>
__device__ __forceinline__ void fast_dense_i32range_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  {
    const uint32_t table_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
  }
  __syncthreads();
  const CAS PgScanLeaf& L = cptr(p.scans)[p.fast_scan];
  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  uint32_t my_matched = 0, my_cand = 0;
  const int wstride = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  int wt = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  uint32_t pv[8];
  auto issue = [&](int tile) {   // tile < n_wtiles; dense postings are addressed arithmetically: dword index = tile * 64 + lane
    const uint32_t di = (uint32_t)tile * 64u + (uint32_t)lane;
#pragma unroll
    for (int j = 0; j < 8; j++) pv[j] = ldnt(gptr<uint32_t>(p.dense_ptr[j]) + di);
  };
  if (wt < p.n_wtiles) issue(wt);
  for (; wt < p.n_wtiles; wt += wstride) {
    const int64_t wbase = (int64_t)wt * PG_WAVE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - wbase;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    // combine: OR inside a group, complement, AND across the groups
    uint32_t grp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int g = p.dense_group[j];
#pragma unroll
      for (int k = 0; k < 4; k++) grp[k] |= g == k ? pv[j] : 0u;
    }
    uint32_t lin = valid_lin_mask(n_valid, lane);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < p.dense_groups) lin &= ((p.dense_excl >> k) & 1) ? ~grp[k] : grp[k];
    if (wt + wstride < p.n_wtiles) issue(wt + wstride);   // the next tile's postings travel while this tile is scanned and aggregated
    uint32_t m = lin_to_quad(lin, lane);
    my_cand += (uint32_t)__popc(m);
    const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)wt * (PG_WAVE_DOCS * 4));
    m = scan_wtile<SK_I32_RANGE>(L, m, tb, lane);
    my_matched += (uint32_t)__popc(m);
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
    if (__ballot(m != 0)) fast_aggregate_wtile<PG_FAST_AGG_B>(p, m, wt, lds_table, lane, rep);
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  const uint32_t csum = wave_sum_u32(my_cand);
  if (lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);
  __syncthreads();
  flush_workgroup(p, lds_table, s_stat, true, t);
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_fast_i32range_d(const PgQueryPlan p) { fast_dense_i32range_body<0>(p); }


// COUNT(*) behind an index-only filter over dense postings — FastFilteredCountOperator's shape
// (core/operator/query/FastFilteredCountOperator.java, BitmapCollection.java:57-126: cardinalities of AND / OR / inverted bitmaps without
// scanning anything): the bitmaps are streamed 16 bytes (128 docs) per lane and load, OR-ed inside a leaf, AND-ed across leaves, popcounted.
// pg_fast_none_f answers the same query tile by tile — one dword per lane and posting, a dependent round trip per 2 048 docs: 32 % of
// 8 TB/s on 0.75 bytes per doc (profiles/r03_l_variants_cfg3_200m.txt); this is a plain stream with 2 x NP loads in flight per lane.
template <int NP>
__device__ __forceinline__ void dense_count_body(const PgQueryPlan& p) {
  const int64_t n_q = ((int64_t)p.num_docs + 127) >> 7;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  uint32_t cnt = 0;
  auto combine = [&](const u32x4 (&v)[NP], int64_t q) __attribute__((always_inline)) {
    uint32_t m[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k >= p.dense_groups) continue;   // wave-uniform
      uint32_t g[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < NP; j++) {
        const bool in = p.dense_group[j] == k;
        g[0] |= in ? v[j].x : 0u; g[1] |= in ? v[j].y : 0u; g[2] |= in ? v[j].z : 0u; g[3] |= in ? v[j].w : 0u;
      }
      const bool excl = ((p.dense_excl >> k) & 1) != 0;
#pragma unroll
      for (int w = 0; w < 4; w++) m[w] &= excl ? ~g[w] : g[w];
    }
    const int64_t rem = (int64_t)p.num_docs - (q << 7);   // docs from this lane's first bit on
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int64_t r = rem - 32 * w;
      const uint32_t valid = r >= 32 ? 0xFFFFFFFFu : (r <= 0 ? 0u : ((1u << (uint32_t)r) - 1u));
      cnt += (uint32_t)__popc(m[w] & valid);
    }
  };
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; q + stride < n_q; q += 2 * stride) {   // two positions per iteration: 2 x NP loads in flight
    u32x4 a[NP], b[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) a[j] = ldnt(gptr<u32x4>(p.dense_ptr[j]) + q);
#pragma unroll
    for (int j = 0; j < NP; j++) b[j] = ldnt(gptr<u32x4>(p.dense_ptr[j]) + q + stride);
    combine(a, q);
    combine(b, q + stride);
  }
  if (q < n_q) {
    u32x4 a[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) a[j] = ldnt(gptr<u32x4>(p.dense_ptr[j]) + q);
    combine(a, q);
  }
  // ONE update of the counter per workgroup: same-address global atomics retire one by one (~10 ns each on this part — with an atomic per
  // wavefront of 2 048 small workgroups they were 0.15 of the kernel's 0.16 ms, profiles/r04_z_count_streams.txt)
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t wsum = wave_sum_u32(cnt);
  if (lane == 0 && wsum) atomicAdd(&s_cnt, wsum);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(p.stats, (unsigned long long)s_cnt);
}
#define PG_DENSE_COUNT(NP) \
  extern "C" __global__ void __launch_bounds__(1024) pg_dense_count_##NP(const PgQueryPlan p) { dense_count_body<NP>(p); }
PG_DENSE_COUNT(1) PG_DENSE_COUNT(2) PG_DENSE_COUNT(3) PG_DENSE_COUNT(4) PG_DENSE_COUNT(5) PG_DENSE_COUNT(6) PG_DENSE_COUNT(7) PG_DENSE_COUNT(8)

// COUNT(*) WHERE <a dictionary column of <= 8 bits> in a dictId range / set, nothing else — ScanBasedFilterOperator over the whole segment
// (core/operator/filter/ScanBasedFilterOperator.java:59-66, SVScanDocIdIterator.java:76-98) — as a stream: thread i owns docs 32 i .. 32 i + 31,
// i.e. B dwords of the MSB-first bit stream at dword offset B i (dword-aligned for every width), cut into fields with compile-time shifts.
// pg_fast_dict{range,lut}_f answer it tile by tile with run-time-width windows at 32-33 % of 8 TB/s (7/8 of a byte per doc: latency per
// tile, profiles/r03_l_variants_cfg3_200m.txt).
typedef uint32_t dc_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t dc_u32x3 __attribute__((ext_vector_type(3)));
typedef dc_u32x2 dc_u32x2_a4 __attribute__((aligned(4)));
typedef dc_u32x3 dc_u32x3_a4 __attribute__((aligned(4)));
typedef u32x4 dc_u32x4_a4 __attribute__((aligned(4)));
// N consecutive dwords (dword-aligned) in as few loads as the width allows
template <int N>
DEVFN void load_dwords(const GAS uint32_t* q, uint32_t* out) {
  if constexpr (N >= 4) {
    const u32x4 v = ldnt((const GAS dc_u32x4_a4*)q);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    load_dwords<N - 4>(q + 4, out + 4);
  } else if constexpr (N == 3) {
    const dc_u32x3 v = ldnt((const GAS dc_u32x3_a4*)q);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
  } else if constexpr (N == 2) {
    const dc_u32x2 v = ldnt((const GAS dc_u32x2_a4*)q);
    out[0] = v.x; out[1] = v.y;
  } else if constexpr (N == 1) {
    out[0] = ldnt(q);
  }
}
template <int B>
__device__ __forceinline__ void dict_count_body(const PgQueryPlan& p) {
  __shared__ uint32_t s_cnt;
  __shared__ uint32_t s_lut[8];
  const CAS PgScanLeaf& L = cptr(p.scans)[p.fast_scan];
  const bool lut = L.pred_kind == PG_P_DICT_LUT;
  if (threadIdx.x == 0) s_cnt = 0;
  if (threadIdx.x < 8) s_lut[threadIdx.x] = lut && (int)threadIdx.x * 32 < (1 << B) ? gptr<uint32_t>(L.lut)[threadIdx.x] : 0u;   // (<= 256 dictIds)
  __syncthreads();
  const uint32_t lo = (uint32_t)L.lo, span = (uint32_t)(L.hi - L.lo);
  const int64_t n_g = ((int64_t)p.num_docs + 31) >> 5;   // 32-doc groups
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const GAS uint32_t* data = gptr<uint32_t>(L.data);
  uint32_t cnt = 0;
  auto test = [&](const uint32_t (&raw)[B], int64_t g, auto is_lut) __attribute__((always_inline)) {   // is_lut: compile-time (the set test reads LDS per value)
    uint32_t w[B + 1];
#pragma unroll
    for (int j = 0; j < B; j++) w[j] = bswap32(raw[j]);
    w[B] = 0;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      constexpr int dummy = 0; (void)dummy;
      const int bit = k * B, j = bit >> 5, s = bit & 31;
      // bits [s, s + B) of (w[j] : w[j + 1]), MSB first
      const uint32_t d = s + B <= 32 ? (w[j] >> (32 - s - B)) & ((1u << B) - 1u)
                                     : (uint32_t)(((((uint64_t)w[j] << 32) | (uint64_t)w[j + 1]) >> (64 - s - B)) & ((1u << B) - 1u));
      bool pass;
      if constexpr (decltype(is_lut)::value) pass = ((s_lut[d >> 5] >> (d & 31u)) & 1u) != 0;
      else pass = (d - lo) <= span;
      m |= (uint32_t)pass << k;
    }
    const int64_t rem = (int64_t)p.num_docs - (g << 5);
    cnt += (uint32_t)__popc(rem >= 32 ? m : (rem <= 0 ? 0u : (m & ((1u << (uint32_t)rem) - 1u))));
  };
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; g + stride < n_g; g += 2 * stride) {   // two groups per iteration in flight
    uint32_t a[B], b[B];
    load_dwords<B>(data + g * B, a);
    load_dwords<B>(data + (g + stride) * B, b);
    if (lut) { test(a, g, std::true_type{}); test(b, g + stride, std::true_type{}); }   // wave-uniform
    else { test(a, g, std::false_type{}); test(b, g + stride, std::false_type{}); }
  }
  if (g < n_g) {
    uint32_t a[B];
    load_dwords<B>(data + g * B, a);
    if (lut) test(a, g, std::true_type{});
    else test(a, g, std::false_type{});
  }
  const uint32_t wsum = wave_sum_u32(cnt);
  if ((threadIdx.x & 63) == 0 && wsum) atomicAdd(&s_cnt, wsum);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(p.stats, (unsigned long long)s_cnt);   // one update per workgroup (see dense_count_body)
}
#define PG_DICT_COUNT(B) \
  extern "C" __global__ void __launch_bounds__(1024) pg_dict_count_##B(const PgQueryPlan p) { dict_count_body<B>(p); }
PG_DICT_COUNT(1) PG_DICT_COUNT(2) PG_DICT_COUNT(3) PG_DICT_COUNT(4) PG_DICT_COUNT(5) PG_DICT_COUNT(6) PG_DICT_COUNT(7) PG_DICT_COUNT(8)
