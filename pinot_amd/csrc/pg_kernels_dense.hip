// pg_fast_i32range_d in a translation unit of its own (see PG_KERNEL in pg_kernels.hip): the device code is shared by inclusion, the
// other kernels become unreferenced static functions here and are dropped.
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

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

