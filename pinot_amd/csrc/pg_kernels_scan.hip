// pg_fast_i32range_fp: BASELINE config 2 — a raw-INT range predicate over the whole segment, COUNT(*) (or the match words of a filter-only
// pass) — as a double-buffered stream on FOUR wavefronts per workgroup.
//
// The pure-scan probe (profiles/r02_scan_bw_probe.txt) streams 7.33 TB/s with 4 wavefronts x 8 KB in flight per CU and 6.5-6.95 TB/s
// with 16: the memory system prefers fewer, longer streams.  pg_fast_i32range_f (16 wavefronts, 4 KB batches, load → test → load) sits
// at the 16-wavefront figure.  This kernel keeps a whole 8 KB tile per wavefront in flight at all times instead: tile i+1's eight
// quads are requested before tile i's are tested, so the ~200 VALU instructions of a tile's test hide behind the next tile's loads
// even with one wavefront per SIMD.  In a translation unit of its own for its workgroup size (see pg_kernels_pipe.hip).
#ifndef PG_WAVES_PER_BLOCK
#define PG_WAVES_PER_BLOCK 4
#endif
#ifndef PG_SCAN_BUFFERS
#define PG_SCAN_BUFFERS 2   // tiles in rotation per wavefront; 3 and 4 measured 2 % SLOWER over 10^9 docs (0.564 / 0.579 / 0.578 ms), the same over 10^8
#endif
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

extern "C" const int pg_scan_waves_per_block = PG_WAVES_PER_BLOCK;   // the host launches pg_fast_i32range_fp with this many wavefronts

extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_fast_i32range_fp(const PgQueryPlan p) {
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  __syncthreads();
  const CAS PgScanLeaf& L = cptr(p.scans)[p.fast_scan];
  const RangeI32 r32 = make_range_i32(L.lo, L.hi);
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int last_wt = p.n_wtiles - 1;
  uint32_t my_matched = 0;

  auto issue = [&](int wt, u32x4 (&a)[8]) __attribute__((always_inline)) {   // the whole tile, unconditionally (a pushed scan visits every doc); clamped beyond the segment
    const int wc = wt < last_wt ? wt : last_wt;
    const uint64_t base = (uint64_t)L.data + (uint64_t)wc * (PG_WAVE_DOCS * 4);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    const GAS uint8_t* tb = (const GAS uint8_t*)(((uint64_t)hi << 32) | (uint64_t)lo);
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = ldnt((const GAS u32x4*)(tb + (uint32_t)(k * 64 + lane) * 16u));
  };
  auto finish = [&](int wt, const u32x4 (&a)[8]) __attribute__((always_inline)) {
    const int64_t rem = (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].x)) << (4 * k);
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].y)) << (4 * k + 1);
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].z)) << (4 * k + 2);
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].w)) << (4 * k + 3);
    }
    m = r32.empty ? 0u : (m & valid_quad_mask(n_valid, lane));
    const uint32_t cnt = (uint32_t)__popc(m);
    my_matched += cnt;
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
    if (p.out_tile_counts) {
      const uint32_t wsum = wave_sum_u32(cnt);
      if (lane == 0 && wsum) atomicAdd(&p.out_tile_counts[wt / PG_WTILES_PER_TILE], wsum);
    }
  };

  // PG_SCAN_BUFFERS whole tiles in rotation per wavefront (one wavefront per SIMD: 512 registers to spend): tile i is tested while tiles i + 1 ..
  // i + NB - 1 travel, and its buffer is re-requested for tile i + NB right behind the test.  Whole rounds in the loop, the rest behind it (a
  // conditional part inside the loop makes the compiler's wait counts conservative, see pg_kernels_spec.hip).
  constexpr int NB = PG_SCAN_BUFFERS;
  u32x4 a[NB][8];
  const int wt0 = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_mine = wt0 < p.n_wtiles ? (p.n_wtiles - wt0 + step - 1) / step : 0;
  if (n_mine > 0) {
#pragma unroll
    for (int k = 0; k < NB; k++) { __builtin_amdgcn_sched_barrier(0); issue(wt0 + k * step, a[k]); }   // harmless (clamped) past the end
    __builtin_amdgcn_sched_barrier(0);
    int i = 0;
    for (; i + NB - 1 < n_mine; i += NB) {
#pragma unroll
      for (int k = 0; k < NB; k++) {
        finish(wt0 + (i + k) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
        issue(wt0 + (i + k + NB) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < NB - 1; k++)
      if (i + k < n_mine) finish(wt0 + (i + k) * step, a[k]);   // wave-uniform
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
}

// pg_nogroup_s1 / _s2 (round 6): aggregation WITHOUT GROUP BY and without a filter over one or two raw INT columns (AggregationOperator over a
// MatchAllFilterOperator: SUM / MIN / MAX / COUNT / AVG / MINMAXRANGE of the columns) in the same frame — four wavefronts per workgroup, whole
// 8 KB tiles per column double-buffered — with the accumulators in REGISTERS: per lane an int64 sum and an int32 minimum / maximum per column,
// folded into the workgroup's partial row once, at the end.  pg_fast_none_a gave every doc one LDS atomic per accumulator (a private slot per
// thread: no conflicts, still 4 x 64 LDS operations per wave tile and accumulator): 43.6 % of 8 TB/s on `sum min max count(m)`.
template <int NS>
__device__ __forceinline__ void nogroup_stream_body(const PgQueryPlan& p) {
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ long long s_acc[PG_MAX_OPS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  if (t < p.n_ops) s_acc[t] = (long long)pg_acc_identity(p.ops[t].fn, p.ops[t].is_float);
  __syncthreads();
  int src_of[2] = {-1, -1};   // the (at most two) distinct sources, in the accumulators' order
  for (int o = 0; o < p.n_ops; o++) {
    const int s = p.ops[o].src;
    if (s < 0 || s == src_of[0] || s == src_of[1]) continue;
    if (src_of[0] < 0) src_of[0] = s; else src_of[1] = s;
  }
  const uint8_t* col[2] = {p.srcs[src_of[0] < 0 ? 0 : src_of[0]].data, p.srcs[src_of[1] < 0 ? (src_of[0] < 0 ? 0 : src_of[0]) : src_of[1]].data};
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int last_wt = p.n_wtiles - 1;
  uint32_t my_docs = 0;
  long long sum[NS];
  int32_t mn[NS], mx[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) { sum[s] = 0; mn[s] = INT32_MAX; mx[s] = INT32_MIN; }

  auto issue = [&](int wt, u32x4 (&a)[NS][8]) __attribute__((always_inline)) {   // the whole tile of every column; clamped beyond the segment
    const int wc = wt < last_wt ? wt : last_wt;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const uint64_t base = (uint64_t)col[s] + (uint64_t)wc * (PG_WAVE_DOCS * 4);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
      const GAS uint8_t* tb = (const GAS uint8_t*)(((uint64_t)hi << 32) | (uint64_t)lo);
#pragma unroll
      for (int k = 0; k < 8; k++) a[s][k] = ldnt((const GAS u32x4*)(tb + (uint32_t)(k * 64 + lane) * 16u));
    }
  };
  auto take = [&](int s, uint32_t x) __attribute__((always_inline)) {
    const int32_t v = (int32_t)bswap32(x);
    sum[s] += (long long)v;
    mn[s] = v < mn[s] ? v : mn[s];
    mx[s] = v > mx[s] ? v : mx[s];
  };
  auto finish = [&](int wt, const u32x4 (&a)[NS][8]) __attribute__((always_inline)) {
    const int64_t rem = (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS;
    if (rem >= PG_WAVE_DOCS) {   // wave-uniform: a whole tile
      my_docs += 32u;
#pragma unroll
      for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 8; k++) { take(s, a[s][k].x); take(s, a[s][k].y); take(s, a[s][k].z); take(s, a[s][k].w); }
    } else {
      const uint32_t m = valid_quad_mask(rem > 0 ? (int32_t)rem : 0, lane);
      my_docs += (uint32_t)__popc(m);
#pragma unroll
      for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if ((m >> (4 * k)) & 1u) take(s, a[s][k].x);
          if ((m >> (4 * k + 1)) & 1u) take(s, a[s][k].y);
          if ((m >> (4 * k + 2)) & 1u) take(s, a[s][k].z);
          if ((m >> (4 * k + 3)) & 1u) take(s, a[s][k].w);
        }
    }
  };

  constexpr int NB = 2;
  u32x4 a[NB][NS][8];
  const int wt0 = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_mine = wt0 < p.n_wtiles ? (p.n_wtiles - wt0 + step - 1) / step : 0;
  if (n_mine > 0) {
#pragma unroll
    for (int k = 0; k < NB; k++) { __builtin_amdgcn_sched_barrier(0); issue(wt0 + k * step, a[k]); }
    __builtin_amdgcn_sched_barrier(0);
    int i = 0;
    for (; i + NB - 1 < n_mine; i += NB) {
#pragma unroll
      for (int k = 0; k < NB; k++) {
        finish(wt0 + (i + k) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
        issue(wt0 + (i + k + NB) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < NB - 1; k++)
      if (i + k < n_mine) finish(wt0 + (i + k) * step, a[k]);   // wave-uniform
  }
  // the lane's accumulators into the workgroup's row (once per kernel: 256 LDS atomics per accumulator)
  if (my_docs) {
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      const int s = (NS > 1 && op.src >= 0 && op.src == src_of[1]) ? 1 : 0;
      if (op.fn == PG_ACC_COUNT) atomicAdd(reinterpret_cast<unsigned long long*>(&s_acc[o]), (unsigned long long)my_docs);
      else if (op.fn == PG_ACC_SUM) atomicAdd(reinterpret_cast<unsigned long long*>(&s_acc[o]), (unsigned long long)sum[s]);
      else if (op.fn == PG_ACC_MIN) atomicMin(&s_acc[o], (long long)mn[s]);
      else atomicMax(&s_acc[o], (long long)mx[s]);
    }
  }
  const uint32_t wsum = wave_sum_u32(my_docs);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  if (t < p.n_ops) p.partials[(int64_t)blockIdx.x * p.n_ops + t] = (int64_t)s_acc[t];
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_nogroup_s1(const PgQueryPlan p) { nogroup_stream_body<1>(p); }
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_nogroup_s2(const PgQueryPlan p) { nogroup_stream_body<2>(p); }

// pg_nogroup_d (round 6): the same aggregation over ONE dictionary-encoded INT column — Pinot's default encoding of a metric.  The fixed-bit dictId
// stream is read in the oct layout (pg_oct_layout.h: lane L owns docs 8L .. 8L+7 of a 512-doc sub-tile = `bits` bytes at byte offset bits x L; a
// whole wave tile — four sub-tiles, eight 16-byte loads per lane — double-buffered per wavefront) and the DICTIDS are accumulated: their sum, their
// minimum and maximum.  The dictionary is sorted, so MIN / MAX are its values at the extreme dictIds, and SUM is nogroup_base x docs +
// nogroup_step x sum(dictIds) for an arithmetic dictionary (nogroup_d = 1); any other dictionary (nogroup_d = 2) is gathered per doc from its
// native-endian copy for SUM (DataFetcher.java:335-386) — from a copy in the workgroup's LDS where it fits (pg_nogroup_dl: <= 36 K values).  The
// width is a template parameter of the loop (one scalar branch per kernel).
#include "pg_oct_layout.h"

template <int B, int MODE>   // MODE 0: arithmetic dictionary; 1: gathered from HBM; 2: gathered from the copy in LDS
__device__ __forceinline__ void nogroup_dict_loop(const PgQueryPlan& p, int lane, int wave, uint32_t& my_docs, unsigned long long& sum_ids, long long& sum_vals, uint32_t& mn,
                                                  uint32_t& mx, const int32_t* lds_dict) {
  const PgValueSrc& V = p.srcs[p.nogroup_src];
  const GAS int32_t* dict = gptr<int32_t>(reinterpret_cast<const uint8_t*>(V.dict));
  const uint32_t woff = ((uint32_t)lane * (uint32_t)B) & ~3u;                 // the lane's window: dword-aligned start, byte selector
  const uint32_t wsel = oct_selector(((uint32_t)lane * (uint32_t)B) & 3u);
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int last_wt = p.n_wtiles - 1;
  constexpr uint32_t TILE_BYTES = (PG_WAVE_DOCS / 8) * B, SUB_BYTES = (OCT_SUB_DOCS / 8) * B;

  auto issue = [&](int wt, u32x4 (&a)[8]) __attribute__((always_inline)) {
    const int wc = wt < last_wt ? wt : last_wt;
    const uint64_t base = (uint64_t)V.data + (uint64_t)wc * TILE_BYTES;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    const GAS uint8_t* tb = (const GAS uint8_t*)(((uint64_t)hi << 32) | (uint64_t)lo);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      a[2 * s] = ldnt((const GAS u32x4_a4*)(tb + (uint32_t)s * SUB_BYTES + woff));
      a[2 * s + 1] = ldnt((const GAS u32x4_a4*)(tb + (uint32_t)s * SUB_BYTES + woff + 16u));
    }
  };
  auto finish = [&](int wt, const u32x4 (&a)[8]) __attribute__((always_inline)) {
    const int64_t rem = (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS;
    const bool whole = rem >= PG_WAVE_DOCS;   // wave-uniform
    uint32_t tile_sum = 0;                    // 32 ids below 2^24 per lane and tile
#pragma unroll
    for (int s = 0; s < 4; s++) {
      uint32_t id[8];
      oct_decode_wide<B>(a[2 * s], a[2 * s + 1], wsel, id);
      if (whole) {   // wave-uniform: no lane masks, the docs counted per tile
#pragma unroll
        for (int j = 0; j < 8; j++) {
          tile_sum += id[j];
          mn = id[j] < mn ? id[j] : mn;
          mx = id[j] > mx ? id[j] : mx;
          if (MODE == 1) sum_vals += (long long)dict[id[j]];
          if (MODE == 2) sum_vals += (long long)lds_dict[id[j]];
        }
      } else {
        const int64_t first = (int64_t)s * OCT_SUB_DOCS + (int64_t)lane * 8;   // the lane's first doc inside the tile
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (first + j < rem) {
            tile_sum += id[j];
            mn = id[j] < mn ? id[j] : mn;
            mx = id[j] > mx ? id[j] : mx;
            my_docs++;
            if (MODE == 1) sum_vals += (long long)dict[id[j]];
            if (MODE == 2) sum_vals += (long long)lds_dict[id[j]];
          }
        }
      }
    }
    if (whole) my_docs += 32u;
    sum_ids += tile_sum;
  };

  u32x4 a[2][8];
  const int wt0 = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_mine = wt0 < p.n_wtiles ? (p.n_wtiles - wt0 + step - 1) / step : 0;
  if (n_mine > 0) {
    __builtin_amdgcn_sched_barrier(0); issue(wt0, a[0]);
    __builtin_amdgcn_sched_barrier(0); issue(wt0 + step, a[1]);
    __builtin_amdgcn_sched_barrier(0);
    int i = 0;
    for (; i + 1 < n_mine; i += 2) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        finish(wt0 + (i + k) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
        issue(wt0 + (i + k + 2) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (i < n_mine) finish(wt0 + i * step, a[0]);   // wave-uniform
  }
}

template <int MODE>
__device__ __forceinline__ void nogroup_dict_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ long long s_acc[PG_MAX_OPS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  constexpr bool GATHER = MODE != 0;
  int32_t* lds_dict = reinterpret_cast<int32_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  if (t < p.n_ops) s_acc[t] = (long long)pg_acc_identity(p.ops[t].fn, p.ops[t].is_float);
  if (MODE == 2) {   // the dictionary into LDS, once per workgroup
    const GAS int32_t* src = gptr<int32_t>(reinterpret_cast<const uint8_t*>(p.srcs[p.nogroup_src].dict));
    for (int i = t; i < p.nogroup_lds_card; i += PG_BLOCK) lds_dict[i] = src[i];
  }
  __syncthreads();
  uint32_t my_docs = 0, mn = 0xFFFFFFFFu, mx = 0;
  unsigned long long sum_ids = 0;
  long long sum_vals = 0;
  switch (uniform(p.nogroup_bits)) {   // wave-uniform, once
#define NGD_CASE(B) case B: nogroup_dict_loop<B, MODE>(p, lane, wave, my_docs, sum_ids, sum_vals, mn, mx, lds_dict); break;
    NGD_CASE(1) NGD_CASE(2) NGD_CASE(3) NGD_CASE(4) NGD_CASE(5) NGD_CASE(6) NGD_CASE(7) NGD_CASE(8) NGD_CASE(9) NGD_CASE(10) NGD_CASE(11) NGD_CASE(12)
    NGD_CASE(13) NGD_CASE(14) NGD_CASE(15) NGD_CASE(16) NGD_CASE(17) NGD_CASE(18) NGD_CASE(19) NGD_CASE(20) NGD_CASE(21) NGD_CASE(22) NGD_CASE(23)
#undef NGD_CASE
    default: nogroup_dict_loop<24, MODE>(p, lane, wave, my_docs, sum_ids, sum_vals, mn, mx, lds_dict); break;
  }
  if (my_docs) {
    const GAS int32_t* dict = gptr<int32_t>(reinterpret_cast<const uint8_t*>(p.srcs[p.nogroup_src].dict));
    const long long base = (long long)p.nogroup_base, stp = (long long)p.nogroup_step;
    const long long vmin = GATHER ? (long long)dict[mn] : base + stp * (long long)mn;
    const long long vmax = GATHER ? (long long)dict[mx] : base + stp * (long long)mx;
    const long long vsum = GATHER ? sum_vals : base * (long long)my_docs + stp * (long long)sum_ids;
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      if (op.fn == PG_ACC_COUNT) atomicAdd(reinterpret_cast<unsigned long long*>(&s_acc[o]), (unsigned long long)my_docs);
      else if (op.fn == PG_ACC_SUM) atomicAdd(reinterpret_cast<unsigned long long*>(&s_acc[o]), (unsigned long long)vsum);
      else if (op.fn == PG_ACC_MIN) atomicMin(&s_acc[o], vmin);
      else atomicMax(&s_acc[o], vmax);
    }
  }
  const uint32_t wsum = wave_sum_u32(my_docs);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  if (t < p.n_ops) p.partials[(int64_t)blockIdx.x * p.n_ops + t] = (int64_t)s_acc[t];
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_nogroup_da(const PgQueryPlan p) { nogroup_dict_body<0>(p); }
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_nogroup_dg(const PgQueryPlan p) { nogroup_dict_body<1>(p); }
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_nogroup_dl(const PgQueryPlan p) { nogroup_dict_body<2>(p); }

// pg_dictrange_fo (round 6): FILTER ONLY — COUNT(*), a docId set, a leaf's match bitmap — over [a dense index program AND] ONE range predicate on a
// dictionary-encoded column (a dictId interval over the fixed-bit stream, RangePredicateEvaluatorFactory.java:126-167; Pinot's default encoding of
// config 3's filter).  The frame of pg_nogroup_d*: the scan column's wave tile in the oct layout, double-buffered per wavefront (two workgroups per
// CU); the index program runs once per tile in the linear layout (32 docs per lane), its candidates reach the oct lanes by one shuffle per
// sub-tile (oct lane L's 8 docs are byte L & 3 of linear lane 16 s + L / 4), and the matches return to the linear layout by four.
// pg_fast_multi_f decoded the stream quad by quad behind every tile's index program: 34 % of 8 TB/s on `dict filter only count`.
template <int B>
__device__ __forceinline__ void dict_filter_loop(const PgQueryPlan& p, const CAS PgScanLeaf& L, int lane, int wave, uint32_t* wscratch, uint32_t& my_matched, uint32_t& my_cand) {
  const RangeI32 r32 = make_range_i32(L.lo, L.hi);
  const uint32_t woff = ((uint32_t)lane * (uint32_t)B) & ~3u;
  const uint32_t wsel = oct_selector(((uint32_t)lane * (uint32_t)B) & 3u);
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int last_wt = p.n_wtiles - 1;
  constexpr uint32_t TILE_BYTES = (PG_WAVE_DOCS / 8) * B, SUB_BYTES = (OCT_SUB_DOCS / 8) * B;
  const int n_index = uniform(p.n_index_instr);

  auto issue = [&](int wt, u32x4 (&a)[8]) __attribute__((always_inline)) {
    const int wc = wt < last_wt ? wt : last_wt;
    const uint64_t base = (uint64_t)L.data + (uint64_t)wc * TILE_BYTES;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    const GAS uint8_t* tb = (const GAS uint8_t*)(((uint64_t)hi << 32) | (uint64_t)lo);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      a[2 * s] = ldnt((const GAS u32x4_a4*)(tb + (uint32_t)s * SUB_BYTES + woff));
      a[2 * s + 1] = ldnt((const GAS u32x4_a4*)(tb + (uint32_t)s * SUB_BYTES + woff + 16u));
    }
  };
  auto finish = [&](int wt, const u32x4 (&a)[8]) __attribute__((always_inline)) {
    const int64_t wbase = (int64_t)wt * PG_WAVE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - wbase;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
    uint32_t cand = valid_lin_mask(n_valid, lane);
    if (n_index > 0) cand = index_program_lin<false>(p, n_index, wt, wbase, cand, wscratch, lane);
    my_cand += (uint32_t)__popc(cand);
    uint32_t out = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      uint32_t id[8];
      oct_decode_wide<B>(a[2 * s], a[2 * s + 1], wsel, id);
      uint32_t m8 = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) m8 |= (uint32_t)in_range_i32(r32, (int32_t)id[j]) << j;
      const uint32_t c = (uint32_t)__shfl((int)cand, 16 * s + (lane >> 2), 64);
      m8 &= (c >> (8 * (lane & 3))) & 0xFFu;
      uint32_t w = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) w |= ((uint32_t)__shfl((int)m8, 4 * (lane & 15) + b, 64) & 0xFFu) << (8 * b);
      if ((lane >> 4) == s) out = w;
    }
    if (r32.empty) out = 0;
    const uint32_t cnt = (uint32_t)__popc(out);
    my_matched += cnt;
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = out;
    if (p.out_tile_counts) {
      const uint32_t wsum = wave_sum_u32(cnt);
      if (lane == 0 && wsum) atomicAdd(&p.out_tile_counts[wt / PG_WTILES_PER_TILE], wsum);
    }
  };

  u32x4 a[2][8];
  const int wt0 = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_mine = wt0 < p.n_wtiles ? (p.n_wtiles - wt0 + step - 1) / step : 0;
  if (n_mine > 0) {
    __builtin_amdgcn_sched_barrier(0); issue(wt0, a[0]);
    __builtin_amdgcn_sched_barrier(0); issue(wt0 + step, a[1]);
    __builtin_amdgcn_sched_barrier(0);
    int i = 0;
    for (; i + 1 < n_mine; i += 2) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        finish(wt0 + (i + k) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
        issue(wt0 + (i + k + 2) * step, a[k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (i < n_mine) finish(wt0 + i * step, a[0]);   // wave-uniform
  }
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_dictrange_fo(const PgQueryPlan p) {
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ uint32_t s_wscratch[PG_WAVES_PER_BLOCK][64];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  __syncthreads();
  const CAS PgScanLeaf& L = cptr(p.scans)[cptr(p.instrs)[p.n_index_instr].arg];
  uint32_t my_matched = 0, my_cand = 0;
  switch (uniform(L.bits)) {   // wave-uniform, once
#define DFO_CASE(B) case B: dict_filter_loop<B>(p, L, lane, wave, s_wscratch[wave], my_matched, my_cand); break;
    DFO_CASE(1) DFO_CASE(2) DFO_CASE(3) DFO_CASE(4) DFO_CASE(5) DFO_CASE(6) DFO_CASE(7) DFO_CASE(8) DFO_CASE(9) DFO_CASE(10) DFO_CASE(11) DFO_CASE(12)
    DFO_CASE(13) DFO_CASE(14) DFO_CASE(15) DFO_CASE(16) DFO_CASE(17) DFO_CASE(18) DFO_CASE(19) DFO_CASE(20) DFO_CASE(21) DFO_CASE(22) DFO_CASE(23)
#undef DFO_CASE
    default: dict_filter_loop<24>(p, L, lane, wave, s_wscratch[wave], my_matched, my_cand); break;
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  if (!p.fast_scan_pushed) {   // a scan behind an index program counts its candidates; a pushed one (no index) every doc, at plan time
    const uint32_t csum = wave_sum_u32(my_cand);
    if (lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);
  }
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
}
