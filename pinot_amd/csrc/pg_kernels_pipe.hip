// pg_fast_i32range_p: the headline shape as a software pipeline, in a translation unit of its own so that its workgroup can be
// smaller than the other kernels' (PG_WAVES_PER_BLOCK is this file's own: fewer wavefronts, each with a bigger register budget).
//
// Shape (PgQueryPlan::pipe_fit): [AND of <= 4 OR-groups of dense postings] AND raw-INT range → LDS-table aggregation over one or two
// <= 8-bit group columns and ONE raw INT column (any number of SUM / MIN / MAX / COUNT accumulators over it; floating accumulators
// stay with pg_fast_i32range_d: hoisted out of the accumulator loop, their 32 doubles and 32 order keys per lane do not fit).
//
// pg_fast_i32range_a / _d walk a tile as a dependent chain — postings → scan column → (value, group columns) → LDS atomics — and rely
// on 16 wavefronts per CU being in different phases to keep HBM busy; the pure-scan probe (profiles/r02_scan_bw_probe.txt) says the
// memory system prefers FEWER, longer streams (4 wavefronts x 8 KB: 7.33 TB/s; 16 x 8 KB: 6.5 TB/s).  Here every wavefront keeps three
// tiles in flight instead: while tile i is aggregated out of registers, tile i+1's scan column and tile i+2's postings are
// travelling, and tile i+1's value / group quads are requested the moment its match mask is known.  Loads return in order, so
// waiting for the oldest group (vmcnt(N)) leaves the younger ones in flight.  Per lane: 32 (scan) + 32 (value) + 16·NG (group
// windows) + 8 (postings) registers of load targets — beyond the 128 a 16-wavefront workgroup allows, which is why this kernel runs
// PG_WAVES_PER_BLOCK <= 8 wavefronts per workgroup.
//
// Loads stay restricted to quads with candidates / matches (lanes without re-read quad 0 of the tile, as everywhere else), tiles
// beyond the segment are clamped to the last tile with an empty mask: the loop body is branch-free apart from the wave-uniform
// skip of an all-zero aggregation.
#ifndef PG_WAVES_PER_BLOCK
#define PG_WAVES_PER_BLOCK 8
#endif
#ifndef PG_PIPE_GRAIN
#define PG_PIPE_GRAIN 8
#endif
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

extern "C" const int pg_pipe_waves_per_block = PG_WAVES_PER_BLOCK;   // the host launches pg_fast_i32range_p with this many wavefronts

// A wave-uniform pointer pinned into SGPRs: the loads below then take the  global_load v, v_offset, s[base:base+1]  form (one 32-bit
// offset register per load) instead of loop-carried 64-bit per-lane addresses, which is what strength reduction makes of them otherwise.
template <typename T> DEVFN const GAS T* sgpr_ptr(const void* ptr) {
  const uint64_t v = (uint64_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const GAS T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}

// the tile's match word of this lane (PgQueryPlan::out_words), addressed SGPR base + 32-bit lane offset: a per-lane 64-bit address kept
// across the main loop cost two vector registers the index-only kernels do not have (they spilled it)
DEVFN void store_match_word(const void* out_words, int wt, int lane, uint32_t word) {
  GAS uint32_t* tile = const_cast<GAS uint32_t*>(sgpr_ptr<uint32_t>((const uint8_t*)out_words + (size_t)wt * 256u));
  *(GAS uint32_t*)((GAS uint8_t*)tile + (uint32_t)lane * 4u) = word;
}

template <int NG>
__device__ __forceinline__ void fast_pipe_i32range_body(const PgQueryPlan& p) {
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
  const RangeI32 r32 = make_range_i32(L.lo, L.hi);
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t rep = (uint32_t)t & (R - 1u);
  const uint32_t stride = (uint32_t)p.n_groups * R;   // slots per op
  const uint8_t* xdata = p.srcs[p.pipe_src].data;
  const int last_wt = p.n_wtiles - 1;
  const int wstride = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  uint32_t my_matched = 0, my_cand = 0;
#ifdef PG_PIPE_BLOCKED   // measurement variant: every workgroup walks one contiguous run of tiles
  const int per_wg = ((p.n_wtiles + (int)gridDim.x - 1) / (int)gridDim.x + PG_WAVES_PER_BLOCK - 1) / PG_WAVES_PER_BLOCK * PG_WAVES_PER_BLOCK;
  const int n_wtiles_loop = min(((int)blockIdx.x + 1) * per_wg, p.n_wtiles);
  const int step = PG_WAVES_PER_BLOCK;
  const int wt_first = (int)blockIdx.x * per_wg + wave;
#else
  const int n_wtiles_loop = p.n_wtiles;
  const int step = wstride;
  const int wt_first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
#endif

  uint32_t pv[8];          // postings dwords (linear layout)
  u32x4 a[8];              // scan column quads
  u32x4 x[8];              // value column quads
  uint32_t g[NG][8][2];    // group column windows

  auto clamp_tile = [&](int wt) { return wt < last_wt ? wt : last_wt; };
  auto issue_postings = [&](int wt) {   // dense postings are addressed arithmetically: dword index = tile * 64 + lane
    const size_t tile_off = (size_t)clamp_tile(wt) * 256u;
#pragma unroll
    for (int j = 0; j < 8; j++) pv[j] = ldnt((const GAS uint32_t*)(sgpr_ptr<uint8_t>(p.dense_ptr[j] + tile_off) + (uint32_t)lane * 4u));
  };
  auto candidates = [&](int wt) -> uint32_t {   // pv → candidate mask in quad layout; empty beyond the segment
    const int64_t rem = wt < n_wtiles_loop ? (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS : 0;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
    uint32_t grp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int gj = p.dense_group[j];
#pragma unroll
      for (int k = 0; k < 4; k++) grp[k] |= gj == k ? pv[j] : 0u;
    }
    uint32_t lin = valid_lin_mask(n_valid, lane);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < p.dense_groups) lin &= ((p.dense_excl >> k) & 1) ? ~grp[k] : grp[k];
    const uint32_t c = lin_to_quad(lin, lane);
    my_cand += (uint32_t)__popc(c);
    return c;
  };
  // quads aggregated per pass over the accumulators (8: the whole tile; smaller keeps fewer hoisted slots / values live)
  constexpr int G = PG_PIPE_GRAIN;
  auto issue_scan = [&](int wt, uint32_t cand, int k0, int n) {
    const GAS uint8_t* tb = sgpr_ptr<uint8_t>(L.data + (size_t)clamp_tile(wt) * (PG_WAVE_DOCS * 4));
#pragma unroll
    for (int k = k0; k < k0 + n; k++) {
      const uint32_t q = ((cand >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
      a[k] = ldnt((const GAS u32x4*)(tb + q * 16u));
    }
  };
  auto test_scan = [&](uint32_t cand) -> uint32_t {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].x)) << (4 * k);
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].y)) << (4 * k + 1);
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].z)) << (4 * k + 2);
      m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].w)) << (4 * k + 3);
    }
    m = r32.empty ? 0u : (m & cand);
    my_matched += (uint32_t)__popc(m);
    return m;
  };
  auto issue_values = [&](int wt, uint32_t m, int k0, int n) {
    const int wc = clamp_tile(wt);
    const GAS uint8_t* xb = sgpr_ptr<uint8_t>(xdata + (size_t)wc * (PG_WAVE_DOCS * 4));
#pragma unroll
    for (int k = k0; k < k0 + n; k++) {
      const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
      x[k] = ldnt((const GAS u32x4*)(xb + q * 16u));
    }
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      const PgGroupCol& gc = p.gcols[gi];
      const GAS uint32_t* tw = sgpr_ptr<uint32_t>((const void*)packed_wtile_base(gc.data, wc, gc.bits));
#pragma unroll
      for (int k = k0; k < k0 + n; k++) {
        const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
        load_packed_quad<true>(tw, q, (uint32_t)gc.bits, g[gi][k]);
      }
    }
  };
  auto aggregate = [&](uint32_t m, int k0) {   // quads k0 .. k0+G-1 of the tile whose quads sit in x / g, matches m
#ifdef PG_PIPE_NO_AGG   // measurement variant (wrong results): the loads alone
    uint32_t acc = 0;
#pragma unroll
    for (int k = k0; k < k0 + G; k++) { acc += x[k].x ^ x[k].y ^ x[k].z ^ x[k].w; for (int gi = 0; gi < NG; gi++) acc += g[gi][k][0] ^ g[gi][k][1]; }
    if (acc == 0x12345678u) atomicAdd(reinterpret_cast<unsigned long long*>(lds_table), 1ULL);
    return;
#endif
    const uint32_t mg = G == 8 ? m : ((m >> (4 * k0)) & ((1u << (4 * (G & 7))) - 1u));
    if (__ballot(mg != 0) == 0) return;   // wave-uniform
    uint32_t sp[G][2];   // packed slots: docs (0,1) and (2,3) of quad k
#pragma unroll
    for (int k = 0; k < G; k++) sp[k][0] = sp[k][1] = rep | (rep << 16);
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      const PgGroupCol& gc = p.gcols[gi];
      const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
      const uint32_t mult = (uint32_t)gc.mult * R;
#pragma unroll
      for (int k = 0; k < G; k++) {
        uint32_t d[4];
        const uint32_t q = ((mg >> (4 * k)) & 0xFu) ? (uint32_t)((k0 + k) * 64 + lane) : 0u;
        decode_packed_quad<true>(g[gi][k0 + k], q, bits, mask, d);
        sp[k][0] += d[0] * mult + ((d[1] * mult) << 16);
        sp[k][1] += d[2] * mult + ((d[3] * mult) << 16);
      }
    }
    uint32_t v[G][4];
#pragma unroll
    for (int k = 0; k < G; k++) { v[k][0] = bswap32(x[k0 + k].x); v[k][1] = bswap32(x[k0 + k].y); v[k][2] = bswap32(x[k0 + k].z); v[k][3] = bswap32(x[k0 + k].w); }
#define PG_SLOT(k, i) ((sp[k][(i) >> 1] >> (((i) & 1) * 16)) & 0xFFFFu)
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      int64_t* base = lds_table + (size_t)o * stride;
      if (op.src < 0) {
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + PG_SLOT(k, i)), 1ULL);
      } else if (op.fn == PG_ACC_SUM) {
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u)
              atomicAdd(reinterpret_cast<unsigned long long*>(base + PG_SLOT(k, i)), (unsigned long long)(int64_t)(int32_t)v[k][i]);
      } else if (op.fn == PG_ACC_MIN) {
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicMin(reinterpret_cast<long long*>(base + PG_SLOT(k, i)), (long long)(int32_t)v[k][i]);
      } else {
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicMax(reinterpret_cast<long long*>(base + PG_SLOT(k, i)), (long long)(int32_t)v[k][i]);
      }
    }
#undef PG_SLOT
  };

  // ---- prologue: tile 0 up to its value loads, tile 1 up to its scan loads, tile 2's postings -------------------------------------
  int wt_cur = wt_first;
  int wt_nxt = wt_cur + step, wt_far = wt_nxt + step;
  issue_postings(wt_cur);
  uint32_t c0 = candidates(wt_cur);
  issue_postings(wt_nxt);
  issue_scan(wt_cur, c0, 0, 8);
  uint32_t m_cur = test_scan(c0);
  issue_values(wt_cur, m_cur, 0, 8);
  uint32_t c_nxt = candidates(wt_nxt);
  issue_postings(wt_far);
  issue_scan(wt_nxt, c_nxt, 0, 8);
  // ---- steady state.  In flight at the top, oldest first: values(cur), postings(far), scan(nxt) — the prologue issues in the same
  // order (the wait counts the compiler derives at the loop head are the more conservative of the two ways in) ---------------------
  while (wt_cur < n_wtiles_loop) {
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt_cur * 64 + lane] = quad_to_lin(m_cur, lane);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += G) aggregate(m_cur, k0);   // waits for values(cur) only: 16 younger loads stay in flight
    const uint32_t m_nxt = test_scan(c_nxt);                   // waits for scan(nxt) (and with it postings(far))
    issue_values(wt_nxt, m_nxt, 0, 8);
    const uint32_t c_far = candidates(wt_far);
    issue_postings(wt_far + step);
    issue_scan(wt_far, c_far, 0, 8);
    wt_cur = wt_nxt; wt_nxt = wt_far; wt_far += step;
    m_cur = m_nxt; c_nxt = c_far;
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  const uint32_t csum = wave_sum_u32(my_cand);
  if (lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);
  __syncthreads();
  flush_workgroup(p, lds_table, s_stat, true, t);
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_fast_i32range_p(const PgQueryPlan p) {
  if (p.n_group_cols == 1) fast_pipe_i32range_body<1>(p);
  else fast_pipe_i32range_body<2>(p);
}

// =====================================================================================================================================
// The same software pipeline for the neighbouring shapes (round 3) — everything that aggregates ONE raw INT column into an LDS table over
// one or two <= 8-bit group columns, whatever the filter's outer shape:
//   HAS_INDEX  the index-only program is the fused dense form (<= 4 AND-ed OR-groups of dense postings), else there is none
//   HAS_SCAN   one raw-INT range scan restricted to the index result (or the whole filter when there is no index)
//   TAIL       one more dense bitmap ANDed in AFTER the scan: the upsert queryableDocIds snapshot of FilterPlanNode.run's outer AND, which
//              must not restrict the scan's candidates (numEntriesScannedInFilter stays the reference's)
//   VSCAN      a second range scan, over the VALUE column itself (WHERE ... AND m < x ... SUM(m)): tested on the value quads when they
//              arrive, no load of its own; its candidates are the first scan's matches (AndDocIdSet applies scans in list order)
//   NP         dense posting pointers loaded per tile (2 when the index program has at most two: the snapshot alone, a single IN)
// pg_fast_i32range_p above is (index, scan, no tail) and stays as it was measured.  With a scan the stage structure is the same (three
// tiles in flight); without one, the registers the scan quads took hold a SECOND set of value / group quads: tile i + 1's loads are
// requested before tile i is aggregated (the loop is unrolled by two: no register rotation, see pg_kernels_part.hip on why a copy of a
// load target drains the pipeline).
// =====================================================================================================================================
template <int NG, bool HAS_INDEX, bool HAS_SCAN, bool TAIL, bool VSCAN = false, int NP = 8>
__device__ __forceinline__ void pipe_general_body(const PgQueryPlan& p) {
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
  const CAS PgScanLeaf& L = cptr(p.scans)[HAS_SCAN ? p.fast_scan : 0];   // only dereferenced when HAS_SCAN
  const RangeI32 r32 = HAS_SCAN ? make_range_i32(L.lo, L.hi) : RangeI32{0, 0u, false};
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t rep = (uint32_t)t & (R - 1u);
  const uint32_t stride = (uint32_t)p.n_groups * R;   // slots per op
  const uint8_t* xdata = p.srcs[p.pipe_src].data;
  const int last_wt = p.n_wtiles - 1;
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int n_wtiles_loop = p.n_wtiles;
  const int wt_first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  uint32_t my_matched = 0, my_cand = 0;

  uint32_t pv[NP];         // postings dwords (linear layout)
  uint32_t pt = 0;         // tail bitmap dword (linear layout)
  uint32_t my_cand2 = 0;   // VSCAN: candidates of the second scan
  const CAS PgScanLeaf& L2 = cptr(p.scans)[VSCAN ? p.pipe_vscan : 0];
  const RangeI32 r32b = VSCAN ? make_range_i32(L2.lo, L2.hi) : RangeI32{0, 0u, false};
  auto clamp_tile = [&](int wt) { return wt < last_wt ? wt : last_wt; };
  auto issue_postings = [&](int wt) {
    const size_t tile_off = (size_t)clamp_tile(wt) * 256u;
    if (HAS_INDEX) {
#pragma unroll
      for (int j = 0; j < NP; j++) pv[j] = ldnt((const GAS uint32_t*)(sgpr_ptr<uint8_t>(p.dense_ptr[j] + tile_off) + (uint32_t)lane * 4u));
    }
    if (TAIL) pt = ldnt((const GAS uint32_t*)(sgpr_ptr<uint8_t>(p.pipe_tail + tile_off) + (uint32_t)lane * 4u));
  };
  // candidate mask of the index program in quad layout (every valid doc without one); tq: the tail bitmap in quad layout
  auto candidates = [&](int wt, uint32_t& tq) -> uint32_t {
    const int64_t rem = wt < n_wtiles_loop ? (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS : 0;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
    uint32_t c;
    if (HAS_INDEX) {
      uint32_t grp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < NP; j++) {
        const int gj = p.dense_group[j];
#pragma unroll
        for (int k = 0; k < 4; k++) grp[k] |= gj == k ? pv[j] : 0u;
      }
      uint32_t lin = valid_lin_mask(n_valid, lane);
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < p.dense_groups) lin &= ((p.dense_excl >> k) & 1) ? ~grp[k] : grp[k];
      c = lin_to_quad(lin, lane);
    } else {
      c = valid_quad_mask(n_valid, lane);
    }
    tq = TAIL ? lin_to_quad(pt & valid_lin_mask(n_valid, lane), lane) : 0xFFFFFFFFu;
    return c;
  };
  auto issue_values = [&](int wt, uint32_t m, u32x4 (&x)[8], uint32_t (&g)[NG][8][2]) {
    const int wc = clamp_tile(wt);
    const GAS uint8_t* xb = sgpr_ptr<uint8_t>(xdata + (size_t)wc * (PG_WAVE_DOCS * 4));
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
      x[k] = ldnt((const GAS u32x4*)(xb + q * 16u));
    }
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      const PgGroupCol& gc = p.gcols[gi];
      const GAS uint32_t* tw = sgpr_ptr<uint32_t>((const void*)packed_wtile_base(gc.data, wc, gc.bits));
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
        load_packed_quad<true>(tw, q, (uint32_t)gc.bits, g[gi][k]);
      }
    }
  };
  auto aggregate = [&](uint32_t mg, const u32x4 (&x)[8], const uint32_t (&g)[NG][8][2]) {   // the whole tile whose quads sit in x / g
    if (__ballot(mg != 0) == 0) return;   // wave-uniform
    uint32_t sp[8][2];   // packed slots: docs (0,1) and (2,3) of quad k
#pragma unroll
    for (int k = 0; k < 8; k++) sp[k][0] = sp[k][1] = rep | (rep << 16);
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      const PgGroupCol& gc = p.gcols[gi];
      const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
      const uint32_t mult = (uint32_t)gc.mult * R;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        uint32_t d[4];
        const uint32_t q = ((mg >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
        decode_packed_quad<true>(g[gi][k], q, bits, mask, d);
        sp[k][0] += d[0] * mult + ((d[1] * mult) << 16);
        sp[k][1] += d[2] * mult + ((d[3] * mult) << 16);
      }
    }
    uint32_t v[8][4];
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k][0] = bswap32(x[k].x); v[k][1] = bswap32(x[k].y); v[k][2] = bswap32(x[k].z); v[k][3] = bswap32(x[k].w); }
#define PG_SLOT(k, i) ((sp[k][(i) >> 1] >> (((i) & 1) * 16)) & 0xFFFFu)
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      int64_t* base = lds_table + (size_t)o * stride;
      if (op.src < 0) {
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + PG_SLOT(k, i)), 1ULL);
      } else if (op.fn == PG_ACC_SUM) {
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u)
              atomicAdd(reinterpret_cast<unsigned long long*>(base + PG_SLOT(k, i)), (unsigned long long)(int64_t)(int32_t)v[k][i]);
      } else if (op.fn == PG_ACC_MIN) {
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicMin(reinterpret_cast<long long*>(base + PG_SLOT(k, i)), (long long)(int32_t)v[k][i]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicMax(reinterpret_cast<long long*>(base + PG_SLOT(k, i)), (long long)(int32_t)v[k][i]);
      }
    }
#undef PG_SLOT
  };

  if (HAS_SCAN) {
    // ---- three tiles in flight: tile i aggregated out of registers, tile i + 1's scan quads and tile i + 2's bitmaps travelling ----
    u32x4 a[8], x[8];
    uint32_t g[NG][8][2];
    auto issue_scan = [&](int wt, uint32_t cand) {
      const GAS uint8_t* tb = sgpr_ptr<uint8_t>(L.data + (size_t)clamp_tile(wt) * (PG_WAVE_DOCS * 4));
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t q = ((cand >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
        a[k] = ldnt((const GAS u32x4*)(tb + q * 16u));
      }
    };
    auto test_scan = [&](uint32_t cand, uint32_t tq) -> uint32_t {
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].x)) << (4 * k);
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].y)) << (4 * k + 1);
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].z)) << (4 * k + 2);
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].w)) << (4 * k + 3);
      }
      my_cand += (uint32_t)__popc(cand);
      m = r32.empty ? 0u : (m & cand & tq);
      if (!VSCAN) my_matched += (uint32_t)__popc(m);
      return m;
    };
    auto value_scan = [&](uint32_t m1) -> uint32_t {   // the second scan, on the value quads of the current tile
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        m |= (uint32_t)in_range_i32(r32b, (int32_t)bswap32(x[k].x)) << (4 * k);
        m |= (uint32_t)in_range_i32(r32b, (int32_t)bswap32(x[k].y)) << (4 * k + 1);
        m |= (uint32_t)in_range_i32(r32b, (int32_t)bswap32(x[k].z)) << (4 * k + 2);
        m |= (uint32_t)in_range_i32(r32b, (int32_t)bswap32(x[k].w)) << (4 * k + 3);
      }
      my_cand2 += (uint32_t)__popc(m1);
      m = r32b.empty ? 0u : (m & m1);
      my_matched += (uint32_t)__popc(m);
      return m;
    };
    int wt_cur = wt_first, wt_nxt = wt_cur + step, wt_far = wt_nxt + step;
    uint32_t tq0 = 0, tq_nxt = 0, tq_far = 0;
    if (HAS_INDEX || TAIL) issue_postings(wt_cur);
    uint32_t c0 = candidates(wt_cur, tq0);
    if (HAS_INDEX || TAIL) issue_postings(wt_nxt);
    issue_scan(wt_cur, c0);
    uint32_t m_cur = test_scan(c0, tq0);
    issue_values(wt_cur, m_cur, x, g);
    uint32_t c_nxt = candidates(wt_nxt, tq_nxt);
    if (HAS_INDEX || TAIL) issue_postings(wt_far);
    issue_scan(wt_nxt, c_nxt);
    while (wt_cur < n_wtiles_loop) {
      if (VSCAN) m_cur = value_scan(m_cur);                        // waits for values(cur), as the aggregation does
      if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt_cur * 64 + lane] = quad_to_lin(m_cur, lane);
      aggregate(m_cur, x, g);                                      // waits for values(cur) only
      const uint32_t m_nxt = test_scan(c_nxt, tq_nxt);             // waits for scan(nxt)
      issue_values(wt_nxt, m_nxt, x, g);
      const uint32_t c_far = candidates(wt_far, tq_far);
      if (HAS_INDEX || TAIL) issue_postings(wt_far + step);
      issue_scan(wt_far, c_far);
      wt_cur = wt_nxt; wt_nxt = wt_far; wt_far += step;
      m_cur = m_nxt; c_nxt = c_far; tq_nxt = tq_far;
    }
  } else {
    // ---- two sets of value / group quads: tile i + 1's loads are requested before tile i is aggregated ----------------------------------
    u32x4 xa[8], xb[8];
    uint32_t ga[NG][8][2], gb[NG][8][2];
    int wt_a = wt_first;
    uint32_t tq = 0;
    if (HAS_INDEX || TAIL) issue_postings(wt_a);
    uint32_t m_a = candidates(wt_a, tq) & tq;
    if (HAS_INDEX || TAIL) issue_postings(wt_a + step);
    issue_values(wt_a, m_a, xa, ga);
    while (wt_a < n_wtiles_loop) {
      const int wt_b = wt_a + step;
      const uint32_t m_b = candidates(wt_b, tq) & tq;
      if (HAS_INDEX || TAIL) issue_postings(wt_b + step);
      issue_values(wt_b, m_b, xb, gb);
      my_matched += (uint32_t)__popc(m_a);
      if (p.out_words) store_match_word(p.out_words, wt_a, lane, quad_to_lin(m_a, lane));
      aggregate(m_a, xa, ga);                                      // waits for values(a): values(b) and the next bitmaps keep travelling
      wt_a = wt_b + step;
      m_a = candidates(wt_a, tq) & tq;
      if (HAS_INDEX || TAIL) issue_postings(wt_a + step);
      issue_values(wt_a, m_a, xa, ga);
      if (wt_b < n_wtiles_loop) {   // wave-uniform
        my_matched += (uint32_t)__popc(m_b);
        if (p.out_words) store_match_word(p.out_words, wt_b, lane, quad_to_lin(m_b, lane));
      }
      aggregate(m_b, xb, gb);
    }
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  if (HAS_SCAN && !p.fast_scan_pushed) {
    const uint32_t csum = wave_sum_u32(my_cand);
    if (lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);
  }
  if (VSCAN) {
    const uint32_t csum = wave_sum_u32(my_cand2);
    if (lane == 0 && csum) atomicAdd(&s_stat[L2.stat_slot], csum);
  }
  __syncthreads();
  flush_workgroup(p, lds_table, s_stat, true, t);
}
#define PG_PIPE_KERNEL(NAME, IDX, SCAN, TAILF, ...) \
  extern "C" __global__ void __launch_bounds__(PG_BLOCK) NAME(const PgQueryPlan p) { \
    if (p.n_group_cols == 1) pipe_general_body<1, IDX, SCAN, TAILF, ##__VA_ARGS__>(p); \
    else pipe_general_body<2, IDX, SCAN, TAILF, ##__VA_ARGS__>(p); \
  }
PG_PIPE_KERNEL(pg_pipe_scan, false, true, false)            // the range scan is the whole filter
PG_PIPE_KERNEL(pg_pipe_scan_tail, false, true, true)        // ... behind an upsert snapshot
PG_PIPE_KERNEL(pg_pipe_index_scan_tail, true, true, true)   // config 3 behind an upsert snapshot
PG_PIPE_KERNEL(pg_pipe_none, false, false, false)           // no filter
PG_PIPE_KERNEL(pg_pipe_tail, false, false, true)            // only the upsert snapshot
PG_PIPE_KERNEL(pg_pipe_index, true, false, false)           // inverted-index leaves only
PG_PIPE_KERNEL(pg_pipe_index_tail, true, false, true)
PG_PIPE_KERNEL(pg_pipe_index2, true, false, false, false, 2)        // ... with at most two dense posting pointers (one leaf: the snapshot alone)
PG_PIPE_KERNEL(pg_pipe_index2_tail, true, false, true, false, 2)
PG_PIPE_KERNEL(pg_pipe_scan_vscan, false, true, false, true)        // range scan AND a range on the value column
PG_PIPE_KERNEL(pg_pipe_index_scan_vscan, true, true, false, true)   // dense index AND range scan AND a range on the value column

// =====================================================================================================================================
// The pipeline for WIDE columns (round 4): a raw LONG value column (8 bytes per doc) and / or group columns of 9 .. 16 bits — the shapes
// pg_fast_none_w / pg_fast_multi_w walked as a dependent chain (46-57 % of 8 TB/s, profiles/r03_m_variants_wide_100m.txt).  Every value
// accumulator reads ONE raw INT or LONG column; zero to two group columns of <= 16 bits; integer accumulators (SumAggregationFunction
// .java:160-179 over LONG sources: exact in int64 while the planner's bound holds).
//
// The unit in flight is a QUARTER of a wave tile (two quads, 512 docs): a LONG column's quad is 32 bytes per lane, so the four quarters
// of a tile are the 64 load-target registers two tiles of a 32-bit column take; the quarters are the four buffers of the software
// pipeline — three travel while one is aggregated.  A lane's 32 bytes are two 16-byte loads at a lane
// stride of 32 bytes: 6.3 TB/s against 6.6-6.9 TB/s fully coalesced; 64 contiguous bytes per lane (an "8 docs per lane" ownership) drops
// to 4.0 TB/s (profiles/r04_lane_stride_probe.txt), which is why this kernel keeps the quad layout.  The filter stages run per whole
// tile as in pipe_general_body: dense postings, one raw-INT range scan.  Group columns take three-dword windows (4 x 16 bits + 31 bits
// of misalignment).
// =====================================================================================================================================
typedef uint32_t u32x3w __attribute__((ext_vector_type(3)));
typedef u32x3w u32x3w_a4 __attribute__((aligned(4)));
DEVFN void load_packed_quad_mid(const GAS uint32_t* __restrict__ tw, uint32_t q, uint32_t bits, uint32_t (&r)[3]) {
  const uint32_t di = __umul24(4u * q, bits) >> 5;   // q < 512, bits <= 16
  const u32x3w v = ldnt((const GAS u32x3w_a4*)(tw + di));
  r[0] = v.x; r[1] = v.y; r[2] = v.z;
}
DEVFN void decode_packed_quad_mid(const uint32_t (&r)[3], uint32_t q, uint32_t bits, uint32_t mask, uint32_t (&out)[4]) {
  const uint32_t sh = __umul24(4u * q, bits) & 31u;
  const uint32_t w0 = bswap32(r[0]), w1 = bswap32(r[1]), w2 = bswap32(r[2]);
  // the quad's 4 x bits <= 64 bits, left-aligned: (w0:w1:w2) << sh, upper 64 bits (two funnel shifts)
  const uint32_t hi = __builtin_amdgcn_alignbit(w0, w1, 32u - sh), lo = __builtin_amdgcn_alignbit(w1, w2, 32u - sh);
  const uint32_t h = sh ? hi : w0, l = sh ? lo : w1;   // alignbit by 32 is a shift by 0 of the LOW operand
  const uint64_t top = ((uint64_t)h << 32) | (uint64_t)l;
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = (uint32_t)(top >> (64u - (uint32_t)(i + 1) * bits)) & mask;
}

// Every lambda is always_inline: one left out of line takes the plan by address, and hipcc then copies the whole kernel argument (2 KB per
// lane) into scratch memory and reads it from there.
// VK: the value column — 0: none (COUNT alone), 1: raw INT, 2: raw LONG, 3: raw DOUBLE (SUMs as fixed-point digits, pg_fixed_point.h: one
// int64 accumulator per base-2^32 digit; MIN / MAX through order-preserving int64 keys, NaN never replaces); NG: group columns.  Both compile-time: a wave-uniform run-time branch around the
// first use of a load target makes hipcc wait for EVERY load in flight there (s_waitcnt vmcnt(0)) — the first cut of this kernel, with
// `if (gi < n_group_cols)` around the decodes, had no wait count above 3 and ran slower with four buffers than with two.
template <int VK, int NG, bool HAS_INDEX, bool HAS_SCAN>
__device__ __forceinline__ void pipe_wide_body(const PgQueryPlan& p) {
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
      const int64_t ident = pg_acc_identity(p.ops[uniform(o)].fn, p.ops[uniform(o)].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
  }
  __syncthreads();
  const CAS PgScanLeaf& L = cptr(p.scans)[HAS_SCAN ? p.fast_scan : 0];   // only dereferenced when HAS_SCAN
  const RangeI32 r32 = HAS_SCAN ? make_range_i32(L.lo, L.hi) : RangeI32{0, 0u, false};
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t rep = (uint32_t)t & (R - 1u);
  const uint32_t stride = (uint32_t)p.n_groups * R;   // slots per op
  constexpr int VW = VK == 0 ? 0 : (VK == 1 ? 1 : 2);   // dwords per value
  constexpr bool DBL = VK == 3;
  constexpr int XW = VW > 0 ? VW : 1;       // (array extents; with VW == 0 nothing is loaded into them)
  constexpr int GN = NG > 0 ? NG : 1;
  constexpr bool has_value = VW > 0;
  const uint8_t* xdata = has_value ? p.srcs[p.pipe_src].data : nullptr;
  const int last_wt = p.n_wtiles - 1;
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int n_wtiles_loop = p.n_wtiles;
  const int wt_first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  uint32_t my_matched = 0, my_cand = 0;

  uint32_t pv[8];
  auto clamp_tile = [&](int wt) __attribute__((always_inline)) { return wt < last_wt ? wt : last_wt; };
  auto issue_postings = [&](int wt) __attribute__((always_inline)) {
    if (!HAS_INDEX) return;
    const size_t tile_off = (size_t)clamp_tile(wt) * 256u;
#pragma unroll
    for (int j = 0; j < 8; j++) pv[j] = ldnt((const GAS uint32_t*)(sgpr_ptr<uint8_t>(p.dense_ptr[j] + tile_off) + (uint32_t)lane * 4u));
  };
  auto candidates = [&](int wt) __attribute__((always_inline)) -> uint32_t {   // index program -> candidate mask in quad layout (every valid doc without one)
    const int64_t rem = wt < n_wtiles_loop ? (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS : 0;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
    if (!HAS_INDEX) return valid_quad_mask(n_valid, lane);
    uint32_t grp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int gj = p.dense_group[j];
#pragma unroll
      for (int k = 0; k < 4; k++) grp[k] |= gj == k ? pv[j] : 0u;
    }
    uint32_t lin = valid_lin_mask(n_valid, lane);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < p.dense_groups) lin &= ((p.dense_excl >> k) & 1) ? ~grp[k] : grp[k];
    return lin_to_quad(lin, lane);
  };
  // value / group quads of part `h` (quads NQ h .. NQ h + NQ - 1) of tile wt, restricted to quads with matches
  constexpr int NQ = 2;   // quads per part: a QUARTER of a tile — four buffers, three of them in flight while one is aggregated
  auto issue_part = [&](int wt, uint32_t m, int h, u32x4 (&x)[NQ][XW], uint32_t (&g)[GN][NQ][3]) __attribute__((always_inline)) {
    const int wc = clamp_tile(wt);
    if (has_value) {
      const GAS uint8_t* xb = sgpr_ptr<uint8_t>(xdata + (size_t)wc * (size_t)(PG_WAVE_DOCS * 4 * VW));
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int kk = NQ * h + k;
        const uint32_t q = ((m >> (4 * kk)) & 0xFu) ? (uint32_t)(kk * 64 + lane) : 0u;
#pragma unroll
        for (int w = 0; w < VW; w++) x[k][w] = ldnt((const GAS u32x4*)(xb + q * (16u * VW) + 16u * w));
      }
    }
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      {
        const PgGroupCol& gc = p.gcols[gi];
        const GAS uint32_t* tw = sgpr_ptr<uint32_t>((const void*)packed_wtile_base(gc.data, wc, gc.bits));
#pragma unroll
        for (int k = 0; k < NQ; k++) {
          const int kk = NQ * h + k;
          const uint32_t q = ((m >> (4 * kk)) & 0xFu) ? (uint32_t)(kk * 64 + lane) : 0u;
          load_packed_quad_mid(tw, q, (uint32_t)gc.bits, g[gi][k]);
        }
      }
    }
  };
  // no GROUP BY: SUM / MIN / MAX / COUNT of the lane's docs stay in registers for the whole kernel and are folded once at the end (the one
  // group's few replica slots would take every lane's atomics: 44 % of 8 TB/s; folding per part across the wavefront: 33 %,
  // profiles/r04_o_variants_wide_100m.txt)
  // DOUBLE values: the rows of the column's accumulators (-1: the query has none of that kind)
  int wd_row_cnt = -1, wd_row_sum = -1, wd_row_min = -1, wd_row_max = -1;
  if (DBL)
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[uniform(o)];
      if (op.src < 0) wd_row_cnt = o;
      else if (op.fn == PG_ACC_SUM) { if (op.limb == 0) wd_row_sum = o; }
      else if (op.fn == PG_ACC_MIN) wd_row_min = o;
      else wd_row_max = o;
    }
  wd_row_cnt = uniform(wd_row_cnt); wd_row_sum = uniform(wd_row_sum); wd_row_min = uniform(wd_row_min); wd_row_max = uniform(wd_row_max);
  int64_t lane_sum = 0, lane_min = INT64_MAX, lane_max = INT64_MIN;   // (DOUBLE values: order keys in lane_min / lane_max)
  int64_t lane_dig[4] = {0, 0, 0, 0};                                  // DOUBLE sums: the lane's digit sums
  uint32_t lane_cnt = 0;
  auto aggregate_part = [&](uint32_t m, int h, const u32x4 (&x)[NQ][XW], const uint32_t (&g)[GN][NQ][3]) __attribute__((always_inline)) {
    const uint32_t mg = (m >> (4 * NQ * h)) & ((1u << (4 * NQ)) - 1u);
    if (__ballot(mg != 0) == 0) return;   // wave-uniform
    uint32_t slot[NQ][4];
#pragma unroll
    for (int k = 0; k < NQ; k++)
#pragma unroll
      for (int i = 0; i < 4; i++) slot[k][i] = rep;
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      {
        const PgGroupCol& gc = p.gcols[gi];
        const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
        const uint32_t mult = (uint32_t)gc.mult * R;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
          uint32_t d[4];
          const int kk = NQ * h + k;
          const uint32_t q = ((mg >> (4 * k)) & 0xFu) ? (uint32_t)(kk * 64 + lane) : 0u;
          decode_packed_quad_mid(g[gi][k], q, bits, mask, d);
#pragma unroll
          for (int i = 0; i < 4; i++) slot[k][i] += __umul24(d[i], mult);   // < 65536 slots (the planner's bound on n_groups x replicas): a 24-bit multiply-add
        }
      }
    }
    int64_t v[NQ][4] = {};
    if (has_value) {
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        if (VW == 1) {
          v[k][0] = (int64_t)(int32_t)bswap32(x[k][0].x); v[k][1] = (int64_t)(int32_t)bswap32(x[k][0].y);
          v[k][2] = (int64_t)(int32_t)bswap32(x[k][0].z); v[k][3] = (int64_t)(int32_t)bswap32(x[k][0].w);
        } else {
          v[k][0] = (int64_t)(((uint64_t)bswap32(x[k][0].x) << 32) | (uint64_t)bswap32(x[k][0].y));
          v[k][1] = (int64_t)(((uint64_t)bswap32(x[k][0].z) << 32) | (uint64_t)bswap32(x[k][0].w));
          v[k][2] = (int64_t)(((uint64_t)bswap32(x[k][XW - 1].x) << 32) | (uint64_t)bswap32(x[k][XW - 1].y));
          v[k][3] = (int64_t)(((uint64_t)bswap32(x[k][XW - 1].z) << 32) | (uint64_t)bswap32(x[k][XW - 1].w));
        }
      }
    }
    // DOUBLE values: digit j (base 2^32) of X = trunc(|x| 2^-q) is bits [32 j, 32 j + 32) of the 53-bit mantissa m shifted by s = exponent - q,
    // i.e. a 32-bit window of (0 : m_hi : m_lo : 0) at bit offset t = 32 j - s.  t mod 32 is the same for every j, so the three windows that
    // can hold mantissa bits are cut ONCE per doc (three funnel shifts) and a digit is a select by (j - ceil(s / 32)) — 32-bit operations
    // only (the first cut shifted the 64-bit mantissa per digit: 84 VALU per doc, profiles/r04_q_*).
    const int fxq = DBL ? p.srcs[p.pipe_src].fx_q : 0;
    // the three windows of value bits b, the digit index j0 of the first, the sign mask (computed where they are used: once per doc)
    auto windows = [&](uint64_t b, uint32_t (&w)[3], int& j0, uint64_t& neg) __attribute__((always_inline)) {
      const uint32_t bh = (uint32_t)(b >> 32), ml = (uint32_t)b;
      const int e = (int)((bh >> 20) & 0x7FFu);
      const uint32_t mh = (bh & 0xFFFFFu) | (e ? (1u << 20) : 0u);
      const int sft = (e ? e - 1075 : -1074) - fxq;             // X = m * 2^sft
      j0 = sft >> 5;                                            // floor(sft / 32): digit j0 holds m's bit 0 at bit (sft & 31)
      const uint32_t sl = (uint32_t)sft & 31u;                  // the windows are (0 : mh : ml) << sl, cut at dword boundaries
      const uint64_t low = (((uint64_t)mh << 32) | (uint64_t)ml) << sl;
      w[0] = (uint32_t)low;
      w[1] = (uint32_t)(low >> 32);
      w[2] = (mh >> 1) >> (31u - sl);                           // mh >> (32 - sl) without the shift by 32 (sl = 0: nothing spills over)
      neg = (uint64_t)(int64_t)((int32_t)bh >> 31);             // all ones for negative values
    };
    if (NG == 0) {
      lane_cnt += (uint32_t)__popc(mg);
      if (DBL) {
#pragma unroll
        for (int k = 0; k < NQ; k++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const bool on = ((mg >> (4 * k + i)) & 1u) != 0;
            const double y = __longlong_as_double(v[k][i]);
            const int64_t key = f64_order_key(y);
            uint32_t w[3];
            int j0;
            uint64_t neg;
            windows((uint64_t)v[k][i], w, j0, neg);
#pragma unroll
            for (int l = 0; l < 4; l++) {
              const int d = l - j0;
              const uint32_t u = d == 0 ? w[0] : (d == 1 ? w[1] : (d == 2 ? w[2] : 0u));
              lane_dig[l] += on ? (int64_t)(((uint64_t)u ^ neg) - neg) : 0;
            }
            lane_min = on && y == y && key < lane_min ? key : lane_min;
            lane_max = on && y == y && key > lane_max ? key : lane_max;
          }
        return;
      }
#pragma unroll
      for (int k = 0; k < NQ; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const bool on = ((mg >> (4 * k + i)) & 1u) != 0;
          const int64_t y = v[k][i];
          lane_sum += on ? y : 0;
          lane_min = on && y < lane_min ? y : lane_min;
          lane_max = on && y > lane_max ? y : lane_max;
        }
      return;
    }
    if (DBL) {
      // DOUBLE values, doc by doc: the accumulators of ONE double column are at most COUNT, the four digit rows of its SUM, MIN and MAX (the
      // planner shares equal accumulators), so a doc's work is four wave-uniform tests on row offsets instead of a loop over the accumulators
      // with the docs inside.  hipcc hoisted everything a doc derives from its value — windows, signs, order keys: 10 registers x 8 docs —
      // out of that loop, which is what pushed pg_pipe_wd_scan / pg_pipe_wd_index_scan into scratch memory (156 / 328 bytes per lane,
      // profiles/r04: VERDICT r4 #8); here those values live for one doc.
#pragma unroll
      for (int k = 0; k < NQ; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (!((mg >> (4 * k + i)) & 1u)) continue;   // one exec-mask change per doc
          int64_t* cell = lds_table + slot[k][i];
          if (wd_row_cnt >= 0) atomicAdd(reinterpret_cast<unsigned long long*>(cell + (uint32_t)wd_row_cnt * stride), 1ULL);
          if (wd_row_sum >= 0) {
            uint32_t w[3];
            int j0;
            uint64_t neg;
            windows((uint64_t)v[k][i], w, j0, neg);
            int64_t* digits = cell + (uint32_t)wd_row_sum * stride;
#pragma unroll
            for (int d = 0; d < 3; d++) {   // a window outside rows 0 .. 3 adds 0 to row 0
              const int row = j0 + d;
              const bool in = (uint32_t)row < 4u;
              const uint32_t u = in ? w[d] : 0u;
              // (24-bit multiplies: a row index x at most 65536 slots; v_mul_lo_u32 runs at a quarter of the rate)
              atomicAdd(reinterpret_cast<unsigned long long*>(digits + __umul24(in ? (uint32_t)row : 0u, stride)), (unsigned long long)(((uint64_t)u ^ neg) - neg));
            }
          }
          if (wd_row_min >= 0 || wd_row_max >= 0) {   // order-preserving keys; NaN never replaces the holder (Java: NaN < x is false)
            const double y = __longlong_as_double(v[k][i]);
            if (y == y) {
              const long long key = (long long)f64_order_key(y);
              if (wd_row_min >= 0) atomicMin(reinterpret_cast<long long*>(cell + (uint32_t)wd_row_min * stride), key);
              if (wd_row_max >= 0) atomicMax(reinterpret_cast<long long*>(cell + (uint32_t)wd_row_max * stride), key);
            }
          }
        }
      return;
    }
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[uniform(o)];   // (a scalar index: a vector one makes hipcc copy the whole plan into scratch memory)
      int64_t* base = lds_table + (size_t)o * stride;
      if (op.src < 0) {
#pragma unroll
        for (int k = 0; k < NQ; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[k][i]), 1ULL);
      } else if (op.fn == PG_ACC_SUM) {
#pragma unroll
        for (int k = 0; k < NQ; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[k][i]), (unsigned long long)v[k][i]);
      } else if (op.fn == PG_ACC_MIN) {
#pragma unroll
        for (int k = 0; k < NQ; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicMin(reinterpret_cast<long long*>(base + slot[k][i]), (long long)v[k][i]);
      } else {
#pragma unroll
        for (int k = 0; k < NQ; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mg >> (4 * k + i)) & 1u) atomicMax(reinterpret_cast<long long*>(base + slot[k][i]), (long long)v[k][i]);
      }
    }
  };

  // four part buffers; at the top of a tile's iteration parts 0 .. 2 of the tile are in flight, part 3 is requested first, and each
  // buffer is re-requested for the NEXT tile right after it has been aggregated — three parts travel while one is aggregated (with two
  // half-tile buffers the oldest request was one aggregation old when it was needed: 59 % of the wave cycles waiting,
  // profiles/r04_l_sq_w_none.txt)
  u32x4 x0[NQ][XW], x1[NQ][XW], x2[NQ][XW], x3[NQ][XW];
  uint32_t g0[GN][NQ][3], g1[GN][NQ][3], g2[GN][NQ][3], g3[GN][NQ][3];
  if (HAS_SCAN) {
    // per tile: scan quads of the NEXT tile and the bitmaps of the one after travel while the current tile's parts are aggregated
    u32x4 a[8];
    auto issue_scan = [&](int wt, uint32_t cand) __attribute__((always_inline)) {
      const GAS uint8_t* tb = sgpr_ptr<uint8_t>(L.data + (size_t)clamp_tile(wt) * (PG_WAVE_DOCS * 4));
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t q = ((cand >> (4 * k)) & 0xFu) ? (uint32_t)(k * 64 + lane) : 0u;
        a[k] = ldnt((const GAS u32x4*)(tb + q * 16u));
      }
    };
    auto test_scan = [&](uint32_t cand) __attribute__((always_inline)) -> uint32_t {
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].x)) << (4 * k);
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].y)) << (4 * k + 1);
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].z)) << (4 * k + 2);
        m |= (uint32_t)in_range_i32(r32, (int32_t)bswap32(a[k].w)) << (4 * k + 3);
      }
      my_cand += (uint32_t)__popc(cand);
      m = r32.empty ? 0u : (m & cand);
      my_matched += (uint32_t)__popc(m);
      return m;
    };
    int wt_cur = wt_first, wt_nxt = wt_cur + step, wt_far = wt_nxt + step;
    issue_postings(wt_cur);
    uint32_t c0 = candidates(wt_cur);
    issue_postings(wt_nxt);
    issue_scan(wt_cur, c0);
    uint32_t m_cur = test_scan(c0);
    issue_part(wt_cur, m_cur, 0, x0, g0);
    issue_part(wt_cur, m_cur, 1, x1, g1);
    issue_part(wt_cur, m_cur, 2, x2, g2);
    uint32_t c_nxt = candidates(wt_nxt);
    issue_postings(wt_far);
    issue_scan(wt_nxt, c_nxt);
    while (wt_cur < n_wtiles_loop) {
      if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt_cur * 64 + lane] = quad_to_lin(m_cur, lane);
      issue_part(wt_cur, m_cur, 3, x3, g3);
      aggregate_part(m_cur, 0, x0, g0);
      const uint32_t m_nxt = test_scan(c_nxt);            // waits for scan(nxt) (requested a whole tile ago)
      issue_part(wt_nxt, m_nxt, 0, x0, g0);
      const uint32_t c_far = candidates(wt_far);
      issue_postings(wt_far + step);
      issue_scan(wt_far, c_far);
      aggregate_part(m_cur, 1, x1, g1);
      issue_part(wt_nxt, m_nxt, 1, x1, g1);
      aggregate_part(m_cur, 2, x2, g2);
      issue_part(wt_nxt, m_nxt, 2, x2, g2);
      aggregate_part(m_cur, 3, x3, g3);
      wt_cur = wt_nxt; wt_nxt = wt_far; wt_far += step;
      m_cur = m_nxt; c_nxt = c_far;
    }
  } else {
    int wt = wt_first;
    issue_postings(wt);
    uint32_t m = candidates(wt);
    issue_postings(wt + step);
    issue_part(wt, m, 0, x0, g0);
    issue_part(wt, m, 1, x1, g1);
    issue_part(wt, m, 2, x2, g2);
    while (wt < n_wtiles_loop) {
      my_matched += (uint32_t)__popc(m);
      if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
      issue_part(wt, m, 3, x3, g3);
      aggregate_part(m, 0, x0, g0);
      const int wt_n = wt + step;
      const uint32_t m_n = candidates(wt_n);              // waits for the bitmaps of tile wt_n (requested a whole tile ago)
      issue_postings(wt_n + step);
      issue_part(wt_n, m_n, 0, x0, g0);
      aggregate_part(m, 1, x1, g1);
      issue_part(wt_n, m_n, 1, x1, g1);
      aggregate_part(m, 2, x2, g2);
      issue_part(wt_n, m_n, 2, x2, g2);
      aggregate_part(m, 3, x3, g3);
      wt = wt_n;
      m = m_n;
    }
  }
  if (NG == 0) {
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[uniform(o)];
      const int kind = op.src < 0 || op.fn == PG_ACC_SUM ? 0 : (op.fn == PG_ACC_MIN ? 1 : 2);
      int64_t acc = op.src < 0 ? (int64_t)lane_cnt : (kind == 0 ? lane_sum : (kind == 1 ? lane_min : lane_max));
      if (DBL && op.src >= 0 && kind == 0) acc = op.limb == 0 ? lane_dig[0] : (op.limb == 1 ? lane_dig[1] : (op.limb == 2 ? lane_dig[2] : lane_dig[3]));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int64_t y = __shfl_xor((long long)acc, off, 64);
        acc = kind == 0 ? acc + y : (kind == 1 ? (y < acc ? y : acc) : (y > acc ? y : acc));
      }
      int64_t* slot = lds_table + (size_t)o * stride + rep;
      if (lane == 0) {
        if (kind == 0) atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)acc);
        else if (kind == 1) atomicMin(reinterpret_cast<long long*>(slot), (long long)acc);
        else atomicMax(reinterpret_cast<long long*>(slot), (long long)acc);
      }
    }
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  if (HAS_SCAN && !p.fast_scan_pushed) {
    const uint32_t csum = wave_sum_u32(my_cand);
    if (lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);
  }
  __syncthreads();
  // statistics and the flush of the LDS table into this workgroup's partial table — flush_workgroup's work with the accumulator loop
  // outermost: its per-thread p.ops[i / groups] is a vector index into the kernel argument, which made hipcc copy the whole plan (2 KB
  // per lane) into scratch memory in the kernels with an index program
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  {
    const int groups = p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * ((int64_t)p.n_ops * groups);
    for (int o = 0; o < p.n_ops; o++) {
      const int fn = p.ops[uniform(o)].fn;   // integer accumulators only (planner)
      if (R >= 64u) {   // a wavefront per slot folds its replicas (flush_workgroup: one lane walking 1 024 replicas per accumulator is 30 us)
        for (int gq = wave; gq < groups; gq += PG_BLOCK / 64) {
          const int64_t* src = lds_table + ((size_t)o * (size_t)groups + (size_t)gq) * R;
          int64_t acc = src[lane];
          if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (uint32_t r = (uint32_t)lane + 64u; r < R; r += 64u) acc += src[r]; }
          else if (fn == PG_ACC_MIN) { for (uint32_t r = (uint32_t)lane + 64u; r < R; r += 64u) acc = src[r] < acc ? src[r] : acc; }
          else { for (uint32_t r = (uint32_t)lane + 64u; r < R; r += 64u) acc = src[r] > acc ? src[r] : acc; }
          acc = wave_fold_i64(acc, fn);
          if (lane == 0) out[(size_t)o * (size_t)groups + (size_t)gq] = acc;
        }
        continue;
      }
      for (int gq = t; gq < groups; gq += PG_BLOCK) {
        const int64_t* src = lds_table + ((size_t)o * (size_t)groups + (size_t)gq) * R;
        int64_t acc = src[0];
        if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (uint32_t r = 1; r < R; r++) acc += src[r]; }
        else if (fn == PG_ACC_MIN) { for (uint32_t r = 1; r < R; r++) acc = src[r] < acc ? src[r] : acc; }
        else { for (uint32_t r = 1; r < R; r++) acc = src[r] > acc ? src[r] : acc; }
        out[(size_t)o * (size_t)groups + (size_t)gq] = acc;
      }
    }
  }
}
#define PG_PIPE_WIDE_KERNEL(NAME, VKV, IDX, SCAN) \
  extern "C" __global__ void __launch_bounds__(PG_BLOCK) NAME(const PgQueryPlan p) { \
    const int ng = p.n_group_cols; \
    if (ng == 0 && (VKV) != 0) pipe_wide_body<VKV, ((VKV) != 0 ? 0 : 1), IDX, SCAN>(p); \
    else if (ng <= 1) pipe_wide_body<VKV, 1, IDX, SCAN>(p); \
    else pipe_wide_body<VKV, 2, IDX, SCAN>(p); \
  }
// one kernel per value kind (0: COUNT alone; raw INT; raw LONG; raw DOUBLE) and filter shape: three bodies each (with all the widths x group
// counts in one kernel hipcc kept the plan in scratch memory in the kernels with an index program)
PG_PIPE_WIDE_KERNEL(pg_pipe_w0_none, 0, false, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_w0_index, 0, true, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_w0_scan, 0, false, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_w0_index_scan, 0, true, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_w32_none, 1, false, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_w32_index, 1, true, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_w32_scan, 1, false, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_w32_index_scan, 1, true, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_w64_none, 2, false, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_w64_index, 2, true, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_w64_scan, 2, false, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_w64_index_scan, 2, true, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_wd_none, 3, false, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_wd_index, 3, true, false)
PG_PIPE_WIDE_KERNEL(pg_pipe_wd_scan, 3, false, true)
PG_PIPE_WIDE_KERNEL(pg_pipe_wd_index_scan, 3, true, true)
