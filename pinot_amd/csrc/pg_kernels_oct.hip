// "Oct layout" kernels — DISTINCTCOUNTHLL / DISTINCTCOUNT next to a small group key, VALU-lean (round 4).
//
// Reference work: DistinctCountHLLAggregationFunction#aggregateGroupBySV (core/query/aggregation/function/
// DistinctCountHLLAggregationFunction.java:152-222: hll.offer(value) per doc into the group's HyperLogLog), BaseDistinctAggregate-
// AggregationFunction's dictId bitmaps (:306-345), CountAggregationFunction#aggregateGroupBySV (:110-143) and the raw keys of
// DictionaryBasedGroupKeyGenerator (:312-354, key = sum dictId_j x prod card_i) over <= 8-bit dictionary columns.
//
// Why another kernel family.  These shapes move few bytes per doc (config 5: 4.375 B) and are bound by instruction issue, not by HBM
// (profiles/r03_n_sq_counters_cfg5_200m.txt: 121 VALU per doc in the partition scatter; the interpreter spends 1.9 k VALU per wave tile
// visit on `hll(u) GROUP BY h1`).  The quad layout of the other kernels (lane L owns quads k*64+L of 4 docs) makes a lane fetch a 64-bit
// window and run a 64-bit shift for FOUR values of a 4-bit column.  Here lane L owns the 8 CONSECUTIVE docs 8L .. 8L+7 of a 512-doc
// sub-tile: 8 values of a b-bit column are exactly b bytes at byte offset b x L — for b = 4 one dword per lane and 8 bit-field
// extracts, for any width one byte permute (v_perm_b32: alignment + endianness in one instruction, its selector a per-lane constant)
// per dword and compile-time field positions after it.  A wave tile (2 048 docs, the unit of match words and tile padding) is four
// sub-tiles; two sub-tiles' loads are in flight per wavefront.
//
// The hash.  stream-lib's MurmurHash.hashLong of an INT value v is  h = k x m^2 ^ C;  h ^= h >>> 13;  h *= m;  h ^= h >>> 15  with
// k = (v x m) ^ ((v x m) >>> 24) and C a constant of the sign of v.  For an arithmetic dictionary (value = base + step x dictId, found at
// registration) v x m = base x m + dictId x (step x m) is two full-rate 24-bit multiply-adds of constants; two 32-bit multiplies
// (quarter rate) remain.
//
// Two back ends:
//   pg_oct_l / _lm   states fit the workgroup's LDS (planner: CompiledPlan::aux_in_lds, e.g. 160 groups x 256 one-byte registers):
//                    a register is read first and compare-and-swapped only when the offer would raise it; DISTINCTCOUNT sets take
//                    ds_or_b32; COUNT the low dword of its int64 slot.  Same LDS layout, partial areas and merge kernels as
//                    pg_generic_query_l, which ran these plans before (5.6 % of the roofline).
//   pg_oct_p / _pm   key space fits LDS but the registers do not (config 5 flat: 12 800 groups x 256 registers = 3.2 MB): PRUNED OFFERS.
//                    A HyperLogLog register only ever grows, so an offer (group, index, rank) with rank <= floor[group] — the smallest
//                    register of the group after the docs aggregated so far — cannot change anything and is dropped on the spot; only
//                    the survivors travel through the partition pipeline (pg_kernels_part.hip).  The segment is walked in a few
//                    passes of growing size (2 %, 6 %, 22 %, 70 % of the docs); after each pass the registers are merged and the
//                    floors recomputed, so later passes drop ~82 % / ~95 % of their offers (13.7 % of all offers survive for uniformly
//                    distributed values; a group whose values are few keeps floor 0 and loses nothing but speed).  COUNT needs every
//                    doc and is kept in an LDS table of 32-bit counters by this kernel, flushed once per pass.
//                    Survivors are appended to a tuple stream in HBM (key << payload bits | index | rank << log2m) in blocks of 1 024
//                    entries claimed per wavefront with one global atomic; the stream is the input of pg_p2_scatter_stream.
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

#include "pg_oct_layout.h"

// one HyperLogLog register (a byte in LDS) raised to `rank`
DEVFN void oct_raise_register(uint32_t* lds_words, uint32_t byte_addr, uint32_t rank) {
  uint32_t* w = lds_words + (byte_addr >> 2);
  const uint32_t sh = (byte_addr & 3u) * 8u;
  uint32_t cur = *reinterpret_cast<volatile uint32_t*>(w);
  while (((cur >> sh) & 0xFFu) < rank) {
    const uint32_t nv = (cur & ~(0xFFu << sh)) | (rank << sh);
    const uint32_t prev = atomicCAS(w, cur, nv);
    if (prev == cur) break;
    cur = prev;
  }
}

// HyperLogLog registers are bytes: a register is raised by compare-and-swap on its dword.  Batched over four of the lane's docs — four
// dword reads in flight, then four compare-and-swaps in flight for the docs whose rank beats the register as read — so a sub-tile costs four
// LDS round trips whatever the number of raises (one read + one CAS chain per doc measured 58 % of the wave's time waiting,
// profiles/r04_b_sq_counters_oct_l_200m.txt; all eight docs at once spilled 96 bytes per lane); a CAS that lost against another writer of the same
// dword is retried in a (rare) serial loop.
template <int J0>
DEVFN void oct_raise_four(uint32_t m8, const uint32_t (&key)[8], const uint32_t (&item)[8], uint32_t* aux_words, uint32_t stride, uint32_t log2m) {
  const uint32_t imask = (1u << log2m) - 1u;
  uint32_t addr[4], w[4];
#pragma unroll
  for (int j = 0; j < 4; j++) addr[j] = key[J0 + j] * stride + (item[J0 + j] & imask);
  volatile uint32_t* words = aux_words;
#pragma unroll
  for (int j = 0; j < 4; j++) w[j] = words[addr[j] >> 2];   // reads first, all in flight
  uint32_t need = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t sh = (addr[j] & 3u) * 8u;
    need |= (uint32_t)((((m8 >> (J0 + j)) & 1u) != 0) && (item[J0 + j] >> log2m) > ((w[j] >> sh) & 0xFFu)) << j;
  }
  if (__builtin_amdgcn_ballot_w64(need != 0) == 0) return;   // wave-uniform: after warm-up most sub-tiles raise nothing
  uint32_t prev[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    prev[j] = w[j];
    if ((need >> j) & 1u) {
      const uint32_t sh = (addr[j] & 3u) * 8u;
      prev[j] = atomicCAS(aux_words + (addr[j] >> 2), w[j], (w[j] & ~(0xFFu << sh)) | ((item[J0 + j] >> log2m) << sh));
    }
  }
  uint32_t again = 0;   // lost against another writer of the dword (same register or a neighbour) and still below the rank
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t sh = (addr[j] & 3u) * 8u;
    again |= (uint32_t)(((need >> j) & 1u) && prev[j] != w[j] && ((prev[j] >> sh) & 0xFFu) < (item[J0 + j] >> log2m)) << j;
  }
  if (__builtin_amdgcn_ballot_w64(again != 0) == 0) return;
#pragma unroll
  for (int j = 0; j < 4; j++)
    if ((again >> j) & 1u) oct_raise_register(aux_words, addr[j], item[J0 + j] >> log2m);
}

// ---- back end L: the states live in this workgroup's LDS ---------------------------------------------------------------------------
DEVFN void oct_apply_lds(const PgQueryPlan& p, uint32_t m8, const uint32_t (&key)[8], const uint32_t (&item)[8], int64_t* table, uint32_t* aux_words,
                         uint32_t rep) {
  const uint32_t rshift = (uint32_t)p.replica_shift;
  if (p.n_ops > 0) {   // COUNT: the low dword of the int64 slot takes the add (a workgroup sees < 2^31 docs)
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((m8 >> j) & 1u) atomicAdd(reinterpret_cast<uint32_t*>(table + (((size_t)key[j] << rshift) + rep)), 1u);
  }
  if (p.n_aux == 0) return;
  const uint32_t stride = (uint32_t)p.aux[0].stride;
  if (p.oct_src_kind == OCT_SRC_DICTID) {   // DISTINCTCOUNT: bit dictId of the group's set
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((m8 >> j) & 1u) atomicOr(aux_words + (size_t)key[j] * stride + (item[j] >> 5), 1u << (item[j] & 31u));
    return;
  }
  // HyperLogLog registers are bytes: a register is raised by compare-and-swap on its dword.  Everything is batched over the lane's 8 docs —
  // 8 dword reads in flight, then 8 compare-and-swaps in flight for the docs whose rank beats the register as read — so a sub-tile costs
  // two LDS round trips whatever the number of raises (one read + one CAS chain per doc measured 58 % of the wave's time waiting,
  // profiles/r04_b_sq_counters_oct_l_200m.txt); a CAS that lost against another writer of the same dword is retried in a (rare) serial loop.
  const uint32_t log2m = (uint32_t)p.oct_log2m;
  if (p.oct_dword) {   // room for a dword per register (planner): one ds_max_u32 per offer, nothing returned
    const uint32_t imask = (1u << log2m) - 1u;
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((m8 >> j) & 1u) atomicMax(aux_words + (key[j] * stride + (item[j] & imask)), item[j] >> log2m);
    return;
  }
  oct_raise_four<0>(m8, key, item, aux_words, stride, log2m);
  oct_raise_four<4>(m8, key, item, aux_words, stride, log2m);
}

typedef uint32_t u32x2o __attribute__((ext_vector_type(2)));
typedef u32x2o u32x2o_a4 __attribute__((aligned(4)));
// Load buffers sized at compile time (back ends L and P): NW dwords per group column (2 when every width is <= 4 bits, else 3) and SW dwords
// of the source (6 up to 21 bits: 8 x 21 bits + 3 bytes of misalignment = 24 bytes; 8 for wider dictionaries and raw INT values) — 15 registers
// for config 5 against OctRaw's 21, which is what lets THREE buffers rotate inside the 128 registers a 16-wavefront workgroup leaves a lane.
// With two, a buffer was requested one turn (decode + hash + pruning, ~2 us with four wavefronts per SIMD) before it was needed: about the
// latency of HBM under load, and the waves waited 52 % of their cycles at 61 % VALU use (profiles/r04_f_sq_counters_cfg5_200m.txt).
template <int NW, int SW> struct OctRawS { uint32_t g[4][NW]; uint32_t s[SW]; uint32_t mw; };
template <bool MASKED, int NW, int SW>
DEVFN void octs_issue(const PgQueryPlan& p, const OctLane& ln, int wt, int sub, int lane, OctRawS<NW, SW>& raw) {
#pragma unroll
  for (int g = 0; g < 4; g++)
    if (g < p.n_group_cols) {
      const PgGroupCol& gc = p.gcols[g];
      const GAS uint8_t* base = gptr<uint8_t>(gc.data) + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)gc.bits + (size_t)sub * (size_t)(OCT_SUB_DOCS / 8) * (size_t)gc.bits;
      if (NW == 2) {
        const u32x2o v = ldnt((const GAS u32x2o_a4*)(base + ln.goff[g]));
        raw.g[g][0] = v.x; raw.g[g][1] = v.y;
      } else {
        const u32x3 v = ldnt((const GAS u32x3_a4*)(base + ln.goff[g]));
        raw.g[g][0] = v.x; raw.g[g][1] = v.y; raw.g[g][NW - 1] = v.z;
      }
    }
  if (p.oct_src_kind != OCT_SRC_NONE) {   // (COUNT(*) alone: no source column)
    const PgValueSrc& V = p.srcs[p.oct_src];
    const uint32_t bits = p.oct_src_kind == OCT_SRC_RAW32 ? 32u : (uint32_t)V.bits;
    const GAS uint8_t* base = gptr<uint8_t>(V.data) + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)bits + (size_t)sub * (size_t)(OCT_SUB_DOCS / 8) * (size_t)bits;
    const u32x4 a = ldnt((const GAS u32x4_a4*)(base + ln.soff));
    raw.s[0] = a.x; raw.s[1] = a.y; raw.s[2] = a.z; raw.s[3] = a.w;
    if (SW == 6) {
      const u32x2o b = ldnt((const GAS u32x2o_a4*)(base + ln.soff + 16u));
      raw.s[4] = b.x; raw.s[SW - 1] = b.y;
    } else {
      const u32x4 b = ldnt((const GAS u32x4_a4*)(base + ln.soff + 16u));
      raw.s[4] = b.x; raw.s[5] = b.y; raw.s[SW - 2] = b.z; raw.s[SW - 1] = b.w;
    }
  }
  if (MASKED) raw.mw = gptr<uint32_t>(p.match_words)[(size_t)wt * 64 + (size_t)sub * 16 + (size_t)(lane >> 2)];
}
// oct_decode over the sized buffer: keys and raw source items of the lane's 8 docs
template <int NW, int SW>
DEVFN void octs_decode(const PgQueryPlan& p, const OctLane& ln, const OctRawS<NW, SW>& raw, uint32_t (&key)[8], uint32_t (&item)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) key[j] = 0;
#pragma unroll
  for (int g = 0; g < 4; g++)
    if (g < p.n_group_cols) {
      uint32_t v[8];
      u32x3 w;
      w.x = raw.g[g][0]; w.y = raw.g[g][1]; w.z = raw.g[g][NW - 1];   // (NW == 2: .z is never read, the widths are <= 4)
      if (NW == 2) {
        switch (p.gcols[g].bits) {   // wave-uniform
          case 1: oct_decode_small<1>(w, ln.gsel[g], v); break;
          case 2: oct_decode_small<2>(w, ln.gsel[g], v); break;
          case 3: oct_decode_small<3>(w, ln.gsel[g], v); break;
          default: oct_decode_small<4>(w, ln.gsel[g], v); break;
        }
      } else {
        oct_decode_group(p.gcols[g].bits, w, ln.gsel[g], v);
      }
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) key[j] = v[j];   // mult of column 0 is 1
      } else {
        const uint32_t mult = (uint32_t)p.gcols[g].mult;
#pragma unroll
        for (int j = 0; j < 8; j++) key[j] = mad24(v[j], mult, key[j]);
      }
    }
  if (p.oct_src_kind == OCT_SRC_NONE) return;
  u32x4 a, b;
  a.x = raw.s[0]; a.y = raw.s[1]; a.z = raw.s[2]; a.w = raw.s[3];
  b.x = raw.s[4]; b.y = raw.s[5]; b.z = raw.s[SW - 2]; b.w = raw.s[SW - 1];   // (SW == 6: .z / .w are never read, the width is <= 21)
  if (SW == 8 && p.oct_src_kind == OCT_SRC_RAW32) {
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; j++) item[j] = bswap32(w[j]);
  } else if (SW == 6) {
    switch (p.srcs[p.oct_src].bits) {   // wave-uniform; <= 21
#define OCT_CASE(B) case B: oct_decode_wide<B>(a, b, ln.ssel, item); break;
      OCT_CASE(1) OCT_CASE(2) OCT_CASE(3) OCT_CASE(4) OCT_CASE(5) OCT_CASE(6) OCT_CASE(7) OCT_CASE(8) OCT_CASE(9) OCT_CASE(10) OCT_CASE(11)
      OCT_CASE(12) OCT_CASE(13) OCT_CASE(14) OCT_CASE(15) OCT_CASE(16) OCT_CASE(17) OCT_CASE(18) OCT_CASE(19) OCT_CASE(20)
#undef OCT_CASE
      default: oct_decode_wide<21>(a, b, ln.ssel, item); break;
    }
  } else {
    oct_decode_source(p.srcs[p.oct_src].bits, a, b, ln.ssel, item);
  }
}

template <bool MASKED, int NW, int SW>
__device__ __forceinline__ void oct_body_lds(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  int64_t* table = reinterpret_cast<int64_t*>(smem);
  const uint32_t table_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;
  for (int o = 0; o < p.n_ops; o++) {
    const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
    for (uint32_t i = (uint32_t)t; i < table_slots; i += PG_BLOCK) table[(size_t)o * table_slots + i] = ident;
  }
  uint32_t* aux_words = reinterpret_cast<uint32_t*>(smem);
  if (p.n_aux > 0) {
    aux_words = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + p.aux[0].lds_offset);
    const int64_t n_words = p.oct_dword ? p.aux[0].rep_bytes : p.aux[0].rep_bytes / 4;
    for (int64_t i = t; i < n_words; i += PG_BLOCK) aux_words[i] = 0;
  }
  // MatchAllFilterOperator: no filter pass ran in front — every doc matches (ExecutionStatistics.numDocsScanned)
  if (!MASKED && blockIdx.x == 0 && t == 0) atomicAdd(p.stats, (unsigned long long)p.num_docs);
  __syncthreads();
  OctLane ln;
  oct_lane_setup(p, lane, ln);
  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  // sub-tile sequence of this wavefront: u = 0, 1, 2, ... -> wave tile first + (u >> 2) x step, sub-tile u & 3
  const int first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave, step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int n_mine = first < p.n_wtiles ? (p.n_wtiles - first + step - 1) / step : 0;
  const int n_sub = n_mine * OCT_SUBS_PER_WTILE;
  const int last_wt = p.n_wtiles - 1;
  auto wt_of = [&](int u) { const int w = first + (u >> 2) * step; return w < p.n_wtiles ? w : last_wt; };   // clamped: loads stay in bounds
  // Load buffers in rotation; a buffer is re-requested (DEPTH sub-tiles ahead) right AFTER it has been decoded, never before: hipcc waits for
  // every load in flight (s_waitcnt vmcnt(0)) where the decode's wave-uniform switch consumes a buffer, so loads requested just before a
  // decode were waited for on the spot — no overlap at all inside a wavefront (r04_b: 58 % of the wave's cycles waiting).  Requested
  // after the decode they travel during this sub-tile's hash / LDS phase and the next DEPTH - 1 sub-tiles.  Three buffers where their
  // compile-time size (OctRawS) leaves room, else two.
  constexpr int DEPTH = (4 * NW + SW) * 3 <= 48 ? 3 : 2;
  OctRawS<NW, SW> r[DEPTH];
  auto turn = [&](OctRawS<NW, SW>& raw, int u) __attribute__((always_inline)) {
    if (u >= n_sub) return;   // wave-uniform: n_sub is a multiple of 4, the last round of three is partial
    uint32_t key[8], item[8];
    octs_decode<NW, SW>(p, ln, raw, key, item);
    const uint32_t m8 = oct_mask8(p, wt_of(u), u & 3, lane, raw.mw, MASKED);
    octs_issue<MASKED, NW, SW>(p, ln, wt_of(u + DEPTH), (u + DEPTH) & 3, lane, raw);
    oct_finish(p, key, item);
    oct_apply_lds(p, m8, key, item, table, aux_words, rep);
  };
  if (n_sub > 0) {
#pragma unroll
    for (int k = 0; k < DEPTH; k++) octs_issue<MASKED, NW, SW>(p, ln, wt_of(k), k & 3, lane, r[k]);
  }
  for (int u = 0; u < n_sub; u += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; k++) turn(r[k], u + k);
  }
  __syncthreads();
  // flush: the partial table [n_ops][G] (replicas folded) and the states, as pg_generic_query_l leaves them
  {
    const int R = p.replicas;
    const int64_t n_out = (int64_t)p.n_ops * p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * n_out;
    for (int64_t i = t; i < n_out; i += PG_BLOCK) {
      const int64_t* src = table + i * R;
      int64_t acc = src[0];
      for (int r = 1; r < R; r++) acc += src[r];   // COUNT only
      out[i] = acc;
    }
  }
  if (p.n_aux > 0) {
    uint32_t* dst = p.aux[0].base + (int64_t)blockIdx.x * (p.aux[0].rep_bytes / 4);
    if (p.oct_dword) {   // dword registers -> the byte registers everything downstream reads
      for (int64_t i = t; i < p.aux[0].rep_bytes / 4; i += PG_BLOCK) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(aux_words + 4 * i);
        dst[i] = r.x | (r.y << 8) | (r.z << 16) | (r.w << 24);
      }
    } else {
      for (int64_t i = t; i < p.aux[0].rep_bytes / 4; i += PG_BLOCK) dst[i] = aux_words[i];
    }
  }
}
template <bool MASKED>
__device__ __forceinline__ void oct_lds_dispatch(const PgQueryPlan& p) {
  bool narrow = true;   // wave-uniform
  for (int g = 0; g < p.n_group_cols; g++) narrow = narrow && p.gcols[g].bits <= 4;
  const bool short_source = p.oct_src_kind == OCT_SRC_NONE || (p.oct_src_kind != OCT_SRC_RAW32 && p.srcs[p.oct_src].bits <= 21);
  if (narrow && short_source) oct_body_lds<MASKED, 2, 6>(p);
  else if (narrow) oct_body_lds<MASKED, 2, 8>(p);
  else if (short_source) oct_body_lds<MASKED, 3, 6>(p);
  else oct_body_lds<MASKED, 3, 8>(p);
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_oct_l(const PgQueryPlan p) { oct_lds_dispatch<false>(p); }
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_oct_lm(const PgQueryPlan p) { oct_lds_dispatch<true>(p); }

// ---- back end C: COUNT(*) alone, no filter in front (round 5) ----------------------------------------------------------------------------
// SELECT dims, COUNT(*) GROUP BY dims reads 0.4 - 2 bytes per doc: with two load buffers per wavefront (oct_body_lds) a CU has ~30 KB in
// flight and the kernel waits for HBM half of its cycles at 43 % VALU use (profiles/r05_sq_count_only_oct.txt).  Here the group-column
// count is compile-time, a buffer is only the group columns' dwords (no source, no match word), and FOUR buffers rotate: the load of
// sub-tile u + 4 is requested where sub-tile u has been decoded.
// Buffers of NW dwords per column — two when every group column has <= 4 bits (8 values: 4 bytes + 3 of misalignment), three up to 8 bits —
// so that four of them fit beside the decode's registers under the 128 a 16-wavefront workgroup leaves each lane (three for 3-4 wide columns).
template <int NG, int NW> struct OctRawG { uint32_t g[NG][NW]; };
template <int NG, int NW>
DEVFN void octc_issue(const PgQueryPlan& p, const OctLane& ln, int wt, int sub, OctRawG<NG, NW>& raw) {
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const PgGroupCol& gc = p.gcols[g];
    const GAS uint8_t* base = gptr<uint8_t>(gc.data) + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)gc.bits + (size_t)sub * (size_t)(OCT_SUB_DOCS / 8) * (size_t)gc.bits;
    if (NW == 2) {
      const u32x2o v = ldnt((const GAS u32x2o_a4*)(base + ln.goff[g]));
      raw.g[g][0] = v.x; raw.g[g][1] = v.y;
    } else {
      const u32x3 v = ldnt((const GAS u32x3_a4*)(base + ln.goff[g]));
      raw.g[g][0] = v.x; raw.g[g][1] = v.y; raw.g[g][NW - 1] = v.z;
    }
  }
}
template <int NG, int NW>
DEVFN void octc_slots(const PgQueryPlan& p, const OctLane& ln, const OctRawG<NG, NW>& raw, uint32_t rep, uint32_t (&slot)[8]) {
  const uint32_t rshift = (uint32_t)p.replica_shift;
#pragma unroll
  for (int g = 0; g < NG; g++) {
    uint32_t v[8];
    u32x3 w;
    w.x = raw.g[g][0]; w.y = raw.g[g][1]; w.z = raw.g[g][NW - 1];   // (NW == 2: .z is never read, the widths are <= 4)
    if (NW == 2) {
      switch (p.gcols[g].bits) {   // wave-uniform
        case 1: oct_decode_small<1>(w, ln.gsel[g], v); break;
        case 2: oct_decode_small<2>(w, ln.gsel[g], v); break;
        case 3: oct_decode_small<3>(w, ln.gsel[g], v); break;
        default: oct_decode_small<4>(w, ln.gsel[g], v); break;
      }
    } else {
      oct_decode_group(p.gcols[g].bits, w, ln.gsel[g], v);
    }
    const uint32_t mult = (uint32_t)p.gcols[g].mult << rshift;   // the table fits LDS: every product fits 24 bits
#pragma unroll
    for (int j = 0; j < 8; j++) slot[j] = g == 0 ? mad24(v[j], mult, rep) : mad24(v[j], mult, slot[j]);
  }
}
template <int NG, int NW>
__device__ __forceinline__ void oct_body_count(const PgQueryPlan& p) {
  // load buffers in rotation.  More than four did not pay: 6 buffers for 4 columns 0.147 ms against 0.118 over 2 x 10^8 docs, 12 buffers for 1 - 2
  // columns within 5 % of four (the unrolled round outgrows the instruction cache; profiles/r05_count_only_oct.txt)
  constexpr int DEPTH = NG * NW > 8 ? 3 : 4;
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  int64_t* table = reinterpret_cast<int64_t*>(smem);
  const uint32_t table_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;
  for (uint32_t i = (uint32_t)t; i < table_slots; i += PG_BLOCK) table[i] = 0;
  if (blockIdx.x == 0 && t == 0) atomicAdd(p.stats, (unsigned long long)p.num_docs);   // MatchAllFilterOperator: every doc matches
  __syncthreads();
  OctLane ln;
  oct_lane_setup(p, lane, ln);
  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  const int first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave, step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int n_mine = first < p.n_wtiles ? (p.n_wtiles - first + step - 1) / step : 0;
  const int n_sub = n_mine * OCT_SUBS_PER_WTILE;
  const int last_wt = p.n_wtiles - 1;
  auto wt_of = [&](int u) __attribute__((always_inline)) { const int w = first + (u >> 2) * step; return w < p.n_wtiles ? w : last_wt; };   // clamped: loads stay in bounds
  auto turn = [&](OctRawG<NG, NW>& r, int u) __attribute__((always_inline)) {
    if (u >= n_sub) return;   // wave-uniform (three buffers: the last round is partial)
    uint32_t slot[8];
    octc_slots<NG, NW>(p, ln, r, rep, slot);
    octc_issue<NG, NW>(p, ln, wt_of(u + DEPTH), (u + DEPTH) & 3, r);
    const int64_t rem = (int64_t)p.num_docs - ((int64_t)wt_of(u) * PG_WAVE_DOCS + (int64_t)(u & 3) * OCT_SUB_DOCS);
    if (rem >= OCT_SUB_DOCS) {   // wave-uniform: a whole sub-tile, no lane masks
#pragma unroll
      for (int j = 0; j < 8; j++) atomicAdd(reinterpret_cast<uint32_t*>(table + slot[j]), 1u);   // low dword: a workgroup sees < 2^31 docs
    } else {
      const uint32_t m8 = oct_mask8(p, wt_of(u), u & 3, lane, 0u, false);
#pragma unroll
      for (int j = 0; j < 8; j++)
        if ((m8 >> j) & 1u) atomicAdd(reinterpret_cast<uint32_t*>(table + slot[j]), 1u);
    }
  };
  OctRawG<NG, NW> r[DEPTH];
  if (n_sub > 0) {
#pragma unroll
    for (int k = 0; k < DEPTH; k++) octc_issue<NG, NW>(p, ln, wt_of(k), k & 3, r[k]);
  }
  for (int u = 0; u < n_sub; u += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; k++) turn(r[k], u + k);
  }
  __syncthreads();
  {   // flush: the partial table [G] (replicas folded), as pg_oct_l leaves it
    const int R = p.replicas;
    int64_t* out = p.partials + (int64_t)blockIdx.x * (int64_t)p.n_groups;
    if (R >= 64) {   // a wavefront per group (16 groups x 1 024 replicas: one lane per group walked them for 30+ us of a 45 us kernel)
      for (int64_t i = wave; i < p.n_groups; i += PG_BLOCK / 64) {
        const int64_t* src = table + i * R;
        int64_t acc = 0;
        for (int r = lane; r < R; r += 64) acc += src[r];
        acc = wave_fold_i64(acc, PG_ACC_SUM);
        if (lane == 0) out[i] = acc;
      }
    } else {
      for (int64_t i = t; i < p.n_groups; i += PG_BLOCK) {
        const int64_t* src = table + i * R;
        int64_t acc = src[0];
        for (int r = 1; r < R; r++) acc += src[r];
        out[i] = acc;
      }
    }
  }
}
template <int NW>
__device__ __forceinline__ void oct_count_dispatch(const PgQueryPlan& p) {
  switch (p.n_group_cols) {   // wave-uniform
    case 1: oct_body_count<1, NW>(p); break;
    case 2: oct_body_count<2, NW>(p); break;
    case 3: oct_body_count<3, NW>(p); break;
    default: oct_body_count<4, NW>(p); break;
  }
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_oct_c(const PgQueryPlan p) {
  bool narrow = true;
  for (int g = 0; g < p.n_group_cols; g++) narrow = narrow && p.gcols[g].bits <= 4;
  if (narrow) oct_count_dispatch<2>(p);
  else oct_count_dispatch<3>(p);
}

// ---- back end P: pruned offers ---------------------------------------------------------------------------------------------------------
// LDS: counts u32 [G] | floors u8 [G] (padded to dwords) | per wavefront a ring of OCT_RING survivor entries.
// Survivors are appended to the wavefront's ring (ds_write) and leave in whole blocks of OCT_STREAM_BLOCK = 256 entries — one coalesced
// 16-byte store per lane — into a block of the stream claimed with one global atomic; the NEXT block is claimed as soon as one is used,
// so the atomic's return is not waited for where it is issued.  (The first cut stored every survivor with a store of its own, 8 store
// instructions per sub-tile between the pipelined column loads, and claimed blocks on demand: 3 x the time of pg_oct_l over the same docs,
// WAIT_INST 46 % — profiles/r04_b_*.)
#define OCT_RING 1024
struct OctStream {
  uint32_t head, tail;   // entries appended / flushed so far (wave-uniform)
};
// inclusive prefix sum across the wavefront (DPP row operations, no LDS)
DEVFN uint32_t oct_wave_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
  return x;
}
// one block of the ring -> the workgroup's region of the stream.  The region's fill cursor lives in LDS (only this workgroup writes the
// region): a claim is one returning ds_add.  (A cursor in HBM shared by all workgroups serialised the claims at ~60 ns each — the passes
// spent 80 % of their time queueing on it, profiles/r04_c_kernels__cfg5_.txt.)
DEVFN void oct_flush_block(const PgQueryPlan& p, uint32_t* ring, uint32_t* s_cur, uint32_t region_base, OctStream& st, int lane) {
  uint32_t pos = 0;
  if (lane == 0) pos = atomicAdd(s_cur, (uint32_t)OCT_STREAM_BLOCK);
  pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
  if (pos + (uint32_t)OCT_STREAM_BLOCK > (uint32_t)p.oct_region) {   // cannot happen: the host sized the region for "every offer survives"
    if (lane == 0) p.oct_cursor[1] = 1u;
    pos = 0;   // keeps the store in bounds; the host fails the query on the flag
  }
  const u32x4 v = *reinterpret_cast<const u32x4*>(ring + (st.tail & (OCT_RING - 1u)) + 4u * (uint32_t)lane);
  *reinterpret_cast<u32x4*>(p.oct_stream + region_base + pos + 4u * (uint32_t)lane) = v;
  st.tail += OCT_STREAM_BLOCK;
}
DEVFN void oct_apply_pruned(const PgQueryPlan& p, uint32_t m8, const uint32_t (&key)[8], const uint32_t (&item)[8], uint32_t* counts,
                            const volatile uint8_t* floors, uint32_t* ring, uint32_t* s_cur, uint32_t region_base, OctStream& st, int lane) {
  // (the floors requested before the hashes, so that the byte reads travel behind them: 8 more live registers, spills, 2.65 ms against 2.57 for
  // config 5 over 10^9 docs — not kept)
  const uint32_t log2m = (uint32_t)p.oct_log2m, pbits = log2m + 5u;
  uint32_t fl[8];
#pragma unroll
  for (int j = 0; j < 8; j++) fl[j] = floors[key[j]];
#pragma unroll
  for (int j = 0; j < 8; j++)
    if ((m8 >> j) & 1u) atomicAdd(counts + key[j], 1u);
  uint32_t surv = 0;   // bit j: doc j's offer can still raise a register of its group
#pragma unroll
  for (int j = 0; j < 8; j++) surv |= (uint32_t)((item[j] >> log2m) > fl[j]) << j;
  surv &= m8;
  const uint32_t n_mine = (uint32_t)__builtin_popcount(surv);
  const uint32_t incl = oct_wave_scan(n_mine);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (total == 0) return;   // wave-uniform
  uint32_t at = st.head + incl - n_mine;   // this lane's first ring position
#pragma unroll
  for (int j = 0; j < 8; j++)
    if ((surv >> j) & 1u) {
      ring[at & (OCT_RING - 1u)] = (key[j] << pbits) | item[j];
      at++;
    }
  st.head += total;
  while (st.head - st.tail >= (uint32_t)OCT_STREAM_BLOCK) oct_flush_block(p, ring, s_cur, region_base, st, lane);   // wave-uniform; <= 3 blocks
}

template <bool MASKED, int NW, int SW>
__device__ __forceinline__ void oct_body_pruned(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_cur;   // entries of this workgroup's stream region claimed so far
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  uint32_t* counts = reinterpret_cast<uint32_t*>(smem);
  const uint32_t G = (uint32_t)p.n_groups;
  uint32_t* floor_words = counts + G;
  for (uint32_t i = (uint32_t)t; i < G; i += PG_BLOCK) counts[i] = 0;
  for (uint32_t i = (uint32_t)t; i < (G + 3u) / 4u; i += PG_BLOCK) floor_words[i] = gptr<uint32_t>(p.oct_floor)[i];
  if (t == 0) s_cur = 0;
  const int t0 = p.oct_t0, t1 = p.oct_t1;   // wave tiles of this pass
  if (!MASKED && blockIdx.x == 0 && t == 0) {
    const int64_t lo = (int64_t)t0 * PG_WAVE_DOCS, hi = (int64_t)t1 * PG_WAVE_DOCS;
    atomicAdd(p.stats, (unsigned long long)((hi < p.num_docs ? hi : (int64_t)p.num_docs) - lo));
  }
  __syncthreads();
  const volatile uint8_t* floors = reinterpret_cast<const volatile uint8_t*>(floor_words);
  OctLane ln;
  oct_lane_setup(p, lane, ln);
  const int first = t0 + (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave, step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int n_mine = first < t1 ? (t1 - first + step - 1) / step : 0;
  const int n_sub = n_mine * OCT_SUBS_PER_WTILE;
  const int last_wt = t1 - 1;
  auto wt_of = [&](int u) { const int w = first + (u >> 2) * step; return w < t1 ? w : last_wt; };
  uint32_t* ring = counts + ((G + (G + 3u) / 4u + 3u) & ~3u) + (uint32_t)wave * OCT_RING;   // 16-byte aligned: blocks leave with ds_read_b128
  const uint32_t region_base = (uint32_t)blockIdx.x * (uint32_t)p.oct_region;
  OctStream st;
  st.head = 0; st.tail = 0;
  // DEPTH buffers in rotation; a buffer is re-requested (DEPTH sub-tiles ahead) right after it has been decoded (the ordering of requests and
  // decodes: see oct_body_lds)
  constexpr int DEPTH = (4 * NW + SW) * 3 <= 48 ? 3 : 2;   // 8-bit group columns: the buffers are OctRaw's size again, two of them fit
  OctRawS<NW, SW> r[DEPTH];
  auto turn = [&](OctRawS<NW, SW>& raw, int u) __attribute__((always_inline)) {
    if (u >= n_sub) return;   // wave-uniform: n_sub is a multiple of 4, the last round of three is partial
    uint32_t key[8], item[8];
    octs_decode<NW, SW>(p, ln, raw, key, item);
    const uint32_t m8 = oct_mask8(p, wt_of(u), u & 3, lane, raw.mw, MASKED);
    octs_issue<MASKED, NW, SW>(p, ln, wt_of(u + DEPTH), (u + DEPTH) & 3, lane, raw);
    oct_finish(p, key, item);
    oct_apply_pruned(p, m8, key, item, counts, floors, ring, &s_cur, region_base, st, lane);
  };
  if (n_sub > 0) {
#pragma unroll
    for (int k = 0; k < DEPTH; k++) octs_issue<MASKED, NW, SW>(p, ln, wt_of(k), k & 3, lane, r[k]);
  }
  for (int u = 0; u < n_sub; u += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; k++) turn(r[k], u + k);
  }
  // what is left in the ring leaves as one block padded with PG_RADIX_INVALID_KEY
  if (st.head != st.tail) {   // wave-uniform; fewer than a block
    for (uint32_t i = st.head + (uint32_t)lane; i < st.tail + (uint32_t)OCT_STREAM_BLOCK; i += 64u) ring[i & (OCT_RING - 1u)] = PG_RADIX_INVALID_KEY;
    oct_flush_block(p, ring, &s_cur, region_base, st, lane);
  }
  __syncthreads();
  if (t == 0) p.oct_cursor[PG_OCT_CTRL_COUNTS + blockIdx.x] = s_cur;   // the region's fill, a multiple of the block
  // COUNT partials accumulate over the passes: workgroup b of every pass adds into row b (zeroed once per query)
  uint32_t* out = p.oct_counts + (size_t)blockIdx.x * G;
  for (uint32_t i = (uint32_t)t; i < G; i += PG_BLOCK) out[i] += counts[i];
}
template <bool MASKED>
__device__ __forceinline__ void oct_pruned_dispatch(const PgQueryPlan& p) {
  bool narrow = true;   // wave-uniform
  for (int g = 0; g < p.n_group_cols; g++) narrow = narrow && p.gcols[g].bits <= 4;
  const bool short_source = p.oct_src_kind != OCT_SRC_RAW32 && p.srcs[p.oct_src].bits <= 21;
  if (narrow && short_source) oct_body_pruned<MASKED, 2, 6>(p);
  else if (narrow) oct_body_pruned<MASKED, 2, 8>(p);
  else if (short_source) oct_body_pruned<MASKED, 3, 6>(p);
  else oct_body_pruned<MASKED, 3, 8>(p);
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_oct_p(const PgQueryPlan p) { oct_pruned_dispatch<false>(p); }
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_oct_pm(const PgQueryPlan p) { oct_pruned_dispatch<true>(p); }

// After a pass: registers of group g = bytewise max of what the table holds and the slices of g's bucket (the passes accumulate), and
// floor[g] = the smallest of them — the next pass's pruning threshold.  One wavefront per group; registers are bytes, 2^log2m per group.
extern "C" __global__ void __launch_bounds__(256) pg_oct_merge_floor_kernel(const uint32_t* __restrict__ partials, uint32_t* __restrict__ regs,
                                                                             uint8_t* __restrict__ floors, int n_groups, int log2m, int radix_shift,
                                                                             int slices) {
  const int g = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (g >= n_groups) return;
  const int n_words = 1 << (log2m - 2);
  const int64_t slots = (int64_t)1 << radix_shift;
  const int64_t b = g >> radix_shift, local = g & (slots - 1);
  uint32_t mn = 0xFFu;
  for (int w = lane; w < n_words; w += 64) {
    uint32_t acc = regs[(size_t)g * n_words + w];
    for (int sl = 0; sl < slices; sl++) acc = bytemax4(acc, partials[((b * slices + sl) * slots + local) * n_words + w]);
    regs[(size_t)g * n_words + w] = acc;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t y = (acc >> (8 * k)) & 0xFFu; mn = y < mn ? y : mn; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mn, off, 64); mn = o < mn ? o : mn; }
  if (lane == 0) floors[g] = (uint8_t)mn;
}
// After pg_oct_p: the regions' tiles numbered across the regions (exclusive prefix of ceil(fill / 2 048)); one small block.
extern "C" __global__ void __launch_bounds__(256) pg_oct_stream_index_kernel(uint32_t* __restrict__ ctrl, int n_regions) {
  __shared__ uint32_t s_tiles[PG_OCT_MAX_REGIONS];
  const int t = threadIdx.x;
  for (int i = t; i < n_regions; i += 256) s_tiles[i] = (ctrl[PG_OCT_CTRL_COUNTS + i] + PG_WAVE_DOCS - 1) / PG_WAVE_DOCS;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int i = 0; i < n_regions; i++) { ctrl[PG_OCT_CTRL_TILE_START + i] = run; run += s_tiles[i]; }
    ctrl[PG_OCT_CTRL_TILE_START + n_regions] = run;
    ctrl[0] = run;   // tiles of the stream (diagnostics)
  }
}
// Before a pass: the chunk records unowned, the chunk and stream cursors and the chunk index's counters zero; the error flags
// (p2_ctrl[1], cursor[1]) stay as they are — they are read once, after the last pass.
extern "C" __global__ void __launch_bounds__(256) pg_oct_pass_reset_kernel(uint32_t* __restrict__ p2_meta, int64_t n_meta, uint32_t* __restrict__ p2_ctrl,
                                                                            uint32_t* __restrict__ cursor) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = i0; i < n_meta; i += step) p2_meta[i] = 0xFFFFFFFFu;
  if (i0 == 0) p2_ctrl[0] = 0;
  (void)cursor;
  if (i0 >= PG_P2_CTRL_COUNTS && i0 < PG_P2_CTRL_DWORDS) p2_ctrl[i0] = 0;
}
// COUNT row of the final table = sum of the workgroups' counters.  64 consecutive groups per block (coalesced rows), the partial rows
// split over the block's four wavefronts, folded through LDS.
extern "C" __global__ void __launch_bounds__(256) pg_oct_reduce_counts_kernel(const uint32_t* __restrict__ counts, int64_t* __restrict__ out, int n_parts,
                                                                               int n_groups) {
  __shared__ unsigned long long s_acc[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = (int)blockIdx.x * 64 + lane;
  unsigned long long acc = 0;
  if (g < n_groups)
    for (int k = wv; k < n_parts; k += 4) acc += (unsigned long long)counts[(size_t)k * n_groups + g];
  s_acc[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && g < n_groups) out[g] = (int64_t)(s_acc[0][lane] + s_acc[1][lane] + s_acc[2][lane] + s_acc[3][lane]);
}
