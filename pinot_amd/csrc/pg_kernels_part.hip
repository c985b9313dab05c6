// Partition pipeline v2 — group-by over key spaces beyond one LDS table (PG_AGG_RADIX with PgQueryPlan::p2), a translation unit
// of its own (256-thread scatter workgroups; the kernels of pg_kernels.hip become uninstantiated templates here).
//
// Reference work: DictionaryBasedGroupKeyGenerator's map-based holders (core/query/aggregation/groupby/
// DictionaryBasedGroupKeyGenerator.java:416-446,629-668,993-1084) + the aggregateGroupBySV loops of the aggregation functions
// (SumAggregationFunction.java:160-179, CountAggregationFunction.java:110-143, Min/MaxAggregationFunction.java:163-188,
// DistinctCountHLLAggregationFunction.java:152-222): raw key = Σ dictId_j · Π card_i, one holder update per matching doc.
//
// Design (what round 2's radix passes did in three reads of the keys and 16-byte tuples is one read and 4-byte tuples here):
//   * ONE pass over the segment.  No counting pass: output space is handed out in chunks of PG_P2_CHUNK tuples from a global
//     counter (claimed PG_P2_BATCH at a time into a ring in LDS), every (workgroup, bucket) stream is a list of chunks recorded in
//     `p2_meta` (owner bucket | filled lines); the aggregation pass finds a bucket's chunks by scanning that small array.
//   * A tuple is `planes` dwords, bit-packed by the planner: the key's low radix_shift bits, then per source the
//     (register index, rank) a DISTINCTCOUNTHLL offers, a dictId, a raw INT minus the column's minimum (as many bits as the column's
//     range needs) or a whole 64-bit value.  The planner lowers radix_shift (more buckets) when that makes a tuple fit one dword.
//     Planes are stored as structure of arrays, so every store and every load is a dword-coalesced stream whatever the tuple width.
//   * The scatter workgroup (4 wavefronts, 4 workgroups per CU) sorts a round of 1 024·Q docs by bucket in LDS — a returning
//     ds_add per doc yields its rank, one wavefront turns the histogram into offsets, the tuples are written to their sorted
//     positions — and then copies every bucket's run out as WHOLE 128-byte lines (32 tuples); what does not fill a line waits in a
//     per-bucket leftover line in LDS for the next round.  Whole aligned lines stream at ~5.5 TB/s with no fetch
//     (profiles/r02_scatter_write_probe.txt); partial ones do not.
//   * The aggregation pass keeps a bucket's accumulators in LDS ([n_ops][2^radix_shift] int64) and HyperLogLog registers as one
//     DWORD each, so that an offer is a single non-returning ds_max_u32 (round 2 kept bytes and paid a read + compare-and-swap chain
//     per tuple); partials are flushed as bytes and merged by pg_radix_reduce_aux_kernel / pg_radix_reduce_kernel as before.
#ifndef PG_WAVES_PER_BLOCK
#define PG_WAVES_PER_BLOCK 4
#endif
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"
#include "pg_oct_layout.h"

#define P2_THREADS (PG_P2_WAVES * 64)
#ifndef PG_P2_STREAM_MIN_QUARTETS
#define PG_P2_STREAM_MIN_QUARTETS 2
#endif

// ---- one source's field for the 4 docs of Q quads ----------------------------------------------------------------------------
template <int Q, bool WIDE>
DEVFN void p2_field(const PgQueryPlan& p, int si, const uint32_t (&qi)[Q], int wt, uint32_t (&f)[Q][4], uint32_t (&fh)[Q][4]) {
  const PgValueSrc& V = p.srcs[si];
  const int kind = p.p2_fkind[si];
  if (V.col_kind == PG_COL_FIXED_BIT) {
    const GAS uint32_t* tw = packed_wtile_base(V.data, wt, V.bits);
    const uint32_t bits = (uint32_t)V.bits, mask = (1u << V.bits) - 1u;
    if (bits <= 8) {
      uint32_t r[Q][2];
#pragma unroll
      for (int u = 0; u < Q; u++) load_packed_quad<true>(tw, qi[u], bits, r[u]);
#pragma unroll
      for (int u = 0; u < Q; u++) decode_packed_quad<true>(r[u], qi[u], bits, mask, f[u]);
    } else {
#pragma unroll
      for (int u = 0; u < Q; u++) {
        uint32_t r[8];
        load_packed_quad<false>(tw, qi[u], bits, r);
        decode_packed_quad<false>(r, qi[u], bits, mask, f[u]);
      }
    }
    if (kind == PG_P2_F_HLL) {
      const int log2m = p.pk_hll[si];
      if (p.pk_affine[si]) {   // arithmetic dictionary: the value is computed and hashed, no cardinality-sized gather
#pragma unroll
        for (int u = 0; u < Q; u++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            f[u][i] = packed_hll_payload(hll_index_rank_dev(murmur_hash_long_dev(p.pk_base[si] + p.pk_step[si] * (int64_t)f[u][i]), log2m), log2m);
      } else {
#pragma unroll
        for (int u = 0; u < Q; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) f[u][i] = packed_hll_payload(gptr<uint32_t>(p.pk_lut[si])[f[u][i]], log2m);
      }
    }
    return;
  }
  if (V.col_kind == PG_COL_RAW32) {
    const GAS uint8_t* tb = gptr<uint8_t>(V.data + (size_t)wt * (PG_WAVE_DOCS * 4));
    u32x4 v[Q];
#pragma unroll
    for (int u = 0; u < Q; u++) v[u] = ldnt((const GAS u32x4*)(tb + qi[u] * 16u));
#pragma unroll
    for (int u = 0; u < Q; u++) { f[u][0] = bswap32(v[u].x); f[u][1] = bswap32(v[u].y); f[u][2] = bswap32(v[u].z); f[u][3] = bswap32(v[u].w); }
    if (kind == PG_P2_F_HLL) {   // hll.offer(value): Integer → hashLong((long) v), Float → raw int bits (MurmurHash.hash(Object))
      const int log2m = p.pk_hll[si];
#pragma unroll
      for (int u = 0; u < Q; u++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          f[u][i] = packed_hll_payload(hll_index_rank_dev(murmur_hash_long_dev((int64_t)(int32_t)f[u][i]), log2m), log2m);
    } else if (V.val_type == PG_V_I32) {
      const uint32_t bias = (uint32_t)p.p2_fbias[si];
#pragma unroll
      for (int u = 0; u < Q; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) f[u][i] -= bias;
    }
    return;
  }
  // PG_COL_RAW64
  if (!WIDE && kind != PG_P2_F_HLL) return;   // a one-plane tuple carries no whole 64-bit value
  const GAS uint8_t* tb = gptr<uint8_t>(V.data + (size_t)wt * (PG_WAVE_DOCS * 8));
#pragma unroll
  for (int u = 0; u < Q; u++) {
    const GAS u32x4* pp = (const GAS u32x4*)(tb + qi[u] * 32u);
    const u32x4 a = ldnt(pp), b = ldnt(pp + 1);
    fh[u][0] = bswap32(a.x); f[u][0] = bswap32(a.y); fh[u][1] = bswap32(a.z); f[u][1] = bswap32(a.w);
    fh[u][2] = bswap32(b.x); f[u][2] = bswap32(b.y); fh[u][3] = bswap32(b.z); f[u][3] = bswap32(b.w);
  }
  if (kind == PG_P2_F_HLL) {
    const int log2m = p.pk_hll[si];
#pragma unroll
    for (int u = 0; u < Q; u++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        f[u][i] = packed_hll_payload(hll_index_rank_dev(murmur_hash_long_dev((int64_t)(((uint64_t)fh[u][i] << 32) | f[u][i])), log2m), log2m);
  }
}

// ---- batched loader of the scatter's phase A ("fast A") ---------------------------------------------------------------------------
// The generic key / field functions above walk the plan's columns in a runtime loop: load, wait, decode, next column — one HBM round
// trip per column and batch, ~10 per round (rocprof, round 3: 1.28 ms per 200 M docs, latency-bound).  Plans of the common shape
// (PgQueryPlan::p2_fast_a: <= 4 group columns, the first <= 24 bits, the others <= 8 bits; at most one source, bit-packed <= 24 bits
// or raw 32-bit) instead issue EVERY load of a batch of two quads back to back, and the loads of a round's first batch are requested
// before the previous round's bookkeeping phases, so they travel while the workgroup sorts and writes.
// inclusive prefix sum across the wavefront with DPP row operations (no LDS round trips: six dependent ds_bpermute per __shfl_up scan
// were ~600 cycles of the serial bookkeeping phase)
DEVFN uint32_t p2_wave_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1 (zero beyond the row)
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8: inclusive within each row of 16
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  return x;
}
// stream-lib MurmurHash.hashLong((long) v) for an INT value: the high word is 0 or -1, its contribution a constant
DEVFN uint32_t p2_murmur_int(int32_t v) {
  const uint32_t m = 0x5bd1e995u;
  uint32_t k = (uint32_t)v * m;
  k ^= k >> 24;
  uint32_t h = k * m;
  constexpr uint32_t km = 0u - 0x5bd1e995u;                 // 0xFFFFFFFF * m
  constexpr uint32_t hi_neg = (km ^ (km >> 24)) * 0x5bd1e995u;
  h *= m;
  h ^= v < 0 ? hi_neg : 0u;
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}

typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
#define P2_QA 2
// Load targets are vector VALUES (never arrays written through a pointer: round 3's first cut passed uint32_t* into the loaders and the
// compiler kept the windows in scratch, waiting for every load right after issuing it).
struct P2Raw {
  u32x4 g0[P2_QA];      // group column 0: a 64-bit window in .xy (<= 8 bits) or a 128-bit window (9..24 bits) per quad
  u32x2 g[3][P2_QA];    // group columns 1..3: 64-bit windows
  u32x4 s0[P2_QA];      // the source: 64- / 128-bit window of a bit-packed column, or the four raw 32-bit values
};
DEVFN u32x2 p2_ld2(const GAS uint32_t* __restrict__ tw, uint32_t q, uint32_t bits) {   // <= 8 bits: the quad lies in one 64-bit window
  return ldnt((const GAS u32x2_a4*)(tw + (__umul24(4u * q, bits) >> 5)));
}
DEVFN u32x4 p2_ld4(const GAS uint32_t* __restrict__ tw, uint32_t q, uint32_t bits) {   // 9..24 bits: four values (<= 96 bits) start within the first dword
  return ldnt((const GAS u32x4_a4*)(tw + (__umul24(4u * q, bits) >> 5)));   // q < 512, bits <= 32
}
DEVFN void p2_dec2(u32x2 r, uint32_t q, uint32_t bits, uint32_t mask, uint32_t (&out)[4]) {
  const uint32_t sh = __umul24(4u * q, bits) & 31u;
  const uint64_t win = ((uint64_t)bswap32(r.x) << 32) | (uint64_t)bswap32(r.y);
  const uint32_t top = (uint32_t)((win << sh) >> 32);   // the quad's four values now start at bit 31
  out[0] = top >> (32u - bits);
#pragma unroll
  for (int i = 1; i < 4; i++) out[i] = (top >> (32u - (uint32_t)(i + 1) * bits)) & mask;
}
DEVFN void p2_dec4(u32x4 r, uint32_t q, uint32_t bits, uint32_t mask, uint32_t (&out)[4]) {
  const uint32_t sh0 = __umul24(4u * q, bits) & 31u;
  const uint32_t w0 = bswap32(r.x), w1 = bswap32(r.y), w2 = bswap32(r.z), w3 = bswap32(r.w);
  // the window shifted left by sh0 (funnel shifts): value i now starts at the wave-uniform bit i * bits
  const uint32_t n0 = (uint32_t)((((uint64_t)w0 << 32) | w1) >> (32u - sh0)), n1 = (uint32_t)((((uint64_t)w1 << 32) | w2) >> (32u - sh0));
  const uint32_t n2 = (uint32_t)((((uint64_t)w2 << 32) | w3) >> (32u - sh0)), n3 = w3 << sh0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t st = (uint32_t)i * bits, j = st >> 5, o = st & 31u;   // wave-uniform
    const uint32_t a = j == 0 ? n0 : (j == 1 ? n1 : n2), b = j == 0 ? n1 : (j == 1 ? n2 : n3);
    const uint64_t win = ((uint64_t)a << 32) | (uint64_t)b;
    out[i] = (uint32_t)(win >> (64u - o - bits)) & mask;
  }
}
// Every load target is written by ONE load instruction on every path (a value assembled from a load's components, or loaded in one
// of two branches, makes the compiler wait for the load where the paths join): column 0 and the source always take a 128-bit window
// (columns are padded, the over-read of a <= 8-bit column stays inside), the source's address is selected, not branched on.
DEVFN void p2_issue(const PgQueryPlan& p, const uint32_t (&qi)[P2_QA], int wt, P2Raw& raw) {
  {
    const PgGroupCol& gc = p.gcols[0];
    const GAS uint32_t* tw = packed_wtile_base(gc.data, wt, gc.bits);
#pragma unroll
    for (int u = 0; u < P2_QA; u++) raw.g0[u] = p2_ld4(tw, qi[u], (uint32_t)gc.bits);
  }
#pragma unroll
  for (int g = 1; g < 4; g++)
    if (g < p.n_group_cols) {
      const PgGroupCol& gc = p.gcols[g];
      const GAS uint32_t* tw = packed_wtile_base(gc.data, wt, gc.bits);
#pragma unroll
      for (int u = 0; u < P2_QA; u++) raw.g[g - 1][u] = p2_ld2(tw, qi[u], (uint32_t)gc.bits);
    }
  if (p.n_srcs > 0) {
    const PgValueSrc& V = p.srcs[0];
    const bool packed = V.col_kind == PG_COL_FIXED_BIT;   // else PG_COL_RAW32
    const uint32_t bits = packed ? (uint32_t)V.bits : 32u;   // a raw 32-bit column is a 32-bit packed one: window = the quad's 16 bytes
    const GAS uint32_t* tw = gptr<uint32_t>(V.data + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)bits);
#pragma unroll
    for (int u = 0; u < P2_QA; u++) raw.s0[u] = p2_ld4(tw, qi[u], bits);
  }
}
// keys and plane dwords of the batch's 8 docs per lane
// SRC: 0 no source; 1 a bit-packed dictionary column offered to a HyperLogLog through an INT arithmetic dictionary (config 5: no
// look-up, four multiplies per hash); 2 any other source the loader takes
template <int T, int SRC>
DEVFN void p2_decode(const PgQueryPlan& p, const P2Raw& raw, const uint32_t (&qi)[P2_QA], uint32_t local_mask, uint32_t (&key)[P2_QA * 4],
                     uint32_t (&d)[T][P2_QA * 4]) {
  {
    const PgGroupCol& gc = p.gcols[0];
    const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
#pragma unroll
    for (int u = 0; u < P2_QA; u++) {
      uint32_t v[4];
      if (bits <= 8) p2_dec2((u32x2){raw.g0[u].x, raw.g0[u].y}, qi[u], bits, mask, v);
      else p2_dec4(raw.g0[u], qi[u], bits, mask, v);
#pragma unroll
      for (int i = 0; i < 4; i++) key[4 * u + i] = v[i];   // mult of column 0 is 1
    }
  }
#pragma unroll
  for (int g = 1; g < 4; g++)
    if (g < p.n_group_cols) {
      const PgGroupCol& gc = p.gcols[g];
      const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u, mult = (uint32_t)gc.mult;
#pragma unroll
      for (int u = 0; u < P2_QA; u++) {
        uint32_t v[4];
        p2_dec2(raw.g[g - 1][u], qi[u], bits, mask, v);
#pragma unroll
        for (int i = 0; i < 4; i++) key[4 * u + i] += __umul24(v[i], mult);   // mult < 2^24 (fast-A condition)
      }
    }
#pragma unroll
  for (int j = 0; j < P2_QA * 4; j++) {
    d[0][j] = key[j] & local_mask;
#pragma unroll
    for (int pl = 1; pl < T; pl++) d[pl][j] = 0;
  }
  if (SRC == 1) {
    const PgValueSrc& V = p.srcs[0];
    const uint32_t bits = (uint32_t)V.bits, mask = (1u << V.bits) - 1u;
    const int log2m = p.pk_hll[0];
    const uint32_t b32 = (uint32_t)p.pk_base[0], s32 = (uint32_t)p.pk_step[0], sh = (uint32_t)p.pk_shift[0];
#pragma unroll
    for (int u = 0; u < P2_QA; u++) {
      uint32_t v[4];
      if (bits <= 8) p2_dec2((u32x2){raw.s0[u].x, raw.s0[u].y}, qi[u], bits, mask, v);
      else p2_dec4(raw.s0[u], qi[u], bits, mask, v);
#pragma unroll
      for (int i = 0; i < 4; i++)
        d[0][4 * u + i] |= packed_hll_payload(hll_index_rank_dev(p2_murmur_int((int32_t)(b32 + __umul24(s32, v[i]))), log2m), log2m) << sh;
    }
  } else if (SRC == 2 && p.n_srcs > 0) {
    const PgValueSrc& V = p.srcs[0];
    const int kind = p.p2_fkind[0];
    uint32_t f[P2_QA * 4];
    if (V.col_kind == PG_COL_FIXED_BIT) {
      const uint32_t bits = (uint32_t)V.bits, mask = (1u << V.bits) - 1u;
#pragma unroll
      for (int u = 0; u < P2_QA; u++) {
        uint32_t v[4];
        if (bits <= 8) p2_dec2((u32x2){raw.s0[u].x, raw.s0[u].y}, qi[u], bits, mask, v);
        else p2_dec4(raw.s0[u], qi[u], bits, mask, v);
#pragma unroll
        for (int i = 0; i < 4; i++) f[4 * u + i] = v[i];
      }
      if (kind == PG_P2_F_HLL) {
        const int log2m = p.pk_hll[0];
        if (p.pk_affine[0] == 2) {   // INT dictionary, value = base + step x dictId with a 24-bit step: 32-bit arithmetic, 4 multiplies
          const uint32_t b32 = (uint32_t)p.pk_base[0], s32 = (uint32_t)p.pk_step[0];
#pragma unroll
          for (int j = 0; j < P2_QA * 4; j++)
            f[j] = packed_hll_payload(hll_index_rank_dev(p2_murmur_int((int32_t)(b32 + __umul24(s32, f[j]))), log2m), log2m);
        } else if (p.pk_affine[0]) {
#pragma unroll
          for (int j = 0; j < P2_QA * 4; j++)
            f[j] = packed_hll_payload(hll_index_rank_dev(murmur_hash_long_dev(p.pk_base[0] + p.pk_step[0] * (int64_t)f[j]), log2m), log2m);
        } else {   // gathers first (all in flight), payloads after
          uint32_t ir[P2_QA * 4];
#pragma unroll
          for (int j = 0; j < P2_QA * 4; j++) ir[j] = gptr<uint32_t>(p.pk_lut[0])[f[j]];
#pragma unroll
          for (int j = 0; j < P2_QA * 4; j++) f[j] = packed_hll_payload(ir[j], log2m);
        }
      }
    } else {   // raw 32-bit values
#pragma unroll
      for (int u = 0; u < P2_QA; u++) {
        f[4 * u] = bswap32(raw.s0[u].x); f[4 * u + 1] = bswap32(raw.s0[u].y); f[4 * u + 2] = bswap32(raw.s0[u].z); f[4 * u + 3] = bswap32(raw.s0[u].w);
      }
      if (kind == PG_P2_F_HLL) {
        const int log2m = p.pk_hll[0];
#pragma unroll
        for (int j = 0; j < P2_QA * 4; j++)
          f[j] = packed_hll_payload(hll_index_rank_dev(p2_murmur_int((int32_t)f[j]), log2m), log2m);
      } else if (V.val_type == PG_V_I32) {
        const uint32_t bias = (uint32_t)p.p2_fbias[0];
#pragma unroll
        for (int j = 0; j < P2_QA * 4; j++) f[j] -= bias;
      }
    }
    const int fpl = p.p2_fplane[0];
    const uint32_t sh = (uint32_t)p.pk_shift[0];
#pragma unroll
    for (int pl = 0; pl < T; pl++)
      if (pl == fpl) {
#pragma unroll
        for (int j = 0; j < P2_QA * 4; j++) d[pl][j] |= f[j] << sh;
      }
  }
}
// SRC == 3: the docs are the entries of the survivor stream pg_oct_p leaves (pg_kernels_oct.hip): one dword per entry,
// key << payload bits | payload, PG_RADIX_INVALID_KEY = padding.  A "tile" is 2 048 consecutive entries, a quad four of them.
// The stream is one REGION per pg_oct_p workgroup (oct_region entries each, filled from its start; the fill counts sit in the control
// block): no global cursor — 27 K claims on one address cost 1.6 ms, profiles/r04_c_kernels__cfg5_.txt.  The regions' 2 048-entry tiles are
// numbered through a prefix array (pg_oct_stream_index_kernel): virtual tile v -> region w = the last with tile_start[w] <= v.
template <bool ON> struct P2StreamLds { };
template <> struct P2StreamLds<true> { uint32_t tile_start[PG_OCT_MAX_REGIONS + 4]; };
struct P2StreamTile { uint32_t base; int32_t n_valid; };
DEVFN P2StreamTile p2_stream_tile(const PgQueryPlan& p, const uint32_t* tile_start, int v, int n_vtiles) {
  P2StreamTile r;
  r.base = 0; r.n_valid = 0;
  if (v >= n_vtiles) return r;   // wave-uniform
  int lo = 0, hi = p.oct_n_regions;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)tile_start[mid] <= v) lo = mid; else hi = mid;
  }
  const uint32_t tile = (uint32_t)v - tile_start[lo];
  const uint32_t count = gptr<uint32_t>(p.oct_cursor)[PG_OCT_CTRL_COUNTS + lo];
  const uint32_t left = count - tile * PG_WAVE_DOCS;
  r.base = (uint32_t)lo * (uint32_t)p.oct_region + tile * PG_WAVE_DOCS;
  r.n_valid = left >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)left;
  return r;
}
DEVFN void p2_issue_stream(const PgQueryPlan& p, const uint32_t (&qi)[P2_QA], uint32_t base, P2Raw& raw) {
  const GAS uint32_t* tw = gptr<uint32_t>(p.oct_stream) + base;
#pragma unroll
  for (int u = 0; u < P2_QA; u++) raw.s0[u] = ldnt((const GAS u32x4*)(tw + 4u * qi[u]));
}
// returns the batch's validity bits (entry != padding)
DEVFN uint32_t p2_decode_stream(const PgQueryPlan& p, const P2Raw& raw, uint32_t local_mask, uint32_t (&key)[P2_QA * 4], uint32_t (&d)[1][P2_QA * 4]) {
  const uint32_t pbits = (uint32_t)p.pk_bits[0], pmask = (1u << pbits) - 1u, sh = (uint32_t)p.pk_shift[0];
  uint32_t valid = 0;
#pragma unroll
  for (int u = 0; u < P2_QA; u++) {
    const uint32_t t4[4] = {raw.s0[u].x, raw.s0[u].y, raw.s0[u].z, raw.s0[u].w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t t = t4[i];
      valid |= (uint32_t)(t != PG_RADIX_INVALID_KEY) << (4 * u + i);
      const uint32_t k = t == PG_RADIX_INVALID_KEY ? 0u : (t >> pbits);
      key[4 * u + i] = k;
      d[0][4 * u + i] = (k & local_mask) | ((t & pmask) << sh);
    }
  }
  return valid;
}
// generic (any plan the pipeline takes): the runtime-loop key / field functions
template <int T>
DEVFN void p2_decode_generic(const PgQueryPlan& p, const uint32_t (&qi)[P2_QA], int wt, uint32_t local_mask, uint32_t (&key)[P2_QA * 4],
                             uint32_t (&d)[T][P2_QA * 4]) {
  uint32_t k2[P2_QA][4];
  radix_keys_of<P2_QA>(p, qi, wt, k2);
#pragma unroll
  for (int u = 0; u < P2_QA; u++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      key[4 * u + i] = k2[u][i];
      d[0][4 * u + i] = k2[u][i] & local_mask;
#pragma unroll
      for (int pl = 1; pl < T; pl++) d[pl][4 * u + i] = 0;
    }
  for (int si = 0; si < p.n_srcs; si++) {
    uint32_t f[P2_QA][4], fh[P2_QA][4];
    p2_field<P2_QA, (T > 1)>(p, si, qi, wt, f, fh);
    const int fpl = p.p2_fplane[si];
    const uint32_t sh = (uint32_t)p.pk_shift[si];
    const bool wide = T > 1 && p.p2_fkind[si] == PG_P2_F_RAW64;
#pragma unroll
    for (int pl = 0; pl < T; pl++) {
      if (pl == fpl) {
#pragma unroll
        for (int u = 0; u < P2_QA; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) d[pl][4 * u + i] |= f[u][i] << sh;
      }
      if (wide && pl == fpl + 1) {
#pragma unroll
        for (int u = 0; u < P2_QA; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) d[pl][4 * u + i] = fh[u][i];
      }
    }
  }
}

// ---- oct-layout phase A (round 5) ---------------------------------------------------------------------------------------------------
// The quad-layout loaders above cut run-time-width fields out of 64- / 128-bit windows: 72 VALU per doc for the whole scatter, 15 of them
// v_readlane / v_writelane of 111 spilled SGPRs (profiles/r04_ab_partition_aggregate_simple.txt).  Plans WITHOUT a filter pass in front
// (PgQueryPlan::p2_oct_a: no match words; group columns <= 8 bits, the first <= 24; no source, a raw INT or a <= 24-bit dictId field; one
// plane) read their columns in the oct layout instead (pg_oct_layout.h): lane L owns docs 8L .. 8L+7 of a 512-doc sub-tile, a round is two
// sub-tiles per wavefront (16 docs per lane, as Q = 4 quads), field positions are compile-time after one byte permute per dword.
// OSK: 0 no source, 1 raw 32-bit INT (minus the column's minimum), 2 bit-packed dictIds (<= 24 bits).
struct P2OctRaw {
  u32x4 g0a, g0b;   // G0WIDE: column 0, 8 dwords from the lane's window start
  u32x3 g[4];       // <= 8-bit group columns: 3 dwords
  u32x4 s0, s1;     // the source: 8 dwords from its window start, or the lane's 8 raw values
};
struct P2OctLane { uint32_t goff[4], gsel[4], soff, ssel; };
template <int OSK>
DEVFN void p2o_lane_setup(const PgQueryPlan& p, int lane, P2OctLane& ln) {
#pragma unroll
  for (int g = 0; g < 4; g++) {
    ln.goff[g] = 0;
    ln.gsel[g] = oct_selector(0);
    if (g < p.n_group_cols) {
      const uint32_t bo = (uint32_t)lane * (uint32_t)p.gcols[g].bits;
      ln.goff[g] = bo & ~3u;
      ln.gsel[g] = oct_selector(bo & 3u);
    }
  }
  const uint32_t sbits = OSK == 1 ? 32u : (OSK == 2 ? (uint32_t)p.srcs[0].bits : 0u);
  const uint32_t bo = (uint32_t)lane * sbits;
  ln.soff = bo & ~3u;
  ln.ssel = oct_selector(bo & 3u);
}
// A wave-uniform address pinned into SGPRs: the loads take the  global_load v, v_offset, s[base:base+1]  form; otherwise the compiler hoists
// one 64-bit per-lane address per column out of the round loop (2 VGPRs each) and the kernel spills.
DEVFN const GAS uint8_t* p2o_sgpr(const GAS uint8_t* ptr) {
  const uint64_t v = (uint64_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const GAS uint8_t*)(((uint64_t)hi << 32) | (uint64_t)lo);
}
template <bool G0WIDE, int OSK>
DEVFN void p2o_issue(const PgQueryPlan& p, const P2OctLane& ln, int wt, int sub, P2OctRaw& raw) {
  const size_t at = (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) + (size_t)sub * (size_t)(OCT_SUB_DOCS / 8);   // x bits = the sub-tile's byte offset
  {
    const PgGroupCol& gc = p.gcols[0];
    const GAS uint8_t* base = p2o_sgpr(gptr<uint8_t>(gc.data) + at * (size_t)gc.bits);
    if (G0WIDE) { raw.g0a = ldnt((const GAS u32x4_a4*)(base + ln.goff[0])); raw.g0b = ldnt((const GAS u32x4_a4*)(base + ln.goff[0]) + 1); }
    else raw.g[0] = ldnt((const GAS u32x3_a4*)(base + ln.goff[0]));
  }
#pragma unroll
  for (int g = 1; g < 4; g++)
    if (g < p.n_group_cols) {
      const PgGroupCol& gc = p.gcols[g];
      raw.g[g] = ldnt((const GAS u32x3_a4*)(p2o_sgpr(gptr<uint8_t>(gc.data) + at * (size_t)gc.bits) + ln.goff[g]));
    }
  if (OSK != 0) {
    const PgValueSrc& V = p.srcs[0];
    const uint32_t bits = OSK == 1 ? 32u : (uint32_t)V.bits;
    const GAS u32x4_a4* q = (const GAS u32x4_a4*)(p2o_sgpr(gptr<uint8_t>(V.data) + at * (size_t)bits) + ln.soff);
    raw.s0 = ldnt(q);
    raw.s1 = ldnt(q + 1);
  }
}
// keys and plane-0 dwords of the lane's 8 docs
template <bool G0WIDE, int OSK>
DEVFN void p2o_decode(const PgQueryPlan& p, const P2OctLane& ln, const P2OctRaw& raw, uint32_t local_mask, uint32_t (&key)[8], uint32_t (&d)[8]) {
  {
    uint32_t v[8];
    if (G0WIDE) oct_decode_source(p.gcols[0].bits, raw.g0a, raw.g0b, ln.gsel[0], v);
    else oct_decode_group(p.gcols[0].bits, raw.g[0], ln.gsel[0], v);
#pragma unroll
    for (int j = 0; j < 8; j++) key[j] = v[j];   // mult of column 0 is 1
  }
#pragma unroll
  for (int g = 1; g < 4; g++)
    if (g < p.n_group_cols) {
      uint32_t v[8];
      oct_decode_group(p.gcols[g].bits, raw.g[g], ln.gsel[g], v);
      const uint32_t mult = (uint32_t)p.gcols[g].mult;
#pragma unroll
      for (int j = 0; j < 8; j++) key[j] = mad24(v[j], mult, key[j]);   // dictId < 2^8, mult < 2^24 (planner)
    }
  if (OSK == 0) {
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = key[j] & local_mask;
    return;
  }
  uint32_t f[8];
  if (OSK == 1) {
    const uint32_t w[8] = {raw.s0.x, raw.s0.y, raw.s0.z, raw.s0.w, raw.s1.x, raw.s1.y, raw.s1.z, raw.s1.w};
    const uint32_t bias = (uint32_t)p.p2_fbias[0];
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = bswap32(w[j]) - bias;
  } else {
    oct_decode_source(p.srcs[0].bits, raw.s0, raw.s1, ln.ssel, f);
  }
  const uint32_t sh = (uint32_t)p.pk_shift[0];
#pragma unroll
  for (int j = 0; j < 8; j++) d[j] = (key[j] & local_mask) | (f[j] << sh);
}
// validity of the lane's 8 docs of sub-tile `sub` of the tile wavefront `wave` owns in quartet g (0 beyond the segment)
DEVFN uint32_t p2o_mask8(int g, int n_quartets, int wave, int lane, int sub, int n_wtiles, int64_t num_docs) {
  const int wt_raw = g * PG_P2_WAVES + wave;
  if (g >= n_quartets || wt_raw >= n_wtiles) return 0u;   // wave-uniform
  const int64_t rem = num_docs - ((int64_t)wt_raw * PG_WAVE_DOCS + (int64_t)sub * OCT_SUB_DOCS);   // wave-uniform
  if (rem >= OCT_SUB_DOCS) return 0xFFu;
  const int64_t r = rem - 8 * lane;
  return r >= 8 ? 0xFFu : (r <= 0 ? 0u : ((1u << (uint32_t)r) - 1u));
}

// PG_P2_TIMING (measurement variant only, tools/build_variants.sh): cycles per phase of the scatter's round, summed over the wavefronts
// (s_memtime; slot 2 k = phase k, slot 2 k + 1 = the wait at the barrier behind it) -> p2_ctrl[PG_P2_CTRL_TIMING ..], printed by the host
#ifdef PG_P2_TIMING
#define P2_TICK(SLOT) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&s_time[SLOT], now_ - tick_); tick_ = now_; } while (0)
#else
#define P2_TICK(SLOT) do { } while (0)
#endif
// LDS of a scatter workgroup (dwords): hist, off, cnt, lo_cnt, cur, left [NBp each] | pool [PG_P2_POOL] | ctrl [8] |
// lines [R / 32 + NB + 1][2] | sorted [T][R] | lo [T][NB][32]
struct P2Stage {
  uint32_t *hist, *off, *cnt, *lo_cnt, *cur, *left, *pool, *ctrl, *lines, *sorted, *lo;
  uint32_t nbp;
};

// quad-layout match mask of wavefront `wave`'s tile of quartet g (0 beyond the doc space: the segment, or — stream mode — the tuple stream)
DEVFN uint32_t p2_tile_mask(const PgQueryPlan& p, int g, int n_quartets, int wave, int lane, int n_wtiles, int64_t num_docs) {
  const int wt_raw = g * PG_P2_WAVES + wave;
  if (g >= n_quartets || wt_raw >= n_wtiles) return 0u;   // wave-uniform
  const int64_t rem = num_docs - (int64_t)wt_raw * PG_WAVE_DOCS;
  const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
  if (p.match_words) return lin_to_quad(gptr<uint32_t>(p.match_words)[(int64_t)wt_raw * 64 + lane] & valid_lin_mask(n_valid, lane), lane);
  return valid_quad_mask(n_valid, lane);
}
DEVFN int p2_tile_of(int g, int wave, int n_wtiles) {   // the tile whose columns the wavefront reads (clamped: loads stay in bounds)
  const int wt_raw = g * PG_P2_WAVES + wave;
  return wt_raw < n_wtiles ? wt_raw : (n_wtiles > 0 ? n_wtiles - 1 : 0);
}
// mb8: a batch's 8 mask bits; kq: its first quad slot.  Quads without a matching doc re-read quad 0 (a line the tile needs anyway).
DEVFN void p2_quads_of(int lane, uint32_t mb8, int kq, uint32_t (&qi)[P2_QA]) {
#pragma unroll
  for (int u = 0; u < P2_QA; u++) qi[u] = ((mb8 >> (4 * u)) & 0xFu) ? (uint32_t)((kq + u) * 64 + lane) : 0u;
}

// PG_P2_BATCH chunk ids from this workgroup's stripe into the LDS ring (wavefront 0); ids beyond the stripe become the spill chunk
DEVFN void p2_claim_batch(const PgQueryPlan& p, uint32_t* pool, uint32_t tail, int lane) {
  const uint32_t stripe = blockIdx.x & (PG_P2_STRIPES - 1u), cap_s = (uint32_t)p.p2_stripe_cap;
  uint32_t basec = 0;
  if (lane == 0) basec = atomicAdd(p.p2_ctrl + PG_P2_CTRL_STRIPE0 + stripe * PG_P2_STRIPE_DWORDS, (uint32_t)PG_P2_BATCH);
  basec = (uint32_t)__builtin_amdgcn_readfirstlane((int)basec);
  for (uint32_t i = (uint32_t)lane; i < PG_P2_BATCH; i += 64u)
    pool[(tail + i) & (PG_P2_POOL - 1u)] = basec + i < cap_s ? stripe * cap_s + basec + i : (uint32_t)p.p2_capacity;
}
template <int T, int Q, bool FAST, int SRC, int OCT = 0, bool G0WIDE = false, int OSK = 0>
__device__ __forceinline__ void p2_scatter_body(const PgQueryPlan& p) {
  static_assert(OCT == 0 || (T == 1 && Q == 4 && !FAST), "the oct-layout phase A: one plane, two sub-tiles (16 docs) per lane and round");
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  constexpr int NBATCH = Q / P2_QA;               // quads per lane and round: Q, decoded in batches of P2_QA
  constexpr uint32_t R = PG_P2_WAVES * Q * 256;   // tuples per round
  constexpr uint32_t CH = PG_P2_CHUNK;
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  const int NB = p.radix_buckets;
  P2Stage S;
  S.nbp = (uint32_t)((NB + 63) & ~63);   // buckets the bookkeeping wavefront visits
  constexpr uint32_t NBA = PG_P2_MAX_BUCKETS;   // array stride: compile-time, so that the array bases are constants, not SGPRs
  uint32_t* base = reinterpret_cast<uint32_t*>(smem);
  S.hist = base; S.off = S.hist + NBA; S.cnt = S.off + NBA; S.lo_cnt = S.cnt + NBA; S.cur = S.lo_cnt + NBA; S.left = S.cur + NBA;
  S.pool = S.left + NBA; S.ctrl = S.pool + PG_P2_POOL; S.lines = S.ctrl + 8; S.sorted = S.lines + 2u * (R / PG_P2_LINE + (uint32_t)NB + 1u); S.lo = S.sorted + (size_t)T * R;
  for (uint32_t i = (uint32_t)t; i < 6u * NBA; i += P2_THREADS) base[i] = 0;
  if (t < 8) S.ctrl[t] = 0;
#ifdef PG_P2_TIMING
  __shared__ unsigned long long s_time[16];
  if (t < 16) s_time[t] = 0;
  unsigned long long tick_ = __builtin_amdgcn_s_memtime();
#endif
  constexpr bool STREAM = SRC == 3;   // the doc space is the survivor stream of pg_oct_p: its length is on the device
  // MatchAllFilterOperator: no filter pass ran in front — every doc matches (ExecutionStatistics.numDocsScanned)
  if (!STREAM && !p.match_words && blockIdx.x == 0 && t == 0) atomicAdd(p.stats, (unsigned long long)p.num_docs);
  __syncthreads();
  const uint32_t local_mask = (1u << p.radix_shift) - 1u;
  const size_t lo_plane = (size_t)NB * PG_P2_LINE;
  uint32_t* const tuples = p.p2_tuples;
  const int64_t num_docs = (int64_t)p.num_docs;
  int n_wtiles = p.n_wtiles;
  __shared__ P2StreamLds<STREAM> s_stream;
  const uint32_t* tile_start = reinterpret_cast<const uint32_t*>(&s_stream);
  if (STREAM) {
    uint32_t* ts = reinterpret_cast<uint32_t*>(&s_stream);
    for (int i = t; i <= p.oct_n_regions; i += P2_THREADS) ts[i] = gptr<uint32_t>(p.oct_cursor)[PG_OCT_CTRL_TILE_START + i];
    __syncthreads();
    n_wtiles = (int)tile_start[p.oct_n_regions];   // the stream's tiles, numbered across the regions
  }
  const int n_quartets = (n_wtiles + PG_P2_WAVES - 1) / PG_P2_WAVES;
  // round sequence of this workgroup: (quartet g, quad slots k0 .. k0 + Q - 1), g = blockIdx.x, blockIdx.x + gstride, ...
  // Stream mode: the host sized the grid for "every offer of the pass survives"; a workgroup that took one quartet of a short stream would
  // leave NB chunks of one or two lines each (the aggregation pass then reads mostly padding).  Only as many workgroups as give each
  // at least PG_P2_STREAM_MIN_QUARTETS quartets take part; the others leave at once (4 quartets per workgroup measured slower: a round
  // costs ~10 us of latency whatever it holds, profiles/r04_b_kernels__cfg5_.txt).
  int gstride = (int)gridDim.x;
  if (STREAM) {
    const int want = (n_quartets + PG_P2_STREAM_MIN_QUARTETS - 1) / PG_P2_STREAM_MIN_QUARTETS;
    gstride = want < gstride ? (want > 0 ? want : 1) : gstride;
    if ((int)blockIdx.x >= gstride) return;   // workgroup-uniform, before any further barrier
  }
  int g = (int)blockIdx.x;
  P2StreamTile stile = STREAM ? p2_stream_tile(p, tile_start, g * PG_P2_WAVES + wave, g < n_quartets ? n_wtiles : 0) : P2StreamTile{0u, 0};
  uint32_t m = STREAM ? valid_quad_mask(stile.n_valid, lane) : p2_tile_mask(p, g, n_quartets, wave, lane, n_wtiles, num_docs);
  int k0 = 0;
  P2Raw raw;   // the loads of the round's first batch, requested one round ahead
  P2OctLane oln;
  P2OctRaw or0, or1;   // OCT: the round's two sub-tiles, each re-requested (for the next round) right after its decode
  if (OCT) {
    p2o_lane_setup<OSK>(p, lane, oln);
    const int wt0 = p2_tile_of(g, wave, n_wtiles);
    p2o_issue<G0WIDE, OSK>(p, oln, wt0, 0, or0);
    p2o_issue<G0WIDE, OSK>(p, oln, wt0, 1, or1);
  }
  if (FAST) {
    uint32_t qi[P2_QA];
    p2_quads_of(lane, m & 0xFFu, 0, qi);
    if (STREAM) p2_issue_stream(p, qi, stile.base, raw);
    else p2_issue(p, qi, p2_tile_of(g, wave, n_wtiles), raw);
  }
  while (g < n_quartets) {   // workgroup-uniform
    const int wt = p2_tile_of(g, wave, n_wtiles);
    const uint32_t mb = (m >> (4 * k0)) & (Q == 8 ? 0xFFFFFFFFu : ((1u << (4 * (Q & 7))) - 1u));
    uint32_t d[T][Q * 4];
    uint32_t br[Q * 4];
    if (OCT) {
      // ---- A (oct layout): sub-tiles 2 sp, 2 sp + 1 of the tile (sp = k0 / 4) ------------------------------------------------------------
      int gn = g, kn = k0 + Q;
      if (kn >= 8) { kn = 0; gn = g + gstride; }
      const int wtn = p2_tile_of(gn, wave, n_wtiles);
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const int sub = (k0 >> 2) * 2 + s2;
        const uint32_t m8 = p2o_mask8(g, n_quartets, wave, lane, sub, n_wtiles, num_docs);
        uint32_t key[8], dd[8];
        p2o_decode<G0WIDE, OSK>(p, oln, s2 == 0 ? or0 : or1, local_mask, key, dd);
#ifdef PG_P2_TIMING
        { const uint32_t probe_ = key[0] + dd[7]; asm volatile("" :: "v"(probe_)); }   // the decode's results exist here (the loads have arrived)
        P2_TICK(9 + 2 * s2);
#endif
        // (rank atomics issued unconditionally — a trash counter per lane for docs beyond the segment, so that the eight returning LDS
        // atomics leave back to back instead of one wait each — were built and measured: 27 spilled VGPRs at four workgroups per CU,
        // 1.04 ms against 0.92 ms per 2 x 10^8 docs on the 40 k-group row, 0.98 ms at three workgroups per CU;
        // profiles/r05_partition_scatter_experiments.txt)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          d[0][s2 * 8 + j] = dd[j];
          const uint32_t b = key[j] >> p.radix_shift;
          br[s2 * 8 + j] = 0xFFFFFFFFu;
          if ((m8 >> j) & 1u) br[s2 * 8 + j] = (b << 16) | atomicAdd(&S.hist[b], 1u);
        }
#ifdef PG_P2_TIMING
        { const uint32_t probe_ = br[s2 * 8] + br[s2 * 8 + 7]; asm volatile("" :: "v"(probe_)); }   // the ranks have returned
        P2_TICK(10 + 2 * s2);
#endif
      }
      // the next round's loads, in flight across the phases below (requested only now: with a buffer re-requested right after its
      // decode both buffers and the second sub-tile's temporaries were live together — 27 spilled VGPRs, whose reloads are VMEM
      // operations that make every wait a vmcnt(0))
      p2o_issue<G0WIDE, OSK>(p, oln, wtn, (kn >> 2) * 2, or0);
      p2o_issue<G0WIDE, OSK>(p, oln, wtn, (kn >> 2) * 2 + 1, or1);
    }
    // ---- A: tuples of this wavefront's Q quads per lane; histogram + rank in one returning LDS add ---------------------------------
#pragma unroll
    for (int h = 0; h < (OCT ? 0 : NBATCH); h++) {
      uint32_t qi[P2_QA];
      p2_quads_of(lane, (mb >> (h * P2_QA * 4)) & 0xFFu, k0 + h * P2_QA, qi);
      uint32_t key[P2_QA * 4], dd[T][P2_QA * 4];
      uint32_t vm = 0xFFu;   // stream mode: the batch's entries that are not padding
      if (STREAM) {
        if (h > 0) p2_issue_stream(p, qi, stile.base, raw);
        vm = p2_decode_stream(p, raw, local_mask, key, reinterpret_cast<uint32_t(&)[1][P2_QA * 4]>(dd));
      } else if (FAST) {
        if (h > 0) p2_issue(p, qi, wt, raw);
        p2_decode<T, SRC>(p, raw, qi, local_mask, key, dd);
      } else {
        p2_decode_generic<T>(p, qi, wt, local_mask, key, dd);
      }
      if (T > 1 && p.p2_docid_plane >= 0) {
#pragma unroll
        for (int pl = 1; pl < T; pl++)
          if (pl == p.p2_docid_plane) {
#pragma unroll
            for (int j = 0; j < P2_QA * 4; j++)
              dd[pl][j] = (uint32_t)wt * PG_WAVE_DOCS + 4u * (uint32_t)((k0 + h * P2_QA + (j >> 2)) * 64 + lane) + (uint32_t)(j & 3);
          }
      }
#pragma unroll
      for (int j = 0; j < P2_QA * 4; j++) {
#pragma unroll
        for (int pl = 0; pl < T; pl++) d[pl][h * P2_QA * 4 + j] = dd[pl][j];
        const uint32_t b = key[j] >> p.radix_shift;
        br[h * P2_QA * 4 + j] = 0xFFFFFFFFu;
        if (((mb >> (h * P2_QA * 4 + j)) & 1u) && ((vm >> j) & 1u)) br[h * P2_QA * 4 + j] = (b << 16) | atomicAdd(&S.hist[b], 1u);
      }
    }
    // next round: its mask, and (FAST) the loads of its first batch — in flight across the phases below
    int g_next = g, k0_next = k0 + Q;
    uint32_t m_next = m;
    P2StreamTile stile_next = stile;
    if (k0_next >= 8) {
      k0_next = 0;
      g_next = g + gstride;
      if (STREAM) {
        stile_next = p2_stream_tile(p, tile_start, g_next * PG_P2_WAVES + wave, g_next < n_quartets ? n_wtiles : 0);
        m_next = valid_quad_mask(stile_next.n_valid, lane);
      } else {
        m_next = p2_tile_mask(p, g_next, n_quartets, wave, lane, n_wtiles, num_docs);
      }
    }
    if (FAST) {
      uint32_t qi[P2_QA];
      p2_quads_of(lane, (m_next >> (4 * k0_next)) & 0xFFu, k0_next, qi);
      if (STREAM) p2_issue_stream(p, qi, stile_next.base, raw);
      else p2_issue(p, qi, p2_tile_of(g_next, wave, n_wtiles), raw);
    }
    P2_TICK(0);
    __syncthreads();
    P2_TICK(1);
    // ---- B: wavefront 0: histogram → offsets; the round's whole lines with their sources and destinations; chunk ids ------------------
    if (wave == 0) {
      uint32_t carry = 0, need = 0;
      for (uint32_t v = 0; v < S.nbp; v += 64u) {
        const uint32_t b = v + (uint32_t)lane;
        const uint32_t n = S.hist[b];
        const uint32_t x = p2_wave_scan(n);
        S.off[b] = carry + x - n;
        S.cnt[b] = n;
        S.hist[b] = 0;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        const uint32_t out = (S.lo_cnt[b] + n) & ~(PG_P2_LINE - 1u), lf = S.left[b];
        if (out > lf) need += (out - lf + CH - 1u) / CH;
      }
      need = (uint32_t)__builtin_amdgcn_readlane((int)p2_wave_scan(need), 63);
      {   // the ring holds the chunk ids this round can take
        const uint32_t head = S.ctrl[0];
        uint32_t tail = S.ctrl[1];
        if (need > PG_P2_POOL - PG_P2_BATCH) need = PG_P2_POOL - PG_P2_BATCH;   // at most R / CHUNK + buckets (272)
        while (tail - head < need) {   // wave-uniform
          p2_claim_batch(p, S.pool, tail, lane);
          tail += PG_P2_BATCH;
        }
        if (lane == 0) S.ctrl[1] = tail;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      uint32_t line_carry = 0;
      for (uint32_t v = 0; v < S.nbp; v += 64u) {
        const uint32_t b = v + (uint32_t)lane;
        const uint32_t n = S.cnt[b], lo_n = S.lo_cnt[b], o = S.off[b];
        const uint32_t n_lines = (lo_n + n) >> 5;
        const uint32_t x = p2_wave_scan(n_lines);
        const uint32_t at = line_carry + x - n_lines;
        line_carry += (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        uint32_t c = S.cur[b], l = S.left[b];
        for (uint32_t i = 0; i < n_lines; i++) {   // per lane: 0..2 lines unless the keys are skewed
          if (l == 0) {   // next chunk of the stream, recorded as full (the epilogue rewrites a stream's last record)
            uint32_t id = S.pool[atomicAdd(&S.ctrl[0], 1u) & (PG_P2_POOL - 1u)];
            if (id >= (uint32_t)p.p2_capacity) { p.p2_ctrl[1] = 1u; id = (uint32_t)p.p2_capacity; }   // spill chunk + error flag
            else p.p2_meta[id] = b | ((uint32_t)(CH / PG_P2_LINE) << 16);
            c = id * CH;
            l = CH;
          }
          // line i of the bucket's run: destination | (run start in `sorted` (13 bits), bucket (8), line (8))
          *reinterpret_cast<u32x2*>(S.lines + 2u * (at + i)) = (u32x2){c, o | (b << 13) | (i << 21)};
          c += PG_P2_LINE; l -= PG_P2_LINE;
        }
        if (n_lines) { S.cur[b] = c; S.left[b] = l; }
      }
      if (lane == 0) S.ctrl[2] = line_carry;
    }
    P2_TICK(2);
    __syncthreads();
    P2_TICK(3);
    // ---- C: tuples to their sorted positions (offsets first, all in flight; then the stores) --------------------------------------------
    {
      uint32_t pos[Q * 4];
#pragma unroll
      for (int j = 0; j < Q * 4; j++) pos[j] = S.off[br[j] == 0xFFFFFFFFu ? 0u : (br[j] >> 16)];
#pragma unroll
      for (int j = 0; j < Q * 4; j++)
        if (br[j] != 0xFFFFFFFFu) {
#pragma unroll
          for (int pl = 0; pl < T; pl++) S.sorted[(size_t)pl * R + pos[j] + (br[j] & 0xFFFFu)] = d[pl][j];
        }
    }
    P2_TICK(4);
    __syncthreads();
    P2_TICK(5);
    // ---- D1: whole lines out: a half wavefront per line (32 tuples = 128 bytes per plane), four lines per step ----------------------------
    {
      const uint32_t n_lines = S.ctrl[2];
      const uint32_t li = (uint32_t)lane & 31u, hw = (uint32_t)(wave * 2 + (lane >> 5));
      constexpr uint32_t NH = PG_P2_WAVES * 2u;
      for (uint32_t e0 = hw; e0 < n_lines; e0 += 4u * NH) {
        u32x2 e[4];
        uint32_t lo_n[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t ei = e0 + (uint32_t)k * NH;
          e[k] = *reinterpret_cast<const u32x2*>(S.lines + 2u * (ei < n_lines ? ei : e0));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) lo_n[k] = S.lo_cnt[(e[k].y >> 13) & 0xFFu];
        uint32_t x[4][T];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t o = e[k].y & 0x1FFFu, b = (e[k].y >> 13) & 0xFFu, v = (e[k].y >> 21) * PG_P2_LINE + li;
#pragma unroll
          for (int pl = 0; pl < T; pl++)
            x[k][pl] = v < lo_n[k] ? S.lo[(size_t)pl * lo_plane + (size_t)b * PG_P2_LINE + v] : S.sorted[(size_t)pl * R + o + v - lo_n[k]];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (e0 + (uint32_t)k * NH < n_lines) {
#pragma unroll
            for (int pl = 0; pl < T; pl++) tuples[(size_t)pl * (size_t)p.p2_plane_stride + e[k].x + li] = x[k][pl];
          }
      }
    }
    P2_TICK(6);
    __syncthreads();
    P2_TICK(7);
    // ---- D2: what did not fill a line is the bucket's leftover for the next round (a half wavefront per bucket, four per step) -----------
    {
      const uint32_t li = (uint32_t)lane & 31u, hw = (uint32_t)(wave * 2 + (lane >> 5));
      constexpr uint32_t NH = PG_P2_WAVES * 2u;
      for (uint32_t b0 = hw; b0 < (uint32_t)NB; b0 += 4u * NH) {
        uint32_t n[4], lo_n[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t b = b0 + (uint32_t)k * NH;
          const bool on = b < (uint32_t)NB;
          n[k] = on ? S.cnt[b] : 0u;
          lo_n[k] = on ? S.lo_cnt[b] : 0u;
          o[k] = on ? S.off[b] : 0u;
        }
        uint32_t x[4][T];
        uint32_t at[4];
        bool st[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t total = lo_n[k] + n[k], out = total & ~(PG_P2_LINE - 1u), rem = total & (PG_P2_LINE - 1u);
          // out == 0: the new tuples join the leftover; else the run's tail (all new tuples: out >= 32 > lo_n) is the next leftover
          st[k] = n[k] != 0 && (out == 0 ? li < n[k] : li < rem);
          const uint32_t src = out == 0 ? o[k] + li : o[k] + out + li - lo_n[k];
          at[k] = out == 0 ? lo_n[k] + li : li;
#pragma unroll
          for (int pl = 0; pl < T; pl++) x[k][pl] = S.sorted[(size_t)pl * R + (st[k] ? src : 0u)];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t b = b0 + (uint32_t)k * NH;
          if (st[k]) {
#pragma unroll
            for (int pl = 0; pl < T; pl++) S.lo[(size_t)pl * lo_plane + (size_t)b * PG_P2_LINE + at[k]] = x[k][pl];
          }
          if (li == 0 && n[k] != 0) S.lo_cnt[b] = (lo_n[k] + n[k]) & (PG_P2_LINE - 1u);
        }
      }
    }
    // no barrier here: the next round's phase A touches only `hist` (reset in B); its barrier orders D2 before the next B
    g = g_next; k0 = k0_next; m = m_next; stile = stile_next;
    P2_TICK(8);
  }
#ifdef PG_P2_TIMING
  __syncthreads();
  if (t < 13) atomicAdd(reinterpret_cast<unsigned long long*>(p.p2_ctrl + PG_P2_CTRL_TIMING) + t, s_time[t]);
#endif
  // ---- epilogue: leftover lines padded with PG_RADIX_INVALID_KEY, the last chunk of every stream recorded with its true fill --------
  __syncthreads();
  if (wave == 0) {
    const uint32_t head = S.ctrl[0];
    uint32_t tail = S.ctrl[1];
    while (tail - head < (uint32_t)NB) {
      p2_claim_batch(p, S.pool, tail, lane);
      tail += PG_P2_BATCH;
    }
    if (lane == 0) S.ctrl[1] = tail;
  }
  __syncthreads();
  for (uint32_t b = (uint32_t)(wave * 2 + (lane >> 5)); b < (uint32_t)NB; b += PG_P2_WAVES * 2u) {   // a half wavefront per bucket
    const uint32_t lo_n = S.lo_cnt[b], li = (uint32_t)lane & 31u;
    uint32_t c = S.cur[b], l = S.left[b];
    if (lo_n > 0) {
      if (l == 0) {
        uint32_t id = 0;
        if (li == 0) {
          id = S.pool[atomicAdd(&S.ctrl[0], 1u) & (PG_P2_POOL - 1u)];
          if (id >= (uint32_t)p.p2_capacity) { p.p2_ctrl[1] = 1u; id = (uint32_t)p.p2_capacity; }
        }
        id = (uint32_t)__shfl((int)id, lane & 32, 64);
        c = id * CH;
        l = CH;
      }
      const uint32_t* lob = S.lo + (size_t)b * PG_P2_LINE;
#pragma unroll
      for (int pl = 0; pl < T; pl++) {
        const uint32_t x = li < lo_n ? lob[(size_t)pl * lo_plane + li] : (pl == 0 ? PG_RADIX_INVALID_KEY : 0u);
        tuples[(size_t)pl * (size_t)p.p2_plane_stride + c + li] = x;
      }
      c += PG_P2_LINE; l -= PG_P2_LINE;
    }
    if (li == 0 && c != 0 && (l > 0 || lo_n > 0)) {   // the stream's last chunk: its true fill (a chunk the main loop filled exactly keeps "full")
      const uint32_t id = (c - 1u) / CH;
      if (id < (uint32_t)p.p2_capacity) p.p2_meta[id] = b | (((CH - l) / PG_P2_LINE) << 16);
    }
  }
}

#define P2_SCATTER(NAME, T, Q, FAST, SRC) \
  extern "C" __global__ void __launch_bounds__(P2_THREADS, 4) NAME(const PgQueryPlan p) { p2_scatter_body<T, Q, FAST, SRC>(p); }
P2_SCATTER(pg_p2_scatter_1, 1, 4, false, 2)
P2_SCATTER(pg_p2_scatter_2, 2, 4, false, 2)
P2_SCATTER(pg_p2_scatter_3, 3, 2, false, 2)
P2_SCATTER(pg_p2_scatter_4, 4, 2, false, 2)
P2_SCATTER(pg_p2_scatter_1f, 1, 4, true, 2)
P2_SCATTER(pg_p2_scatter_2f, 2, 4, true, 2)
P2_SCATTER(pg_p2_scatter_1f_key, 1, 4, true, 0)   // key only (COUNT over a big key space)
P2_SCATTER(pg_p2_scatter_1f_hll, 1, 4, true, 1)   // config 5
P2_SCATTER(pg_p2_scatter_stream, 1, 4, true, 3)   // the survivors of the pruned-offer passes (pg_oct_p)
// oct-layout phase A (plans without a filter pass, PgQueryPlan::p2_oct_a): [first group column <= 8 / <= 24 bits] x [no source, raw INT, dictIds]
#define P2_SCATTER_OCT(NAME, G0WIDE, OSK, WGS) \
  extern "C" __global__ void __launch_bounds__(P2_THREADS, WGS) NAME(const PgQueryPlan p) { p2_scatter_body<1, 4, false, 2, 1, G0WIDE, OSK>(p); }
P2_SCATTER_OCT(pg_p2_scatter_o_key, false, 0, 4)
P2_SCATTER_OCT(pg_p2_scatter_o_raw, false, 1, 4)
P2_SCATTER_OCT(pg_p2_scatter_o_dict, false, 2, 4)
P2_SCATTER_OCT(pg_p2_scatter_ow_key, true, 0, 4)
P2_SCATTER_OCT(pg_p2_scatter_ow_raw, true, 1, 4)
P2_SCATTER_OCT(pg_p2_scatter_ow_dict, true, 2, 4)
extern "C" const int pg_p2_round_quads[5] = {0, 4, 4, 2, 2};   // Q per plane count (the host sizes the LDS with it)

// ---- chunk index: the chunk records grouped by bucket (counting sort of p2_meta), three small launches ------------------------------
extern "C" __global__ void __launch_bounds__(1024) pg_p2_index_count_kernel(const PgQueryPlan p) {
  __shared__ uint32_t s_hist[PG_P2_MAX_BUCKETS];
  const int t = threadIdx.x;
  if (t < PG_P2_MAX_BUCKETS) s_hist[t] = 0;
  __syncthreads();
  const uint32_t n_alloc = (uint32_t)p.p2_capacity;   // ids are handed out per stripe: the records of unclaimed ids stay 0xFFFFFFFF
  for (uint32_t i = blockIdx.x * 1024u + (uint32_t)t; i < n_alloc; i += gridDim.x * 1024u) {
    const uint32_t mt = p.p2_meta[i];
    if ((mt & 0xFFFFu) < (uint32_t)p.radix_buckets && (mt >> 16) != 0) atomicAdd(&s_hist[mt & 0xFFFFu], 1u);
  }
  __syncthreads();
  if (t < p.radix_buckets && s_hist[t]) atomicAdd(&p.p2_ctrl[PG_P2_CTRL_COUNTS + t], s_hist[t]);
}
extern "C" __global__ void __launch_bounds__(PG_P2_MAX_BUCKETS) pg_p2_index_scan_kernel(const PgQueryPlan p) {
  __shared__ uint32_t s_c[PG_P2_MAX_BUCKETS];
  const int t = threadIdx.x;
  s_c[t] = t < p.radix_buckets ? p.p2_ctrl[PG_P2_CTRL_COUNTS + t] : 0u;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int b = 0; b < p.radix_buckets; b++) {
      p.p2_ctrl[PG_P2_CTRL_STARTS + b] = run;
      p.p2_ctrl[PG_P2_CTRL_CURSOR + b] = run;
      run += s_c[b];
    }
    p.p2_ctrl[PG_P2_CTRL_STARTS + p.radix_buckets] = run;
  }
}
extern "C" __global__ void __launch_bounds__(1024) pg_p2_index_fill_kernel(const PgQueryPlan p) {
  __shared__ uint32_t s_hist[PG_P2_MAX_BUCKETS], s_base[PG_P2_MAX_BUCKETS];
  const int t = threadIdx.x;
  const uint32_t n_alloc = (uint32_t)p.p2_capacity;   // ids are handed out per stripe: the records of unclaimed ids stay 0xFFFFFFFF
  // the workgroup's share of the records is a contiguous range: count, claim one range per bucket, then place
  const uint32_t per = (n_alloc + gridDim.x - 1u) / gridDim.x;
  const uint32_t lo = blockIdx.x * per;
  uint32_t hi = lo + per;
  if (hi > n_alloc) hi = n_alloc;
  if (t < PG_P2_MAX_BUCKETS) s_hist[t] = 0;
  __syncthreads();
  for (uint32_t i = lo + (uint32_t)t; i < hi; i += 1024u) {
    const uint32_t mt = p.p2_meta[i];
    if ((mt & 0xFFFFu) < (uint32_t)p.radix_buckets && (mt >> 16) != 0) atomicAdd(&s_hist[mt & 0xFFFFu], 1u);
  }
  __syncthreads();
  if (t < p.radix_buckets) {
    s_base[t] = s_hist[t] ? atomicAdd(&p.p2_ctrl[PG_P2_CTRL_CURSOR + t], s_hist[t]) : 0u;
    s_hist[t] = 0;
  }
  __syncthreads();
  for (uint32_t i = lo + (uint32_t)t; i < hi; i += 1024u) {
    const uint32_t mt = p.p2_meta[i];
    const uint32_t b = mt & 0xFFFFu;
    if (b < (uint32_t)p.radix_buckets && (mt >> 16) != 0) p.p2_list[s_base[b] + atomicAdd(&s_hist[b], 1u)] = i | ((mt >> 16) << 27);
  }
}

// ---- aggregation pass ------------------------------------------------------------------------------------------------------------
// work item w = bucket * slices + slice aggregates its share of the bucket's chunk list: windows of PG_P2_LIST records are copied into
// LDS, one wavefront per chunk (4 tuples per lane and plane), the loads of the wavefront's next two chunks in flight meanwhile.
template <int T>
DEVFN uint32_t p2_pick(const u32x4 (&v)[T], int pl, int e) {
  uint32_t x = 0;
#pragma unroll
  for (int q = 0; q < T; q++) {
    const uint32_t y = e == 0 ? v[q].x : (e == 1 ? v[q].y : (e == 2 ? v[q].z : v[q].w));
    if (q == pl) x = y;
  }
  return x;
}

// Loads one chunk's tuples of this lane (4 per plane) and returns whether the lane has any.  The load targets are touched by nothing
// but the load (a select between the loaded value and a constant is a USE: the compiler then waits for the load on the spot); lanes
// beyond the chunk's filled lines re-read its first line and are masked by the returned flag.
template <int T>
DEVFN bool p2_fetch(const GAS uint32_t* tuples, size_t plane_stride, const uint32_t* list, uint32_t n_list, uint32_t ci, int lane, u32x4 (&v)[T]) {
  if (ci >= n_list) return false;   // wave-uniform; v stays undefined and is never consumed
  const uint32_t e = list[ci];
  const bool on = (uint32_t)(lane >> 3) < (e >> 27);
  const size_t at = (size_t)(e & 0x7FFFFFFu) * PG_P2_CHUNK + (size_t)(on ? lane : (lane & 7)) * 4u;
#pragma unroll
  for (int pl = 0; pl < T; pl++) v[pl] = ldnt((const GAS u32x4*)(tuples + (size_t)pl * plane_stride + at));
  return on;
}
// GATHER: some source travels as a dictId and is looked up here.  Without such a source the consumer issues no global load at all, so
// the wait in front of it is for the OLDEST chunk only and the younger two keep travelling (any gather drains them: vmcnt counts
// in order).
template <int T, bool GATHER>
DEVFN void p2_consume(const PgQueryPlan& p, const u32x4 (&cur)[T], bool on, int64_t* table, uint32_t* aux_lds, uint32_t slots, uint32_t local_mask) {
  if (!on) return;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t d0 = p2_pick<T>(cur, 0, e);
    if (d0 == PG_RADIX_INVALID_KEY) continue;
    const uint32_t k = d0 & local_mask;
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      int64_t* acc = table + (size_t)o * slots + k;
      if (op.src < 0) {
        // COUNT: a work item sees < 2^32 tuples, so the slot's low dword takes the add (ds_add_u32: half the data of ds_add_u64)
        if (op.fn == PG_ACC_COUNT) atomicAdd(reinterpret_cast<uint32_t*>(acc), 1u);
        else atomicMin(reinterpret_cast<long long*>(acc), (long long)p2_pick<T>(cur, p.p2_docid_plane, e));   // MIN(docId)
        continue;
      }
      const PgValueSrc& V = p.srcs[op.src];
      const int kind = p.p2_fkind[op.src], fpl = p.p2_fplane[op.src];
      const uint32_t bits = (uint32_t)p.pk_bits[op.src];
      uint32_t f = p2_pick<T>(cur, fpl, e) >> p.pk_shift[op.src];
      if (bits < 32u) f &= (1u << bits) - 1u;
      if (GATHER && kind == PG_P2_F_DICTID) {
        if (V.val_type == PG_V_I32) acc_from_int(acc, op, (int64_t)(int32_t)gptr<uint32_t>(V.dict)[f]);
        else if (V.val_type == PG_V_I64) acc_from_int(acc, op, (int64_t)gptr<uint64_t>(V.dict)[f]);
        else if (V.val_type == PG_V_F32) acc_from_double(acc, op, (double)__uint_as_float(gptr<uint32_t>(V.dict)[f]), V.fx_q);
        else acc_from_double(acc, op, __longlong_as_double((int64_t)gptr<uint64_t>(V.dict)[f]), V.fx_q);
      } else if (kind == PG_P2_F_RAW32) {
        if (V.val_type == PG_V_I32) acc_from_int(acc, op, (int64_t)(int32_t)(f + (uint32_t)p.p2_fbias[op.src]));
        else acc_from_double(acc, op, (double)__uint_as_float(f), V.fx_q);
      } else {   // PG_P2_F_RAW64
        const uint64_t v64 = ((uint64_t)p2_pick<T>(cur, fpl + 1, e) << 32) | (uint64_t)f;
        if (V.val_type == PG_V_I64) acc_from_int(acc, op, (int64_t)v64);
        else acc_from_double(acc, op, __longlong_as_double((int64_t)v64), V.fx_q);
      }
    }
    uint32_t aux_off = 0;
    for (int x = 0; x < p.n_aux; x++) {
      const PgAuxOp& A = p.aux[x];
      uint32_t f = p2_pick<T>(cur, p.p2_fplane[A.src], e) >> p.pk_shift[A.src];
      f &= (1u << p.pk_bits[A.src]) - 1u;
      atomicMax(&aux_lds[aux_off + (k << A.log2m) + (f & ((1u << A.log2m) - 1u))], f >> A.log2m);
      aux_off += slots << A.log2m;
    }
  }
}

// The common accumulator shapes — COUNT(*) and SUM / MIN / MAX over raw INT fields, no auxiliary states — with the ops' descriptors read
// once into scalar registers: p2_consume's walk over p.ops / p.srcs / pk_* costs ~63 scalar instructions and 8 scalar loads per wavefront
// and tuple (profiles/r04_ab_*: the pass was SALU-bound at 1.4 TB/s of tuples).
#define PG_P2_SIMPLE_OPS 4
#define PG_P2_SIMPLE_DOCID 4   // P2SimpleOp::fn of the MIN(docId) accumulator of numGroupsLimit trimming
// COUNT(*) next to SUM over a raw INT field share ONE 64-bit LDS atomic per tuple (round 6): the slot of the SUM row holds count << S | sum of the
// fields (S = 64 - bits of the work item's tuple count; chosen per work item, only where 2 x count bits + field bits <= 64), unpacked into the two
// rows when the work item's table leaves LDS.  The pass is bound by its LDS atomics — random local keys: 8-9 lanes per bank pair against the 4 of a
// conflict-free instruction, 20 cycles per wave-level atomic (the same stream without them: 149 us against 357 us on the 40 k-group row)

struct P2SimpleOp { int32_t fn, plane; uint32_t shift, mask, bias, step; int32_t vt; const GAS uint8_t* dict; };   // vt: 0 raw INT field (value = field x step + bias: a raw column minus its minimum, or the dictId of an arithmetic INT dictionary), 1 / 2 dictId of an INT / LONG dictionary
template <int T, bool GATHER>
DEVFN void p2_consume_simple(const P2SimpleOp (&so)[PG_P2_SIMPLE_OPS], int n_ops, const u32x4 (&cur)[T], bool on, int64_t* table, uint32_t slots, uint32_t local_mask, uint32_t pack_shift,
                             int pk_cnt, int pk_sum, bool narrow) {
  if (!on) return;
#ifdef PG_P2_AGG_NO_ATOMICS   // measurement variant (wrong results): the stream of the aggregation pass alone
  if ((p2_pick<T>(cur, 0, 0) ^ p2_pick<T>(cur, 0, 1) ^ p2_pick<T>(cur, 0, 2) ^ p2_pick<T>(cur, 0, 3)) == 0x12345u) table[0] = 1;
  return;
#endif
  // dictionary look-ups of the lane's four tuples first, all in flight (padding tuples look up dictId 0), the LDS atomics after
  int64_t gv[4][PG_P2_SIMPLE_OPS];
  if (GATHER) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t d0 = p2_pick<T>(cur, 0, e);
#pragma unroll
      for (int o = 0; o < PG_P2_SIMPLE_OPS; o++) {
        gv[e][o] = 0;
        if (o < n_ops && so[o].vt != 0) {
          const uint32_t f = d0 == PG_RADIX_INVALID_KEY ? 0u : ((T == 1 ? d0 : p2_pick<T>(cur, so[o].plane, e)) >> so[o].shift) & so[o].mask;
          if (so[o].vt == 1) gv[e][o] = (int64_t)(int32_t)((const GAS uint32_t*)so[o].dict)[f];
          else gv[e][o] = (int64_t)((const GAS uint64_t*)so[o].dict)[f];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const uint32_t d0 = p2_pick<T>(cur, 0, e);
    if (d0 == PG_RADIX_INVALID_KEY) continue;
    const uint32_t k = d0 & local_mask;
#pragma unroll
    for (int o = 0; o < PG_P2_SIMPLE_OPS; o++) {
      if (o >= n_ops) break;
      int64_t* acc = table + (size_t)o * slots + k;
      if (pack_shift && o == pk_cnt) continue;     // (wave-uniform; the accumulators' descriptions themselves are never written: that sent them to scratch memory)
      if (pack_shift && o == pk_sum) {
        const uint32_t fp = ((T == 1 ? d0 : p2_pick<T>(cur, so[o].plane, e)) >> so[o].shift) & so[o].mask;
        atomicAdd(reinterpret_cast<unsigned long long*>(acc), (1ULL << pack_shift) + (unsigned long long)fp);
        continue;
      }
      if (so[o].fn == PG_ACC_COUNT) { atomicAdd(reinterpret_cast<uint32_t*>(acc), 1u); continue; }   // (a work item sees < 2^32 tuples)
      if (so[o].fn == PG_P2_SIMPLE_DOCID) { atomicMin(reinterpret_cast<long long*>(acc), (long long)p2_pick<T>(cur, so[o].plane, e)); continue; }   // MIN(docId)
      const uint32_t f = ((T == 1 ? d0 : p2_pick<T>(cur, so[o].plane, e)) >> so[o].shift) & so[o].mask;
      if (!GATHER && narrow && so[o].vt == 0 && so[o].mask != 0xFFFFFFFFu && so[o].fn != PG_ACC_SUM) {
        // MIN / MAX of a raw INT field narrower than 32 bits: a 32-bit LDS atomic on the slot's low dword over the FIELD (value - column minimum:
        // the same order) — MAX keeps field + 1 (0 = nothing yet: the low dword of the int64 identity), MIN the field (0xFFFFFFFF likewise);
        // half the bank footprint of the 64-bit atomic, unpacked when the table leaves LDS
        if (so[o].fn == PG_ACC_MIN) atomicMin(reinterpret_cast<uint32_t*>(acc), f);
        else atomicMax(reinterpret_cast<uint32_t*>(acc), f + 1u);
        continue;
      }
      int64_t v = (int64_t)(int32_t)(f * so[o].step + so[o].bias);
      if (GATHER && so[o].vt != 0) v = gv[e][o];
      if (so[o].fn == PG_ACC_SUM) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)v);
      else if (so[o].fn == PG_ACC_MIN) atomicMin(reinterpret_cast<long long*>(acc), (long long)v);
      else atomicMax(reinterpret_cast<long long*>(acc), (long long)v);
    }
  }
}

template <int T, bool GATHER, bool SIMPLE = false>
__device__ __forceinline__ void p2_aggregate_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  int64_t* table = reinterpret_cast<int64_t*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  constexpr uint32_t WAVES = PG_P2_AGG_THREADS / 64;
  const uint32_t slots = 1u << p.radix_shift, local_mask = slots - 1u;
  uint32_t* const aux_lds = reinterpret_cast<uint32_t*>(table + (size_t)p.n_ops * slots);
  uint32_t aux_dwords = 0;
  for (int x = 0; x < p.n_aux; x++) aux_dwords += slots << p.aux[x].log2m;
  uint32_t* const list = aux_lds + aux_dwords;
  const int n_items = p.radix_buckets * p.radix_slices;
  const GAS uint32_t* const tuples = gptr<uint32_t>(p.p2_tuples);
  const size_t plane_stride = (size_t)p.p2_plane_stride;
  P2SimpleOp so[PG_P2_SIMPLE_OPS];
  if (SIMPLE) {
#pragma unroll
    for (int o = 0; o < PG_P2_SIMPLE_OPS; o++) {
      const int oo = o < p.n_ops ? o : 0, src = p.ops[oo].src < 0 ? 0 : p.ops[oo].src;
      const uint32_t bits = (uint32_t)p.pk_bits[src];
      so[o].fn = p.ops[oo].src < 0 ? (p.ops[oo].fn == PG_ACC_COUNT ? PG_ACC_COUNT : PG_P2_SIMPLE_DOCID) : p.ops[oo].fn;
      so[o].plane = p.ops[oo].src < 0 ? p.p2_docid_plane : p.p2_fplane[src];
      so[o].shift = (uint32_t)p.pk_shift[src];
      so[o].mask = bits < 32u ? (1u << bits) - 1u : 0xFFFFFFFFu;
      const bool affine = p.p2_fkind[src] == PG_P2_F_DICTID && p.pk_affine[src] == 3;
      so[o].bias = affine ? (uint32_t)(int32_t)p.pk_base[src] : (uint32_t)p.p2_fbias[src];
      so[o].step = affine ? (uint32_t)p.pk_step[src] : 1u;
      so[o].vt = p.p2_fkind[src] == PG_P2_F_DICTID && !affine ? (p.srcs[src].val_type == PG_V_I32 ? 1 : 2) : 0;
      so[o].dict = gptr<uint8_t>(p.srcs[src].dict);
    }
  }
  // the COUNT(*) + SUM(raw INT field) pair that may share one atomic (see PG_P2_SIMPLE_PACKED)
  int pk_cnt = -1, pk_sum = -1;
  uint32_t pk_field_bits = 0;
  const bool narrow = SIMPLE && !GATHER && !p.p2_no_pack;   // MIN / MAX of narrow raw fields as 32-bit atomics (p2_consume_simple)
  if (SIMPLE && !GATHER && !p.p2_no_pack) {
#pragma unroll
    for (int o = 0; o < PG_P2_SIMPLE_OPS; o++) {   // (unrolled: a run-time index into so[] sends the array to scratch memory — 144 B, 0.94 -> 1.57 ms)
      if (o >= p.n_ops) continue;
      if (so[o].fn == PG_ACC_COUNT && pk_cnt < 0) pk_cnt = o;
      if (so[o].fn == PG_ACC_SUM && so[o].vt == 0 && pk_sum < 0 && so[o].mask != 0xFFFFFFFFu) { pk_sum = o; pk_field_bits = (uint32_t)__popc(so[o].mask); }
    }
    if (pk_cnt < 0 || pk_sum < 0) pk_cnt = pk_sum = -1;
    pk_cnt = uniform(pk_cnt); pk_sum = uniform(pk_sum); pk_field_bits = (uint32_t)uniform((int)pk_field_bits);
  }
  for (int w = (int)blockIdx.x; w < n_items; w += (int)gridDim.x) {
    const uint32_t b = (uint32_t)(w / p.radix_slices), sl = (uint32_t)(w % p.radix_slices);
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = (uint32_t)t; i < slots; i += PG_P2_AGG_THREADS) table[(size_t)o * slots + i] = ident;
    }
    for (uint32_t i = (uint32_t)t; i < aux_dwords; i += PG_P2_AGG_THREADS) aux_lds[i] = 0u;
    const uint32_t bstart = gptr<uint32_t>(p.p2_ctrl)[PG_P2_CTRL_STARTS + b], bend = gptr<uint32_t>(p.p2_ctrl)[PG_P2_CTRL_STARTS + b + 1];
    const uint32_t per = (bend - bstart + (uint32_t)p.radix_slices - 1u) / (uint32_t)p.radix_slices;
    const uint32_t lo_i = bstart + sl * per;
    uint32_t hi_i = lo_i + per;
    if (hi_i > bend) hi_i = bend;
    uint32_t pack_shift = 0;   // 0: not packed in this work item
    if (SIMPLE && pk_sum >= 0) {
      const uint64_t n_max = (uint64_t)(hi_i > lo_i ? hi_i - lo_i : 0u) * (uint64_t)PG_P2_CHUNK;   // tuples of this work item, at most
      uint32_t cbits = 1;
      while (cbits < 40 && (n_max >> cbits) != 0) cbits++;
      if (2u * cbits + pk_field_bits <= 64u) pack_shift = 64u - cbits;
      pack_shift = (uint32_t)uniform((int)pack_shift);
    }
    for (uint32_t win = lo_i; win < hi_i; win += PG_P2_LIST) {
      __syncthreads();   // the table is initialised / the previous window's list is done with
      const uint32_t n_list = hi_i - win < PG_P2_LIST ? hi_i - win : PG_P2_LIST;
      for (uint32_t i = (uint32_t)t; i < n_list; i += PG_P2_AGG_THREADS) list[i] = gptr<uint32_t>(p.p2_list)[win + i];
      __syncthreads();
      // two chunks per wavefront in flight, NO register rotation (a copy c0 = c1 reads the younger load's target: the compiler then
      // waits for every load in flight); the loop is unrolled by two, each half consuming one buffer while the other travels
      // (three chunks per wavefront in flight instead of two were measured for the lean consumer: no change — 318 us against 317 us on the
      // 40 k-group row, profiles/r05_partition_scatter_experiments.txt)
      u32x4 c0[T], c1[T];
      bool on0 = p2_fetch<T>(tuples, plane_stride, list, n_list, (uint32_t)wave, lane, c0), on1 = false;
      for (uint32_t ci = (uint32_t)wave; ci < n_list; ci += 2u * WAVES) {
        on1 = p2_fetch<T>(tuples, plane_stride, list, n_list, ci + WAVES, lane, c1);
        if (SIMPLE) p2_consume_simple<T, GATHER>(so, p.n_ops, c0, on0, table, slots, local_mask, pack_shift, pk_cnt, pk_sum, narrow);
        else p2_consume<T, GATHER>(p, c0, on0, table, aux_lds, slots, local_mask);
        on0 = p2_fetch<T>(tuples, plane_stride, list, n_list, ci + 2u * WAVES, lane, c0);
        if (SIMPLE) p2_consume_simple<T, GATHER>(so, p.n_ops, c1, on1, table, slots, local_mask, pack_shift, pk_cnt, pk_sum, narrow);
        else p2_consume<T, GATHER>(p, c1, on1, table, aux_lds, slots, local_mask);
      }
    }
    __syncthreads();
    if (SIMPLE && pack_shift) {   // unpack: count into the COUNT row, sum of the fields + count x bias into the SUM row
      uint32_t bias_u = 0, step_u = 1;
#pragma unroll
      for (int o = 0; o < PG_P2_SIMPLE_OPS; o++) if (o == pk_sum) { bias_u = so[o].bias; step_u = so[o].step; }
      const int64_t bias = (int64_t)(int32_t)bias_u, step = (int64_t)step_u;
      for (uint32_t i = (uint32_t)t; i < slots; i += PG_P2_AGG_THREADS) {
        const uint64_t v = (uint64_t)table[(size_t)pk_sum * slots + i];
        const int64_t cnt = (int64_t)(v >> pack_shift);
        table[(size_t)pk_cnt * slots + i] = cnt;
        table[(size_t)pk_sum * slots + i] = (int64_t)(v & ((1ULL << pack_shift) - 1ULL)) * step + cnt * bias;
      }
      __syncthreads();
    }
    if (narrow) {   // the 32-bit MIN / MAX rows back to int64 values (identities where nothing arrived)
      bool any = false;
#pragma unroll
      for (int o = 0; o < PG_P2_SIMPLE_OPS; o++) {
        if (o >= p.n_ops || so[o].vt != 0 || so[o].mask == 0xFFFFFFFFu || (so[o].fn != PG_ACC_MIN && so[o].fn != PG_ACC_MAX)) continue;
        any = true;
        const int64_t bias = (int64_t)(int32_t)so[o].bias, step = (int64_t)so[o].step;
        for (uint32_t i = (uint32_t)t; i < slots; i += PG_P2_AGG_THREADS) {
          const uint32_t x = (uint32_t)(uint64_t)table[(size_t)o * slots + i];
          if (so[o].fn == PG_ACC_MIN) table[(size_t)o * slots + i] = x == 0xFFFFFFFFu ? INT64_MAX : (int64_t)x * step + bias;
          else table[(size_t)o * slots + i] = x == 0u ? INT64_MIN : (int64_t)(x - 1u) * step + bias;
        }
      }
      if (any) __syncthreads();
    }
    int64_t* out = p.partials + (int64_t)w * p.n_ops * slots;
    for (int64_t i = t; i < (int64_t)p.n_ops * slots; i += PG_P2_AGG_THREADS) out[i] = table[i];
    {
      uint32_t aux_off = 0;
      for (int x = 0; x < p.n_aux; x++) {   // registers leave as bytes: [slots][2^log2m], merged bytewise by pg_radix_reduce_aux_kernel
        const uint32_t n_words = (slots << p.aux[x].log2m) >> 2;
        const uint32_t* src = aux_lds + aux_off;
        uint32_t* dst = p.aux[x].base + (int64_t)w * n_words;
        for (uint32_t i = (uint32_t)t; i < n_words; i += PG_P2_AGG_THREADS)
          dst[i] = src[4 * i] | (src[4 * i + 1] << 8) | (src[4 * i + 2] << 16) | (src[4 * i + 3] << 24);
        aux_off += slots << p.aux[x].log2m;
      }
    }
    __syncthreads();
  }
}

#define P2_AGGREGATE(NAME, T, GATHER) \
  extern "C" __global__ void __launch_bounds__(PG_P2_AGG_THREADS) NAME(const PgQueryPlan p) { p2_aggregate_body<T, GATHER>(p); }
P2_AGGREGATE(pg_p2_aggregate_1, 1, true)
P2_AGGREGATE(pg_p2_aggregate_2, 2, true)
P2_AGGREGATE(pg_p2_aggregate_3, 3, true)
P2_AGGREGATE(pg_p2_aggregate_4, 4, true)
P2_AGGREGATE(pg_p2_aggregate_1n, 1, false)
P2_AGGREGATE(pg_p2_aggregate_2n, 2, false)
P2_AGGREGATE(pg_p2_aggregate_3n, 3, false)
P2_AGGREGATE(pg_p2_aggregate_4n, 4, false)
#define P2_AGGREGATE_SIMPLE(NAME, T, GATHER) \
  extern "C" __global__ void __launch_bounds__(PG_P2_AGG_THREADS) NAME(const PgQueryPlan p) { p2_aggregate_body<T, GATHER, true>(p); }
P2_AGGREGATE_SIMPLE(pg_p2_aggregate_1s, 1, false)
P2_AGGREGATE_SIMPLE(pg_p2_aggregate_2s, 2, false)
P2_AGGREGATE_SIMPLE(pg_p2_aggregate_1sg, 1, true)
P2_AGGREGATE_SIMPLE(pg_p2_aggregate_2sg, 2, true)

// ---- aggregation pass of the pruned-offer passes (pg_kernels_oct.hip): HyperLogLog offers only (COUNT is kept by pg_oct_p), registers as
// BYTES — 512 groups x 256 registers per bucket instead of 82 as dwords, i.e. 25 buckets instead of 157 for config 5: the stream scatter
// fills whole lines and chunks, the chunk lists are short.  A survivor beat its group's floor but mostly not its own register: the
// register's dword is read first (four tuples per lane in flight), compare-and-swapped only where the rank is higher, and a lost CAS
// (another writer of the dword) is retried serially — rare.
DEVFN void p2_raise_byte(uint32_t* words, uint32_t byte_addr, uint32_t rank) {
  uint32_t* w = words + (byte_addr >> 2);
  const uint32_t sh = (byte_addr & 3u) * 8u;
  uint32_t cur = *reinterpret_cast<volatile uint32_t*>(w);
  while (((cur >> sh) & 0xFFu) < rank) {
    const uint32_t nv = (cur & ~(0xFFu << sh)) | (rank << sh);
    const uint32_t prev = atomicCAS(w, cur, nv);
    if (prev == cur) break;
    cur = prev;
  }
}
DEVFN void p2_consume_bytes(const PgQueryPlan& p, const u32x4 cur, bool on, uint32_t* words, uint32_t local_mask) {
  const uint32_t log2m = (uint32_t)p.aux[0].log2m, imask = (1u << log2m) - 1u, sh0 = (uint32_t)p.pk_shift[0], fmask = (1u << p.pk_bits[0]) - 1u;
  const uint32_t t4[4] = {cur.x, cur.y, cur.z, cur.w};
  uint32_t addr[4], rank[4], w[4];
  uint32_t valid = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const bool v = on && t4[e] != PG_RADIX_INVALID_KEY;
    valid |= (uint32_t)v << e;
    const uint32_t f = (t4[e] >> sh0) & fmask;
    addr[e] = v ? (((t4[e] & local_mask) << log2m) + (f & imask)) : 0u;
    rank[e] = f >> log2m;
  }
  volatile uint32_t* vw = words;
#pragma unroll
  for (int e = 0; e < 4; e++) w[e] = vw[addr[e] >> 2];
  uint32_t need = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) need |= (uint32_t)(((valid >> e) & 1u) && rank[e] > ((w[e] >> ((addr[e] & 3u) * 8u)) & 0xFFu)) << e;
  if (__builtin_amdgcn_ballot_w64(need != 0) == 0) return;   // wave-uniform
  uint32_t prev[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    prev[e] = w[e];
    if ((need >> e) & 1u) {
      const uint32_t sh = (addr[e] & 3u) * 8u;
      prev[e] = atomicCAS(words + (addr[e] >> 2), w[e], (w[e] & ~(0xFFu << sh)) | (rank[e] << sh));
    }
  }
  uint32_t again = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) again |= (uint32_t)(((need >> e) & 1u) && prev[e] != w[e] && ((prev[e] >> ((addr[e] & 3u) * 8u)) & 0xFFu) < rank[e]) << e;
  if (__builtin_amdgcn_ballot_w64(again != 0) == 0) return;
#pragma unroll
  for (int e = 0; e < 4; e++)
    if ((again >> e) & 1u) p2_raise_byte(words, addr[e], rank[e]);
}
// ---- aggregation pass of DISTINCTCOUNT over a dictionary column (round 6): one-plane tuples key | dictId << shift, the bucket's
// dictId sets as bit sets in LDS — [2^radix_shift groups][stride words]; a 2^20-value dictionary's set is 128 KB: one group per bucket — ORed with
// ds_or_b32 (no memory-side atomic anywhere: 16 M first sightings x 24 G/s were >= 0.7 ms on the HBM-resident sets), COUNT(*) accumulators as
// 32-bit adds next to them.  BaseDistinctAggregateAggregationFunction.java:306-345 keeps a RoaringBitmap of dictIds per group.
extern "C" __global__ void __launch_bounds__(PG_P2_AGG_THREADS) pg_p2_aggregate_1set(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  constexpr uint32_t WAVES = PG_P2_AGG_THREADS / 64;
  const uint32_t slots = 1u << p.radix_shift, local_mask = slots - 1u;
  int64_t* const table = reinterpret_cast<int64_t*>(smem);
  const int n_ops = uniform(p.n_ops);
  uint32_t* const words = reinterpret_cast<uint32_t*>(table + (size_t)n_ops * slots);
  const uint32_t stride = (uint32_t)p.aux[0].stride, n_words = slots * stride;
  uint32_t* const list = words + n_words;
  const uint32_t sh0 = (uint32_t)p.pk_shift[p.aux[0].src], fmask = (1u << p.pk_bits[p.aux[0].src]) - 1u;
  const int n_items = p.radix_buckets * p.radix_slices;
  const GAS uint32_t* const tuples = gptr<uint32_t>(p.p2_tuples);
  for (int w = (int)blockIdx.x; w < n_items; w += (int)gridDim.x) {
    const uint32_t b = (uint32_t)(w / p.radix_slices), sl = (uint32_t)(w % p.radix_slices);
    for (uint32_t i = (uint32_t)t; i < (uint32_t)n_ops * slots; i += PG_P2_AGG_THREADS) table[i] = 0;   // COUNTs only (planner)
    for (uint32_t i = (uint32_t)t; i < n_words; i += PG_P2_AGG_THREADS) words[i] = 0u;
    const uint32_t bstart = gptr<uint32_t>(p.p2_ctrl)[PG_P2_CTRL_STARTS + b], bend = gptr<uint32_t>(p.p2_ctrl)[PG_P2_CTRL_STARTS + b + 1];
    const uint32_t per = (bend - bstart + (uint32_t)p.radix_slices - 1u) / (uint32_t)p.radix_slices;
    const uint32_t lo_i = bstart + sl * per;
    uint32_t hi_i = lo_i + per;
    if (hi_i > bend) hi_i = bend;
    // one group per bucket (a 2^20-value dictionary's set fills the LDS): every tuple of the work item counts for the SAME slot — 64 lanes
    // on one address are served one after the other (0.33 of the pass's 0.75 ms per 2 x 10^8 docs): counted in a register, added once
    uint32_t my_count = 0;
    auto consume = [&](const u32x4 cur, bool on) __attribute__((always_inline)) {
      if (!on) return;
      const uint32_t t4[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (t4[e] == PG_RADIX_INVALID_KEY) continue;
        const uint32_t k = t4[e] & local_mask, id = (t4[e] >> sh0) & fmask;
        if (slots == 1u) my_count++;
        else for (int o = 0; o < n_ops; o++) atomicAdd(reinterpret_cast<uint32_t*>(table + (size_t)o * slots + k), 1u);   // (a work item sees < 2^32 tuples)
        atomicOr(&words[k * stride + (id >> 5)], 1u << (id & 31u));
      }
    };
    for (uint32_t win = lo_i; win < hi_i; win += PG_P2_LIST) {
      __syncthreads();   // the sets are zeroed / the previous window's list is done with
      const uint32_t n_list = hi_i - win < PG_P2_LIST ? hi_i - win : PG_P2_LIST;
      for (uint32_t i = (uint32_t)t; i < n_list; i += PG_P2_AGG_THREADS) list[i] = gptr<uint32_t>(p.p2_list)[win + i];
      __syncthreads();
      u32x4 c0[1], c1[1];   // two chunks per wavefront in flight
      bool on0 = p2_fetch<1>(tuples, 0, list, n_list, (uint32_t)wave, lane, c0), on1 = false;
      for (uint32_t ci = (uint32_t)wave; ci < n_list; ci += 2u * WAVES) {
        on1 = p2_fetch<1>(tuples, 0, list, n_list, ci + WAVES, lane, c1);
        consume(c0[0], on0);
        on0 = p2_fetch<1>(tuples, 0, list, n_list, ci + 2u * WAVES, lane, c0);
        consume(c1[0], ci + WAVES < n_list ? on1 : false);
      }
    }
    if (slots == 1u) {
      const uint32_t wsum = wave_sum_u32(my_count);
      if (lane == 0 && wsum) for (int o = 0; o < n_ops; o++) atomicAdd(reinterpret_cast<uint32_t*>(table + o), wsum);
    }
    __syncthreads();
    int64_t* out = p.partials + (int64_t)w * n_ops * slots;
    for (uint32_t i = (uint32_t)t; i < (uint32_t)n_ops * slots; i += PG_P2_AGG_THREADS) out[i] = table[i];
    uint32_t* dst = p.aux[0].base + (int64_t)w * n_words;
    for (uint32_t i = (uint32_t)t; i < n_words; i += PG_P2_AGG_THREADS) dst[i] = words[i];
    __syncthreads();
  }
}
extern "C" __global__ void __launch_bounds__(PG_P2_AGG_THREADS) pg_p2_aggregate_1b(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  constexpr uint32_t WAVES = PG_P2_AGG_THREADS / 64;
  const uint32_t slots = 1u << p.radix_shift, local_mask = slots - 1u;
  uint32_t* const words = reinterpret_cast<uint32_t*>(smem);
  const uint32_t n_words = (slots << p.aux[0].log2m) >> 2;   // one byte per register
  uint32_t* const list = words + n_words;
  const int n_items = p.radix_buckets * p.radix_slices;
  const GAS uint32_t* const tuples = gptr<uint32_t>(p.p2_tuples);
  for (int w = (int)blockIdx.x; w < n_items; w += (int)gridDim.x) {
    const uint32_t b = (uint32_t)(w / p.radix_slices), sl = (uint32_t)(w % p.radix_slices);
    for (uint32_t i = (uint32_t)t; i < n_words; i += PG_P2_AGG_THREADS) words[i] = 0u;
    const uint32_t bstart = gptr<uint32_t>(p.p2_ctrl)[PG_P2_CTRL_STARTS + b], bend = gptr<uint32_t>(p.p2_ctrl)[PG_P2_CTRL_STARTS + b + 1];
    const uint32_t per = (bend - bstart + (uint32_t)p.radix_slices - 1u) / (uint32_t)p.radix_slices;
    const uint32_t lo_i = bstart + sl * per;
    uint32_t hi_i = lo_i + per;
    if (hi_i > bend) hi_i = bend;
    for (uint32_t win = lo_i; win < hi_i; win += PG_P2_LIST) {
      __syncthreads();   // the registers are zeroed / the previous window's list is done with
      const uint32_t n_list = hi_i - win < PG_P2_LIST ? hi_i - win : PG_P2_LIST;
      for (uint32_t i = (uint32_t)t; i < n_list; i += PG_P2_AGG_THREADS) list[i] = gptr<uint32_t>(p.p2_list)[win + i];
      __syncthreads();
      u32x4 c0[1], c1[1];   // two chunks per wavefront in flight, no register rotation (see p2_aggregate_body)
      bool on0 = p2_fetch<1>(tuples, 0, list, n_list, (uint32_t)wave, lane, c0), on1 = false;
      for (uint32_t ci = (uint32_t)wave; ci < n_list; ci += 2u * WAVES) {
        on1 = p2_fetch<1>(tuples, 0, list, n_list, ci + WAVES, lane, c1);
        p2_consume_bytes(p, c0[0], on0, words, local_mask);
        on0 = p2_fetch<1>(tuples, 0, list, n_list, ci + 2u * WAVES, lane, c0);
        p2_consume_bytes(p, c1[0], ci + WAVES < n_list ? on1 : false, words, local_mask);
      }
    }
    __syncthreads();
    uint32_t* dst = p.aux[0].base + (int64_t)w * n_words;
    for (uint32_t i = (uint32_t)t; i < n_words; i += PG_P2_AGG_THREADS) dst[i] = words[i];
    __syncthreads();
  }
}
