// HIP kernels for gfx950 (MI355X / CDNA4): fused filter → (bitmap) → group-by aggregation over one Pinot segment.
//
// Reference semantics reproduced (SURVEY.md §2 kernel inventory):
//   K1/K2 scan leaves        SVScanDocIdIterator + PredicateEvaluator.applySV   (core/operator/dociditerators/SVScanDocIdIterator.java:76-142)
//   K3    posting leaves     ImmutableRoaringBitmap.or / flip / and             (core/operator/filter/InvertedIndexFilterOperator.java:60-96,
//                                                                                core/operator/docidsets/AndDocIdSet.java:127-186)
//   K4    bitmap → docIds    DocIdSetOperator                                   (core/operator/DocIdSetOperator.java:59-86)
//   K5    projection         DataFetcher.readDictIds / readDoubleValues         (core/common/DataFetcher.java:335-386)
//   K6    group keys         DictionaryBasedGroupKeyGenerator raw keys          (core/query/aggregation/groupby/DictionaryBasedGroupKeyGenerator.java:312-354)
//   K7    aggregation        Sum/Count/Min/Max aggregateGroupBySV               (core/query/aggregation/function/SumAggregationFunction.java:160-179 ...)
//
// Hardware mapping: the path is HBM-bound integer/bitmap work (no MFMA).  Persistent 256-thread workgroups walk 16 384-doc
// tiles; every column is read once with loads that are contiguous across the wavefront (16 B/lane for raw columns, dword
// pairs for bit-packed ones) through a wave-uniform tile base (SGPR) + 32-bit lane offset; match bits are assembled with
// DPP row operations into 64-bit words held in LDS; group accumulators live in LDS (ds_add_u64 / ds_max_i64 / ds_add_f64)
// and are flushed once per workgroup.  Register use is kept under 64 VGPRs so that 8 workgroups (32 waves) share a CU:
// occupancy, not instruction-level unrolling, is what hides HBM latency here.
#include <hip/hip_runtime.h>

#include "pg_device.h"

#define DEVFN __device__ __forceinline__
// Pointers that were themselves loaded from memory (plan leaves) have no address space the compiler can infer and would
// be accessed with flat_load; every column / index byte lives in HBM, so say so.
#define GAS __attribute__((address_space(1)))
template <typename T> DEVFN const GAS T* gptr(const void* p) { return (const GAS T*)p; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

DEVFN uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// OR across aligned groups of 8 lanes (two quads) with DPP row operations.
DEVFN uint32_t or_reduce8(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  return v;
}

DEVFN uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

DEVFN uint64_t tile_valid_word(int32_t n_valid, int word) {   // n_valid: docs of this tile below numDocs
  int rem = n_valid - word * 64;
  if (rem >= 64) return ~0ULL;
  if (rem <= 0) return 0ULL;
  return (1ULL << rem) - 1ULL;
}
DEVFN uint32_t quad_valid_nibble(int32_t n_valid, int q) {
  int nv = n_valid - 4 * q;
  return nv >= 4 ? 0xFu : (nv <= 0 ? 0u : ((1u << nv) - 1u));
}

// ---- bit-packed (FixedBitSVForwardIndexReaderV2) extraction ---------------------------------------------------------------
// tw: wave-uniform pointer to the tile's first dword (a tile of 16 384 values starts on a dword boundary for any width);
// q: quad index inside the tile; values 4q .. 4q+3.
template <bool SMALL>
DEVFN void extract4(const GAS uint32_t* __restrict__ tw, uint32_t q, uint32_t bits, uint32_t mask, uint32_t out[4]) {
  if (SMALL) {   // bits <= 8: the four values sit inside one 64-bit window
    uint32_t bitpos = 4u * q * bits;
    uint32_t di = bitpos >> 5, sh = bitpos & 31u;
    uint64_t win = ((uint64_t)bswap32(tw[di]) << 32) | (uint64_t)bswap32(tw[di + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = (uint32_t)(win >> (64u - sh - (uint32_t)(i + 1) * bits)) & mask;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t bitpos = (4u * q + (uint32_t)i) * bits;
      uint32_t di = bitpos >> 5, sh = bitpos & 31u;
      uint64_t win = ((uint64_t)bswap32(tw[di]) << 32) | (uint64_t)bswap32(tw[di + 1]);
      out[i] = (uint32_t)(win >> (64u - sh - bits)) & mask;
    }
  }
}
DEVFN const GAS uint32_t* packed_tile_base(const uint8_t* data, int tile, int bits) {
  return gptr<uint32_t>(data + (size_t)tile * (size_t)(PG_TILE_DOCS / 8) * (size_t)bits);
}

// ---- predicates ---------------------------------------------------------------------------------------------------------------
struct RangeI32 { int32_t lo; uint32_t span; bool empty; };
DEVFN RangeI32 make_range_i32(int64_t lo, int64_t hi) {
  RangeI32 r;
  r.empty = hi < lo;
  r.lo = (int32_t)lo;
  r.span = (uint32_t)(hi - lo);
  return r;
}
DEVFN bool in_range_i32(const RangeI32& r, int32_t v) { return (uint32_t)(v - r.lo) <= r.span; }

DEVFN bool in_set_i64(const PgScanLeaf& L, int64_t v) {
  const GAS int64_t* s = gptr<int64_t>(L.set_values);
  bool hit = false;
#pragma unroll 1
  for (int i = 0; i < L.n_set; i++) hit |= (s[i] == v);
  return hit != (L.exclusive != 0);
}
DEVFN bool in_set_f64(const PgScanLeaf& L, double v) {
  const GAS double* s = gptr<double>(L.set_values);
  bool hit = false;
#pragma unroll 1
  for (int i = 0; i < L.n_set; i++) hit |= (s[i] == v);
  return hit != (L.exclusive != 0);
}

// Predicate over the 4 docs of quad q of the tile → nibble.  KIND selects column layout × value type × predicate form.
enum ScanKind : int {
  SK_DICT_RANGE_SMALL, SK_DICT_RANGE_WIDE, SK_DICT_LUT_SMALL, SK_DICT_LUT_WIDE,
  SK_I32_RANGE, SK_I32_SET, SK_F32_RANGE, SK_F32_SET, SK_I64_RANGE, SK_I64_SET, SK_F64_RANGE, SK_F64_SET
};

template <int KIND>
DEVFN uint32_t eval_quad(const PgScanLeaf& L, const GAS uint8_t* __restrict__ tb, uint32_t q, const RangeI32& r32) {
  uint32_t r = 0;
  if (KIND <= SK_DICT_LUT_WIDE) {
    uint32_t d[4];
    const uint32_t mask = (1u << L.bits) - 1u;
    extract4<(KIND == SK_DICT_RANGE_SMALL || KIND == SK_DICT_LUT_SMALL)>((const GAS uint32_t*)tb, q, (uint32_t)L.bits, mask, d);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bool m = (KIND == SK_DICT_RANGE_SMALL || KIND == SK_DICT_RANGE_WIDE) ? in_range_i32(r32, (int32_t)d[i])
                                                                            : (bool)((gptr<uint32_t>(L.lut)[d[i] >> 5] >> (d[i] & 31u)) & 1u);
      r |= (uint32_t)m << i;
    }
  } else if (KIND <= SK_F32_SET) {
    u32x4 v = *(const GAS u32x4*)(tb + q * 16u);
    uint32_t x[4] = {bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bool m;
      if (KIND == SK_I32_RANGE) m = in_range_i32(r32, (int32_t)x[i]);
      else if (KIND == SK_I32_SET) m = in_set_i64(L, (int64_t)(int32_t)x[i]);
      else if (KIND == SK_F32_RANGE) { double f = (double)__uint_as_float(x[i]); m = f >= __longlong_as_double(L.lo) && f <= __longlong_as_double(L.hi); }
      else m = in_set_f64(L, (double)__uint_as_float(x[i]));
      r |= (uint32_t)m << i;
    }
  } else {
    const GAS u32x4* p = (const GAS u32x4*)(tb + q * 32u);
    u32x4 a = p[0], b = p[1];
    uint64_t x[4] = {((uint64_t)bswap32(a.x) << 32) | bswap32(a.y), ((uint64_t)bswap32(a.z) << 32) | bswap32(a.w),
                     ((uint64_t)bswap32(b.x) << 32) | bswap32(b.y), ((uint64_t)bswap32(b.z) << 32) | bswap32(b.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bool m;
      if (KIND == SK_I64_RANGE) m = (int64_t)x[i] >= L.lo && (int64_t)x[i] <= L.hi;
      else if (KIND == SK_I64_SET) m = in_set_i64(L, (int64_t)x[i]);
      else if (KIND == SK_F64_RANGE) { double f = __longlong_as_double((int64_t)x[i]); m = f >= __longlong_as_double(L.lo) && f <= __longlong_as_double(L.hi); }
      else m = in_set_f64(L, __longlong_as_double((int64_t)x[i]));
      r |= (uint32_t)m << i;
    }
  }
  return r;
}

// Scan leaf over one tile.  MASKED: AND into `words32` in place, evaluating only quads that still have candidates
// (ScanBasedDocIdIterator.applyAnd); returns the number of candidate docs this thread evaluated.
template <int KIND, bool MASKED>
DEVFN uint32_t scan_tile(const PgScanLeaf& L, uint32_t* __restrict__ words32, const GAS uint8_t* __restrict__ tb, int32_t n_valid) {
  const int t = threadIdx.x;
  const uint32_t sh = (uint32_t)(t & 7) * 4u;
  const RangeI32 r32 = make_range_i32(L.lo, L.hi);
  uint32_t n_cand = 0;
  constexpr int U = (KIND == SK_DICT_RANGE_WIDE || KIND == SK_DICT_LUT_WIDE || KIND >= SK_I64_RANGE) ? 1 : 2;
  for (int q0 = t; q0 < PG_TILE_QUADS; q0 += PG_BLOCK * U) {
    uint32_t cand[U], res[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = q0 + u * PG_BLOCK;
      const uint32_t vn = quad_valid_nibble(n_valid, q);
      cand[u] = MASKED ? ((words32[q >> 3] >> sh) & vn) : vn;
      if ((KIND == SK_DICT_RANGE_SMALL || KIND == SK_DICT_RANGE_WIDE || KIND == SK_I32_RANGE) && r32.empty) cand[u] = 0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      res[u] = 0;
      if (cand[u]) res[u] = eval_quad<KIND>(L, tb, (uint32_t)(q0 + u * PG_BLOCK), r32) & cand[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int q = q0 + u * PG_BLOCK;
      if (MASKED) n_cand += __popc(cand[u]);
      const uint32_t x = or_reduce8(res[u] << sh);
      if ((t & 7) == 0) words32[q >> 3] = x;
    }
  }
  return n_cand;
}

template <bool MASKED>
DEVFN uint32_t scan_dispatch(const PgScanLeaf& L, uint32_t* words32, int tile, int32_t n_valid) {
  if (L.col_kind == PG_COL_FIXED_BIT) {
    const GAS uint8_t* tb = (const GAS uint8_t*)packed_tile_base(L.data, tile, L.bits);
    if (L.pred_kind == PG_P_RANGE)
      return L.bits <= 8 ? scan_tile<SK_DICT_RANGE_SMALL, MASKED>(L, words32, tb, n_valid) : scan_tile<SK_DICT_RANGE_WIDE, MASKED>(L, words32, tb, n_valid);
    return L.bits <= 8 ? scan_tile<SK_DICT_LUT_SMALL, MASKED>(L, words32, tb, n_valid) : scan_tile<SK_DICT_LUT_WIDE, MASKED>(L, words32, tb, n_valid);
  }
  if (L.col_kind == PG_COL_RAW32) {
    const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)tile * (PG_TILE_DOCS * 4));
    if (L.val_type == PG_V_I32)
      return L.pred_kind == PG_P_RANGE ? scan_tile<SK_I32_RANGE, MASKED>(L, words32, tb, n_valid) : scan_tile<SK_I32_SET, MASKED>(L, words32, tb, n_valid);
    return L.pred_kind == PG_P_RANGE ? scan_tile<SK_F32_RANGE, MASKED>(L, words32, tb, n_valid) : scan_tile<SK_F32_SET, MASKED>(L, words32, tb, n_valid);
  }
  const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)tile * (PG_TILE_DOCS * 8));
  if (L.val_type == PG_V_I64)
    return L.pred_kind == PG_P_RANGE ? scan_tile<SK_I64_RANGE, MASKED>(L, words32, tb, n_valid) : scan_tile<SK_I64_SET, MASKED>(L, words32, tb, n_valid);
  return L.pred_kind == PG_P_RANGE ? scan_tile<SK_F64_RANGE, MASKED>(L, words32, tb, n_valid) : scan_tile<SK_F64_SET, MASKED>(L, words32, tb, n_valid);
}

// ---- posting leaf: OR the leaf's RoaringBitmap containers that intersect the tile ----------------------------------------
// Container descriptors of the chunk are fetched with ONE vector load (lane e holds entry e) and broadcast with readlane,
// so the dependent chain per leaf is chunk_start → entries → payload regardless of how many postings are OR-ed.
DEVFN void postings_tile(const PgPostingLeaf& L, uint64_t* __restrict__ dst, int tile, uint64_t valid) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int chunk = tile / PG_TILES_PER_CHUNK;
  const int sub = tile % PG_TILES_PER_CHUNK;
  const uint32_t cs = gptr<uint32_t>(L.chunk_start)[chunk], ce = gptr<uint32_t>(L.chunk_start)[chunk + 1];
  uint64_t acc = 0;
  bool scatter = false;
  for (uint32_t base = cs; base < ce; base += 64) {
    const uint32_t n = (ce - base) < 64u ? (ce - base) : 64u;
    u32x4 ent = {0, 0, 0, 0};
    if ((uint32_t)lane < n) ent = gptr<u32x4>(L.entries)[base + lane];
    for (uint32_t e = 0; e < n; e++) {
      const uint32_t off_lo = __builtin_amdgcn_readlane(ent.x, e), off_hi = __builtin_amdgcn_readlane(ent.y, e);
      const uint32_t kt = __builtin_amdgcn_readlane(ent.w, e);   // key | type << 16
      if ((kt >> 16) == 1u) {
        const GAS uint64_t* w = gptr<uint64_t>(L.containers + (((uint64_t)off_hi << 32) | off_lo));
        acc |= w[sub * PG_TILE_WORDS + t];
      } else {
        scatter = true;
      }
    }
  }
  if (scatter) {  // workgroup-uniform: array / run containers set bits with LDS atomics
    dst[t] = acc;
    __syncthreads();
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
    const uint32_t lo = (uint32_t)sub * PG_TILE_DOCS, hi = lo + PG_TILE_DOCS;  // low-16 range of this tile
    for (uint32_t e = cs; e < ce; e++) {
      const u32x4 ce4 = gptr<u32x4>(L.entries)[e];
      PgContainer c;
      c.offset = ((uint64_t)ce4.y << 32) | ce4.x; c.n = ce4.z; c.key = (uint16_t)(ce4.w & 0xFFFF); c.type = (uint16_t)(ce4.w >> 16);
      if (c.type == 0) {
        const GAS uint16_t* vals = gptr<uint16_t>(L.containers + c.offset);
        uint32_t a = 0, b = c.n;  // lower_bound(lo)
        while (a < b) {
          uint32_t m = (a + b) >> 1;
          if (vals[m] < lo) a = m + 1; else b = m;
        }
        for (uint32_t i = a + t; i < c.n; i += PG_BLOCK) {
          uint32_t v = vals[i];
          if (v >= hi) break;
          v -= lo;
          atomicOr(&d32[v >> 5], 1u << (v & 31));
        }
      } else if (c.type == 2) {
        const GAS uint16_t* runs = gptr<uint16_t>(L.containers + c.offset);
        for (uint32_t r = t; r < c.n; r += PG_BLOCK) {
          uint32_t s = runs[2 * r], eend = s + runs[2 * r + 1] + 1;  // [s, eend)
          if (eend <= lo || s >= hi) continue;
          s = (s < lo ? lo : s) - lo;
          eend = (eend > hi ? hi : eend) - lo;
          uint32_t fw = s >> 5, lw = (eend - 1) >> 5;
          uint32_t fm = 0xFFFFFFFFu << (s & 31), lm = 0xFFFFFFFFu >> (31 - ((eend - 1) & 31));
          if (fw == lw) {
            atomicOr(&d32[fw], fm & lm);
          } else {
            atomicOr(&d32[fw], fm);
            for (uint32_t w = fw + 1; w < lw; w++) atomicOr(&d32[w], 0xFFFFFFFFu);
            atomicOr(&d32[lw], lm);
          }
        }
      }
    }
    __syncthreads();
    acc = dst[t];
  }
  dst[t] = L.exclusive ? ((~acc) & valid) : (acc & valid);
}

DEVFN void ranges_tile(const PgRangeLeaf& L, uint64_t* __restrict__ dst, int64_t tile_base, uint64_t valid) {
  const int t = threadIdx.x;
  const int64_t wb = tile_base + (int64_t)t * 64, we = wb + 63;
  const int64_t tile_end = tile_base + PG_TILE_DOCS - 1;
  int a = 0, b = L.n;   // first range whose hi >= tile_base (ranges ascending, disjoint)
  while (a < b) {
    int m = (a + b) >> 1;
    if ((int64_t)gptr<int32_t>(L.hi)[m] < tile_base) a = m + 1; else b = m;
  }
  uint64_t acc = 0;
  for (int r = a; r < L.n; r++) {
    int64_t lo = gptr<int32_t>(L.lo)[r], hi = gptr<int32_t>(L.hi)[r];
    if (lo > tile_end) break;
    if (hi < wb || lo > we) continue;
    int64_t s = lo > wb ? lo - wb : 0, e = hi < we ? hi - wb : 63;
    acc |= (~0ULL << s) & (~0ULL >> (63 - e));
  }
  dst[t] = acc & valid;
}

// ---- accumulator updates ----------------------------------------------------------------------------------------------
DEVFN int64_t f64_order_key(double v) {  // order-preserving map double → int64 (MIN/MAX through integer atomics)
  int64_t b = __double_as_longlong(v);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
DEVFN void acc_int(int64_t* slot, int fn, int64_t v) {
  if (fn == PG_ACC_SUM) atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)v);
  else if (fn == PG_ACC_MIN) atomicMin(reinterpret_cast<long long*>(slot), (long long)v);
  else atomicMax(reinterpret_cast<long long*>(slot), (long long)v);
}
DEVFN void acc_float(int64_t* slot, int fn, double v) {
  if (fn == PG_ACC_SUM) { atomicAdd(reinterpret_cast<double*>(slot), v); return; }
  if (v != v) return;   // Java: NaN > x and NaN < x are false → NaN never replaces the holder
  const int64_t k = f64_order_key(v);
  if (fn == PG_ACC_MIN) atomicMin(reinterpret_cast<long long*>(slot), (long long)k);
  else atomicMax(reinterpret_cast<long long*>(slot), (long long)k);
}

// Aggregates the matching docs of one tile into `table` ([n_ops][G*R] int64 slots, LDS or HBM).
DEVFN void aggregate_tile(const PgQueryPlan& p, const uint32_t* __restrict__ mask32, int tile, int64_t* table) {
  const int t = threadIdx.x;
  const uint32_t sh = (uint32_t)(t & 7) * 4u;
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t stride = (uint32_t)p.n_groups * R;   // slots per op
  const uint32_t rep = (uint32_t)t & (R - 1u);
  for (int q = t; q < PG_TILE_QUADS; q += PG_BLOCK) {
    const uint32_t nib = (mask32[q >> 3] >> sh) & 0xFu;
    if (nib == 0) continue;
    uint32_t slot[4] = {rep, rep, rep, rep};
    for (int g = 0; g < p.n_group_cols; g++) {
      const PgGroupCol& gc = p.gcols[g];
      const GAS uint32_t* tw = packed_tile_base(gc.data, tile, gc.bits);
      const uint32_t mask = (1u << gc.bits) - 1u;
      const uint32_t mult = (uint32_t)gc.mult * R;
      uint32_t d[4];
      if (gc.bits <= 8) extract4<true>(tw, (uint32_t)q, (uint32_t)gc.bits, mask, d);
      else extract4<false>(tw, (uint32_t)q, (uint32_t)gc.bits, mask, d);
#pragma unroll
      for (int i = 0; i < 4; i++) slot[i] += d[i] * mult;
    }
    int o = 0;
    // COUNT ops (src < 0) come first
    for (; o < p.n_ops && p.ops[o].src < 0; o++) {
      int64_t* base = table + (size_t)o * stride;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if ((nib >> i) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[i]), 1ULL);
    }
    while (o < p.n_ops) {
      const int src = p.ops[o].src;
      const PgValueSrc& S = p.srcs[src];
      int o_end = o;
      while (o_end < p.n_ops && p.ops[o_end].src == src) o_end++;
      if (S.col_kind == PG_COL_RAW32 || (S.col_kind == PG_COL_FIXED_BIT && (S.val_type == PG_V_I32 || S.val_type == PG_V_F32))) {
        uint32_t x[4];
        if (S.col_kind == PG_COL_RAW32) {
          u32x4 v = *gptr<u32x4>(S.data + (size_t)tile * (PG_TILE_DOCS * 4) + (uint32_t)q * 16u);
          x[0] = bswap32(v.x); x[1] = bswap32(v.y); x[2] = bswap32(v.z); x[3] = bswap32(v.w);
        } else {
          uint32_t d[4];
          const GAS uint32_t* tw = packed_tile_base(S.data, tile, S.bits);
          const uint32_t mask = (1u << S.bits) - 1u;
          if (S.bits <= 8) extract4<true>(tw, (uint32_t)q, (uint32_t)S.bits, mask, d);
          else extract4<false>(tw, (uint32_t)q, (uint32_t)S.bits, mask, d);
#pragma unroll
          for (int i = 0; i < 4; i++) x[i] = ((nib >> i) & 1u) ? gptr<uint32_t>(S.dict)[d[i]] : 0u;
        }
        for (int k = o; k < o_end; k++) {
          const int fn = p.ops[k].fn;
          int64_t* base = table + (size_t)k * stride;
          if (S.val_type == PG_V_I32) {
#pragma unroll
            for (int i = 0; i < 4; i++) if ((nib >> i) & 1u) acc_int(base + slot[i], fn, (int64_t)(int32_t)x[i]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if ((nib >> i) & 1u) acc_float(base + slot[i], fn, (double)__uint_as_float(x[i]));
          }
        }
      } else {
        uint64_t x[4];
        if (S.col_kind == PG_COL_RAW64) {
          const GAS u32x4* pp = gptr<u32x4>(S.data + (size_t)tile * (PG_TILE_DOCS * 8) + (uint32_t)q * 32u);
          u32x4 a = pp[0], b = pp[1];
          x[0] = ((uint64_t)bswap32(a.x) << 32) | bswap32(a.y); x[1] = ((uint64_t)bswap32(a.z) << 32) | bswap32(a.w);
          x[2] = ((uint64_t)bswap32(b.x) << 32) | bswap32(b.y); x[3] = ((uint64_t)bswap32(b.z) << 32) | bswap32(b.w);
        } else {
          uint32_t d[4];
          const GAS uint32_t* tw = packed_tile_base(S.data, tile, S.bits);
          const uint32_t mask = (1u << S.bits) - 1u;
          if (S.bits <= 8) extract4<true>(tw, (uint32_t)q, (uint32_t)S.bits, mask, d);
          else extract4<false>(tw, (uint32_t)q, (uint32_t)S.bits, mask, d);
#pragma unroll
          for (int i = 0; i < 4; i++) x[i] = ((nib >> i) & 1u) ? gptr<uint64_t>(S.dict)[d[i]] : 0ull;
        }
        for (int k = o; k < o_end; k++) {
          const int fn = p.ops[k].fn;
          int64_t* base = table + (size_t)k * stride;
          if (S.val_type == PG_V_I64) {
#pragma unroll
            for (int i = 0; i < 4; i++) if ((nib >> i) & 1u) acc_int(base + slot[i], fn, (int64_t)x[i]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if ((nib >> i) & 1u) acc_float(base + slot[i], fn, __longlong_as_double((int64_t)x[i]));
          }
        }
      }
      o = o_end;
    }
  }
}

// =====================================================================================================================
// The segment query kernel: persistent workgroups, tile = wg + k * gridDim.
// dynamic LDS: [stack_depth][256] u64 filter stack, then the LDS accumulator table (LDS / SINGLE modes)
// =====================================================================================================================
extern "C" __global__ void __launch_bounds__(PG_BLOCK, PG_WG_PER_CU) pg_segment_query_kernel(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  uint64_t* stack = smem;
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem + (size_t)p.stack_depth * PG_TILE_WORDS);
  const bool lds_agg = (p.agg_mode == PG_AGG_LDS || p.agg_mode == PG_AGG_SINGLE);
  const uint32_t table_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;

  if (t < PG_MAX_STATS) s_stat[t] = 0;
  if (lds_agg) {
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
  }
  __syncthreads();

  uint32_t my_matched = 0;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t tile_base = (int64_t)tile * PG_TILE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - tile_base;
    const int32_t n_valid = rem >= PG_TILE_DOCS ? PG_TILE_DOCS : (int32_t)rem;
    const uint64_t valid = tile_valid_word(n_valid, t);

    // ---- filter program -------------------------------------------------------------------------------------------
    int sp = 0;
    for (int i = 0; i < p.n_instr; i++) {
      const PgFInstr ins = p.instrs[i];
      switch (ins.op) {
        case PG_F_PUSH_POSTINGS:
          postings_tile(p.postings[ins.arg], stack + sp * PG_TILE_WORDS, tile, valid);
          sp++;
          break;
        case PG_F_PUSH_RANGES:
          ranges_tile(p.ranges[ins.arg], stack + sp * PG_TILE_WORDS, tile_base, valid);
          sp++;
          break;
        case PG_F_PUSH_NONE:
          stack[sp * PG_TILE_WORDS + t] = 0;
          sp++;
          break;
        case PG_F_PUSH_SCAN:
          __syncthreads();
          scan_dispatch<false>(p.scans[ins.arg], reinterpret_cast<uint32_t*>(stack + sp * PG_TILE_WORDS), tile, n_valid);
          sp++;
          __syncthreads();
          break;
        case PG_F_AND_SCAN: {
          __syncthreads();
          const PgScanLeaf& L = p.scans[ins.arg];
          uint32_t nc = scan_dispatch<true>(L, reinterpret_cast<uint32_t*>(stack + (sp - 1) * PG_TILE_WORDS), tile, n_valid);
          nc = wave_sum_u32(nc);
          if ((t & 63) == 0 && nc) atomicAdd(&s_stat[L.stat_slot], nc);
          __syncthreads();
          break;
        }
        case PG_F_AND:
          sp--;
          stack[(sp - 1) * PG_TILE_WORDS + t] &= stack[sp * PG_TILE_WORDS + t];
          break;
        case PG_F_OR:
          sp--;
          stack[(sp - 1) * PG_TILE_WORDS + t] |= stack[sp * PG_TILE_WORDS + t];
          break;
        case PG_F_NOT:
          stack[(sp - 1) * PG_TILE_WORDS + t] = (~stack[(sp - 1) * PG_TILE_WORDS + t]) & valid;
          break;
        default: break;
      }
    }
    const uint64_t word = stack[t];
    const uint32_t cnt = (uint32_t)__popcll(word);
    my_matched += cnt;
    if (p.out_words) p.out_words[(int64_t)tile * PG_TILE_WORDS + t] = word;
    if (p.out_tile_counts) {
      uint32_t wsum = wave_sum_u32(cnt);
      if ((t & 63) == 0) atomicAdd(&p.out_tile_counts[tile], wsum);
    }

    // ---- aggregation ----------------------------------------------------------------------------------------------
    if (p.agg_mode != PG_AGG_NONE) {
      __syncthreads();   // stack[0] complete for every quad reader
      if (lds_agg) aggregate_tile(p, reinterpret_cast<const uint32_t*>(stack), tile, lds_table);
      else aggregate_tile(p, reinterpret_cast<const uint32_t*>(stack), tile, p.partials);
    }
    __syncthreads();     // before the next tile overwrites the stack
  }

  // ---- epilogue: statistics and accumulator flush -----------------------------------------------------------------------
  uint32_t wsum = wave_sum_u32(my_matched);
  if ((t & 63) == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);

  if (lds_agg) {
    const int R = p.replicas;
    const int64_t n_out = (int64_t)p.n_ops * p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * n_out;
    for (int64_t i = t; i < n_out; i += PG_BLOCK) {
      const int o = (int)(i / p.n_groups);
      const PgAccOp op = p.ops[o];
      const int64_t* src = lds_table + i * R;   // (o * G + g) * R
      int64_t acc = src[0];
      if (op.fn == PG_ACC_SUM && op.is_float) {
        double d = __longlong_as_double(acc);
        for (int r = 1; r < R; r++) d += __longlong_as_double(src[r]);
        acc = __double_as_longlong(d);
      } else if (op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) {
        for (int r = 1; r < R; r++) acc += src[r];
      } else if (op.fn == PG_ACC_MIN) {
        for (int r = 1; r < R; r++) acc = src[r] < acc ? src[r] : acc;
      } else {
        for (int r = 1; r < R; r++) acc = src[r] > acc ? src[r] : acc;
      }
      out[i] = acc;
    }
  }
}

// Combines the per-workgroup partial tables in workgroup order (deterministic): out[op][g].
extern "C" __global__ void __launch_bounds__(256) pg_reduce_partials_kernel(const int64_t* __restrict__ partials,
                                                                             int64_t* __restrict__ out, int n_wg,
                                                                             int n_ops, int n_groups,
                                                                             const PgAccOp* __restrict__ ops) {
  const int64_t n_out = (int64_t)n_ops * n_groups;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const PgAccOp op = ops[i / n_groups];
  int64_t acc = partials[i];
  if (op.fn == PG_ACC_SUM && op.is_float) {
    double d = __longlong_as_double(acc);
    for (int w = 1; w < n_wg; w++) d += __longlong_as_double(partials[(int64_t)w * n_out + i]);
    acc = __double_as_longlong(d);
  } else if (op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) {
    for (int w = 1; w < n_wg; w++) acc += partials[(int64_t)w * n_out + i];
  } else if (op.fn == PG_ACC_MIN) {
    for (int w = 1; w < n_wg; w++) { int64_t v = partials[(int64_t)w * n_out + i]; acc = v < acc ? v : acc; }
  } else {
    for (int w = 1; w < n_wg; w++) { int64_t v = partials[(int64_t)w * n_out + i]; acc = v > acc ? v : acc; }
  }
  out[i] = acc;
}

extern "C" __global__ void __launch_bounds__(256) pg_fill_i64_kernel(int64_t* dst, int64_t n_per_op, int n_ops,
                                                                      const PgAccOp* __restrict__ ops) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_per_op * n_ops) return;
  const PgAccOp op = ops[i / n_per_op];
  dst[i] = pg_acc_identity(op.fn, op.is_float);
}

// K4: match words → ascending docIds (DocIdSetOperator).  One workgroup per tile; tile_offsets = exclusive prefix of
// the per-tile match counts.
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_expand_docids_kernel(const uint64_t* __restrict__ words,
                                                                                const int64_t* __restrict__ tile_offsets,
                                                                                int32_t* __restrict__ out, int n_tiles) {
  __shared__ uint32_t s_scan[PG_BLOCK];
  const int t = threadIdx.x;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint64_t w = words[(int64_t)tile * PG_TILE_WORDS + t];
    uint32_t c = (uint32_t)__popcll(w);
    s_scan[t] = c;
    __syncthreads();
    for (int off = 1; off < PG_BLOCK; off <<= 1) {   // Hillis–Steele inclusive scan
      uint32_t v = (t >= off) ? s_scan[t - off] : 0;
      __syncthreads();
      s_scan[t] += v;
      __syncthreads();
    }
    int64_t pos = tile_offsets[tile] + (s_scan[t] - c);
    const int32_t base = tile * PG_TILE_DOCS + t * 64;
    while (w) {
      int b = __ffsll((long long)w) - 1;
      out[pos++] = base + b;
      w &= w - 1;
    }
    __syncthreads();
  }
}
