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
// Hardware mapping: the path is HBM-bound integer/bitmap work (no MFMA).  One persistent 1024-thread workgroup per CU;
// each of its 16 wavefronts walks 2 048-doc wave tiles on its own — there is no barrier and no LDS staging in the main
// loop.  The filter program runs on a register stack of match masks in "quad layout": lane L of the wavefront owns the
// quads k*64+L (k = 0..7) of 4 consecutive docs, bit 4k+i of its mask dword is doc 4(k*64+L)+i of the tile.  In that
// layout every forward-index load is contiguous across the wavefront (16 B/lane for raw columns, a dword pair for
// bit-packed ones) and all quads of a lane are fetched before the first is used (up to 8 KB in flight per wavefront).
// Posting bitmaps and docId ranges are produced one dword (32 consecutive docs) per lane and transposed into quad layout
// with 8 ds_bpermute.  Group accumulators live in the workgroup's LDS table (ds_add_u64 / ds_max_i64 / ds_add_f64),
// shared by the 16 wavefronts and flushed once per workgroup.
#include <hip/hip_runtime.h>

#include "pg_device.h"
#include "pg_fixed_point.h"

#define DEVFN __device__ __forceinline__
// Kernels are exported with C linkage.  pg_kernels_dense.hip includes this file with PG_KERNEL = static to reuse the device code in a
// translation unit of its own: the register allocation of the kernels at the 128-VGPR limit turned out to depend on which other
// kernels share their translation unit (adding pg_fast_i32range_d here put 16 bytes of scratch into pg_fast_i32range_a).
#ifndef PG_KERNEL
#define PG_KERNEL extern "C"
#endif
#ifndef PG_FAST_AGG_B
#define PG_FAST_AGG_B 4
#endif
#ifndef PG_SCAN_B
#define PG_SCAN_B 4
#endif
#ifndef PG_WIDE_AGG_B
#define PG_WIDE_AGG_B 2      // quads in flight in the general aggregator inside the 1024-thread kernels (4 spills ~180 VGPRs)
#endif
#ifndef PG_GENERIC_AGG_B
#define PG_GENERIC_AGG_B 4   // quads in flight per lane in the interpreter kernels' aggregation (8 spills ~200 VGPRs)
#endif
// Pointers that were themselves loaded from memory (plan leaves) have no address space the compiler can infer and would
// be accessed with flat_load; every column / index byte lives in HBM, so say so.
#define GAS __attribute__((address_space(1)))
template <typename T> DEVFN const GAS T* gptr(const void* p) { return (const GAS T*)p; }
// Plan descriptors (filter program, leaves) are written by the host before the launch and never by a kernel: reading
// them through the constant address space lets the compiler use scalar loads (s_load → SGPRs, hoistable out of the tile
// loop) instead of a chain of vector loads each waited for with vmcnt(0).
#define CAS __attribute__((address_space(4)))
template <typename T> DEVFN const CAS T* cptr(const T* p) { return (const CAS T*)p; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));   // dword-aligned pair (bit-packed windows)

// Column and posting bytes are streamed once per query: non-temporal loads (measured +5 points of HBM roofline on cfg 3).
template <typename T> DEVFN T ldnt(const GAS T* p) { return __builtin_nontemporal_load(p); }
DEVFN uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
DEVFN int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

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
// fold of an int64 accumulator across the wavefront (fn: PgAccFn; COUNT / SUM add, MIN, MAX) — every lane gets the result
DEVFN int64_t wave_fold_i64(int64_t v, int fn) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off, 64), hi = __shfl_xor((uint32_t)((uint64_t)v >> 32), off, 64);
    const int64_t o = (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
    v = fn == PG_ACC_MIN ? (o < v ? o : v) : (fn == PG_ACC_MAX ? (o > v ? o : v) : v + o);
  }
  return v;
}

// ---- mask layouts ---------------------------------------------------------------------------------------------------------
// linear: lane L holds docs 32L .. 32L+31 of the wave tile (bit i = doc 32L+i) — what bitmap containers deliver
// quad:   lane L holds quads k*64+L, k = 0..7 (bit 4k+i = doc 4(k*64+L)+i)    — what 16 B/lane column loads deliver
DEVFN uint32_t lin_to_quad(uint32_t w, int lane) {
  uint32_t out = 0;
  const uint32_t sh = (uint32_t)(lane & 7) * 4u;
  const int src0 = lane >> 3;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t x = (uint32_t)__shfl((int)w, k * 8 + src0, 64);   // dword holding quads 8(k*8+src0) .. +7
    out |= ((x >> sh) & 0xFu) << (4 * k);
  }
  return out;
}
DEVFN uint32_t quad_to_lin(uint32_t m, int lane) {
  uint32_t out = 0;
  const uint32_t sh = (uint32_t)(lane & 7) * 4u;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t x = ((m >> (4 * k)) & 0xFu) << sh;
    x = or_reduce8(x);                                                 // lanes 8g..8g+7 hold linear dword k*8+g
    const uint32_t z = (uint32_t)__shfl((int)x, (lane & 7) * 8, 64);
    if ((lane >> 3) == k) out = z;
  }
  return out;
}
DEVFN uint32_t valid_lin_mask(int32_t n_valid, int lane) {   // n_valid: docs of this wave tile below numDocs
  const int rem = n_valid - lane * 32;
  return rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
}
DEVFN uint32_t valid_quad_mask(int32_t n_valid, int lane) {
  if (n_valid >= PG_WAVE_DOCS) return 0xFFFFFFFFu;
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int nv = n_valid - 4 * (k * 64 + lane);
    const uint32_t nib = nv >= 4 ? 0xFu : (nv <= 0 ? 0u : ((1u << nv) - 1u));
    m |= nib << (4 * k);
  }
  return m;
}

// ---- predicates -----------------------------------------------------------------------------------------------------------
struct RangeI32 { int32_t lo; uint32_t span; bool empty; };
DEVFN RangeI32 make_range_i32(int64_t lo, int64_t hi) {
  RangeI32 r;
  r.empty = hi < lo;
  r.lo = (int32_t)lo;
  r.span = (uint32_t)(hi - lo);
  return r;
}
DEVFN bool in_range_i32(const RangeI32& r, int32_t v) { return (uint32_t)(v - r.lo) <= r.span; }

template <class LeafT> DEVFN bool in_set_i64(const LeafT& L, int64_t v) {
  const GAS int64_t* s = gptr<int64_t>(L.set_values);
  bool hit = false;
#pragma unroll 1
  for (int i = 0; i < L.n_set; i++) hit |= (s[i] == v);
  return hit != (L.exclusive != 0);
}
template <class LeafT> DEVFN bool in_set_f64(const LeafT& L, double v) {
  const GAS double* s = gptr<double>(L.set_values);
  bool hit = false;
#pragma unroll 1
  for (int i = 0; i < L.n_set; i++) hit |= (s[i] == v);
  return hit != (L.exclusive != 0);
}

// KIND selects column layout × value type × predicate form.
enum ScanKind : int {
  SK_DICT_RANGE_SMALL, SK_DICT_RANGE_WIDE, SK_DICT_LUT_SMALL, SK_DICT_LUT_WIDE,
  SK_I32_RANGE, SK_I32_SET, SK_F32_RANGE, SK_F32_SET, SK_I64_RANGE, SK_I64_SET, SK_F64_RANGE, SK_F64_SET
};
DEVFN constexpr bool sk_small(int k) { return k == SK_DICT_RANGE_SMALL || k == SK_DICT_LUT_SMALL; }
DEVFN constexpr bool sk_dict(int k) { return k <= SK_DICT_LUT_WIDE; }
DEVFN constexpr bool sk_raw32(int k) { return k >= SK_I32_RANGE && k <= SK_F32_SET; }
DEVFN constexpr int sk_words(int k) { return sk_small(k) ? 2 : (sk_raw32(k) ? 4 : 8); }   // dwords fetched per quad

// ---- bit-packed (FixedBitSVForwardIndexReaderV2) access ---------------------------------------------------------------------
// tw: wave-uniform pointer to the wave tile's first dword (2 048 values start on a dword boundary for any width);
// q: quad index inside the wave tile (values 4q .. 4q+3).  SMALL (bits <= 8): the four values sit inside one 64-bit
// window; otherwise one window per value.
template <bool SMALL>
DEVFN void load_packed_quad(const GAS uint32_t* __restrict__ tw, uint32_t q, uint32_t bits, uint32_t* r) {
  if (SMALL) {
    const uint32_t di = __umul24(4u * q, bits) >> 5;   // q < 512, bits <= 8: v_mul_u32_u24 is full rate, v_mul_lo_u32 a quarter
    const u32x2 v = ldnt((const GAS u32x2_a4*)(tw + di));
    r[0] = v.x; r[1] = v.y;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t di = ((4u * q + (uint32_t)i) * bits) >> 5;
      const u32x2 v = ldnt((const GAS u32x2_a4*)(tw + di));
      r[2 * i] = v.x; r[2 * i + 1] = v.y;
    }
  }
}
template <bool SMALL>
DEVFN void decode_packed_quad(const uint32_t* r, uint32_t q, uint32_t bits, uint32_t mask, uint32_t out[4]) {
  if (SMALL) {
    const uint32_t sh = __umul24(4u * q, bits) & 31u;
    const uint64_t win = ((uint64_t)bswap32(r[0]) << 32) | (uint64_t)bswap32(r[1]);
    const uint32_t top = (uint32_t)((win << sh) >> 32);   // one 64-bit shift: the quad's four values now start at bit 31
    out[0] = top >> (32u - bits);
#pragma unroll
    for (int i = 1; i < 4; i++) out[i] = (top >> (32u - (uint32_t)(i + 1) * bits)) & mask;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t sh = ((4u * q + (uint32_t)i) * bits) & 31u;
      const uint64_t win = ((uint64_t)bswap32(r[2 * i]) << 32) | (uint64_t)bswap32(r[2 * i + 1]);
      out[i] = (uint32_t)(win >> (64u - sh - bits)) & mask;
    }
  }
}
DEVFN const GAS uint32_t* packed_wtile_base(const uint8_t* data, int wtile, int bits) {
  return gptr<uint32_t>(data + (size_t)wtile * (size_t)(PG_WAVE_DOCS / 8) * (size_t)bits);
}

template <int KIND, class LeafT>
DEVFN void load_quad(const LeafT& L, const GAS uint8_t* __restrict__ tb, uint32_t q, uint32_t* r) {
  if (sk_dict(KIND)) {
    load_packed_quad<sk_small(KIND)>((const GAS uint32_t*)tb, q, (uint32_t)L.bits, r);
  } else if (sk_raw32(KIND)) {
    const u32x4 v = ldnt((const GAS u32x4*)(tb + q * 16u));
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else {
    const GAS u32x4* p = (const GAS u32x4*)(tb + q * 32u);
    const u32x4 a = ldnt(p), b = ldnt(p + 1);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
  }
}

// Predicate over the 4 docs of a fetched quad → nibble.
template <int KIND, class LeafT>
DEVFN uint32_t test_quad(const LeafT& L, const uint32_t* r, uint32_t q, const RangeI32& r32) {
  uint32_t res = 0;
  if (sk_dict(KIND)) {
    uint32_t d[4];
    decode_packed_quad<sk_small(KIND)>(r, q, (uint32_t)L.bits, (1u << L.bits) - 1u, d);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool m = (KIND == SK_DICT_RANGE_SMALL || KIND == SK_DICT_RANGE_WIDE)
                         ? in_range_i32(r32, (int32_t)d[i])
                         : (bool)((gptr<uint32_t>(L.lut)[d[i] >> 5] >> (d[i] & 31u)) & 1u);
      res |= (uint32_t)m << i;
    }
  } else if (sk_raw32(KIND)) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t x = bswap32(r[i]);
      bool m;
      if (KIND == SK_I32_RANGE) m = in_range_i32(r32, (int32_t)x);
      else if (KIND == SK_I32_SET) m = in_set_i64(L, (int64_t)(int32_t)x);
      else if (KIND == SK_F32_RANGE) { const double f = (double)__uint_as_float(x); m = f >= __longlong_as_double(L.lo) && f <= __longlong_as_double(L.hi); }
      else m = in_set_f64(L, (double)__uint_as_float(x));
      res |= (uint32_t)m << i;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint64_t x = ((uint64_t)bswap32(r[2 * i]) << 32) | bswap32(r[2 * i + 1]);
      bool m;
      if (KIND == SK_I64_RANGE) m = (int64_t)x >= L.lo && (int64_t)x <= L.hi;
      else if (KIND == SK_I64_SET) m = in_set_i64(L, (int64_t)x);
      else if (KIND == SK_F64_RANGE) { const double f = __longlong_as_double((int64_t)x); m = f >= __longlong_as_double(L.lo) && f <= __longlong_as_double(L.hi); }
      else m = in_set_f64(L, __longlong_as_double((int64_t)x));
      res |= (uint32_t)m << i;
    }
  }
  return res;
}

// Scan leaf over one wave tile: evaluates the predicate for the docs of `cand` (quad layout) only — the whole tile for a
// pushed scan, the surviving candidates for a restricted one (ScanBasedDocIdIterator.applyAnd).  All quads of a lane
// are fetched before the first is tested; quads without a candidate are neither fetched nor tested.
template <int KIND, class LeafT>
DEVFN uint32_t scan_wtile(const LeafT& L, uint32_t cand, const GAS uint8_t* __restrict__ tb, int lane) {
  constexpr int W = sk_words(KIND);
  // quads in flight per lane: 4 (4 KB per wavefront for a raw INT column).  With 16 wavefronts per CU, 8 in flight measured slower
  // (cfg 3: 79.8 -> 80.5 % of the HBM roofline; the pure-scan probe 6.5 -> 6.95 TB/s, profiles/r02_scan_bw_probe.txt)
  constexpr int B = W <= 4 ? PG_SCAN_B : (PG_SCAN_B > 4 ? 4 : PG_SCAN_B);
  const RangeI32 r32 = make_range_i32(L.lo, L.hi);
  if ((KIND == SK_DICT_RANGE_SMALL || KIND == SK_DICT_RANGE_WIDE || KIND == SK_I32_RANGE) && r32.empty) return 0;
  uint32_t res = 0;
#pragma unroll
  for (int k0 = 0; k0 < 8; k0 += B) {
    // Loads are unconditional (a load under a per-lane branch is waited for inside the branch, which serialises the
    // batch); lanes whose quad has no candidate re-read quad `lane` of the tile's first KB instead, so no cache line is
    // touched that candidates do not need.
    uint32_t r[B][W];
#pragma unroll
    for (int u = 0; u < B; u++) {
      const uint32_t nib = (cand >> (4 * (k0 + u))) & 0xFu;
      load_quad<KIND>(L, tb, nib ? (uint32_t)((k0 + u) * 64 + lane) : 0u, r[u]);
    }
#pragma unroll
    for (int u = 0; u < B; u++) {
      const uint32_t nib = (cand >> (4 * (k0 + u))) & 0xFu;
      const uint32_t q = nib ? (uint32_t)((k0 + u) * 64 + lane) : 0u;
      if (KIND == SK_DICT_LUT_SMALL || KIND == SK_DICT_LUT_WIDE || KIND == SK_I32_SET || KIND == SK_F32_SET || KIND == SK_I64_SET || KIND == SK_F64_SET) {
        if (nib) res |= (test_quad<KIND>(L, r[u], q, r32) & nib) << (4 * (k0 + u));   // these read a LUT / value set
      } else {
        res |= (test_quad<KIND>(L, r[u], q, r32) & nib) << (4 * (k0 + u));
      }
    }
  }
  return res;
}

template <class LeafT>
DEVFN uint32_t scan_dispatch(const LeafT& L, uint32_t cand, int wtile, int lane) {
  if (L.col_kind == PG_COL_FIXED_BIT) {
    const GAS uint8_t* tb = (const GAS uint8_t*)packed_wtile_base(L.data, wtile, L.bits);
    if (L.pred_kind == PG_P_RANGE)
      return L.bits <= 8 ? scan_wtile<SK_DICT_RANGE_SMALL>(L, cand, tb, lane) : scan_wtile<SK_DICT_RANGE_WIDE>(L, cand, tb, lane);
    return L.bits <= 8 ? scan_wtile<SK_DICT_LUT_SMALL>(L, cand, tb, lane) : scan_wtile<SK_DICT_LUT_WIDE>(L, cand, tb, lane);
  }
  if (L.col_kind == PG_COL_RAW32) {
    const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)wtile * (PG_WAVE_DOCS * 4));
    if (L.val_type == PG_V_I32)
      return L.pred_kind == PG_P_RANGE ? scan_wtile<SK_I32_RANGE>(L, cand, tb, lane) : scan_wtile<SK_I32_SET>(L, cand, tb, lane);
    return L.pred_kind == PG_P_RANGE ? scan_wtile<SK_F32_RANGE>(L, cand, tb, lane) : scan_wtile<SK_F32_SET>(L, cand, tb, lane);
  }
  const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)wtile * (PG_WAVE_DOCS * 8));
  if (L.val_type == PG_V_I64)
    return L.pred_kind == PG_P_RANGE ? scan_wtile<SK_I64_RANGE>(L, cand, tb, lane) : scan_wtile<SK_I64_SET>(L, cand, tb, lane);
  return L.pred_kind == PG_P_RANGE ? scan_wtile<SK_F64_RANGE>(L, cand, tb, lane) : scan_wtile<SK_F64_SET>(L, cand, tb, lane);
}

// ---- posting leaf: OR the leaf's RoaringBitmap containers that intersect the wave tile (linear layout) ---------------------
// Container descriptors of the chunk are fetched with ONE vector load (lane e holds entry e) and broadcast with readlane,
// so the dependent chain per leaf is chunk_start → entries → payload regardless of how many postings are OR-ed.
// Array / run containers set their bits in the wavefront's private 2 048-bit LDS scratch.
template <class LeafT>
DEVFN uint32_t postings_wtile(const LeafT& L, int wtile, uint32_t valid_lin, uint32_t* __restrict__ wscratch, int lane) {
  const int chunk = wtile / PG_WTILES_PER_CHUNK;
  const int sub = wtile % PG_WTILES_PER_CHUNK;
  uint32_t acc = 0;
  if (chunk < L.dense_chunks) {   // wave-uniform: dense bitmap postings, address = base + 8 KB * chunk
    const uint32_t di = (uint32_t)wtile * 64u + (uint32_t)lane;   // = chunk * 2048 + sub * 64 + lane
    // unused pointers repeat dense[0] (OR is idempotent): 8 unconditional loads, no branches (skipping the unused slots with
    // wave-uniform branches costs the headline kernel its last free VGPRs: 16 bytes of scratch and 8 points of roofline, measured)
    uint32_t v[PG_MAX_DENSE];
#pragma unroll
    for (int j = 0; j < PG_MAX_DENSE; j++) v[j] = ldnt(gptr<uint32_t>(L.dense[j]) + di);
#pragma unroll
    for (int j = 0; j < PG_MAX_DENSE; j++) acc |= v[j];
  }
  uint32_t cs = 0, ce = 0;
  if (L.has_csr) { cs = gptr<uint32_t>(L.chunk_start)[chunk]; ce = gptr<uint32_t>(L.chunk_start)[chunk + 1]; }
  bool scatter = false;
  for (uint32_t base = cs; base < ce; base += 64) {
    const uint32_t n = (ce - base) < 64u ? (ce - base) : 64u;
    u32x4 ent = {0, 0, 0, 0};
    if ((uint32_t)lane < n) ent = gptr<u32x4>(L.entries)[base + lane];
    for (uint32_t e = 0; e < n; e++) {
      const uint32_t off_lo = __builtin_amdgcn_readlane(ent.x, e), off_hi = __builtin_amdgcn_readlane(ent.y, e);
      const uint32_t kt = __builtin_amdgcn_readlane(ent.w, e);   // key | type << 16
      if ((kt >> 16) == 1u) {
        const GAS uint32_t* w = gptr<uint32_t>(L.containers + (((uint64_t)off_hi << 32) | off_lo));
        acc |= w[sub * 64 + lane];
      } else {
        scatter = true;
      }
    }
  }
  if (scatter) {  // wave-uniform
    wscratch[lane] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const uint32_t lo = (uint32_t)sub * PG_WAVE_DOCS, hi = lo + PG_WAVE_DOCS;  // low-16 range of this wave tile
    for (uint32_t e = cs; e < ce; e++) {
      const u32x4 ce4 = gptr<u32x4>(L.entries)[e];
      const uint64_t off = ((uint64_t)ce4.y << 32) | ce4.x;
      const uint32_t n = ce4.z, type = ce4.w >> 16;
      if (type == 0) {
        const GAS uint16_t* vals = gptr<uint16_t>(L.containers + off);
        uint32_t a = 0, b = n;  // lower_bound(lo)
        while (a < b) {
          const uint32_t m = (a + b) >> 1;
          if (vals[m] < lo) a = m + 1; else b = m;
        }
        for (uint32_t i = a + lane; i < n; i += 64) {
          uint32_t v = vals[i];
          if (v >= hi) break;
          v -= lo;
          atomicOr(&wscratch[v >> 5], 1u << (v & 31));
        }
      } else if (type == 2) {
        const GAS uint16_t* runs = gptr<uint16_t>(L.containers + off);
        for (uint32_t r = lane; r < n; r += 64) {
          uint32_t s = runs[2 * r], eend = s + runs[2 * r + 1] + 1;  // [s, eend)
          if (eend <= lo || s >= hi) continue;
          s = (s < lo ? lo : s) - lo;
          eend = (eend > hi ? hi : eend) - lo;
          const uint32_t fw = s >> 5, lw = (eend - 1) >> 5;
          const uint32_t fm = 0xFFFFFFFFu << (s & 31), lm = 0xFFFFFFFFu >> (31 - ((eend - 1) & 31));
          if (fw == lw) {
            atomicOr(&wscratch[fw], fm & lm);
          } else {
            atomicOr(&wscratch[fw], fm);
            for (uint32_t w = fw + 1; w < lw; w++) atomicOr(&wscratch[w], 0xFFFFFFFFu);
            atomicOr(&wscratch[lw], lm);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    acc = *(volatile uint32_t*)&wscratch[lane];
  }
  return (L.exclusive ? ~acc : acc) & valid_lin;
}

// One container's bits of the wave tile `sub` (2 048 docs of its 2^16-row chunk) OR-ed into the wavefront's LDS scratch (array / run
// containers; bitmap containers are read directly).  The caller zeroes / fences the scratch.
DEVFN void container_scatter(const GAS uint8_t* payload, uint32_t n, uint32_t type, int sub, uint32_t* __restrict__ wscratch, int lane) {
  const uint32_t lo = (uint32_t)sub * PG_WAVE_DOCS, hi = lo + PG_WAVE_DOCS;
  if (type == 0) {
    const GAS uint16_t* vals = (const GAS uint16_t*)payload;
    uint32_t a = 0, b = n;
    while (a < b) {
      const uint32_t m = (a + b) >> 1;
      if (vals[m] < lo) a = m + 1; else b = m;
    }
    for (uint32_t i = a + lane; i < n; i += 64) {
      uint32_t v = vals[i];
      if (v >= hi) break;
      v -= lo;
      atomicOr(&wscratch[v >> 5], 1u << (v & 31));
    }
  } else if (type == 2) {
    const GAS uint16_t* runs = (const GAS uint16_t*)payload;
    for (uint32_t r = lane; r < n; r += 64) {
      uint32_t s = runs[2 * r], eend = s + runs[2 * r + 1] + 1;
      if (eend <= lo || s >= hi) continue;
      s = (s < lo ? lo : s) - lo;
      eend = (eend > hi ? hi : eend) - lo;
      const uint32_t fw = s >> 5, lw = (eend - 1) >> 5;
      const uint32_t fm = 0xFFFFFFFFu << (s & 31), lm = 0xFFFFFFFFu >> (31 - ((eend - 1) & 31));
      if (fw == lw) atomicOr(&wscratch[fw], fm & lm);
      else {
        atomicOr(&wscratch[fw], fm);
        for (uint32_t w = fw + 1; w < lw; w++) atomicOr(&wscratch[w], 0xFFFFFFFFu);
        atomicOr(&wscratch[lw], lm);
      }
    }
  }
}

// Bit-sliced range index leaf (PgRangeIdxLeaf): the descriptors of the chunk's slices arrive with one vector load (lane s holds slice
// s), each slice's 2 048-bit window is folded into the two running "<= threshold" masks.
template <class LeafT>
DEVFN uint32_t rangeidx_wtile(const LeafT& L, int wtile, uint32_t valid_lin, uint32_t* __restrict__ wscratch, int lane) {
  const int chunk = wtile / PG_WTILES_PER_CHUNK, sub = wtile % PG_WTILES_PER_CHUNK;
  u32x4 ent = {0u, 0u, 0u, 3u << 16};
  if (lane < L.n_slices) ent = gptr<u32x4>(L.descs)[(size_t)chunk * (size_t)L.n_slices + (size_t)lane];
  uint32_t le_hi = 0xFFFFFFFFu, le_lo = 0xFFFFFFFFu;
  for (int s = 0; s < L.n_slices; s++) {
    const uint32_t off_lo = __builtin_amdgcn_readlane(ent.x, s), off_hi = __builtin_amdgcn_readlane(ent.y, s);
    const uint32_t n = __builtin_amdgcn_readlane(ent.z, s), type = __builtin_amdgcn_readlane(ent.w, s) >> 16;
    const GAS uint8_t* payload = gptr<uint8_t>(L.containers + (((uint64_t)off_hi << 32) | off_lo));
    uint32_t S = 0;
    if (type == 1) S = ((const GAS uint32_t*)payload)[sub * 64 + lane];
    else if (type != 3) {   // wave-uniform
      wscratch[lane] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      container_scatter(payload, n, type, sub, wscratch, lane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      S = *(volatile uint32_t*)&wscratch[lane];
    }
    le_hi = ((L.hi >> s) & 1ULL) ? (le_hi | S) : (le_hi & S);
    le_lo = ((L.lo_m1 >> s) & 1ULL) ? (le_lo | S) : (le_lo & S);
  }
  uint32_t r = L.has_hi ? le_hi : 0xFFFFFFFFu;
  if (L.has_lo) r &= ~le_lo;
  return r & valid_lin;
}

// docId ranges (sorted index / match-all), linear layout
template <class LeafT>
DEVFN uint32_t ranges_wtile(const LeafT& L, int64_t wbase, uint32_t valid_lin, int lane) {
  const int64_t wb = wbase + (int64_t)lane * 32, we = wb + 31;
  const int64_t tile_end = wbase + PG_WAVE_DOCS - 1;
  int a = 0, b = L.n;   // first range whose hi >= wbase (ranges ascending, disjoint)
  while (a < b) {
    const int m = (a + b) >> 1;
    if ((int64_t)gptr<int32_t>(L.hi)[m] < wbase) a = m + 1; else b = m;
  }
  uint32_t acc = 0;
  for (int r = a; r < L.n; r++) {
    const int64_t lo = gptr<int32_t>(L.lo)[r], hi = gptr<int32_t>(L.hi)[r];
    if (lo > tile_end) break;
    if (hi < wb || lo > we) continue;
    const int64_t s = lo > wb ? lo - wb : 0, e = hi < we ? hi - wb : 31;
    acc |= (0xFFFFFFFFu << s) & (0xFFFFFFFFu >> (31 - e));
  }
  return acc & valid_lin;
}

// ---- accumulator updates ------------------------------------------------------------------------------------------------
DEVFN int64_t f64_order_key(double v) {  // order-preserving map double → int64 (MIN/MAX through integer atomics)
  const int64_t b = __double_as_longlong(v);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
DEVFN void acc_int(int64_t* slot, int fn, int64_t v) {
  if (fn == PG_ACC_SUM) atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)v);
  else if (fn == PG_ACC_MIN) atomicMin(reinterpret_cast<long long*>(slot), (long long)v);
  else atomicMax(reinterpret_cast<long long*>(slot), (long long)v);
}
DEVFN int64_t fx_digit(double x, int q, int limb) { return pg_fx_digit(x, q, limb); }   // pg_fixed_point.h
DEVFN int64_t long_digit(int64_t v, int limb) { return pg_long_digit(v, limb); }
// one accumulator update from a double / an int64 value, whatever the accumulator's value kind
DEVFN void acc_float(int64_t* slot, int fn, double v);
DEVFN void acc_from_double(int64_t* slot, const PgAccOp& op, double v, int q) {
  if (op.is_float == PG_ACCV_FIXED_DIGIT) atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)fx_digit(v, q, op.limb));
  else acc_float(slot, op.fn, v);
}
DEVFN void acc_from_int(int64_t* slot, const PgAccOp& op, int64_t v) {
  if (op.is_float == PG_ACCV_LONG_DIGIT) atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)long_digit(v, op.limb));
  else acc_int(slot, op.fn, v);
}
DEVFN void acc_float(int64_t* slot, int fn, double v) {
  if (fn == PG_ACC_SUM) { atomicAdd(reinterpret_cast<double*>(slot), v); return; }
  if (v != v) return;   // Java: NaN > x and NaN < x are false → NaN never replaces the holder
  const int64_t k = f64_order_key(v);
  if (fn == PG_ACC_MIN) atomicMin(reinterpret_cast<long long*>(slot), (long long)k);
  else atomicMax(reinterpret_cast<long long*>(slot), (long long)k);
}

// ---- auxiliary accumulators (DISTINCTCOUNT dictId sets, HyperLogLog registers) in HBM ------------------------------------------
// Both are monotone (bits only get set, registers only grow), so an update first reads the current state and skips the
// atomic when it would change nothing — after warm-up almost every doc does (a stale read only costs a redundant atomic).
DEVFN uint32_t murmur_hash_long_dev(int64_t data) {   // stream-lib MurmurHash.hashLong
  const uint32_t m = 0x5bd1e995u;
  uint32_t h = 0;
  uint32_t k = (uint32_t)(uint64_t)data * m;
  k ^= k >> 24;
  h ^= k * m;
  k = (uint32_t)((uint64_t)data >> 32) * m;
  k ^= k >> 24;
  h *= m;
  h ^= k * m;
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}
DEVFN uint32_t hll_index_rank_dev(uint32_t x, int log2m) {   // register index | rank << 16
  const uint32_t j = x >> (32 - log2m);
  const uint32_t w = (x << log2m) | ((1u << (log2m - 1)) + 1u);
  return j | ((uint32_t)(__clz((int)w) + 1) << 16);
}
DEVFN void set_add(uint32_t* words, uint32_t id) {
  uint32_t* w = words + (id >> 5);
  const uint32_t bit = 1u << (id & 31u);
  if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(w, bit);
}
DEVFN uint32_t bytemax4(uint32_t a, uint32_t b) {   // per-byte unsigned max (four HyperLogLog registers)
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t x = (a >> (8 * k)) & 0xFFu, y = (b >> (8 * k)) & 0xFFu;
    r |= (x > y ? x : y) << (8 * k);
  }
  return r;
}
DEVFN void hll_update(uint8_t* regs, uint32_t idx, uint32_t rank) {
  uint32_t* w = reinterpret_cast<uint32_t*>(regs + (idx & ~3u));
  const uint32_t sh = (idx & 3u) * 8u;
  uint32_t cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (((cur >> sh) & 0xFFu) < rank) {
    const uint32_t nv = (cur & ~(0xFFu << sh)) | (rank << sh);
    const uint32_t prev = atomicCAS(w, cur, nv);
    if (prev == cur) break;
    cur = prev;
  }
}

// ---- auxiliary accumulators of one batch of quads (slot[u][i] = table slot of doc i of quad k0+u, mb = the batch's mask) ----
template <int B>
DEVFN void aux_update_batch(const PgQueryPlan& p, uint32_t mb, int k0, int wtile, const uint32_t (&slot_in)[B][4], int lane, uint32_t part_lo) {
  uint32_t slot[B][4];   // the auxiliary regions are indexed by the GLOBAL group (slots are range-local in PG_AGG_LDS_PART)
#pragma unroll
  for (int u = 0; u < B; u++)
#pragma unroll
    for (int i = 0; i < 4; i++) slot[u][i] = slot_in[u][i] + (part_lo << p.replica_shift);
  // ---- auxiliary accumulators: DISTINCTCOUNT dictId sets / HyperLogLog registers (HBM regions) ---------------------------
  // Dictionary sources run in three batched stages over the 4·B docs of the lane — dictIds, (index, rank) look-ups,
  // current state words — so that 4·B gathers are in flight per lane; only docs that would change the state go on to
  // the atomic.  Lanes / docs outside the mask use clamped (valid) addresses and are masked at the atomic.
  for (int xa = 0; xa < p.n_aux; xa++) {
    PgAuxOp A = p.aux[xa];
    if (A.lds_offset >= 0) {   // this workgroup's own state in LDS (generic pointer into the LDS aperture: flat atomics)
      extern __shared__ __attribute__((aligned(16))) uint64_t smem_aux[];
      A.base = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem_aux) + A.lds_offset);
    } else {
      A.base = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(A.base) + (size_t)(blockIdx.x & (uint32_t)(A.n_rep - 1)) * (size_t)A.rep_bytes);
    }
    const PgValueSrc& S = p.srcs[A.src];
    if (A.kind == PG_AUX_HLL_BYTES) {
      // Serialized HyperLogLogs of a star-tree pair column (one byte per register after upload): HyperLogLog#addAll = register-wise
      // max.  One matching doc at a time, the whole wavefront on its registers: lane L merges dwords L, L+64, ... of the doc
      // into the group's registers (read first, CAS only where a register grows).
      const uint32_t n_dw = (uint32_t)A.stride >> 2;
      const GAS uint8_t* src = gptr<uint8_t>(S.data);
#pragma unroll
      for (int u = 0; u < B; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          unsigned long long ball = __ballot((mb >> (4 * u + i)) & 1u);
          while (ball) {
            // four docs in flight (their loads are independent); short of four, the first is repeated — max is idempotent
            const GAS uint32_t* sw[4];
            uint32_t* dst[4];
            int l0 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              int l = l0;
              if (j == 0 || ball) { l = __builtin_ctzll(ball); ball &= ball - 1; }
              if (j == 0) l0 = l;
              const size_t g = (size_t)((uint32_t)__builtin_amdgcn_readlane((int)slot[u][i], l) >> p.replica_shift);
              const int64_t doc = (int64_t)wtile * PG_WAVE_DOCS + 4 * ((k0 + u) * 64 + l) + i;
              sw[j] = (const GAS uint32_t*)(src + doc * (int64_t)A.stride);
              dst[j] = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(A.base) + g * (size_t)A.stride);
            }
            for (uint32_t w = (uint32_t)lane; w < n_dw; w += 64) {
              uint32_t v[4], cur[4];
#pragma unroll
              for (int j = 0; j < 4; j++) v[j] = sw[j][w];
#pragma unroll
              for (int j = 0; j < 4; j++) cur[j] = __hip_atomic_load(dst[j] + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                uint32_t c = cur[j];
                for (;;) {
                  const uint32_t nv = bytemax4(c, v[j]);
                  if (nv == c) break;
                  const uint32_t prev = atomicCAS(dst[j] + w, c, nv);
                  if (prev == c) break;
                  c = prev;
                }
              }
            }
          }
        }
    } else if (S.col_kind == PG_COL_FIXED_BIT) {
      const GAS uint32_t* tw = packed_wtile_base(S.data, wtile, S.bits);
      const uint32_t bits = (uint32_t)S.bits, mask = (1u << S.bits) - 1u;
      uint32_t d[B][4];
      if (bits <= 8) {
        uint32_t r[B][2];
#pragma unroll
        for (int u = 0; u < B; u++) load_packed_quad<true>(tw, ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u, bits, r[u]);
#pragma unroll
        for (int u = 0; u < B; u++) decode_packed_quad<true>(r[u], ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u, bits, mask, d[u]);
      } else {
        uint32_t r[B][8];
#pragma unroll
        for (int u = 0; u < B; u++) load_packed_quad<false>(tw, ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u, bits, r[u]);
#pragma unroll
        for (int u = 0; u < B; u++) decode_packed_quad<false>(r[u], ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u, bits, mask, d[u]);
      }
      uint32_t* wp[B][4];
      uint32_t want[B][4];   // DICT_SET: the bit; HLL: rank << shift-in-word, with the shift in the low 5 bits of `sh`
      uint32_t sh[B][4];
      if (A.kind == PG_AUX_DICT_SET) {
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const size_t g = (size_t)(slot[u][i] >> p.replica_shift);
            wp[u][i] = A.base + g * (size_t)A.stride + (d[u][i] >> 5);
            want[u][i] = 1u << (d[u][i] & 31u);
            sh[u][i] = 0;
          }
      } else {
        uint32_t e[B][4];
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) e[u][i] = gptr<uint32_t>(A.lut)[d[u][i]];
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const size_t g = (size_t)(slot[u][i] >> p.replica_shift);
            const uint32_t idx = e[u][i] & 0xFFFFu;
            wp[u][i] = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(A.base) + g * (size_t)A.stride + (idx & ~3u));
            sh[u][i] = (idx & 3u) * 8u;
            want[u][i] = e[u][i] >> 16;
          }
      }
      uint32_t cur[B][4];
#pragma unroll
      for (int u = 0; u < B; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) cur[u][i] = __hip_atomic_load(wp[u][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < B; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if ((mb >> (4 * u + i)) & 1u) {
            if (A.kind == PG_AUX_DICT_SET) {
              if (!(cur[u][i] & want[u][i])) atomicOr(wp[u][i], want[u][i]);
            } else {
              uint32_t c = cur[u][i];
              while (((c >> sh[u][i]) & 0xFFu) < want[u][i]) {
                const uint32_t nv = (c & ~(0xFFu << sh[u][i])) | (want[u][i] << sh[u][i]);
                const uint32_t prev = atomicCAS(wp[u][i], c, nv);
                if (prev == c) break;
                c = prev;
              }
            }
          }
        }
    } else {   // raw column: DISTINCTCOUNTHLL hashes the value on the fly
#pragma unroll
      for (int u = 0; u < B; u++) {
        const uint32_t nib = (mb >> (4 * u)) & 0xFu;
        if (nib) {
          const uint32_t q = (uint32_t)((k0 + u) * 64 + lane);
          int64_t v[4];   // the long the value hashes as (Integer/Long value, Float raw int bits, Double raw long bits)
          if (S.col_kind == PG_COL_RAW32) {
            const u32x4 x = *gptr<u32x4>(S.data + (size_t)wtile * (PG_WAVE_DOCS * 4) + q * 16u);
            v[0] = (int64_t)(int32_t)bswap32(x.x); v[1] = (int64_t)(int32_t)bswap32(x.y);
            v[2] = (int64_t)(int32_t)bswap32(x.z); v[3] = (int64_t)(int32_t)bswap32(x.w);
          } else {
            const GAS u32x4* pp = gptr<u32x4>(S.data + (size_t)wtile * (PG_WAVE_DOCS * 8) + q * 32u);
            const u32x4 a = pp[0], b = pp[1];
            v[0] = (int64_t)(((uint64_t)bswap32(a.x) << 32) | bswap32(a.y)); v[1] = (int64_t)(((uint64_t)bswap32(a.z) << 32) | bswap32(a.w));
            v[2] = (int64_t)(((uint64_t)bswap32(b.x) << 32) | bswap32(b.y)); v[3] = (int64_t)(((uint64_t)bswap32(b.z) << 32) | bswap32(b.w));
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {
            if ((nib >> i) & 1u) {
              const size_t g = (size_t)(slot[u][i] >> p.replica_shift);
              const uint32_t e = hll_index_rank_dev(murmur_hash_long_dev(v[i]), A.log2m);
              hll_update(reinterpret_cast<uint8_t*>(A.base) + g * (size_t)A.stride, e & 0xFFFFu, e >> 16);
            }
          }
        }
      }
    }
  }

}

// Aggregates the matching docs (quad-layout mask m) of one wave tile into `table` ([n_ops][G*R] int64 slots, LDS or HBM):
// group columns of any width, raw 32/64-bit and dictionary-encoded sources.  B quads per lane are in flight at a time; like
// the filter stage, every load of a stage is issued unconditionally (lanes whose quad has no match re-read quad 0 of the
// tile) before the first is used — a load under a per-lane branch is waited for inside the branch.
// DIG: the plan has digit accumulators (exact SUMs of FLOAT / DOUBLE / wide LONG sources, PgAccValueKind); kernels without them are
// compiled without that code (it costs the others ~40 spilled VGPRs).
template <int B, bool AUX = true, bool DIG = false>
DEVFN void aggregate_wtile(const PgQueryPlan& p, uint32_t m, int wtile, int64_t* table, int lane, uint32_t rep, uint32_t part_lo = 0) {
  const uint32_t R = (uint32_t)p.replicas;
  const bool part = p.agg_mode == PG_AGG_LDS_PART;
  const uint32_t stride = part ? (uint32_t)p.part_groups : (uint32_t)p.n_groups * R;   // slots per op
#pragma unroll
  for (int k0 = 0; k0 < 8; k0 += B) {
    uint32_t mb = B == 8 ? m : ((m >> (4 * k0)) & ((1u << (4 * (B & 7))) - 1u));
    if (__ballot(mb != 0) == 0) continue;   // wave-uniform: the HLL-bytes merge needs every lane of the wavefront
    uint32_t qi[B];                         // quad index of each in-flight quad, clamped to 0 where the lane has no match
#pragma unroll
    for (int u = 0; u < B; u++) qi[u] = ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u;
    uint32_t slot[B][4];
#pragma unroll
    for (int u = 0; u < B; u++)
#pragma unroll
      for (int i = 0; i < 4; i++) slot[u][i] = rep;
    // Up to PG_BATCH_GCOLS narrow group columns are requested together (one memory round trip for the whole key instead of one
    // per column: with few bytes per load this stage is latency-, not bandwidth-bound), together with the first raw 32-bit
    // source.  Wider / further columns follow one at a time.
    constexpr int PG_BATCH_GCOLS = 4;
    int n_batched = 0;
    while (n_batched < p.n_group_cols && n_batched < PG_BATCH_GCOLS && p.gcols[n_batched].bits <= 8) n_batched++;
    int o_first = 0;
    while (o_first < p.n_ops && p.ops[o_first].src < 0) o_first++;
    const bool prefetch0 = o_first < p.n_ops && p.srcs[p.ops[o_first].src].col_kind == PG_COL_RAW32;
    uint32_t x0[B][4];
    {
      uint32_t rg[PG_BATCH_GCOLS][B][2];
#pragma unroll
      for (int g = 0; g < PG_BATCH_GCOLS; g++)
        if (g < n_batched) {
          const GAS uint32_t* tw = packed_wtile_base(p.gcols[g].data, wtile, p.gcols[g].bits);
#pragma unroll
          for (int u = 0; u < B; u++) load_packed_quad<true>(tw, qi[u], (uint32_t)p.gcols[g].bits, rg[g][u]);
        }
      if (prefetch0) {
        const GAS uint8_t* tb = gptr<uint8_t>(p.srcs[p.ops[o_first].src].data + (size_t)wtile * (PG_WAVE_DOCS * 4));
#pragma unroll
        for (int u = 0; u < B; u++) {
          const u32x4 v = ldnt((const GAS u32x4*)(tb + qi[u] * 16u));
          x0[u][0] = v.x; x0[u][1] = v.y; x0[u][2] = v.z; x0[u][3] = v.w;
        }
      }
#pragma unroll
      for (int g = 0; g < PG_BATCH_GCOLS; g++)
        if (g < n_batched) {
          const uint32_t bits = (uint32_t)p.gcols[g].bits, mask = (1u << bits) - 1u, mult = (uint32_t)p.gcols[g].mult * R;
#pragma unroll
          for (int u = 0; u < B; u++) {
            uint32_t d[4];
            decode_packed_quad<true>(rg[g][u], qi[u], bits, mask, d);
#pragma unroll
            for (int i = 0; i < 4; i++) slot[u][i] += d[i] * mult;
          }
        }
    }
    for (int g = n_batched; g < p.n_group_cols; g++) {
      const PgGroupCol& gc = p.gcols[g];
      const GAS uint32_t* tw = packed_wtile_base(gc.data, wtile, gc.bits);
      const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
      const uint32_t mult = (uint32_t)gc.mult * R;
      if (bits <= 8) {
        uint32_t r[B][2];
#pragma unroll
        for (int u = 0; u < B; u++) load_packed_quad<true>(tw, qi[u], bits, r[u]);
#pragma unroll
        for (int u = 0; u < B; u++) {
          uint32_t d[4];
          decode_packed_quad<true>(r[u], qi[u], bits, mask, d);
#pragma unroll
          for (int i = 0; i < 4; i++) slot[u][i] += d[i] * mult;
        }
      } else {
        uint32_t r[B][8];
#pragma unroll
        for (int u = 0; u < B; u++) load_packed_quad<false>(tw, qi[u], bits, r[u]);
#pragma unroll
        for (int u = 0; u < B; u++) {
          uint32_t d[4];
          decode_packed_quad<false>(r[u], qi[u], bits, mask, d);
#pragma unroll
          for (int i = 0; i < 4; i++) slot[u][i] += d[i] * mult;
        }
      }
    }
    if (part) {   // keep only the docs whose key falls in this workgroup's range; slots become range-local
      uint32_t keep = 0;
#pragma unroll
      for (int u = 0; u < B; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          slot[u][i] -= part_lo;
          keep |= (uint32_t)(slot[u][i] < (uint32_t)p.part_groups) << (4 * u + i);
        }
      mb &= keep;
      if (__ballot(mb != 0) == 0) continue;
#pragma unroll
      for (int u = 0; u < B; u++) qi[u] = ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u;
    }
    int o = 0;
    // ops without a source column (src < 0) come first: COUNT, and MIN(docId) when numGroupsLimit can bite
    for (; o < p.n_ops && p.ops[o].src < 0; o++) {
      int64_t* base = table + (size_t)o * stride;
      if (p.ops[o].fn == PG_ACC_COUNT) {
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mb >> (4 * u + i)) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[u][i]), 1ULL);
      } else {
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mb >> (4 * u + i)) & 1u)
              atomicMin(reinterpret_cast<long long*>(base + slot[u][i]),
                        (long long)wtile * PG_WAVE_DOCS + 4 * ((k0 + u) * 64 + lane) + i);
      }
    }
    while (o < p.n_ops) {
      const int src = p.ops[o].src;
      const PgValueSrc& S = p.srcs[src];
      int o_end = o;
      while (o_end < p.n_ops && p.ops[o_end].src == src) o_end++;
      const bool wide_val = S.val_type == PG_V_I64 || S.val_type == PG_V_F64;
      if (!wide_val) {
        // ---- 32-bit values: raw big-endian INT / FLOAT, or dictionary-encoded ------------------------------------------------
        uint32_t x[B][4];
        if (S.col_kind == PG_COL_RAW32) {
          if (prefetch0 && o == o_first) {   // requested together with the group columns (docs later found out of range included)
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++) x[u][i] = x0[u][i];
          } else {
            const GAS uint8_t* tb = gptr<uint8_t>(S.data + (size_t)wtile * (PG_WAVE_DOCS * 4));
#pragma unroll
            for (int u = 0; u < B; u++) {
              const u32x4 v = ldnt((const GAS u32x4*)(tb + qi[u] * 16u));
              x[u][0] = v.x; x[u][1] = v.y; x[u][2] = v.z; x[u][3] = v.w;
            }
          }
#pragma unroll
          for (int u = 0; u < B; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) x[u][i] = bswap32(x[u][i]);
        } else {
          const GAS uint32_t* tw = packed_wtile_base(S.data, wtile, S.bits);
          const uint32_t bits = (uint32_t)S.bits, mask = (1u << S.bits) - 1u;
          uint32_t d[B][4];
          if (bits <= 8) {
            uint32_t r[B][2];
#pragma unroll
            for (int u = 0; u < B; u++) load_packed_quad<true>(tw, qi[u], bits, r[u]);
#pragma unroll
            for (int u = 0; u < B; u++) decode_packed_quad<true>(r[u], qi[u], bits, mask, d[u]);
          } else {
            uint32_t r[B][8];
#pragma unroll
            for (int u = 0; u < B; u++) load_packed_quad<false>(tw, qi[u], bits, r[u]);
#pragma unroll
            for (int u = 0; u < B; u++) decode_packed_quad<false>(r[u], qi[u], bits, mask, d[u]);
          }
#pragma unroll
          for (int u = 0; u < B; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) x[u][i] = gptr<uint32_t>(S.dict)[d[u][i]];   // every dictId read is a valid one
        }
        for (int k = o; k < o_end; k++) {
          const PgAccOp opk = p.ops[k];
          const int fn = opk.fn;
          int64_t* base = table + (size_t)k * stride;
          if (S.val_type == PG_V_I32) {
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++)
                if ((mb >> (4 * u + i)) & 1u) acc_int(base + slot[u][i], fn, (int64_t)(int32_t)x[u][i]);
          } else if (DIG && opk.is_float == PG_ACCV_FIXED_DIGIT) {   // wave-uniform: the branch stays outside the unrolled loops
            const int q = S.fx_q, limb = opk.limb;
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++)
                if ((mb >> (4 * u + i)) & 1u)
                  atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[u][i]), (unsigned long long)fx_digit((double)__uint_as_float(x[u][i]), q, limb));
          } else {
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++)
                if ((mb >> (4 * u + i)) & 1u) acc_float(base + slot[u][i], fn, (double)__uint_as_float(x[u][i]));
          }
        }
      } else {
        // ---- 64-bit values: raw big-endian LONG / DOUBLE (32 B per quad), or dictionary-encoded; four quads at a time -------------
        constexpr int WB = B < 4 ? B : 4;
#pragma unroll
        for (int h = 0; h < B; h += WB) {
          if (__ballot(((mb >> (4 * h)) & ((1u << (4 * WB)) - 1u)) != 0) == 0) continue;
          uint64_t x[WB][4];
          if (S.col_kind == PG_COL_RAW64) {
            const GAS uint8_t* tb = gptr<uint8_t>(S.data + (size_t)wtile * (PG_WAVE_DOCS * 8));
            u32x4 a[WB], b[WB];
#pragma unroll
            for (int u = 0; u < WB; u++) {
              const GAS u32x4* pp = (const GAS u32x4*)(tb + qi[h + u] * 32u);
              a[u] = ldnt(pp);
              b[u] = ldnt(pp + 1);
            }
#pragma unroll
            for (int u = 0; u < WB; u++) {
              x[u][0] = ((uint64_t)bswap32(a[u].x) << 32) | bswap32(a[u].y); x[u][1] = ((uint64_t)bswap32(a[u].z) << 32) | bswap32(a[u].w);
              x[u][2] = ((uint64_t)bswap32(b[u].x) << 32) | bswap32(b[u].y); x[u][3] = ((uint64_t)bswap32(b[u].z) << 32) | bswap32(b[u].w);
            }
          } else {
            const GAS uint32_t* tw = packed_wtile_base(S.data, wtile, S.bits);
            const uint32_t bits = (uint32_t)S.bits, mask = (1u << S.bits) - 1u;
            uint32_t d[WB][4];
            if (bits <= 8) {
              uint32_t r[WB][2];
#pragma unroll
              for (int u = 0; u < WB; u++) load_packed_quad<true>(tw, qi[h + u], bits, r[u]);
#pragma unroll
              for (int u = 0; u < WB; u++) decode_packed_quad<true>(r[u], qi[h + u], bits, mask, d[u]);
            } else {
              uint32_t r[WB][8];
#pragma unroll
              for (int u = 0; u < WB; u++) load_packed_quad<false>(tw, qi[h + u], bits, r[u]);
#pragma unroll
              for (int u = 0; u < WB; u++) decode_packed_quad<false>(r[u], qi[h + u], bits, mask, d[u]);
            }
#pragma unroll
            for (int u = 0; u < WB; u++)
#pragma unroll
              for (int i = 0; i < 4; i++) x[u][i] = gptr<uint64_t>(S.dict)[d[u][i]];
          }
          for (int k = o; k < o_end; k++) {
            const PgAccOp opk = p.ops[k];
            int64_t* base = table + (size_t)k * stride;
            if (DIG && opk.is_float == PG_ACCV_LONG_DIGIT) {   // the value-kind branches are wave-uniform and stay outside the unrolled loops
              const int limb = opk.limb;
#pragma unroll
              for (int u = 0; u < WB; u++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                  if ((mb >> (4 * (h + u) + i)) & 1u)
                    atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[h + u][i]), (unsigned long long)long_digit((int64_t)x[u][i], limb));
            } else if (DIG && opk.is_float == PG_ACCV_FIXED_DIGIT) {
              const int q = S.fx_q, limb = opk.limb;
#pragma unroll
              for (int u = 0; u < WB; u++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                  if ((mb >> (4 * (h + u) + i)) & 1u)
                    atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[h + u][i]), (unsigned long long)fx_digit(__longlong_as_double((int64_t)x[u][i]), q, limb));
            } else if (S.val_type == PG_V_I64) {
#pragma unroll
              for (int u = 0; u < WB; u++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                  if ((mb >> (4 * (h + u) + i)) & 1u) acc_int(base + slot[h + u][i], opk.fn, (int64_t)x[u][i]);
            } else {
#pragma unroll
              for (int u = 0; u < WB; u++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                  if ((mb >> (4 * (h + u) + i)) & 1u) acc_float(base + slot[h + u][i], opk.fn, __longlong_as_double((int64_t)x[u][i]));
            }
          }
        }
      }
      o = o_end;
    }
    if (AUX && p.n_aux > 0) {   // four quads at a time: the three-stage gathers keep 16 addresses + states per lane in registers
      constexpr int AB = B < 4 ? B : 4;
#pragma unroll
      for (int h = 0; h < B; h += AB)
        aux_update_batch<AB>(p, (mb >> (4 * h)) & ((1u << (4 * AB)) - 1u), k0 + h, wtile, reinterpret_cast<const uint32_t(&)[AB][4]>(slot[h]), lane, part_lo);
    }
  }
}

// =====================================================================================================================
// Fast path.  Plans of the shape  [index-only program]  AND  [at most one scan leaf]  →  [LDS-table aggregation over
// 32-bit sources and <= 8-bit group columns]  run in a kernel specialised at compile time on the scan kind and on
// whether there is an aggregation table, so that the per-tile code is straight-line: leaf descriptors are hoisted into
// SGPRs, every load of a stage is issued before the first use, no statistics atomics inside the loop.  Everything else
// takes the interpreter kernel below.
// =====================================================================================================================
struct LinStack {
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
  DEVFN void push(uint32_t v) { s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v; }
  DEVFN void drop() { s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; }
};

// index-only filter program in linear layout (one dword = 32 consecutive docs per lane)
template <bool WORDS>   // WORDS: also understands PG_F_PUSH_WORDS (interpreter kernels only: the specialised kernels sit at 128 VGPRs)
DEVFN uint32_t index_program_lin(const PgQueryPlan& p, int n_instr, int wt, int64_t wbase, uint32_t valid_l, uint32_t* wscratch, int lane) {
  LinStack st;
  for (int i = 0; i < n_instr; i++) {
    const int op = cptr(p.instrs)[i].op, arg = cptr(p.instrs)[i].arg;
    switch (op) {
      case PG_F_PUSH_POSTINGS: st.push(postings_wtile(cptr(p.postings)[arg], wt, valid_l, wscratch, lane)); break;
      case PG_F_PUSH_RANGES: st.push(ranges_wtile(cptr(p.ranges)[arg], wbase, valid_l, lane)); break;
      case PG_F_PUSH_WORDS: if (WORDS) st.push(gptr<uint32_t>(cptr(p.ranges)[arg].words)[(int64_t)wt * 64 + lane] & valid_l); break;
      case PG_F_PUSH_RANGEIDX: if (WORDS) st.push(rangeidx_wtile(cptr(p.rangeidx)[arg], wt, valid_l, wscratch, lane)); break;
      case PG_F_PUSH_ALL: st.push(valid_l); break;
      case PG_F_PUSH_NONE: st.push(0u); break;
      case PG_F_AND: { const uint32_t b = st.s0; st.drop(); st.s0 &= b; break; }
      case PG_F_OR: { const uint32_t b = st.s0; st.drop(); st.s0 |= b; break; }
      case PG_F_NOT: st.s0 = (~st.s0) & valid_l; break;
      default: break;
    }
  }
  return st.s0;
}

// LDS-table aggregation of one wave tile for the fast path: group columns of <= 8 bits, 32-bit value sources, slots
// < 65 536 (two packed per register).  B quads per lane are in flight at a time; the first source's quads are requested
// before the group columns are decoded.
template <int B>
DEVFN void fast_aggregate_wtile(const PgQueryPlan& p, uint32_t mask_all, int wtile, int64_t* __restrict__ table, int lane, uint32_t rep) {
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t stride = (uint32_t)p.n_groups * R;   // slots per op
  int o_first = 0;
  while (o_first < p.n_ops && p.ops[o_first].src < 0) o_first++;
  const bool prefetch0 = o_first < p.n_ops && p.srcs[p.ops[o_first].src].col_kind == PG_COL_RAW32;
#pragma unroll
  for (int k0 = 0; k0 < 8; k0 += B) {
    const uint32_t m = B == 8 ? mask_all : ((mask_all >> (4 * k0)) & ((1u << (4 * (B & 7))) - 1u));
    if (__ballot(m != 0) == 0) continue;   // wave-uniform
    uint32_t x[B][4];
    if (prefetch0) {
      const GAS uint8_t* tb = gptr<uint8_t>(p.srcs[p.ops[o_first].src].data + (size_t)wtile * (PG_WAVE_DOCS * 4));
#pragma unroll
      for (int k = 0; k < B; k++) {
        const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)((k0 + k) * 64 + lane) : 0u;
        const u32x4 v = ldnt((const GAS u32x4*)(tb + q * 16u));
        x[k][0] = v.x; x[k][1] = v.y; x[k][2] = v.z; x[k][3] = v.w;
      }
    }
    uint32_t sp[B][2];   // packed slots: docs (0,1) and (2,3) of quad k
#pragma unroll
    for (int k = 0; k < B; k++) sp[k][0] = sp[k][1] = rep | (rep << 16);
    for (int g = 0; g < p.n_group_cols; g++) {
      const PgGroupCol& gc = p.gcols[g];
      const GAS uint32_t* tw = packed_wtile_base(gc.data, wtile, gc.bits);
      const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
      const uint32_t mult = (uint32_t)gc.mult * R;
      uint32_t r[B][2];
#pragma unroll
      for (int k = 0; k < B; k++) {
        const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)((k0 + k) * 64 + lane) : 0u;
        load_packed_quad<true>(tw, q, bits, r[k]);
      }
#pragma unroll
      for (int k = 0; k < B; k++) {
        uint32_t d[4];
        const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)((k0 + k) * 64 + lane) : 0u;
        decode_packed_quad<true>(r[k], q, bits, mask, d);
        sp[k][0] += __umul24(d[0], mult) + (__umul24(d[1], mult) << 16);   // dictIds < 256, mult x replicas < 65 536
        sp[k][1] += __umul24(d[2], mult) + (__umul24(d[3], mult) << 16);
      }
    }
#define PG_SLOT(k, i) ((sp[k][(i) >> 1] >> (((i) & 1) * 16)) & 0xFFFFu)
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      int64_t* base = table + (size_t)o * stride;
      if (op.src < 0) {
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((m >> (4 * k + i)) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + PG_SLOT(k, i)), 1ULL);
        continue;
      }
      const bool new_src = (o == 0 || p.ops[o - 1].src != op.src);
      if (new_src) {
        const PgValueSrc& S = p.srcs[op.src];
        if (S.col_kind == PG_COL_RAW32) {
          if (!(prefetch0 && o == o_first)) {
            const GAS uint8_t* tb = gptr<uint8_t>(S.data + (size_t)wtile * (PG_WAVE_DOCS * 4));
#pragma unroll
            for (int k = 0; k < B; k++) {
              const uint32_t q = ((m >> (4 * k)) & 0xFu) ? (uint32_t)((k0 + k) * 64 + lane) : 0u;
              const u32x4 v = ldnt((const GAS u32x4*)(tb + q * 16u));
              x[k][0] = v.x; x[k][1] = v.y; x[k][2] = v.z; x[k][3] = v.w;
            }
          }
#pragma unroll
          for (int k = 0; k < B; k++)
#pragma unroll
            for (int i = 0; i < 4; i++) x[k][i] = bswap32(x[k][i]);
        } else {   // dictionary-encoded 32-bit values
          const GAS uint32_t* tw = packed_wtile_base(S.data, wtile, S.bits);
          const uint32_t bits = (uint32_t)S.bits, mask = (1u << S.bits) - 1u;
#pragma unroll
          for (int k = 0; k < B; k++) {
#pragma unroll
            for (int i = 0; i < 4; i++) x[k][i] = 0;
            if ((m >> (4 * k)) & 0xFu) {
              uint32_t r[8], d[4];
              const uint32_t q = (uint32_t)((k0 + k) * 64 + lane);
              if (bits <= 8) { load_packed_quad<true>(tw, q, bits, r); decode_packed_quad<true>(r, q, bits, mask, d); }
              else { load_packed_quad<false>(tw, q, bits, r); decode_packed_quad<false>(r, q, bits, mask, d); }
#pragma unroll
              for (int i = 0; i < 4; i++) x[k][i] = ((m >> (4 * k + i)) & 1u) ? gptr<uint32_t>(S.dict)[d[i]] : 0u;
            }
          }
        }
      }
      if (!op.is_float) {
        if (op.fn == PG_ACC_SUM) {
#pragma unroll
          for (int k = 0; k < B; k++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              if ((m >> (4 * k + i)) & 1u)
                atomicAdd(reinterpret_cast<unsigned long long*>(base + PG_SLOT(k, i)), (unsigned long long)(int64_t)(int32_t)x[k][i]);
        } else if (op.fn == PG_ACC_MIN) {
#pragma unroll
          for (int k = 0; k < B; k++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              if ((m >> (4 * k + i)) & 1u) atomicMin(reinterpret_cast<long long*>(base + PG_SLOT(k, i)), (long long)(int32_t)x[k][i]);
        } else {
#pragma unroll
          for (int k = 0; k < B; k++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              if ((m >> (4 * k + i)) & 1u) atomicMax(reinterpret_cast<long long*>(base + PG_SLOT(k, i)), (long long)(int32_t)x[k][i]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((m >> (4 * k + i)) & 1u) acc_float(base + PG_SLOT(k, i), op.fn, (double)__uint_as_float(x[k][i]));
      }
    }
#undef PG_SLOT
  }
}

// Shared epilogue: statistics and the flush of the LDS accumulator table into this workgroup's partial table.
DEVFN void flush_workgroup(const PgQueryPlan& p, const int64_t* lds_table, const uint32_t* s_stat, bool lds_agg, int t) {
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  if (lds_agg) {
    const int R = p.replicas;
    const int groups = p.agg_mode == PG_AGG_LDS_PART ? p.part_groups : p.n_groups;   // this workgroup's table: [n_ops][groups]
    const int64_t n_out = (int64_t)p.n_ops * groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * n_out;
    // Many replicas per slot (no GROUP BY — a replica per lane, PG_AGG_SINGLE —, or a handful of groups): one lane per slot walking R replicas is
    // a chain of ~70-cycle LDS round trips, 30-60 us at the end of the kernel for R = 1 024.  A WAVEFRONT per slot instead: its lanes fold
    // replicas lane, lane + 64, ... and the wavefront folds the 64 partial results (integer folds commute; floating sums keep their one order).
    if (R >= 64) {
      const int lane = t & 63, wave = t >> 6;
      for (int64_t i = wave; i < n_out; i += PG_BLOCK / 64) {
        const PgAccOp op = p.ops[(int)(i / groups)];
        const int64_t* src = lds_table + i * R;
        if (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) {
          if (lane == 0) {
            double d = __longlong_as_double(src[0]);
            for (int r = 1; r < R; r++) d += __longlong_as_double(src[r]);
            out[i] = __double_as_longlong(d);
          }
          continue;
        }
        int64_t acc = src[lane];
        if (op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) { for (int r = lane + 64; r < R; r += 64) acc += src[r]; }
        else if (op.fn == PG_ACC_MIN) { for (int r = lane + 64; r < R; r += 64) acc = src[r] < acc ? src[r] : acc; }
        else { for (int r = lane + 64; r < R; r += 64) acc = src[r] > acc ? src[r] : acc; }
        acc = wave_fold_i64(acc, op.fn);
        if (lane == 0) out[i] = acc;
      }
      return;
    }
    for (int64_t i = t; i < n_out; i += PG_BLOCK) {
      const int o = (int)(i / groups);
      const PgAccOp op = p.ops[o];
      const int64_t* src = lds_table + i * R;   // (o * G + g) * R
      int64_t acc = src[0];
      if (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) {
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

// SK: ScanKind of the single scan leaf, -1 for none.  AGG: 0 = no accumulator table, 1 = LDS table (LDS / SINGLE modes).
template <int SK, int AGG>
__device__ __forceinline__ void fast_query_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ uint32_t s_wscratch[PG_WAVES_PER_BLOCK][64];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  const bool part_agg = AGG >= 2 && p.agg_mode == PG_AGG_LDS_PART;
  if (AGG) {
    const uint32_t table_slots = part_agg ? (uint32_t)p.part_groups : (uint32_t)p.n_groups * (uint32_t)p.replicas;
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
  }
  __syncthreads();

  const CAS PgScanLeaf& L = cptr(p.scans)[SK >= 0 ? p.fast_scan : 0];   // only dereferenced when SK >= 0
  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  uint32_t my_matched = 0, my_cand = 0;
  // tile walk: see generic_query_body (range-partitioned aggregation lets the ranges' workgroups of an XCD share its chunks)
  int chunk0 = (int)blockIdx.x, cstride = (int)gridDim.x;
  uint32_t part_lo = 0;
  bool count_stats = true;
  if (part_agg) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int range = idx % p.n_parts, j = idx / p.n_parts, nj = per_xcd / p.n_parts;
    chunk0 = xcd + 8 * j;
    cstride = 8 * nj;
    part_lo = (uint32_t)range * (uint32_t)p.part_groups;
    count_stats = range == 0;
  }
  const int wstride = cstride * PG_WAVES_PER_BLOCK;
  for (int wt = chunk0 * PG_WAVES_PER_BLOCK + wave; wt < p.n_wtiles; wt += wstride) {
    const int64_t wbase = (int64_t)wt * PG_WAVE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - wbase;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    uint32_t m = valid_quad_mask(n_valid, lane);
    if (p.n_index_instr > 0)
      m = lin_to_quad(index_program_lin<false>(p, p.n_index_instr, wt, wbase, valid_lin_mask(n_valid, lane), s_wscratch[wave], lane), lane);
    if (SK >= 0) {
      my_cand += (uint32_t)__popc(m);
      const GAS uint8_t* tb = sk_dict(SK) ? (const GAS uint8_t*)packed_wtile_base(L.data, wt, L.bits)
                                          : gptr<uint8_t>(L.data + (size_t)wt * (PG_WAVE_DOCS * (sk_raw32(SK) ? 4 : 8)));
      m = scan_wtile<(SK >= 0 ? SK : 0)>(L, m, tb, lane);
    }
    const uint32_t cnt = (uint32_t)__popc(m);
    my_matched += cnt;
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
    if (p.out_tile_counts) {
      const uint32_t wsum = wave_sum_u32(cnt);
      if (lane == 0 && wsum) atomicAdd(&p.out_tile_counts[wt / PG_WTILES_PER_TILE], wsum);
    }
    if (AGG && p.agg_mode != PG_AGG_NONE && __ballot(m != 0)) {
      if (AGG == 3) aggregate_wtile<PG_WIDE_AGG_B, false, true>(p, m, wt, lds_table, lane, rep, part_lo);   // + digit accumulators
      else if (AGG == 2) aggregate_wtile<PG_WIDE_AGG_B, false>(p, m, wt, lds_table, lane, rep, part_lo);   // any group width, 32/64-bit sources
      else fast_aggregate_wtile<PG_FAST_AGG_B>(p, m, wt, lds_table, lane, rep);
    }
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum && count_stats) atomicAdd(&s_stat[0], wsum);
  if (SK >= 0 && !p.fast_scan_pushed) {
    const uint32_t csum = wave_sum_u32(my_cand);
    if (lane == 0 && csum && count_stats) atomicAdd(&s_stat[L.stat_slot], csum);
  }
  __syncthreads();
  flush_workgroup(p, lds_table, s_stat, AGG && p.agg_mode != PG_AGG_NONE, t);
}

// Chain of up to PG_MAX_FAST_SCANS scan leaves, kind dispatched per leaf (wave-uniform): the shape of most multi-predicate
// filters (several range / IN predicates ANDed, optionally behind inverted-index leaves).
template <int AGG>
__device__ __forceinline__ void fast_multi_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ uint32_t s_wscratch[PG_WAVES_PER_BLOCK][64];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  const bool part_agg = AGG >= 2 && p.agg_mode == PG_AGG_LDS_PART;
  if (AGG) {
    const uint32_t table_slots = part_agg ? (uint32_t)p.part_groups : (uint32_t)p.n_groups * (uint32_t)p.replicas;
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
  }
  __syncthreads();

  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  uint32_t my_matched = 0;
  uint32_t my_cand[PG_MAX_FAST_SCANS] = {0, 0, 0, 0};
  int chunk0 = (int)blockIdx.x, cstride = (int)gridDim.x;   // tile walk: see generic_query_body
  uint32_t part_lo = 0;
  bool count_stats = true;
  if (part_agg) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int range = idx % p.n_parts, j = idx / p.n_parts, nj = per_xcd / p.n_parts;
    chunk0 = xcd + 8 * j;
    cstride = 8 * nj;
    part_lo = (uint32_t)range * (uint32_t)p.part_groups;
    count_stats = range == 0;
  }
  const int wstride = cstride * PG_WAVES_PER_BLOCK;
  for (int wt = chunk0 * PG_WAVES_PER_BLOCK + wave; wt < p.n_wtiles; wt += wstride) {
    const int64_t wbase = (int64_t)wt * PG_WAVE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - wbase;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    uint32_t m = valid_quad_mask(n_valid, lane);
    if (p.n_index_instr > 0)
      m = lin_to_quad(index_program_lin<false>(p, p.n_index_instr, wt, wbase, valid_lin_mask(n_valid, lane), s_wscratch[wave], lane), lane);
#pragma unroll
    for (int sidx = 0; sidx < PG_MAX_FAST_SCANS; sidx++) {
      if (sidx < p.n_fast_scans && __ballot(m != 0)) {   // wave-uniform
        const CAS PgScanLeaf& L = cptr(p.scans)[cptr(p.instrs)[p.n_index_instr + sidx].arg];
        my_cand[sidx] += (uint32_t)__popc(m);
        if (L.col_kind == PG_COL_FIXED_BIT) {
          const GAS uint8_t* tb = (const GAS uint8_t*)packed_wtile_base(L.data, wt, L.bits);
          if (L.bits <= 8) m = L.pred_kind == PG_P_RANGE ? scan_wtile<SK_DICT_RANGE_SMALL>(L, m, tb, lane) : scan_wtile<SK_DICT_LUT_SMALL>(L, m, tb, lane);
          else m = L.pred_kind == PG_P_RANGE ? scan_wtile<SK_DICT_RANGE_WIDE>(L, m, tb, lane) : scan_wtile<SK_DICT_LUT_WIDE>(L, m, tb, lane);
        } else if (L.col_kind == PG_COL_RAW32) {
          const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)wt * (PG_WAVE_DOCS * 4));
          m = L.val_type == PG_V_I32 ? scan_wtile<SK_I32_RANGE>(L, m, tb, lane) : scan_wtile<SK_F32_RANGE>(L, m, tb, lane);
        } else {
          const GAS uint8_t* tb = gptr<uint8_t>(L.data + (size_t)wt * (PG_WAVE_DOCS * 8));
          m = L.val_type == PG_V_I64 ? scan_wtile<SK_I64_RANGE>(L, m, tb, lane) : scan_wtile<SK_F64_RANGE>(L, m, tb, lane);
        }
      }
    }
    if (p.tail_posting >= 0 && __ballot(m != 0))   // wave-uniform
      m &= lin_to_quad(postings_wtile(cptr(p.postings)[p.tail_posting], wt, valid_lin_mask(n_valid, lane), s_wscratch[wave], lane), lane);
    const uint32_t cnt = (uint32_t)__popc(m);
    my_matched += cnt;
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
    if (p.out_tile_counts) {
      const uint32_t wsum = wave_sum_u32(cnt);
      if (lane == 0 && wsum) atomicAdd(&p.out_tile_counts[wt / PG_WTILES_PER_TILE], wsum);
    }
    if (AGG && p.agg_mode != PG_AGG_NONE && __ballot(m != 0)) {
      if (AGG == 3) aggregate_wtile<PG_WIDE_AGG_B, false, true>(p, m, wt, lds_table, lane, rep, part_lo);
      else if (AGG == 2) aggregate_wtile<PG_WIDE_AGG_B, false>(p, m, wt, lds_table, lane, rep, part_lo);
      else fast_aggregate_wtile<PG_FAST_AGG_B>(p, m, wt, lds_table, lane, rep);
    }
  }
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum && count_stats) atomicAdd(&s_stat[0], wsum);
#pragma unroll
  for (int sidx = 0; sidx < PG_MAX_FAST_SCANS; sidx++) {
    if (sidx < p.n_fast_scans && !(sidx == 0 && p.fast_scan_pushed)) {
      const uint32_t csum = wave_sum_u32(my_cand[sidx]);
      const int slot = cptr(p.scans)[cptr(p.instrs)[p.n_index_instr + sidx].arg].stat_slot;
      if (lane == 0 && csum && count_stats) atomicAdd(&s_stat[slot], csum);
    }
  }
  __syncthreads();
  flush_workgroup(p, lds_table, s_stat, AGG && p.agg_mode != PG_AGG_NONE, t);
}
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_fast_multi_f(const PgQueryPlan p) { fast_multi_body<0>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_fast_multi_a(const PgQueryPlan p) { fast_multi_body<1>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_fast_multi_w(const PgQueryPlan p) { fast_multi_body<2>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_fast_multi_wd(const PgQueryPlan p) { fast_multi_body<3>(p); }

#ifndef PG_FAST_MIN_WAVES_PER_SIMD
#define PG_FAST_MIN_WAVES_PER_SIMD 1   // measurement knob: 8 asks for <= 64 VGPRs (two 1024-thread workgroups per CU)
#endif
#define PG_FAST_KERNEL(NAME, SK, AGG) \
  PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK, PG_FAST_MIN_WAVES_PER_SIMD) NAME(const PgQueryPlan p) { fast_query_body<SK, AGG>(p); }
PG_FAST_KERNEL(pg_fast_none_f, -1, 0)
PG_FAST_KERNEL(pg_fast_none_a, -1, 1)
PG_FAST_KERNEL(pg_fast_none_w, -1, 2)
PG_FAST_KERNEL(pg_fast_none_wd, -1, 3)
PG_FAST_KERNEL(pg_fast_i32range_f, SK_I32_RANGE, 0)
PG_FAST_KERNEL(pg_fast_i32range_a, SK_I32_RANGE, 1)
PG_FAST_KERNEL(pg_fast_dictrange_f, SK_DICT_RANGE_SMALL, 0)
PG_FAST_KERNEL(pg_fast_dictrange_a, SK_DICT_RANGE_SMALL, 1)
PG_FAST_KERNEL(pg_fast_dictlut_f, SK_DICT_LUT_SMALL, 0)
PG_FAST_KERNEL(pg_fast_dictlut_a, SK_DICT_LUT_SMALL, 1)

// Register stack of match masks; the top is always s0 (push / pop shift the others), so no dynamic register indexing.
struct MaskStack {
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
  DEVFN void push(uint32_t v) { s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v; }
  DEVFN void drop() { s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; }
  DEVFN void pop_and() { const uint32_t b = s0; drop(); s0 &= b; }
  DEVFN void pop_or() { const uint32_t b = s0; drop(); s0 |= b; }
};

// =====================================================================================================================
// The interpreter kernel (any supported plan): same wave-tile model, runtime dispatch.  512-thread workgroups so that the
// compiler has 256 VGPRs per lane (no spills); two workgroups per CU when the LDS table allows.
// dynamic LDS: the accumulator table (LDS / SINGLE modes)
// =====================================================================================================================
// TABLE: 0 = no accumulator table (filter only / COUNT from the match count), 1 = LDS table (LDS / SINGLE / LDS_PART modes),
// 2 = dense HBM table — three kernels instead of one so that each carries one inlined copy of the aggregation code.
template <int TABLE, bool DIG = false>
__device__ __forceinline__ void generic_query_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ uint32_t s_wscratch[PG_GENERIC_BLOCK / 64][64];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  const bool part_agg = p.agg_mode == PG_AGG_LDS_PART;
  const bool lds_agg = (p.agg_mode == PG_AGG_LDS || p.agg_mode == PG_AGG_SINGLE || part_agg);
  const uint32_t table_slots = part_agg ? (uint32_t)p.part_groups : (uint32_t)p.n_groups * (uint32_t)p.replicas;

  if (t < PG_MAX_STATS) s_stat[t] = 0;
  if (lds_agg) {
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_GENERIC_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
  }
  if (TABLE == 1)
    for (int x = 0; x < p.n_aux; x++)
      if (p.aux[x].lds_offset >= 0) {
        uint32_t* z = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + p.aux[x].lds_offset);
        for (int64_t i = t; i < p.aux[x].rep_bytes / 4; i += PG_GENERIC_BLOCK) z[i] = 0;
      }
  __syncthreads();

  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  uint32_t my_matched = 0;
  // Tile walk.  Default: workgroup b takes chunks (= one wave tile per wavefront) b, b + grid, ...  Range-partitioned
  // aggregation: the grid is 8 x per_xcd workgroups, dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8 —
  // HW_REG_XCC_ID, tools/probes/atomic_scope.hip); XCD x owns the chunks c with c % 8 == x, and the n_parts ranges'
  // workgroups of that XCD walk the same chunks (HBM once, L2 for the others), per_xcd / n_parts workgroups per range.
  int chunk0 = (int)blockIdx.x, cstride = (int)gridDim.x;
  uint32_t part_lo = 0;
  bool count_stats = true;
  if (part_agg) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3;
    const int range = idx % p.n_parts, j = idx / p.n_parts, nj = per_xcd / p.n_parts;
    chunk0 = xcd + 8 * j;
    cstride = 8 * nj;
    part_lo = (uint32_t)range * (uint32_t)p.part_groups;
    count_stats = range == 0;   // every range sees every doc: only range 0 reports the filter statistics
  }
  const int wstride = cstride * (PG_GENERIC_BLOCK / 64);
  const int split_shift = p.tile_split_shift, n_vtiles = p.n_wtiles << split_shift;   // PgQueryPlan::tile_split_shift
  for (int vt = chunk0 * (PG_GENERIC_BLOCK / 64) + wave; vt < n_vtiles; vt += wstride) {
    const int wt = vt >> split_shift, share = vt & ((1 << split_shift) - 1);
    const bool first_share = share == 0;
    const int64_t wbase = (int64_t)wt * PG_WAVE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - wbase;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    const uint32_t valid_q = valid_quad_mask(n_valid, lane);
    const uint32_t valid_l = valid_lin_mask(n_valid, lane);

    // ---- filter program ---------------------------------------------------------------------------------------------
    MaskStack st;
    if (p.n_lin_prefix > 0) st.push(lin_to_quad(index_program_lin<true>(p, p.n_lin_prefix, wt, wbase, valid_l, s_wscratch[wave], lane), lane));
    for (int i = p.n_lin_prefix; i < p.n_instr; i++) {
      const int fop = cptr(p.instrs)[i].op, farg = cptr(p.instrs)[i].arg;
      switch (fop) {
        case PG_F_PUSH_POSTINGS:
          st.push(lin_to_quad(postings_wtile(cptr(p.postings)[farg], wt, valid_l, s_wscratch[wave], lane), lane));
          break;
        case PG_F_PUSH_RANGES:
          st.push(lin_to_quad(ranges_wtile(cptr(p.ranges)[farg], wbase, valid_l, lane), lane));
          break;
        case PG_F_PUSH_WORDS:
          st.push(lin_to_quad(gptr<uint32_t>(cptr(p.ranges)[farg].words)[(int64_t)wt * 64 + lane] & valid_l, lane));
          break;
        case PG_F_PUSH_RANGEIDX:
          st.push(lin_to_quad(rangeidx_wtile(cptr(p.rangeidx)[farg], wt, valid_l, s_wscratch[wave], lane), lane));
          break;
        case PG_F_PUSH_ALL:
          st.push(valid_q);
          break;
        case PG_F_PUSH_NONE:
          st.push(0u);
          break;
        case PG_F_PUSH_SCAN:
          st.push(scan_dispatch(cptr(p.scans)[farg], valid_q, wt, lane));
          break;
        case PG_F_AND_SCAN: {
          const CAS PgScanLeaf& L = cptr(p.scans)[farg];
          const uint32_t cand = st.s0;
          const uint32_t nc = wave_sum_u32((uint32_t)__popc(cand));
          if (nc) {   // wave-uniform
            st.s0 = scan_dispatch(L, cand, wt, lane);
            if (lane == 0 && count_stats && first_share) atomicAdd(&s_stat[L.stat_slot], nc);
          }
          break;
        }
        case PG_F_AND: st.pop_and(); break;
        case PG_F_OR: st.pop_or(); break;
        case PG_F_NOT: st.s0 = (~st.s0) & valid_q; break;
        default: break;
      }
    }
    uint32_t m = st.s0;
    if (first_share) {
      if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
      if (p.out_tile_counts) {
        const uint32_t wsum = wave_sum_u32((uint32_t)__popc(m));
        if (lane == 0 && wsum) atomicAdd(&p.out_tile_counts[wt / PG_WTILES_PER_TILE], wsum);
      }
    }
    if (split_shift) {   // this wavefront's share of the tile's matches: quad slots k with k = share (mod 8 or fewer), then lane classes
      const int kbits = split_shift < 3 ? split_shift : 3;
      const uint32_t kmask = (1u << kbits) - 1u, ksel = (uint32_t)share & kmask;
      uint32_t keep = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) keep |= (((uint32_t)k & kmask) == ksel ? 0xFu : 0u) << (4 * k);
      const uint32_t lmask = (1u << (split_shift - kbits)) - 1u;
      if (((uint32_t)lane & lmask) != ((uint32_t)share >> kbits)) keep = 0;
      m &= keep;
    }
    const uint32_t cnt = (uint32_t)__popc(m);
    my_matched += cnt;

    // ---- aggregation ----------------------------------------------------------------------------------------------
    if (TABLE != 0 && __ballot(m != 0)) {
      if (TABLE == 1) {
        if (!DIG && p.fast_agg_shape) fast_aggregate_wtile<PG_FAST_AGG_B>(p, m, wt, lds_table, lane, rep);
        else aggregate_wtile<PG_GENERIC_AGG_B, true, DIG>(p, m, wt, lds_table, lane, rep, part_lo);
      } else {
        aggregate_wtile<PG_GENERIC_AGG_B, true, DIG>(p, m, wt, p.partials, lane, rep);
      }
    }
  }

  // ---- epilogue: statistics and accumulator flush -----------------------------------------------------------------------
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum && count_stats) atomicAdd(&s_stat[0], wsum);
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);

  if (lds_agg) {
    const int R = p.replicas;
    const int groups = part_agg ? p.part_groups : p.n_groups;   // this workgroup's partial table: [n_ops][groups]
    const int64_t n_out = (int64_t)p.n_ops * groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * n_out;
    // (>= 64 replicas per slot — no GROUP BY, a handful of groups: a wavefront per slot folds them, see flush_workgroup)
    for (int64_t i = (t >> 6); R >= 64 && i < n_out; i += PG_GENERIC_BLOCK / 64) {
      const PgAccOp op = p.ops[(int)(i / groups)];
      const int64_t* src = lds_table + i * R;
      const int ln = t & 63;
      if (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) {
        if (ln == 0) {
          double d = __longlong_as_double(src[0]);
          for (int r = 1; r < R; r++) d += __longlong_as_double(src[r]);
          out[i] = __double_as_longlong(d);
        }
        continue;
      }
      int64_t acc = src[ln];
      if (op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) { for (int r = ln + 64; r < R; r += 64) acc += src[r]; }
      else if (op.fn == PG_ACC_MIN) { for (int r = ln + 64; r < R; r += 64) acc = src[r] < acc ? src[r] : acc; }
      else { for (int r = ln + 64; r < R; r += 64) acc = src[r] > acc ? src[r] : acc; }
      acc = wave_fold_i64(acc, op.fn);
      if (ln == 0) out[i] = acc;
    }
    for (int64_t i = t; R < 64 && i < n_out; i += PG_GENERIC_BLOCK) {
      const int o = (int)(i / groups);
      const PgAccOp op = p.ops[o];
      const int64_t* src = lds_table + i * R;   // (o * G + g) * R
      int64_t acc = src[0];
      if (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) {
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
  if (TABLE == 1)   // LDS-resident DISTINCTCOUNT / HLL states: this workgroup's partial, merged by pg_reduce_aux_kernel
    for (int x = 0; x < p.n_aux; x++)
      if (p.aux[x].lds_offset >= 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(smem) + p.aux[x].lds_offset);
        uint32_t* dst = p.aux[x].base + (int64_t)blockIdx.x * (p.aux[x].rep_bytes / 4);
        for (int64_t i = t; i < p.aux[x].rep_bytes / 4; i += PG_GENERIC_BLOCK) dst[i] = src[i];
      }
}
PG_KERNEL __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_generic_query_f(const PgQueryPlan p) { generic_query_body<0>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_generic_query_l(const PgQueryPlan p) { generic_query_body<1>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_generic_query_g(const PgQueryPlan p) { generic_query_body<2>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_generic_query_ld(const PgQueryPlan p) { generic_query_body<1, true>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_generic_query_gd(const PgQueryPlan p) { generic_query_body<2, true>(p); }

// Combines the per-workgroup partial tables: out[op][g].  One wavefront per output slot, lanes stride over the
// workgroups, fixed butterfly order (deterministic also for floating sums).  The last block also moves the statistics
// counters behind the table (out[n_out .. n_out+PG_MAX_STATS)) and re-zeroes them for the next query on this stream, so
// that one device→host copy returns everything.
PG_KERNEL __global__ void __launch_bounds__(256) pg_reduce_partials_kernel(const int64_t* __restrict__ partials,
                                                                             int64_t* __restrict__ out, int n_wg,
                                                                             int n_ops, int n_groups,
                                                                             const PgAccOp* __restrict__ ops,
                                                                             unsigned long long* __restrict__ stats,
                                                                             int reduce) {
  const int64_t n_out = (int64_t)n_ops * n_groups;
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x < PG_MAX_STATS) {
      out[n_out + threadIdx.x] = (int64_t)stats[threadIdx.x];
      stats[threadIdx.x] = 0;
    }
    return;
  }
  if (!reduce) return;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_out) return;
  const PgAccOp op = ops[i / n_groups];
  const int kind = (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) ? 0 : ((op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) ? 1 : (op.fn == PG_ACC_MIN ? 2 : 3));
  int64_t acc = pg_acc_identity(op.fn, op.is_float);
  auto combine = [&](int64_t a, int64_t b) -> int64_t {
    if (kind == 0) return __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b));
    if (kind == 1) return a + b;
    if (kind == 2) return b < a ? b : a;
    return b > a ? b : a;
  };
  for (int w = lane; w < n_wg; w += 64) acc = combine(acc, partials[(int64_t)w * n_out + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int64_t other = __shfl_xor((long long)acc, off, 64);
    acc = combine(acc, other);
  }
  if (lane == 0) out[i] = acc;
}

// =====================================================================================================================
// Radix-partitioned group-by (PG_AGG_RADIX): pass 1 / pass 2 over the match words, then per-bucket LDS aggregation.
// =====================================================================================================================
// raw keys of the 4 docs of B quads: Σ dictId_j · mult_j, one group column at a time (any width)
template <int B>
DEVFN void radix_keys_of(const PgQueryPlan& p, const uint32_t (&qi)[B], int wtile, uint32_t (&key)[B][4]) {
#pragma unroll
  for (int u = 0; u < B; u++)
#pragma unroll
    for (int i = 0; i < 4; i++) key[u][i] = 0;
  for (int g = 0; g < p.n_group_cols; g++) {
    const PgGroupCol& gc = p.gcols[g];
    const GAS uint32_t* tw = packed_wtile_base(gc.data, wtile, gc.bits);
    const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u, mult = (uint32_t)gc.mult;
    if (bits <= 8) {
      uint32_t r[B][2];
#pragma unroll
      for (int u = 0; u < B; u++) load_packed_quad<true>(tw, qi[u], bits, r[u]);
#pragma unroll
      for (int u = 0; u < B; u++) {
        uint32_t d[4];
        decode_packed_quad<true>(r[u], qi[u], bits, mask, d);
#pragma unroll
        for (int i = 0; i < 4; i++) key[u][i] += d[i] * mult;
      }
    } else {
      uint32_t r[B][8];
#pragma unroll
      for (int u = 0; u < B; u++) load_packed_quad<false>(tw, qi[u], bits, r[u]);
#pragma unroll
      for (int u = 0; u < B; u++) {
        uint32_t d[4];
        decode_packed_quad<false>(r[u], qi[u], bits, mask, d);
#pragma unroll
        for (int i = 0; i < 4; i++) key[u][i] += d[i] * mult;
      }
    }
  }
}

// 64-bit raw keys (PG_AGG_RADIX_HASH): Σ dictId_j · mult_j in int64
template <int B>
DEVFN void radix_keys_of64(const PgQueryPlan& p, const uint32_t (&qi)[B], int wtile, uint64_t (&key)[B][4]) {
#pragma unroll
  for (int u = 0; u < B; u++)
#pragma unroll
    for (int i = 0; i < 4; i++) key[u][i] = 0;
  for (int g = 0; g < p.n_group_cols; g++) {
    const PgGroupCol& gc = p.gcols[g];
    if (gc.col_kind != PG_COL_FIXED_BIT) {   // no-dictionary group column: key = value ^ 2^63
      if (gc.col_kind == PG_COL_RAW32) {
        const GAS uint8_t* tb = gptr<uint8_t>(gc.data + (size_t)wtile * (PG_WAVE_DOCS * 4));
#pragma unroll
        for (int u = 0; u < B; u++) {
          const u32x4 v = ldnt((const GAS u32x4*)(tb + qi[u] * 16u));
          key[u][0] = (uint64_t)(int64_t)(int32_t)bswap32(v.x) ^ (1ULL << 63); key[u][1] = (uint64_t)(int64_t)(int32_t)bswap32(v.y) ^ (1ULL << 63);
          key[u][2] = (uint64_t)(int64_t)(int32_t)bswap32(v.z) ^ (1ULL << 63); key[u][3] = (uint64_t)(int64_t)(int32_t)bswap32(v.w) ^ (1ULL << 63);
        }
      } else {
        const GAS uint8_t* tb = gptr<uint8_t>(gc.data + (size_t)wtile * (PG_WAVE_DOCS * 8));
#pragma unroll
        for (int u = 0; u < B; u++) {
          const GAS u32x4* pp = (const GAS u32x4*)(tb + qi[u] * 32u);
          const u32x4 a = ldnt(pp), b = ldnt(pp + 1);
          key[u][0] = ((((uint64_t)bswap32(a.x) << 32) | bswap32(a.y))) ^ (1ULL << 63); key[u][1] = ((((uint64_t)bswap32(a.z) << 32) | bswap32(a.w))) ^ (1ULL << 63);
          key[u][2] = ((((uint64_t)bswap32(b.x) << 32) | bswap32(b.y))) ^ (1ULL << 63); key[u][3] = ((((uint64_t)bswap32(b.z) << 32) | bswap32(b.w))) ^ (1ULL << 63);
        }
      }
      continue;
    }
    const GAS uint32_t* tw = packed_wtile_base(gc.data, wtile, gc.bits);
    const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
    const uint64_t mult = (uint64_t)gc.mult;
#pragma unroll
    for (int u = 0; u < B; u++) {
      uint32_t r[8], d[4];
      if (bits <= 8) { load_packed_quad<true>(tw, qi[u], bits, r); decode_packed_quad<true>(r, qi[u], bits, mask, d); }
      else { load_packed_quad<false>(tw, qi[u], bits, r); decode_packed_quad<false>(r, qi[u], bits, mask, d); }
#pragma unroll
      for (int i = 0; i < 4; i++) key[u][i] += (uint64_t)d[i] * mult;
    }
  }
}
DEVFN uint64_t radix_mix64(uint64_t x) {   // splitmix64 finaliser: bucket = low bits, LDS slot = bits 16..
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

// values of one source for the 4 docs of B quads, as the int64 the accumulators take (INT / LONG sign-extended, FLOAT / DOUBLE as
// the bits of the double) — raw 32/64-bit columns and dictionary-encoded ones
template <int B>
DEVFN void radix_source_values(const PgValueSrc& S, const uint32_t (&qi)[B], int wt, int64_t (&out)[B][4]) {
  if (S.col_kind == PG_COL_RAW32) {
    const GAS uint8_t* tb = gptr<uint8_t>(S.data + (size_t)wt * (PG_WAVE_DOCS * 4));
    u32x4 v[B];
#pragma unroll
    for (int u = 0; u < B; u++) v[u] = ldnt((const GAS u32x4*)(tb + qi[u] * 16u));
#pragma unroll
    for (int u = 0; u < B; u++) {
      const uint32_t x[4] = {bswap32(v[u].x), bswap32(v[u].y), bswap32(v[u].z), bswap32(v[u].w)};
#pragma unroll
      for (int i = 0; i < 4; i++)
        out[u][i] = S.val_type == PG_V_I32 ? (int64_t)(int32_t)x[i] : __double_as_longlong((double)__uint_as_float(x[i]));
    }
  } else if (S.col_kind == PG_COL_RAW64) {
    const GAS uint8_t* tb = gptr<uint8_t>(S.data + (size_t)wt * (PG_WAVE_DOCS * 8));
    u32x4 a[B], b[B];
#pragma unroll
    for (int u = 0; u < B; u++) {
      const GAS u32x4* pp = (const GAS u32x4*)(tb + qi[u] * 32u);
      a[u] = ldnt(pp);
      b[u] = ldnt(pp + 1);
    }
#pragma unroll
    for (int u = 0; u < B; u++) {
      out[u][0] = (int64_t)(((uint64_t)bswap32(a[u].x) << 32) | bswap32(a[u].y)); out[u][1] = (int64_t)(((uint64_t)bswap32(a[u].z) << 32) | bswap32(a[u].w));
      out[u][2] = (int64_t)(((uint64_t)bswap32(b[u].x) << 32) | bswap32(b[u].y)); out[u][3] = (int64_t)(((uint64_t)bswap32(b[u].z) << 32) | bswap32(b[u].w));
    }
  } else {   // dictionary-encoded: dictIds → values (every dictId read is a valid one: idle lanes re-read quad 0 of the tile)
    const GAS uint32_t* tw = packed_wtile_base(S.data, wt, S.bits);
    const uint32_t bits = (uint32_t)S.bits, mask = (1u << S.bits) - 1u;
#pragma unroll
    for (int u = 0; u < B; u++) {
      uint32_t r[8], d[4];
      if (bits <= 8) { load_packed_quad<true>(tw, qi[u], bits, r); decode_packed_quad<true>(r, qi[u], bits, mask, d); }
      else { load_packed_quad<false>(tw, qi[u], bits, r); decode_packed_quad<false>(r, qi[u], bits, mask, d); }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (S.val_type == PG_V_I32) out[u][i] = (int64_t)(int32_t)gptr<uint32_t>(S.dict)[d[i]];
        else if (S.val_type == PG_V_F32) out[u][i] = __double_as_longlong((double)__uint_as_float(gptr<uint32_t>(S.dict)[d[i]]));
        else out[u][i] = (int64_t)gptr<uint64_t>(S.dict)[d[i]];
      }
    }
  }
}

// one value of a bit-packed column at in-tile doc index `doc` (any width)
DEVFN uint32_t packed_value_at(const GAS uint32_t* __restrict__ tw, uint32_t doc, uint32_t bits) {
  const uint32_t bit0 = doc * bits;
  const u32x2 v = *(const GAS u32x2_a4*)(tw + (bit0 >> 5));
  const uint64_t win = ((uint64_t)bswap32(v.x) << 32) | (uint64_t)bswap32(v.y);
  return (uint32_t)(win >> (64u - (bit0 & 31u) - bits)) & ((1u << bits) - 1u);
}
DEVFN int64_t source_value_at(const PgValueSrc& S, int wt, uint32_t doc) {
  if (S.col_kind == PG_COL_RAW32) {
    const uint32_t x = bswap32(gptr<uint32_t>(S.data + (size_t)wt * (PG_WAVE_DOCS * 4))[doc]);
    return S.val_type == PG_V_I32 ? (int64_t)(int32_t)x : __double_as_longlong((double)__uint_as_float(x));
  }
  if (S.col_kind == PG_COL_RAW64) {
    const u32x2 v = gptr<u32x2>(S.data + (size_t)wt * (PG_WAVE_DOCS * 8))[doc];
    return (int64_t)(((uint64_t)bswap32(v.x) << 32) | bswap32(v.y));
  }
  const uint32_t d = packed_value_at(packed_wtile_base(S.data, wt, S.bits), doc, (uint32_t)S.bits);
  if (S.val_type == PG_V_I32) return (int64_t)(int32_t)gptr<uint32_t>(S.dict)[d];
  if (S.val_type == PG_V_F32) return __double_as_longlong((double)__uint_as_float(gptr<uint32_t>(S.dict)[d]));
  return (int64_t)gptr<uint64_t>(S.dict)[d];
}

// Sparse tiles (at most PG_SELVEC_MAX matching docs): the matches are compacted into a per-wavefront list (the linear mask
// layout gives ascending docIds) and handled 64 per round with every lane busy — one gather per column and doc — instead of
// decoding all 2 048 positions of the tile twice (count and scatter pass) for a few dozen matches.
#define PG_SELVEC_MAX 512
template <int PASS, bool HASH>
DEVFN void radix_selvec_tile(const PgQueryPlan& p, uint32_t mlin, uint32_t n_match, int wt, uint32_t* __restrict__ s_cnt,
                             const uint32_t* __restrict__ s_base, uint16_t* __restrict__ list, int lane) {
  {
    const uint32_t c = (uint32_t)__popc(mlin);
    uint32_t x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)x, off, 64);
      if (lane >= off) x += y;
    }
    uint32_t pos = x - c, mm = mlin;
    while (__ballot(mm != 0)) {
      if (mm) {
        const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
        mm &= mm - 1u;
        list[pos++] = (uint16_t)((uint32_t)lane * 32u + b);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint32_t local_mask = (1u << p.radix_shift) - 1u, bmask = (uint32_t)p.radix_buckets - 1u, stride = (uint32_t)p.radix_stride;
  for (uint32_t r0 = 0; r0 < n_match; r0 += 64) {
    const bool on = r0 + (uint32_t)lane < n_match;
    const uint32_t doc = on ? (uint32_t)list[r0 + lane] : (uint32_t)list[0];
    uint64_t key = 0;
    for (int g = 0; g < p.n_group_cols; g++) {
      const PgGroupCol& gc = p.gcols[g];
      if (HASH && gc.col_kind == PG_COL_RAW32) key = (uint64_t)(int64_t)(int32_t)bswap32(gptr<uint32_t>(gc.data + (size_t)wt * (PG_WAVE_DOCS * 4))[doc]) ^ (1ULL << 63);
      else if (HASH && gc.col_kind == PG_COL_RAW64) {
        const u32x2 v = gptr<u32x2>(gc.data + (size_t)wt * (PG_WAVE_DOCS * 8))[doc];
        key = (((uint64_t)bswap32(v.x) << 32) | bswap32(v.y)) ^ (1ULL << 63);
      } else key += (uint64_t)packed_value_at(packed_wtile_base(gc.data, wt, gc.bits), doc, (uint32_t)gc.bits) * (uint64_t)gc.mult;
    }
    const uint32_t b = HASH ? ((uint32_t)radix_mix64(key) & bmask) : ((uint32_t)key >> p.radix_shift);
    if (PASS == 1) {
      if (on) atomicAdd(&s_cnt[b], 1u);
    } else {
      uint8_t* tp = nullptr;
      if (on) tp = p.radix_tuples + (size_t)(s_base[b] + atomicAdd(&s_cnt[b], 1u)) * stride;
      const uint32_t docid = (uint32_t)wt * PG_WAVE_DOCS + doc;
      if (HASH) {
        if (on) { u32x4 w4 = {(uint32_t)key, (uint32_t)(key >> 32), docid, 0u}; *reinterpret_cast<u32x4*>(tp) = w4; }
        for (int si = 0; si < p.n_srcs; si++) {
          const int64_t v = source_value_at(p.srcs[si], wt, doc);
          if (on) *reinterpret_cast<int64_t*>(tp + 16 + 8 * si) = v;
        }
      } else {
        if (p.n_srcs > 0) {
          const int64_t v0 = source_value_at(p.srcs[0], wt, doc);
          if (on) { u32x4 w4 = {(uint32_t)key & local_mask, docid, (uint32_t)(uint64_t)v0, (uint32_t)((uint64_t)v0 >> 32)}; *reinterpret_cast<u32x4*>(tp) = w4; }
        } else if (on) {
          u32x2 w2 = {(uint32_t)key & local_mask, docid};
          *reinterpret_cast<u32x2*>(tp) = w2;
        }
        for (int si = 1; si < p.n_srcs; si++) {
          const int64_t v = source_value_at(p.srcs[si], wt, doc);
          if (on) *reinterpret_cast<int64_t*>(tp + 8 + 8 * si) = v;
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();   // the list is rewritten by this wavefront's next sparse tile
}

// PASS 1: count the matching docs per bucket → radix_hist[workgroup][bucket].
// PASS 2: radix_hist holds exact offsets; every matching doc claims the next slot of its (workgroup, bucket) range with an LDS
//         counter and writes its tuple there.  Tuples are arrays of structs — {local key, docId} then 8 bytes per source, the
//         stride a multiple of 16 — so that a tuple goes out in 16-byte stores: with tens of buckets every lane of a store hits
//         its own cache line, and the number of store instructions is what the pass costs (one 16 B store for key + one value).
template <int PASS, bool HASH>
__device__ __forceinline__ void radix_pass_body(const PgQueryPlan& p) {
  __shared__ uint32_t s_cnt[PG_MAX_RADIX_BUCKETS];
  __shared__ uint32_t s_base[PASS == 2 ? PG_MAX_RADIX_BUCKETS : 1];
  __shared__ uint16_t s_list[PG_WAVES_PER_BLOCK][PG_SELVEC_MAX];
  constexpr int B = PASS == 1 ? (HASH ? 2 : 4) : 2;
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6);
  const int P = p.radix_buckets;
  for (int i = t; i < P; i += PG_BLOCK) {
    s_cnt[i] = 0;
    if (PASS == 2) s_base[i] = p.radix_bucket_start[i] + p.radix_hist[(int64_t)blockIdx.x * P + i];
  }
  __syncthreads();
  const uint32_t local_mask = (1u << p.radix_shift) - 1u;
  const uint32_t stride = (uint32_t)p.radix_stride;
  const int wstride = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  for (int wt = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave; wt < p.n_wtiles; wt += wstride) {
    const int64_t rem = (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    const uint32_t mlin = gptr<uint32_t>(p.match_words)[(int64_t)wt * 64 + lane] & valid_lin_mask(n_valid, lane);
    if (__ballot(mlin != 0) == 0) continue;
    const uint32_t n_match = wave_sum_u32((uint32_t)__popc(mlin));
    if (n_match <= PG_SELVEC_MAX) {   // wave-uniform
      radix_selvec_tile<PASS, HASH>(p, mlin, n_match, wt, s_cnt, s_base, s_list[wave], lane);
      continue;
    }
    const uint32_t m = lin_to_quad(mlin, lane);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += B) {
      const uint32_t mb = (m >> (4 * k0)) & ((1u << (4 * B)) - 1u);
      if (__ballot(mb != 0) == 0) continue;
      uint32_t qi[B];
#pragma unroll
      for (int u = 0; u < B; u++) qi[u] = ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u;
      if (HASH) {
        uint64_t key64[B][4];
        radix_keys_of64<B>(p, qi, wt, key64);
        const uint32_t bmask = (uint32_t)p.radix_buckets - 1u;
        if (PASS == 1) {
#pragma unroll
          for (int u = 0; u < B; u++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              if ((mb >> (4 * u + i)) & 1u) atomicAdd(&s_cnt[(uint32_t)radix_mix64(key64[u][i]) & bmask], 1u);
        } else {
          uint8_t* tp[B][4];
#pragma unroll
          for (int u = 0; u < B; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
              tp[u][i] = nullptr;
              if ((mb >> (4 * u + i)) & 1u) {
                const uint32_t b = (uint32_t)radix_mix64(key64[u][i]) & bmask;
                tp[u][i] = p.radix_tuples + (size_t)(s_base[b] + atomicAdd(&s_cnt[b], 1u)) * stride;
                const uint32_t docid = (uint32_t)wt * PG_WAVE_DOCS + 4u * (uint32_t)((k0 + u) * 64 + lane) + (uint32_t)i;
                u32x4 w4 = {(uint32_t)key64[u][i], (uint32_t)(key64[u][i] >> 32), docid, 0u};
                *reinterpret_cast<u32x4*>(tp[u][i]) = w4;
              }
            }
          for (int si = 0; si < p.n_srcs; si += 2) {   // sources, two per 16-byte store, behind the 16-byte header
            int64_t va[B][4], vb[B][4];
            radix_source_values<B>(p.srcs[si], qi, wt, va);
            const bool two = si + 1 < p.n_srcs;
            if (two) radix_source_values<B>(p.srcs[si + 1], qi, wt, vb);
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++)
                if ((mb >> (4 * u + i)) & 1u) {
                  u32x4 w4 = {(uint32_t)(uint64_t)va[u][i], (uint32_t)((uint64_t)va[u][i] >> 32), 0u, 0u};
                  if (two) { w4.z = (uint32_t)(uint64_t)vb[u][i]; w4.w = (uint32_t)((uint64_t)vb[u][i] >> 32); }
                  *reinterpret_cast<u32x4*>(tp[u][i] + 16 + 8 * si) = w4;
                }
          }
        }
        continue;
      }
      uint32_t key[B][4];
      radix_keys_of<B>(p, qi, wt, key);
      if (PASS == 1) {
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++)
            if ((mb >> (4 * u + i)) & 1u) atomicAdd(&s_cnt[key[u][i] >> p.radix_shift], 1u);
      } else {
        uint8_t* tp[B][4];
        int64_t v0[B][4];
        if (p.n_srcs > 0) radix_source_values<B>(p.srcs[0], qi, wt, v0);
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            tp[u][i] = nullptr;
            if ((mb >> (4 * u + i)) & 1u) {
              const uint32_t b = key[u][i] >> p.radix_shift;
              tp[u][i] = p.radix_tuples + (size_t)(s_base[b] + atomicAdd(&s_cnt[b], 1u)) * stride;
              const uint32_t docid = (uint32_t)wt * PG_WAVE_DOCS + 4u * (uint32_t)((k0 + u) * 64 + lane) + (uint32_t)i;
              if (p.n_srcs > 0) {
                u32x4 w4 = {key[u][i] & local_mask, docid, (uint32_t)(uint64_t)v0[u][i], (uint32_t)((uint64_t)v0[u][i] >> 32)};
                *reinterpret_cast<u32x4*>(tp[u][i]) = w4;
              } else {
                u32x2 w2 = {key[u][i] & local_mask, docid};
                *reinterpret_cast<u32x2*>(tp[u][i]) = w2;
              }
            }
          }
        for (int si = 1; si < p.n_srcs; si += 2) {   // further sources, two per 16-byte store
          int64_t va[B][4], vb[B][4];
          radix_source_values<B>(p.srcs[si], qi, wt, va);
          const bool two = si + 1 < p.n_srcs;
          if (two) radix_source_values<B>(p.srcs[si + 1], qi, wt, vb);
#pragma unroll
          for (int u = 0; u < B; u++)
#pragma unroll
            for (int i = 0; i < 4; i++)
              if ((mb >> (4 * u + i)) & 1u) {
                u32x4 w4 = {(uint32_t)(uint64_t)va[u][i], (uint32_t)((uint64_t)va[u][i] >> 32), 0u, 0u};
                if (two) { w4.z = (uint32_t)(uint64_t)vb[u][i]; w4.w = (uint32_t)((uint64_t)vb[u][i] >> 32); }
                *reinterpret_cast<u32x4*>(tp[u][i] + 8 + 8 * si) = w4;
              }
        }
      }
    }
  }
  if (PASS == 1) {
    __syncthreads();
    for (int i = t; i < P; i += PG_BLOCK) p.radix_hist[(int64_t)blockIdx.x * P + i] = s_cnt[i];
  }
}
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_radix_count_kernel(const PgQueryPlan p) { radix_pass_body<1, false>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_radix_scatter_kernel(const PgQueryPlan p) { radix_pass_body<2, false>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_hash_count_kernel(const PgQueryPlan p) { radix_pass_body<1, true>(p); }
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_hash_scatter_kernel(const PgQueryPlan p) { radix_pass_body<2, true>(p); }

// =====================================================================================================================
// Packed radix tuples (PgQueryPlan::radix_packed): when everything the aggregation pass needs from a doc fits 32 bits — the local key
// (radix_shift bits) and, per source, either the (register index, rank) a DISTINCTCOUNTHLL offers (log2m + 5 bits) or the dictId of a
// dictionary-encoded source (its value is looked up in the aggregation pass) — a tuple is ONE dword instead of 16+ bytes: a quarter
// of the bytes written and read back (config 5: 200 M docs x 16 B = 3.2 GB each way → 0.8 GB).
// The scatter stages them per wavefront and bucket in LDS rings of 64 tuples and writes whole, aligned 128-byte lines of 32 tuples:
// whole-line flushes stream at ~5.5 TB/s where scattered 16-byte stores reach 0.35 - 2 TB/s (tools/probes/scatter_bw.hip,
// profiles/r02_scatter_write_probe.txt).  Returning LDS atomics hand out ring positions, eight per lane in flight (one batch of docs)
// before the first is waited for; flushes are decided once per batch, eight lines per step.
// A (workgroup, bucket) range of the tuple area holds whole lines; what a line does not fill carries PG_RADIX_INVALID_KEY.
// dynamic LDS per workgroup: [16 wavefronts][buckets] tail, [16][buckets] head, [16][64 + 8] work list, [16][buckets][64] tuples.
// =====================================================================================================================
#define PG_PK_RING 64u       // tuples per ring (two 128-byte lines)
#define PG_PK_LINE 32u       // tuples per flushed line
// dwords of a wavefront's flush work list: one (bucket, destination) pair per line, and a flush can complete
// floor((buckets x 31 leftovers + 64 lanes x 8 new tuples) / 32) = 78 lines at 64 buckets — 80 pairs (round 2 sized it for 72)
#define PG_PK_WORK 160
struct PackedStage {
  uint32_t* tail;        // this wavefront's [buckets]: ring positions handed out so far
  uint32_t* head;        // this wavefront's [buckets]: ring positions flushed so far (multiple of PG_PK_LINE)
  uint32_t* work;        // this wavefront's work list: (bucket | line-of-ring << 16) per line to flush
  uint32_t* ring;        // this wavefront's [buckets][PG_PK_RING] tuples
  uint32_t* s_cnt;       // workgroup: tuples claimed so far per bucket in the tuple area
  const uint32_t* s_base;
  uint32_t* tuples;
  int buckets;
};
// Writes out every whole line the wavefront's rings hold (pad_partial: also the partial ones, padded — the epilogue).
DEVFN void packed_flush(const PackedStage& S, bool pad_partial, int lane) {
  // lanes 0 .. buckets-1 own one bucket each (buckets <= 64)
  uint32_t lines = 0, hd = 0, avail = 0;
  if (lane < S.buckets) {
    hd = S.head[lane];
    avail = S.tail[lane] - hd;
    if (avail > PG_PK_RING) avail = PG_PK_RING;   // positions beyond the ring are claimed but not written yet
    lines = avail / PG_PK_LINE;
    if (pad_partial && (avail % PG_PK_LINE)) {
      for (uint32_t i = avail; i < (lines + 1) * PG_PK_LINE; i++) S.ring[(size_t)lane * PG_PK_RING + ((hd + i) & (PG_PK_RING - 1u))] = PG_RADIX_INVALID_KEY;
      lines++;
    }
  }
  if (__ballot(lines != 0) == 0) return;
  // exclusive scan of `lines` over the lanes → positions in the work list; the owner claims its lines' slots in the tuple area
  uint32_t x = lines;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, off, 64);
    if (lane >= off) x += y;
  }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  uint32_t at = 0;
  if (lines) at = atomicAdd(&S.s_cnt[lane], lines * PG_PK_LINE);
  for (uint32_t i = 0; i < lines; i++) {
    const uint32_t w = x - lines + i;
    S.work[2 * w] = (uint32_t)lane | ((((hd / PG_PK_LINE) + i) & 1u) << 16);   // which half of the ring
    S.work[2 * w + 1] = S.s_base[lane] + at + i * PG_PK_LINE;                  // destination (in tuples)
  }
  if (lines) S.head[lane] = hd + lines * PG_PK_LINE;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t l0 = 0; l0 < total; l0 += 8) {   // eight lines per step, 16 bytes per lane
    const uint32_t li = l0 + (uint32_t)(lane >> 3);
    if (li < total) {
      const uint32_t e = S.work[2 * li], dst = S.work[2 * li + 1];
      const uint32_t b = e & 0xFFFFu, half = e >> 16;
      const u32x4 v = *reinterpret_cast<const u32x4*>(S.ring + (size_t)b * PG_PK_RING + half * PG_PK_LINE + (uint32_t)(lane & 7) * 4u);
      *reinterpret_cast<u32x4*>(S.tuples + (size_t)dst + (uint32_t)(lane & 7) * 4u) = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
}
// One batch: N tuples per lane (active[j]: the lane has one), positions handed out by N returning LDS adds in flight at once.
template <int N>
DEVFN void packed_emit(const PackedStage& S, const bool (&active)[N], const uint32_t (&bucket)[N], const uint32_t (&tuple)[N], int lane) {
  uint32_t pos[N];
#pragma unroll
  for (int j = 0; j < N; j++) pos[j] = active[j] ? atomicAdd(&S.tail[bucket[j]], 1u) : 0u;
  bool pending[N];
  bool any_pending = false;
#pragma unroll
  for (int j = 0; j < N; j++) {
    pending[j] = active[j];
    if (active[j] && pos[j] - S.head[bucket[j]] < PG_PK_RING) {
      S.ring[(size_t)bucket[j] * PG_PK_RING + (pos[j] & (PG_PK_RING - 1u))] = tuple[j];
      pending[j] = false;
    }
    any_pending |= pending[j];
  }
  packed_flush(S, false, lane);
  while (__ballot(any_pending)) {   // a ring overflowed in this batch (skewed keys): its claimed positions become writable as lines leave
    any_pending = false;
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (pending[j] && pos[j] - S.head[bucket[j]] < PG_PK_RING) {
        S.ring[(size_t)bucket[j] * PG_PK_RING + (pos[j] & (PG_PK_RING - 1u))] = tuple[j];
        pending[j] = false;
      }
      any_pending |= pending[j];
    }
    packed_flush(S, false, lane);
  }
}
// payload of one source for a doc: (register index | rank << log2m) of the value a DISTINCTCOUNTHLL offers, or the dictId
DEVFN uint32_t packed_hll_payload(uint32_t index_rank, int log2m) { return (index_rank & 0xFFFFu) | ((index_rank >> 16) << log2m); }

PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_radix_scatter_packed_kernel(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_cnt[64];
  __shared__ uint32_t s_base[64];
  constexpr int B = 2;
  // 16 wavefronts (up to 32 buckets) or 8 (up to 64: the rings must fit the LDS); the workgroup walks the tiles of count-pass
  // workgroup blockIdx.x — 16 per sweep — whatever its own width, so that the counted ranges are the ones it fills
  const int t = threadIdx.x, lane = t & 63, wave = uniform(t >> 6), n_waves = (int)(blockDim.x >> 6);
  const int P = p.radix_buckets;
  PackedStage S;
  uint32_t* base = reinterpret_cast<uint32_t*>(smem);
  S.tail = base + (size_t)wave * P;
  S.head = base + (size_t)n_waves * P + (size_t)wave * P;
  S.work = base + (size_t)2 * n_waves * P + (size_t)wave * PG_PK_WORK;
  S.ring = base + (size_t)2 * n_waves * P + (size_t)n_waves * PG_PK_WORK + (size_t)wave * P * PG_PK_RING;
  S.s_cnt = s_cnt;
  S.s_base = s_base;
  S.tuples = reinterpret_cast<uint32_t*>(p.radix_tuples);
  S.buckets = P;
  for (int i = t; i < P; i += (int)blockDim.x) {
    s_cnt[i] = 0;
    s_base[i] = p.radix_bucket_start[i] + p.radix_hist[(int64_t)blockIdx.x * P + i];
  }
  for (int i = t; i < 2 * n_waves * P; i += (int)blockDim.x) base[i] = 0;
  __syncthreads();
  const uint32_t local_mask = (1u << p.radix_shift) - 1u;
  const int sweep = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  for (int wt0 = (int)blockIdx.x * PG_WAVES_PER_BLOCK; wt0 < p.n_wtiles; wt0 += sweep)
  for (int wt = wt0 + wave; wt < wt0 + PG_WAVES_PER_BLOCK && wt < p.n_wtiles; wt += n_waves) {
    const int64_t rem = (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    const uint32_t mlin = gptr<uint32_t>(p.match_words)[(int64_t)wt * 64 + lane] & valid_lin_mask(n_valid, lane);
    if (__ballot(mlin != 0) == 0) continue;
    const uint32_t m = lin_to_quad(mlin, lane);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += B) {
      const uint32_t mb = (m >> (4 * k0)) & ((1u << (4 * B)) - 1u);
      if (__ballot(mb != 0) == 0) continue;
      uint32_t qi[B];
#pragma unroll
      for (int u = 0; u < B; u++) qi[u] = ((mb >> (4 * u)) & 0xFu) ? (uint32_t)((k0 + u) * 64 + lane) : 0u;
      uint32_t key[B][4];
      radix_keys_of<B>(p, qi, wt, key);
      uint32_t tuple[B * 4];
#pragma unroll
      for (int u = 0; u < B; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) tuple[4 * u + i] = key[u][i] & local_mask;
      // payload fields: dictIds of the sources, turned into (index, rank) for the HyperLogLog ones
      for (int si = 0; si < p.n_srcs; si++) {
        const PgValueSrc& V = p.srcs[si];
        const int hll_log2m = p.pk_hll[si];   // > 0: the source feeds a DISTINCTCOUNTHLL
        uint32_t d[B][4];
        if (V.col_kind == PG_COL_FIXED_BIT) {
          const GAS uint32_t* tw = packed_wtile_base(V.data, wt, V.bits);
          const uint32_t bits = (uint32_t)V.bits, mask = (1u << V.bits) - 1u;
#pragma unroll
          for (int u = 0; u < B; u++) {
            uint32_t r[8];
            if (bits <= 8) { load_packed_quad<true>(tw, qi[u], bits, r); decode_packed_quad<true>(r, qi[u], bits, mask, d[u]); }
            else { load_packed_quad<false>(tw, qi[u], bits, r); decode_packed_quad<false>(r, qi[u], bits, mask, d[u]); }
          }
          if (hll_log2m > 0 && p.pk_affine[si]) {
            // an arithmetic dictionary (value = base + step x dictId: ids, dense enumerations): the value is computed and hashed here
            // instead of gathering (index, rank) from a cardinality-sized table (1 M entries: ~0.8 ms of L2 misses per 200 M docs)
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++)
                d[u][i] = packed_hll_payload(hll_index_rank_dev(murmur_hash_long_dev(p.pk_base[si] + p.pk_step[si] * (int64_t)d[u][i]), hll_log2m), hll_log2m);
          } else if (hll_log2m > 0) {
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
              for (int i = 0; i < 4; i++) d[u][i] = packed_hll_payload(gptr<uint32_t>(p.pk_lut[si])[d[u][i]], hll_log2m);
          }
        } else {   // raw INT / LONG value offered to a HyperLogLog: hashed here (stream-lib MurmurHash.hashLong)
          int64_t v[B][4];
          radix_source_values<B>(V, qi, wt, v);
#pragma unroll
          for (int u = 0; u < B; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) d[u][i] = packed_hll_payload(hll_index_rank_dev(murmur_hash_long_dev(v[u][i]), hll_log2m), hll_log2m);
        }
        const uint32_t sh = (uint32_t)p.pk_shift[si];
#pragma unroll
        for (int u = 0; u < B; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) tuple[4 * u + i] |= d[u][i] << sh;
      }
      bool active[B * 4];
      uint32_t bucket[B * 4];
#pragma unroll
      for (int u = 0; u < B; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          active[4 * u + i] = ((mb >> (4 * u + i)) & 1u) != 0;
          bucket[4 * u + i] = key[u][i] >> p.radix_shift;
        }
      packed_emit<B * 4>(S, active, bucket, tuple, lane);
    }
  }
  packed_flush(S, true, lane);   // the wavefront's partial lines, padded
  __syncthreads();
  for (int b = wave; b < P; b += n_waves) {   // slots of the workgroup's ranges that no line claimed
    const uint32_t begin = p.radix_hist[(int64_t)blockIdx.x * P + b];
    const uint32_t end = blockIdx.x + 1 < gridDim.x ? p.radix_hist[(int64_t)(blockIdx.x + 1) * P + b]
                                                     : p.radix_bucket_start[b + 1] - p.radix_bucket_start[b];
    for (uint32_t i = s_cnt[b] + (uint32_t)lane; i < end - begin; i += 64) S.tuples[(size_t)s_base[b] + i] = PG_RADIX_INVALID_KEY;
  }
}

// Aggregation pass over packed tuples: work item w = bucket * slices + slice, as pg_radix_aggregate_kernel; four tuples per 16-byte load.
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_radix_aggregate_packed_kernel(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  int64_t* table = reinterpret_cast<int64_t*>(smem);
  const int t = threadIdx.x;
  const uint32_t slots = 1u << p.radix_shift, local_mask = slots - 1u;
  const int n_items = p.radix_buckets * p.radix_slices;
  uint8_t* const aux_lds = reinterpret_cast<uint8_t*>(table + (size_t)p.n_ops * slots);
  uint32_t aux_words = 0;
  for (int x = 0; x < p.n_aux; x++) aux_words += (slots * (uint32_t)p.aux[x].stride) >> 2;
  const uint32_t* tuples = reinterpret_cast<const uint32_t*>(p.radix_tuples);
  for (int w = (int)blockIdx.x; w < n_items; w += (int)gridDim.x) {
    const int b = w / p.radix_slices, sl = w % p.radix_slices;
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < slots; i += PG_BLOCK) table[(size_t)o * slots + i] = ident;
    }
    for (uint32_t i = t; i < aux_words; i += PG_BLOCK) reinterpret_cast<uint32_t*>(aux_lds)[i] = 0u;
    __syncthreads();
    // the bucket's range is a whole number of 32-tuple lines: slices take whole lines, threads 16-byte pieces
    const uint32_t start = p.radix_bucket_start[b], total_lines = (p.radix_bucket_start[b + 1] - start) / PG_PK_LINE;
    const uint32_t per = (total_lines + (uint32_t)p.radix_slices - 1u) / (uint32_t)p.radix_slices;
    const uint32_t lo = start + (uint32_t)sl * per * PG_PK_LINE;
    uint32_t hi = lo + per * PG_PK_LINE;
    if (hi > start + total_lines * PG_PK_LINE) hi = start + total_lines * PG_PK_LINE;
    constexpr int U = 2;   // 16-byte pieces per thread in flight
    for (uint32_t i0 = lo; i0 < hi; i0 += PG_BLOCK * 4 * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = i0 + ((uint32_t)u * PG_BLOCK + (uint32_t)t) * 4u;
        const u32x4 inv = {PG_RADIX_INVALID_KEY, PG_RADIX_INVALID_KEY, PG_RADIX_INVALID_KEY, PG_RADIX_INVALID_KEY};
        v[u] = i < hi ? ldnt((const GAS u32x4*)gptr<uint32_t>(tuples + i)) : inv;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t tp[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (tp[e] == PG_RADIX_INVALID_KEY) continue;
          const uint32_t k = tp[e] & local_mask;
          for (int o = 0; o < p.n_ops; o++) {
            const PgAccOp op = p.ops[o];
            int64_t* acc = table + (size_t)o * slots + k;
            if (op.src < 0) { atomicAdd(reinterpret_cast<unsigned long long*>(acc), 1ULL); continue; }   // COUNT (MIN(docId) plans are not packed)
            const PgValueSrc& V = p.srcs[op.src];
            const uint32_t d = (tp[e] >> p.pk_shift[op.src]) & ((1u << p.pk_bits[op.src]) - 1u);   // dictId
            if (V.val_type == PG_V_I32) acc_from_int(acc, op, (int64_t)(int32_t)gptr<uint32_t>(V.dict)[d]);
            else if (V.val_type == PG_V_I64) acc_from_int(acc, op, (int64_t)gptr<uint64_t>(V.dict)[d]);
            else if (V.val_type == PG_V_F32) acc_from_double(acc, op, (double)__uint_as_float(gptr<uint32_t>(V.dict)[d]), V.fx_q);
            else acc_from_double(acc, op, __longlong_as_double((int64_t)gptr<uint64_t>(V.dict)[d]), V.fx_q);
          }
          uint32_t aux_off = 0;
          for (int x = 0; x < p.n_aux; x++) {
            const PgAuxOp& A = p.aux[x];
            const uint32_t f = (tp[e] >> p.pk_shift[A.src]) & ((1u << p.pk_bits[A.src]) - 1u);
            hll_update(aux_lds + aux_off + (size_t)k * (uint32_t)A.stride, f & ((1u << A.log2m) - 1u), f >> A.log2m);
            aux_off += slots * (uint32_t)A.stride;
          }
        }
      }
    }
    __syncthreads();
    int64_t* out = p.partials + (int64_t)w * p.n_ops * slots;
    for (int64_t i = t; i < (int64_t)p.n_ops * slots; i += PG_BLOCK) out[i] = table[i];
    {
      uint32_t aux_off = 0;
      for (int x = 0; x < p.n_aux; x++) {
        const uint32_t n_words = (slots * (uint32_t)p.aux[x].stride) >> 2;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(aux_lds + aux_off);
        uint32_t* dst = p.aux[x].base + (int64_t)w * n_words;
        for (uint32_t i = t; i < n_words; i += PG_BLOCK) dst[i] = src[i];
        aux_off += n_words << 2;
      }
    }
    __syncthreads();
  }
}

// Per-bucket hash aggregation: the bucket's tuples go into an open-addressing table in LDS — keys[cap] claimed with a 64-bit
// compare-and-swap (linear probing), accumulators [n_ops][cap] updated with LDS atomics — whose occupied slots are then appended
// to the result (one global atomic per wavefront).  A bucket with more distinct keys than slots raises the overflow flag.
#define PG_HASH_MAX_ITERS 16   // hash_cap <= 16 384 slots = 16 sweeps of the 1024-thread workgroup
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_hash_aggregate_kernel(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_local;
  __shared__ unsigned long long s_gbase;
  const int t = threadIdx.x, lane = t & 63;
  const uint32_t cap = (uint32_t)p.hash_cap, cmask = cap - 1u;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  int64_t* table = reinterpret_cast<int64_t*>(smem) + cap;
  const uint32_t stride = (uint32_t)p.radix_stride;
  const unsigned long long kEmpty = ~0ULL;
  for (int b = (int)blockIdx.x; b < p.radix_buckets; b += (int)gridDim.x) {
    for (uint32_t i = t; i < cap; i += PG_BLOCK) keys[i] = kEmpty;
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < cap; i += PG_BLOCK) table[(size_t)o * cap + i] = ident;
    }
    __syncthreads();
    const uint32_t lo = p.radix_bucket_start[b], hi = p.radix_bucket_start[b + 1];
    for (uint32_t i = lo + (uint32_t)t; i < hi; i += PG_BLOCK) {
      const GAS uint8_t* tp = gptr<uint8_t>(p.radix_tuples + (size_t)i * stride);
      const u32x4 h = *(const GAS u32x4*)tp;
      const unsigned long long key = ((unsigned long long)h.y << 32) | h.x;
      if (key == kEmpty) { p.hash_out_count[1] = 2; continue; }   // a raw LONG group key of Long.MAX_VALUE collides with the empty marker
      uint32_t slot = (uint32_t)(radix_mix64(key) >> 16) & cmask;
      bool found = false;
      for (uint32_t probes = 0; probes < cap; probes++) {
        unsigned long long cur = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == kEmpty) cur = atomicCAS(&keys[slot], kEmpty, key);
        if (cur == kEmpty || cur == key) { found = true; break; }
        slot = (slot + 1u) & cmask;
      }
      if (!found) { p.hash_out_count[1] = 1; continue; }   // table full: more distinct keys than slots
      for (int o = 0; o < p.n_ops; o++) {
        const PgAccOp op = p.ops[o];
        int64_t* acc = table + (size_t)o * cap + slot;
        if (op.src < 0) {
          if (op.fn == PG_ACC_COUNT) atomicAdd(reinterpret_cast<unsigned long long*>(acc), 1ULL);
          else atomicMin(reinterpret_cast<long long*>(acc), (long long)h.z);   // MIN(docId)
          continue;
        }
        const int64_t v = *(const GAS int64_t*)(tp + 16 + 8 * op.src);
        if (op.is_float == PG_ACCV_DOUBLE || op.is_float == PG_ACCV_FIXED_DIGIT) acc_from_double(acc, op, __longlong_as_double(v), p.srcs[op.src].fx_q);
        else acc_from_int(acc, op, v);
      }
    }
    __syncthreads();
    // append the occupied slots: local positions from an LDS counter, ONE global atomic per bucket (a counter shared by every
    // wavefront of the chip serialises at ~12 ns per atomic: 131 k of them were 1.6 ms)
    if (t == 0) s_local = 0;
    __syncthreads();
    uint32_t my_pos[PG_HASH_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < PG_HASH_MAX_ITERS; it++) {
      const uint32_t i = (uint32_t)it * PG_BLOCK + (uint32_t)t;
      const bool occ = i < cap && keys[i] != kEmpty;
      const unsigned long long ball = __ballot(occ);
      uint32_t wbase = 0;
      if (ball && lane == 0) wbase = atomicAdd(&s_local, (uint32_t)__popcll(ball));
      wbase = (uint32_t)__shfl((int)wbase, 0, 64);
      my_pos[it] = occ ? wbase + (uint32_t)__popcll(ball & ((1ULL << lane) - 1ULL)) : 0xFFFFFFFFu;
    }
    __syncthreads();
    if (t == 0) s_gbase = atomicAdd(&p.hash_out_count[0], (unsigned long long)s_local);
    __syncthreads();
    const unsigned long long gbase = s_gbase;
#pragma unroll
    for (int it = 0; it < PG_HASH_MAX_ITERS; it++) {
      if (my_pos[it] != 0xFFFFFFFFu) {
        const uint32_t i = (uint32_t)it * PG_BLOCK + (uint32_t)t;
        const unsigned long long pos = gbase + my_pos[it];
        if ((int64_t)pos < p.hash_out_cap) {
          p.hash_out_keys[pos] = (int64_t)keys[i];
          for (int o = 0; o < p.n_ops; o++) p.hash_out_acc[(int64_t)o * p.hash_out_cap + (int64_t)pos] = table[(size_t)o * cap + i];
        } else {
          p.hash_out_count[1] = 1;
        }
      }
    }
    __syncthreads();
  }
}

// Counts → offsets.  Step 1, one wavefront per bucket: hist[wg][b] becomes the exclusive prefix over the workgroups (lanes own
// consecutive workgroups), total[b] the bucket's size.  Step 2, one wavefront: bucket_start = exclusive scan of the totals.
// (The scatter pass adds bucket_start[b] when it loads its bases.)
// stage > 0 (packed tuples): a (workgroup, bucket) range holds whole lines of `stage` tuples — each of the scatter workgroup's
// `waves` wavefronts may end on a partial one — so c tuples reserve (c / stage + waves) * stage slots; unused slots carry PG_RADIX_INVALID_KEY.
DEVFN uint32_t radix_capacity(uint32_t c, int stage, int waves) {
  if (stage <= 0 || c == 0) return c;
  return (c / (uint32_t)stage + (uint32_t)waves) * (uint32_t)stage;
}
PG_KERNEL __global__ void __launch_bounds__(1024) pg_radix_offsets_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ bucket_total,
                                                                           int n_wg, int n_buckets, int stage, int stage_waves) {
  const int lane = threadIdx.x & 63;
  const int b = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 6);
  if (b >= n_buckets) return;
  const int per_lane = (n_wg + 63) / 64;
  uint32_t run_total = 0;
  // lanes own per_lane consecutive workgroups
  uint32_t mine = 0;
  for (int k = 0; k < per_lane; k++) {
    const int w = lane * per_lane + k;
    if (w < n_wg) mine += radix_capacity(hist[(int64_t)w * n_buckets + b], stage, stage_waves);
  }
  uint32_t x = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, off, 64);
    if (lane >= off) x += y;
  }
  run_total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  uint32_t run = x - mine;
  for (int k = 0; k < per_lane; k++) {
    const int w = lane * per_lane + k;
    if (w < n_wg) {
      const uint32_t c = radix_capacity(hist[(int64_t)w * n_buckets + b], stage, stage_waves);
      hist[(int64_t)w * n_buckets + b] = run;
      run += c;
    }
  }
  if (lane == 0) bucket_total[b] = run_total;
}
PG_KERNEL __global__ void __launch_bounds__(64) pg_radix_bucket_scan_kernel(const uint32_t* __restrict__ bucket_total,
                                                                             uint32_t* __restrict__ bucket_start, int n_buckets) {
  const int lane = threadIdx.x;
  const int per_lane = (n_buckets + 63) / 64;
  uint32_t mine = 0;
  for (int k = 0; k < per_lane; k++) {
    const int b = lane * per_lane + k;
    if (b < n_buckets) mine += bucket_total[b];
  }
  uint32_t x = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, off, 64);
    if (lane >= off) x += y;
  }
  uint32_t run = x - mine;
  for (int k = 0; k < per_lane; k++) {
    const int b = lane * per_lane + k;
    if (b < n_buckets) { const uint32_t c = bucket_total[b]; bucket_start[b] = run; run += c; }
  }
  if (lane == 63) bucket_start[n_buckets] = run;
}

// Per-bucket aggregation: work item w = bucket * slices + slice aggregates its share of the bucket's tuples into an LDS table
// [n_ops][2^radix_shift] and flushes it to partials[w].
PG_KERNEL __global__ void __launch_bounds__(PG_BLOCK) pg_radix_aggregate_kernel(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  int64_t* table = reinterpret_cast<int64_t*>(smem);
  const int t = threadIdx.x;
  const uint32_t slots = 1u << p.radix_shift;
  const int n_items = p.radix_buckets * p.radix_slices;
  // DISTINCTCOUNTHLL over INT / LONG values: the bucket's registers [slots][2^log2m] live behind the accumulator table; the
  // tuple carries the doc's value, hashed here (stream-lib MurmurHash.hashLong, as hll.offer(value) does per doc)
  uint8_t* const aux_lds = reinterpret_cast<uint8_t*>(table + (size_t)p.n_ops * slots);
  uint32_t aux_words = 0;
  for (int x = 0; x < p.n_aux; x++) aux_words += (slots * (uint32_t)p.aux[x].stride) >> 2;
  for (int w = (int)blockIdx.x; w < n_items; w += (int)gridDim.x) {
    const int b = w / p.radix_slices, sl = w % p.radix_slices;
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < slots; i += PG_BLOCK) table[(size_t)o * slots + i] = ident;
    }
    for (uint32_t i = t; i < aux_words; i += PG_BLOCK) reinterpret_cast<uint32_t*>(aux_lds)[i] = 0u;
    __syncthreads();
    const uint32_t start = p.radix_bucket_start[b], total = p.radix_bucket_start[b + 1] - start;
    const uint32_t per = (total + (uint32_t)p.radix_slices - 1u) / (uint32_t)p.radix_slices;
    const uint32_t lo = start + (uint32_t)sl * per;
    const uint32_t hi = lo + per < start + total ? lo + per : start + total;
    constexpr int U = 4;   // tuples per thread in flight: the loop is a chain of dependent loads otherwise
    const uint32_t stride = (uint32_t)p.radix_stride;
    for (uint32_t i0 = lo; i0 < hi; i0 += PG_BLOCK * U) {
      uint32_t k[U], docid[U];
      const GAS uint8_t* tp[U];
      bool on[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = i0 + (uint32_t)u * PG_BLOCK + (uint32_t)t;
        on[u] = i < hi;
        tp[u] = gptr<uint8_t>(p.radix_tuples + (size_t)(on[u] ? i : lo) * stride);
        const u32x2 h = *(const GAS u32x2*)tp[u];
        k[u] = h.x;
        docid[u] = h.y;
        if (h.x == PG_RADIX_INVALID_KEY) { on[u] = false; k[u] = 0; }   // padding of a staged flush
      }
      for (int o = 0; o < p.n_ops; o++) {
        const PgAccOp op = p.ops[o];
        int64_t* base = table + (size_t)o * slots;
        if (op.src < 0) {
          if (op.fn == PG_ACC_COUNT) {
#pragma unroll
            for (int u = 0; u < U; u++) if (on[u]) atomicAdd(reinterpret_cast<unsigned long long*>(base + k[u]), 1ULL);
          } else {   // MIN(docId): which groups numGroupsLimit admits
#pragma unroll
            for (int u = 0; u < U; u++) if (on[u]) atomicMin(reinterpret_cast<long long*>(base + k[u]), (long long)docid[u]);
          }
          continue;
        }
        int64_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = *(const GAS int64_t*)(tp[u] + 8 + 8 * op.src);
#pragma unroll
        for (int u = 0; u < U; u++)
          if (on[u]) {
            if (op.is_float == PG_ACCV_DOUBLE || op.is_float == PG_ACCV_FIXED_DIGIT) acc_from_double(base + k[u], op, __longlong_as_double(v[u]), p.srcs[op.src].fx_q);
            else acc_from_int(base + k[u], op, v[u]);
          }
      }
      uint32_t aux_off = 0;
      for (int x = 0; x < p.n_aux; x++) {
        const PgAuxOp& A = p.aux[x];
        int64_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = *(const GAS int64_t*)(tp[u] + 8 + 8 * A.src);
#pragma unroll
        for (int u = 0; u < U; u++)
          if (on[u]) {
            const uint32_t ir = hll_index_rank_dev(murmur_hash_long_dev(v[u]), A.log2m);
            hll_update(aux_lds + aux_off + (size_t)k[u] * (uint32_t)A.stride, ir & 0xFFFFu, ir >> 16);
          }
        aux_off += slots * (uint32_t)A.stride;
      }
    }
    __syncthreads();
    int64_t* out = p.partials + (int64_t)w * p.n_ops * slots;
    for (int64_t i = t; i < (int64_t)p.n_ops * slots; i += PG_BLOCK) out[i] = table[i];
    {
      uint32_t aux_off = 0;
      for (int x = 0; x < p.n_aux; x++) {   // this work item's registers → its partial, merged by pg_radix_reduce_aux_kernel
        const uint32_t n_words = (slots * (uint32_t)p.aux[x].stride) >> 2;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(aux_lds + aux_off);
        uint32_t* dst = p.aux[x].base + (int64_t)w * n_words;
        for (uint32_t i = t; i < n_words; i += PG_BLOCK) dst[i] = src[i];
        aux_off += n_words << 2;
      }
    }
    __syncthreads();
  }
}

// registers of group g = bytewise max over the slices of g's bucket
PG_KERNEL __global__ void __launch_bounds__(256) pg_radix_reduce_aux_kernel(const uint32_t* __restrict__ partials, uint32_t* __restrict__ out,
                                                                              int slices, int64_t bucket_words, int64_t n_words, int bitwise_or) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const int64_t b = w / bucket_words, l = w % bucket_words;
  uint32_t acc = 0;
  if (bitwise_or) { for (int sl = 0; sl < slices; sl++) acc |= partials[(b * slices + sl) * bucket_words + l]; }   // dictId sets
  else { for (int sl = 0; sl < slices; sl++) acc = bytemax4(acc, partials[(b * slices + sl) * bucket_words + l]); }
  out[w] = acc;
}

// out[op][g] = combine over the slices of g's bucket
PG_KERNEL __global__ void __launch_bounds__(256) pg_radix_reduce_kernel(const int64_t* __restrict__ partials, int64_t* __restrict__ out,
                                                                          int n_ops, int n_groups, int radix_shift, int slices,
                                                                          const PgAccOp* __restrict__ ops) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_ops * n_groups) return;
  const int o = (int)(i / n_groups), g = (int)(i % n_groups);
  const int b = g >> radix_shift, l = g & ((1 << radix_shift) - 1);
  const int64_t slots = (int64_t)1 << radix_shift;
  const PgAccOp op = ops[o];
  const int kind = (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) ? 0 : ((op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) ? 1 : (op.fn == PG_ACC_MIN ? 2 : 3));
  int64_t acc = pg_acc_identity(op.fn, op.is_float);
  for (int sl = 0; sl < slices; sl++) {
    const int64_t v = partials[((int64_t)(b * slices + sl) * n_ops + o) * slots + l];
    if (kind == 0) acc = __double_as_longlong(__longlong_as_double(acc) + __longlong_as_double(v));
    else if (kind == 1) acc += v;
    else if (kind == 2) acc = v < acc ? v : acc;
    else acc = v > acc ? v : acc;
  }
  out[i] = acc;
}

// Range-partitioned partials: workgroup b holds [n_ops][part_groups] for range (b >> 3) % n_parts.  64 output slots (op, group)
// per workgroup — neighbouring lanes read neighbouring slots of the same workgroup's table — and its four wavefronts split the
// workgroups that own the slots' range, eight loads in flight each.
PG_KERNEL __global__ void __launch_bounds__(256) pg_reduce_parts_kernel(const int64_t* __restrict__ partials,
                                                                          int64_t* __restrict__ out, int n_wg, int n_ops,
                                                                          int n_groups, int n_parts, int part_groups,
                                                                          const PgAccOp* __restrict__ ops) {
  const int64_t n_out = (int64_t)n_ops * n_groups;
  __shared__ int64_t s_acc[4][64];
  const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;   // wavefront q takes the workgroups j = q (mod 4) of the slot's range
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const bool live = i < n_out;
  const int o = live ? (int)(i / n_groups) : 0, g = live ? (int)(i % n_groups) : 0;
  const int range = g / part_groups, l = g % part_groups;
  const PgAccOp op = ops[o];
  const int kind = (op.fn == PG_ACC_SUM && op.is_float == PG_ACCV_DOUBLE) ? 0 : ((op.fn == PG_ACC_COUNT || op.fn == PG_ACC_SUM) ? 1 : (op.fn == PG_ACC_MIN ? 2 : 3));
  int64_t acc = pg_acc_identity(op.fn, op.is_float);
  auto combine = [&](int64_t a, int64_t b) -> int64_t {
    if (kind == 0) return __double_as_longlong(__longlong_as_double(a) + __longlong_as_double(b));
    if (kind == 1) return a + b;
    if (kind == 2) return b < a ? b : a;
    return b > a ? b : a;
  };
  const int64_t wg_stride = (int64_t)n_ops * part_groups;
  const int64_t* src = partials + (int64_t)o * part_groups + l;
  const int per_range = (n_wg >> 3) / n_parts;   // launch_shape: the grid is 8 x per_range x n_parts
  for (int j = quarter; j < per_range; j += 4) {
    const int64_t w0 = 8 * ((int64_t)range + (int64_t)n_parts * j);
    int64_t v[8];
#pragma unroll
    for (int x = 0; x < 8; x++) v[x] = src[(w0 + x) * wg_stride];
#pragma unroll
    for (int x = 0; x < 8; x++) acc = combine(acc, v[x]);
  }
  s_acc[quarter][lane] = acc;
  __syncthreads();
  if (quarter == 0 && live) out[i] = combine(combine(s_acc[0][lane], s_acc[1][lane]), combine(s_acc[2][lane], s_acc[3][lane]));
}

// Merges the workgroups' LDS-resident DISTINCTCOUNT / HLL partials: out[w] = OR (sets) or per-byte max (registers) over n_wg.
PG_KERNEL __global__ void __launch_bounds__(256) pg_reduce_aux_kernel(const uint32_t* __restrict__ partials, uint32_t* __restrict__ out,
                                                                        int n_wg, int64_t n_words, int bytewise_max) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint32_t acc = 0;
  for (int b = 0; b < n_wg; b++) {
    const uint32_t v = partials[(int64_t)b * n_words + w];
    acc = bytewise_max ? bytemax4(acc, v) : (acc | v);
  }
  out[w] = acc;
}

PG_KERNEL __global__ void __launch_bounds__(256) pg_fill_i64_kernel(int64_t* dst, int64_t n_per_op, int n_ops,
                                                                      const PgAccOp* __restrict__ ops) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_per_op * n_ops) return;
  const PgAccOp op = ops[i / n_per_op];
  dst[i] = pg_acc_identity(op.fn, op.is_float);
}

// K4: match words → ascending docIds (DocIdSetOperator).  One 256-thread workgroup per 16 384-doc tile; tile_offsets =
// exclusive prefix of the per-tile match counts.
PG_KERNEL __global__ void __launch_bounds__(PG_TILE_WORDS) pg_expand_docids_kernel(const uint64_t* __restrict__ words,
                                                                                     const int64_t* __restrict__ tile_offsets,
                                                                                     int32_t* __restrict__ out, int n_tiles) {
  __shared__ uint32_t s_scan[PG_TILE_WORDS];
  const int t = threadIdx.x;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint64_t w = words[(int64_t)tile * PG_TILE_WORDS + t];
    uint32_t c = (uint32_t)__popcll(w);
    s_scan[t] = c;
    __syncthreads();
    for (int off = 1; off < PG_TILE_WORDS; off <<= 1) {   // Hillis–Steele inclusive scan
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
