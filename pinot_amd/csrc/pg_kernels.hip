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
// Hardware mapping: the path is HBM-bound integer/bitmap work (no MFMA).  One persistent workgroup per tile stream;
// every column is read once with 16-byte (raw columns) or dword-pair (bit-packed columns) loads that are contiguous
// across the wavefront; match bits are assembled with DPP row operations (no LDS round trip) into 64-bit words held in
// LDS; group accumulators live in LDS (ds_add_u64 / ds_max_i64 / ds_add_f64) and are flushed once per workgroup.
#include <hip/hip_runtime.h>

#include "pg_device.h"

#define DEVFN __device__ __forceinline__

DEVFN uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
DEVFN uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

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

DEVFN uint64_t tile_valid_word(int32_t num_docs, int64_t doc_base) {
  int64_t rem = (int64_t)num_docs - doc_base;
  if (rem >= 64) return ~0ULL;
  if (rem <= 0) return 0ULL;
  return (1ULL << rem) - 1ULL;
}

// ---- bit-packed (FixedBitSVForwardIndexReaderV2) extraction: 4 consecutive docs starting at doc0 (multiple of 4) ----
DEVFN void extract4(const uint8_t* __restrict__ data, int64_t doc0, int bits, uint32_t out[4]) {
  const uint32_t* __restrict__ w = reinterpret_cast<const uint32_t*>(data);
  const uint32_t mask = (1u << bits) - 1u;
  if (bits <= 8) {
    int64_t bitpos = doc0 * bits;
    int64_t di = bitpos >> 5;
    int sh = (int)(bitpos & 31);
    uint64_t win = ((uint64_t)bswap32(w[di]) << 32) | (uint64_t)bswap32(w[di + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = (uint32_t)(win >> (64 - sh - (i + 1) * bits)) & mask;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int64_t bitpos = (doc0 + i) * bits;
      int64_t di = bitpos >> 5;
      int sh = (int)(bitpos & 31);
      uint64_t win = ((uint64_t)bswap32(w[di]) << 32) | (uint64_t)bswap32(w[di + 1]);
      out[i] = (uint32_t)(win >> (64 - sh - bits)) & mask;
    }
  }
}

// ---- predicate evaluation ---------------------------------------------------------------------------------------------
template <int PK>
DEVFN bool pred_dict(const PgScanLeaf& L, uint32_t d) {
  if (PK == PG_P_RANGE) return (int64_t)d >= L.lo && (int64_t)d <= L.hi;
  return (L.lut[d >> 5] >> (d & 31)) & 1u;
}
template <int PK>
DEVFN bool pred_i64(const PgScanLeaf& L, int64_t v) {
  if (PK == PG_P_RANGE) return v >= L.lo && v <= L.hi;
  const int64_t* s = reinterpret_cast<const int64_t*>(L.set_values);
  bool hit = false;
  for (int i = 0; i < L.n_set; i++) hit |= (s[i] == v);
  return hit != (L.exclusive != 0);
}
template <int PK>
DEVFN bool pred_f64(const PgScanLeaf& L, double v) {
  if (PK == PG_P_RANGE) return v >= __longlong_as_double(L.lo) && v <= __longlong_as_double(L.hi);
  const double* s = reinterpret_cast<const double*>(L.set_values);
  bool hit = false;
  for (int i = 0; i < L.n_set; i++) hit |= (s[i] == v);
  return hit != (L.exclusive != 0);
}

// Evaluates the predicate for the 4 docs of a quad; returns a nibble.
template <int CK, int VT, int PK>
DEVFN uint32_t eval_quad(const PgScanLeaf& L, int64_t doc0) {
  uint32_t r = 0;
  if (CK == PG_COL_FIXED_BIT) {
    uint32_t d[4];
    extract4(L.data, doc0, L.bits, d);
#pragma unroll
    for (int i = 0; i < 4; i++) r |= (uint32_t)pred_dict<PK>(L, d[i]) << i;
  } else if (CK == PG_COL_RAW32) {
    uint4 v = *reinterpret_cast<const uint4*>(L.data + doc0 * 4);
    uint32_t x[4] = {bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bool m = (VT == PG_V_I32) ? pred_i64<PK>(L, (int64_t)(int32_t)x[i]) : pred_f64<PK>(L, (double)__uint_as_float(x[i]));
      r |= (uint32_t)m << i;
    }
  } else {
    const uint4* p = reinterpret_cast<const uint4*>(L.data + doc0 * 8);
    uint4 a = p[0], b = p[1];
    uint64_t x[4] = {((uint64_t)bswap32(a.x) << 32) | bswap32(a.y), ((uint64_t)bswap32(a.z) << 32) | bswap32(a.w),
                     ((uint64_t)bswap32(b.x) << 32) | bswap32(b.y), ((uint64_t)bswap32(b.z) << 32) | bswap32(b.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bool m = (VT == PG_V_I64) ? pred_i64<PK>(L, (int64_t)x[i]) : pred_f64<PK>(L, __longlong_as_double((int64_t)x[i]));
      r |= (uint32_t)m << i;
    }
  }
  return r;
}

// Scan leaf over one tile.  MASKED: AND into `words` in place, evaluating only quads with candidates.
template <int CK, int VT, int PK, bool MASKED>
DEVFN uint32_t scan_tile(const PgScanLeaf& L, uint32_t* __restrict__ words32, int64_t tile_base, int32_t n_valid) {
  const int t = threadIdx.x;
  const int sh = (t & 7) * 4;
  uint32_t n_cand = 0;
  constexpr int U = 4;
  for (int q0 = t; q0 < PG_TILE_QUADS; q0 += PG_BLOCK * U) {
    uint32_t cand[U], res[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int q = q0 + u * PG_BLOCK;
      int nv = n_valid - 4 * q;
      uint32_t vn = nv >= 4 ? 0xFu : (nv <= 0 ? 0u : ((1u << nv) - 1u));
      cand[u] = MASKED ? ((words32[q >> 3] >> sh) & vn) : vn;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      int q = q0 + u * PG_BLOCK;
      res[u] = 0;
      if (cand[u]) res[u] = eval_quad<CK, VT, PK>(L, tile_base + 4 * (int64_t)q) & cand[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      int q = q0 + u * PG_BLOCK;
      if (MASKED) n_cand += __popc(cand[u]);
      uint32_t x = or_reduce8(res[u] << sh);
      if ((t & 7) == 0) words32[q >> 3] = x;
    }
  }
  return n_cand;
}

template <bool MASKED>
DEVFN uint32_t scan_dispatch(const PgScanLeaf& L, uint32_t* words32, int64_t tile_base, int32_t n_valid) {
  if (L.col_kind == PG_COL_FIXED_BIT) {
    if (L.pred_kind == PG_P_RANGE) return scan_tile<PG_COL_FIXED_BIT, PG_V_I32, PG_P_RANGE, MASKED>(L, words32, tile_base, n_valid);
    return scan_tile<PG_COL_FIXED_BIT, PG_V_I32, PG_P_DICT_LUT, MASKED>(L, words32, tile_base, n_valid);
  }
  if (L.col_kind == PG_COL_RAW32) {
    if (L.val_type == PG_V_I32) {
      if (L.pred_kind == PG_P_RANGE) return scan_tile<PG_COL_RAW32, PG_V_I32, PG_P_RANGE, MASKED>(L, words32, tile_base, n_valid);
      return scan_tile<PG_COL_RAW32, PG_V_I32, PG_P_SET, MASKED>(L, words32, tile_base, n_valid);
    }
    if (L.pred_kind == PG_P_RANGE) return scan_tile<PG_COL_RAW32, PG_V_F32, PG_P_RANGE, MASKED>(L, words32, tile_base, n_valid);
    return scan_tile<PG_COL_RAW32, PG_V_F32, PG_P_SET, MASKED>(L, words32, tile_base, n_valid);
  }
  if (L.val_type == PG_V_I64) {
    if (L.pred_kind == PG_P_RANGE) return scan_tile<PG_COL_RAW64, PG_V_I64, PG_P_RANGE, MASKED>(L, words32, tile_base, n_valid);
    return scan_tile<PG_COL_RAW64, PG_V_I64, PG_P_SET, MASKED>(L, words32, tile_base, n_valid);
  }
  if (L.pred_kind == PG_P_RANGE) return scan_tile<PG_COL_RAW64, PG_V_F64, PG_P_RANGE, MASKED>(L, words32, tile_base, n_valid);
  return scan_tile<PG_COL_RAW64, PG_V_F64, PG_P_SET, MASKED>(L, words32, tile_base, n_valid);
}

// ---- posting leaf: OR the leaf's RoaringBitmap containers that intersect the tile ----------------------------------------
DEVFN void postings_tile(const PgPostingLeaf& L, uint64_t* __restrict__ dst, int tile, uint64_t valid) {
  const int t = threadIdx.x;
  const int chunk = tile / PG_TILES_PER_CHUNK;
  const int sub = tile % PG_TILES_PER_CHUNK;
  const uint32_t cs = L.chunk_start[chunk], ce = L.chunk_start[chunk + 1];
  uint64_t acc = 0;
  bool scatter = false;
  for (uint32_t e = cs; e < ce; e++) {
    const PgContainer c = L.descs[L.chunk_desc[e]];
    if (c.type == 1) {
      acc |= reinterpret_cast<const uint64_t*>(L.containers + c.offset)[sub * PG_TILE_WORDS + t];
    } else {
      scatter = true;
    }
  }
  if (scatter) {  // workgroup-uniform: array / run containers set bits with LDS atomics
    dst[t] = acc;
    __syncthreads();
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
    const uint32_t lo = (uint32_t)sub * PG_TILE_DOCS, hi = lo + PG_TILE_DOCS;  // low-16 range of this tile
    for (uint32_t e = cs; e < ce; e++) {
      const PgContainer c = L.descs[L.chunk_desc[e]];
      if (c.type == 0) {
        const uint16_t* vals = reinterpret_cast<const uint16_t*>(L.containers + c.offset);
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
        const uint16_t* runs = reinterpret_cast<const uint16_t*>(L.containers + c.offset);
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
  // first range whose hi >= tile_base (ranges ascending, disjoint)
  int a = 0, b = L.n;
  while (a < b) {
    int m = (a + b) >> 1;
    if ((int64_t)L.hi[m] < tile_base) a = m + 1; else b = m;
  }
  uint64_t acc = 0;
  for (int r = a; r < L.n; r++) {
    int64_t lo = L.lo[r], hi = L.hi[r];
    if (lo > tile_end) break;
    if (hi < wb || lo > we) continue;
    int64_t s = lo > wb ? lo - wb : 0, e = hi < we ? hi - wb : 63;
    uint64_t m = (~0ULL << s) & (~0ULL >> (63 - e));
    acc |= m;
  }
  dst[t] = acc & valid;
}

// ---- accumulator updates ----------------------------------------------------------------------------------------------
DEVFN int64_t f64_order_key(double v) {  // order-preserving map double → int64 (for MIN/MAX via integer atomics)
  int64_t b = __double_as_longlong(v);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}

template <typename P>
DEVFN void acc_update(P* slot, int fn, int is_float, int64_t iv, double fv) {
  if (fn == PG_ACC_COUNT) {
    atomicAdd(reinterpret_cast<unsigned long long*>(slot), 1ULL);
  } else if (fn == PG_ACC_SUM) {
    if (is_float) atomicAdd(reinterpret_cast<double*>(slot), fv);
    else atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)iv);
  } else {
    if (is_float) {
      if (fv != fv) return;  // Java: NaN > x and NaN < x are false → NaN never replaces the holder
      iv = f64_order_key(fv);
    }
    if (fn == PG_ACC_MIN) atomicMin(reinterpret_cast<long long*>(slot), (long long)iv);
    else atomicMax(reinterpret_cast<long long*>(slot), (long long)iv);
  }
}

// loads the 4 values of a quad from a metric source as (int64, double) pairs
DEVFN void load_values4(const PgValueSrc& S, int64_t doc0, uint32_t nib, int64_t iv[4], double fv[4]) {
  if (S.col_kind == PG_COL_FIXED_BIT) {
    uint32_t d[4];
    extract4(S.data, doc0, S.bits, d);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!((nib >> i) & 1)) { iv[i] = 0; fv[i] = 0; continue; }
      switch (S.val_type) {
        case PG_V_I32: iv[i] = reinterpret_cast<const int32_t*>(S.dict)[d[i]]; fv[i] = (double)iv[i]; break;
        case PG_V_I64: iv[i] = reinterpret_cast<const int64_t*>(S.dict)[d[i]]; fv[i] = (double)iv[i]; break;
        case PG_V_F32: fv[i] = (double)reinterpret_cast<const float*>(S.dict)[d[i]]; iv[i] = 0; break;
        default: fv[i] = reinterpret_cast<const double*>(S.dict)[d[i]]; iv[i] = 0; break;
      }
    }
  } else if (S.col_kind == PG_COL_RAW32) {
    uint4 v = *reinterpret_cast<const uint4*>(S.data + doc0 * 4);
    uint32_t x[4] = {bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (S.val_type == PG_V_I32) { iv[i] = (int32_t)x[i]; fv[i] = (double)iv[i]; }
      else { fv[i] = (double)__uint_as_float(x[i]); iv[i] = 0; }
    }
  } else {
    const uint4* p = reinterpret_cast<const uint4*>(S.data + doc0 * 8);
    uint4 a = p[0], b = p[1];
    uint64_t x[4] = {((uint64_t)bswap32(a.x) << 32) | bswap32(a.y), ((uint64_t)bswap32(a.z) << 32) | bswap32(a.w),
                     ((uint64_t)bswap32(b.x) << 32) | bswap32(b.y), ((uint64_t)bswap32(b.z) << 32) | bswap32(b.w)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (S.val_type == PG_V_I64) { iv[i] = (int64_t)x[i]; fv[i] = (double)iv[i]; }
      else { fv[i] = __longlong_as_double((int64_t)x[i]); iv[i] = 0; }
    }
  }
}

// Aggregates the matching docs of one tile.  TABLE: accumulator table [n_ops][G*R] (LDS or HBM).
template <typename TABLE>
DEVFN void aggregate_tile(const PgQueryPlan& p, const uint32_t* __restrict__ mask32, int64_t tile_base, TABLE* table) {
  const int t = threadIdx.x;
  const int sh = (t & 7) * 4;
  const int R = p.replicas;
  const int64_t stride = (int64_t)p.n_groups * R;   // slots per op
  const int rep = t & (R - 1);
  for (int q = t; q < PG_TILE_QUADS; q += PG_BLOCK) {
    const uint32_t nib = (mask32[q >> 3] >> sh) & 0xFu;
    if (nib == 0) continue;
    const int64_t doc0 = tile_base + 4 * (int64_t)q;
    int64_t key[4] = {0, 0, 0, 0};
    for (int g = 0; g < p.n_group_cols; g++) {
      uint32_t d[4];
      extract4(p.gcols[g].data, doc0, p.gcols[g].bits, d);
#pragma unroll
      for (int i = 0; i < 4; i++) key[i] += (int64_t)d[i] * p.gcols[g].mult;
    }
    int64_t slot[4];
#pragma unroll
    for (int i = 0; i < 4; i++) slot[i] = key[i] * R + rep;
    int cur_src = -2;
    int64_t iv[4] = {0, 0, 0, 0};
    double fv[4] = {0, 0, 0, 0};
    for (int o = 0; o < p.n_ops; o++) {   // ops are sorted by src on the host
      const PgAccOp op = p.ops[o];
      if (op.src != cur_src) {
        cur_src = op.src;
        if (cur_src >= 0) load_values4(p.srcs[cur_src], doc0, nib, iv, fv);
      }
      TABLE* base = table + (int64_t)o * stride;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if ((nib >> i) & 1) acc_update(base + slot[i], op.fn, op.is_float, iv[i], fv[i]);
    }
  }
}

// =====================================================================================================================
// The segment query kernel: persistent workgroups, tile = wg + k * gridDim.
// dynamic LDS: [stack_depth][256] u64 filter stack, then the LDS accumulator table (LDS / SINGLE modes)
// =====================================================================================================================
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_segment_query_kernel(const PgQueryPlan p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  uint64_t* stack = smem;
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem + (size_t)p.stack_depth * PG_TILE_WORDS);
  const bool lds_agg = (p.agg_mode == PG_AGG_LDS || p.agg_mode == PG_AGG_SINGLE);
  const int64_t table_slots = (int64_t)p.n_groups * p.replicas;

  if (t < PG_MAX_STATS) s_stat[t] = 0;
  if (lds_agg) {
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (int64_t i = t; i < table_slots; i += PG_BLOCK) lds_table[o * table_slots + i] = ident;
    }
  }
  __syncthreads();

  uint32_t my_matched = 0;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t tile_base = (int64_t)tile * PG_TILE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - tile_base;
    const int32_t n_valid = rem >= PG_TILE_DOCS ? PG_TILE_DOCS : (int32_t)rem;
    const uint64_t valid = tile_valid_word(p.num_docs, tile_base + (int64_t)t * 64);

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
          scan_dispatch<false>(p.scans[ins.arg], reinterpret_cast<uint32_t*>(stack + sp * PG_TILE_WORDS), tile_base, n_valid);
          sp++;
          __syncthreads();
          break;
        case PG_F_AND_SCAN: {
          __syncthreads();
          const PgScanLeaf& L = p.scans[ins.arg];
          uint32_t nc = scan_dispatch<true>(L, reinterpret_cast<uint32_t*>(stack + (sp - 1) * PG_TILE_WORDS), tile_base, n_valid);
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
      if (lds_agg) aggregate_tile<int64_t>(p, reinterpret_cast<const uint32_t*>(stack), tile_base, lds_table);
      else aggregate_tile<int64_t>(p, reinterpret_cast<const uint32_t*>(stack), tile_base, p.partials);
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
