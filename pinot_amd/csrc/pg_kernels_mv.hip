// Multi-value columns: filter, group-by and aggregation over FixedBitMVForwardIndexReader columns.
//
// Reference semantics reproduced:
//   MVScanDocIdIterator + PredicateEvaluator#applyMV   core/operator/dociditerators/MVScanDocIdIterator.java:65-117,184-192 and
//       core/operator/filter/predicate/BaseDictionaryBasedPredicateEvaluator.java:164-180 — a doc passes when ANY of its dictIds passes,
//       ALL of them for the exclusive predicates (NOT_EQ / NOT_IN); every entry of every evaluated doc counts as scanned
//   DictionaryBasedGroupKeyGenerator#processMultiValue  core/query/aggregation/groupby/DictionaryBasedGroupKeyGenerator.java:357-368,
//       504-573 — one raw key per combination of the doc's entries over the multi-value group columns; repeated entries repeat the key
//   aggregateGroupByMV of every function, and CountMV / SumMV / MinMV / MaxMV / AvgMV / MinMaxRangeMV / DistinctCountMV /
//       DistinctCountHLLMV AggregationFunction.java — all entries of a doc are aggregated (into every key of the doc)
//
// Layout: a multi-value column is the bit stream of all its entries (dictIds, bits_per_value each, doc after doc) plus the docs' first
// entries as an int32 array — the row-start bitmap of the index expanded once at registration (pg_segment.cpp); the reader's chunk
// offsets + bitmap walk per doc has no place on the device.
//
// The work is irregular (entries per doc vary, keys per doc multiply), so the kernels keep the interpreter's frame — wave tiles, the
// filter program on a register stack of match masks in quad layout, LDS / HBM accumulator tables flushed like pg_generic_query_* —
// and walk the matching docs of a lane one by one: entries of neighbouring docs are neighbours in the stream, so the lanes of a
// wavefront still read neighbouring lines.  A translation unit of its own: the single-value kernels are compiled without any of this.
#include <hip/hip_runtime.h>

#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

// entry `idx` of a multi-value column's bit stream (64-bit bit offsets: 2^31 entries x 31 bits)
DEVFN uint32_t mv_entry_at(const uint8_t* data, uint32_t idx, uint32_t bits) {
  const uint64_t bit0 = (uint64_t)idx * bits;
  const GAS uint32_t* w = gptr<uint32_t>(data) + (bit0 >> 5);
  const u32x2 v = *(const GAS u32x2_a4*)w;
  const uint64_t win = ((uint64_t)bswap32(v.x) << 32) | (uint64_t)bswap32(v.y);
  return (uint32_t)(win >> (64u - (uint32_t)(bit0 & 31u) - bits)) & ((1u << bits) - 1u);
}

// MVScanDocIdIterator over the candidates of one wave tile (quad-layout mask); `entries` += the entries of the docs evaluated
template <class LeafT>
DEVFN uint32_t mv_scan_wtile(const LeafT& L, uint32_t cand, int wt, int lane, uint32_t& entries) {
  const GAS int32_t* off = gptr<int32_t>(L.set_values);
  const uint32_t bits = (uint32_t)L.bits;
  const uint32_t lo = (uint32_t)L.lo, span = (uint32_t)(L.hi - L.lo);
  const bool lut = L.pred_kind == PG_P_DICT_LUT, all = L.exclusive != 0;
  uint32_t out = 0;
  for (uint32_t m = cand; m;) {
    const int bit = __builtin_ctz(m);
    m &= m - 1;
    const int64_t doc = (int64_t)wt * PG_WAVE_DOCS + 4 * ((bit >> 2) * 64 + lane) + (bit & 3);
    const uint32_t s = (uint32_t)off[doc], e = (uint32_t)off[doc + 1];
    entries += e - s;
    bool any_pass = false, all_pass = true;
    for (uint32_t k = s; k < e; k++) {
      const uint32_t d = mv_entry_at(L.data, k, bits);
      const bool pass = lut ? ((gptr<uint32_t>(L.lut)[d >> 5] >> (d & 31u)) & 1u) != 0 : (d - lo) <= span;
      any_pass |= pass;
      all_pass &= pass;
    }
    if (all ? all_pass : any_pass) out |= 1u << bit;
  }
  return out;
}

// The same over a FULL wave tile, ENTRY-parallel (round 4).  The doc-by-doc walk above is a chain of dependent loads — a lane's 32 docs one
// after the other, every entry an 8-byte load of its own — and 128 instructions per entry once the compiler has predicated its loops: 0.70 ms
// for a 5 x 10^7-doc scan, 3.7 % of 8 TB/s (profiles/r04_i_variants_mv_50m.txt; a first rewrite that walked a quad's entries out of a 32-byte
// register window was no better: 0.60 ms, 12 300 wave instructions per tile, profiles/r04_u_*).  A tile's entries are one contiguous run of
// the stream, so the wavefront tests them 64 at a time — lane L the entry base + L: neighbouring lanes read neighbouring bits, one ballot
// per 64 entries — and keeps the pass bits as a bitmap in LDS; a doc then is a bit range [row start, row end) of that bitmap: ANY entry
// passes = the range is not zero, ALL = it is all ones.  Tiles with more than PG_MV_TILE_ENTRIES entries keep the walk above.
#define PG_MV_TILE_ENTRIES 8192
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
template <class LeafT>
DEVFN uint32_t mv_scan_wtile_bitmap(const LeafT& L, uint32_t cand, int wt, int lane, uint32_t& entries, uint32_t* bm /* [PG_MV_TILE_ENTRIES / 32 + 2] */,
                                    bool& taken) {
  const GAS int32_t* off = gptr<int32_t>(L.set_values) + (size_t)wt * PG_WAVE_DOCS;
  const uint32_t E0 = (uint32_t)__builtin_amdgcn_readfirstlane(off[0]), E1 = (uint32_t)__builtin_amdgcn_readfirstlane(off[PG_WAVE_DOCS]);
  const uint32_t n_e = E1 - E0;
  taken = n_e <= (uint32_t)PG_MV_TILE_ENTRIES;
  if (!taken) return 0u;   // wave-uniform
  const uint32_t bits = (uint32_t)L.bits, vmask = (1u << bits) - 1u;
  const uint32_t lo = (uint32_t)L.lo, span = (uint32_t)(L.hi - L.lo);
  const bool lut = L.pred_kind == PG_P_DICT_LUT, all = L.exclusive != 0;
  // row starts of the lane's eight quads (16 bytes each, requested together)
  i32x4 o[8];
#pragma unroll
  for (int k = 0; k < 8; k++) o[k] = *(const GAS i32x4*)(off + 4 * (k * 64 + lane));
  // ---- entry phase: pass bits of entries E0 .. E1 - 1 -> bm -------------------------------------------------------------------------------
  const GAS uint32_t* stream = gptr<uint32_t>(L.data);
  for (uint32_t base = 0; base < n_e; base += 256u) {   // four ballots per iteration: four loads in flight
    uint32_t dv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t idx = min(base + 64u * (uint32_t)u + (uint32_t)lane, n_e - 1u);   // (lanes beyond the run re-read its last entry)
      const uint64_t bit0 = (uint64_t)(E0 + idx) * bits;
      const u32x2 v = *(const GAS u32x2_a4*)(stream + (bit0 >> 5));
      const uint64_t win = ((uint64_t)bswap32(v.x) << 32) | (uint64_t)bswap32(v.y);
      dv[u] = (uint32_t)(win >> (64u - (uint32_t)(bit0 & 31u) - bits)) & vmask;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t idx = base + 64u * (uint32_t)u + (uint32_t)lane;
      const uint32_t d = dv[u];
      const bool pass = (lut ? ((gptr<uint32_t>(L.lut)[d >> 5] >> (d & 31u)) & 1u) != 0 : (d - lo) <= span) && idx < n_e;
      const uint64_t m = __ballot(pass);
      if (lane < 2) bm[(base >> 5) + 2u * (uint32_t)u + (uint32_t)lane] = lane ? (uint32_t)(m >> 32) : (uint32_t)m;   // (beyond the run: zeros)
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // ---- doc phase: a candidate doc is the bit range [row start, row end) ---------------------------------------------------------------------
  const volatile uint32_t* vb = bm;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t c4 = (cand >> (4 * k)) & 0xFu;
    // end of the quad's rows = the next lane's first row start (lane 63: lane 0 of the next quad, or the next tile)
    const int32_t next_q = k < 7 ? __builtin_amdgcn_readfirstlane(o[k < 7 ? k + 1 : 7].x) : (int32_t)E1;
    int32_t o4 = __shfl_down(o[k].x, 1, 64);
    if (lane == 63) o4 = next_q;
    const uint32_t rs[5] = {(uint32_t)o[k].x - E0, (uint32_t)o[k].y - E0, (uint32_t)o[k].z - E0, (uint32_t)o[k].w - E0, (uint32_t)o4 - E0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!((c4 >> i) & 1u)) continue;
      uint32_t sb = rs[i];
      const uint32_t eb = rs[i + 1];
      entries += eb - sb;
      bool any_pass = false, all_pass = true;
      while (sb < eb) {   // 32 bits of the row at a time (one pass for rows of up to 32 entries)
        const uint32_t n = min(eb - sb, 32u);
        const uint32_t w0 = vb[sb >> 5], w1 = vb[(sb >> 5) + 1u];
        const uint32_t x = (uint32_t)((((uint64_t)w1 << 32) | (uint64_t)w0) >> (sb & 31u));
        const uint32_t full = n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u);
        any_pass |= (x & full) != 0u;
        all_pass &= (x & full) == full;
        sb += n;
      }
      out |= (uint32_t)(all ? all_pass : any_pass) << (4 * k + i);
    }
  }
  return out;
}

#define PG_MV_MAX_GROUP_COLS 4
struct MvKeys {   // the multi-value group columns' entries of one doc (the planner admits at most PG_MV_MAX_GROUP_COLS such columns)
  uint32_t base;        // slot of the single-value part of the key (replica included)
  uint32_t combos;      // keys of the doc
  uint32_t start[PG_MV_MAX_GROUP_COLS], len[PG_MV_MAX_GROUP_COLS], mult[PG_MV_MAX_GROUP_COLS], bits[PG_MV_MAX_GROUP_COLS];
  const uint8_t* data[PG_MV_MAX_GROUP_COLS];
  int n;
};
DEVFN uint32_t mv_key_slot(const MvKeys& K, uint32_t c) {   // combination c of the doc: digit j = (c / (len_0 ... len_{j-1})) mod len_j
  uint32_t slot = K.base, r = c;
#pragma unroll
  for (int j = 0; j < PG_MV_MAX_GROUP_COLS; j++)
    if (j < K.n) {
      const uint32_t i = j + 1 < K.n ? r % K.len[j] : r;
      r = j + 1 < K.n ? r / K.len[j] : 0u;
      slot += mv_entry_at(K.data[j], K.start[j] + i, K.bits[j]) * K.mult[j];
    }
  return slot;
}

// one dictionary value as the int64 the accumulators take (integers as they are, FLOAT / DOUBLE as double bits)
DEVFN int64_t mv_dict_value(const PgValueSrc& S, uint32_t d) {
  if (S.val_type == PG_V_I32) return (int64_t)(int32_t)gptr<uint32_t>(S.dict)[d];
  if (S.val_type == PG_V_F32) return __double_as_longlong((double)__uint_as_float(gptr<uint32_t>(S.dict)[d]));
  return (int64_t)gptr<uint64_t>(S.dict)[d];
}

// Aggregates the matching docs of one wave tile: doc by doc, accumulator by accumulator, key by key.
DEVFN void mv_aggregate_wtile(const PgQueryPlan& p, uint32_t m, int wt, int64_t* table, int lane, uint32_t rep) {
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t stride = (uint32_t)p.n_groups * R;   // slots per accumulator
  while (m) {
    const int bit = __builtin_ctz(m);
    m &= m - 1;
    const uint32_t in_tile = (uint32_t)(4 * ((bit >> 2) * 64 + lane) + (bit & 3));
    const int64_t doc = (int64_t)wt * PG_WAVE_DOCS + in_tile;
    MvKeys K;
    K.base = rep; K.combos = 1; K.n = 0;
#pragma unroll
    for (int j = 0; j < PG_MV_MAX_GROUP_COLS; j++) { K.start[j] = 0; K.len[j] = 1; K.mult[j] = 0; K.bits[j] = 1; K.data[j] = nullptr; }
    for (int g = 0; g < p.n_group_cols; g++) {
      const PgGroupCol& gc = p.gcols[g];
      const uint32_t mult = (uint32_t)gc.mult * R;
      if (p.mv_gcol_offsets[g]) {
        const GAS int32_t* off = gptr<int32_t>(p.mv_gcol_offsets[g]);
        const uint32_t s = (uint32_t)off[doc], e = (uint32_t)off[doc + 1];
        // (constant indices: a run-time index into the struct's arrays puts them into scratch memory — 72 bytes per lane in round 3)
#pragma unroll
        for (int j = 0; j < PG_MV_MAX_GROUP_COLS; j++)
          if (K.n == j) { K.start[j] = s; K.len[j] = e - s; K.mult[j] = mult; K.bits[j] = (uint32_t)gc.bits; K.data[j] = gc.data; }
        K.combos *= e - s;
        K.n++;
      } else {
        K.base += packed_value_at(packed_wtile_base(gc.data, wt, gc.bits), in_tile, (uint32_t)gc.bits) * mult;
      }
    }
    // ---- accumulators --------------------------------------------------------------------------------------------------------
    for (int o = 0; o < p.n_ops; o++) {
      const PgAccOp op = p.ops[o];
      int64_t* base = table + (size_t)o * stride;
      int64_t v = 1;
      bool as_double = false;
      if (op.src >= 0) {
        const PgValueSrc& S = p.srcs[op.src];
        as_double = S.val_type == PG_V_F32 || S.val_type == PG_V_F64;
        if (!p.mv_src_offsets[op.src]) {
          v = source_value_at(S, wt, in_tile);
        } else {
          const GAS int32_t* off = gptr<int32_t>(p.mv_src_offsets[op.src]);
          const uint32_t s = (uint32_t)off[doc], e = (uint32_t)off[doc + 1];
          if (p.mv_src_len[op.src]) {
            v = (int64_t)(e - s);
            as_double = false;
          } else if (op.is_float == PG_ACCV_FIXED_DIGIT) {   // SUMMV / AVGMV over FLOAT / DOUBLE entries: digit `limb` of every entry, summed in int64
            v = 0;
            for (uint32_t k = s; k < e; k++)
              v += fx_digit(__longlong_as_double(mv_dict_value(S, mv_entry_at(S.data, k, (uint32_t)S.bits))), S.fx_q, op.limb);
            as_double = false;
          } else if (!as_double) {   // SUM / MIN / MAX of the doc's entries in int64, then one update per key
            v = op.fn == PG_ACC_SUM ? 0 : (op.fn == PG_ACC_MIN ? INT64_MAX : INT64_MIN);
            for (uint32_t k = s; k < e; k++) {
              const int64_t x = mv_dict_value(S, mv_entry_at(S.data, k, (uint32_t)S.bits));
              v = op.fn == PG_ACC_SUM ? v + x : (op.fn == PG_ACC_MIN ? (x < v ? x : v) : (x > v ? x : v));
            }
          } else {                   // MIN / MAX of FLOAT / DOUBLE entries (the planner keeps floating SUMMV off the device)
            double dv = op.fn == PG_ACC_MIN ? INFINITY : -INFINITY;
            for (uint32_t k = s; k < e; k++) {
              const double x = __longlong_as_double(mv_dict_value(S, mv_entry_at(S.data, k, (uint32_t)S.bits)));
              if (op.fn == PG_ACC_MIN ? x < dv : x > dv) dv = x;   // NaN never replaces the holder, as in Java
            }
            v = __double_as_longlong(dv);
          }
        }
      }
      const bool mv_source = op.src >= 0 && p.mv_src_offsets[op.src] != nullptr;
      for (uint32_t c = 0; c < K.combos; c++) {
        int64_t* slot = base + mv_key_slot(K, c);
        if (op.src < 0) atomicAdd(reinterpret_cast<unsigned long long*>(slot), 1ULL);   // COUNT
        else if (!mv_source && op.is_float == PG_ACCV_FIXED_DIGIT) acc_from_double(slot, op, __longlong_as_double(v), p.srcs[op.src].fx_q);   // a single-value FLOAT / DOUBLE SUM next to multi-value columns
        else if (!mv_source && op.is_float == PG_ACCV_LONG_DIGIT) acc_from_int(slot, op, v);
        else if (as_double) acc_float(slot, op.fn, __longlong_as_double(v));
        else acc_int(slot, op.fn, v);
      }
    }
    // ---- DISTINCTCOUNT dictId sets / HyperLogLog registers ---------------------------------------------------------------------------
    for (int xa = 0; xa < p.n_aux; xa++) {
      PgAuxOp A = p.aux[xa];
      if (A.lds_offset >= 0) {
        extern __shared__ __attribute__((aligned(16))) uint64_t smem_aux[];
        A.base = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem_aux) + A.lds_offset);
      } else {
        A.base = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(A.base) + (size_t)(blockIdx.x & (uint32_t)(A.n_rep - 1)) * (size_t)A.rep_bytes);
      }
      const PgValueSrc& S = p.srcs[A.src];
      uint32_t s = 0, e = 1;
      const bool mv_src = p.mv_src_offsets[A.src] != nullptr;
      if (mv_src) {
        const GAS int32_t* off = gptr<int32_t>(p.mv_src_offsets[A.src]);
        s = (uint32_t)off[doc];
        e = (uint32_t)off[doc + 1];
      }
      for (uint32_t k = s; k < e; k++) {
        uint32_t d = 0, idx_rank = 0;
        if (A.kind == PG_AUX_HLL_RAW) {   // raw single-value column: the value is hashed on the fly
          int64_t hv;
          if (S.col_kind == PG_COL_RAW32) hv = (int64_t)(int32_t)bswap32(gptr<uint32_t>(S.data + (size_t)wt * (PG_WAVE_DOCS * 4))[in_tile]);
          else { const u32x2 x = gptr<u32x2>(S.data + (size_t)wt * (PG_WAVE_DOCS * 8))[in_tile]; hv = (int64_t)(((uint64_t)bswap32(x.x) << 32) | bswap32(x.y)); }
          idx_rank = hll_index_rank_dev(murmur_hash_long_dev(hv), A.log2m);
        } else {
          d = mv_src ? mv_entry_at(S.data, k, (uint32_t)S.bits) : packed_value_at(packed_wtile_base(S.data, wt, S.bits), in_tile, (uint32_t)S.bits);
          if (A.kind == PG_AUX_HLL_DICT) idx_rank = gptr<uint32_t>(A.lut)[d];
        }
        for (uint32_t c = 0; c < K.combos; c++) {
          const size_t g = (size_t)(mv_key_slot(K, c) >> p.replica_shift);
          if (A.kind == PG_AUX_DICT_SET) set_add(A.base + g * (size_t)A.stride, d);
          else hll_update(reinterpret_cast<uint8_t*>(A.base) + g * (size_t)A.stride, idx_rank & 0xFFFFu, idx_rank >> 16);
        }
      }
    }
  }
}

// TABLE: 0 = no accumulator table, 1 = LDS table (PG_AGG_LDS / PG_AGG_SINGLE), 2 = dense HBM table (PG_AGG_GLOBAL)
template <int TABLE>
__device__ __forceinline__ void mv_query_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  __shared__ uint32_t s_wscratch[PG_GENERIC_BLOCK / 64][64];
  __shared__ uint32_t s_mvbits[PG_GENERIC_BLOCK / 64][PG_MV_TILE_ENTRIES / 32 + 8];   // per wavefront: pass bits of a tile's entries (mv_scan_wtile_bitmap)
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  const bool lds_agg = TABLE == 1;
  const uint32_t table_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;

  if (t < PG_MAX_STATS) s_stat[t] = 0;
  if (lds_agg) {
    for (int o = 0; o < p.n_ops; o++) {
      const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
      for (uint32_t i = t; i < table_slots; i += PG_GENERIC_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
    }
    for (int x = 0; x < p.n_aux; x++)
      if (p.aux[x].lds_offset >= 0) {
        uint32_t* z = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + p.aux[x].lds_offset);
        for (int64_t i = t; i < p.aux[x].rep_bytes / 4; i += PG_GENERIC_BLOCK) z[i] = 0;
      }
  }
  __syncthreads();

  const uint32_t rep = (uint32_t)t & ((uint32_t)p.replicas - 1u);
  uint32_t my_matched = 0;
  const int wstride = (int)gridDim.x * (PG_GENERIC_BLOCK / 64);
  for (int wt = (int)blockIdx.x * (PG_GENERIC_BLOCK / 64) + wave; wt < p.n_wtiles; wt += wstride) {
    const int64_t wbase = (int64_t)wt * PG_WAVE_DOCS;
    const int64_t rem = (int64_t)p.num_docs - wbase;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (int32_t)rem;
    const uint32_t valid_q = valid_quad_mask(n_valid, lane);
    const uint32_t valid_l = valid_lin_mask(n_valid, lane);

    // ---- filter program (the interpreter's, with the multi-value scan leaves) ----------------------------------------------------------
    MaskStack st;
    if (p.n_lin_prefix > 0) st.push(lin_to_quad(index_program_lin<true>(p, p.n_lin_prefix, wt, wbase, valid_l, s_wscratch[wave], lane), lane));
    for (int i = p.n_lin_prefix; i < p.n_instr; i++) {
      const int fop = cptr(p.instrs)[i].op, farg = cptr(p.instrs)[i].arg;
      switch (fop) {
        case PG_F_PUSH_POSTINGS: st.push(lin_to_quad(postings_wtile(cptr(p.postings)[farg], wt, valid_l, s_wscratch[wave], lane), lane)); break;
        case PG_F_PUSH_RANGES: st.push(lin_to_quad(ranges_wtile(cptr(p.ranges)[farg], wbase, valid_l, lane), lane)); break;
        case PG_F_PUSH_WORDS: st.push(lin_to_quad(gptr<uint32_t>(cptr(p.ranges)[farg].words)[(int64_t)wt * 64 + lane] & valid_l, lane)); break;
        case PG_F_PUSH_RANGEIDX: st.push(lin_to_quad(rangeidx_wtile(cptr(p.rangeidx)[farg], wt, valid_l, s_wscratch[wave], lane), lane)); break;
        case PG_F_PUSH_ALL: st.push(valid_q); break;
        case PG_F_PUSH_NONE: st.push(0u); break;
        case PG_F_PUSH_SCAN: {
          const CAS PgScanLeaf& L = cptr(p.scans)[farg];
          uint32_t ignored = 0;   // every doc is evaluated: the planner counted the column's entries already
          if (L.mv) {
            bool taken = false;
            uint32_t mm = 0;
            if (n_valid == PG_WAVE_DOCS && !p.mv_no_windows) mm = mv_scan_wtile_bitmap(L, valid_q, wt, lane, ignored, s_mvbits[wave], taken);
            st.push(taken ? mm : mv_scan_wtile(L, valid_q, wt, lane, ignored));
          } else {
            st.push(scan_dispatch(L, valid_q, wt, lane));
          }
          break;
        }
        case PG_F_AND_SCAN: {
          const CAS PgScanLeaf& L = cptr(p.scans)[farg];
          const uint32_t cand = st.s0;
          const uint32_t nc = wave_sum_u32((uint32_t)__popc(cand));
          if (nc) {   // wave-uniform
            if (L.mv) {
              uint32_t entries = 0;
              bool taken = false;
              uint32_t mm = 0;
              if (n_valid == PG_WAVE_DOCS && !p.mv_no_windows) mm = mv_scan_wtile_bitmap(L, cand, wt, lane, entries, s_mvbits[wave], taken);
              st.s0 = taken ? mm : mv_scan_wtile(L, cand, wt, lane, entries);
              const uint32_t ne = wave_sum_u32(entries);
              if (lane == 0) atomicAdd(&s_stat[L.stat_slot], ne);
            } else {
              st.s0 = scan_dispatch(L, cand, wt, lane);
              if (lane == 0) atomicAdd(&s_stat[L.stat_slot], nc);
            }
          }
          break;
        }
        case PG_F_AND: st.pop_and(); break;
        case PG_F_OR: st.pop_or(); break;
        case PG_F_NOT: st.s0 = (~st.s0) & valid_q; break;
        default: break;
      }
    }
    const uint32_t m = st.s0;
    if (p.out_words) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + lane] = quad_to_lin(m, lane);
    if (p.out_tile_counts) {
      const uint32_t wsum = wave_sum_u32((uint32_t)__popc(m));
      if (lane == 0 && wsum) atomicAdd(&p.out_tile_counts[wt / PG_WTILES_PER_TILE], wsum);
    }
    my_matched += (uint32_t)__popc(m);
    if (TABLE != 0 && m) mv_aggregate_wtile(p, m, wt, TABLE == 1 ? lds_table : p.partials, lane, rep);
  }

  // ---- epilogue: statistics and accumulator flush (as pg_generic_query_*) ------------------------------------------------------------------
  const uint32_t wsum = wave_sum_u32(my_matched);
  if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  if (lds_agg) {
    const int R = p.replicas;
    const int groups = p.n_groups;
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
      const int64_t* src = lds_table + i * R;
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
    for (int x = 0; x < p.n_aux; x++)
      if (p.aux[x].lds_offset >= 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(smem) + p.aux[x].lds_offset);
        uint32_t* dst = p.aux[x].base + (int64_t)blockIdx.x * (p.aux[x].rep_bytes / 4);
        for (int64_t i = t; i < p.aux[x].rep_bytes / 4; i += PG_GENERIC_BLOCK) dst[i] = src[i];
      }
  }
}
extern "C" __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_mv_query_f(const PgQueryPlan p) { mv_query_body<0>(p); }
extern "C" __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_mv_query_l(const PgQueryPlan p) { mv_query_body<1>(p); }
extern "C" __global__ void __launch_bounds__(PG_GENERIC_BLOCK) pg_mv_query_g(const PgQueryPlan p) { mv_query_body<2>(p); }
