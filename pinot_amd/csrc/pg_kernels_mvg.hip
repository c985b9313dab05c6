// pg_mv_group: GROUP BY ONE multi-value dictionary column (alone or next to one single-value dictionary column) with integer accumulators over at most one raw INT column and no filter —
// `SELECT mv, COUNT(*), SUM(m) FROM t GROUP BY mv` — the commonest multi-value shape, as a kernel of its own (round 6).
//
// Reference: DictionaryBasedGroupKeyGenerator#processMultiValue (DictionaryBasedGroupKeyGenerator.java:357-368, 504-573): every entry of the
// doc's multi-value column is a group key of the doc — repeated entries repeat the key — and aggregateGroupByMV applies the doc's value to every
// one of its keys (CountAggregationFunction.java:120-131, SumAggregationFunction.java:181-200).
//
// pg_mv_query_l (pg_kernels_mv.hip) answers this shape inside the interpreter's frame: a lane walks its 32 docs one after the other, every doc
// a chain of dependent loads (row start -> entries -> value), 231 VGPRs, 2 wavefronts per SIMD: 2.6 % of 8 TB/s.  Here the shape is fixed, so
// the loads are not: a wavefront takes its tile's docs a BATCH of rows at a time (a row = 64 consecutive docs, one per lane); the row starts
// and the values of batch b + 1 are requested while the entries of batch b travel, and the entries of a doc — KU of them, the column's maximum
// where that is <= KU, positions past the doc's last entry clamped onto it — are requested together, before the first is used.  Neighbouring
// docs' entries are neighbours in the bit stream: a row's entry loads touch one or two cache lines.  Each entry that exists is one LDS atomic
// per accumulator at slot = dictId x R + replica.  16 wavefronts per workgroup, one workgroup per CU, <= 128 VGPRs.
// Bytes per doc: entries x bits / 8 + the row starts (the int32 offsets of pg_segment.cpp: 4 B per doc, where the index's row-start bitmap
// would be entries / 8) + 4 B of the value column.
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

template <typename T> DEVFN const GAS T* mvg_sgpr_ptr(const void* ptr) {
  const uint64_t v = (uint64_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const GAS T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}

// KU: entries of a doc requested up front; RB: rows of a batch.  HAS_SRC: an accumulator reads the raw INT column srcs[pipe_src].
template <int KU, int RB, bool HAS_SRC, bool HAS_SV>   // HAS_SV: a second, single-value dictionary group column next to the multi-value one
__device__ __forceinline__ void mv_group_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t real_slots = (uint32_t)p.n_groups * R;
  const uint32_t table_slots = real_slots + 64u;   // (+ a trash slot per lane: unused here, the layout of the other LDS-table kernels' launches)
  const int n_ops = uniform(p.n_ops);
  for (int o = 0; o < n_ops; o++) {
    const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
    for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
  }
  // the accumulators as 2-bit codes in one 64-bit scalar (0 COUNT, 1 SUM, 2 MIN, 3 MAX)
  uint64_t ops_code = 0;
  for (int o = 0; o < n_ops; o++) {
    const PgAccOp op = p.ops[uniform(o)];
    ops_code |= (uint64_t)(op.src < 0 ? 0u : (op.fn == PG_ACC_SUM ? 1u : (op.fn == PG_ACC_MIN ? 2u : 3u))) << (2 * o);
  }
  ops_code = ((uint64_t)(uint32_t)uniform((int)(uint32_t)(ops_code >> 32)) << 32) | (uint64_t)(uint32_t)uniform((int)(uint32_t)ops_code);
  // (compile-time indices into the kernel argument only: which of the two group columns is the multi-value one is a select, not an index)
  const bool mv_first = !HAS_SV || p.mv_gcol_offsets[0] != nullptr;
  const uint32_t bits = (uint32_t)uniform(mv_first ? p.gcols[0].bits : p.gcols[1].bits);
  const uint32_t mask = (1u << bits) - 1u;
  const GAS int32_t* off = mvg_sgpr_ptr<int32_t>(mv_first ? p.mv_gcol_offsets[0] : p.mv_gcol_offsets[1]);
  const GAS uint32_t* ent = mvg_sgpr_ptr<uint32_t>(mv_first ? p.gcols[0].data : p.gcols[1].data);
  const uint32_t mv_mul = (uint32_t)uniform((int)((uint32_t)(mv_first ? p.gcols[0].mult : p.gcols[1].mult) * R));
  const GAS uint32_t* sv_data = HAS_SV ? mvg_sgpr_ptr<uint32_t>(mv_first ? p.gcols[1].data : p.gcols[0].data) : nullptr;
  const uint32_t sv_bits = HAS_SV ? (uint32_t)uniform(mv_first ? p.gcols[1].bits : p.gcols[0].bits) : 1u;
  const uint32_t sv_mul = HAS_SV ? (uint32_t)uniform((int)((uint32_t)(mv_first ? p.gcols[1].mult : p.gcols[0].mult) * R)) : 0u;
  const GAS uint32_t* val = HAS_SRC ? mvg_sgpr_ptr<uint32_t>(p.srcs[p.pipe_src].data) : nullptr;
  const uint32_t rep = (uint32_t)t & (R - 1u);
  const int32_t n_docs = uniform(p.num_docs);
  // batches of this wavefront: wave tile first + (b / BPT) * step, rows (b % BPT) * RB ...
  constexpr int BPT = 32 / RB;   // batches per wave tile (32 rows of 64 docs)
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_tiles = first < p.n_wtiles ? (p.n_wtiles - first + step - 1) / step : 0;
  const int n_batches = n_tiles * BPT;
  uint32_t my_docs = 0;
  __syncthreads();

  auto doc_of = [&](int b, int r) __attribute__((always_inline)) -> int32_t {   // this lane's doc of row r of batch b
    const int wt = first + (b / BPT) * step;
    return (int32_t)((int64_t)wt * PG_WAVE_DOCS + (int64_t)(((b % BPT) * RB + r) * 64 + lane));
  };
  int32_t s_nx[RB], e_nx[RB];
  uint32_t v_nx[RB];
  u32x2 g_nx[RB];
  auto request_rows = [&](int b) __attribute__((always_inline)) {   // row starts (and values) of batch b; docs past the segment: its last doc, never applied
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int32_t d = doc_of(b < n_batches ? b : n_batches - 1, r);
      const int32_t dc = d < n_docs ? d : n_docs - 1;
      s_nx[r] = off[dc];
      e_nx[r] = off[dc + 1];
      if (HAS_SRC) v_nx[r] = val[dc];
      if (HAS_SV) g_nx[r] = *(const GAS u32x2_a4*)(sv_data + (((uint64_t)(uint32_t)dc * sv_bits) >> 5));
    }
  };
  auto apply = [&](uint32_t slot, int32_t v) __attribute__((always_inline)) {
    for (int o = 0; o < n_ops; o++) {
      const uint32_t code = (uint32_t)(ops_code >> (2 * o)) & 3u;   // 0 COUNT, 1 SUM, 2 MIN, 3 MAX
      int64_t* base = lds_table + (size_t)o * table_slots;
      if (code == 0u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot), 1ULL);
      else if (code == 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot), (unsigned long long)(int64_t)v);
      else if (code == 2u) atomicMin(reinterpret_cast<long long*>(base + slot), (long long)v);
      else atomicMax(reinterpret_cast<long long*>(base + slot), (long long)v);
    }
  };
  auto entry_of = [&](u32x2 w, uint32_t idx) __attribute__((always_inline)) -> uint32_t {   // entry idx out of the dword pair that holds its first bit
    const uint64_t win = ((uint64_t)bswap32(w.x) << 32) | (uint64_t)bswap32(w.y);
    return (uint32_t)(win >> (64u - (uint32_t)(((uint64_t)idx * bits) & 31u) - bits)) & mask;
  };

  if (n_batches > 0) {
    request_rows(0);
    for (int b = 0; b < n_batches; b++) {
      int32_t s[RB], e[RB];
      uint32_t v[RB];
      bool ok[RB];
      uint32_t base[RB];   // replica + the single-value column's part of the slot
#pragma unroll
      for (int r = 0; r < RB; r++) {
        s[r] = s_nx[r]; e[r] = e_nx[r]; v[r] = HAS_SRC ? v_nx[r] : 0u;
        const int32_t d = doc_of(b, r);
        ok[r] = d < n_docs;
        base[r] = rep;
        if (HAS_SV) {
          const uint64_t bit0 = (uint64_t)(uint32_t)(ok[r] ? d : n_docs - 1) * sv_bits;
          const uint64_t win = ((uint64_t)bswap32(g_nx[r].x) << 32) | (uint64_t)bswap32(g_nx[r].y);
          base[r] += ((uint32_t)(win >> (64u - (uint32_t)(bit0 & 31u) - sv_bits)) & ((1u << sv_bits) - 1u)) * sv_mul;
        }
      }
      // the batch's entries: KU per doc, all requested before the first is used (positions past the doc's last entry: that entry again)
      u32x2 w[RB][KU];
#pragma unroll
      for (int r = 0; r < RB; r++) {
#pragma unroll
        for (int k = 0; k < KU; k++) {
          const uint32_t idx = (uint32_t)(s[r] + k < e[r] ? s[r] + k : e[r] - 1);
          w[r][k] = *(const GAS u32x2_a4*)(ent + (((uint64_t)idx * bits) >> 5));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      request_rows(b + 1);   // the next batch's row starts travel behind them
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RB; r++) {
        my_docs += ok[r] ? 1u : 0u;
        const int32_t mv = HAS_SRC ? (int32_t)bswap32(v[r]) : 0;   // raw INT forward index: big-endian
#pragma unroll
        for (int k = 0; k < KU; k++) {
          if (ok[r] && s[r] + k < e[r]) {
            const uint32_t id = entry_of(w[r][k], (uint32_t)(s[r] + k));
            apply(id * mv_mul + base[r], mv);
          }
        }
        // a doc with more than KU entries (only where the column's maximum exceeds KU): the rest one by one
        for (int32_t i = s[r] + KU; ok[r] && i < e[r]; i++) {
          const u32x2 wi = *(const GAS u32x2_a4*)(ent + (((uint64_t)(uint32_t)i * bits) >> 5));
          apply(entry_of(wi, (uint32_t)i) * mv_mul + base[r], mv);
        }
      }
    }
  }
  {
    const uint32_t wsum = wave_sum_u32(my_docs);
    if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  }
  __syncthreads();
  // statistics and this workgroup's partial table [n_ops][n_groups], replicas folded
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  {
    const int Rr = p.replicas, groups = p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * ((int64_t)p.n_ops * groups);
    for (int o = 0; o < p.n_ops; o++) {
      const int fn = p.ops[uniform(o)].fn;   // integer accumulators only (planner)
      for (int gq = t >> 6; Rr >= 64 && gq < groups; gq += PG_BLOCK / 64) {   // a wavefront per slot folds its replicas (see flush_workgroup)
        const int64_t* src = lds_table + (size_t)o * table_slots + (size_t)gq * Rr;
        const int ln = t & 63;
        int64_t acc = src[ln];
        if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (int r = ln + 64; r < Rr; r += 64) acc += src[r]; }
        else if (fn == PG_ACC_MIN) { for (int r = ln + 64; r < Rr; r += 64) acc = src[r] < acc ? src[r] : acc; }
        else { for (int r = ln + 64; r < Rr; r += 64) acc = src[r] > acc ? src[r] : acc; }
        acc = wave_fold_i64(acc, fn);
        if (ln == 0) out[(size_t)o * groups + gq] = acc;
      }
      for (int gq = t; Rr < 64 && gq < groups; gq += PG_BLOCK) {
        const int64_t* src = lds_table + (size_t)o * table_slots + (size_t)gq * Rr;
        int64_t acc = src[0];
        if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (int r = 1; r < Rr; r++) acc += src[r]; }
        else if (fn == PG_ACC_MIN) { for (int r = 1; r < Rr; r++) acc = src[r] < acc ? src[r] : acc; }
        else { for (int r = 1; r < Rr; r++) acc = src[r] > acc ? src[r] : acc; }
        out[(size_t)o * groups + gq] = acc;
      }
    }
  }
}

// ---- the *MV aggregation functions over ONE multi-value INT column, grouped by one or two single-value dictionary columns ---------------------
// `SELECT g1, SUMMV(mv), COUNTMV(mv), MAXMV(mv) FROM t GROUP BY g1`: all entries of a doc are aggregated into the doc's key
// (SumMVAggregationFunction.java / CountMVAggregationFunction.java / MinMVAggregationFunction.java / MaxMVAggregationFunction.java:
// aggregateGroupBySV over getIntValuesMV / getNumMVEntries).  The same batches; behind the entries one more level of loads — the entries'
// dictionary values (a 4 KB dictionary stays in L1) — and the doc's entries are reduced in registers first: ONE LDS atomic per accumulator and doc.
// Accumulator kinds (3 bits each): 0 COUNT(*), 1 / 2 / 3 SUM / MIN / MAX of the entries' values, 4 / 5 / 6 SUM / MIN / MAX of the doc's number of entries.
template <int KU, int RB, int NG, bool HAS_ENT>
__device__ __forceinline__ void mv_aggr_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t real_slots = (uint32_t)p.n_groups * R;
  const uint32_t table_slots = real_slots + 64u;
  const int n_ops = uniform(p.n_ops);
  for (int o = 0; o < n_ops; o++) {
    const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
    for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
  }
  const int es = uniform(p.pipe_src);   // the multi-value source (its row starts serve the entry count too)
  uint64_t ops_code = 0;
  for (int o = 0; o < n_ops; o++) {
    const PgAccOp op = p.ops[uniform(o)];
    const uint32_t f = op.fn == PG_ACC_SUM ? 1u : (op.fn == PG_ACC_MIN ? 2u : 3u);
    ops_code |= (uint64_t)(op.src < 0 ? 0u : (p.mv_src_len[op.src] ? 3u + f : f)) << (3 * o);
  }
  ops_code = ((uint64_t)(uint32_t)uniform((int)(uint32_t)(ops_code >> 32)) << 32) | (uint64_t)(uint32_t)uniform((int)(uint32_t)ops_code);
  const uint32_t bits = (uint32_t)uniform(p.srcs[es].bits);
  const uint32_t mask = (1u << bits) - 1u;
  const GAS int32_t* off = mvg_sgpr_ptr<int32_t>(p.mv_src_offsets[es]);
  const GAS uint32_t* ent = mvg_sgpr_ptr<uint32_t>(p.srcs[es].data);
  const GAS int32_t* dict = mvg_sgpr_ptr<int32_t>(p.srcs[es].dict);
  // a dictionary of up to 4 096 values is copied into LDS behind the table (PgQueryPlan::mvg_dict_card): the entries' values are then LDS reads,
  // not a third level of global loads
  const int dict_card = uniform(p.mvg_dict_card);
  int32_t* lds_dict = reinterpret_cast<int32_t*>(lds_table + (size_t)n_ops * table_slots);
  if (HAS_ENT) for (int i = t; i < dict_card; i += PG_BLOCK) lds_dict[i] = dict[i];
  const GAS uint32_t* gdata[NG];
  uint32_t gbits[NG], gmul[NG];
#pragma unroll
  for (int j = 0; j < NG; j++) {
    gdata[j] = mvg_sgpr_ptr<uint32_t>(p.gcols[j].data);
    gbits[j] = (uint32_t)uniform(p.gcols[j].bits);
    gmul[j] = (uint32_t)uniform((int)((uint32_t)p.gcols[j].mult * R));
  }
  const uint32_t rep = (uint32_t)t & (R - 1u);
  const int32_t n_docs = uniform(p.num_docs);
  constexpr int BPT = 32 / RB;
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_tiles = first < p.n_wtiles ? (p.n_wtiles - first + step - 1) / step : 0;
  const int n_batches = n_tiles * BPT;
  uint32_t my_docs = 0;
  __syncthreads();

  auto doc_of = [&](int b, int r) __attribute__((always_inline)) -> int32_t {
    const int wt = first + (b / BPT) * step;
    return (int32_t)((int64_t)wt * PG_WAVE_DOCS + (int64_t)(((b % BPT) * RB + r) * 64 + lane));
  };
  auto field_of = [&](u32x2 w, uint64_t bit0, uint32_t width) __attribute__((always_inline)) -> uint32_t {   // the field whose first bit is bit0, out of the dword pair that holds it
    const uint64_t win = ((uint64_t)bswap32(w.x) << 32) | (uint64_t)bswap32(w.y);
    return (uint32_t)(win >> (64u - (uint32_t)(bit0 & 31u) - width)) & ((1u << width) - 1u);
  };
  int32_t s_nx[RB], e_nx[RB];
  u32x2 g_nx[RB][NG];
  auto request_rows = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int32_t d = doc_of(b < n_batches ? b : n_batches - 1, r);
      const int32_t dc = d < n_docs ? d : n_docs - 1;
      s_nx[r] = off[dc];
      e_nx[r] = off[dc + 1];
#pragma unroll
      for (int j = 0; j < NG; j++) g_nx[r][j] = *(const GAS u32x2_a4*)(gdata[j] + (((uint64_t)(uint32_t)dc * gbits[j]) >> 5));
    }
  };
  auto apply = [&](uint32_t slot, int64_t dsum, int64_t dmin, int64_t dmax, int64_t len) __attribute__((always_inline)) {
    for (int o = 0; o < n_ops; o++) {
      const uint32_t code = (uint32_t)(ops_code >> (3 * o)) & 7u;
      int64_t* at = lds_table + (size_t)o * table_slots + slot;
      if (code == 0u) atomicAdd(reinterpret_cast<unsigned long long*>(at), 1ULL);
      else if (code == 1u) atomicAdd(reinterpret_cast<unsigned long long*>(at), (unsigned long long)dsum);
      else if (code == 2u) atomicMin(reinterpret_cast<long long*>(at), (long long)dmin);
      else if (code == 3u) atomicMax(reinterpret_cast<long long*>(at), (long long)dmax);
      else if (code == 4u) atomicAdd(reinterpret_cast<unsigned long long*>(at), (unsigned long long)len);
      else if (code == 5u) atomicMin(reinterpret_cast<long long*>(at), (long long)len);
      else atomicMax(reinterpret_cast<long long*>(at), (long long)len);
    }
  };

  if (n_batches > 0) {
    request_rows(0);
    for (int b = 0; b < n_batches; b++) {
      int32_t s[RB], e[RB];
      u32x2 gw[RB][NG];
      bool ok[RB];
      int32_t doc[RB];
#pragma unroll
      for (int r = 0; r < RB; r++) {
        s[r] = s_nx[r]; e[r] = e_nx[r]; doc[r] = doc_of(b, r); ok[r] = doc[r] < n_docs;
#pragma unroll
        for (int j = 0; j < NG; j++) gw[r][j] = g_nx[r][j];
      }
      u32x2 w[RB][KU];
      if (HAS_ENT) {
#pragma unroll
        for (int r = 0; r < RB; r++) {
#pragma unroll
          for (int k = 0; k < KU; k++) {
            const uint32_t idx = (uint32_t)(s[r] + k < e[r] ? s[r] + k : e[r] - 1);
            w[r][k] = *(const GAS u32x2_a4*)(ent + (((uint64_t)idx * bits) >> 5));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      request_rows(b + 1);
      __builtin_amdgcn_sched_barrier(0);
      int32_t dv[RB][KU];
      if (HAS_ENT) {   // the entries' dictionary values: all requested before the first is used
#pragma unroll
        for (int r = 0; r < RB; r++) {
#pragma unroll
          for (int k = 0; k < KU; k++) {
            const uint32_t idx = (uint32_t)(s[r] + k < e[r] ? s[r] + k : e[r] - 1);
            const uint32_t id = field_of(w[r][k], (uint64_t)idx * bits, bits);
            dv[r][k] = dict_card ? lds_dict[id] : dict[id];   // (wave-uniform choice)
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RB; r++) {
        my_docs += ok[r] ? 1u : 0u;
        int64_t dsum = 0, dmin = INT64_MAX, dmax = INT64_MIN;
        if (HAS_ENT) {
#pragma unroll
          for (int k = 0; k < KU; k++) {
            if (s[r] + k < e[r]) {
              const int64_t x = (int64_t)dv[r][k];
              dsum += x; dmin = x < dmin ? x : dmin; dmax = x > dmax ? x : dmax;
            }
          }
          for (int32_t i = s[r] + KU; ok[r] && i < e[r]; i++) {   // rows longer than KU: the rest one by one
            const u32x2 wi = *(const GAS u32x2_a4*)(ent + (((uint64_t)(uint32_t)i * bits) >> 5));
            const uint32_t id = field_of(wi, (uint64_t)(uint32_t)i * bits, bits);
            const int64_t x = (int64_t)(dict_card ? lds_dict[id] : dict[id]);
            dsum += x; dmin = x < dmin ? x : dmin; dmax = x > dmax ? x : dmax;
          }
        }
        if (ok[r]) {
          uint32_t slot = rep;
#pragma unroll
          for (int j = 0; j < NG; j++) slot += field_of(gw[r][j], (uint64_t)(uint32_t)doc[r] * gbits[j], gbits[j]) * gmul[j];
          apply(slot, dsum, dmin, dmax, (int64_t)(e[r] - s[r]));
        }
      }
    }
  }
  {
    const uint32_t wsum = wave_sum_u32(my_docs);
    if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  }
  __syncthreads();
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  {
    const int Rr = p.replicas, groups = p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * ((int64_t)p.n_ops * groups);
    for (int o = 0; o < p.n_ops; o++) {
      const int fn = p.ops[uniform(o)].fn;
      for (int gq = t >> 6; Rr >= 64 && gq < groups; gq += PG_BLOCK / 64) {   // a wavefront per slot folds its replicas (see flush_workgroup)
        const int64_t* src = lds_table + (size_t)o * table_slots + (size_t)gq * Rr;
        const int ln = t & 63;
        int64_t acc = src[ln];
        if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (int r = ln + 64; r < Rr; r += 64) acc += src[r]; }
        else if (fn == PG_ACC_MIN) { for (int r = ln + 64; r < Rr; r += 64) acc = src[r] < acc ? src[r] : acc; }
        else { for (int r = ln + 64; r < Rr; r += 64) acc = src[r] > acc ? src[r] : acc; }
        acc = wave_fold_i64(acc, fn);
        if (ln == 0) out[(size_t)o * groups + gq] = acc;
      }
      for (int gq = t; Rr < 64 && gq < groups; gq += PG_BLOCK) {
        const int64_t* src = lds_table + (size_t)o * table_slots + (size_t)gq * Rr;
        int64_t acc = src[0];
        if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (int r = 1; r < Rr; r++) acc += src[r]; }
        else if (fn == PG_ACC_MIN) { for (int r = 1; r < Rr; r++) acc = src[r] < acc ? src[r] : acc; }
        else { for (int r = 1; r < Rr; r++) acc = src[r] > acc ? src[r] : acc; }
        out[(size_t)o * groups + gq] = acc;
      }
    }
  }
}

// PgQueryPlan::mvg = the entries requested up front (4: columns of at most 4 entries per doc and the usual case; 8: up to 8, two rows per batch)
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_mv_group_4(const PgQueryPlan p) {
  if (p.n_group_cols == 1) { if (p.pipe_src >= 0) mv_group_body<4, 4, true, false>(p); else mv_group_body<4, 4, false, false>(p); }
  else { if (p.pipe_src >= 0) mv_group_body<4, 4, true, true>(p); else mv_group_body<4, 4, false, true>(p); }
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_mv_group_8(const PgQueryPlan p) {
  if (p.n_group_cols == 1) { if (p.pipe_src >= 0) mv_group_body<8, 2, true, false>(p); else mv_group_body<8, 2, false, false>(p); }
  else { if (p.pipe_src >= 0) mv_group_body<8, 2, true, true>(p); else mv_group_body<8, 2, false, true>(p); }
}
// PgQueryPlan::mvg = 16 + 4 / 16 + 8: the *MV functions over one multi-value column grouped by single-value columns (mv_has_entries: an accumulator reads the entries' values)
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_mv_aggr_4(const PgQueryPlan p) {
  if (p.n_group_cols == 1) { if (p.mvg_has_entries) mv_aggr_body<4, 4, 1, true>(p); else mv_aggr_body<4, 4, 1, false>(p); }
  else { if (p.mvg_has_entries) mv_aggr_body<4, 4, 2, true>(p); else mv_aggr_body<4, 4, 2, false>(p); }
}
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_mv_aggr_8(const PgQueryPlan p) {
  if (p.n_group_cols == 1) mv_aggr_body<8, 2, 1, true>(p); else mv_aggr_body<8, 2, 2, true>(p);
}
