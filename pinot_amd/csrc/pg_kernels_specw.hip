// pg_fast_dictrange_w: the shapes of pg_kernels_specd.hip (config 3 over dictionary-encoded scan / value columns) with a SHARED stage per workgroup
// (late round 6).  What the independent wavefronts of pg_fast_dictrange_s cannot do is stream well: 16 wavefronts x 13 short streams per CU
// (1.25 KB of a 20-bit column per wavefront and request, eight 256-byte posting rows per tile) reach 76-78 % of 8 TB/s with NOTHING behind the
// loads (profiles/r06_specd_stream_only_ab.txt) where four wavefronts requesting whole tiles reach 88 % (pg_kernels_spec.hip) — the memory system
// prefers few, long streams.  Here the workgroup requests a STAGE — SW_WAVES x 512 consecutive docs of every column and of every posting bitmap,
// tens of KB per request round — as whole 1 KB rows, one row per wavefront and instruction, straight into LDS (global_load_lds_dwordx4: no
// staging registers, no ds_write), into one of two stage buffers; one s_barrier per stage; behind it every wavefront filters and aggregates ITS
// sub-tile (512 docs) of the stage exactly as pg_fast_dictrange_s does — oct layout, branch-free field reads out of LDS, ballot ranks, selection
// list, 64 matches per round — while the next stage travels.  The candidates come out of the stage too: lane L's 8 docs are byte L of the
// sub-tile's 64 bytes of every bitmap (the linear layout read bytewise), the dense index program runs on bytes.
// Same plans, same results, same statistics as the _s family (tests/test_gpu_dict_headline.py runs both frames); the planner falls back to _s
// where two stage buffers do not fit beside the table (wide raw columns, many groups).
#ifndef SW_WAVES
#define SW_WAVES 16     // wavefronts per workgroup = sub-tiles per stage
#endif
#ifndef SW_DMA_AUX
#define SW_DMA_AUX 0    // cache policy bits of the LDS-DMA loads (2: non-temporal)
#endif
#ifndef SW_MIN_WAVES_PER_SIMD
#define SW_MIN_WAVES_PER_SIMD 4   // 16 wavefronts per CU whatever the workgroup size (16 / SW_WAVES workgroups)
#endif
#define PG_WAVES_PER_BLOCK SW_WAVES
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"
#include "pg_oct_layout.h"

#define SW_PAD 16u                                      // a field's dword pair may reach one dword past the column's last byte
#define SW_LIST_BYTES ((OCT_SUB_DOCS + 64u) * 2u)       // a wavefront's selection list: 512 uint16 entries + a dummy entry per lane
#define SW_SUB_BYTES (OCT_SUB_DOCS / 8u)                // bytes of one sub-tile in a bitmap, and per bit of width in a column
#define SW_ROWS_MAX ((32 + 24 + 8 + 8 + 9 + SW_WAVES - 1) / SW_WAVES + 1)   // rows (1 KB pieces of a stage) one wavefront may own
__host__ __device__ static inline uint32_t sw_region(uint32_t bits) { return bits ? (uint32_t)SW_WAVES * SW_SUB_BYTES * bits + SW_PAD : 0u; }
extern "C" const int pg_specw_waves_per_block = PG_WAVES_PER_BLOCK;
// bytes of ONE stage buffer (the launch holds two, and SW_WAVES selection lists, behind the table and its trash slots)
extern "C" int pg_specw_stage_bytes(int scan_bits, int value_bits, int bits0, int bits1, int n_bitmaps) {
  return (int)(sw_region((uint32_t)scan_bits) + sw_region((uint32_t)value_bits) + sw_region((uint32_t)bits0) + sw_region((uint32_t)bits1) + (uint32_t)n_bitmaps * SW_WAVES * SW_SUB_BYTES);
}
extern "C" int pg_specw_list_bytes() { return (int)(SW_WAVES * SW_LIST_BYTES); }

template <typename T> DEVFN const GAS T* sw_sgpr_ptr(const void* ptr) {
  const uint64_t v = (uint64_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const GAS T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}
DEVFN uint32_t sw_or_reduce4(uint32_t v) {   // OR across aligned groups of 4 lanes
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  return v;
}
// workgroup barrier fenced on the LDS address space only (pg_kernels_spec.hip: a full fence drains and re-reads far more than the stage needs)
DEVFN void sw_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
enum { SW_V_RAW32 = 1, SW_V_AFFINE = 2, SW_V_GATHER = 3 };   // PgQueryPlan::specd_vkind

template <int NG, bool HAS_INDEX, bool HAS_SCAN, bool HAS_TAIL, int VK>
__device__ __forceinline__ void specw_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  typedef __attribute__((address_space(3))) uint8_t LdsByte;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  const uint32_t real_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;
  const uint32_t table_slots = real_slots + 64u;   // + one trash slot per lane behind every accumulator's row (the dead lanes of a list's last round)
  for (int o = 0; o < p.n_ops; o++) {
    const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
    for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
  }
  const uint32_t sbits = HAS_SCAN ? (uint32_t)uniform(p.specd_sbits) : 0u, vbits = (uint32_t)uniform(p.specd_vbits);
  uint32_t gbits[NG];
#pragma unroll
  for (int gi = 0; gi < NG; gi++) gbits[gi] = (uint32_t)uniform(p.gcols[gi].bits);
  // the dense index program as three scalar bit masks over the eight pointer slots (pg_kernels_specd.hip): live (a real pointer — the planner
  // pads from the top down with copies of slot 0, so the live slots are a prefix), first (opens a group), excl (the group closed behind slot j
  // is complemented; bit 7: the last group)
  uint32_t idx_live = 0u, idx_first = 0u, idx_excl = 0u;
  if (HAS_INDEX) {
    bool pad = true;
    uint32_t live = 1u;
#pragma unroll
    for (int j = 7; j >= 1; j--) {
      pad = pad && p.dense_ptr[j] == p.dense_ptr[0] && p.dense_group[j] == p.dense_group[0];
      if (!pad) live |= 1u << j;
    }
    idx_live = live;
    int last_g = -1;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int gj = p.dense_group[j];
      if (((live >> j) & 1u) && gj != last_g) {
        idx_first |= 1u << j;
        if (j > 0 && ((p.dense_excl >> last_g) & 1)) idx_excl |= 1u << (j - 1);
        last_g = gj;
      }
    }
    if ((p.dense_excl >> last_g) & 1) idx_excl |= 1u << 7;
    idx_live = (uint32_t)uniform((int)idx_live); idx_first = (uint32_t)uniform((int)idx_first); idx_excl = (uint32_t)uniform((int)idx_excl);
  }
  const uint32_t n_bm = HAS_INDEX ? (uint32_t)__builtin_popcount(idx_live) : 0u;
  // one stage buffer: [scan bytes][value bytes][group bytes ...][bitmap 0 .. n_bm - 1][upsert snapshot]; a column's bytes of the stage are
  // contiguous in memory (SW_WAVES x 64 x bits) and its sub-tile w starts 64 w bits bytes in
  const uint32_t col_unit = (uint32_t)SW_WAVES * SW_SUB_BYTES;   // bytes of a stage per bit of width (and per bitmap)
  const uint32_t off_val = sw_region(sbits), off_g0 = off_val + sw_region(vbits), off_g1 = off_g0 + sw_region(gbits[0]);
  const uint32_t off_bm = off_g1 + (NG > 1 ? sw_region(gbits[NG - 1]) : 0u);
  const uint32_t stage_bytes = off_bm + (n_bm + (HAS_TAIL ? 1u : 0u)) * col_unit;
  uint8_t* stage0 = reinterpret_cast<uint8_t*>(smem) + (((size_t)p.n_ops * table_slots * 8u + 15u) & ~(size_t)15u);
  uint16_t* my_list = reinterpret_cast<uint16_t*>(stage0 + 2u * stage_bytes + (uint32_t)wave * SW_LIST_BYTES);
  const CAS PgScanLeaf& L = cptr(p.scans)[HAS_SCAN ? p.fast_scan : 0];   // only dereferenced when HAS_SCAN
  const uint8_t* xdata = p.srcs[p.pipe_src].data;
  const uint8_t* sdata = HAS_SCAN ? L.data : xdata;
  // stages of this workgroup: blockIdx.x, + gridDim.x, ...
  const int64_t stage_docs = (int64_t)SW_WAVES * OCT_SUB_DOCS;
  const int n_stages = (int)(((int64_t)p.num_docs + stage_docs - 1) / stage_docs);
  const int first = (int)blockIdx.x, step = (int)gridDim.x;
  const int n_mine = first < n_stages ? (n_stages - first + step - 1) / step : 0;

  // the rows (1 KB pieces of a stage, the last of a column possibly shorter) this wavefront requests: rows wave, wave + SW_WAVES, ... over
  // [scan rows][value rows][group rows][one row group per bitmap] — source at stage 0, bytes per stage, offset in the stage buffer, length
  const GAS uint8_t* r_src[SW_ROWS_MAX];
  uint32_t r_stride[SW_ROWS_MAX], r_dst[SW_ROWS_MAX], r_len[SW_ROWS_MAX];
  {
    const uint32_t rows_per_bm = (col_unit + 1023u) / 1024u;
    auto rows_of = [&](uint32_t bits) __attribute__((always_inline)) { return (col_unit * bits + 1023u) / 1024u; };
    const uint32_t e_scan = rows_of(sbits), e_val = e_scan + rows_of(vbits), e_g0 = e_val + rows_of(gbits[0]), e_g1 = e_g0 + (NG > 1 ? rows_of(gbits[NG - 1]) : 0u);
    const uint32_t e_all = e_g1 + (n_bm + (HAS_TAIL ? 1u : 0u)) * rows_per_bm;
#pragma unroll
    for (int q = 0; q < SW_ROWS_MAX; q++) {
      const uint32_t r = (uint32_t)wave + (uint32_t)SW_WAVES * (uint32_t)q;
      const uint8_t* base = nullptr;
      uint32_t stride = 0, dst = 0, len = 0;
      if (r < e_all) {
        uint32_t j, total;   // row j of an area of `total` bytes per stage
        if (r < e_scan) { j = r; base = sdata; total = col_unit * sbits; dst = 0u; }
        else if (r < e_val) { j = r - e_scan; base = xdata; total = col_unit * vbits; dst = off_val; }
        else if (r < e_g0) { j = r - e_val; base = p.gcols[0].data; total = col_unit * gbits[0]; dst = off_g0; }
        else if (r < e_g1) { j = r - e_g0; base = p.gcols[NG - 1].data; total = col_unit * gbits[NG - 1]; dst = off_g1; }
        else {
          const uint32_t b = (r - e_g1) / rows_per_bm;
          j = (r - e_g1) % rows_per_bm; total = col_unit; dst = off_bm + b * col_unit;
          base = p.pipe_tail;   // (b == n_bm: the upsert snapshot)
          // (compile-time slot indices only: a run-time index into the kernel argument makes hipcc copy it to scratch memory)
#pragma unroll
          for (int s = 0; s < 8; s++) if (b == (uint32_t)s && (uint32_t)s < n_bm) base = p.dense_ptr[s];
        }
        stride = total; dst += j * 1024u; len = total - j * 1024u < 1024u ? total - j * 1024u : 1024u;
        base += (size_t)j * 1024u;
      }
      r_src[q] = sw_sgpr_ptr<uint8_t>(base);
      r_stride[q] = (uint32_t)uniform((int)stride); r_dst[q] = (uint32_t)uniform((int)dst); r_len[q] = (uint32_t)uniform((int)len);
    }
  }
  const uint32_t pc = (uint32_t)lane * 16u;
  auto request_stage = [&](int k, uint32_t buf) __attribute__((always_inline)) {
    const int st = first + (k < n_mine ? k : n_mine - 1) * step;   // (past the end: the last stage again, never consumed)
    uint8_t* dst = stage0 + buf * stage_bytes;
#pragma unroll
    for (int q = 0; q < SW_ROWS_MAX; q++) {
      if (r_len[q] != 0u) {          // wave-uniform
        if (pc < r_len[q])           // (the short last row of an area: the lanes past it sit out — their 16 bytes would land in the next area)
          __builtin_amdgcn_global_load_lds(r_src[q] + (size_t)st * (size_t)r_stride[q] + pc, (LdsByte*)(dst + r_dst[q]), 16, 0, SW_DMA_AUX);
      }
    }
  };

  const uint32_t R = (uint32_t)p.replicas;
  const uint32_t rep = (uint32_t)t & (R - 1u);
  const uint32_t stride = table_slots;
  const uint32_t trash_slot = real_slots + (uint32_t)lane;
  // per-lane constants of the scan fields: LDS address of the dword pair, byte selector, and the (wave-uniform) shift
  uint32_t sc_at[8], sc_sel[8], sc_sh[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t P = mul24((uint32_t)lane * 8u + (uint32_t)j, sbits);
    sc_at[j] = (P >> 5) << 2;
    sc_sel[j] = oct_selector((P >> 3) & 3u);
    sc_sh[j] = (uint32_t)uniform((int)(32u - (((uint32_t)j * sbits) & 7u) - (sbits > 24u ? 32u : sbits)));   // (raw INT: 32 bits, byte aligned)
  }
  const uint32_t sc_mask = sbits >= 32u ? 0xFFFFFFFFu : (1u << sbits) - 1u;
  const uint32_t lin_sh = ((uint32_t)lane & 3u) * 8u;
  const RangeI32 r32 = HAS_SCAN ? make_range_i32(L.lo, L.hi) : RangeI32{0, 0u, false};
  uint32_t my_matched = 0, my_cand = 0;
  // the accumulators as 2-bit codes in one 64-bit scalar (0 COUNT, 1 SUM, 2 MIN, 3 MAX)
  const int n_ops = uniform(p.n_ops);
  uint64_t ops_code = 0;
  for (int o = 0; o < n_ops; o++) {
    const PgAccOp op = p.ops[uniform(o)];
    ops_code |= (uint64_t)(op.src < 0 ? 0u : (op.fn == PG_ACC_SUM ? 1u : (op.fn == PG_ACC_MIN ? 2u : 3u))) << (2 * o);
  }
  ops_code = ((uint64_t)(uint32_t)uniform((int)(uint32_t)(ops_code >> 32)) << 32) | (uint64_t)(uint32_t)uniform((int)(uint32_t)ops_code);
  const int has_out_words = uniform(p.out_words != nullptr ? 1 : 0);
  const uint32_t vbase = (uint32_t)uniform(p.specd_base), vstep = (uint32_t)uniform(p.specd_step);
  const GAS int32_t* vdict = VK == SW_V_GATHER ? sw_sgpr_ptr<int32_t>(p.srcs[p.pipe_src].dict) : nullptr;
  uint32_t gmul[NG];
#pragma unroll
  for (int gi = 0; gi < NG; gi++) gmul[gi] = (uint32_t)uniform((int)((uint32_t)p.gcols[gi].mult * R));
  const uint32_t list_dummy = OCT_SUB_DOCS + (uint32_t)lane;
  // this wavefront's sub-tile inside a stage buffer
  const uint32_t sub_scan = (uint32_t)wave * SW_SUB_BYTES * sbits, sub_val = off_val + (uint32_t)wave * SW_SUB_BYTES * vbits;
  const uint32_t sub_g0 = off_g0 + (uint32_t)wave * SW_SUB_BYTES * gbits[0], sub_g1 = off_g1 + (uint32_t)wave * SW_SUB_BYTES * gbits[NG - 1];
  const uint32_t sub_bm = off_bm + (uint32_t)wave * SW_SUB_BYTES + (uint32_t)lane;   // this lane's byte of bitmap 0

  auto field_pair = [&](const uint8_t* col, uint32_t doc, uint32_t bits) __attribute__((always_inline)) -> u32x2 {
    return *reinterpret_cast<const u32x2_a4*>(col + ((mul24(doc, bits) >> 5) << 2));
  };
  auto field_of = [&](u32x2 w, uint32_t doc, uint32_t bits) __attribute__((always_inline)) -> uint32_t {
    const uint32_t P = mul24(doc, bits);
    const uint32_t s8 = (P >> 3) & 3u;   // the field's first byte inside the pair
    const uint32_t be = perm(w.y, w.x, perm(s8, s8, 0u) + 0x00010203u);
    return bits >= 32u ? be : bfe(be, 32u - (P & 7u) - bits, bits);
  };
  auto value_of = [&](uint32_t id) __attribute__((always_inline)) -> int32_t {
    if (VK == SW_V_RAW32) return (int32_t)id;
    if (VK == SW_V_AFFINE) return (int32_t)mad24(id, vstep, vbase);   // dictId, step < 2^24 (planner); the sum wraps to the int value
    return vdict[id];
  };
  auto apply = [&](uint32_t slot, int32_t v) __attribute__((always_inline)) {
    for (int o = 0; o < n_ops; o++) {
      const uint32_t code = (uint32_t)(ops_code >> (2 * o)) & 3u;   // 0 COUNT, 1 SUM, 2 MIN, 3 MAX
      int64_t* base = lds_table + (size_t)o * stride;
      if (code == 0u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot), 1ULL);
      else if (code == 1u) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot), (unsigned long long)(int64_t)v);
      else if (code == 2u) atomicMin(reinterpret_cast<long long*>(base + slot), (long long)v);
      else atomicMax(reinterpret_cast<long long*>(base + slot), (long long)v);
    }
  };
  // SW_V_GATHER: the look-ups of a sub-tile's first two rounds are applied behind the NEXT stage's filter (they travel meanwhile)
  uint32_t pend_slot0 = 0, pend_slot1 = 0;
  int32_t pend_v0 = 0, pend_v1 = 0;
  int pend_n = 0;
  auto flush_pending = [&]() __attribute__((always_inline)) {
    if (pend_n > 0) apply(pend_slot0, pend_v0);
    if (pend_n > 1) apply(pend_slot1, pend_v1);
    pend_n = 0;
  };
  auto consume = [&](int k, const uint8_t* sb) __attribute__((always_inline)) {   // sb: the stage buffer
    const int64_t sub_index = ((int64_t)(first + k * step)) * SW_WAVES + wave;    // this sub-tile of the segment
    const int64_t rem = (int64_t)p.num_docs - sub_index * OCT_SUB_DOCS - (int64_t)lane * 8;
    uint32_t m = rem >= 8 ? 0xFFu : (rem > 0 ? (1u << (uint32_t)rem) - 1u : 0u);
    if (HAS_INDEX) {
      uint32_t bm[8];
#pragma unroll
      for (int j = 0; j < 8; j++) bm[j] = ((idx_live >> j) & 1u) ? (uint32_t)sb[sub_bm + (uint32_t)__builtin_popcount(idx_live & ((1u << j) - 1u)) * col_unit] : 0u;
      uint32_t acc = 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if ((idx_live >> j) & 1u) {          // wave-uniform
          if ((idx_first >> j) & 1u) {       // slot j opens a group: close the previous one
            if (j > 0) m &= ((idx_excl >> (j - 1)) & 1u) ? ~acc : acc;
            acc = 0u;
          }
          acc |= bm[j];
        }
      }
      m &= ((idx_excl >> 7) & 1u) ? ~acc : acc;
    }
    my_cand += (uint32_t)__popc(m);   // the scan leaf's candidates (numEntriesScannedInFilter)
    if (HAS_TAIL) m &= (uint32_t)sb[sub_bm + n_bm * col_unit];
    if (HAS_SCAN) {
      uint32_t rm = 0;
      u32x2 w[8];   // all eight pairs requested before the first is used (one LDS round trip, not eight)
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = *reinterpret_cast<const u32x2_a4*>(sb + sub_scan + sc_at[j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t id = (perm(w[j].y, w[j].x, sc_sel[j]) >> sc_sh[j]) & sc_mask;
        rm |= (uint32_t)in_range_i32(r32, (int32_t)id) << j;
      }
      m &= r32.empty ? 0u : rm;
    }
    my_matched += (uint32_t)__popc(m);
    if (has_out_words) {   // the sub-tile's match words, linear layout: lanes 4 g .. 4 g + 3 hold the bytes of its dword g
      const uint32_t word = sw_or_reduce4(m << lin_sh);
      if ((lane & 3) == 0) reinterpret_cast<uint32_t*>(p.out_words)[sub_index * 16 + (int64_t)(lane >> 2)] = word;
    }
#ifdef PG_SW_NO_TABLE   // measurement variant (wrong results): the filter alone
    return;
#endif
    // the selection list in position-major order: one ballot per doc position ranks its matches; a doc that does not match writes to the lane's dummy entry
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const bool hit = ((m >> j) & 1u) != 0u;
      const uint64_t b = __builtin_amdgcn_ballot_w64(hit);
      const uint32_t rank = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
      my_list[hit ? rank : list_dummy] = (uint16_t)((uint32_t)lane * 8u + (uint32_t)j);
      total += (uint32_t)__builtin_popcountll(b);
    }
    total = (uint32_t)uniform((int)total);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (a wavefront's LDS operations execute in order: the reads below see the writes above)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (VK == SW_V_GATHER) flush_pending();
    int round = 0;
    for (uint32_t at = 0; at < total; at += 64u, round++) {
      const uint32_t idx = at + (uint32_t)lane;
      const bool live = idx < total;
      const uint32_t doc = live ? (uint32_t)my_list[idx] : 0u;
      const u32x2 wv = field_pair(sb + sub_val, doc, vbits);
      u32x2 wg[NG];
#pragma unroll
      for (int gi = 0; gi < NG; gi++) wg[gi] = field_pair(sb + (gi == 0 ? sub_g0 : sub_g1), doc, gbits[gi]);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t vid = field_of(wv, doc, vbits);
      uint32_t slot = rep;
#pragma unroll
      for (int gi = 0; gi < NG; gi++) slot = mad24(field_of(wg[gi], doc, gbits[gi]), gmul[gi], slot);   // < 65536 slots (planner)
      slot = live ? slot : trash_slot;
      const int32_t v = value_of(live ? vid : 0u);
      if (VK == SW_V_GATHER && round < 2) {   // applied one stage later
        if (round == 0) { pend_slot0 = slot; pend_v0 = v; } else { pend_slot1 = slot; pend_v1 = v; }
        pend_n = round + 1;
      } else {
        apply(slot, v);
      }
    }
  };

  // ---- main loop: stage k is filtered and aggregated out of buffer k & 1 while stage k + 1 lands in the other.  Behind barrier k every
  // wavefront's rows of stage k have landed (each waited for its own: vmcnt(0)) and every wavefront has left stage k - 1 — whose buffer the
  // request for stage k + 1 then overwrites ------------------------------------------------------------------------------------------------
  __syncthreads();
  if (n_mine > 0) {
    request_stage(0, 0u);
    for (int k = 0; k < n_mine; k++) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
      sw_barrier();
      request_stage(k + 1, (uint32_t)((k + 1) & 1));
      consume(k, stage0 + (uint32_t)(k & 1) * stage_bytes);
    }
    if (VK == SW_V_GATHER) flush_pending();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (the last, unused request)
  }
  {
    const uint32_t wsum = wave_sum_u32(my_matched), csum = wave_sum_u32(my_cand);
    if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
    if (HAS_SCAN && !p.fast_scan_pushed && lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);   // (a pushed scan covers the segment: the host adds numDocs)
  }
  __syncthreads();
  // statistics and this workgroup's partial table [n_ops][n_groups], replicas folded
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  {
    const int Rr = p.replicas, groups = p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * ((int64_t)p.n_ops * groups);
    for (int o = 0; o < p.n_ops; o++) {
      const int fn = p.ops[uniform(o)].fn;   // integer accumulators only (planner)
      for (int gq = t; gq < groups; gq += PG_BLOCK) {
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

// one kernel per filter shape and value kind; the group-column count is a wave-uniform branch between two bodies
#define PG_SPECW_KERNEL(NAME, IDX, SCAN, TAIL, VK) \
  extern "C" __global__ void __launch_bounds__(PG_BLOCK, SW_MIN_WAVES_PER_SIMD) NAME(const PgQueryPlan p) { \
    if (p.n_group_cols == 1) specw_body<1, IDX, SCAN, TAIL, VK>(p); \
    else specw_body<2, IDX, SCAN, TAIL, VK>(p); \
  }
#define PG_SPECW_FAMILY(SUFFIX, VK) \
  PG_SPECW_KERNEL(pg_fast_dictrange_w##SUFFIX, true, true, false, VK)     /* the headline shape: dense index program AND range scan */ \
  PG_SPECW_KERNEL(pg_fast_dictrange_wt##SUFFIX, true, true, true, VK)     /* ... behind an upsert snapshot */ \
  PG_SPECW_KERNEL(pg_specw_none##SUFFIX, false, false, false, VK)         /* no filter */ \
  PG_SPECW_KERNEL(pg_specw_scan##SUFFIX, false, true, false, VK)          /* the range scan is the whole filter */ \
  PG_SPECW_KERNEL(pg_specw_index##SUFFIX, true, false, false, VK)         /* inverted-index leaves only */
PG_SPECW_FAMILY(_r, SW_V_RAW32)    // value column raw INT (the scan column is dictionary-encoded)
PG_SPECW_FAMILY(_a, SW_V_AFFINE)   // value = base + step x dictId
PG_SPECW_FAMILY(_g, SW_V_GATHER)   // value = dictionary[dictId]
#undef PG_SPECW_FAMILY
#undef PG_SPECW_KERNEL
