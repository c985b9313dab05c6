// pg_fast_dictrange_s: the headline shape over Pinot's DEFAULT encoding (round 6) — the scan column and / or the value column
// are dictionary-encoded: fixed-bit dictId streams (FixedBitSVForwardIndexReaderV2.java:65-99), the range predicate is a dictId interval
// [start, end) found by binary search in the sorted dictionary at plan time (RangePredicateEvaluatorFactory.java:126-167) and the aggregated
// value is dictionary.get(dictId) (DataFetcher.java:335-386).
//
// Design: 16 independent wavefronts per CU, NO barrier in the main loop, each with a PRIVATE strip of LDS that serves as its transposition
// buffer.  A wavefront walks wave tiles (2 048 docs) as four sub-tiles of 512 docs:
//   * stream: the sub-tile's bytes of every column — 64 x bits bytes, contiguous — arrive as coalesced 16 B / lane loads (a 20-bit column:
//     80 lanes' worth, two instructions), two sub-tiles in flight per wavefront (~110 KB per CU), and are stored to the strip as they are;
//     the tile's posting dwords (linear layout: 32 docs per lane) arrive once per tile and the dense index program runs over them at once;
//   * transpose by address: lane L owns docs 8 L .. 8 L + 7 of the sub-tile (oct layout).  Field j of any width <= 24 bits (or 32: raw INT) is
//     read WITHOUT a branch on the width: bit P = (8 L + j) x bits of the strip's MSB-first stream, the dword pair that holds it is read at a
//     per-lane address, one byte permute (per-lane selector) puts the 4 bytes from byte P >> 3 on into big-endian order, and the field then
//     starts (j x bits) & 7 bits below the top — the same for every lane: one shift, one mask.  (A switch over the widths with compile-time field
//     positions is a chain of taken scalar branches per sub-tile: 870 cycles for 50 vector instructions, profiles/r06_specd_steps.txt; an LDS
//     read at an address that is not dword aligned is served one lane per cycle, tools/probes/lds_unaligned.hip.)
//   * filter: range test on the 8 dictIds (or raw values) AND the candidates' byte of the linear dword (one ds_bpermute per sub-tile);
//   * aggregate in proportion to the MATCHES: one ballot per doc position ranks the matching docs, they are compacted into a selection list
//     (uint16 per match, in the strip) and the list is walked 64 docs at a time with every lane busy — value dictId and group dictIds gathered
//     from the strip (the same two-dword window with per-lane positions), value = raw | base + step x dictId (arithmetic dictionaries, found at
//     registration) | dictionary[dictId] (gathered from the L2-resident native-endian copy, applied one sub-tile later), ONE LDS atomic per
//     accumulator and 64 matches.  The position-wise form (8 docs per lane, misses aimed at a trash slot) issued 16 atomics per sub-tile
//     whatever matched and kept the CU's LDS pipe busy 60 % of the time at 47 % of the roofline.
// The first cut of this file was the loader / consumer frame of pg_fast_i32range_s with these consumers (4 + 8 wavefronts, one barrier per two
// tiles): 44-47 % of 8 TB/s whatever the consumers did — each stage was a chain of three LDS round trips on every consumer at the same time,
// between two rendezvous of all twelve wavefronts.  Independent wavefronts overlap one another's round trips instead.
// Bytes per doc: (scan bits + value bits + group bits) / 8 + postings — 6.625 B for config 3 with 20-bit r_int / m against 9.625 raw.
#ifndef SD_WAVES
#define SD_WAVES 16     // wavefronts per workgroup
#endif
#ifndef SD_SETS
#define SD_SETS 1       // sub-tiles of loads in flight per wavefront (register sets); 2: measured the same, and the scan fields' eight LDS reads then serialise on one register pair
#endif
// DMA (template parameter): the columns' bytes go global memory -> LDS directly (global_load_lds_dwordx4: no staging registers, no
// ds_write_b128), into one of two column areas of the strip; otherwise through registers into a single area.  +3-4 % with non-temporal DMA
// (profiles/r06_specd_steps.txt) for twice the strip: the planner takes the DMA kernels (pg_fast_dictrange_s_*_dma) where the table still
// keeps >= 8 replicas beside the doubled strips (PgQueryPlan::specd_dma)
#ifndef SD_DMA_AUX
#define SD_DMA_AUX 2    // cache policy bits of the LDS-DMA loads (2: non-temporal)
#endif
#ifndef SD_MIN_WAVES_PER_SIMD
#define SD_MIN_WAVES_PER_SIMD 4   // register budget: 4 -> 128 VGPRs, 5 -> 96, 6 -> 80 (two workgroups per CU where their LDS fits)
#endif
#ifndef SD_PIPE_ROUNDS
#define SD_PIPE_ROUNDS 0
#endif
#ifndef SD_EXEC_ROUNDS
#define SD_EXEC_ROUNDS 0
#endif
#define PG_WAVES_PER_BLOCK SD_WAVES
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"
#include "pg_oct_layout.h"

#define SD_PAD 16u                                      // a field's dword pair may reach one dword past the column's last byte
#define SD_LIST_BYTES ((OCT_SUB_DOCS + 64u) * 2u)       // a wavefront's selection list: 512 uint16 entries + a dummy entry per lane
__host__ __device__ static inline uint32_t sd_region(uint32_t bits) { return bits ? (bits * 64u + SD_PAD + 15u) & ~15u : 0u; }
extern "C" const int pg_specd_waves_per_block = PG_WAVES_PER_BLOCK;
// bytes of the launch's LDS behind the table and its trash slots: one strip per wavefront (`areas` column areas — 1, or 2 for the DMA kernels —
// of the sub-tile's column bytes + the selection list)
extern "C" int pg_specd_stage_bytes(int scan_bits, int value_bits, int bits0, int bits1, int areas) {
  const int strip = areas * (int)(sd_region((uint32_t)scan_bits) + sd_region((uint32_t)value_bits) + sd_region((uint32_t)bits0) + sd_region((uint32_t)bits1)) + (int)SD_LIST_BYTES;
  return PG_WAVES_PER_BLOCK * strip + 16;
}

template <typename T> DEVFN const GAS T* sd_sgpr_ptr(const void* ptr) {
  const uint64_t v = (uint64_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const GAS T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}
DEVFN uint32_t or_reduce4(uint32_t v) {   // OR across aligned groups of 4 lanes
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  return v;
}

// the loads of one sub-tile in flight: two 16-byte pieces per lane of the scan and the value column (64 x bits <= 2 048 bytes), one of a group column
template <int NG> struct SdSub { u32x4 sc[2], va[2], g[NG > 0 ? NG : 1]; };   // NG = 0: no GROUP BY (one group, the slot is the lane's replica)
// value kinds (PgQueryPlan::specd_vkind)
enum { SD_V_RAW32 = 1, SD_V_AFFINE = 2, SD_V_GATHER = 3 };

// HAS_INDEX: the fused dense index program (else every valid doc is a candidate); HAS_SCAN: one range scan (dictId interval of a fixed-bit
// column, or raw INT range) restricted to the candidates; HAS_TAIL: the upsert queryableDocIds snapshot ANDed in after the candidates have been
// counted (FilterPlanNode.run's outer AND); VK: how a dictId of the value column becomes the value.
template <int NG, bool HAS_INDEX, bool HAS_SCAN, bool HAS_TAIL, int VK, bool DMA>
__device__ __forceinline__ void specd_body(const PgQueryPlan& p) {
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
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
  uint32_t gbits[NG > 0 ? NG : 1] = {0u};
#pragma unroll
  for (int gi = 0; gi < NG; gi++) gbits[gi] = (uint32_t)uniform(p.gcols[gi].bits);
  // this wavefront's strip: one or two column areas [scan bytes][value bytes][group bytes ...], then the selection list
  const uint32_t off_val = sd_region(sbits), off_g0 = off_val + sd_region(vbits), off_g1 = off_g0 + sd_region(gbits[0]);
  const uint32_t area_bytes = off_g1 + (NG > 1 ? sd_region(gbits[NG - 1]) : 0u);   // one sub-tile's column bytes
  const uint32_t off_list = (DMA ? 2u : 1u) * area_bytes;
  const uint32_t strip_bytes = off_list + SD_LIST_BYTES;
  uint8_t* strip = reinterpret_cast<uint8_t*>(smem) + (((size_t)p.n_ops * table_slots * 8u + 15u) & ~(size_t)15u) + (uint32_t)wave * strip_bytes;
  uint16_t* my_list = reinterpret_cast<uint16_t*>(strip + off_list);
  const CAS PgScanLeaf& L = cptr(p.scans)[HAS_SCAN ? p.fast_scan : 0];   // only dereferenced when HAS_SCAN
  // tiles of this wavefront: blockIdx.x * 16 + wave, + gridDim.x * 16, ...
  const int step = (int)gridDim.x * PG_WAVES_PER_BLOCK;
  const int first = (int)blockIdx.x * PG_WAVES_PER_BLOCK + wave;
  const int n_mine = first < p.n_wtiles ? (p.n_wtiles - first + step - 1) / step : 0;
  const uint8_t* xdata = p.srcs[p.pipe_src].data;
  const uint8_t* sdata = HAS_SCAN ? L.data : xdata;

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
  // the 16-byte pieces this lane carries of a sub-tile: piece q of a column covers bytes [1024 q + 16 lane, + 16) of its 64 x bits
  const uint32_t pc = (uint32_t)lane * 16u;
  bool s_on[2], v_on[2], g_on[NG > 0 ? NG : 1] = {false};
#pragma unroll
  for (int q = 0; q < 2; q++) { s_on[q] = pc + 1024u * q < 64u * sbits; v_on[q] = pc + 1024u * q < 64u * vbits; }
#pragma unroll
  for (int gi = 0; gi < NG; gi++) g_on[gi] = pc < 64u * gbits[gi];
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
  const GAS int32_t* vdict = VK == SD_V_GATHER ? sd_sgpr_ptr<int32_t>(p.srcs[p.pipe_src].dict) : nullptr;
  uint32_t gmul[NG > 0 ? NG : 1] = {0u};
#pragma unroll
  for (int gi = 0; gi < NG; gi++) gmul[gi] = (uint32_t)uniform((int)((uint32_t)p.gcols[gi].mult * R));
  const uint32_t list_dummy = OCT_SUB_DOCS + (uint32_t)lane;
  // the dense index program as three scalar bit masks over the eight pointer slots: live (a real pointer), first (opens a group), and excl —
  // bit j: the group that is CLOSED behind slot j is complemented (bit 7: the last group)
  uint32_t idx_live = 0u, idx_first = 0u, idx_excl = 0u;
  if (HAS_INDEX) {
    // (compile-time slot indices only: a run-time index into the kernel argument makes hipcc copy it to scratch memory)
    bool pad = true;   // slots from the top down that repeat slot 0: the planner's padding
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
  __syncthreads();

  // ---- the stream ------------------------------------------------------------------------------------------------------------------------
  auto tile_of = [&](int k) __attribute__((always_inline)) { return first + (k < n_mine ? k : n_mine - 1) * step; };   // (past the end: the last tile again, never consumed)
  // Lanes past a column's bytes re-read its first 16 bytes of the sub-tile (a line the wavefront requests anyway): no load sits under a branch.
  auto issue_sub = [&](int k, int sub, SdSub<NG>& r) __attribute__((always_inline)) {
    const int wt = tile_of(k);
    __builtin_amdgcn_sched_barrier(0);
    if (HAS_SCAN) {
      const uint8_t* base = sdata + ((size_t)wt * 4u + (size_t)sub) * 64u * (size_t)sbits;
#pragma unroll
      for (int q = 0; q < 2; q++) r.sc[q] = ldnt((const GAS u32x4*)(sd_sgpr_ptr<uint8_t>(base) + (s_on[q] ? pc + 1024u * q : 0u)));
    }
    {
      const uint8_t* base = xdata + ((size_t)wt * 4u + (size_t)sub) * 64u * (size_t)vbits;
#pragma unroll
      for (int q = 0; q < 2; q++) r.va[q] = ldnt((const GAS u32x4*)(sd_sgpr_ptr<uint8_t>(base) + (v_on[q] ? pc + 1024u * q : 0u)));
    }
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      const uint8_t* base = p.gcols[gi].data + ((size_t)wt * 4u + (size_t)sub) * 64u * (size_t)gbits[gi];
      r.g[gi] = ldnt((const GAS u32x4*)(sd_sgpr_ptr<uint8_t>(base) + (g_on[gi] ? pc : 0u)));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_sub = [&](const SdSub<NG>& r) __attribute__((always_inline)) {
    if (HAS_SCAN) {
#pragma unroll
      for (int q = 0; q < 2; q++) if (s_on[q]) *reinterpret_cast<u32x4*>(strip + pc + 1024u * q) = r.sc[q];
    }
#pragma unroll
    for (int q = 0; q < 2; q++) if (v_on[q]) *reinterpret_cast<u32x4*>(strip + off_val + pc + 1024u * q) = r.va[q];
#pragma unroll
    for (int gi = 0; gi < NG; gi++) if (g_on[gi]) *reinterpret_cast<u32x4*>(strip + (gi == 0 ? off_g0 : off_g1) + pc) = r.g[gi];
  };
  typedef __attribute__((address_space(3))) uint8_t LdsByte;
  // sub-tile (k, sub) of every column straight into column area `area` of the strip: lane l's 16 bytes land at piece base + 16 l
  auto dma_sub = [&](int k, int sub, uint32_t area) __attribute__((always_inline)) {
    const int wt = tile_of(k);
    LdsByte* dst = (LdsByte*)(strip + area * area_bytes);
    if (HAS_SCAN) {
      const GAS uint8_t* src = gptr<uint8_t>(sdata + ((size_t)wt * 4u + (size_t)sub) * 64u * (size_t)sbits) + pc;
#pragma unroll
      for (int q = 0; q < 2; q++) if (s_on[q]) __builtin_amdgcn_global_load_lds(src + 1024u * q, dst + 1024u * q, 16, 0, SD_DMA_AUX);
    }
    {
      const GAS uint8_t* src = gptr<uint8_t>(xdata + ((size_t)wt * 4u + (size_t)sub) * 64u * (size_t)vbits) + pc;
#pragma unroll
      for (int q = 0; q < 2; q++) if (v_on[q]) __builtin_amdgcn_global_load_lds(src + 1024u * q, dst + off_val + 1024u * q, 16, 0, SD_DMA_AUX);
    }
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      const GAS uint8_t* src = gptr<uint8_t>(p.gcols[gi].data + ((size_t)wt * 4u + (size_t)sub) * 64u * (size_t)gbits[gi]) + pc;
      if (g_on[gi]) __builtin_amdgcn_global_load_lds(src, dst + (gi == 0 ? off_g0 : off_g1), 16, 0, SD_DMA_AUX);
    }
  };
  uint32_t post[8], tail = 0;
  auto issue_post = [&](int k) __attribute__((always_inline)) {
    const int wt = tile_of(k);
    if (HAS_INDEX) {
#pragma unroll
      for (int j = 0; j < 8; j++) post[j] = ldnt((const GAS uint32_t*)(sd_sgpr_ptr<uint8_t>(p.dense_ptr[j] + (size_t)wt * 256u) + (uint32_t)lane * 4u));
    }
    if (HAS_TAIL) tail = ldnt((const GAS uint32_t*)(sd_sgpr_ptr<uint8_t>(p.pipe_tail + (size_t)wt * 256u) + (uint32_t)lane * 4u));
    __builtin_amdgcn_sched_barrier(0);
  };
  // the index program over this lane's 32 docs of tile k (linear layout); counts the scan's candidates
  auto tile_candidates = [&](int k) __attribute__((always_inline)) -> uint32_t {
    const int wt = first + k * step;
    const int64_t rem = (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS;
    const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
    uint32_t lin = valid_lin_mask(n_valid, lane);
    if (HAS_INDEX) {
      // the planner lays the eight pointers out group after group (slots past the last real pointer repeat slot 0 and are not looked at): a
      // running OR, folded into the result where the group changes — two scalar bit tests per slot.  (Selecting every slot into four group
      // registers by compare masks kept 32 64-bit masks alive: ~180 instructions and 64 spilled-SGPR reloads per tile.)
      uint32_t acc = 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if ((idx_live >> j) & 1u) {          // wave-uniform
          if ((idx_first >> j) & 1u) {       // slot j opens a group: close the previous one
            if (j > 0) lin &= ((idx_excl >> (j - 1)) & 1u) ? ~acc : acc;
            acc = 0u;
          }
          acc |= post[j];
        }
      }
      lin &= ((idx_excl >> 7) & 1u) ? ~acc : acc;
    }
    my_cand += (uint32_t)__popc(lin);   // the scan leaf's candidates (numEntriesScannedInFilter)
    return HAS_TAIL ? lin & tail : lin;
  };

  // ---- filter and aggregation of the sub-tile in the strip --------------------------------------------------------------------------------
  // one field of a column's bytes in the strip: doc `doc` of the sub-tile, width <= 24 (or 32: raw INT) — the dword pair is requested by
  // field_pair (every field of a round first: one LDS round trip), cut by field_of
  auto field_pair = [&](const uint8_t* col, uint32_t doc, uint32_t bits) __attribute__((always_inline)) -> u32x2 {
    return *reinterpret_cast<const u32x2_a4*>(col + ((mul24(doc, bits) >> 5) << 2));
  };
  auto field_of = [&](u32x2 w, uint32_t doc, uint32_t bits) __attribute__((always_inline)) -> uint32_t {
    const uint32_t P = mul24(doc, bits);
    const uint32_t s8 = (P >> 3) & 3u;   // the field's first byte inside the pair
    const uint32_t be = perm(w.y, w.x, perm(s8, s8, 0u) + 0x00010203u);   // (oct_selector(s8): s8 in every byte + 0, 1, 2, 3 — a 24-bit multiply would cut the constant)
    return bits >= 32u ? be : bfe(be, 32u - (P & 7u) - bits, bits);
  };
  auto value_of = [&](uint32_t id) __attribute__((always_inline)) -> int32_t {
    if (VK == SD_V_RAW32) return (int32_t)id;
    if (VK == SD_V_AFFINE) return (int32_t)mad24(id, vstep, vbase);   // dictId, step < 2^24 (planner); the sum wraps to the int value
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
  // SD_V_GATHER: the look-ups of a sub-tile's first two rounds are applied behind the NEXT sub-tile's filter (they travel meanwhile); plain
  // scalars, not an indexed struct: hipcc kept the struct in scratch memory
  uint32_t pend_slot0 = 0, pend_slot1 = 0;
  int32_t pend_v0 = 0, pend_v1 = 0;
  int pend_n = 0;
  auto flush_pending = [&]() __attribute__((always_inline)) {
    if (pend_n > 0) apply(pend_slot0, pend_v0);
    if (pend_n > 1) apply(pend_slot1, pend_v1);
    pend_n = 0;
  };
  auto consume = [&](int k, int sub, uint32_t lin, const uint8_t* cols) __attribute__((always_inline)) {   // cols: the sub-tile's column area
    // candidates of this lane's 8 docs: byte (lane & 3) of linear dword 16 sub + (lane >> 2) (docs past the segment: none)
    const uint32_t cd = (uint32_t)__builtin_amdgcn_ds_bpermute((sub * 16 + (lane >> 2)) * 4, (int)lin);
    uint32_t m = (cd >> lin_sh) & 0xFFu;
#ifdef PG_SD_LOADS_ONLY   // measurement variant (wrong results): the stream alone — loads, stores to the strip, one LDS read
    my_matched += *reinterpret_cast<const uint32_t*>(cols + (uint32_t)lane * 4u) & m & 1u;
    return;
#endif
    if (HAS_SCAN) {
      uint32_t rm = 0;
      u32x2 w[8];   // all eight pairs requested before the first is used (one LDS round trip, not eight)
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = *reinterpret_cast<const u32x2_a4*>(cols + sc_at[j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t id = (perm(w[j].y, w[j].x, sc_sel[j]) >> sc_sh[j]) & sc_mask;
        rm |= (uint32_t)in_range_i32(r32, (int32_t)id) << j;
      }
      m &= r32.empty ? 0u : rm;
#ifdef PG_SD_TWICE   // measurement variant: the scan fields decoded and tested a second time (what do ~56 more vector instructions per sub-tile cost?)
      {
        uint32_t rm2 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t id = (perm(w[j].x, w[j].y, sc_sel[j] ^ 0x01010101u) >> sc_sh[j]) & sc_mask;
          rm2 |= (uint32_t)in_range_i32(r32, (int32_t)id + 1) << j;
        }
        if (rm2 == 0x12345u) m ^= 1u;
      }
#endif
    }
    my_matched += (uint32_t)__popc(m);
    if (has_out_words) {   // the tile's match words, linear layout: lanes 4 g .. 4 g + 3 hold the bytes of dword 16 sub + g
      const int wt = first + k * step;
      const uint32_t word = or_reduce4(m << lin_sh);
      if ((lane & 3) == 0) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + (int64_t)(sub * 16 + (lane >> 2))] = word;
    }
#ifdef PG_SD_NO_TABLE   // measurement variant (wrong results): the filter alone
    return;
#endif
    // the selection list in position-major order (any order serves: the accumulators commute): one ballot per doc position ranks its matches;
    // a doc that does not match writes to the lane's dummy entry
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
    if (VK == SD_V_GATHER) flush_pending();
    if (total == 0u) return;
#if SD_PIPE_ROUNDS
    // two rounds per trip: both rounds' list entries, then both rounds' field pairs are requested before anything is used — a sub-tile with
    // 65 .. 128 matches (every second one at 1 match in 8) pays the chain list -> fields -> atomics once, not twice
    for (uint32_t at = 0; at < total; at += 128u) {
      const uint32_t idx0 = at + (uint32_t)lane, idx1 = idx0 + 64u;
      const bool live0 = idx0 < total, live1 = idx1 < total;
      const bool two = at + 64u < total;   // wave-uniform
      const uint32_t doc0 = live0 ? (uint32_t)my_list[idx0] : 0u;
      const uint32_t doc1 = live1 ? (uint32_t)my_list[idx1] : 0u;
      const u32x2 wv0 = field_pair(cols + off_val, doc0, vbits);
      u32x2 wg0[NG > 0 ? NG : 1], wg1[NG > 0 ? NG : 1];
#pragma unroll
      for (int gi = 0; gi < NG; gi++) wg0[gi] = field_pair(cols + (gi == 0 ? off_g0 : off_g1), doc0, gbits[gi]);
      const u32x2 wv1 = field_pair(cols + off_val, doc1, vbits);
#pragma unroll
      for (int gi = 0; gi < NG; gi++) wg1[gi] = field_pair(cols + (gi == 0 ? off_g0 : off_g1), doc1, gbits[gi]);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t vid0 = field_of(wv0, doc0, vbits);
      uint32_t slot0 = rep;
#pragma unroll
      for (int gi = 0; gi < NG; gi++) slot0 = mad24(field_of(wg0[gi], doc0, gbits[gi]), gmul[gi], slot0);   // < 65536 slots (planner)
      slot0 = live0 ? slot0 : trash_slot;
      const int32_t v0 = value_of(live0 ? vid0 : 0u);
      uint32_t slot1 = trash_slot;
      int32_t v1 = 0;
      if (two) {
        const uint32_t vid1 = field_of(wv1, doc1, vbits);
        slot1 = rep;
#pragma unroll
        for (int gi = 0; gi < NG; gi++) slot1 = mad24(field_of(wg1[gi], doc1, gbits[gi]), gmul[gi], slot1);
        slot1 = live1 ? slot1 : trash_slot;
        v1 = value_of(live1 ? vid1 : 0u);
      }
      if (VK == SD_V_GATHER && at == 0u) {   // applied one sub-tile later
        pend_slot0 = slot0; pend_v0 = v0; pend_slot1 = slot1; pend_v1 = v1;
        pend_n = two ? 2 : 1;
      } else {
#if SD_EXEC_ROUNDS
        if (live0) apply(slot0, v0);
        if (two) { if (live1) apply(slot1, v1); }
#else
        apply(slot0, v0);
        if (two) apply(slot1, v1);
#endif
      }
    }
#else
    const bool all = false;
    int round = 0;
    for (uint32_t at = 0; at < total; at += 64u, round++) {
      const uint32_t idx = at + (uint32_t)lane;
      const bool live = idx < total;
      const uint32_t doc = all ? idx : (live ? (uint32_t)my_list[idx] : 0u);
      const u32x2 wv = field_pair(cols + off_val, doc, vbits);
      u32x2 wg[NG > 0 ? NG : 1];
#pragma unroll
      for (int gi = 0; gi < NG; gi++) wg[gi] = field_pair(cols + (gi == 0 ? off_g0 : off_g1), doc, gbits[gi]);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t vid = field_of(wv, doc, vbits);
      uint32_t slot = rep;
#pragma unroll
      for (int gi = 0; gi < NG; gi++) slot = mad24(field_of(wg[gi], doc, gbits[gi]), gmul[gi], slot);   // < 65536 slots (planner)
      slot = live ? slot : trash_slot;
      const int32_t v = value_of(live ? vid : 0u);
      if (VK == SD_V_GATHER && round < 2) {   // applied one sub-tile later
        if (round == 0) { pend_slot0 = slot; pend_v0 = v; } else { pend_slot1 = slot; pend_v1 = v; }
        pend_n = round + 1;
      } else {
#if SD_EXEC_ROUNDS
        if (live) apply(slot, v);   // the dead lanes of a list's last round sit out (EXEC) instead of adding to their trash slots
#else
        apply(slot, v);
#endif
      }
    }
#endif
  };

  // ---- main loop: one tile per iteration, sub-tiles 0 .. 3.  SD_SETS = 2: two register sets, sub-tile (k, s + 2) is requested where (k, s) is
  // stored; SD_SETS = 1: one set — the next sub-tile travels while this one is filtered and aggregated (twice the wavefronts fit then) --------
  if (n_mine > 0) {
    issue_post(0);
    if constexpr (DMA) {
      // sub-tile u lands in column area u & 1 while sub-tile u - 1 is filtered and aggregated out of the other; what has to have landed is waited
      // for with vmcnt(0) just before the next request goes out (the posting dwords and the dictionary look-ups in flight are older than that)
      dma_sub(0, 0, 0u);
      for (int k = 0; k < n_mine; k++) {
        const uint32_t lin = tile_candidates(k);
        issue_post(k + 1);
#pragma unroll
        for (int sub = 0; sub < 4; sub++) {
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched: sub-tile (k, sub) is in its area
          if (sub < 3) dma_sub(k, sub + 1, (uint32_t)((sub + 1) & 1)); else dma_sub(k + 1, 0, 0u);
          consume(k, sub, lin, strip + (uint32_t)(sub & 1) * area_bytes);
        }
      }
    } else {
#if SD_SETS == 2
      SdSub<NG> ra, rb;
      issue_sub(0, 0, ra);
      issue_sub(0, 1, rb);
      for (int k = 0; k < n_mine; k++) {
        const uint32_t lin = tile_candidates(k);   // (waits for tile k's posting dwords only: the sub-tiles behind them stay in flight)
        issue_post(k + 1);
        store_sub(ra); issue_sub(k, 2, ra); consume(k, 0, lin, strip);
        store_sub(rb); issue_sub(k, 3, rb); consume(k, 1, lin, strip);
        store_sub(ra); issue_sub(k + 1, 0, ra); consume(k, 2, lin, strip);
        store_sub(rb); issue_sub(k + 1, 1, rb); consume(k, 3, lin, strip);
      }
#else
      SdSub<NG> ra;
      issue_sub(0, 0, ra);
      for (int k = 0; k < n_mine; k++) {
        const uint32_t lin = tile_candidates(k);
        issue_post(k + 1);
        store_sub(ra); issue_sub(k, 1, ra); consume(k, 0, lin, strip);
        store_sub(ra); issue_sub(k, 2, ra); consume(k, 1, lin, strip);
        store_sub(ra); issue_sub(k, 3, ra); consume(k, 2, lin, strip);
        store_sub(ra); issue_sub(k + 1, 0, ra); consume(k, 3, lin, strip);
      }
#endif
    }
    if (VK == SD_V_GATHER) flush_pending();
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
    if (NG == 0) {   // one group, a replica per lane: every lane folds its replica into replica 0 (one lane walking 1 024 replicas per accumulator took ~60 us)
      for (int o = 0; o < p.n_ops; o++) {
        const int fn = p.ops[uniform(o)].fn;
        long long* slot0 = reinterpret_cast<long long*>(lds_table + (size_t)o * table_slots);
        if (t > 0 && t < Rr) {
          const long long v = slot0[t];
          if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) atomicAdd(reinterpret_cast<unsigned long long*>(slot0), (unsigned long long)v);
          else if (fn == PG_ACC_MIN) atomicMin(slot0, v);
          else atomicMax(slot0, v);
        }
      }
      __syncthreads();
      if (t < p.n_ops) out[t] = lds_table[(size_t)t * table_slots];
      return;
    }
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
#define PG_SPECD_KERNEL(NAME, IDX, SCAN, TAIL, VK, DMA) \
  extern "C" __global__ void __launch_bounds__(PG_BLOCK, SD_MIN_WAVES_PER_SIMD) NAME(const PgQueryPlan p) { \
    if (p.n_group_cols == 1) specd_body<1, IDX, SCAN, TAIL, VK, DMA>(p); \
    else if (p.n_group_cols == 0) specd_body<0, IDX, SCAN, TAIL, VK, DMA>(p);   /* no GROUP BY: AggregationOperator's shapes */ \
    else specd_body<2, IDX, SCAN, TAIL, VK, DMA>(p); \
  }
#define PG_SPECD_FAMILY(SUFFIX, VK) \
  PG_SPECD_KERNEL(pg_fast_dictrange_s##SUFFIX, true, true, false, VK, false)     /* the headline shape: dense index program AND range scan */ \
  PG_SPECD_KERNEL(pg_fast_dictrange_s##SUFFIX##_dma, true, true, false, VK, true) /* ... its columns by LDS-DMA into two areas per strip */ \
  PG_SPECD_KERNEL(pg_fast_dictrange_st##SUFFIX, true, true, true, VK, false)     /* ... behind an upsert snapshot */ \
  PG_SPECD_KERNEL(pg_specd_none##SUFFIX, false, false, false, VK, false)         /* no filter */ \
  PG_SPECD_KERNEL(pg_specd_scan##SUFFIX, false, true, false, VK, false)          /* the range scan is the whole filter */ \
  PG_SPECD_KERNEL(pg_specd_index##SUFFIX, true, false, false, VK, false)         /* inverted-index leaves only */
PG_SPECD_FAMILY(_r, SD_V_RAW32)    // value column raw INT (the scan column is dictionary-encoded)
PG_SPECD_FAMILY(_a, SD_V_AFFINE)   // value = base + step x dictId
PG_SPECD_FAMILY(_g, SD_V_GATHER)   // value = dictionary[dictId]
#undef PG_SPECD_FAMILY
#undef PG_SPECD_KERNEL
