// pg_fast_i32range_s: the headline shape with SPECIALISED wavefronts (round 5, VERDICT r4 #4b).  Steps, measurement variants and SQ counters:
// profiles/r05_wave_specialised.txt; the default where the plan's last execution counted >= 15 % candidates (pg_exec.hip, spec_shape).
//
// pg_fast_i32range_p (pg_kernels_pipe.hip) runs 8 identical wavefronts per CU, each streaming AND aggregating; it sits at the stream its own
// loads reach with 8 wavefronts (80-84 % of 8 TB/s; 84 % with the aggregation compiled out), while the same loads issued by FOUR wavefronts
// reach 88-90 % (profiles/r02_scan_bw_probe.txt, r02_ab_pipeline.txt: the memory system prefers fewer, longer streams).  Four wavefronts cannot
// hide the LDS atomics themselves, hence this split of one 12-wavefront workgroup per CU:
//   * 4 LOADER wavefronts stream whole wave tiles (2 048 docs) from HBM into registers — the scan column, the value column and the group
//     columns as full coalesced 1 KB rows, SPEC_SETS stages of SPEC_TILES tiles in flight — and hand the oldest stage to the consumers through
//     one of two LDS stage buffers.  What a loader holds whole it evaluates on the way: the index program over the tile's posting dwords, once
//     per tile (-> 64 candidate dwords, linear layout); the range predicate on the scan quads (-> 4 result bits per lane: the scan column
//     never goes through LDS); the value column's byte swaps; the candidate count (numEntriesScannedInFilter);
//   * 8 CONSUMER wavefronts, one per quad row of a tile (256 docs, 4 per lane), read stage s + 1's slices out of LDS (candidate dword, range
//     bits, value quad, group windows) and then aggregate stage s's out of registers: match = candidates AND range bits, group decode, LDS
//     atomics into the workgroup's table — unpredicated: a doc that does not match aims at a per-lane trash slot behind its accumulator's row.
//     The reference's loop (SVScanDocIdIterator.java:213-291 + SumAggregationFunction.java:160-179), without a single global load.
// One barrier per stage, fenced on the LDS address space only: behind barrier s the consumers read buffer s & 1 while the loaders fill buffer
// (s + 1) & 1; a buffer is refilled only behind the next barrier, which the consumers reach after their reads of it have completed.
//
// Same plans, same results, same statistics as pg_fast_i32range_p (tests/test_gpu_headline_kernels.py runs both, and the general shapes the
// template also instantiates).  pg_fast_i32range_p stays the kernel of selective filters: it requests only the quads that hold a candidate.
#define PG_WAVES_PER_BLOCK 12
#define PG_KERNEL template <int PG_NOT_INSTANTIATED> static
#include "pg_kernels.hip"

#define SPEC_LOADERS 4
#ifndef SPEC_TILES
#define SPEC_TILES 2   // wave tiles per stage: one barrier per 4 096 docs, two independent quad rows per consumer wavefront
#endif
#ifndef SPEC_SETS
#define SPEC_SETS 2    // stages of loads in flight per loader wavefront (register sets).  Three: 1 % at best, and beyond 168 registers (3 wavefronts per SIMD) with two tiles
#endif
#define SPEC_CONSUMERS 8
// stage buffer, per wave tile, byte offsets
#define SPEC_OFF_LIN 0u                       // the index program's result: 64 dwords, linear layout (computed by a loader wavefront)
#define SPEC_OFF_RNG 256u                     // the range predicate's result: per quad row 64 dwords, bits 0 .. 3 = the lane's four docs (computed by the loaders)
#define SPEC_OFF_VAL (256u + 2048u)           // 2 048 raw INT values
#define SPEC_OFF_G0 (256u + 2048u + 8192u)    // bits x 256 bytes + 16 (the packed-quad window reads one dword past its values)
extern "C" const int pg_spec_waves_per_block = PG_WAVES_PER_BLOCK;
// bytes of one stage buffer, SPEC_TILES wave tiles (the host sizes the launch's LDS with it: table + 2 stages)
extern "C" int pg_spec_stage_bytes(int bits0, int bits1) { return SPEC_TILES * (((int)SPEC_OFF_G0 + bits0 * 256 + 16 + (bits1 > 0 ? bits1 * 256 + 16 : 0) + 15) & ~15); }

// Workgroup barrier that waits for this wavefront's LDS operations only.  __syncthreads() is a workgroup-scope release fence in front of
// s_barrier — s_waitcnt vmcnt(0) as well — which would drain the loaders' stages of global loads at every barrier.
// (Fences on the LDS address space alone — not asm with a "memory" clobber: that made the compiler re-read every plan field from the kernel
// argument behind each barrier, 48 scalar loads per tile, each one a stall — profiles/r05_wave_specialised.txt.)
DEVFN void spec_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T> DEVFN const GAS T* spec_sgpr_ptr(const void* ptr) {
  const uint64_t v = (uint64_t)ptr;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (const GAS T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}

// what a loader wavefront holds of one wave tile: 4 rows of the scan / value columns, 1 row of a group column — and of ONE tile of the stage
// (tile w % SPEC_TILES for loader w) the 8 posting dwords of its lane's 32 docs (+ the upsert snapshot's): loaders 0 .. SPEC_TILES - 1 evaluate
// the index program there, once per tile, as pg_fast_i32range_p does; evaluated by the consumers it ran once per QUAD ROW — 8 x 64 VALU
// instructions per tile (first cut, 2.35 ms)
struct SpecTile {
  u32x4 col[4];
  u32x4 grp;
};
struct SpecStage { SpecTile tile[SPEC_TILES]; uint32_t post[8]; uint32_t tail; };
// what a consumer lane reads of one wave tile: everything up front, one LDS round trip per stage
template <int NG> struct SpecQuad {
  uint32_t lin, rng;
  u32x4 val;
  uint32_t win[NG][2];
};

// HAS_INDEX: the fused dense index program (else every valid doc is a candidate); HAS_SCAN: one raw-INT range scan restricted to the candidates
// (else every candidate matches) — pipe_general_body's shapes without the tail / value-scan extras (pg_spec_none / _scan / _index).
// HAS_TAIL: one more dense bitmap ANDed in AFTER the scan — the upsert queryableDocIds snapshot of FilterPlanNode.run's outer AND, which must not
// restrict the scan's candidates (numEntriesScannedInFilter stays the reference's): the loader counts the candidates, then masks them.
template <int NG, bool HAS_INDEX, bool HAS_SCAN, bool HAS_TAIL = false>
__device__ __forceinline__ void spec_body(const PgQueryPlan& p) {
  constexpr int NCOL = HAS_SCAN ? 4 : 2;   // 1 KB rows of (scan column |) value column per loader and tile
  extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
  __shared__ uint32_t s_stat[PG_MAX_STATS];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = uniform(t >> 6);
  int64_t* lds_table = reinterpret_cast<int64_t*>(smem);
  if (t < PG_MAX_STATS) s_stat[t] = 0;
  // the table: per accumulator n_groups x replicas slots and 64 TRASH slots behind them (one per lane: where a doc that does not match aims)
  const uint32_t real_slots = (uint32_t)p.n_groups * (uint32_t)p.replicas;
  const uint32_t table_slots = real_slots + 64u;
  for (int o = 0; o < p.n_ops; o++) {
    const int64_t ident = pg_acc_identity(p.ops[o].fn, p.ops[o].is_float);
    for (uint32_t i = t; i < table_slots; i += PG_BLOCK) lds_table[(size_t)o * table_slots + i] = ident;
  }
  const uint32_t bits0 = (uint32_t)p.gcols[0].bits, bits1 = NG > 1 ? (uint32_t)p.gcols[NG - 1].bits : 0u;
  const uint32_t off_g1 = SPEC_OFF_G0 + bits0 * 256u + 16u;
  const uint32_t tile_bytes = (off_g1 + (NG > 1 ? bits1 * 256u + 16u : 0u) + 15u) & ~15u;
  const uint32_t stage_bytes = tile_bytes * SPEC_TILES;
  uint8_t* stage0 = reinterpret_cast<uint8_t*>(smem) + (((size_t)p.n_ops * table_slots * 8u + 15u) & ~(size_t)15u);
  const CAS PgScanLeaf& L = cptr(p.scans)[HAS_SCAN ? p.fast_scan : 0];   // only dereferenced when HAS_SCAN
  const int grid = (int)gridDim.x;
  const int n_mine = (int)blockIdx.x < p.n_wtiles ? (p.n_wtiles - (int)blockIdx.x + grid - 1) / grid : 0;   // tiles blockIdx.x, + grid, ...
  const int n_stages = (n_mine + SPEC_TILES - 1) / SPEC_TILES;   // stage s: tiles SPEC_TILES s .. of this workgroup's sequence
  __syncthreads();

  if (wave < SPEC_LOADERS) {
    // ---- loaders ------------------------------------------------------------------------------------------------------------------
    const int w = wave;
    const uint8_t* xdata = p.srcs[p.pipe_src].data;
    const RangeI32 r32 = HAS_SCAN ? make_range_i32(L.lo, L.hi) : RangeI32{0, 0u, false};
    uint32_t ld_cand = 0;
    const int ggi = (w >> 1) < NG ? (w >> 1) : 0;   // this loader's group column (loaders without one repeat column 0's first 16 bytes)
    const uint32_t goff = (uint32_t)(w & 1) * 1024u + (uint32_t)lane * 16u;
    const uint32_t geff = (w >> 1) < NG && goff < (uint32_t)p.gcols[ggi].bits * 256u ? goff : 0u;   // ... as do lanes past the column's row
    // slot s of loader w: row n = 4 s + w of the 16 rows (1 KB each) of scan column | value column
    auto issue = [&](int sidx, SpecStage& st) __attribute__((always_inline)) {
      // The scheduling barriers pin the ORDER of the loads of a stage, and of the stages: loads return in order and the wait counts in
      // front of publish() are derived per path into the loop — with the prologue's two stages interleaved by the scheduler the loop head
      // merged to vmcnt(0), i.e. every stage was waited for where the previous one was published.
#pragma unroll
      for (int tt = 0; tt < SPEC_TILES; tt++) {
        const int i = sidx * SPEC_TILES + tt;
        const int wt = (int)blockIdx.x + (i < n_mine ? i : n_mine - 1) * grid;   // (past the end: the last tile again, never consumed)
        SpecTile& tl = st.tile[tt];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NCOL; s++) {
          const int n = 4 * s + w;
          const uint8_t* base = (HAS_SCAN && n < 8 ? L.data : xdata) + (size_t)wt * (PG_WAVE_DOCS * 4) + (size_t)(n & 7) * 1024u;
          tl.col[s] = ldnt((const GAS u32x4*)(spec_sgpr_ptr<uint8_t>(base) + (uint32_t)lane * 16u));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const uint8_t* base = p.gcols[ggi].data + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)p.gcols[ggi].bits;
          tl.grp = ldnt((const GAS u32x4*)(spec_sgpr_ptr<uint8_t>(base) + geff));   // a tile of a packed column is bits x 256 bytes: 16-byte aligned
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (HAS_INDEX) {
        const int i = sidx * SPEC_TILES + (w % SPEC_TILES);
        const int wt = (int)blockIdx.x + (i < n_mine ? i : n_mine - 1) * grid;
#pragma unroll
        // (plain loads, not non-temporal ones: two loader wavefronts request each tile's posting dwords, the second should find them in L2 —
        // streaming loads left 1.027 x the algorithmic bytes in the HBM counters, profiles/r05_wave_specialised.txt)
        for (int j = 0; j < 8; j++) st.post[j] = *(const GAS uint32_t*)(spec_sgpr_ptr<uint8_t>(p.dense_ptr[j] + (size_t)wt * 256u) + (uint32_t)lane * 4u);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (HAS_TAIL) {
        const int i = sidx * SPEC_TILES + (w % SPEC_TILES);
        const int wt = (int)blockIdx.x + (i < n_mine ? i : n_mine - 1) * grid;
        st.tail = *(const GAS uint32_t*)(spec_sgpr_ptr<uint8_t>(p.pipe_tail + (size_t)wt * 256u) + (uint32_t)lane * 4u);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto publish = [&](uint8_t* stage, const SpecStage& st) __attribute__((always_inline)) {
#ifdef PG_SPEC_NO_PUBLISH   // measurement variant (wrong results): the loads are waited for, nothing is written to LDS
      {
        uint32_t acc = 0;
#pragma unroll
        for (int tt = 0; tt < SPEC_TILES; tt++) { for (int k = 0; k < NCOL; k++) acc += st.tile[tt].col[k].x ^ st.tile[tt].col[k].w; acc += st.tile[tt].grp.x; }
        if (acc == 0x12345678u) *reinterpret_cast<uint32_t*>(stage) = acc;
        return;
      }
#endif
#pragma unroll
      for (int tt = 0; tt < SPEC_TILES; tt++) {
        uint8_t* buf = stage + (uint32_t)tt * tile_bytes;
        const SpecTile& tl = st.tile[tt];
        // rows 0 .. 7 are the scan column: the loader holds quad (row, lane) — exactly what consumer `row`'s lane reads — and tests the range
        // there; four result bits per lane go to LDS instead of the row's 1 KB (the consumers' path is the long one: 20 VALU instructions and a
        // 16-byte LDS read per quad row less on it, on wavefronts that otherwise wait for HBM)
        if (HAS_SCAN) {
#pragma unroll
          for (int s = 0; s < 2; s++) {
            const u32x4 a = tl.col[s];
            uint32_t m = (uint32_t)in_range_i32(r32, (int32_t)bswap32(a.x)) | ((uint32_t)in_range_i32(r32, (int32_t)bswap32(a.y)) << 1) |
                         ((uint32_t)in_range_i32(r32, (int32_t)bswap32(a.z)) << 2) | ((uint32_t)in_range_i32(r32, (int32_t)bswap32(a.w)) << 3);
            m = r32.empty ? 0u : m;
            *reinterpret_cast<uint32_t*>(buf + SPEC_OFF_RNG + (uint32_t)(4 * s + w) * 256u + (uint32_t)lane * 4u) = m;
          }
        }
#pragma unroll
        for (int s = HAS_SCAN ? 2 : 0; s < NCOL; s++) {   // the value column: host byte order already (four bswaps per quad row less for the consumers)
          u32x4 v = tl.col[s];
          v.x = bswap32(v.x); v.y = bswap32(v.y); v.z = bswap32(v.z); v.w = bswap32(v.w);
          *reinterpret_cast<u32x4*>(buf + SPEC_OFF_VAL + (uint32_t)(4 * s + w - (HAS_SCAN ? 8 : 0)) * 1024u + (uint32_t)lane * 16u) = v;
        }
        *reinterpret_cast<u32x4*>(buf + (ggi == 0 ? SPEC_OFF_G0 : off_g1) + geff) = tl.grp;
      }
    };
    // loaders 0 and 1: the index program of tile w of the stage over this lane's 32 docs (linear layout), into the tile's buffer
    auto publish_lin = [&](uint8_t* stage, const SpecStage& st, int sidx) __attribute__((always_inline)) {
      if (w >= SPEC_TILES) return;   // wave-uniform (the other loaders loaded dwords too: their loads keep every loader's wait counts alike)
      const int i = sidx * SPEC_TILES + w;
      const int wt = (int)blockIdx.x + i * grid;
      const int64_t rem = i < n_mine ? (int64_t)p.num_docs - (int64_t)wt * PG_WAVE_DOCS : 0;   // (the second tile of an odd last stage: empty)
      const int32_t n_valid = rem >= PG_WAVE_DOCS ? PG_WAVE_DOCS : (rem > 0 ? (int32_t)rem : 0);
      uint32_t lin = valid_lin_mask(n_valid, lane);
      if (HAS_INDEX) {
        uint32_t grp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int gj = p.dense_group[j];
#pragma unroll
          for (int k = 0; k < 4; k++) grp[k] |= gj == k ? st.post[j] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k < p.dense_groups) lin &= ((p.dense_excl >> k) & 1) ? ~grp[k] : grp[k];
      }
      *reinterpret_cast<uint32_t*>(stage + (uint32_t)w * tile_bytes + SPEC_OFF_LIN + (uint32_t)lane * 4u) = HAS_TAIL ? lin & st.tail : lin;
      ld_cand += (uint32_t)__popc(lin);   // the scan leaf's candidates (numEntriesScannedInFilter), counted where the whole dword is at hand
    };
    // SPEC_SETS stages of loads in flight per loader (a register set each).  With two sets of two tiles 80 KB per CU were in flight against the
    // ~160 KB of pg_fast_i32range_p's eight wavefronts; the LDS buffers stay two: buffer = stage & 1.
    constexpr int SETS = HAS_SCAN ? SPEC_SETS : 2 * SPEC_SETS;   // (without a scan column a stage is half the bytes: twice the stages in flight)
    SpecStage st[SETS];
    auto buf_of = [&](int sg) __attribute__((always_inline)) { return stage0 + (uint32_t)(sg & 1) * stage_bytes; };
    if (n_stages > 0) {
#pragma unroll
      for (int k = 0; k < SETS; k++) issue(k, st[k]);
    }
    // Whole rounds of SPEC_SETS stages in the loop, the rest behind it: a conditional part INSIDE the loop gives the compiler a path on which a
    // younger stage's loads are older at the loop head, and it then waits for all of them.
    int s = 0;
    for (; s + SETS - 1 < n_stages; s += SETS) {
#pragma unroll
      for (int k = 0; k < SETS; k++) {
        publish(buf_of(s + k), st[k]);         // waits for stage s + k's loads only: the younger stages stay in flight
        publish_lin(buf_of(s + k), st[k], s + k);
        issue(s + k + SETS, st[k]);
        spec_barrier();                        // barrier s + k: its buffer holds the stage
      }
    }
#pragma unroll
    for (int k = 0; k < SETS - 1; k++)
      if (s + k < n_stages) {                  // workgroup-uniform
        publish(buf_of(s + k), st[k]);
        publish_lin(buf_of(s + k), st[k], s + k);
        spec_barrier();
      }
    const uint32_t csum = wave_sum_u32(ld_cand);
    if (HAS_SCAN && !p.fast_scan_pushed && lane == 0 && csum) atomicAdd(&s_stat[L.stat_slot], csum);   // (a pushed scan covers the segment: the host adds numDocs)
  } else {
    // ---- consumers: wavefront c aggregates quad row c (quads 64 c .. 64 c + 63) of every tile ------------------------------------------------
    const int c = wave - SPEC_LOADERS;
    const uint32_t R = (uint32_t)p.replicas;
    const uint32_t rep = (uint32_t)t & (R - 1u);
    const uint32_t stride = table_slots;                 // slots per accumulator (with the trash tail)
    const uint32_t trash_slot = real_slots + (uint32_t)lane;
    const uint32_t q = (uint32_t)c * 64u + (uint32_t)lane;   // this lane's quad of a tile: docs 4 q .. 4 q + 3
    const uint32_t nib = ((uint32_t)lane & 7u) * 4u;          // ... whose match bits are bits nib .. nib + 3 of linear dword q >> 3
    uint32_t my_matched = 0;
    // the accumulators as 2-bit codes in one 64-bit scalar (0 COUNT, 1 SUM, 2 MIN, 3 MAX; PG_MAX_OPS = 32): read from the kernel argument per
    // tile they were 46 scalar loads per tile, each a stall of the wavefront (profiles/r05_wave_specialised.txt)
    const int n_ops = uniform(p.n_ops);
    uint64_t ops_code = 0;
    for (int o = 0; o < n_ops; o++) {
      const PgAccOp op = p.ops[uniform(o)];
      ops_code |= (uint64_t)(op.src < 0 ? 0u : (op.fn == PG_ACC_SUM ? 1u : (op.fn == PG_ACC_MIN ? 2u : 3u))) << (2 * o);
    }
    ops_code = ((uint64_t)(uint32_t)uniform((int)(uint32_t)(ops_code >> 32)) << 32) | (uint64_t)(uint32_t)uniform((int)(uint32_t)ops_code);
    const int has_out_words = uniform(p.out_words != nullptr ? 1 : 0);
    auto fetch = [&](const uint8_t* buf, SpecQuad<NG>& d) __attribute__((always_inline)) {
      d.lin = *reinterpret_cast<const uint32_t*>(buf + SPEC_OFF_LIN + (q >> 3) * 4u);   // 8 lanes share a dword: a broadcast
      d.rng = HAS_SCAN ? *reinterpret_cast<const uint32_t*>(buf + SPEC_OFF_RNG + q * 4u) : 0xFu;
      d.val = *reinterpret_cast<const u32x4*>(buf + SPEC_OFF_VAL + q * 16u);
#pragma unroll
      for (int gi = 0; gi < NG; gi++) {
        const uint32_t di = __umul24(4u * q, (uint32_t)p.gcols[gi].bits) >> 5;
        const uint32_t* win = reinterpret_cast<const uint32_t*>(buf + (gi == 0 ? SPEC_OFF_G0 : off_g1)) + di;
        d.win[gi][0] = win[0]; d.win[gi][1] = win[1];
      }
    };
    // both quad rows of a stage (SPEC_TILES x 4 docs per lane) in one pass over the accumulators: two independent chains per wavefront
    auto aggregate = [&](const SpecQuad<NG> (&d)[SPEC_TILES], int sidx) __attribute__((always_inline)) {
      constexpr int ND = 4 * SPEC_TILES;
      uint32_t slot[ND];
      int32_t v[ND];
#pragma unroll
      for (int tt = 0; tt < SPEC_TILES; tt++) {
        const int i = sidx * SPEC_TILES + tt;
        const uint32_t cand = (d[tt].lin >> nib) & 0xFu;   // candidates of this lane's four docs (docs past the segment: none, the loader masked them)
        const uint32_t m = d[tt].rng & cand;
        my_matched += (uint32_t)__popc(m);
        if (has_out_words && i < n_mine) {   // the tile's match words, linear layout: lanes 8 g .. 8 g + 7 hold the nibbles of dword 8 c + g
          const int wt = (int)blockIdx.x + i * grid;
          const uint32_t word = or_reduce8(m << nib);
          if ((lane & 7) == 0) reinterpret_cast<uint32_t*>(p.out_words)[(int64_t)wt * 64 + (int64_t)(q >> 3)] = word;
        }
#ifndef PG_SPEC_NO_TABLE
#pragma unroll
        for (int e = 0; e < 4; e++) slot[4 * tt + e] = rep;
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
          const PgGroupCol& gc = p.gcols[gi];
          const uint32_t bits = (uint32_t)gc.bits, mask = (1u << gc.bits) - 1u;
          const uint32_t mult = (uint32_t)gc.mult * R;
          uint32_t dv[4];
          decode_packed_quad<true>(d[tt].win[gi], q, bits, mask, dv);
#pragma unroll
          for (int e = 0; e < 4; e++) slot[4 * tt + e] += __umul24(dv[e], mult);   // < 65536 slots (planner)
        }
        // Unpredicated LDS atomics: a doc that does not match aims at this lane's own trash slot (the 64 slots behind every accumulator's
        // row, never read) instead of being skipped — one select per doc in place of a compare, an exec-mask save / restore and a branch per
        // (doc, accumulator).  The consumers are bound by instruction issue and LDS round trips, not by LDS throughput (27 % busy): predicated,
        // this loop was 0.24 ms of 1.69 (profiles/r05_wave_specialised.txt).
#pragma unroll
        for (int e = 0; e < 4; e++) slot[4 * tt + e] = ((m >> e) & 1u) ? slot[4 * tt + e] : trash_slot;
        v[4 * tt + 0] = (int32_t)d[tt].val.x; v[4 * tt + 1] = (int32_t)d[tt].val.y; v[4 * tt + 2] = (int32_t)d[tt].val.z; v[4 * tt + 3] = (int32_t)d[tt].val.w;   // (the loaders swapped the bytes)
#endif
      }
#ifdef PG_SPEC_NO_TABLE   // measurement variant (wrong results): the filter alone, nothing aggregated
      return;
#endif
      for (int o = 0; o < n_ops; o++) {
        const uint32_t code = (uint32_t)(ops_code >> (2 * o)) & 3u;   // 0 COUNT, 1 SUM, 2 MIN, 3 MAX
        int64_t* base = lds_table + (size_t)o * stride;
        if (code == 0u) {
#pragma unroll
          for (int e = 0; e < ND; e++) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[e]), 1ULL);
        } else if (code == 1u) {
#pragma unroll
          for (int e = 0; e < ND; e++) atomicAdd(reinterpret_cast<unsigned long long*>(base + slot[e]), (unsigned long long)(int64_t)v[e]);
        } else if (code == 2u) {
#pragma unroll
          for (int e = 0; e < ND; e++) atomicMin(reinterpret_cast<long long*>(base + slot[e]), (long long)v[e]);
        } else {
#pragma unroll
          for (int e = 0; e < ND; e++) atomicMax(reinterpret_cast<long long*>(base + slot[e]), (long long)v[e]);
        }
      }
    };
    auto fetch_stage = [&](int sidx, SpecQuad<NG> (&d)[SPEC_TILES]) __attribute__((always_inline)) {
      const uint8_t* stage = stage0 + (uint32_t)(sidx & 1) * stage_bytes;
#pragma unroll
      for (int tt = 0; tt < SPEC_TILES; tt++) fetch(stage + (uint32_t)tt * tile_bytes, d[tt]);
    };
    auto aggregate_stage = [&](int sidx, const SpecQuad<NG> (&d)[SPEC_TILES]) __attribute__((always_inline)) {
#ifdef PG_SPEC_NO_CONSUME   // measurement variant (wrong results): the loaders' stream alone
      return;
#endif
#ifdef PG_SPEC_FETCH_ONLY   // measurement variant (wrong results): the consumers' LDS reads alone
      {
        uint32_t acc = 0;
#pragma unroll
        for (int tt = 0; tt < SPEC_TILES; tt++) { acc += d[tt].lin ^ d[tt].rng ^ d[tt].val.x ^ d[tt].val.w; for (int gi = 0; gi < NG; gi++) acc += d[tt].win[gi][0] ^ d[tt].win[gi][1]; }
        if (acc == 0x12345678u) my_matched++;
        return;
      }
#endif
      aggregate(d, sidx);
    };
    // A stage's slice is read out of LDS as soon as its barrier has been passed, and aggregated out of registers behind the NEXT barrier — the
    // LDS round trip of stage s + 1 travels during the aggregation of stage s, and the loaders may refill a buffer whose slices sit in registers.
    SpecQuad<NG> da[SPEC_TILES], db[SPEC_TILES];
    if (n_stages > 0) {
      spec_barrier();                          // barrier 0
#ifndef PG_SPEC_NO_CONSUME
      fetch_stage(0, da);
#endif
    }
    for (int s = 0; s < n_stages; s += 2) {
      if (s + 1 < n_stages) {                  // workgroup-uniform
        spec_barrier();                        // barrier s + 1 (every slice of stage s has been read: the release in front of it waits for them)
#ifndef PG_SPEC_NO_CONSUME
        fetch_stage(s + 1, db);
#endif
      }
      aggregate_stage(s, da);
      if (s + 1 < n_stages) {
        if (s + 2 < n_stages) {
          spec_barrier();                      // barrier s + 2
#ifndef PG_SPEC_NO_CONSUME
          fetch_stage(s + 2, da);
#endif
        }
        aggregate_stage(s + 1, db);
      }
    }
    const uint32_t wsum = wave_sum_u32(my_matched);
    if (lane == 0 && wsum) atomicAdd(&s_stat[0], wsum);
  }
  __syncthreads();
  // statistics and this workgroup's partial table [n_ops][n_groups], replicas folded (flush_workgroup's work over this kernel's row stride)
  if (t < PG_MAX_STATS && s_stat[t]) atomicAdd(&p.stats[t], (unsigned long long)s_stat[t]);
  {
    const int R = p.replicas, groups = p.n_groups;
    int64_t* out = p.partials + (int64_t)blockIdx.x * ((int64_t)p.n_ops * groups);
    for (int o = 0; o < p.n_ops; o++) {
      const int fn = p.ops[uniform(o)].fn;   // integer accumulators only (pipe_fit)
      for (int gq = t; gq < groups; gq += PG_BLOCK) {
        const int64_t* src = lds_table + (size_t)o * table_slots + (size_t)gq * R;
        int64_t acc = src[0];
        if (fn == PG_ACC_COUNT || fn == PG_ACC_SUM) { for (int r = 1; r < R; r++) acc += src[r]; }
        else if (fn == PG_ACC_MIN) { for (int r = 1; r < R; r++) acc = src[r] < acc ? src[r] : acc; }
        else { for (int r = 1; r < R; r++) acc = src[r] > acc ? src[r] : acc; }
        out[(size_t)o * groups + gq] = acc;
      }
    }
  }
}
#define PG_SPEC_KERNEL(NAME, IDX, SCAN) \
  extern "C" __global__ void __launch_bounds__(PG_BLOCK) NAME(const PgQueryPlan p) { \
    if (p.n_group_cols == 1) spec_body<1, IDX, SCAN>(p); \
    else spec_body<2, IDX, SCAN>(p); \
  }
PG_SPEC_KERNEL(pg_fast_i32range_s, true, true)   // the headline shape: dense index program AND range scan
PG_SPEC_KERNEL(pg_spec_none, false, false)       // no filter
PG_SPEC_KERNEL(pg_spec_scan, false, true)        // the range scan is the whole filter
PG_SPEC_KERNEL(pg_spec_index, true, false)       // inverted-index leaves only
#undef PG_SPEC_KERNEL
extern "C" __global__ void __launch_bounds__(PG_BLOCK) pg_fast_i32range_st(const PgQueryPlan p) {   // the headline shape behind an upsert snapshot
  if (p.n_group_cols == 1) spec_body<1, true, true, true>(p);
  else spec_body<2, true, true, true>(p);
}
