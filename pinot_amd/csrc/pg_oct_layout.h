// Oct layout (shared by pg_kernels_oct.hip and the oct-layout scatter of pg_kernels_part.hip): lane L owns the 8 CONSECUTIVE docs
// 8L .. 8L+7 of a 512-doc sub-tile, so the 8 values of a b-bit column are exactly b bytes at byte offset b x L — one byte permute
// (v_perm_b32: alignment + endianness in one instruction, its selector a per-lane constant) per dword and compile-time field positions
// after it (1.5-4 VALU per value; the quad layout's run-time-width windows cost 5-9).  Included after pg_kernels.hip.
#pragma once

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

#define OCT_SUB_DOCS 512          // docs per sub-tile: 64 lanes x 8
#define OCT_SUBS_PER_WTILE 4
#define OCT_STREAM_BLOCK 256      // entries a wavefront claims per global atomic and writes with one 16-byte store per lane

// byte permute: result byte i = byte sel[i] of the 8 bytes {hi (4..7), lo (0..3)}
DEVFN uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
DEVFN uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }
DEVFN uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
DEVFN uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
DEVFN uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }

// selector that turns the little-endian dword pair {w[k+1], w[k]} into the big-endian dword starting at byte `bs` of w[k]
DEVFN uint32_t oct_selector(uint32_t bs) { return (bs << 24) | ((bs + 1u) << 16) | ((bs + 2u) << 8) | (bs + 3u); }

// The 8 values of a B-bit column whose lane window starts at the most significant bit of n[0] (big-endian dwords)
template <int B, int ND>
DEVFN void oct_fields(const uint32_t (&n)[ND], uint32_t (&out)[8]) {
  constexpr uint32_t mask = B >= 32 ? 0xFFFFFFFFu : ((1u << B) - 1u);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int p = i * B, k = p >> 5, o = p & 31;
    if (o + B <= 32) out[i] = bfe(n[k], (uint32_t)(32 - o - B), (uint32_t)B);
    else out[i] = alignbit(n[k], n[k + 1 < ND ? k + 1 : k], (uint32_t)(64 - o - B)) & mask;
  }
}
// group column (<= 8 bits): raw = the 3 dwords from the lane's dword-aligned window start
template <int B>
DEVFN void oct_decode_small(const u32x3 raw, uint32_t sel, uint32_t (&out)[8]) {
  if (B <= 4) {
    uint32_t n[1] = {perm(raw.y, raw.x, sel)};
    oct_fields<B, 1>(n, out);
  } else {
    uint32_t n[2] = {perm(raw.y, raw.x, sel), perm(raw.z, raw.y, sel)};
    oct_fields<B, 2>(n, out);
  }
}
DEVFN void oct_decode_group(int bits, const u32x3 raw, uint32_t sel, uint32_t (&out)[8]) {
  switch (bits) {   // wave-uniform
    case 1: oct_decode_small<1>(raw, sel, out); break;
    case 2: oct_decode_small<2>(raw, sel, out); break;
    case 3: oct_decode_small<3>(raw, sel, out); break;
    case 4: oct_decode_small<4>(raw, sel, out); break;
    case 5: oct_decode_small<5>(raw, sel, out); break;
    case 6: oct_decode_small<6>(raw, sel, out); break;
    case 7: oct_decode_small<7>(raw, sel, out); break;
    default: oct_decode_small<8>(raw, sel, out); break;
  }
}
// source column (<= 24 bits): raw = 8 dwords from the lane's dword-aligned window start (8 x 24 bits + 3 bytes of misalignment = 27 bytes)
template <int B>
DEVFN void oct_decode_wide(const u32x4 a, const u32x4 b, uint32_t sel, uint32_t (&out)[8]) {
  constexpr int ND = (8 * B + 31) / 32;   // big-endian dwords the 8 fields span
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t n[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) n[k] = perm(w[k + 1 < 8 ? k + 1 : k], w[k], sel);
  oct_fields<B, ND>(n, out);
}
DEVFN void oct_decode_source(int bits, const u32x4 a, const u32x4 b, uint32_t sel, uint32_t (&out)[8]) {
  switch (bits) {   // wave-uniform
#define OCT_CASE(B) case B: oct_decode_wide<B>(a, b, sel, out); break;
    OCT_CASE(1) OCT_CASE(2) OCT_CASE(3) OCT_CASE(4) OCT_CASE(5) OCT_CASE(6) OCT_CASE(7) OCT_CASE(8) OCT_CASE(9) OCT_CASE(10) OCT_CASE(11)
    OCT_CASE(12) OCT_CASE(13) OCT_CASE(14) OCT_CASE(15) OCT_CASE(16) OCT_CASE(17) OCT_CASE(18) OCT_CASE(19) OCT_CASE(20) OCT_CASE(21)
    OCT_CASE(22) OCT_CASE(23)
#undef OCT_CASE
    default: oct_decode_wide<24>(a, b, sel, out); break;
  }
}

// source kinds (PgQueryPlan::oct_src_kind)
enum { OCT_SRC_NONE = 0, OCT_SRC_AFFINE = 1, OCT_SRC_LUT = 2, OCT_SRC_RAW32 = 3, OCT_SRC_DICTID = 4 };

// tail of stream-lib MurmurHash.hashLong for an INT value whose first product k0 = (uint32) v x m is given; sign = v < 0
DEVFN uint32_t oct_murmur_tail(uint32_t k0, uint32_t c_sign) {
  constexpr uint32_t m = 0x5bd1e995u, m2 = m * m;
  uint32_t k = k0 ^ (k0 >> 24);
  uint32_t h = k * m2;
  h ^= c_sign;
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return h;
}
DEVFN constexpr uint32_t oct_hi_neg() {   // contribution of a high word of all ones: ((0xFFFFFFFF x m) ^ (… >>> 24)) x m
  constexpr uint32_t m = 0x5bd1e995u, km = 0u - m;
  return (km ^ (km >> 24)) * m;
}

struct OctRaw {
  u32x3 g[4];     // group columns: 3 dwords from the lane's window start
  u32x4 s0, s1;   // the source: 8 dwords from its window start, or the 8 raw 32-bit values
  uint32_t mw;    // match word holding the lane's 8 mask bits (MASKED)
};

struct OctLane {   // per-lane constants of the columns
  uint32_t goff[4], gsel[4];   // byte offset (dword aligned) of the lane's window inside a sub-tile, permute selector
  uint32_t soff, ssel;
};

template <bool MASKED>
DEVFN void oct_issue(const PgQueryPlan& p, const OctLane& ln, int wt, int sub, int lane, OctRaw& raw) {
#pragma unroll
  for (int g = 0; g < 4; g++)
    if (g < p.n_group_cols) {
      const PgGroupCol& gc = p.gcols[g];
      const GAS uint8_t* base = gptr<uint8_t>(gc.data) + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)gc.bits + (size_t)sub * (size_t)(OCT_SUB_DOCS / 8) * (size_t)gc.bits;
      raw.g[g] = ldnt((const GAS u32x3_a4*)(base + ln.goff[g]));
    }
  if (p.oct_src_kind != OCT_SRC_NONE) {
    const PgValueSrc& V = p.srcs[p.oct_src];
    const uint32_t bits = p.oct_src_kind == OCT_SRC_RAW32 ? 32u : (uint32_t)V.bits;
    const GAS uint8_t* base = gptr<uint8_t>(V.data) + (size_t)wt * (size_t)(PG_WAVE_DOCS / 8) * (size_t)bits + (size_t)sub * (size_t)(OCT_SUB_DOCS / 8) * (size_t)bits;
    const GAS u32x4_a4* q = (const GAS u32x4_a4*)(base + ln.soff);
    raw.s0 = ldnt(q);
    raw.s1 = ldnt(q + 1);
  }
  if (MASKED) raw.mw = gptr<uint32_t>(p.match_words)[(size_t)wt * 64 + (size_t)sub * 16 + (size_t)(lane >> 2)];
}

// keys and raw source items (dictIds / raw values) of the lane's 8 docs: everything that reads the load buffer
DEVFN void oct_decode(const PgQueryPlan& p, const OctLane& ln, const OctRaw& raw, uint32_t (&key)[8], uint32_t (&item)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) key[j] = 0;
#pragma unroll
  for (int g = 0; g < 4; g++)
    if (g < p.n_group_cols) {
      uint32_t v[8];
      oct_decode_group(p.gcols[g].bits, raw.g[g], ln.gsel[g], v);
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) key[j] = v[j];   // mult of column 0 is 1
      } else {
        const uint32_t mult = (uint32_t)p.gcols[g].mult;
#pragma unroll
        for (int j = 0; j < 8; j++) key[j] = mad24(v[j], mult, key[j]);   // dictId < 2^8, mult < 2^24 (planner)
      }
    }
  const int kind = p.oct_src_kind;
  if (kind == OCT_SRC_NONE) return;
  if (kind == OCT_SRC_RAW32) {
    const uint32_t w[8] = {raw.s0.x, raw.s0.y, raw.s0.z, raw.s0.w, raw.s1.x, raw.s1.y, raw.s1.z, raw.s1.w};
#pragma unroll
    for (int j = 0; j < 8; j++) item[j] = bswap32(w[j]);
  } else {
    oct_decode_source(p.srcs[p.oct_src].bits, raw.s0, raw.s1, ln.ssel, item);
  }
}
// item[] in: the docs' dictIds / raw values (oct_decode); out: what the back end applies — HyperLogLog index | rank << log2m, or the dictId.
// Nothing here touches a load buffer: the caller re-requests the buffer between oct_decode and this.
DEVFN void oct_finish(const PgQueryPlan& p, const uint32_t (&key)[8], uint32_t (&item)[8]) {
  (void)key;
  const int kind = p.oct_src_kind;
  if (kind == OCT_SRC_NONE || kind == OCT_SRC_DICTID) return;
  uint32_t id[8];
#pragma unroll
  for (int j = 0; j < 8; j++) id[j] = item[j];
  const uint32_t log2m = (uint32_t)p.oct_log2m;
  if (kind == OCT_SRC_LUT) {   // any dictionary: (index | rank << 16) per dictId, computed on the host at plan time; gathers first
    uint32_t ir[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ir[j] = gptr<uint32_t>(p.oct_lut)[id[j]];
#pragma unroll
    for (int j = 0; j < 8; j++) item[j] = (ir[j] & 0xFFFFu) | ((ir[j] >> 16) << log2m);
    return;
  }
  uint32_t h[8];
  if (kind == OCT_SRC_AFFINE) {
    const uint32_t c0 = p.oct_c0, c1lo = p.oct_c1 & 0xFFFFFFu, c1hi = p.oct_c1 >> 24;
    if (p.oct_nonneg) {
#pragma unroll
      for (int j = 0; j < 8; j++) h[j] = oct_murmur_tail(mad24(id[j], c1lo, c0) + (mul24(id[j], c1hi) << 24), 0u);
    } else {
      const uint32_t base = (uint32_t)p.oct_base, step = (uint32_t)p.oct_step;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int32_t v = (int32_t)mad24(id[j], step, base);
        h[j] = oct_murmur_tail(mad24(id[j], c1lo, c0) + (mul24(id[j], c1hi) << 24), (uint32_t)(v >> 31) & oct_hi_neg());
      }
    }
  } else {   // raw INT values: hll.offer(Integer) = hashLong((long) v)
#pragma unroll
    for (int j = 0; j < 8; j++) h[j] = oct_murmur_tail(id[j] * 0x5bd1e995u, (uint32_t)((int32_t)id[j] >> 31) & oct_hi_neg());
  }
  const uint32_t tail = (1u << (log2m - 1u)) + 1u;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t idx = h[j] >> (32u - log2m);
    const uint32_t w = (h[j] << log2m) | tail;
    item[j] = idx | ((uint32_t)(__clz((int)w) + 1) << log2m);
  }
}

DEVFN uint32_t oct_mask8(const PgQueryPlan& p, int wt, int sub, int lane, uint32_t mw, bool masked) {
  const int64_t first = (int64_t)wt * PG_WAVE_DOCS + (int64_t)sub * OCT_SUB_DOCS + 8 * lane;
  const int64_t rem = (int64_t)p.num_docs - first;
  uint32_t m = rem >= 8 ? 0xFFu : (rem <= 0 ? 0u : ((1u << (uint32_t)rem) - 1u));
  if (masked) m &= mw >> (8u * ((uint32_t)lane & 3u));
  return m;
}

DEVFN void oct_lane_setup(const PgQueryPlan& p, int lane, OctLane& ln) {
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
  ln.soff = 0;
  ln.ssel = oct_selector(0);
  if (p.oct_src_kind != OCT_SRC_NONE) {
    const uint32_t bits = p.oct_src_kind == OCT_SRC_RAW32 ? 32u : (uint32_t)p.srcs[p.oct_src].bits;
    const uint32_t bo = (uint32_t)lane * bits;
    ln.soff = bo & ~3u;
    ln.ssel = oct_selector(bo & 3u);
  }
}

