// Exact floating SUMs in fixed point (shared by the kernels, the planner and a CPU test harness; plain C++, no HIP types).
//
// The reference adds every value to a double in docId order (SumAggregationFunction.java:160-179); any other order of floating
// additions rounds differently.  Instead of chasing that order the GPU path keeps SUMs EXACT: a value x of a column whose finite
// magnitudes stay below 2^E is the integer X = trunc(|x| * 2^-q) with q = E - 32 L + 1, cut into L base-2^32 digits; digit j of every
// value goes into int64 accumulator ("limb") j.  |digit| < 2^32 and a segment has < 2^31 docs, so no limb overflows; integer
// additions commute, so workgroup count, atomics order and merge order (also across GPUs) cannot change the result.  The host
// combines the limbs — sum_j limb_j * 2^(32 j + q) — and rounds to double ONCE (nearest-even).  FLOAT columns use L = 3 (96 bits),
// DOUBLE columns L = 4 (128 bits): values within 2^-56 / 2^-59 of the largest magnitude are represented exactly, smaller ones are
// truncated below 2^q, an absolute error < docs * 2^q <= 2^-64 * 2^E — below one ulp of any sum that does not cancel to < 2^-11 of
// the largest magnitude.  LONG columns whose sum could leave int64 use two digits of the value itself (exact).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PG_FX_HD __host__ __device__ __forceinline__
#else
#define PG_FX_HD static inline
#endif

// digit `limb` (base 2^32) of trunc(|x| * 2^-q), carrying x's sign; x finite, |x| * 2^-q < 2^(32 L - 1)
PG_FX_HD int64_t pg_fx_digit(double x, int q, int limb) {
  uint64_t b;
  __builtin_memcpy(&b, &x, 8);
  const int e = (int)((b >> 52) & 0x7FFu);
  uint64_t m = b & 0xFFFFFFFFFFFFFULL;
  int ex = -1074;                       // value = m * 2^ex with a 53-bit m
  if (e) { m |= 1ULL << 52; ex = e - 1075; }
  const int sh = ex - q - 32 * limb;    // digit = floor(m * 2^sh) mod 2^32
  uint32_t d;
  if (sh >= 0) d = sh >= 32 ? 0u : ((uint32_t)m << sh);
  else d = sh <= -64 ? 0u : (uint32_t)(m >> (-sh));
  const int64_t v = (int64_t)d;
  return (b >> 63) ? -v : v;
}
// digit of a LONG value: 0 = low 32 bits (unsigned), 1 = high 32 bits (signed)
PG_FX_HD int64_t pg_long_digit(int64_t v, int limb) { return limb ? (v >> 32) : (int64_t)(uint32_t)v; }

#include <math.h>
// sum_j limbs[j] * 2^(32 j + q) rounded to nearest-even once: two's-complement base-2^32 digits -> magnitude -> top 53 bits + round + sticky
static inline double pg_limbs_to_double(const int64_t* limbs, int n_limbs, int q) {
  uint32_t dig[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // n_limbs <= 4 digits + carries + sign extension
  __int128 carry = 0;
  for (int j = 0; j < 7; j++) {
    const __int128 t = carry + (j < n_limbs ? (__int128)limbs[j] : 0);
    dig[j] = (uint32_t)(t & 0xFFFFFFFF);
    carry = t >> 32;   // arithmetic shift: floor
  }
  dig[7] = (uint32_t)(carry & 0xFFFFFFFF);
  const int neg = (dig[7] >> 31) != 0;
  if (neg) {
    uint64_t c = 1;
    for (int j = 0; j < 8; j++) { const uint64_t t = (uint64_t)(uint32_t)~dig[j] + c; dig[j] = (uint32_t)t; c = t >> 32; }
  }
  int top = -1;
  for (int j = 7; j >= 0; j--) if (dig[j]) { top = 32 * j + 31 - __builtin_clz(dig[j]); break; }
  if (top < 0) return 0.0;
#define PG_FX_BIT(i) ((i) < 0 ? 0ULL : (uint64_t)((dig[(i) >> 5] >> ((i) & 31)) & 1u))
  uint64_t mant = 0;
  for (int i = top; i > top - 53; i--) mant = (mant << 1) | PG_FX_BIT(i);
  if (top >= 53) {
    const uint64_t round = PG_FX_BIT(top - 53);
    int sticky = 0;
    for (int i = top - 54; i >= 0 && !sticky; i--) sticky = PG_FX_BIT(i) != 0;
    if (round && (sticky || (mant & 1))) mant++;
  }
#undef PG_FX_BIT
  const double v = ldexp((double)mant, top - 52 + q);
  return neg ? -v : v;
}
