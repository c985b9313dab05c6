/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 *
 * Bit-sliced range index: BitSlicedRangeIndexReader (pinot-segment-local/.../index/readers/BitSlicedRangeIndexReader.java:41-246)
 * over a RoaringBitmap `RangeBitmap` (third-party, RoaringBitmap 1.3.0 — NOT in the reference tree; format restated from the
 * published source, parity of the byte format unpinned by any reference fixture):
 *   Pinot header (big-endian, BitSlicedRangeIndexCreator.java:125-133):  int version = 2, long min
 *   RangeBitmap (little-endian):  u16 cookie 0xF00D, u8 base = 2, u8 sliceCount, u16 maxKey, u32 maxRid,
 *                                 maxKey x mask[(sliceCount + 7) / 8]   (bit i: slice i has a container in this 2^16-row chunk)
 *                                 then per chunk, per present slice:  u8 type (0 bitmap, 1 run, 2 array),
 *                                     bitmap: 8192 bytes;  run: u16 nRuns, nRuns x (u16 start, u16 length - 1);  array: u16 n, n x u16
 *   slice i holds the rows whose value has bit i CLEAR (the appender adds row to slice i for every set bit of ~value & mask).
 * Values: dictIds for dictionary columns (min 0), value - min for raw INT / LONG, FPOrdering.ordinalOf for FLOAT / DOUBLE
 * (pinot-segment-local/.../utils/FPOrdering.java).
 *
 * The oracle answers a query by DECODING every row's value from the slices and comparing — not by the bit-sliced lte algebra
 * the GPU leaf runs — so that the two are independent readings of the same bytes.
 */
#include <math.h>

#include "po_internal.h"

static uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

uint64_t po_fp_ordinal_double(double v) {   /* FPOrdering.ordinalOf(double) */
  if (v == (double)INFINITY) return 0xFFFFFFFFFFFFFFFFULL;
  if (v == -(double)INFINITY || v != v) return 0;
  uint64_t bits;
  memcpy(&bits, &v, 8);
  if (bits & 0x8000000000000000ULL) bits = bits == 0x8000000000000000ULL ? 0x8000000000000000ULL : ~bits;
  else bits ^= 0x8000000000000000ULL;
  return bits;
}
uint64_t po_fp_ordinal_float(float v) {     /* FPOrdering.ordinalOf(float) */
  if (v == INFINITY) return 0xFFFFFFFFULL;
  if (v == -INFINITY || v != v) return 0;
  uint32_t bits;
  memcpy(&bits, &v, 4);
  if (bits & 0x80000000u) bits = bits == 0x80000000u ? 0x80000000u : ~bits;
  else bits ^= 0x80000000u;
  return bits;
}

/* decodes every row's stored value; returns NULL (error set) on a malformed index */
static uint64_t* decode_values(const po_column* c, int32_t num_docs, int64_t* out_min) {
  const uint8_t* p = c->range_idx;
  const uint64_t len = c->range_len;
  if (len < 22 || (int32_t)po_be32(p) != 2) { po_set_error("range index of %s: bad header / version", c->name); return NULL; }
  *out_min = (int64_t)po_be64(p + 4);
  const uint8_t* r = p + 12;
  if (le16(r) != 0xF00D || r[2] != 2) { po_set_error("range index of %s: bad RangeBitmap cookie / base", c->name); return NULL; }
  const int slice_count = r[3];
  const uint32_t max_key = le16(r + 4), max_rid = le32(r + 6);
  const int bytes_per_mask = (slice_count + 7) >> 3;
  if (slice_count < 1 || slice_count > 64 || (int64_t)max_rid < num_docs) { po_set_error("range index of %s: %d slices, %u rows", c->name, slice_count, max_rid); return NULL; }
  const uint64_t range_mask = slice_count == 64 ? ~0ULL : ((1ULL << slice_count) - 1ULL);
  uint64_t pos = 12 + 10 + (uint64_t)max_key * (uint64_t)bytes_per_mask;
  if (pos > len) { po_set_error("range index of %s: truncated masks", c->name); return NULL; }
  uint64_t* values = (uint64_t*)po_xmalloc(sizeof(uint64_t) * (size_t)(num_docs > 0 ? num_docs : 1));
  for (int32_t i = 0; i < num_docs; i++) values[i] = range_mask;   /* a row absent from slice i has bit i set */
  for (uint32_t key = 0; key < max_key; key++) {
    uint64_t mask = 0;
    for (int b = 0; b < bytes_per_mask; b++) mask |= (uint64_t)p[22 + (uint64_t)key * bytes_per_mask + b] << (8 * b);
    for (int s = 0; s < slice_count; s++) {
      if (!((mask >> s) & 1)) continue;
      if (pos + 1 > len) goto truncated;
      const int type = p[pos++];
      const int64_t base = (int64_t)key << 16;
      if (type == 0) {
        if (pos + 8192 > len) goto truncated;
        for (int w = 0; w < 1024; w++) {
          uint64_t word = 0;
          for (int b = 0; b < 8; b++) word |= (uint64_t)p[pos + 8 * w + b] << (8 * b);
          while (word) {
            const int bit = __builtin_ctzll(word);
            word &= word - 1;
            const int64_t row = base + 64 * w + bit;
            if (row < num_docs) values[row] &= ~(1ULL << s);
          }
        }
        pos += 8192;
      } else if (type == 1) {
        if (pos + 2 > len) goto truncated;
        const uint32_t n = le16(p + pos);
        pos += 2;
        if (pos + 4ULL * n > len) goto truncated;
        for (uint32_t k = 0; k < n; k++) {
          const int64_t st = le16(p + pos + 4 * k), run = le16(p + pos + 4 * k + 2);
          for (int64_t row = base + st; row <= base + st + run; row++) if (row < num_docs) values[row] &= ~(1ULL << s);
        }
        pos += 4ULL * n;
      } else if (type == 2) {
        if (pos + 2 > len) goto truncated;
        const uint32_t n = le16(p + pos);
        pos += 2;
        if (pos + 2ULL * n > len) goto truncated;
        for (uint32_t k = 0; k < n; k++) {
          const int64_t row = base + le16(p + pos + 2 * k);
          if (row < num_docs) values[row] &= ~(1ULL << s);
        }
        pos += 2ULL * n;
      } else {
        po_set_error("range index of %s: container type %d", c->name, type);
        free(values);
        return NULL;
      }
    }
  }
  return values;
truncated:
  po_set_error("range index of %s: truncated container", c->name);
  free(values);
  return NULL;
}

/* RangeIndexBasedFilterOperator#getMatchingDocIds (core/operator/filter/RangeIndexBasedFilterOperator.java:112-145) over
 * BitSlicedRangeIndexReader#getMatchingDocIds: the rows whose value lies in the predicate's inclusive bounds */
po_bitmap* po_range_index_matching(const po_column* c, const po_pred_eval* e, int32_t num_docs) {
  int64_t min = 0;
  uint64_t* values = decode_values(c, num_docs, &min);
  if (!values) return NULL;
  po_bitmap* out = po_bitmap_new(num_docs);
  const int eq = e->pred_type == PG_PRED_EQ;
  if (e->dictionary_based) {   /* dictIds: IntRange [start, end - 1] / IntValue */
    const int64_t lo = eq ? e->matching_dict_ids[0] : e->start_dict_id, hi = eq ? e->matching_dict_ids[0] : (int64_t)e->end_dict_id - 1;
    for (int32_t d = 0; d < num_docs; d++) if ((int64_t)values[d] >= lo && (int64_t)values[d] <= hi) po_bitmap_add(out, d);
  } else if (c->data_type == PG_TYPE_INT || c->data_type == PG_TYPE_LONG) {
    const int64_t lo = eq ? e->raw_i[0] : e->lo_i, hi = eq ? e->raw_i[0] : e->hi_i;
    for (int32_t d = 0; d < num_docs; d++) {
      const int64_t v = (int64_t)(values[d] + (uint64_t)min);
      if (v >= lo && v <= hi) po_bitmap_add(out, d);
    }
  } else if (c->data_type == PG_TYPE_FLOAT) {
    const float flo = eq ? (float)e->raw_d[0] : e->lo_f, fhi = eq ? (float)e->raw_d[0] : e->hi_f;
    if (!(flo > fhi)) {
      const uint64_t lo = po_fp_ordinal_float(flo), hi = po_fp_ordinal_float(fhi);
      for (int32_t d = 0; d < num_docs; d++) if (values[d] >= lo && values[d] <= hi) po_bitmap_add(out, d);
    }
  } else {
    const double dlo = eq ? e->raw_d[0] : e->lo_d, dhi = eq ? e->raw_d[0] : e->hi_d;
    if (!(dlo > dhi)) {
      const uint64_t lo = po_fp_ordinal_double(dlo), hi = po_fp_ordinal_double(dhi);
      for (int32_t d = 0; d < num_docs; d++) if (values[d] >= lo && values[d] <= hi) po_bitmap_add(out, d);
    }
  }
  free(values);
  return out;
}
