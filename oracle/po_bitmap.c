/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 *
 * Dense-bitset stand-in for org.roaringbitmap.{Immutable,Mutable}RoaringBitmap.  RoaringBitmap 1.3.0 is a Maven
 * dependency (pom.xml:798-802) whose source is not under /root/reference; the reference only uses its *set algebra*
 * (or / and / flip / cardinality / ordered iteration — InvertedIndexFilterOperator.java:60-96, AndDocIdSet.java:127-186),
 * which any correct bitset reproduces bit-exactly, and its *portable serialization*, restated here from the public
 * format specification (https://github.com/RoaringBitmap/RoaringFormatSpec):
 *   cookie 12346: [u32 cookie][u32 size][size x (u16 key, u16 card-1)][size x u32 offset][containers]
 *   cookie 12347: [u32 cookie | (size-1)<<16][ceil(size/8) run-flag bytes][size x (u16 key, u16 card-1)]
 *                 [size x u32 offset, only if size >= 4][containers]
 *   container: run flag set -> [u16 n_runs][n_runs x (u16 start, u16 length-1)]
 *              card > 4096  -> 1024 x u64 little-endian words
 *              else         -> card x u16 sorted values
 * "parity unpinned" for the byte format by the reference's own tests (they only round-trip through the library);
 * it is pinned here indirectly by the inverted-index goldens (any mis-parse changes filtered counts).
 */
#include "po_internal.h"

po_bitmap* po_bitmap_new(int64_t universe) {
  po_bitmap* b = (po_bitmap*)po_xcalloc(1, sizeof(po_bitmap));
  b->universe = universe;
  /* round up to whole 65536-bit containers so deserialization never needs bounds checks per word */
  int64_t containers = (universe + 65535) / 65536;
  if (containers < 1) containers = 1;
  b->n_words = containers * 1024;
  b->words = (uint64_t*)po_xcalloc((size_t)b->n_words, 8);
  return b;
}

/* a set over a small universe (the dictIds one group has seen): whole 64-bit words, no container rounding */
po_bitmap* po_bitmap_new_small(int64_t universe) {
  po_bitmap* b = (po_bitmap*)po_xcalloc(1, sizeof(po_bitmap));
  b->universe = universe;
  b->n_words = (universe + 63) / 64;
  if (b->n_words < 1) b->n_words = 1;
  if (universe > 4096) {
    b->cap_sparse = 8;
    b->sparse = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)b->cap_sparse);
    return b;
  }
  b->words = (uint64_t*)po_xcalloc((size_t)b->n_words, 8);
  return b;
}

static void densify(po_bitmap* b) {
  if (!b->sparse) return;
  b->words = (uint64_t*)po_xcalloc((size_t)b->n_words, 8);
  for (int32_t i = 0; i < b->n_sparse; i++) b->words[b->sparse[i] >> 6] |= 1ULL << (b->sparse[i] & 63);
  free(b->sparse);
  b->sparse = NULL;
  b->n_sparse = b->cap_sparse = 0;
}

po_bitmap* po_bitmap_clone(const po_bitmap* s) {
  po_bitmap* b = (po_bitmap*)po_xmalloc(sizeof(po_bitmap));
  *b = *s;
  if (s->sparse) {
    b->sparse = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)s->cap_sparse);
    memcpy(b->sparse, s->sparse, sizeof(int32_t) * (size_t)s->n_sparse);
    return b;
  }
  b->words = (uint64_t*)po_xmalloc((size_t)s->n_words * 8);
  memcpy(b->words, s->words, (size_t)s->n_words * 8);
  return b;
}

void po_bitmap_free(po_bitmap* b) {
  if (!b) return;
  free(b->words);
  free(b->sparse);
  free(b);
}

void po_bitmap_add(po_bitmap* b, int32_t x) {
  if (b->sparse) {
    int32_t lo = 0, hi = b->n_sparse;
    while (lo < hi) {
      int32_t mid = (lo + hi) >> 1;
      if (b->sparse[mid] < x) lo = mid + 1;
      else hi = mid;
    }
    if (lo < b->n_sparse && b->sparse[lo] == x) return;
    if ((int64_t)b->n_sparse + 1 > b->n_words) { densify(b); b->words[x >> 6] |= 1ULL << (x & 63); return; }
    if (b->n_sparse == b->cap_sparse) {
      b->cap_sparse *= 2;
      b->sparse = (int32_t*)po_xrealloc(b->sparse, sizeof(int32_t) * (size_t)b->cap_sparse);
    }
    memmove(b->sparse + lo + 1, b->sparse + lo, sizeof(int32_t) * (size_t)(b->n_sparse - lo));
    b->sparse[lo] = x;
    b->n_sparse++;
    return;
  }
  b->words[x >> 6] |= 1ULL << (x & 63);
}

void po_bitmap_add_range(po_bitmap* b, int64_t start, int64_t end) {
  if (start >= end) return;
  densify(b);
  int64_t fw = start >> 6, lw = (end - 1) >> 6;
  uint64_t fm = ~0ULL << (start & 63);
  uint64_t lm = ~0ULL >> (63 - ((end - 1) & 63));
  if (fw == lw) {
    b->words[fw] |= fm & lm;
    return;
  }
  b->words[fw] |= fm;
  for (int64_t w = fw + 1; w < lw; w++) b->words[w] = ~0ULL;
  b->words[lw] |= lm;
}

void po_bitmap_or(po_bitmap* d, const po_bitmap* s) {
  int64_t n = d->n_words < s->n_words ? d->n_words : s->n_words;
  for (int64_t i = 0; i < n; i++) d->words[i] |= s->words[i];
}

void po_bitmap_and(po_bitmap* d, const po_bitmap* s) {
  int64_t n = d->n_words < s->n_words ? d->n_words : s->n_words;
  for (int64_t i = 0; i < n; i++) d->words[i] &= s->words[i];
  for (int64_t i = n; i < d->n_words; i++) d->words[i] = 0;
}

void po_bitmap_flip(po_bitmap* b, int64_t start, int64_t end) {
  if (start >= end) return;
  int64_t fw = start >> 6, lw = (end - 1) >> 6;
  uint64_t fm = ~0ULL << (start & 63);
  uint64_t lm = ~0ULL >> (63 - ((end - 1) & 63));
  if (fw == lw) {
    b->words[fw] ^= fm & lm;
    return;
  }
  b->words[fw] ^= fm;
  for (int64_t w = fw + 1; w < lw; w++) b->words[w] = ~b->words[w];
  b->words[lw] ^= lm;
}

int64_t po_bitmap_cardinality(const po_bitmap* b) {
  if (b->sparse) return b->n_sparse;
  int64_t c = 0;
  for (int64_t i = 0; i < b->n_words; i++) c += __builtin_popcountll(b->words[i]);
  return c;
}

int po_bitmap_contains(const po_bitmap* b, int32_t x) {
  if (x < 0 || (x >> 6) >= b->n_words) return 0;
  return (int)((b->words[x >> 6] >> (x & 63)) & 1);
}

int64_t po_bitmap_next_set(const po_bitmap* b, int64_t from) {
  if (from < 0) from = 0;
  if (b->sparse) {
    int32_t lo = 0, hi = b->n_sparse;
    while (lo < hi) {
      int32_t mid = (lo + hi) >> 1;
      if (b->sparse[mid] < from) lo = mid + 1;
      else hi = mid;
    }
    return lo < b->n_sparse ? b->sparse[lo] : -1;
  }
  int64_t w = from >> 6;
  if (w >= b->n_words) return -1;
  uint64_t cur = b->words[w] & (~0ULL << (from & 63));
  while (1) {
    if (cur) return (w << 6) + __builtin_ctzll(cur);
    if (++w >= b->n_words) return -1;
    cur = b->words[w];
  }
}

static inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

static int roaring_walk(const uint8_t* blob, uint64_t len, po_bitmap* dst, int* n_array, int* n_bitmap, int* n_run) {
  if (len < 8) {
    po_set_error("roaring: blob too short (%llu)", (unsigned long long)len);
    return -1;
  }
  uint32_t cookie = le32(blob);
  uint64_t pos = 4;
  uint32_t size;
  const uint8_t* run_flags = NULL;
  int has_run = 0;
  if ((cookie & 0xFFFF) == 12347) {
    size = (cookie >> 16) + 1;
    run_flags = blob + pos;
    pos += (size + 7) / 8;
    has_run = 1;
  } else if (cookie == 12346) {
    size = le32(blob + pos);
    pos += 4;
  } else {
    po_set_error("roaring: bad cookie %u", cookie);
    return -1;
  }
  if (size > 65536 || pos + 4ULL * size > len) {
    po_set_error("roaring: bad size %u", size);
    return -1;
  }
  const uint8_t* desc = blob + pos;
  pos += 4ULL * size;
  if (!has_run || size >= 4) pos += 4ULL * size; /* offset header (not needed for a sequential walk) */
  for (uint32_t i = 0; i < size; i++) {
    uint32_t key = le16(desc + 4 * i);
    uint32_t card = (uint32_t)le16(desc + 4 * i + 2) + 1;
    int is_run = has_run && ((run_flags[i >> 3] >> (i & 7)) & 1);
    int64_t base = (int64_t)key << 16;
    if (is_run) {
      if (pos + 2 > len) goto trunc;
      uint32_t n_runs = le16(blob + pos);
      pos += 2;
      if (pos + 4ULL * n_runs > len) goto trunc;
      if (dst) {
        if ((base >> 6) + 1024 > dst->n_words) goto range;
        for (uint32_t r = 0; r < n_runs; r++) {
          uint32_t s = le16(blob + pos + 4 * r), l = le16(blob + pos + 4 * r + 2);
          po_bitmap_add_range(dst, base + s, base + s + l + 1);
        }
      }
      pos += 4ULL * n_runs;
      if (n_run) (*n_run)++;
    } else if (card > 4096) {
      if (pos + 8192 > len) goto trunc;
      if (dst) {
        if ((base >> 6) + 1024 > dst->n_words) goto range;
        uint64_t* w = dst->words + (base >> 6);
        for (int k = 0; k < 1024; k++) w[k] |= le64(blob + pos + 8 * k);
      }
      pos += 8192;
      if (n_bitmap) (*n_bitmap)++;
    } else {
      if (pos + 2ULL * card > len) goto trunc;
      if (dst) {
        if ((base >> 6) + 1024 > dst->n_words) goto range;
        for (uint32_t k = 0; k < card; k++) po_bitmap_add(dst, (int32_t)(base + le16(blob + pos + 2 * k)));
      }
      pos += 2ULL * card;
      if (n_array) (*n_array)++;
    }
  }
  return 0;
trunc:
  po_set_error("roaring: truncated blob");
  return -1;
range:
  po_set_error("roaring: container key beyond the segment's doc range");
  return -1;
}

int po_roaring_deserialize_or(const uint8_t* blob, uint64_t len, po_bitmap* dst) {
  return roaring_walk(blob, len, dst, NULL, NULL, NULL);
}

int po_roaring_container_stats(const uint8_t* blob, uint64_t len, int* n_array, int* n_bitmap, int* n_run) {
  *n_array = *n_bitmap = *n_run = 0;
  return roaring_walk(blob, len, NULL, n_array, n_bitmap, n_run);
}
