/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 * Storage readers of the hot path: SURVEY.md §8a rows a5, a12, a13, a14 and the sorted index.
 */
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "po_internal.h"

/* ---- error / alloc helpers ------------------------------------------------------------------------------------------ */
static __thread char g_err[4096];

void po_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* po_get_error(void) { return g_err; }

void* po_xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) abort();
  return p;
}
void* po_xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz ? sz : 1);
  if (!p) abort();
  return p;
}
void* po_xrealloc(void* p, size_t n) {
  void* q = realloc(p, n ? n : 1);
  if (!q) abort();
  return q;
}

/* ---- FixedBitIntReader ------------------------------------------------------------------------------------------------
 * pinot-segment-local/.../io/reader/impl/FixedBitIntReader.java:27-119 dispatches to one unrolled class per bit width
 * (e.g. Bit5Reader :375-420); all of them are specialisations of the generic MSB-first read in
 * pinot-segment-local/.../io/util/PinotDataBitSet.java:74-97 (readInt), restated here once. */
int32_t po_fixedbit_read(const po_column* c, int32_t index) {
  int bits = c->bits_per_value;
  int64_t bit_offset = (int64_t)index * bits;
  int64_t byte_offset = bit_offset >> 3;
  int bit_in_first = (int)(bit_offset & 7);
  const uint8_t* p = c->fwd;
  int32_t cur = p[byte_offset] & (0xff >> bit_in_first);
  int left = bits - (8 - bit_in_first);
  if (left <= 0) return cur >> -left;
  while (left > 8) {
    byte_offset++;
    cur = (cur << 8) | p[byte_offset];
    left -= 8;
  }
  return (cur << left) | (p[byte_offset + 1] >> (8 - left));
}

/* FixedBitIntReader.BitNReader#read32: 32 values from `bits` big-endian ints starting at index (multiple of 32). */
static void fixedbit_read32(const po_column* c, int32_t index, int32_t* out) {
  int bits = c->bits_per_value;
  const uint8_t* p = c->fwd + ((int64_t)(index >> 3)) * bits;
  uint32_t mask = (bits == 32) ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  uint64_t acc = 0;   /* bit accumulator, MSB first */
  int have = 0;
  int w = 0;
  for (int i = 0; i < 32; i++) {
    while (have < bits) {
      acc = (acc << 32) | po_be32(p + 4 * w);
      w++;
      have += 32;
    }
    out[i] = (int32_t)((acc >> (have - bits)) & mask);
    have -= bits;
    acc &= (have == 0) ? 0 : ((1ULL << have) - 1);
  }
}

/* FixedBitSVForwardIndexReaderV2#readDictIds, pinot-segment-local/.../readers/forward/FixedBitSVForwardIndexReaderV2.java:65-99
 * (bulk read32 for runs of >= 64 sequential docIds; the checked/unchecked split only guards buffer ends and returns
 * the same values, so both map onto po_fixedbit_read). */
void po_fwd_read_dict_ids(const po_column* c, const int32_t* doc_ids, int32_t length, int32_t* out) {
  if (length <= 0) return;
  if (c->fwd_encoding == PG_FWD_DICT_SORTED) {
    /* SortedIndexReaderImpl#readDictIds: per-doc binary search with a moving context */
    for (int i = 0; i < length; i++) out[i] = po_sorted_get_dict_id(c, doc_ids[i]);
    return;
  }
  int32_t first = doc_ids[0], last = doc_ids[length - 1];
  int index = 0;
  if (last - first + 1 == length && length >= 64) {
    int32_t bulk_start = (first + 31) & (int32_t)0xffffffe0;
    int32_t bulk_end = last & (int32_t)0xffffffe0;
    for (int32_t i = first; i < bulk_start; i++) out[index++] = po_fixedbit_read(c, i);
    for (int32_t i = bulk_start; i < bulk_end; i += 32) {
      fixedbit_read32(c, i, out + index);
      index += 32;
    }
  }
  for (int i = index; i < length; i++) out[i] = po_fixedbit_read(c, doc_ids[i]);
}

/* ---- FixedBitMVForwardIndexReader -------------------------------------------------------------------------------------
 * pinot-segment-local/.../segment/index/readers/forward/FixedBitMVForwardIndexReader.java:57-76 (layout), :94-131 (getDictIdMV);
 * PinotDataBitSet#getNextSetBitOffset / getNextNthSetBitOffset (pinot-segment-local/.../io/util/PinotDataBitSet.java) — bits are
 * MSB first within a byte. */
int po_mv_parse(po_column* c) {
  const int32_t num_docs = c->num_docs, num_values = c->total_entries;
  if (num_docs <= 0 || num_values < num_docs) {
    po_set_error("multi-value column %s: %d values over %d docs", c->name, num_values, num_docs);
    return -1;
  }
  c->is_mv = 1;
  c->mv_docs_per_chunk = (int32_t)ceilf((float)2048 / (float)(num_values / num_docs));   /* :62 (PREFERRED_NUM_VALUES_PER_CHUNK, int division) */
  const int64_t num_chunks = ((int64_t)num_docs + c->mv_docs_per_chunk - 1) / c->mv_docs_per_chunk;
  const int64_t bitmap_size = ((int64_t)num_values + 7) / 8;
  const int64_t raw_size = ((int64_t)num_values * c->bits_per_value + 7) / 8;
  if ((int64_t)c->fwd_len < num_chunks * 4 + bitmap_size + raw_size) {
    po_set_error("multi-value forward index of %s is %llu bytes, need %lld", c->name, (unsigned long long)c->fwd_len,
                 (long long)(num_chunks * 4 + bitmap_size + raw_size));
    return -1;
  }
  c->mv_chunk_offsets = c->fwd;
  c->mv_bitmap = c->fwd + num_chunks * 4;
  c->mv_raw = c->mv_bitmap + bitmap_size;
  /* getMaxNumberOfMultiValues: the longest gap between row starts */
  int32_t prev = -1, longest = 0;
  for (int32_t i = 0; i < num_values; i++) {
    if (c->mv_bitmap[i >> 3] & (0x80 >> (i & 7))) {
      if (prev >= 0 && i - prev > longest) longest = i - prev;
      prev = i;
    }
  }
  if (prev >= 0 && num_values - prev > longest) longest = num_values - prev;
  c->mv_max_values = longest;
  return 0;
}
/* FixedBitMVEntryDictForwardIndexReader (.../readers/forward/FixedBitMVEntryDictForwardIndexReader.java; the MV_ENTRY_DICT format of
 * FixedBitMVEntryDictForwardIndexWriter.java:80-130): header magic 0xffabcdef, short version 1, byte bitsPerValue, byte bitsPerId, int
 * uniqueEntries, int totalValues, int offsetBufferOffset, int valueBufferOffset; then three PinotDataBitSet arrays — the docs' entry ids,
 * the entries' start offsets, the entries' dictIds.  getDictIdMV(doc) = values[offsets[id] .. offsets[id + 1]).  Read once into
 * FixedBitMVForwardIndexReader's layout (owned), which the rest of this file walks. */
static uint32_t bitset_read(const uint8_t* base, int64_t index, int bits) {   /* PinotDataBitSet#readInt */
  uint32_t v = 0;
  const int64_t bit0 = index * bits;
  for (int b = 0; b < bits; b++) { const int64_t at = bit0 + b; v = (v << 1) | ((base[at >> 3] >> (7 - (at & 7))) & 1u); }
  return v;
}
int po_mv_entry_dict_attach(po_column* c) {
  const uint8_t* f = c->fwd;
  if (c->fwd_len < 24 || po_be32(f) != 0xffabcdefu || (po_be32(f + 4) >> 16) != 1u) { po_set_error("column %s: not an MV_ENTRY_DICT forward index of version 1", c->name); return -1; }
  const int bits_v = f[6], bits_id = f[7];
  const int64_t n_unique = (int32_t)po_be32(f + 8), n_total = (int32_t)po_be32(f + 12), off_at = (int32_t)po_be32(f + 16), val_at = (int32_t)po_be32(f + 20);
  int bits_off = 1;
  while (bits_off < 31 && ((int64_t)1 << bits_off) <= n_total) bits_off++;
  const int64_t nd = c->num_docs;
  if (bits_v != c->bits_per_value || n_unique <= 0 || off_at != 24 + (nd * bits_id + 7) / 8 || val_at != off_at + ((n_unique + 1) * bits_off + 7) / 8 ||
      (uint64_t)val_at + (uint64_t)((n_total * bits_v + 7) / 8) > c->fwd_len) {
    po_set_error("MV_ENTRY_DICT forward index of %s: inconsistent header", c->name);
    return -1;
  }
  int64_t total = 0;
  int32_t* starts = (int32_t*)po_xcalloc((size_t)nd + 1, sizeof(int32_t));
  for (int64_t d = 0; d < nd; d++) {
    const uint32_t id = bitset_read(f + 24, d, bits_id);
    const int64_t a = bitset_read(f + off_at, id, bits_off), b = bitset_read(f + off_at, (int64_t)id + 1, bits_off);
    if ((int64_t)id >= n_unique || b <= a || b > n_total) { free(starts); po_set_error("MV_ENTRY_DICT forward index of %s: bad entry of doc %lld", c->name, (long long)d); return -1; }
    starts[d] = (int32_t)total;
    total += b - a;
  }
  starts[nd] = (int32_t)total;
  const int bits = c->bits_per_value;
  const int64_t per_chunk = (int64_t)ceilf((float)2048 / (float)(total / nd));
  const int64_t num_chunks = (nd + per_chunk - 1) / per_chunk;
  const int64_t bitmap_size = (total + 7) / 8, raw_size = (total * bits + 7) / 8;
  uint8_t* fwd = (uint8_t*)po_xcalloc((size_t)(num_chunks * 4 + bitmap_size + raw_size + 8), 1);
  for (int64_t ch = 0; ch < num_chunks; ch++) {
    const uint32_t o = (uint32_t)starts[ch * per_chunk];
    fwd[ch * 4] = (uint8_t)(o >> 24); fwd[ch * 4 + 1] = (uint8_t)(o >> 16); fwd[ch * 4 + 2] = (uint8_t)(o >> 8); fwd[ch * 4 + 3] = (uint8_t)o;
  }
  uint8_t* bm = fwd + num_chunks * 4;
  uint8_t* packed = bm + bitmap_size;
  int64_t e = 0;
  for (int64_t d = 0; d < nd; d++) {
    bm[starts[d] >> 3] |= (uint8_t)(0x80u >> (starts[d] & 7));
    const uint32_t id = bitset_read(f + 24, d, bits_id);
    const int64_t a = bitset_read(f + off_at, id, bits_off), b = bitset_read(f + off_at, (int64_t)id + 1, bits_off);
    for (int64_t k = a; k < b; k++, e++) {
      const uint32_t v = bitset_read(f + val_at, k, bits_v);
      const int64_t bit0 = e * bits;
      for (int bb = 0; bb < bits; bb++)
        if ((v >> (bits - 1 - bb)) & 1u) packed[(bit0 + bb) >> 3] |= (uint8_t)(0x80u >> ((bit0 + bb) & 7));
    }
  }
  free(starts);
  c->mv_owned_fwd = fwd;
  c->fwd = fwd;
  c->fwd_len = (uint64_t)(num_chunks * 4 + bitmap_size + raw_size);
  c->total_entries = (int32_t)total;
  return po_mv_parse(c);
}

static inline int mv_bit(const po_column* c, int32_t i) { return (c->mv_bitmap[i >> 3] >> (7 - (i & 7))) & 1; }
static int32_t mv_next_set_bit(const po_column* c, int32_t from) {          /* getNextSetBitOffset(bitOffset) */
  while (!mv_bit(c, from)) from++;
  return from;
}
static int32_t mv_next_nth_set_bit(const po_column* c, int32_t from, int32_t n) {   /* getNextNthSetBitOffset(bitOffset, n), n >= 1 */
  for (;; from++) {
    if (mv_bit(c, from) && --n == 0) return from;
  }
}
static int32_t mv_fixedbit(const po_column* c, int32_t index) {   /* FixedBitIntReaderWriter#readInt over the raw data view */
  po_column v;
  memset(&v, 0, sizeof(v));
  v.fwd = c->mv_raw;
  v.bits_per_value = c->bits_per_value;
  return po_fixedbit_read(&v, index);
}
int32_t po_mv_get_dict_ids(const po_column* c, int32_t doc_id, int32_t* buf, po_mv_ctx* ctx) {
  int32_t start;
  if (doc_id == ctx->doc_id + 1) {
    start = ctx->end_offset;
  } else {
    const int32_t chunk = doc_id / c->mv_docs_per_chunk;
    if (doc_id > ctx->doc_id && chunk == ctx->doc_id / c->mv_docs_per_chunk) {   /* same chunk (a fresh context: -1 / n == 0, as in Java) */
      start = mv_next_nth_set_bit(c, ctx->end_offset + 1, doc_id - ctx->doc_id - 1);
    } else {
      const int32_t chunk_offset = (int32_t)po_be32(c->mv_chunk_offsets + (int64_t)chunk * 4);
      const int32_t in_chunk = doc_id % c->mv_docs_per_chunk;
      start = in_chunk == 0 ? chunk_offset : mv_next_nth_set_bit(c, chunk_offset + 1, in_chunk);
    }
  }
  const int32_t end = doc_id == c->num_docs - 1 ? c->total_entries : mv_next_set_bit(c, start + 1);
  const int32_t n = end - start;
  for (int32_t i = 0; i < n; i++) buf[i] = mv_fixedbit(c, start + i);
  ctx->doc_id = doc_id;
  ctx->end_offset = end;
  return n;
}

/* ---- FixedByteChunkSVForwardIndexReader (PASS_THROUGH) ---------------------------------------------------------------
 * header parse: BaseChunkForwardIndexReader.java:61-111; value access: FixedByteChunkSVForwardIndexReader.java:53-94
 * (`_rawData.getInt(docId * Integer.BYTES)`; the reference multiplies in int, so it wraps for docId >= 2^29 — the
 * oracle uses 64-bit offsets, i.e. restates the intent, and tests stay below that bound). */
int po_raw_parse_header(po_column* c) {
  const uint8_t* b = c->fwd;
  if (c->fwd_len < 16) {
    po_set_error("raw forward index of %s too short", c->name);
    return -1;
  }
  int32_t version = (int32_t)po_be32(b);
  c->raw_version = version;
  c->raw_num_chunks = (int32_t)po_be32(b + 4);
  c->raw_docs_per_chunk = (int32_t)po_be32(b + 8);
  c->raw_entry_len = (int32_t)po_be32(b + 12);
  int32_t data_header_start = 16;
  if (version > 1) {
    c->raw_compression = (int32_t)po_be32(b + 20);
    data_header_start = (int32_t)po_be32(b + 24);
  } else {
    c->raw_compression = 1; /* SNAPPY */
  }
  int off_size = version <= 2 ? 4 : 8;
  if (c->raw_compression != 0) {
    /* BaseChunkForwardIndexReader#decompressChunk (:141-163) decompresses the chunk of the docId being read into the reader
     * context; the values are the same if every chunk is decompressed once, up front, into the PASS_THROUGH layout (version 3
     * header, 8-byte chunk offsets) the accessors below read. */
    if (c->raw_compression < 1 || c->raw_compression > 5) {
      po_set_error("column %s: chunk compression type %d is not a ChunkCompressionType", c->name, c->raw_compression);
      return -1;
    }
    const int64_t nc = c->raw_num_chunks;
    const uint64_t cap = (uint64_t)c->raw_docs_per_chunk * (uint64_t)(4 + c->raw_entry_len);
    const uint64_t head = 28 + (uint64_t)nc * 8;
    uint8_t* out = (uint8_t*)po_xcalloc(1, head + (uint64_t)nc * cap + 16);
    uint64_t pos = head;
    for (int64_t i = 0; i < nc; i++) {
      const uint8_t* o = b + data_header_start + i * off_size;
      int64_t start = off_size == 4 ? (int64_t)(int32_t)po_be32(o) : (int64_t)po_be64(o);
      int64_t end = i == nc - 1 ? (int64_t)c->fwd_len : (off_size == 4 ? (int64_t)(int32_t)po_be32(o + 4) : (int64_t)po_be64(o + 8));
      if (start < 0 || end < start || (uint64_t)end > c->fwd_len) { free(out); po_set_error("column %s: bad chunk offsets", c->name); return -1; }
      int64_t got = po_chunk_decompress(c->raw_compression, b + start, (uint64_t)(end - start), out + pos, cap);
      if (got < 0) { free(out); po_set_error("column %s: chunk %lld does not decompress", c->name, (long long)i); return -1; }
      for (int k = 0; k < 8; k++) out[28 + i * 8 + k] = (uint8_t)(pos >> (56 - 8 * k));
      pos += (uint64_t)got;
    }
    const uint32_t hdr[7] = {3, (uint32_t)nc, (uint32_t)c->raw_docs_per_chunk, (uint32_t)c->raw_entry_len, (uint32_t)c->num_docs, 0, 28};
    for (int w = 0; w < 7; w++) for (int k = 0; k < 4; k++) out[w * 4 + k] = (uint8_t)(hdr[w] >> (24 - 8 * k));
    c->raw_owned = out;
    c->fwd = out;
    c->fwd_len = pos;
    c->raw_version = 3;
    c->raw_compression = 0;
    c->raw_data = out + head;
    return 0;
  }
  int64_t raw_start = (int64_t)data_header_start + (int64_t)c->raw_num_chunks * off_size;
  c->raw_data = b + raw_start;
  return 0;
}
int32_t po_raw_get_int(const po_column* c, int32_t d) { return (int32_t)po_be32(c->raw_data + (int64_t)d * 4); }
int64_t po_raw_get_long(const po_column* c, int32_t d) { return (int64_t)po_be64(c->raw_data + (int64_t)d * 8); }
float po_raw_get_float(const po_column* c, int32_t d) { return po_bef32(c->raw_data + (int64_t)d * 4); }
double po_raw_get_double(const po_column* c, int32_t d) { return po_bef64(c->raw_data + (int64_t)d * 8); }

/* VarByteChunkSVForwardIndexReader#getBytesUncompressed (:158-217): chunk = numDocsPerChunk BE int offsets relative to the
 * chunk start (0 for the absent rows of the last chunk), then the values */
const uint8_t* po_raw_get_bytes(const po_column* c, int32_t doc_id, int32_t* len) {
  const uint8_t* b = c->fwd;
  int32_t chunk = doc_id / c->raw_docs_per_chunk, row = doc_id % c->raw_docs_per_chunk;
  int off_size = c->raw_version <= 2 ? 4 : 8;
  const uint8_t* offs = c->raw_data - (int64_t)c->raw_num_chunks * off_size;   /* chunk position table */
  int64_t chunk_start = off_size == 4 ? (int64_t)(int32_t)po_be32(offs + (int64_t)chunk * 4) : (int64_t)po_be64(offs + (int64_t)chunk * 8);
  int64_t chunk_end = chunk == c->raw_num_chunks - 1
                          ? (int64_t)c->fwd_len
                          : (off_size == 4 ? (int64_t)(int32_t)po_be32(offs + (int64_t)(chunk + 1) * 4) : (int64_t)po_be64(offs + (int64_t)(chunk + 1) * 8));
  int64_t start = chunk_start + (int32_t)po_be32(b + chunk_start + (int64_t)row * 4);
  int64_t end;
  if (row == c->raw_docs_per_chunk - 1) end = chunk_end;
  else {
    int32_t nxt = (int32_t)po_be32(b + chunk_start + (int64_t)(row + 1) * 4);
    end = nxt == 0 ? chunk_end : chunk_start + nxt;
  }
  *len = (int32_t)(end - start);
  return b + start;
}

/* ---- FixedByteChunkMVForwardIndexReader (raw multi-value column of INT / LONG / FLOAT / DOUBLE) --------------------------------
 * pinot-segment-local/.../readers/forward/FixedByteChunkMVForwardIndexReader.java:35-140: slice(docId) is the var-byte chunk value of the
 * doc (sliceBytesUncompressed :118-131 = the offsets walk of po_raw_get_bytes above; compressed chunks through getChunkBuffer), and
 * ArraySerDeUtils.deserialize…ArrayWithLength reads a big-endian int numValues followed by the values big-endian
 * (pinot-segment-local/.../utils/ArraySerDeUtils.java); getNumValuesMV = slice.getInt().
 *
 * The oracle's multi-value operators (mvscan_*, gkg_generate_mv, the *MV branches of agg_process_block) are written over dictIds.  A raw
 * column is attached to them through an equivalent dictionary encoding built here from the values this reader returns — sorted distinct
 * values (the order SegmentDictionaryCreator would give them) and the docs' ids in FixedBitMVForwardIndexReader's layout — instead of a
 * second copy of those operators over raw values.  What this does NOT restate separately are the raw-value predicate evaluators and the
 * NoDictionary…GroupKeyGenerators over multi-value blocks: their RESULTS are pinned by MultiValueRawQueriesTest, which asserts raw ==
 * dictionary twin for every query it runs (tests/test_mv_reference_goldens.py).  `raw_mv` keeps the two differences that are visible:
 * no NonScanBasedAggregationOperator (no dictionary to answer from), DISTINCTCOUNTMV refused (its intermediate is a VALUE set). */
static uint64_t raw_mv_key(int32_t data_type, const uint8_t* p) {   /* order-preserving 64-bit image of a stored value */
  if (data_type == PG_TYPE_INT) return (uint64_t)(int64_t)(int32_t)po_be32(p) ^ (1ULL << 63);
  if (data_type == PG_TYPE_LONG) return po_be64(p) ^ (1ULL << 63);
  if (data_type == PG_TYPE_FLOAT) { const uint32_t f = po_be32(p); return (f >> 31) ? (uint64_t)(uint32_t)~f : (uint64_t)(f ^ 0x80000000u); }
  const uint64_t f = po_be64(p);
  return (f >> 63) ? ~f : (f ^ (1ULL << 63));
}
static int cmp_u64(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
int po_raw_mv_attach(po_column* c) {
  if (c->has_dictionary || c->data_type > PG_TYPE_DOUBLE) { po_set_error("column %s: raw multi-value index of a dictionary / non-numeric column", c->name); return -1; }
  if (po_raw_parse_header(c)) return -1;
  const int width = (c->data_type == PG_TYPE_INT || c->data_type == PG_TYPE_FLOAT) ? 4 : 8;
  const int32_t n_docs = c->num_docs;
  if (n_docs <= 0) { po_set_error("raw multi-value column %s of %d docs", c->name, n_docs); return -1; }
  /* pass 1: getNumValuesMV of every doc */
  int64_t total = 0;
  int32_t* starts = (int32_t*)po_xcalloc((size_t)n_docs + 1, sizeof(int32_t));
  for (int32_t d = 0; d < n_docs; d++) {
    int32_t len = 0;
    const uint8_t* v = po_raw_get_bytes(c, d, &len);
    if (len < 4) { free(starts); po_set_error("raw multi-value index of %s: doc %d has %d bytes", c->name, d, len); return -1; }
    const int64_t n = (int64_t)(int32_t)po_be32(v);
    if (n <= 0 || n * width + 4 != len) { free(starts); po_set_error("raw multi-value index of %s: doc %d: %lld values in %d bytes", c->name, d, (long long)n, len); return -1; }
    starts[d] = (int32_t)total;
    total += n;
  }
  starts[n_docs] = (int32_t)total;
  /* pass 2: the values (deserialize…ArrayWithLength) as keys; dictionary = sorted distinct */
  uint64_t* keys = (uint64_t*)po_xcalloc((size_t)total, sizeof(uint64_t));
  for (int32_t d = 0; d < n_docs; d++) {
    int32_t len = 0;
    const uint8_t* v = po_raw_get_bytes(c, d, &len);
    const int32_t n = starts[d + 1] - starts[d];
    for (int32_t i = 0; i < n; i++) keys[starts[d] + i] = raw_mv_key(c->data_type, v + 4 + (int64_t)i * width);
  }
  uint64_t* sorted = (uint64_t*)po_xcalloc((size_t)total, sizeof(uint64_t));
  memcpy(sorted, keys, (size_t)total * sizeof(uint64_t));
  qsort(sorted, (size_t)total, sizeof(uint64_t), cmp_u64);
  int64_t card = 0;
  for (int64_t i = 0; i < total; i++) if (i == 0 || sorted[i] != sorted[i - 1]) sorted[card++] = sorted[i];
  int bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < card) bits++;
  uint8_t* dict = (uint8_t*)po_xcalloc((size_t)card * (size_t)width + 8, 1);
  for (int64_t i = 0; i < card; i++) {
    uint64_t raw;
    const uint64_t k = sorted[i];
    if (c->data_type == PG_TYPE_INT || c->data_type == PG_TYPE_LONG) raw = k ^ (1ULL << 63);
    else if (c->data_type == PG_TYPE_FLOAT) { const uint32_t kk = (uint32_t)k; raw = (kk >> 31) ? (kk ^ 0x80000000u) : (uint32_t)~kk; }
    else raw = (k >> 63) ? (k ^ (1ULL << 63)) : ~k;
    for (int b = 0; b < width; b++) dict[i * width + b] = (uint8_t)(raw >> (8 * (width - 1 - b)));
  }
  const int64_t per_chunk = (int64_t)ceilf((float)2048 / (float)(total / n_docs));
  const int64_t num_chunks = ((int64_t)n_docs + per_chunk - 1) / per_chunk;
  const int64_t bitmap_size = (total + 7) / 8, raw_size = (total * bits + 7) / 8;
  uint8_t* fwd = (uint8_t*)po_xcalloc((size_t)(num_chunks * 4 + bitmap_size + raw_size + 8), 1);
  for (int64_t ch = 0; ch < num_chunks; ch++) {
    const uint32_t o = (uint32_t)starts[ch * per_chunk];
    fwd[ch * 4] = (uint8_t)(o >> 24); fwd[ch * 4 + 1] = (uint8_t)(o >> 16); fwd[ch * 4 + 2] = (uint8_t)(o >> 8); fwd[ch * 4 + 3] = (uint8_t)o;
  }
  uint8_t* bm = fwd + num_chunks * 4;
  for (int32_t d = 0; d < n_docs; d++) bm[starts[d] >> 3] |= (uint8_t)(0x80u >> (starts[d] & 7));
  uint8_t* packed = bm + bitmap_size;
  for (int64_t e = 0; e < total; e++) {
    int64_t lo = 0, hi = card - 1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted[mid] < keys[e]) lo = mid + 1; else hi = mid; }
    const uint32_t id = (uint32_t)lo;
    const int64_t bit0 = e * bits;
    for (int b = 0; b < bits; b++)
      if ((id >> (bits - 1 - b)) & 1u) packed[(bit0 + b) >> 3] |= (uint8_t)(0x80u >> ((bit0 + b) & 7));
  }
  free(keys); free(sorted); free(starts);
  c->raw_mv = 1;
  c->mv_owned_fwd = fwd;
  c->mv_owned_dict = dict;
  c->has_dictionary = 1;
  c->cardinality = (int32_t)card;
  c->bits_per_value = bits;
  c->dict_bytes_per_value = width;
  c->dict = dict;
  c->dict_len = (uint64_t)card * (uint64_t)width;
  c->fwd = fwd;
  c->fwd_len = (uint64_t)(num_chunks * 4 + bitmap_size + raw_size);
  c->fwd_encoding = PG_FWD_DICT_FIXED_BIT_MV;
  c->total_entries = (int32_t)total;
  return po_mv_parse(c);
}

/* VarByteChunkMVForwardIndexReader over STRING values (.../readers/forward/VarByteChunkMVForwardIndexReader.java; getStringMV =
 * ArraySerDeUtils.deserializeStringArray, .../utils/ArraySerDeUtils.java:282-307: int numValues, numValues int lengths, the UTF-8 bytes): the
 * column is read once into a dictionary-encoded twin like the fixed-width form above — distinct strings in byte order as a zero-padded
 * fixed-width dictionary, ids in FixedBitMVForwardIndexReader's layout.  Group keys are reported as ids of that dictionary (the tests
 * decode them through the host model's sorted distinct values). */
typedef struct { const uint8_t* p; int32_t len; } po_str;
static int cmp_str(const void* a, const void* b) {
  const po_str* x = (const po_str*)a; const po_str* y = (const po_str*)b;
  const int32_t n = x->len < y->len ? x->len : y->len;
  const int c = n ? memcmp(x->p, y->p, (size_t)n) : 0;
  return c ? c : (x->len < y->len ? -1 : (x->len > y->len ? 1 : 0));
}
int po_raw_mv_attach_strings(po_column* c) {
  if (c->has_dictionary || c->data_type != PG_TYPE_STRING) { po_set_error("column %s: raw multi-value var-byte index of a dictionary / non-STRING column", c->name); return -1; }
  if (po_raw_parse_header(c)) return -1;
  const int32_t n_docs = c->num_docs;
  if (n_docs <= 0) { po_set_error("raw multi-value column %s of %d docs", c->name, n_docs); return -1; }
  int64_t total = 0;
  int32_t* starts = (int32_t*)po_xcalloc((size_t)n_docs + 1, sizeof(int32_t));
  for (int32_t d = 0; d < n_docs; d++) {
    int32_t len = 0;
    const uint8_t* v = po_raw_get_bytes(c, d, &len);
    const int64_t n = len >= 4 ? (int64_t)(int32_t)po_be32(v) : 0;
    if (n <= 0 || 4 + n * 4 > len) { free(starts); po_set_error("raw multi-value index of %s: doc %d: %lld values in %d bytes", c->name, d, (long long)n, len); return -1; }
    starts[d] = (int32_t)total;
    total += n;
  }
  starts[n_docs] = (int32_t)total;
  po_str* vals = (po_str*)po_xcalloc((size_t)total, sizeof(po_str));
  for (int32_t d = 0; d < n_docs; d++) {
    int32_t len = 0;
    const uint8_t* v = po_raw_get_bytes(c, d, &len);
    const int32_t n = starts[d + 1] - starts[d];
    int64_t at = 4 + (int64_t)n * 4;
    for (int32_t i = 0; i < n; i++) {
      const int32_t l = (int32_t)po_be32(v + 4 + (int64_t)i * 4);
      if (l < 0 || at + l > len) { free(vals); free(starts); po_set_error("raw multi-value index of %s: doc %d: value lengths beyond %d bytes", c->name, d, len); return -1; }
      vals[starts[d] + i].p = v + at;
      vals[starts[d] + i].len = l;
      at += l;
    }
  }
  po_str* sorted = (po_str*)po_xcalloc((size_t)total, sizeof(po_str));
  memcpy(sorted, vals, (size_t)total * sizeof(po_str));
  qsort(sorted, (size_t)total, sizeof(po_str), cmp_str);
  int64_t card = 0;
  int32_t width = 1;
  for (int64_t i = 0; i < total; i++) if (i == 0 || cmp_str(&sorted[i], &sorted[card - 1]) != 0) sorted[card++] = sorted[i];
  for (int64_t i = 0; i < card; i++) if (sorted[i].len > width) width = sorted[i].len;
  int bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < card) bits++;
  uint8_t* dict = (uint8_t*)po_xcalloc((size_t)card * (size_t)width + 8, 1);
  for (int64_t i = 0; i < card; i++) memcpy(dict + i * width, sorted[i].p, (size_t)sorted[i].len);
  const int64_t per_chunk = (int64_t)ceilf((float)2048 / (float)(total / n_docs));
  const int64_t num_chunks = ((int64_t)n_docs + per_chunk - 1) / per_chunk;
  const int64_t bitmap_size = (total + 7) / 8, raw_size = (total * bits + 7) / 8;
  uint8_t* fwd = (uint8_t*)po_xcalloc((size_t)(num_chunks * 4 + bitmap_size + raw_size + 8), 1);
  for (int64_t ch = 0; ch < num_chunks; ch++) {
    const uint32_t o = (uint32_t)starts[ch * per_chunk];
    fwd[ch * 4] = (uint8_t)(o >> 24); fwd[ch * 4 + 1] = (uint8_t)(o >> 16); fwd[ch * 4 + 2] = (uint8_t)(o >> 8); fwd[ch * 4 + 3] = (uint8_t)o;
  }
  uint8_t* bm = fwd + num_chunks * 4;
  for (int32_t d = 0; d < n_docs; d++) bm[starts[d] >> 3] |= (uint8_t)(0x80u >> (starts[d] & 7));
  uint8_t* packed = bm + bitmap_size;
  for (int64_t e = 0; e < total; e++) {
    int64_t lo = 0, hi = card - 1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cmp_str(&sorted[mid], &vals[e]) < 0) lo = mid + 1; else hi = mid; }
    const uint32_t id = (uint32_t)lo;
    const int64_t bit0 = e * bits;
    for (int b = 0; b < bits; b++)
      if ((id >> (bits - 1 - b)) & 1u) packed[(bit0 + b) >> 3] |= (uint8_t)(0x80u >> ((bit0 + b) & 7));
  }
  free(vals); free(sorted); free(starts);
  c->raw_mv = 1;
  c->mv_owned_fwd = fwd;
  c->mv_owned_dict = dict;
  c->has_dictionary = 1;
  c->cardinality = (int32_t)card;
  c->bits_per_value = bits;
  c->dict_bytes_per_value = width;
  c->dict = dict;
  c->dict_len = (uint64_t)card * (uint64_t)width;
  c->fwd = fwd;
  c->fwd_len = (uint64_t)(num_chunks * 4 + bitmap_size + raw_size);
  c->fwd_encoding = PG_FWD_DICT_FIXED_BIT_MV;
  c->total_entries = (int32_t)total;
  return po_mv_parse(c);
}

/* ---- SortedIndexReaderImpl, pinot-segment-local/.../readers/sorted/SortedIndexReaderImpl.java:35-110 ----------------- */
void po_sorted_get_doc_ids(const po_column* c, int32_t dict_id, int32_t* start, int32_t* end) {
  *start = (int32_t)po_be32(c->fwd + (int64_t)dict_id * 8);
  *end = (int32_t)po_be32(c->fwd + (int64_t)dict_id * 8 + 4);
}
int32_t po_sorted_get_dict_id(const po_column* c, int32_t doc_id) {
  int32_t lo = 0, hi = c->cardinality - 1;
  while (lo <= hi) {
    int32_t mid = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
    int32_t s, e;
    po_sorted_get_doc_ids(c, mid, &s, &e);
    if (e < doc_id) lo = mid + 1;
    else if (s > doc_id) hi = mid - 1;
    else return mid;
  }
  return -1;
}

/* ---- dictionaries: BaseImmutableDictionary.java:124-245, IntDictionary.java:28-80 & siblings -------------------------- */
int32_t po_dict_get_int(const po_column* c, int32_t id) { return (int32_t)po_be32(c->dict + (int64_t)id * 4); }
int64_t po_dict_get_long(const po_column* c, int32_t id) { return (int64_t)po_be64(c->dict + (int64_t)id * 8); }
float po_dict_get_float(const po_column* c, int32_t id) { return po_bef32(c->dict + (int64_t)id * 4); }
static double dict_raw_double(const po_column* c, int32_t id) { return po_bef64(c->dict + (int64_t)id * 8); }

/* Dictionary#getDoubleValue: (double) of the stored value (IntDictionary.java: getDoubleValue = getInt(dictId)). */
double po_dict_get_double(const po_column* c, int32_t id) {
  switch (c->data_type) {
    case PG_TYPE_INT: return (double)po_dict_get_int(c, id);
    case PG_TYPE_LONG: return (double)po_dict_get_long(c, id);
    case PG_TYPE_FLOAT: return (double)po_dict_get_float(c, id);
    case PG_TYPE_DOUBLE: return dict_raw_double(c, id);
    default: {
      /* StringDictionary#getDoubleValue = Double.parseDouble(unpadded string) */
      char buf[512];
      int w = c->dict_bytes_per_value;
      int n = w < 511 ? w : 511;
      memcpy(buf, c->dict + (int64_t)id * w, (size_t)n);
      buf[n] = 0;
      return strtod(buf, NULL);
    }
  }
}

static int parse_i64(const char* s, int64_t lo, int64_t hi, int64_t* out) {
  /* Integer.parseInt / Long.parseLong: optional sign, decimal digits only, range checked */
  if (!s || !*s) return -1;
  errno = 0;
  char* end = NULL;
  long long v = strtoll(s, &end, 10);
  if (errno || *end != 0 || end == s) return -1;
  for (const char* p = s; *p; p++)
    if (!((*p >= '0' && *p <= '9') || ((p == s) && (*p == '-' || *p == '+')))) return -1;
  if (v < lo || v > hi) return -1;
  *out = v;
  return 0;
}
int po_parse_int(const char* s, int32_t* out) {
  int64_t v;
  if (parse_i64(s, INT32_MIN, INT32_MAX, &v)) {
    po_set_error("NumberFormatException: For input string: \"%s\"", s ? s : "null");
    return -1;
  }
  *out = (int32_t)v;
  return 0;
}
int po_parse_long(const char* s, int64_t* out) {
  if (parse_i64(s, INT64_MIN, INT64_MAX, out)) {
    po_set_error("NumberFormatException: For input string: \"%s\"", s ? s : "null");
    return -1;
  }
  return 0;
}
int po_parse_double(const char* s, double* out) {
  if (!s || !*s) goto bad;
  {
    char* end = NULL;
    errno = 0;
    double v = strtod(s, &end);
    if (end == s || *end != 0) goto bad;
    *out = v;
    return 0;
  }
bad:
  po_set_error("NumberFormatException: For input string: \"%s\"", s ? s : "null");
  return -1;
}
int po_parse_float(const char* s, float* out) {
  if (!s || !*s) goto bad;
  {
    char* end = NULL;
    float v = strtof(s, &end);
    if (end == s || *end != 0) goto bad;
    *out = v;
    return 0;
  }
bad:
  po_set_error("NumberFormatException: For input string: \"%s\"", s ? s : "null");
  return -1;
}

/* ValueReaderComparisons.compareUtf8Bytes with padded entries: compare the unpadded stored value with `s` (byte order
 * equals code-point order for the BMP/ASCII values the tests use). */
static int compare_padded_string(const uint8_t* entry, int width, const uint8_t* s, int slen) {
  int elen = width;
  while (elen > 0 && entry[elen - 1] == 0) elen--;
  int n = elen < slen ? elen : slen;
  for (int i = 0; i < n; i++)
    if (entry[i] != s[i]) return entry[i] < s[i] ? -1 : 1;
  return elen == slen ? 0 : (elen < slen ? -1 : 1);
}

int32_t po_dict_insertion_index_of(const po_column* c, const char* sv) {
  int32_t low = 0, high = c->cardinality - 1;
  switch (c->data_type) {
    case PG_TYPE_INT: {
      int32_t v;
      if (po_parse_int(sv, &v)) return INT32_MIN;
      while (low <= high) {
        int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
        int32_t mv = po_dict_get_int(c, mid);
        if (mv < v) low = mid + 1; else if (mv > v) high = mid - 1; else return mid;
      }
      return -(low + 1);
    }
    case PG_TYPE_LONG: {
      int64_t v;
      if (po_parse_long(sv, &v)) return INT32_MIN;
      while (low <= high) {
        int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
        int64_t mv = po_dict_get_long(c, mid);
        if (mv < v) low = mid + 1; else if (mv > v) high = mid - 1; else return mid;
      }
      return -(low + 1);
    }
    case PG_TYPE_FLOAT: {
      float v;
      if (po_parse_float(sv, &v)) return INT32_MIN;
      while (low <= high) {
        int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
        float mv = po_dict_get_float(c, mid);
        if (mv < v) low = mid + 1; else if (mv > v) high = mid - 1; else return mid;
      }
      return -(low + 1);
    }
    case PG_TYPE_DOUBLE: {
      double v;
      if (po_parse_double(sv, &v)) return INT32_MIN;
      while (low <= high) {
        int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
        double mv = dict_raw_double(c, mid);
        if (mv < v) low = mid + 1; else if (mv > v) high = mid - 1; else return mid;
      }
      return -(low + 1);
    }
    default: {
      int slen = (int)strlen(sv);
      int w = c->dict_bytes_per_value;
      while (low <= high) {
        int32_t mid = (int32_t)(((uint32_t)low + (uint32_t)high) >> 1);
        int cmp = compare_padded_string(c->dict + (int64_t)mid * w, w, (const uint8_t*)sv, slen);
        if (cmp < 0) low = mid + 1; else if (cmp > 0) high = mid - 1; else return mid;
      }
      return -(low + 1);
    }
  }
}

/* ---- BitmapInvertedIndexReader#getDocIds, pinot-segment-local/.../readers/BitmapInvertedIndexReader.java:45-62 -------- */
int po_inv_get_doc_ids_or(const po_column* c, int32_t dict_id, po_bitmap* dst) {
  if (c->fwd_encoding == PG_FWD_DICT_SORTED) {
    int32_t s, e;
    po_sorted_get_doc_ids(c, dict_id, &s, &e);
    po_bitmap_add_range(dst, s, (int64_t)e + 1);
    return 0;
  }
  uint64_t off_end = ((uint64_t)c->cardinality + 1) * 4;
  uint64_t first = po_be32(c->inv);                                    /* _firstOffset */
  uint64_t off = po_be32(c->inv + (uint64_t)dict_id * 4);
  uint64_t len = po_be32(c->inv + (uint64_t)(dict_id + 1) * 4) - off;
  const uint8_t* bitmap_buffer = c->inv + off_end;                     /* _bitmapBuffer */
  return po_roaring_deserialize_or(bitmap_buffer + (off - first), len, dst);
}
