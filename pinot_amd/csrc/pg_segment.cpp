// Segment registration: pins the Pinot index entries of one column into HBM (pg_segment_add_column).
//
//   forward index   FixedBitSVForwardIndexReaderV2 bytes are uploaded verbatim (MSB-first big-endian bit stream,
//                   pinot-segment-local/.../readers/forward/FixedBitSVForwardIndexReaderV2.java:65-99); raw
//                   FixedByteChunkSVForwardIndexReader entries (PASS_THROUGH) are uploaded from `_rawDataStart`
//                   (BaseChunkForwardIndexReader.java:61-111) so that doc 0 sits on a 256-byte boundary; a sorted
//                   column's (start,end) pairs are expanded once to the fixed-bit layout.  Values stay big-endian in
//                   HBM; kernels byte-swap in registers.
//   dictionary      kept on the host big-endian (binary search like BaseImmutableDictionary.java:124-245) and
//                   uploaded native-endian for dictionary-encoded metric columns.
//   inverted index  BitmapInvertedIndexReader.java:45-62 offsets + portable RoaringBitmap blobs are parsed once;
//                   container payloads are copied 16-byte aligned into one device buffer with a descriptor per
//                   container (array / bitmap / run are all kept as is).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "pg_internal.hpp"

// Largest magnitude of a raw column, for exact SUMs: out[0] = max |value| (LONG: as uint64; FLOAT / DOUBLE: the bits of the largest
// finite |double|), out[1] = 1 when a NaN / Inf occurs.  One pass at registration, values big-endian as stored.
extern "C" __global__ void __launch_bounds__(256) pg_column_magnitude_kernel(const uint8_t* __restrict__ data, int64_t n, int val_type,
                                                                             unsigned long long* __restrict__ out) {
  unsigned long long mx = 0, bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long a;
    if (val_type == PG_V_I64) {
      const unsigned long long v = __builtin_bswap64(reinterpret_cast<const unsigned long long*>(data)[i]);
      a = (long long)v < 0 ? 0ULL - v : v;
    } else {
      double d;
      if (val_type == PG_V_F32) d = (double)__uint_as_float(__builtin_bswap32(reinterpret_cast<const uint32_t*>(data)[i]));
      else d = __longlong_as_double((long long)__builtin_bswap64(reinterpret_cast<const unsigned long long*>(data)[i]));
      a = (unsigned long long)__double_as_longlong(d) & 0x7FFFFFFFFFFFFFFFULL;
      if (a >= 0x7FF0000000000000ULL) { bad = 1; a = 0; }
    }
    mx = a > mx ? a : mx;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(mx, off, 64);
    mx = o > mx ? o : mx;
    bad |= __shfl_xor(bad, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (mx) atomicMax(&out[0], mx);
    if (bad) atomicOr(&out[1], 1ULL);
  }
}

// Smallest and largest value of a raw INT / LONG column (out[0] = min, out[1] = max, both initialised by the caller): the partition
// pipeline packs such a value into a tuple as (value - min) in as many bits as the column's range needs.
extern "C" __global__ void __launch_bounds__(256) pg_column_int_range_kernel(const uint8_t* __restrict__ data, int64_t n, int val_type,
                                                                             long long* __restrict__ out) {
  long long mn = INT64_MAX, mx = INT64_MIN;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long v = val_type == PG_V_I64 ? (long long)__builtin_bswap64(reinterpret_cast<const unsigned long long*>(data)[i])
                                             : (long long)(int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(data)[i]);
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const long long a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0 && mn <= mx) {
    atomicMin(&out[0], mn);
    atomicMax(&out[1], mx);
  }
}

// Largest dictId of a bit-packed forward index (PinotDataBitSet layout: value i at bit i x bits, most significant bit first).  A forward
// index whose width leaves room above the cardinality (7 bits, 100 values) can HOLD dictIds the dictionary does not have — a corrupt or
// mismatched file; the reference would throw ArrayIndexOutOfBoundsException on the first such doc, the kernels here would index past a
// dictionary or an LDS table.  Checked once, at registration: one pass over the column at HBM speed.
extern "C" __global__ void __launch_bounds__(256) pg_column_max_dict_id_kernel(const uint8_t* __restrict__ data, int64_t n, int bits,
                                                                                unsigned int* __restrict__ out) {
  unsigned int mx = 0;
  const uint64_t mask = (1ULL << bits) - 1ULL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t bit = (uint64_t)i * (uint64_t)bits;
    const uint8_t* p = data + (bit >> 3);   // (the buffer is padded by 64 bytes: the 8-byte window stays inside)
    uint64_t w = 0;
    for (int k = 0; k < 8; k++) w = (w << 8) | p[k];
    const unsigned int v = (unsigned int)((w >> (64 - bits - (int)(bit & 7))) & mask);
    mx = v > mx ? v : mx;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned int o = __shfl_xor(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(out, mx);
}

namespace pg {

// every finite |value| < 2^fx_exp, fx_exp a multiple of 16 (so that segments with similar data choose the same scale and their
// fixed-point tables merge)
static int32_t fx_exp_of(double max_abs) {
  if (!(max_abs > 0)) return 0;
  int e = 0;
  (void)std::frexp(max_abs, &e);   // max_abs = f * 2^e, f in [0.5, 1): max_abs < 2^e
  return (int32_t)(16 * (int)std::floor(((double)e + 15.0) / 16.0));
}

static inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
static inline uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
static inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

Column* Segment::find(const char* n) {
  if (!n) return nullptr;
  auto it = columns.find(n);
  return it == columns.end() ? nullptr : it->second.get();
}

static size_t padded_docs(const Segment& seg) { return (size_t)seg.n_tiles * PG_TILE_DOCS; }

// the n bit-packed dictIds c.fwd_dev holds (a single-value column's docs, a multi-value column's entries) against the dictionary's cardinality
// (pg_column_max_dict_id_kernel)
static void check_dict_ids(const Column& c, int64_t n) {
  if (n <= 0 || c.cardinality <= 0 || !(c.bits >= 31 || ((int64_t)1 << c.bits) > (int64_t)c.cardinality)) return;   // no room above the cardinality
  DeviceBuffer out(4, true);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n + 255) / 256));
  hipLaunchKernelGGL(pg_column_max_dict_id_kernel, dim3(grid), dim3(256), 0, 0, c.fwd_dev.as<uint8_t>(), n, c.bits, out.as<unsigned int>());
  PG_HIP(hipGetLastError());
  unsigned int mx = 0;
  PG_HIP(hipMemcpy(&mx, out.ptr, 4, hipMemcpyDeviceToHost));
  if ((int64_t)mx >= (int64_t)c.cardinality)
    fail(PG_ERR_INVALID_ARGUMENT, "forward index of %s holds dictId %u, the dictionary has %d values", c.name.c_str(), mx, c.cardinality);
}

static void upload_fixed_bit(Segment& seg, Column& c, const uint8_t* src, uint64_t len) {
  uint64_t need = ((uint64_t)seg.total_docs * (uint64_t)c.bits + 7) / 8;
  if (len < need) fail(PG_ERR_INVALID_ARGUMENT, "forward index of %s is %llu bytes, need %llu", c.name.c_str(),
                       (unsigned long long)len, (unsigned long long)need);
  size_t alloc = (padded_docs(seg) * (size_t)c.bits + 7) / 8 + 64;  // +64: the kernels read a dword pair past the value
  c.fwd_dev.alloc(alloc, true);
  c.fwd_dev.upload(src, need);
  c.col_kind = PG_COL_FIXED_BIT;
  c.fwd_bytes_logical = need;
  check_dict_ids(c, (int64_t)seg.total_docs);
}

static void parse_inverted_index(Segment& seg, Column& c, const uint8_t* inv, uint64_t len) {
  const int32_t card = c.cardinality;
  const uint64_t off_end = ((uint64_t)card + 1) * 4;
  if (len < off_end) fail(PG_ERR_INVALID_ARGUMENT, "inverted index of %s too short", c.name.c_str());
  const uint64_t first = be32(inv);
  const uint8_t* bitmap_buffer = inv + off_end;
  c.posting_begin.assign((size_t)card + 1, 0);
  std::vector<uint8_t> staging;
  staging.reserve(len);
  const uint32_t max_key = (uint32_t)((padded_docs(seg) + PG_CHUNK_DOCS - 1) / PG_CHUNK_DOCS);
  c.posting_card.assign((size_t)card, 0);
  for (int32_t d = 0; d < card; d++) {
    c.posting_begin[d] = (uint32_t)c.descs_host.size();
    uint64_t off = be32(inv + (uint64_t)d * 4), end = be32(inv + (uint64_t)(d + 1) * 4);
    if (end < off || off < first || (end - first) > len - off_end)
      fail(PG_ERR_INVALID_ARGUMENT, "inverted index of %s: bad offsets for dictId %d", c.name.c_str(), d);
    const uint8_t* blob = bitmap_buffer + (off - first);
    uint64_t blen = end - off;
    if (blen < 8) fail(PG_ERR_INVALID_ARGUMENT, "roaring blob too short (%s dictId %d)", c.name.c_str(), d);
    uint32_t cookie = le32(blob);
    uint64_t pos = 4;
    uint32_t size;
    const uint8_t* run_flags = nullptr;
    bool has_run = false;
    if ((cookie & 0xFFFF) == 12347) {
      size = (cookie >> 16) + 1;
      run_flags = blob + pos;
      pos += (size + 7) / 8;
      has_run = true;
    } else if (cookie == 12346) {
      size = le32(blob + pos);
      pos += 4;
    } else {
      fail(PG_ERR_INVALID_ARGUMENT, "roaring: bad cookie %u (%s dictId %d)", cookie, c.name.c_str(), d);
    }
    if (pos + 4ULL * size > blen) fail(PG_ERR_INVALID_ARGUMENT, "roaring: truncated header");
    const uint8_t* desc = blob + pos;
    pos += 4ULL * size;
    if (!has_run || size >= 4) pos += 4ULL * size;
    for (uint32_t i = 0; i < size; i++) {
      PgContainer pc{};
      pc.key = le16(desc + 4 * i);
      uint32_t cardm1 = le16(desc + 4 * i + 2);
      bool is_run = has_run && ((run_flags[i >> 3] >> (i & 7)) & 1);
      uint64_t payload;
      if (is_run) {
        if (pos + 2 > blen) fail(PG_ERR_INVALID_ARGUMENT, "roaring: truncated run container");
        pc.n = le16(blob + pos);
        pos += 2;
        pc.type = 2;
        payload = 4ULL * pc.n;
      } else if (cardm1 + 1 > 4096) {
        pc.type = 1;
        pc.n = cardm1 + 1;
        payload = 8192;
      } else {
        pc.type = 0;
        pc.n = cardm1 + 1;
        payload = 2ULL * pc.n;
      }
      if (pos + payload > blen) fail(PG_ERR_INVALID_ARGUMENT, "roaring: truncated container");
      if (pc.type == 2) for (uint32_t r = 0; r < pc.n; r++) c.posting_card[(size_t)d] += (int64_t)le16(blob + pos + 4 * r + 2) + 1;
      else c.posting_card[(size_t)d] += pc.n;
      if (pc.key >= max_key) fail(PG_ERR_INVALID_ARGUMENT, "roaring: container key %u beyond the segment", pc.key);
      size_t aligned = (staging.size() + 15) & ~(size_t)15;
      staging.resize(aligned + payload);
      memcpy(staging.data() + aligned, blob + pos, payload);
      pc.offset = aligned;
      pos += payload;
      c.descs_host.push_back(pc);
    }
  }
  c.posting_begin[card] = (uint32_t)c.descs_host.size();
  staging.resize(((staging.size() + 15) & ~(size_t)15) + 16);
  c.containers_dev.alloc(staging.size());
  c.containers_dev.upload(staging.data(), staging.size());
  c.descs_dev = upload_vector(c.descs_host);
  c.has_inverted = true;
}

// One RoaringBitmap of docIds → a column-less inverted index with a single posting list ("dictId 0"): the filter kernels read
// it through the same posting leaf as BitmapInvertedIndexReader's bitmaps (array / bitmap / run containers decoded on the GPU).
static std::shared_ptr<Column> bitmap_column(Segment& seg, const std::string& name, const void* roaring, uint64_t size) {
  if (!roaring || size < 8) fail(PG_ERR_INVALID_ARGUMENT, "%s: not a serialized RoaringBitmap", name.c_str());
  if (size > 0xFFFFFFF0ULL) fail(PG_ERR_UNSUPPORTED, "%s: bitmap larger than 4 GB", name.c_str());
  auto col = std::make_shared<Column>();
  col->name = name;
  col->cardinality = 1;
  std::vector<uint8_t> inv(8 + size);
  const uint32_t offs[2] = {8, (uint32_t)(8 + size)};
  for (int i = 0; i < 2; i++) for (int b = 0; b < 4; b++) inv[(size_t)i * 4 + b] = (uint8_t)(offs[i] >> (24 - 8 * b));
  memcpy(inv.data() + 8, roaring, size);
  parse_inverted_index(seg, *col, inv.data(), inv.size());
  return col;
}

void segment_set_null_vector(Segment& seg, const char* column, const void* roaring, uint64_t size) {
  if (!column || !seg.find(column)) fail(PG_ERR_NOT_FOUND, "column not found: %s", column ? column : "(null)");
  auto col = bitmap_column(seg, std::string("nullvalue_vector of ") + column, roaring, size);
  std::lock_guard<std::mutex> g(seg.mu);
  seg.null_vectors[column] = std::move(col);
  seg.plan_cache.clear();   // running queries keep their plan, which pins the bitmap it reads
}

// DataSource#getRangeIndex: BitSlicedRangeIndexReader bytes (big-endian int version 2, long min) + a RoaringBitmap RangeBitmap
// (little-endian: u16 cookie 0xF00D, u8 base 2, u8 sliceCount, u16 maxKey, u32 maxRid, maxKey masks of (sliceCount + 7) / 8 bytes, then
// per chunk and present slice u8 type (0 bitmap / 1 run / 2 array) + container).  RoaringBitmap is not in the reference tree: the
// format is restated from its published source (pinot_amd/formats.py holds the writer the tests use).
void segment_set_range_index(Segment& seg, const char* column, const void* bytes, uint64_t size) {
  Column* c = column ? seg.find(column) : nullptr;
  if (!c) fail(PG_ERR_NOT_FOUND, "column not found: %s", column ? column : "(null)");
  const uint8_t* p = static_cast<const uint8_t*>(bytes);
  if (!p || size < 22) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s too short", column);
  const int32_t version = (int32_t)be32(p);
  if (version != 2) fail(PG_ERR_UNSUPPORTED, "range index of %s: version %d (only the exact bit-sliced index, version 2, is on the GPU path)", column, version);
  const int64_t min = (int64_t)be64(p + 4);
  const uint8_t* r = p + 12;
  if (le16(r) != 0xF00D || r[2] != 2) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: bad RangeBitmap cookie / base", column);
  const int slices = r[3];
  const uint32_t max_key = le16(r + 4), max_rid = le32(r + 6);
  const int bytes_per_mask = (slices + 7) >> 3;
  if (slices < 1 || slices > 64) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: %d slices", column, slices);
  if ((int64_t)max_rid < (int64_t)seg.total_docs) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s covers %u rows, the segment has %d", column, max_rid, seg.total_docs);
  const uint32_t n_chunks = (uint32_t)((padded_docs(seg) + PG_CHUNK_DOCS - 1) / PG_CHUNK_DOCS);
  if (max_key > n_chunks) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: %u chunks beyond the segment", column, max_key);
  uint64_t pos = 22 + (uint64_t)max_key * (uint64_t)bytes_per_mask;
  if (pos > size) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: truncated masks", column);
  std::vector<PgContainer> descs((size_t)n_chunks * (size_t)slices);
  for (auto& d : descs) { d.offset = 0; d.n = 0; d.key = 0; d.type = 3; }
  std::vector<uint8_t> staging;
  staging.reserve(size);
  for (uint32_t key = 0; key < max_key; key++) {
    uint64_t mask = 0;
    for (int b = 0; b < bytes_per_mask; b++) mask |= (uint64_t)p[22 + (uint64_t)key * bytes_per_mask + b] << (8 * b);
    for (int s = 0; s < slices; s++) {
      if (!((mask >> s) & 1)) continue;
      if (pos + 1 > size) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: truncated container", column);
      const int type = p[pos++];
      PgContainer pc{};
      pc.key = (uint16_t)key;
      uint64_t payload;
      if (type == 0) { pc.type = 1; pc.n = 0; payload = 8192; }
      else if (type == 1 || type == 2) {
        if (pos + 2 > size) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: truncated container", column);
        pc.n = le16(p + pos);
        pos += 2;
        pc.type = type == 1 ? 2 : 0;
        payload = type == 1 ? 4ULL * pc.n : 2ULL * pc.n;
      } else {
        fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: container type %d", column, type);
      }
      if (pos + payload > size) fail(PG_ERR_INVALID_ARGUMENT, "range index of %s: truncated container", column);
      const size_t aligned = (staging.size() + 15) & ~(size_t)15;
      staging.resize(aligned + payload);
      memcpy(staging.data() + aligned, p + pos, payload);
      pc.offset = aligned;
      pos += payload;
      descs[(size_t)key * (size_t)slices + (size_t)s] = pc;
    }
  }
  staging.resize(((staging.size() + 15) & ~(size_t)15) + 16);
  DeviceBuffer cont(staging.size());
  cont.upload(staging.data(), staging.size());
  DeviceBuffer dd = upload_vector(descs);
  std::lock_guard<std::mutex> g(seg.mu);
  seg.device_bytes += cont.size + dd.size;
  c->ri_containers_dev = std::move(cont);
  c->ri_descs_dev = std::move(dd);
  c->ri_slices = slices;
  c->ri_chunks = (int32_t)n_chunks;
  c->ri_min = min;
  c->ri_bytes = size;
  c->has_range_index = true;
  seg.plan_cache.clear();
}

void segment_set_queryable_doc_ids(Segment& seg, const void* roaring, uint64_t size) {
  std::shared_ptr<Column> col;
  if (size) col = bitmap_column(seg, "queryableDocIds", roaring, size);
  std::lock_guard<std::mutex> g(seg.mu);
  seg.queryable_doc_ids = std::move(col);
  seg.plan_cache.clear();
}

// VarByteChunkForwardIndexWriter's layout (writer versions 2 / 3; readers VarByteChunkSVForwardIndexReader.java:158-217,
// FixedByteChunkMVForwardIndexReader.java:104-140): 7 big-endian ints {version, numChunks, numDocsPerChunk, lengthOfLongestEntry, totalDocs,
// compressionType, dataHeaderStart}, the chunks' file offsets (int for version 2, long for 3), and per chunk numDocsPerChunk big-endian
// int offsets relative to the chunk start (0 for the absent rows of the last chunk) followed by the values.  Calls `each(doc, ptr, len)`
// for every doc; compressed chunks are decoded on the host (pg_host_codecs.cpp).
template <typename F>
static void walk_var_byte_chunks(const Segment& seg, const uint8_t* fwd, uint64_t fwd_len, const char* name, bool pass_through_only, F each) {
  if (fwd_len < 28) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s too short", name);
  const int32_t version = (int32_t)be32(fwd), num_chunks = (int32_t)be32(fwd + 4), per_chunk = (int32_t)be32(fwd + 8);
  if (version != 2 && version != 3) fail(PG_ERR_UNSUPPORTED, "column %s: var-byte chunk writer version %d", name, version);
  const int32_t longest = (int32_t)be32(fwd + 12), compression = (int32_t)be32(fwd + 20), header_start = (int32_t)be32(fwd + 24);
  if (compression != 0 && pass_through_only)
    fail(PG_ERR_UNSUPPORTED, "column %s: compressed var-byte chunks (type %d) are outside the GPU path", name, compression);
  const int off_size = version == 2 ? 4 : 8;
  if (per_chunk <= 0 || num_chunks < 0 || longest < 0 || (int64_t)num_chunks * per_chunk < seg.total_docs || header_start < 28 ||
      (uint64_t)header_start + (uint64_t)num_chunks * (uint64_t)off_size > fwd_len)
    fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: %d chunks of %d docs for %d docs", name, num_chunks, per_chunk, seg.total_docs);
  const uint64_t max_chunk = (uint64_t)per_chunk * (4ULL + (uint64_t)longest);   // BaseChunkForwardIndexWriter's chunkSize
  for (int32_t ch = 0; ch < num_chunks && (int64_t)ch * per_chunk < seg.total_docs; ch++) {
    auto chunk_pos = [&](int32_t i) -> uint64_t {
      if (i == num_chunks) return fwd_len;
      const uint8_t* o = fwd + header_start + (uint64_t)i * (uint64_t)off_size;
      return off_size == 4 ? (uint64_t)be32(o) : be64(o);
    };
    const uint64_t start = chunk_pos(ch), end = chunk_pos(ch + 1);
    if (start > end || end > fwd_len) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: bad chunk offsets", name);
    std::vector<uint8_t> plain;
    const uint8_t* cb = fwd + start;
    uint64_t clen = end - start;
    if (compression != 0) {
      plain = host_decompress_chunk(compression, cb, clen, max_chunk, name);
      cb = plain.data();
      clen = plain.size();
    }
    if (clen < (uint64_t)per_chunk * 4) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: chunk %d is %llu bytes", name, ch, (unsigned long long)clen);
    const int32_t rows = (int32_t)std::min<int64_t>(per_chunk, (int64_t)seg.total_docs - (int64_t)ch * per_chunk);
    for (int32_t r = 0; r < rows; r++) {
      const uint64_t vs = be32(cb + (size_t)r * 4);
      uint64_t ve = clen;                                       // getValueEndOffset: the last row, or a following absent row (offset 0)
      if (r + 1 < per_chunk) { const uint64_t nx = be32(cb + (size_t)(r + 1) * 4); if (nx != 0) ve = nx; }
      if (vs < (uint64_t)per_chunk * 4 || ve < vs || ve > clen) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: bad value offsets in chunk %d", name, ch);
      each((int64_t)ch * per_chunk + r, cb + vs, ve - vs);
    }
  }
}

void segment_add_column(Segment& seg, const pg_column_desc& d) {
  if (!d.name) fail(PG_ERR_INVALID_ARGUMENT, "column name is null");
  if (seg.columns.count(d.name)) fail(PG_ERR_INVALID_ARGUMENT, "column %s already added", d.name);
  auto col = std::make_unique<Column>();
  Column& c = *col;
  c.name = d.name;
  c.data_type = d.data_type;
  c.fwd_encoding = d.fwd_encoding;
  c.has_dictionary = d.has_dictionary != 0;
  c.cardinality = d.cardinality;
  c.bits = d.bits_per_value;
  c.is_sorted = d.is_sorted != 0;
  c.dict_bytes_per_value = d.dict_bytes_per_value;
  const uint8_t* fwd = (const uint8_t*)d.forward_index.addr;
  const uint64_t fwd_len = d.forward_index.size;
  if (!fwd && seg.total_docs > 0) fail(PG_ERR_INVALID_ARGUMENT, "column %s has no forward index", d.name);

  switch (c.data_type) {
    case PG_TYPE_INT: c.val_type = PG_V_I32; break;
    case PG_TYPE_LONG: c.val_type = PG_V_I64; break;
    case PG_TYPE_FLOAT: c.val_type = PG_V_F32; break;
    case PG_TYPE_DOUBLE: c.val_type = PG_V_F64; break;
    default: c.val_type = PG_V_I32; break;
  }

  if (c.has_dictionary) {
    if (c.cardinality <= 0) fail(PG_ERR_INVALID_ARGUMENT, "column %s: dictionary with cardinality %d", d.name, c.cardinality);
    const uint8_t* dp = (const uint8_t*)d.dictionary.addr;
    uint64_t dict_size = d.dictionary.size;
    // A variable-length STRING dictionary (useVarLengthDictionary: VarLengthValueWriter.java:78-108 / VarLengthValueReader.java — ".vl;", int
    // version 1, int numValues, int dataSectionStartOffset, numValues + 1 absolute int offsets, the values; the reference recognises it by
    // the same magic, BaseImmutableDictionary.java:58-66) becomes the fixed-width zero-padded form here, once: everything downstream
    // (binary search of predicate values, group key decode, data tables) reads padded entries.
    std::vector<uint8_t> padded;
    if (c.data_type == PG_TYPE_STRING && dp && dict_size >= 20 && !memcmp(dp, ".vl;", 4) && be32(dp + 4) == 1) {
      const uint32_t n = be32(dp + 8), start = be32(dp + 12);
      if ((int64_t)n != (int64_t)c.cardinality || (uint64_t)start + ((uint64_t)n + 1) * 4 > dict_size)
        fail(PG_ERR_INVALID_ARGUMENT, "variable-length dictionary of %s: %u values at %u in %llu bytes, cardinality %d", d.name, n, start, (unsigned long long)dict_size, c.cardinality);
      size_t width = 1;
      for (uint32_t i = 0; i < n; i++) {
        const uint64_t a = be32(dp + start + (size_t)i * 4), b = be32(dp + start + (size_t)(i + 1) * 4);
        if (b < a || b > dict_size) fail(PG_ERR_INVALID_ARGUMENT, "variable-length dictionary of %s: bad offsets of value %u", d.name, i);
        if (b > a && dp[b - 1] == 0) fail(PG_ERR_UNSUPPORTED, "column %s: a dictionary value ending in a zero byte cannot be padded", d.name);
        width = std::max(width, (size_t)(b - a));
      }
      padded.assign((size_t)n * width, 0);
      for (uint32_t i = 0; i < n; i++) {
        const uint64_t a = be32(dp + start + (size_t)i * 4), b = be32(dp + start + (size_t)(i + 1) * 4);
        memcpy(padded.data() + (size_t)i * width, dp + a, (size_t)(b - a));
      }
      dp = padded.data();
      dict_size = padded.size();
      c.dict_bytes_per_value = (int32_t)width;
    }
    uint64_t need = (uint64_t)c.cardinality * (uint64_t)c.dict_bytes_per_value;
    if (dict_size != need)   // BaseImmutableDictionary.java:52-55 "Buffer size mismatch"
      fail(PG_ERR_INVALID_ARGUMENT, "Buffer size mismatch: bufferSize = %llu, numValues = %d, numByesPerValue = %d",
           (unsigned long long)dict_size, c.cardinality, c.dict_bytes_per_value);
    c.dict_host.assign(dp, dp + need);
    {
      uint64_t h = 1469598103934665603ULL;
      for (size_t i = 0; i < need; i++) { h ^= dp[i]; h *= 1099511628211ULL; }
      c.dict_hash = h ^ ((uint64_t)c.data_type << 56) ^ (uint64_t)(uint32_t)c.cardinality;
    }
    // native-endian copy for dictionary-encoded metric / value lookups on the device
    if (c.data_type == PG_TYPE_INT || c.data_type == PG_TYPE_FLOAT) {
      std::vector<uint32_t> v((size_t)c.cardinality);
      for (int32_t i = 0; i < c.cardinality; i++) v[i] = be32(dp + (size_t)i * 4);
      c.dict_dev = upload_vector(v);
    } else if (c.data_type == PG_TYPE_LONG || c.data_type == PG_TYPE_DOUBLE) {
      std::vector<uint64_t> v((size_t)c.cardinality);
      for (int32_t i = 0; i < c.cardinality; i++) v[i] = be64(dp + (size_t)i * 8);
      c.dict_dev = upload_vector(v);
    }
  }

  if (c.has_dictionary && (c.data_type == PG_TYPE_INT || c.data_type == PG_TYPE_LONG) && c.cardinality >= 2) {
    // arithmetic dictionary?  (value = base + step x dictId lets kernels compute values instead of gathering them)
    const uint8_t* dp = c.dict_host.data();
    auto at = [&](int32_t i) -> int64_t { return c.data_type == PG_TYPE_INT ? (int64_t)(int32_t)be32(dp + (size_t)i * 4) : (int64_t)be64(dp + (size_t)i * 8); };
    const int64_t base = at(0);
    const __int128 step = (__int128)at(1) - base;
    bool affine = step > 0 && step < ((__int128)1 << 40);
    for (int32_t i = 2; i < c.cardinality && affine; i++) affine = (__int128)at(i) - at(i - 1) == step;
    if (affine) { c.dict_affine = true; c.dict_base = base; c.dict_step = (int64_t)step; }
  }

  if (c.fwd_encoding == PG_FWD_DICT_FIXED_BIT) {
    if (c.bits < 1 || c.bits > 31) fail(PG_ERR_INVALID_ARGUMENT, "column %s: bits_per_value %d", d.name, c.bits);
    if (!c.has_dictionary || c.cardinality <= 0) fail(PG_ERR_INVALID_ARGUMENT, "column %s: a fixed-bit forward index needs a dictionary (cardinality %d)", d.name, c.cardinality);
    if (c.bits < 31 && ((int64_t)1 << c.bits) < (int64_t)c.cardinality)
      fail(PG_ERR_INVALID_ARGUMENT, "column %s: %d bits per value cannot hold %d dictIds", d.name, c.bits, c.cardinality);
    upload_fixed_bit(seg, c, fwd, fwd_len);
  } else if (c.fwd_encoding == PG_FWD_DICT_FIXED_BIT_MV) {
    // FixedBitMVForwardIndexReader.java:57-76: chunk offset header | row-start bitmap (MSB first, one set bit per doc) | bit-packed dictIds.
    // The header only accelerates the reader's per-doc walk; here every doc's first entry is expanded once from the bitmap.
    if (c.bits < 1 || c.bits > 31) fail(PG_ERR_INVALID_ARGUMENT, "column %s: bits_per_value %d", d.name, c.bits);
    if (!c.has_dictionary || c.cardinality <= 0) fail(PG_ERR_UNSUPPORTED, "column %s: raw multi-value columns are outside the hot path", d.name);
    // ForwardIndexReaderFactory.java:82-86 looks for FixedBitMVEntryDictForwardIndexReader's marker FIRST (the MV_ENTRY_DICT format:
    // per-doc ids into a dictionary of distinct entry lists) — a different layout, left to the Java plan (refused, not misparsed)
    if (fwd_len > 4 && be32(fwd) == 0xffabcdefu) {
      // FixedBitMVEntryDictForwardIndexReader.java (written by FixedBitMVEntryDictForwardIndexWriter.java:80-130): 24-byte header — magic,
      // short version 1, byte bitsPerValue, byte bitsPerId, int uniqueEntries, int totalValues, int offsetBufferOffset, int valueBufferOffset —
      // then three MSB-first fixed-bit arrays: the docs' entry ids, the entries' start offsets (unique + 1), the entries' dictIds.  Expanded
      // once, here, into what the kernels read for every multi-value column: the docs' dictIds back to back + every doc's first entry.
      if (fwd_len < 24 || (be32(fwd + 4) >> 16) != 1u) fail(PG_ERR_UNSUPPORTED, "column %s: MV_ENTRY_DICT forward index version %u", d.name, fwd_len < 24 ? 0u : be32(fwd + 4) >> 16);
      const int bits_v = fwd[6], bits_id = fwd[7];
      const int64_t n_unique = (int32_t)be32(fwd + 8), n_total = (int32_t)be32(fwd + 12), off_at = (int32_t)be32(fwd + 16), val_at = (int32_t)be32(fwd + 20);
      int bits_off = 1;
      while (bits_off < 31 && ((int64_t)1 << bits_off) <= n_total) bits_off++;   // PinotDataBitSet.getNumBitsPerValue(numTotalValues)
      const int64_t nd = seg.total_docs;
      if (bits_v != c.bits || bits_id < 1 || bits_id > 31 || n_unique <= 0 || n_total < n_unique || off_at != 24 + (nd * bits_id + 7) / 8 ||
          val_at != off_at + ((n_unique + 1) * bits_off + 7) / 8 || (uint64_t)val_at + (uint64_t)((n_total * bits_v + 7) / 8) > fwd_len)
        fail(PG_ERR_INVALID_ARGUMENT, "MV_ENTRY_DICT forward index of %s: inconsistent header (%d / %d bits, %lld entries, %lld values)", d.name, bits_v, bits_id,
             (long long)n_unique, (long long)n_total);
      auto read_bits = [](const uint8_t* base, int64_t index, int bits) -> uint32_t {   // PinotDataBitSet#readInt
        uint32_t v = 0;
        const int64_t bit0 = index * bits;
        for (int b = 0; b < bits; b++) { const int64_t at = bit0 + b; v = (v << 1) | ((base[at >> 3] >> (7 - (at & 7))) & 1u); }
        return v;
      };
      std::vector<int32_t>& off = c.mv_offsets_host;
      off.reserve((size_t)nd + 1);
      int64_t total = 0;
      for (int64_t doc = 0; doc < nd; doc++) {
        const uint32_t id = read_bits(fwd + 24, doc, bits_id);
        if ((int64_t)id >= n_unique) fail(PG_ERR_INVALID_ARGUMENT, "MV_ENTRY_DICT forward index of %s: entry id %u of %lld", d.name, id, (long long)n_unique);
        const int64_t a = read_bits(fwd + off_at, id, bits_off), b = read_bits(fwd + off_at, (int64_t)id + 1, bits_off);
        if (b <= a || b > n_total) fail(PG_ERR_INVALID_ARGUMENT, "MV_ENTRY_DICT forward index of %s: entry %u spans [%lld, %lld)", d.name, id, (long long)a, (long long)b);
        off.push_back((int32_t)total);
        total += b - a;
        if (total > 0x7FFFFFFF) fail(PG_ERR_UNSUPPORTED, "multi-value column %s: more than 2^31 entries", d.name);
        c.max_entries_per_doc = std::max(c.max_entries_per_doc, (int32_t)(b - a));
      }
      off.push_back((int32_t)total);
      if (d.total_number_of_entries > 0 && total != d.total_number_of_entries)
        fail(PG_ERR_INVALID_ARGUMENT, "MV_ENTRY_DICT forward index of %s expands to %lld entries, the metadata says %d", d.name, (long long)total, d.total_number_of_entries);
      std::vector<uint8_t> stream((size_t)((total * c.bits + 7) / 8) + 8, 0);
      int64_t e = 0;
      for (int64_t doc = 0; doc < nd; doc++) {
        const uint32_t id = read_bits(fwd + 24, doc, bits_id);
        const int64_t a = read_bits(fwd + off_at, id, bits_off), b = read_bits(fwd + off_at, (int64_t)id + 1, bits_off);
        for (int64_t k = a; k < b; k++, e++) {
          const uint32_t v = read_bits(fwd + val_at, k, bits_v);
          if ((int64_t)v >= c.cardinality) fail(PG_ERR_INVALID_ARGUMENT, "MV_ENTRY_DICT forward index of %s: dictId %u of %d", d.name, v, c.cardinality);
          const int64_t bit0 = e * c.bits;
          for (int bb = 0; bb < c.bits; bb++)
            if ((v >> (c.bits - 1 - bb)) & 1u) stream[(size_t)((bit0 + bb) >> 3)] |= (uint8_t)(0x80u >> ((bit0 + bb) & 7));
        }
      }
      const uint64_t raw_bytes = ((uint64_t)total * (uint64_t)c.bits + 7) / 8;
      c.is_mv = true;
      c.total_entries = (int32_t)total;
      c.fwd_dev.alloc((size_t)raw_bytes + 64, true);
      c.fwd_dev.upload(stream.data(), raw_bytes);
      c.mv_offsets_dev = upload_vector(off);
      c.col_kind = PG_COL_FIXED_BIT;
      c.fwd_bytes_logical = fwd_len;
    } else {
    const int64_t num_docs = seg.total_docs, num_values = d.total_number_of_entries;
    if (num_docs <= 0 || num_values < num_docs) fail(PG_ERR_INVALID_ARGUMENT, "multi-value column %s: %lld entries over %lld docs", d.name, (long long)num_values, (long long)num_docs);
    const int64_t per_chunk = (int64_t)std::ceil((float)2048 / (float)(num_values / num_docs));   // the reader's integer division
    const int64_t num_chunks = (num_docs + per_chunk - 1) / per_chunk;
    const uint64_t bitmap_bytes = ((uint64_t)num_values + 7) / 8, raw_bytes = ((uint64_t)num_values * (uint64_t)c.bits + 7) / 8;
    const uint64_t need = (uint64_t)num_chunks * 4 + bitmap_bytes + raw_bytes;
    if (fwd_len < need) fail(PG_ERR_INVALID_ARGUMENT, "multi-value forward index of %s is %llu bytes, need %llu", d.name, (unsigned long long)fwd_len, (unsigned long long)need);
    const uint8_t* bitmap = fwd + (uint64_t)num_chunks * 4;
    std::vector<int32_t>& off = c.mv_offsets_host;
    off.reserve((size_t)num_docs + 1);
    for (int64_t byte = 0; byte < (int64_t)bitmap_bytes; byte++) {
      uint32_t b = bitmap[byte];
      while (b) {
        const int lead = __builtin_clz(b) - 24;   // MSB first
        const int64_t pos = byte * 8 + lead;
        if (pos < num_values) off.push_back((int32_t)pos);
        b &= ~(0x80u >> lead);
      }
    }
    if ((int64_t)off.size() != num_docs || off[0] != 0)
      fail(PG_ERR_INVALID_ARGUMENT, "multi-value forward index of %s: %zu row starts for %lld docs", d.name, off.size(), (long long)num_docs);
    off.push_back((int32_t)num_values);
    for (int64_t i = 0; i < num_docs; i++) c.max_entries_per_doc = std::max(c.max_entries_per_doc, off[(size_t)i + 1] - off[(size_t)i]);
    // the chunk offset header must agree with the bitmap (both come from a file)
    for (int64_t ch = 0; ch < num_chunks; ch++)
      if ((int32_t)be32(fwd + (size_t)ch * 4) != off[(size_t)(ch * per_chunk)])
        fail(PG_ERR_INVALID_ARGUMENT, "multi-value forward index of %s: chunk %lld starts at entry %d, its first doc at %d", d.name, (long long)ch,
             (int32_t)be32(fwd + (size_t)ch * 4), off[(size_t)(ch * per_chunk)]);
    c.is_mv = true;
    c.total_entries = (int32_t)num_values;
    c.fwd_dev.alloc((size_t)raw_bytes + 64, true);   // +64: the kernels read a dword pair past the value
    c.fwd_dev.upload(bitmap + bitmap_bytes, raw_bytes);
    check_dict_ids(c, num_values);   // (the entries index dictionaries, look-up tables and LDS group tables exactly as a single-value column's docs do)
    c.mv_offsets_dev = upload_vector(off);
    c.col_kind = PG_COL_FIXED_BIT;
    c.fwd_bytes_logical = need;
    }
  } else if (c.fwd_encoding == PG_FWD_DICT_SORTED) {
    // SortedIndexReaderImpl: 2 big-endian ints (startDocId, endDocId inclusive) per dictId.  The pairs come from a file: every one is
    // checked (inside the segment, ascending, disjoint) before anything is expanded from them.
    if (!c.has_dictionary || c.cardinality <= 0) fail(PG_ERR_INVALID_ARGUMENT, "sorted index of %s needs a dictionary (cardinality %d)", d.name, c.cardinality);
    if (fwd_len < (uint64_t)c.cardinality * 8) fail(PG_ERR_INVALID_ARGUMENT, "sorted index of %s too short", d.name);
    c.sorted_start.resize((size_t)c.cardinality);
    c.sorted_end.resize((size_t)c.cardinality);
    int64_t prev_end = -1;
    for (int32_t i = 0; i < c.cardinality; i++) {
      const int32_t st = (int32_t)be32(fwd + (size_t)i * 8), en = (int32_t)be32(fwd + (size_t)i * 8 + 4);
      if (st < 0 || en < st || en >= seg.total_docs || (int64_t)st <= prev_end)
        fail(PG_ERR_INVALID_ARGUMENT, "sorted index of %s: dictId %d has the doc range [%d, %d] (segment of %d docs, previous range ends at %lld)",
             d.name, i, st, en, seg.total_docs, (long long)prev_end);
      c.sorted_start[(size_t)i] = st;
      c.sorted_end[(size_t)i] = en;
      prev_end = en;
    }
    // expand to the fixed-bit layout so that projection / group-by see one dictionary-column encoding
    {
      const int32_t mv = c.cardinality - 1;
      const int32_t need_bits = mv <= 1 ? 1 : 32 - __builtin_clz((uint32_t)mv);   // PinotDataBitSet.getNumBitsPerValue(cardinality - 1)
      if (c.bits < 1) c.bits = need_bits;
      if (c.bits > 31 || c.bits < need_bits) fail(PG_ERR_INVALID_ARGUMENT, "column %s: bits_per_value %d cannot hold dictIds up to %d", d.name, c.bits, mv);
    }
    size_t nbytes = ((size_t)seg.total_docs * (size_t)c.bits + 7) / 8;
    std::vector<uint8_t> packed(nbytes + 8, 0);
    for (int32_t id = 0; id < c.cardinality; id++) {
      for (int64_t doc = c.sorted_start[id]; doc <= c.sorted_end[id]; doc++) {
        int64_t bitpos = doc * c.bits;
        for (int b = 0; b < c.bits; b++) {
          if ((id >> (c.bits - 1 - b)) & 1) {
            int64_t bp = bitpos + b;
            packed[(size_t)(bp >> 3)] |= (uint8_t)(0x80 >> (bp & 7));
          }
        }
      }
    }
    upload_fixed_bit(seg, c, packed.data(), nbytes);
    c.fwd_bytes_logical = fwd_len;
  } else if (c.fwd_encoding == PG_FWD_RAW_VAR_BYTE_CHUNK) {
    // VarByteChunkSVForwardIndexReader.java:158-217 (writer versions 2 / 3): per chunk numDocsPerChunk BE int offsets relative to the chunk
    // start (0 for the absent rows of the last chunk), then the values.  The values are re-laid back to back with one int64 offset
    // per doc: what a device kernel can index.  Only a GROUP BY reads such a column (through its virtual dictionary).
    if (c.has_dictionary || (c.data_type != PG_TYPE_STRING && c.data_type != PG_TYPE_BYTES))
      fail(PG_ERR_INVALID_ARGUMENT, "column %s: a var-byte chunk forward index belongs to a raw STRING / BYTES column", d.name);
    if (fwd_len < 28) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s too short", d.name);
    const int32_t version = (int32_t)be32(fwd), num_chunks = (int32_t)be32(fwd + 4), per_chunk = (int32_t)be32(fwd + 8);
    if (version != 2 && version != 3) fail(PG_ERR_UNSUPPORTED, "column %s: var-byte chunk writer version %d", d.name, version);
    const int32_t compression = (int32_t)be32(fwd + 20), header_start = (int32_t)be32(fwd + 24);
    if (compression != 0) fail(PG_ERR_UNSUPPORTED, "column %s: compressed var-byte chunks (type %d) are outside the GPU path", d.name, compression);
    const int off_size = version == 2 ? 4 : 8;
    if (per_chunk <= 0 || num_chunks < 0 || (int64_t)num_chunks * per_chunk < seg.total_docs || header_start < 28 ||
        (uint64_t)header_start + (uint64_t)num_chunks * (uint64_t)off_size > fwd_len)
      fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: %d chunks of %d docs for %d docs", d.name, num_chunks, per_chunk, seg.total_docs);
    std::vector<int64_t> off((size_t)seg.total_docs + 1, 0);
    std::vector<uint8_t> blob;
    blob.reserve((size_t)fwd_len);
    for (int32_t ch = 0; ch < num_chunks && (int64_t)ch * per_chunk < seg.total_docs; ch++) {
      auto chunk_pos = [&](int32_t i) -> uint64_t {
        if (i == num_chunks) return fwd_len;
        const uint8_t* o = fwd + header_start + (uint64_t)i * (uint64_t)off_size;
        return off_size == 4 ? (uint64_t)be32(o) : be64(o);
      };
      const uint64_t start = chunk_pos(ch), end = chunk_pos(ch + 1);
      if (start > end || end > fwd_len || end - start < (uint64_t)per_chunk * 4) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: bad chunk offsets", d.name);
      const uint8_t* cb = fwd + start;
      const uint64_t clen = end - start;
      const int32_t rows = (int32_t)std::min<int64_t>(per_chunk, (int64_t)seg.total_docs - (int64_t)ch * per_chunk);
      for (int32_t r = 0; r < rows; r++) {
        const uint64_t vs = be32(cb + (size_t)r * 4);
        uint64_t ve = clen;                                       // getValueEndOffset: the last row, or a following absent row (offset 0)
        if (r + 1 < per_chunk) { const uint64_t nx = be32(cb + (size_t)(r + 1) * 4); if (nx != 0) ve = nx; }
        if (vs < (uint64_t)per_chunk * 4 || ve < vs || ve > clen) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: bad value offsets in chunk %d", d.name, ch);
        const size_t doc = (size_t)ch * (size_t)per_chunk + (size_t)r;
        off[doc] = (int64_t)blob.size();
        blob.insert(blob.end(), cb + vs, cb + ve);
      }
    }
    off[(size_t)seg.total_docs] = (int64_t)blob.size();
    c.vb_total_bytes = blob.size();
    c.fwd_dev.alloc(blob.size() + 64, true);
    if (!blob.empty()) c.fwd_dev.upload(blob.data(), blob.size());
    c.vb_offsets_dev = upload_vector(off);
    c.col_kind = PG_COL_VAR_BYTES;
    c.fwd_bytes_logical = fwd_len;
  } else if (c.fwd_encoding == PG_FWD_RAW_MV_VAR_BYTE_CHUNK) {
    // VarByteChunkMVForwardIndexReader over STRING values: doc d's value = ArraySerDeUtils.serializeStringArray — int numValues, numValues
    // int lengths, the UTF-8 bytes.  As for the fixed-width form below the column becomes a dictionary-encoded multi-value column once, here:
    // the distinct strings in byte order as a fixed-width, zero-padded STRING dictionary (what SegmentDictionaryCreator writes), ids in
    // FixedBitMVForwardIndexReader's layout; group keys go back as the byte strings (virtual dictionary kind 4).
    if (c.has_dictionary || c.data_type != PG_TYPE_STRING)
      fail(PG_ERR_UNSUPPORTED, "column %s: a raw multi-value var-byte forward index is taken for no-dictionary STRING columns only", d.name);
    const int64_t num_docs = seg.total_docs;
    std::vector<int32_t> starts;
    std::vector<std::string> vals;   // all entries, docs back to back
    walk_var_byte_chunks(seg, fwd, fwd_len, d.name, false, [&](int64_t doc, const uint8_t* p, uint64_t len) {
      (void)doc;
      if (len < 4) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: a value of %llu bytes", d.name, (unsigned long long)len);
      const uint32_t n = be32(p);
      if (n == 0 || 4 + (uint64_t)n * 4 > len) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: %u values in %llu bytes", d.name, n, (unsigned long long)len);
      if (vals.size() + n > 0x7FFFFFFFu) fail(PG_ERR_UNSUPPORTED, "raw multi-value index of %s: more than 2^31 entries", d.name);
      starts.push_back((int32_t)vals.size());
      uint64_t at = 4 + (uint64_t)n * 4;
      for (uint32_t i = 0; i < n; i++) {
        const uint64_t l = be32(p + 4 + (size_t)i * 4);
        if (at + l > len) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: value lengths beyond the doc's %llu bytes", d.name, (unsigned long long)len);
        vals.emplace_back(reinterpret_cast<const char*>(p + at), (size_t)l);
        at += l;
      }
      if (at != len) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: %llu bytes used of %llu", d.name, (unsigned long long)at, (unsigned long long)len);
    });
    if ((int64_t)starts.size() != num_docs || num_docs <= 0) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: %zu docs, the segment has %lld", d.name, starts.size(), (long long)num_docs);
    if (d.total_number_of_entries > 0 && (int64_t)vals.size() != (int64_t)d.total_number_of_entries)
      fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s holds %zu entries, the metadata says %d", d.name, vals.size(), d.total_number_of_entries);
    std::vector<std::string> distinct(vals);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    const int32_t card = (int32_t)distinct.size();
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < (int64_t)card) bits++;
    size_t width = 1;
    for (const std::string& v : distinct) {
      width = std::max(width, v.size());
      if (!v.empty() && v.back() == '\0') fail(PG_ERR_UNSUPPORTED, "column %s: a value ending in a zero byte cannot live in a padded dictionary", d.name);
    }
    std::vector<uint8_t> dict((size_t)card * width, 0);
    std::vector<uint8_t> vbytes;
    std::vector<int64_t> voff((size_t)card + 1, 0);
    for (int32_t i = 0; i < card; i++) {
      memcpy(dict.data() + (size_t)i * width, distinct[(size_t)i].data(), distinct[(size_t)i].size());
      vbytes.insert(vbytes.end(), distinct[(size_t)i].begin(), distinct[(size_t)i].end());
      voff[(size_t)i + 1] = (int64_t)vbytes.size();
    }
    const int64_t num_values = (int64_t)vals.size();
    const int64_t per_chunk = (int64_t)std::ceil((float)2048 / (float)(num_values / num_docs));
    const int64_t num_chunks = (num_docs + per_chunk - 1) / per_chunk;
    const uint64_t bitmap_bytes = ((uint64_t)num_values + 7) / 8, raw_bytes = ((uint64_t)num_values * (uint64_t)bits + 7) / 8;
    std::vector<uint8_t> twin((size_t)num_chunks * 4 + bitmap_bytes + raw_bytes, 0);
    for (int64_t ch = 0; ch < num_chunks; ch++) {
      const uint32_t o = (uint32_t)starts[(size_t)(ch * per_chunk)];
      twin[(size_t)ch * 4] = (uint8_t)(o >> 24); twin[(size_t)ch * 4 + 1] = (uint8_t)(o >> 16); twin[(size_t)ch * 4 + 2] = (uint8_t)(o >> 8); twin[(size_t)ch * 4 + 3] = (uint8_t)o;
    }
    uint8_t* bm = twin.data() + (size_t)num_chunks * 4;
    for (int64_t dd = 0; dd < num_docs; dd++) { const int64_t pos = starts[(size_t)dd]; bm[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7)); }
    uint8_t* packed = bm + bitmap_bytes;
    for (int64_t e = 0; e < num_values; e++) {
      const uint32_t id = (uint32_t)(std::lower_bound(distinct.begin(), distinct.end(), vals[(size_t)e]) - distinct.begin());
      const int64_t bit0 = e * bits;
      for (int b = 0; b < bits; b++)
        if ((id >> (bits - 1 - b)) & 1u) packed[(bit0 + b) >> 3] |= (uint8_t)(0x80u >> ((bit0 + b) & 7));
    }
    const std::string twin_name = std::string(d.name) + "$ids";
    pg_column_desc td{};
    td.name = twin_name.c_str();
    td.data_type = PG_TYPE_STRING;
    td.fwd_encoding = PG_FWD_DICT_FIXED_BIT_MV;
    td.has_dictionary = 1;
    td.cardinality = card;
    td.bits_per_value = bits;
    td.is_sorted = 0;
    td.dict_bytes_per_value = (int32_t)width;
    td.total_number_of_entries = (int32_t)num_values;
    td.forward_index.addr = twin.data();
    td.forward_index.size = twin.size();
    td.dictionary.addr = dict.data();
    td.dictionary.size = dict.size();
    segment_add_column(seg, td);   // (the caller holds the segment's lock: pg_segment_add_column)
    {
      auto it = seg.columns.find(twin_name);
      c.vdict = std::move(it->second);
      seg.columns.erase(it);
    }
    c.vdict->vdict_kind = 4;
    c.vdict->vdict_bytes = std::move(vbytes);
    c.vdict->vdict_bytes_off = std::move(voff);
    c.vdict->vdict_hash = c.vdict->dict_hash;
    c.vdict->public_col = &c;
    c.raw_mv = true;
    c.is_mv = true;
    c.total_entries = (int32_t)num_values;
    c.max_entries_per_doc = c.vdict->max_entries_per_doc;
    c.fwd_bytes_logical = fwd_len;
    c.vdict->fwd_bytes_logical = fwd_len;
  } else if (c.fwd_encoding == PG_FWD_RAW_MV_FIXED_BYTE_CHUNK) {
    // FixedByteChunkMVForwardIndexReader: doc d's value = ArraySerDeUtils.serialize…ArrayWithLength: big-endian int numValues, then the
    // values big-endian.  The column becomes a dictionary-encoded multi-value column here, once: sorted distinct values (what the segment
    // creator would have written as the dictionary: same order, same bytes) and the docs' ids in FixedBitMVForwardIndexReader's layout,
    // registered through the ordinary path as this column's internal twin (c.vdict).
    if (c.has_dictionary || c.data_type > PG_TYPE_DOUBLE)
      fail(PG_ERR_UNSUPPORTED, "column %s: a raw multi-value forward index of fixed-width values belongs to a no-dictionary INT / LONG / FLOAT / DOUBLE column", d.name);
    const int width = (c.data_type == PG_TYPE_INT || c.data_type == PG_TYPE_FLOAT) ? 4 : 8;
    const int64_t num_docs = seg.total_docs;
    std::vector<int32_t> starts;
    starts.reserve((size_t)num_docs + 1);
    std::vector<uint64_t> keys;   // order-preserving 64-bit keys of all entries, docs back to back (pg_vdict.hip's key encoding)
    auto key_of = [&](const uint8_t* p) -> uint64_t {
      if (c.data_type == PG_TYPE_INT) return (uint64_t)(int64_t)(int32_t)be32(p) ^ (1ULL << 63);
      if (c.data_type == PG_TYPE_LONG) return be64(p) ^ (1ULL << 63);
      if (c.data_type == PG_TYPE_FLOAT) { const uint32_t f = be32(p); return (f >> 31) ? (uint64_t)(uint32_t)~f : (uint64_t)(f ^ 0x80000000u); }
      const uint64_t f = be64(p);
      return (f >> 63) ? ~f : (f ^ (1ULL << 63));
    };
    walk_var_byte_chunks(seg, fwd, fwd_len, d.name, false, [&](int64_t doc, const uint8_t* p, uint64_t len) {
      (void)doc;
      if (len < 4) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: a value of %llu bytes", d.name, (unsigned long long)len);
      const uint32_t n = be32(p);
      if ((uint64_t)n * (uint64_t)width + 4 != len || n == 0)   // (an empty array is stored as the default null value: one entry)
        fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: %u values in %llu bytes", d.name, n, (unsigned long long)len);
      if (keys.size() + n > 0x7FFFFFFFu) fail(PG_ERR_UNSUPPORTED, "raw multi-value index of %s: more than 2^31 entries", d.name);
      starts.push_back((int32_t)keys.size());
      for (uint32_t i = 0; i < n; i++) keys.push_back(key_of(p + 4 + (size_t)i * (size_t)width));
    });
    if ((int64_t)starts.size() != num_docs || num_docs <= 0) fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s: %zu docs, the segment has %lld", d.name, starts.size(), (long long)num_docs);
    if (d.total_number_of_entries > 0 && (int64_t)keys.size() != (int64_t)d.total_number_of_entries)
      fail(PG_ERR_INVALID_ARGUMENT, "raw multi-value index of %s holds %zu entries, the metadata says %d", d.name, keys.size(), d.total_number_of_entries);
    std::vector<uint64_t> distinct(keys);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    const int32_t card = (int32_t)distinct.size();
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < (int64_t)card) bits++;
    // the dictionary: the values big-endian in key order (= value order: what SegmentDictionaryCreator writes)
    std::vector<uint8_t> dict((size_t)card * (size_t)width);
    for (int32_t i = 0; i < card; i++) {
      double unused;
      const int64_t v = vdict_value_of_key(distinct[(size_t)i], c.data_type == PG_TYPE_INT ? 0 : c.data_type == PG_TYPE_LONG ? 1 : c.data_type == PG_TYPE_FLOAT ? 2 : 3, &unused);
      uint64_t raw;   // the stored bits
      if (c.data_type == PG_TYPE_FLOAT) { const uint64_t k = distinct[(size_t)i]; const uint32_t kk = (uint32_t)k; raw = (kk >> 31) ? (kk ^ 0x80000000u) : (uint32_t)~kk; }
      else raw = (uint64_t)v;   // INT / LONG: the value; DOUBLE: its IEEE bits
      for (int b = 0; b < width; b++) dict[(size_t)i * (size_t)width + (size_t)b] = (uint8_t)(raw >> (8 * (width - 1 - b)));
    }
    // FixedBitMVForwardIndexReader's layout: chunk offsets | row-start bitmap | bit-packed ids
    const int64_t num_values = (int64_t)keys.size();
    const int64_t per_chunk = (int64_t)std::ceil((float)2048 / (float)(num_values / num_docs));
    const int64_t num_chunks = (num_docs + per_chunk - 1) / per_chunk;
    const uint64_t bitmap_bytes = ((uint64_t)num_values + 7) / 8, raw_bytes = ((uint64_t)num_values * (uint64_t)bits + 7) / 8;
    std::vector<uint8_t> twin((size_t)num_chunks * 4 + bitmap_bytes + raw_bytes, 0);
    for (int64_t ch = 0; ch < num_chunks; ch++) {
      const uint32_t o = (uint32_t)starts[(size_t)(ch * per_chunk)];
      twin[(size_t)ch * 4] = (uint8_t)(o >> 24); twin[(size_t)ch * 4 + 1] = (uint8_t)(o >> 16); twin[(size_t)ch * 4 + 2] = (uint8_t)(o >> 8); twin[(size_t)ch * 4 + 3] = (uint8_t)o;
    }
    uint8_t* bm = twin.data() + (size_t)num_chunks * 4;
    for (int64_t dd = 0; dd < num_docs; dd++) { const int64_t pos = starts[(size_t)dd]; bm[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7)); }
    uint8_t* packed = bm + bitmap_bytes;
    for (int64_t e = 0; e < num_values; e++) {
      const uint32_t id = (uint32_t)(std::lower_bound(distinct.begin(), distinct.end(), keys[(size_t)e]) - distinct.begin());
      const int64_t bit0 = e * bits;
      for (int b = 0; b < bits; b++)
        if ((id >> (bits - 1 - b)) & 1u) packed[(bit0 + b) >> 3] |= (uint8_t)(0x80u >> ((bit0 + b) & 7));
    }
    const std::string twin_name = std::string(d.name) + "$ids";
    pg_column_desc td{};
    td.name = twin_name.c_str();
    td.data_type = c.data_type;
    td.fwd_encoding = PG_FWD_DICT_FIXED_BIT_MV;
    td.has_dictionary = 1;
    td.cardinality = card;
    td.bits_per_value = bits;
    td.is_sorted = 0;
    td.dict_bytes_per_value = width;
    td.total_number_of_entries = (int32_t)num_values;
    td.forward_index.addr = twin.data();
    td.forward_index.size = twin.size();
    td.dictionary.addr = dict.data();
    td.dictionary.size = dict.size();
    segment_add_column(seg, td);   // (the caller holds the segment's lock: pg_segment_add_column)
    {
      auto it = seg.columns.find(twin_name);
      c.vdict = std::move(it->second);
      seg.columns.erase(it);
    }
    c.vdict->vdict_keys = std::move(distinct);
    c.vdict->vdict_kind = c.data_type == PG_TYPE_INT ? 0 : c.data_type == PG_TYPE_LONG ? 1 : c.data_type == PG_TYPE_FLOAT ? 2 : 3;
    c.vdict->vdict_hash = c.vdict->dict_hash;
    c.vdict->public_col = &c;
    c.raw_mv = true;
    c.is_mv = true;
    c.total_entries = (int32_t)num_values;
    c.max_entries_per_doc = c.vdict->max_entries_per_doc;
    c.fwd_bytes_logical = fwd_len;
    c.vdict->fwd_bytes_logical = fwd_len;
  } else if (c.fwd_encoding == PG_FWD_RAW_FIXED_BYTE_CHUNK) {
    if (fwd_len < 16) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s too short", d.name);
    int32_t version = (int32_t)be32(fwd);
    int32_t num_chunks = (int32_t)be32(fwd + 4);
    int32_t entry_len = (int32_t)be32(fwd + 12);
    int32_t data_header_start = 16;
    int32_t compression = 1;  // version 1: SNAPPY
    if (version > 1) {
      if (fwd_len < 28) fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s too short", d.name);
      compression = (int32_t)be32(fwd + 20);
      data_header_start = (int32_t)be32(fwd + 24);
    }
    int width = (c.data_type == PG_TYPE_INT || c.data_type == PG_TYPE_FLOAT) ? 4
                : (c.data_type == PG_TYPE_LONG || c.data_type == PG_TYPE_DOUBLE) ? 8 : 0;
    if (width == 0 || entry_len != width)
      fail(PG_ERR_UNSUPPORTED, "column %s: raw type %d / entry length %d is outside the hot path", d.name, c.data_type, entry_len);
    const int off_size = version <= 2 ? 4 : 8;
    uint64_t raw_start = (uint64_t)data_header_start + (uint64_t)num_chunks * (uint64_t)off_size;
    uint64_t need = (uint64_t)seg.total_docs * (uint64_t)width;
    if (raw_start > fwd_len || (compression == 0 && raw_start + need > fwd_len))
      fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s truncated", d.name);
    c.fwd_dev.alloc(padded_docs(seg) * (size_t)width + 64, true);
    if (compression == 0) {
      c.fwd_dev.upload(fwd + raw_start, need);
    } else {
      // compressed chunks: uploaded as stored, decompressed in HBM (pg_decompress.hip)
      const int32_t docs_per_chunk = (int32_t)be32(fwd + 8);
      if (docs_per_chunk <= 0 || num_chunks < 0 || (int64_t)num_chunks * docs_per_chunk < seg.total_docs)
        fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: %d chunks of %d docs for %d docs", d.name, num_chunks, docs_per_chunk, seg.total_docs);
      std::vector<uint64_t> offs((size_t)num_chunks + 1);
      for (int32_t i = 0; i < num_chunks; i++) {
        const uint8_t* o = fwd + data_header_start + (uint64_t)i * (uint64_t)off_size;
        offs[(size_t)i] = off_size == 4 ? (uint64_t)be32(o) : be64(o);
      }
      offs[(size_t)num_chunks] = fwd_len;
      for (int32_t i = 0; i < num_chunks; i++)
        if (offs[(size_t)i] > offs[(size_t)i + 1] || offs[(size_t)i] < raw_start)
          fail(PG_ERR_INVALID_ARGUMENT, "raw forward index of %s: bad chunk offsets", d.name);
      const int32_t used_chunks = (int32_t)(((int64_t)seg.total_docs + docs_per_chunk - 1) / docs_per_chunk);
      offs.resize((size_t)used_chunks + 1);   // offs[used_chunks]: start of the first unused chunk, or the end of the index
      decompress_fixed_byte_chunks(compression, fwd, offs, (uint32_t)docs_per_chunk * (uint32_t)width, need, c.fwd_dev.as<uint8_t>(), d.name);
    }
    c.col_kind = width == 4 ? PG_COL_RAW32 : PG_COL_RAW64;
    c.fwd_bytes_logical = need;
  } else {
    fail(PG_ERR_UNSUPPORTED, "column %s: forward index encoding %d", d.name, c.fwd_encoding);
  }

  // ---- value statistics for exact SUMs ---------------------------------------------------------------------------------------
  if (c.has_dictionary && (c.data_type == PG_TYPE_LONG || c.data_type == PG_TYPE_FLOAT || c.data_type == PG_TYPE_DOUBLE)) {
    const uint8_t* dp = c.dict_host.data();
    double mx = 0;
    for (int32_t i = 0; i < c.cardinality; i++) {
      if (c.data_type == PG_TYPE_LONG) {
        const int64_t v = (int64_t)be64(dp + (size_t)i * 8);
        const uint64_t a = v < 0 ? 0ULL - (uint64_t)v : (uint64_t)v;
        c.max_abs_int = std::max(c.max_abs_int, a);
      } else {
        double v;
        if (c.data_type == PG_TYPE_FLOAT) { uint32_t u = be32(dp + (size_t)i * 4); float f; memcpy(&f, &u, 4); v = (double)f; }
        else { uint64_t u = be64(dp + (size_t)i * 8); memcpy(&v, &u, 8); }
        if (std::isfinite(v)) mx = std::max(mx, std::fabs(v));
        else c.has_nonfinite = true;
      }
    }
    c.fx_exp = fx_exp_of(mx);
  } else if (!c.has_dictionary && (c.val_type == PG_V_I64 || c.val_type == PG_V_F32 || c.val_type == PG_V_F64) &&
             (c.col_kind == PG_COL_RAW32 || c.col_kind == PG_COL_RAW64) && seg.total_docs > 0) {
    DeviceBuffer out(16, true);
    hipLaunchKernelGGL(pg_column_magnitude_kernel, dim3(1024), dim3(256), 0, 0, c.fwd_dev.as<uint8_t>(), (int64_t)seg.total_docs, c.val_type,
                       out.as<unsigned long long>());
    PG_HIP(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    PG_HIP(hipMemcpy(h, out.ptr, sizeof(h), hipMemcpyDeviceToHost));
    if (c.val_type == PG_V_I64) c.max_abs_int = h[0];
    else {
      double mx;
      memcpy(&mx, &h[0], 8);
      c.fx_exp = fx_exp_of(mx);
      c.has_nonfinite = h[1] != 0;
    }
  }

  if (!c.has_dictionary && (c.val_type == PG_V_I32 || c.val_type == PG_V_I64) && (c.col_kind == PG_COL_RAW32 || c.col_kind == PG_COL_RAW64) &&
      seg.total_docs > 0) {
    const long long init[2] = {INT64_MAX, INT64_MIN};
    DeviceBuffer out(16);
    out.upload(init, sizeof(init));
    hipLaunchKernelGGL(pg_column_int_range_kernel, dim3(1024), dim3(256), 0, 0, c.fwd_dev.as<uint8_t>(), (int64_t)seg.total_docs, c.val_type,
                       out.as<long long>());
    PG_HIP(hipGetLastError());
    long long h[2] = {0, 0};
    PG_HIP(hipMemcpy(h, out.ptr, sizeof(h), hipMemcpyDeviceToHost));
    if (h[0] <= h[1]) { c.has_int_range = true; c.int_min = h[0]; c.int_max = h[1]; }
  }

  if (d.inverted_index.size > 0 && c.has_dictionary && c.fwd_encoding != PG_FWD_DICT_SORTED)
    parse_inverted_index(seg, c, (const uint8_t*)d.inverted_index.addr, d.inverted_index.size);

  seg.device_bytes += c.fwd_dev.size + c.dict_dev.size + c.containers_dev.size + c.descs_dev.size + c.mv_offsets_dev.size + c.vb_offsets_dev.size;
  seg.columns.emplace(c.name, std::move(col));
}

}  // namespace pg
