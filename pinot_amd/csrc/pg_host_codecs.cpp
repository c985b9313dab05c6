// ZSTANDARD and GZIP chunks (ChunkCompressionType 2 and 5) are entropy-coded bit streams (FSE / Huffman): one symbol at a time, no
// wavefront-wide step to map them onto, so — unlike SNAPPY / LZ4 (pg_decompress.hip) — they are decoded on the host, chunk-parallel over
// the host's cores, ONCE at segment registration, and uploaded as the same flat value array every kernel reads.  Nothing on the query
// path sees them.  The decoders are the libraries the reference itself calls through JNI / the JDK:
//   GZIP       java.util.zip.Inflater (zlib stream, RFC 1950) over the chunk minus its last 4 bytes — GzipCompressor appends the
//              uncompressed length as a big-endian int (pinot-segment-local/.../io/compression/GzipCompressor.java:41-51,
//              GzipDecompressor.java:38-56)                                                        → zlib inflate()
//   ZSTANDARD  zstd-jni 1.5.6-9 Zstd.decompress (one zstd frame per chunk: .../ZstandardDecompressor.java:36-45)
//                                                                                                   → ZSTD_decompress of libzstd.so.1,
//              bound at first use with dlopen (the image ships the runtime library without its header).
#include <dlfcn.h>
#include <zlib.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>

#include "pg_internal.hpp"

namespace pg {
namespace {

typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_is_error_fn)(size_t);
struct ZstdLib {
  zstd_decompress_fn decompress = nullptr;
  zstd_is_error_fn is_error = nullptr;
};
const ZstdLib& zstd_lib() {
  static ZstdLib lib;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    lib.decompress = reinterpret_cast<zstd_decompress_fn>(dlsym(h, "ZSTD_decompress"));
    lib.is_error = reinterpret_cast<zstd_is_error_fn>(dlsym(h, "ZSTD_isError"));
  });
  return lib;
}

// one chunk → `want` bytes at dst; false when it does not decode to exactly that
bool inflate_chunk(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t want) {
  if (n < 4) return false;
  const uint64_t stated = ((uint64_t)src[n - 4] << 24) | ((uint64_t)src[n - 3] << 16) | ((uint64_t)src[n - 2] << 8) | (uint64_t)src[n - 1];
  if (stated != want) return false;
  z_stream z;
  std::memset(&z, 0, sizeof z);
  if (inflateInit(&z) != Z_OK) return false;
  z.next_in = const_cast<Bytef*>(src);
  z.avail_in = (uInt)(n - 4);
  z.next_out = dst;
  z.avail_out = (uInt)want;
  const int rc = inflate(&z, Z_FINISH);
  const bool ok = rc == Z_STREAM_END && z.total_out == want;
  inflateEnd(&z);
  return ok;
}

// LZ4 block format (lz4-java's fast decompressor, .../io/compression/LZ4Decompressor.java: one block per chunk, decompressed size known to
// the caller as an upper bound here): token = literal length (high nibble) | match length - 4 (low nibble), 255-continued lengths,
// little-endian 16-bit match offset; the last sequence has literals only.  Returns the bytes produced, -1 on malformed input.
int64_t lz4_block(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) {
  uint64_t ip = 0, op = 0;
  while (ip < n) {
    const uint32_t token = src[ip++];
    uint64_t lit = token >> 4;
    if (lit == 15) { uint8_t b; do { if (ip >= n) return -1; b = src[ip++]; lit += b; } while (b == 255); }
    if (ip + lit > n || op + lit > cap) return -1;
    std::memcpy(dst + op, src + ip, lit);
    ip += lit; op += lit;
    if (ip >= n) break;   // last sequence: literals only
    if (ip + 2 > n) return -1;
    const uint64_t off = (uint64_t)src[ip] | ((uint64_t)src[ip + 1] << 8);
    ip += 2;
    uint64_t len = (token & 15u) + 4;
    if ((token & 15u) == 15) { uint8_t b; do { if (ip >= n) return -1; b = src[ip++]; len += b; } while (b == 255); }
    if (off == 0 || off > op || op + len > cap) return -1;
    for (uint64_t k = 0; k < len; k++) dst[op + k] = dst[op - off + k];   // overlapping copies repeat their window
    op += len;
  }
  return (int64_t)op;
}

}  // namespace

std::vector<uint8_t> host_decompress_chunk(int compression, const uint8_t* src, uint64_t n, uint64_t capacity, const char* column) {
  std::vector<uint8_t> out;
  switch (compression) {
    case 0:   // PASS_THROUGH
      out.assign(src, src + n);
      return out;
    case 3: {   // LZ4
      out.resize(capacity);
      const int64_t got = lz4_block(src, n, out.data(), capacity);
      if (got < 0) fail(PG_ERR_INVALID_ARGUMENT, "column %s: malformed LZ4 chunk", column);
      out.resize((size_t)got);
      return out;
    }
    case 4: {   // LZ4_LENGTH_PREFIXED: little-endian int decompressed length, then the block (LZ4WithLengthDecompressor.java)
      if (n < 4) fail(PG_ERR_INVALID_ARGUMENT, "column %s: LZ4_LENGTH_PREFIXED chunk of %llu bytes", column, (unsigned long long)n);
      const uint64_t want = (uint64_t)src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24);
      if (want > capacity) fail(PG_ERR_INVALID_ARGUMENT, "column %s: chunk states %llu bytes, at most %llu fit", column, (unsigned long long)want, (unsigned long long)capacity);
      out.resize(want);
      if (lz4_block(src + 4, n - 4, out.data(), want) != (int64_t)want) fail(PG_ERR_INVALID_ARGUMENT, "column %s: malformed LZ4 chunk", column);
      return out;
    }
    case 2: {   // ZSTANDARD
      if (!zstd_lib().decompress || !zstd_lib().is_error) fail(PG_ERR_UNSUPPORTED, "column %s: ZSTANDARD chunks need libzstd.so.1, which this host does not have", column);
      out.resize(capacity);
      const size_t got = zstd_lib().decompress(out.data(), capacity, src, n);
      if (zstd_lib().is_error(got)) fail(PG_ERR_INVALID_ARGUMENT, "column %s: malformed ZSTANDARD chunk", column);
      out.resize(got);
      return out;
    }
    case 5: {   // GZIP: zlib stream + the decompressed length as a big-endian int
      if (n < 4) fail(PG_ERR_INVALID_ARGUMENT, "column %s: GZIP chunk of %llu bytes", column, (unsigned long long)n);
      const uint64_t want = ((uint64_t)src[n - 4] << 24) | ((uint64_t)src[n - 3] << 16) | ((uint64_t)src[n - 2] << 8) | (uint64_t)src[n - 1];
      if (want > capacity) fail(PG_ERR_INVALID_ARGUMENT, "column %s: chunk states %llu bytes, at most %llu fit", column, (unsigned long long)want, (unsigned long long)capacity);
      out.resize(want);
      if (!inflate_chunk(src, n, out.data(), want)) fail(PG_ERR_INVALID_ARGUMENT, "column %s: malformed GZIP chunk", column);
      return out;
    }
    default:
      fail(PG_ERR_UNSUPPORTED, "column %s: ChunkCompressionType %d of a var-byte chunk is outside the GPU path", column, compression);
  }
}

bool host_codec(int compression) { return compression == 2 || compression == 5; }

void host_decompress_fixed_byte_chunks(int compression, const uint8_t* file, const std::vector<uint64_t>& offs, uint32_t chunk_bytes,
                                       uint64_t total_bytes, uint8_t* dst_device, const char* column) {
  const int64_t n_chunks = (int64_t)offs.size() - 1;
  if (n_chunks <= 0 || total_bytes == 0) return;
  if (chunk_bytes == 0) fail(PG_ERR_INVALID_ARGUMENT, "column %s: empty chunks", column);
  if (compression == 2 && (!zstd_lib().decompress || !zstd_lib().is_error))
    fail(PG_ERR_UNSUPPORTED, "column %s: ZSTANDARD chunks need libzstd.so.1, which this host does not have", column);
  std::vector<uint8_t> flat(total_bytes);
  std::atomic<int64_t> next{0}, bad{-1};
  auto work = [&] {
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= n_chunks || bad.load() >= 0) return;
      const uint64_t pos = (uint64_t)i * chunk_bytes;
      if (pos >= total_bytes) return;                                   // chunks past the last doc carry nothing
      const uint64_t want = std::min<uint64_t>(chunk_bytes, total_bytes - pos);
      const uint8_t* src = file + offs[(size_t)i];
      const uint64_t n = offs[(size_t)i + 1] - offs[(size_t)i];
      bool ok;
      if (compression == 5) {
        ok = inflate_chunk(src, n, flat.data() + pos, want);
      } else {
        const size_t got = zstd_lib().decompress(flat.data() + pos, want, src, n);
        ok = !zstd_lib().is_error(got) && got == want;
      }
      if (!ok) { int64_t none = -1; bad.compare_exchange_strong(none, i); return; }
    }
  };
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int n_threads = (int)std::min<int64_t>(std::min<unsigned>(hw, 32u), std::max<int64_t>(1, n_chunks / 64));
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; t++) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  if (bad.load() >= 0)
    fail(PG_ERR_INVALID_ARGUMENT, "column %s: chunk %lld does not decompress to its %u bytes", column, (long long)bad.load(), chunk_bytes);
  PG_HIP(hipMemcpy(dst_device, flat.data(), total_bytes, hipMemcpyHostToDevice));
}

}  // namespace pg
