// Synthetic segment generator (host, multi-threaded): writes the `gpuBench` table of BASELINE.md / SURVEY.md §8d directly
// in Pinot's index-entry byte layouts.  Every value is a pure function of (seed, column salt, docId):
//     h = splitmix64(splitmix64(seed ^ salt) + docId);   value = ((h >> 32) * range) >> 32
// so a prefix segment equals the prefix of a bigger one, and pinot_amd/synth.py restates the same function in numpy for
// cross-checking.  This plays the role of the reference's SegmentIndexCreationDriverImpl for synthetic data only.
//   fixed-bit forward index   FixedBitSVForwardIndexWriter.java:42-45 (MSB-first bit stream)
//   raw INT forward index     BaseChunkForwardIndexWriter.java:131-165 (7-int header + chunk offsets + BE values)
//   inverted index            BitmapInvertedIndexWriter.java:36-104 + RoaringBitmap portable format (array / bitmap / run
//                             containers chosen as RoaringBitmapWriter.writer().get() would after runOptimize)
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline uint32_t value_at(uint64_t col_key, uint64_t doc, uint32_t range) {
  uint64_t h = splitmix64(col_key + doc);
  return (uint32_t)(((h >> 32) * (uint64_t)range) >> 32);
}
inline uint64_t col_key_of(uint64_t seed, uint64_t salt) { return splitmix64(seed ^ salt); }

template <typename F>
void parallel_for(int64_t n_items, int n_threads, F&& f) {
  if (n_threads <= 1 || n_items <= 1) {
    for (int64_t i = 0; i < n_items; i++) f(i);
    return;
  }
  std::atomic<int64_t> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back([&] {
      for (;;) {
        int64_t i = next.fetch_add(1);
        if (i >= n_items) break;
        f(i);
      }
    });
  for (auto& x : th) x.join();
}

inline void put_be32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
inline void put_be64(uint8_t* p, uint64_t v) { put_be32(p, (uint32_t)(v >> 32)); put_be32(p + 4, (uint32_t)v); }

}  // namespace

extern "C" {

int pgs_default_threads(void) {
  unsigned n = std::thread::hardware_concurrency();
  return n ? (int)std::min(n, 64u) : 8;
}

// plain values (int32, native endian) — used by tests to cross-check the packed outputs
void pgs_fill_values(int32_t* out, int64_t n, uint64_t seed, uint64_t salt, uint32_t range, int threads) {
  const uint64_t key = col_key_of(seed, salt);
  const int64_t blk = 1 << 20;
  parallel_for((n + blk - 1) / blk, threads, [&](int64_t b) {
    int64_t e = std::min(n, (b + 1) * blk);
    for (int64_t d = b * blk; d < e; d++) out[d] = (int32_t)value_at(key, (uint64_t)d, range);
  });
}

// FixedBitSVForwardIndexWriter layout: `out` must hold ceil(n*bits/8) bytes
void pgs_fill_fixed_bit(uint8_t* out, int64_t n, int bits, uint64_t seed, uint64_t salt, uint32_t range, int threads) {
  const uint64_t key = col_key_of(seed, salt);
  const int64_t blk = 1 << 20;   // multiple of 32 values => every block starts on a 32-bit word boundary
  const int64_t total_bytes = (n * bits + 7) / 8;
  parallel_for((n + blk - 1) / blk, threads, [&](int64_t b) {
    int64_t s = b * blk, e = std::min(n, (b + 1) * blk);
    uint8_t* p = out + (s * bits) / 8;
    uint64_t acc = 0;
    int have = 0;
    for (int64_t d = s; d < e; d++) {
      acc = (acc << bits) | value_at(key, (uint64_t)d, range);
      have += bits;
      while (have >= 8) {
        *p++ = (uint8_t)(acc >> (have - 8));
        have -= 8;
      }
    }
    if (have > 0 && p < out + total_bytes) *p = (uint8_t)(acc << (8 - have));
  });
}

// the same layout for the docs [first, first + n) of the segment, packed from bit 0 of `out` (doc-sharded oracle runs of the full-size tests)
void pgs_fill_fixed_bit_from(uint8_t* out, int64_t first, int64_t n, int bits, uint64_t seed, uint64_t salt, uint32_t range, int threads) {
  const uint64_t key = col_key_of(seed, salt);
  const int64_t blk = 1 << 20;
  const int64_t total_bytes = (n * bits + 7) / 8;
  parallel_for((n + blk - 1) / blk, threads, [&](int64_t b) {
    int64_t s = b * blk, e = std::min(n, (b + 1) * blk);
    uint8_t* p = out + (s * bits) / 8;
    uint64_t acc = 0;
    int have = 0;
    for (int64_t d = s; d < e; d++) {
      acc = (acc << bits) | value_at(key, (uint64_t)(first + d), range);
      have += bits;
      while (have >= 8) {
        *p++ = (uint8_t)(acc >> (have - 8));
        have -= 8;
      }
    }
    if (have > 0 && p < out + total_bytes) *p = (uint8_t)(acc << (8 - have));
  });
}

// raw fixed-byte chunk forward index (PASS_THROUGH).  Returns the total size; fills `out` when non-null.
int64_t pgs_raw_int_index(uint8_t* out, int64_t n, uint64_t seed, uint64_t salt, uint32_t range, int version,
                          int docs_per_chunk, int threads) {
  const int64_t num_chunks = (n + docs_per_chunk - 1) / docs_per_chunk;
  const int off_size = version == 2 ? 4 : 8;
  const int64_t header = 28 + num_chunks * off_size;
  const int64_t total = header + n * 4;
  if (!out) return total;
  put_be32(out, (uint32_t)version);
  put_be32(out + 4, (uint32_t)num_chunks);
  put_be32(out + 8, (uint32_t)docs_per_chunk);
  put_be32(out + 12, 4);
  put_be32(out + 16, (uint32_t)n);
  put_be32(out + 20, 0);   // PASS_THROUGH
  put_be32(out + 24, 28);
  for (int64_t c = 0; c < num_chunks; c++) {
    int64_t off = header + c * (int64_t)docs_per_chunk * 4;
    if (off_size == 4) put_be32(out + 28 + c * 4, (uint32_t)off);
    else put_be64(out + 28 + c * 8, (uint64_t)off);
  }
  const uint64_t key = col_key_of(seed, salt);
  uint8_t* data = out + header;
  const int64_t blk = 1 << 20;
  parallel_for((n + blk - 1) / blk, threads, [&](int64_t b) {
    int64_t e = std::min(n, (b + 1) * blk);
    for (int64_t d = b * blk; d < e; d++) put_be32(data + d * 4, value_at(key, (uint64_t)d, range));
  });
  return total;
}

// ---- inverted index ---------------------------------------------------------------------------------------------------
struct InvBuilder {
  int64_t n;
  uint32_t card;
  uint64_t key;
  int threads;
  int64_t n_chunks;
  std::vector<uint8_t> type;     // [card][n_chunks]: 0 array 1 bitmap 2 run 255 absent
  std::vector<uint32_t> cnt;     // cardinality
  std::vector<uint32_t> nruns;
  std::vector<int64_t> blob_size, blob_off;
};

static void chunk_bitsets(const InvBuilder& B, int64_t chunk, std::vector<uint64_t>& words /* card*1024 */) {
  std::fill(words.begin(), words.end(), 0);
  int64_t s = chunk * 65536, e = std::min(B.n, s + 65536);
  for (int64_t d = s; d < e; d++) {
    uint32_t v = value_at(B.key, (uint64_t)d, B.card);
    int64_t o = d - s;
    words[(size_t)v * 1024 + (size_t)(o >> 6)] |= 1ULL << (o & 63);
  }
}
static void container_stats(const uint64_t* w, uint32_t* card, uint32_t* runs) {
  uint32_t c = 0, r = 0;
  uint64_t prev_msb = 0;
  for (int i = 0; i < 1024; i++) {
    uint64_t x = w[i];
    c += (uint32_t)__builtin_popcountll(x);
    r += (uint32_t)__builtin_popcountll(x & ~((x << 1) | prev_msb));   // run starts
    prev_msb = x >> 63;
  }
  *card = c;
  *runs = r;
}

void* pgs_inverted_begin(int64_t n, uint32_t card, uint64_t seed, uint64_t salt, int threads, int64_t* out_total_size) {
  auto* B = new InvBuilder();
  B->n = n; B->card = card; B->key = col_key_of(seed, salt); B->threads = threads;
  B->n_chunks = (n + 65535) / 65536;
  size_t m = (size_t)card * (size_t)B->n_chunks;
  B->type.assign(m, 255); B->cnt.assign(m, 0); B->nruns.assign(m, 0);
  parallel_for(B->n_chunks, threads, [&](int64_t ch) {
    std::vector<uint64_t> words((size_t)card * 1024);
    chunk_bitsets(*B, ch, words);
    for (uint32_t v = 0; v < card; v++) {
      uint32_t c, r;
      container_stats(words.data() + (size_t)v * 1024, &c, &r);
      size_t i = (size_t)v * B->n_chunks + ch;
      B->cnt[i] = c; B->nruns[i] = r;
      if (c == 0) continue;
      uint32_t size_now = c <= 4096 ? 2 * c : 8192;
      if (2 + 4 * r < size_now) B->type[i] = 2;          // runOptimize
      else B->type[i] = c <= 4096 ? 0 : 1;
    }
  });
  B->blob_size.assign(card, 0); B->blob_off.assign(card + 1, 0);
  int64_t pos = ((int64_t)card + 1) * 4;
  for (uint32_t v = 0; v < card; v++) {
    int64_t size = 0; bool has_run = false; int64_t payload = 0;
    for (int64_t ch = 0; ch < B->n_chunks; ch++) {
      size_t i = (size_t)v * B->n_chunks + ch;
      if (B->type[i] == 255) continue;
      size++;
      if (B->type[i] == 2) { has_run = true; payload += 2 + 4LL * B->nruns[i]; }
      else if (B->type[i] == 1) payload += 8192;
      else payload += 2LL * B->cnt[i];
    }
    int64_t hdr;
    if (size == 0) hdr = 8;
    else if (has_run) hdr = 4 + (size + 7) / 8 + 4 * size + (size >= 4 ? 4 * size : 0);
    else hdr = 8 + 4 * size + 4 * size;
    B->blob_off[v] = pos;
    B->blob_size[v] = hdr + payload;
    pos += hdr + payload;
  }
  B->blob_off[card] = pos;
  *out_total_size = pos;
  return B;
}

void pgs_inverted_fill(void* handle, uint8_t* out) {
  InvBuilder& B = *(InvBuilder*)handle;
  const uint32_t card = B.card;
  for (uint32_t v = 0; v <= card; v++) put_be32(out + (size_t)v * 4, (uint32_t)B.blob_off[v]);
  // headers + per-container payload offsets
  std::vector<int64_t> payload_off((size_t)card * B.n_chunks, -1);
  for (uint32_t v = 0; v < card; v++) {
    uint8_t* blob = out + B.blob_off[v];
    int64_t size = 0; bool has_run = false;
    for (int64_t ch = 0; ch < B.n_chunks; ch++) {
      uint8_t t = B.type[(size_t)v * B.n_chunks + ch];
      if (t == 255) continue;
      size++;
      has_run |= (t == 2);
    }
    auto le16 = [](uint8_t* p, uint32_t x) { p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); };
    auto le32 = [](uint8_t* p, uint32_t x) { p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); p[3] = (uint8_t)(x >> 24); };
    int64_t pos;
    if (size == 0) { le32(blob, 12346); le32(blob + 4, 0); continue; }
    if (has_run) {
      le32(blob, 12347u | ((uint32_t)(size - 1) << 16));
      pos = 4;
      memset(blob + pos, 0, (size_t)((size + 7) / 8));
      int64_t k = 0;
      for (int64_t ch = 0; ch < B.n_chunks; ch++) {
        uint8_t t = B.type[(size_t)v * B.n_chunks + ch];
        if (t == 255) continue;
        if (t == 2) blob[pos + (k >> 3)] |= (uint8_t)(1 << (k & 7));
        k++;
      }
      pos += (size + 7) / 8;
    } else {
      le32(blob, 12346); le32(blob + 4, (uint32_t)size);
      pos = 8;
    }
    for (int64_t ch = 0; ch < B.n_chunks; ch++) {
      size_t i = (size_t)v * B.n_chunks + ch;
      if (B.type[i] == 255) continue;
      le16(blob + pos, (uint32_t)ch);
      le16(blob + pos + 2, B.cnt[i] - 1);
      pos += 4;
    }
    const bool with_offsets = !has_run || size >= 4;
    int64_t off_hdr = pos;
    if (with_offsets) pos += 4 * size;
    int64_t k = 0;
    for (int64_t ch = 0; ch < B.n_chunks; ch++) {
      size_t i = (size_t)v * B.n_chunks + ch;
      if (B.type[i] == 255) continue;
      if (with_offsets) le32(blob + off_hdr + 4 * k, (uint32_t)pos);
      payload_off[i] = B.blob_off[v] + pos;
      pos += B.type[i] == 2 ? 2 + 4LL * B.nruns[i] : (B.type[i] == 1 ? 8192 : 2LL * B.cnt[i]);
      k++;
    }
  }
  parallel_for(B.n_chunks, B.threads, [&](int64_t ch) {
    std::vector<uint64_t> words((size_t)card * 1024);
    chunk_bitsets(B, ch, words);
    for (uint32_t v = 0; v < card; v++) {
      size_t i = (size_t)v * B.n_chunks + ch;
      if (B.type[i] == 255) continue;
      const uint64_t* w = words.data() + (size_t)v * 1024;
      uint8_t* p = out + payload_off[i];
      if (B.type[i] == 1) {
        memcpy(p, w, 8192);   // little-endian host == Roaring's LE words
      } else if (B.type[i] == 0) {
        for (int k = 0; k < 1024; k++) {
          uint64_t x = w[k];
          while (x) {
            uint32_t val = (uint32_t)(k * 64 + __builtin_ctzll(x));
            p[0] = (uint8_t)val; p[1] = (uint8_t)(val >> 8); p += 2;
            x &= x - 1;
          }
        }
      } else {
        uint32_t nr = B.nruns[i];
        p[0] = (uint8_t)nr; p[1] = (uint8_t)(nr >> 8); p += 2;
        int run_start = -1;
        for (int bit = 0; bit <= 65536; bit++) {
          bool set = bit < 65536 && ((w[bit >> 6] >> (bit & 63)) & 1);
          if (set && run_start < 0) run_start = bit;
          if (!set && run_start >= 0) {
            uint32_t len1 = (uint32_t)(bit - 1 - run_start);
            p[0] = (uint8_t)run_start; p[1] = (uint8_t)(run_start >> 8); p[2] = (uint8_t)len1; p[3] = (uint8_t)(len1 >> 8); p += 4;
            run_start = -1;
          }
        }
      }
    }
  });
}

void pgs_inverted_end(void* handle) { delete (InvBuilder*)handle; }

}  // extern "C"
