// Virtual dictionaries for no-dictionary group-by columns.
//
// The reference groups raw columns through on-the-fly maps: NoDictionarySingleColumnGroupKeyGenerator (one value → group id map per
// stored type: Int2Int / Long2Int / Float2Int / Double2Int, pinot-core/.../groupby/NoDictionarySingleColumnGroupKeyGenerator.java:53-90,
// 238-262) and NoDictionaryMultiColumnGroupKeyGenerator (an on-the-fly dictionary per raw column, the tuple of per-column ids → group id,
// .../NoDictionaryMultiColumnGroupKeyGenerator.java:60-130).  What those maps compute is a dictionary encoding of the raw column that
// nobody kept.  Here it is built ONCE per (segment, column) at first use and kept next to the column in HBM: the column's distinct
// values in value order (radix sort + unique over order-preserving 64-bit keys) and a bit-packed id per doc in the layout of a
// dictionary-encoded forward index (FixedBitSVForwardIndexReaderV2), so that every group-by kernel — LDS tables, range-partitioned
// tables, packed radix tuples, hashed 64-bit composite keys — takes a raw key column exactly as it takes a dictionary column, and any
// mix of raw and dictionary columns is one composite key.  Equality follows the fastutil maps: FLOAT / DOUBLE keys compare by
// floatToIntBits / doubleToLongBits (every NaN is one key, -0.0 and 0.0 are two).
//
// Raw STRING / BYTES columns (NoDictionarySingleColumnGroupKeyGenerator.java:132-140: Object2IntOpenHashMap of the values;
// NoDictionaryMultiColumnGroupKeyGenerator's String / Bytes on-the-fly dictionaries) take the same route with a 64-bit HASH of the
// value as the key: equality is all a group key needs, so ids follow hash order, not value order.  A hash is not an identity, so the
// build proves it one: every doc's bytes are compared with the bytes of its id's representative doc (the smallest docId holding the
// hash), and two different values under one hash (≈ n² / 2^65) fail the build (PG_ERR_UNSUPPORTED: the Java plan groups that column)
// instead of merging two groups.  The distinct values themselves (the representatives' bytes, in id order) are copied to the host once
// and handed back as the groups' keys.
//
// Load-path work (one pass to build keys, a device radix sort, one pass to assign ids): rocPRIM provides the sort and the unique.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/device/device_scan.hpp>

#include "pg_internal.hpp"

namespace pg {
namespace {

// order-preserving unsigned keys; kind: 0 INT, 1 LONG, 2 FLOAT, 3 DOUBLE
__device__ __host__ inline uint64_t key_of_bits(uint64_t be_value, int kind) {
  switch (kind) {
    case 0: return (uint64_t)(int64_t)(int32_t)(uint32_t)be_value ^ (1ULL << 63);
    case 1: return be_value ^ (1ULL << 63);
    case 2: {
      uint32_t f = (uint32_t)be_value;
      if ((f & 0x7FFFFFFFu) > 0x7F800000u) f = 0x7FC00000u;          // floatToIntBits: one NaN
      return (uint64_t)(f ^ ((f >> 31) ? 0xFFFFFFFFu : 0x80000000u));
    }
    default: {
      uint64_t d = be_value;
      if ((d & 0x7FFFFFFFFFFFFFFFULL) > 0x7FF0000000000000ULL) d = 0x7FF8000000000000ULL;
      return d ^ ((d >> 63) ? ~0ULL : (1ULL << 63));
    }
  }
}

__global__ void pg_vdict_keys_kernel(const uint8_t* __restrict__ raw, int width, int kind, uint64_t* __restrict__ keys, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t v;
  if (width == 4) v = __builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[i]);
  else v = __builtin_bswap64(reinterpret_cast<const uint64_t*>(raw)[i]);
  keys[i] = key_of_bits(v, kind);
}

// 64-bit hash of a byte string: two FNV-1a lanes over even / odd bytes, mixed (splitmix64 finaliser) with the length
__global__ void pg_vdict_hash_kernel(const uint8_t* __restrict__ blob, const int64_t* __restrict__ off, uint64_t* __restrict__ keys, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = off[i], e = off[i + 1];
  uint64_t a = 1469598103934665603ULL, b = 0x9E3779B97F4A7C15ULL;
  for (int64_t k = s; k < e; k++) {
    const uint64_t x = blob[k];
    a = (a ^ x) * 1099511628211ULL;
    b = (b + x + (b << 6) + (b >> 2)) * 0xFF51AFD7ED558CCDULL;
  }
  uint64_t h = a ^ (b + (uint64_t)(e - s) * 0xC2B2AE3D27D4EB4FULL);
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 27; h *= 0x94D049BB133111EBULL; h ^= h >> 31;
  keys[i] = h;
}
// representative doc of every id: the smallest docId holding it
__global__ void pg_vdict_rep_kernel(const uint32_t* __restrict__ ids, uint32_t* __restrict__ rep, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMin(&rep[ids[i]], (uint32_t)i);
}
// every doc's bytes against its representative's: flag[0] != 0 ⇒ two values share a hash
__global__ void pg_vdict_verify_kernel(const uint8_t* __restrict__ blob, const int64_t* __restrict__ off, const uint32_t* __restrict__ ids,
                                       const uint32_t* __restrict__ rep, uint32_t* __restrict__ flag, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = rep[ids[i]];
  if (r == i) return;
  const int64_t s = off[i], len = off[i + 1] - s, rs = off[r];
  bool same = len == off[r + 1] - rs;
  for (int64_t k = 0; same && k < len; k++) same = blob[s + k] == blob[rs + k];
  if (!same) atomicOr(flag, 1u);
}
__global__ void pg_vdict_value_len_kernel(const int64_t* __restrict__ off, const uint32_t* __restrict__ rep, int64_t* __restrict__ len, uint32_t card) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < card) len[d] = off[rep[d] + 1] - off[rep[d]];
}
__global__ void pg_vdict_value_copy_kernel(const uint8_t* __restrict__ blob, const int64_t* __restrict__ off, const uint32_t* __restrict__ rep,
                                           const int64_t* __restrict__ out_off, uint8_t* __restrict__ out, uint32_t card) {
  const uint32_t d = blockIdx.x;   // one workgroup per distinct value
  if (d >= card) return;
  const int64_t s = off[rep[d]], len = off[rep[d] + 1] - s, o = out_off[d];
  for (int64_t k = threadIdx.x; k < len; k += blockDim.x) out[o + k] = blob[s + k];
}

__global__ void pg_vdict_ids_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ distinct, uint32_t card,
                                    uint32_t* __restrict__ ids, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  uint32_t lo = 0, hi = card;   // lower_bound
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (distinct[mid] < k) lo = mid + 1; else hi = mid;
  }
  ids[i] = lo;
}

// One thread per output dword of the bit-packed forward index: a wave tile of 2 048 values is a big-endian bit stream of 64 x bits
// dwords starting on a dword boundary (packed_wtile_base in pg_kernels.hip).
__global__ void pg_vdict_pack_kernel(const uint32_t* __restrict__ ids, int64_t n, int bits, uint32_t* __restrict__ out, int64_t n_dwords) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_dwords) return;
  const int64_t per_tile = 64 * (int64_t)bits;
  const int64_t tile = w / per_tile;
  const int64_t bit0 = (w - tile * per_tile) * 32;        // first bit of this dword inside the tile's stream
  uint32_t acc = 0;
  int64_t v = bit0 / bits;
  for (int64_t pos = v * bits; pos < bit0 + 32; pos += bits, v++) {
    const int64_t doc = tile * PG_WAVE_DOCS + v;
    const uint64_t id = doc < n && v < PG_WAVE_DOCS ? ids[doc] : 0u;
    // value occupies stream bits [pos, pos + bits); dword covers [bit0, bit0 + 32): MSB first
    const int64_t shift = (bit0 + 32) - (pos + bits);     // position of the value's LSB counted from the dword's LSB
    if (shift >= 0) acc |= (uint32_t)(id << shift);
    else acc |= (uint32_t)(id >> (-shift));
  }
  out[w] = __builtin_bswap32(acc);
}

}  // namespace

int64_t vdict_value_of_key(uint64_t key, int kind, double* as_double) {
  switch (kind) {
    case 0: case 1: { const int64_t v = (int64_t)(key ^ (1ULL << 63)); if (as_double) *as_double = (double)v; return v; }
    case 2: {
      const uint32_t k = (uint32_t)key;
      const uint32_t f = (k >> 31) ? (k ^ 0x80000000u) : ~k;
      float x; memcpy(&x, &f, 4);
      if (as_double) *as_double = (double)x;
      const double d = (double)x; int64_t bits; memcpy(&bits, &d, 8); return bits;
    }
    default: {
      const uint64_t d = (key >> 63) ? (key ^ (1ULL << 63)) : ~key;
      double x; memcpy(&x, &d, 8);
      if (as_double) *as_double = x;
      return (int64_t)d;
    }
  }
}

// Builds c.vdict (idempotent; the caller holds the segment's lock).  The segment's device is current.
void ensure_virtual_dictionary(Segment& seg, Column& c) {
  if (c.vdict) return;
  const bool var_bytes = c.col_kind == PG_COL_VAR_BYTES;
  if (c.has_dictionary || (!var_bytes && ((c.col_kind != PG_COL_RAW32 && c.col_kind != PG_COL_RAW64) || c.data_type > PG_TYPE_DOUBLE)))
    fail(PG_ERR_UNSUPPORTED, "no-dictionary group-by column %s: only raw INT / LONG / FLOAT / DOUBLE / STRING / BYTES columns get a virtual dictionary", c.name.c_str());
  const int kind = var_bytes ? 4 : c.data_type == PG_TYPE_INT ? 0 : c.data_type == PG_TYPE_LONG ? 1 : c.data_type == PG_TYPE_FLOAT ? 2 : 3;
  const int width = c.col_kind == PG_COL_RAW32 ? 4 : 8;
  const int64_t n = seg.total_docs;
  auto vd = std::make_unique<Column>();
  vd->name = c.name + "$ids";
  vd->data_type = c.data_type;
  vd->has_dictionary = true;
  vd->col_kind = PG_COL_FIXED_BIT;
  vd->val_type = c.val_type;
  std::vector<uint64_t> distinct_host;
  if (n > 0) {
    DeviceBuffer keys((size_t)n * 8), sorted((size_t)n * 8), count(8, true);
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (var_bytes) hipLaunchKernelGGL(pg_vdict_hash_kernel, dim3(grid), dim3(256), 0, 0, c.fwd_dev.as<uint8_t>(), c.vb_offsets_dev.as<int64_t>(), keys.as<uint64_t>(), n);
    else hipLaunchKernelGGL(pg_vdict_keys_kernel, dim3(grid), dim3(256), 0, 0, c.fwd_dev.as<uint8_t>(), width, kind, keys.as<uint64_t>(), n);
    PG_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    PG_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys.as<uint64_t>(), sorted.as<uint64_t>(), (size_t)n));
    {
      DeviceBuffer tmp(std::max<size_t>(tmp_bytes, 16));
      PG_HIP(rocprim::radix_sort_keys(tmp.ptr, tmp_bytes, keys.as<uint64_t>(), sorted.as<uint64_t>(), (size_t)n));
    }
    // unique into a fresh buffer (it cannot run in place)
    DeviceBuffer distinct((size_t)n * 8);
    size_t u_bytes = 0;
    PG_HIP(rocprim::unique(nullptr, u_bytes, sorted.as<uint64_t>(), distinct.as<uint64_t>(), count.as<uint32_t>(), (size_t)n));
    {
      DeviceBuffer tmp(std::max<size_t>(u_bytes, 16));
      PG_HIP(rocprim::unique(tmp.ptr, u_bytes, sorted.as<uint64_t>(), distinct.as<uint64_t>(), count.as<uint32_t>(), (size_t)n));
    }
    uint32_t card = 0;
    PG_HIP(hipMemcpy(&card, count.ptr, 4, hipMemcpyDeviceToHost));
    if (card == 0 || card > 0x7FFFFFFFu) fail(PG_ERR_UNSUPPORTED, "column %s: %u distinct values", c.name.c_str(), card);
    distinct_host.resize(card);
    PG_HIP(hipMemcpy(distinct_host.data(), distinct.ptr, (size_t)card * 8, hipMemcpyDeviceToHost));
    // ids (reusing `sorted` as the 32-bit id array), then the bit-packed forward index
    uint32_t* ids = sorted.as<uint32_t>();
    hipLaunchKernelGGL(pg_vdict_ids_kernel, dim3(grid), dim3(256), 0, 0, keys.as<uint64_t>(), distinct.as<uint64_t>(), card, ids, n);
    PG_HIP(hipGetLastError());
    int bits = 1;
    while (bits < 31 && (1u << bits) < card) bits++;
    vd->bits = bits;
    vd->cardinality = (int32_t)card;
    const size_t padded = (size_t)seg.n_tiles * PG_TILE_DOCS;
    const size_t bytes = (padded * (size_t)bits + 7) / 8 + 64;
    vd->fwd_dev.alloc(bytes, true);
    const int64_t n_dwords = (int64_t)(padded / PG_WAVE_DOCS) * 64 * bits;
    hipLaunchKernelGGL(pg_vdict_pack_kernel, dim3((unsigned)((n_dwords + 255) / 256)), dim3(256), 0, 0, ids, n, bits, vd->fwd_dev.as<uint32_t>(), n_dwords);
    PG_HIP(hipGetLastError());
    if (var_bytes) {
      // the hash is an identity on this column?  representatives, then every doc against its representative
      DeviceBuffer rep((size_t)card * 4), flag(8, true);
      PG_HIP(hipMemset(rep.ptr, 0xFF, (size_t)card * 4));
      PG_HIP(hipStreamSynchronize(nullptr));   // (the fill runs on the legacy default stream, the kernels below on a non-blocking one)
      hipLaunchKernelGGL(pg_vdict_rep_kernel, dim3(grid), dim3(256), 0, 0, ids, rep.as<uint32_t>(), n);
      hipLaunchKernelGGL(pg_vdict_verify_kernel, dim3(grid), dim3(256), 0, 0, c.fwd_dev.as<uint8_t>(), c.vb_offsets_dev.as<int64_t>(), ids, rep.as<uint32_t>(),
                         flag.as<uint32_t>(), n);
      PG_HIP(hipGetLastError());
      uint32_t collided = 0;
      PG_HIP(hipMemcpy(&collided, flag.ptr, 4, hipMemcpyDeviceToHost));
      if (collided) fail(PG_ERR_UNSUPPORTED, "column %s: two distinct values share a 64-bit hash; its GROUP BY stays with the Java plan", c.name.c_str());
      // the distinct values in id order: lengths, exclusive scan, gather; copied to the host once (the groups' keys come from here)
      DeviceBuffer len(((size_t)card + 1) * 8, true), out_off(((size_t)card + 1) * 8, true);
      hipLaunchKernelGGL(pg_vdict_value_len_kernel, dim3((card + 255) / 256), dim3(256), 0, 0, c.vb_offsets_dev.as<int64_t>(), rep.as<uint32_t>(), len.as<int64_t>(), card);
      size_t s_bytes = 0;
      PG_HIP(rocprim::exclusive_scan(nullptr, s_bytes, len.as<int64_t>(), out_off.as<int64_t>(), (int64_t)0, (size_t)card + 1, rocprim::plus<int64_t>()));
      {
        DeviceBuffer tmp(std::max<size_t>(s_bytes, 16));
        PG_HIP(rocprim::exclusive_scan(tmp.ptr, s_bytes, len.as<int64_t>(), out_off.as<int64_t>(), (int64_t)0, (size_t)card + 1, rocprim::plus<int64_t>()));
      }
      vd->vdict_bytes_off.resize((size_t)card + 1);
      PG_HIP(hipMemcpy(vd->vdict_bytes_off.data(), out_off.ptr, ((size_t)card + 1) * 8, hipMemcpyDeviceToHost));
      const int64_t total = vd->vdict_bytes_off[card];
      DeviceBuffer values((size_t)std::max<int64_t>(total, 1));
      hipLaunchKernelGGL(pg_vdict_value_copy_kernel, dim3(card), dim3(64), 0, 0, c.fwd_dev.as<uint8_t>(), c.vb_offsets_dev.as<int64_t>(), rep.as<uint32_t>(),
                         out_off.as<int64_t>(), values.as<uint8_t>(), card);
      PG_HIP(hipGetLastError());
      vd->vdict_bytes.resize((size_t)total);
      if (total > 0) PG_HIP(hipMemcpy(vd->vdict_bytes.data(), values.ptr, (size_t)total, hipMemcpyDeviceToHost));
    }
    PG_HIP(hipDeviceSynchronize());
    seg.device_bytes += bytes;
  } else {
    vd->bits = 1;
    vd->cardinality = 1;
    vd->fwd_dev.alloc(64, true);
    distinct_host.push_back(kind == 4 ? 0 : key_of_bits(0, kind));
    if (kind == 4) vd->vdict_bytes_off.assign(2, 0);
  }
  vd->vdict_kind = kind;
  vd->vdict_keys = std::move(distinct_host);
  uint64_t h = 1469598103934665603ULL;
  for (uint64_t k : vd->vdict_keys) { h ^= k; h *= 1099511628211ULL; }
  for (uint8_t b : vd->vdict_bytes) { h ^= b; h *= 1099511628211ULL; }   // var-byte keys: the values themselves (ids follow hash order)
  vd->vdict_hash = h;
  c.vdict = std::move(vd);
}

}  // namespace pg
