// Chunk decompression at segment load: compressed raw forward indexes (BaseChunkForwardIndexReader.java:61-111 — SNAPPY,
// LZ4, LZ4_LENGTH_PREFIXED chunks of numDocsPerChunk values) are uploaded as they are on disk and decompressed in HBM, once,
// into the PASS_THROUGH layout the query kernels read (flat big-endian values).  The reference decompresses the chunk of the
// docId being read into a per-thread ChunkReaderContext on every access (decompressChunk :141-163); here the cost is paid at
// IndexSegment load and queries stream the column at HBM speed.
//
// One wavefront per chunk.  The compressed chunk is staged into LDS with coalesced loads; the element stream (tags, lengths,
// offsets) is parsed from LDS with wave-uniform control flow (readfirstlane → scalar registers, scalar branches), and every
// literal run / back-reference is copied by the 64 lanes in parallel inside LDS (a back-reference of length > offset repeats
// its pattern: lane i reads position i mod offset of the already complete window, so the lanes never depend on one another).
// The decompressed chunk leaves LDS with coalesced dword stores.  Formats: snappy format_description.txt; LZ4 block format.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pg_internal.hpp"

namespace pg {

#define PG_DC_BLOCK 256
#define PG_DC_MAX_CHUNK_BYTES 65536   // out window + staged input must fit the 160 KB LDS of one CU

enum { kSnappy = 1, kLz4 = 3, kLz4Len = 4 };   // ChunkCompressionType values

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int CODEC>
__global__ __launch_bounds__(PG_DC_BLOCK) void pg_decompress_chunks_kernel(const uint8_t* __restrict__ in,
                                                                         const uint64_t* __restrict__ chunk_off,
                                                                         uint8_t* __restrict__ out, uint32_t chunk_bytes,
                                                                         uint64_t total_bytes, uint32_t in_cap, int n_chunks,
                                                                         int waves_per_block, int* __restrict__ error) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int wave = (int)(threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  const int chunk = (int)blockIdx.x * waves_per_block + wave;
  if (wave >= waves_per_block || chunk >= n_chunks) return;
  const uint32_t out_cap = (chunk_bytes + 15u) & ~15u;
  uint8_t* s_out = lds + (size_t)wave * (out_cap + in_cap);
  uint8_t* s_in = s_out + out_cap;
  const uint64_t src0 = chunk_off[chunk];
  const uint64_t n64 = chunk_off[chunk + 1] - src0;
  const uint64_t dst0 = (uint64_t)chunk * chunk_bytes;
  const uint32_t want = (uint32_t)min((uint64_t)chunk_bytes, total_bytes - dst0);
  bool bad = n64 > in_cap;
  const uint32_t n = bad ? 0u : (uint32_t)n64;
  for (uint32_t i = lane; i < n; i += 64) s_in[i] = in[src0 + i];
  __builtin_amdgcn_wave_barrier();

  uint32_t ip = 0, op = 0;
  if (CODEC == kSnappy && !bad) {   // preamble: varint32 uncompressed length
    uint32_t v = 0, shift = 0;
    bool done = false;
    while (!done && ip < n && shift < 35) {
      const uint32_t b = uni(s_in[ip++]);
      v |= (b & 0x7Fu) << shift;
      shift += 7;
      done = !(b & 0x80u);
    }
    if (!done || v != want) bad = true;
  }
  if (CODEC == kLz4Len && !bad) {   // LZ4DecompressorWithLength: little-endian int, the decompressed length
    if (n < 4) bad = true;
    else {
      const uint32_t v = uni((uint32_t)s_in[0] | ((uint32_t)s_in[1] << 8) | ((uint32_t)s_in[2] << 16) | ((uint32_t)s_in[3] << 24));
      if (v != want) bad = true;
      ip = 4;
    }
  }
  while (!bad && ip < n) {
    uint32_t lit = 0, mlen = 0, offset = 0;
    if (CODEC == kSnappy) {
      const uint32_t tag = uni(s_in[ip++]);
      const uint32_t kind = tag & 3u;
      if (kind == 0) {
        lit = (tag >> 2) + 1;
        if (lit > 60) {
          const uint32_t extra = lit - 60;
          if (ip + extra > n) { bad = true; break; }
          uint32_t v = 0;
          for (uint32_t k = 0; k < extra; k++) v |= (uint32_t)s_in[ip + k] << (8 * k);
          if (uni(v) >= want) { bad = true; break; }   // v + 1 > want (and 0xFFFFFFFF + 1 must not wrap to 0)
          lit = uni(v) + 1;
          ip += extra;
        }
      } else if (kind == 1) {
        if (ip + 1 > n) { bad = true; break; }
        mlen = ((tag >> 2) & 7u) + 4;
        offset = ((tag >> 5) << 8) | uni(s_in[ip]);
        ip += 1;
      } else if (kind == 2) {
        if (ip + 2 > n) { bad = true; break; }
        mlen = (tag >> 2) + 1;
        offset = uni((uint32_t)s_in[ip] | ((uint32_t)s_in[ip + 1] << 8));
        ip += 2;
      } else {
        if (ip + 4 > n) { bad = true; break; }
        mlen = (tag >> 2) + 1;
        offset = uni((uint32_t)s_in[ip] | ((uint32_t)s_in[ip + 1] << 8) | ((uint32_t)s_in[ip + 2] << 16) | ((uint32_t)s_in[ip + 3] << 24));
        ip += 4;
      }
      if (kind != 0 && offset == 0) { bad = true; break; }
    } else {
      const uint32_t token = uni(s_in[ip++]);
      lit = token >> 4;
      if (lit == 15) {
        uint32_t b;
        do {
          if (ip >= n) { bad = true; break; }
          b = uni(s_in[ip++]);
          lit += b;
          if (lit > want) { bad = true; break; }   // a length chain cannot exceed the chunk (and must not wrap 32 bits)
        } while (b == 255);
        if (bad) break;
      }
      mlen = token & 15u;   // resolved after the literals: the last sequence of a block has none
    }
    if (lit) {
      if (lit > n - ip || lit > want - op) { bad = true; break; }   // ip <= n, op <= want: no 32-bit wrap for 4-byte snappy lengths
      for (uint32_t i = lane; i < lit; i += 64) s_out[op + i] = s_in[ip + i];
      ip += lit;
      op += lit;
      __builtin_amdgcn_wave_barrier();
    }
    if (CODEC != kSnappy) {
      if (ip >= n) break;   // end of the block
      if (ip + 2 > n) { bad = true; break; }
      offset = uni((uint32_t)s_in[ip] | ((uint32_t)s_in[ip + 1] << 8));
      ip += 2;
      if (mlen == 15) {
        uint32_t b;
        do {
          if (ip >= n) { bad = true; break; }
          b = uni(s_in[ip++]);
          mlen += b;
          if (mlen > want) { bad = true; break; }
        } while (b == 255);
        if (bad) break;
      }
      mlen += 4;
      if (offset == 0) { bad = true; break; }
    }
    if (mlen) {
      if (offset > op || mlen > want - op) { bad = true; break; }
      const uint8_t* window = s_out + (op - offset);
      if (offset >= mlen) {
        for (uint32_t i = lane; i < mlen; i += 64) s_out[op + i] = window[i];
      } else {
        for (uint32_t i = lane; i < mlen; i += 64) s_out[op + i] = window[i % offset];
      }
      op += mlen;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (bad || op != want) {
    if (lane == 0) atomicCAS(error, 0, chunk + 1);
    return;
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(out + dst0);   // chunk_bytes is a multiple of the value width (4 or 8)
  const uint32_t* srcw = reinterpret_cast<const uint32_t*>(s_out);
  for (uint32_t i = lane; i < want / 4; i += 64) dst[i] = srcw[i];
}

// `file`: the forward index as stored; offs: numChunks + 1 byte positions in it (chunk i = [offs[i], offs[i+1])); dst: flat values
void decompress_fixed_byte_chunks(int compression, const uint8_t* file, const std::vector<uint64_t>& offs, uint32_t chunk_bytes,
                                  uint64_t total_bytes, uint8_t* dst, const char* column) {
  const int n_chunks = (int)offs.size() - 1;
  if (n_chunks <= 0 || total_bytes == 0) return;
  if (host_codec(compression)) {   // ZSTANDARD / GZIP: entropy-coded streams, decoded on the host once (pg_host_codecs.cpp)
    host_decompress_fixed_byte_chunks(compression, file, offs, chunk_bytes, total_bytes, dst, column);
    return;
  }
  if (compression != kSnappy && compression != kLz4 && compression != kLz4Len)
    fail(PG_ERR_UNSUPPORTED, "column %s: chunk compression type %d is not a ChunkCompressionType", column, compression);
  if (chunk_bytes == 0 || chunk_bytes > PG_DC_MAX_CHUNK_BYTES || (chunk_bytes & 3u))
    fail(PG_ERR_UNSUPPORTED, "column %s: %u-byte chunks (compressed chunks up to %d bytes are decompressed on the GPU)", column,
         chunk_bytes, PG_DC_MAX_CHUNK_BYTES);
  // worst-case compressed size of a chunk (snappy: 32 + n + n/6; LZ4: n + n/255 + 16; + the 4-byte length prefix)
  const uint32_t in_cap = (chunk_bytes + chunk_bytes / 6 + 64 + 15) & ~15u;
  const size_t per_wave = (size_t)((chunk_bytes + 15u) & ~15u) + in_cap;
  const int waves = (int)std::max<size_t>(1, std::min<size_t>(PG_DC_BLOCK / 64, (64 * 1024) / per_wave));
  const size_t lds = per_wave * (size_t)waves;
  std::vector<uint64_t> rel(offs.size());
  for (size_t i = 0; i < offs.size(); i++) rel[i] = offs[i] - offs[0];
  DeviceBuffer in_dev(rel.back() + 16);
  in_dev.upload(file + offs[0], rel.back());
  DeviceBuffer off_dev = upload_vector(rel);
  DeviceBuffer err_dev(sizeof(int), true);
  const int grid = (n_chunks + waves - 1) / waves;
  auto launch = [&](auto kernel) {
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PG_DC_BLOCK), lds, 0, in_dev.as<uint8_t>(), off_dev.as<uint64_t>(), dst, chunk_bytes,
                       total_bytes, in_cap, n_chunks, waves, err_dev.as<int>());
  };
  if (compression == kSnappy) launch(pg_decompress_chunks_kernel<kSnappy>);
  else if (compression == kLz4) launch(pg_decompress_chunks_kernel<kLz4>);
  else launch(pg_decompress_chunks_kernel<kLz4Len>);
  PG_HIP(hipGetLastError());
  int err = 0;
  PG_HIP(hipMemcpy(&err, err_dev.ptr, sizeof(int), hipMemcpyDeviceToHost));   // synchronises with the kernel (null stream)
  if (err) fail(PG_ERR_INVALID_ARGUMENT, "column %s: chunk %d does not decompress to its %u bytes", column, err - 1, chunk_bytes);
}

}  // namespace pg
