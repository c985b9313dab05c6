// Probe: what does a partitioning write pattern cost on MI355X?  Every wavefront writes `chunk`-byte pieces (16 B per lane, chunk/16
// lanes active) to pseudo-random chunk-aligned places of a 3.2 GB area — the write side of a radix scatter with LDS staging — with
// plain and non-temporal stores; chunk = 16 is the unstaged scatter (every lane its own place).  Reports time and GB/s; run under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to see whether partial / full-line writes are fetched first.  Dev tool (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; return x ^ (x >> 31);
}

// total_chunks pieces of CHUNK bytes; piece i lands at chunk slot perm(i) (a bijection: multiply by an odd number mod 2^k)
template <int CHUNK, bool NT>
__global__ void __launch_bounds__(1024) scatter_kernel(uint8_t* __restrict__ area, uint64_t total_chunks, uint64_t mul, uint64_t mask) {
  const int lane = threadIdx.x & 63;
  constexpr int LANES = CHUNK / 16;            // lanes per piece
  constexpr int PER_WAVE = 64 / LANES;         // pieces per wave instruction
  const uint64_t wave = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 6);
  const uint64_t n_waves = (uint64_t)gridDim.x * 16;
  for (uint64_t i0 = wave * PER_WAVE; i0 < total_chunks; i0 += n_waves * PER_WAVE) {
    const uint64_t piece = i0 + (uint64_t)(lane / LANES);
    if (piece >= total_chunks) continue;
    const uint64_t slot = (piece * mul) & mask;
    u32x4 v = {(uint32_t)piece, (uint32_t)lane, 1u, 2u};
    u32x4* dst = reinterpret_cast<u32x4*>(area + slot * CHUNK + (uint64_t)(lane % LANES) * 16);
    if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
  }
}

template <int CHUNK, bool NT>
void run(uint8_t* area, uint64_t bytes, const char* name) {
  const uint64_t total_chunks = bytes / CHUNK;   // a power of two
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 4; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((scatter_kernel<CHUNK, NT>), dim3(256), dim3(1024), 0, 0, area, total_chunks, 0x9E3779B97F4A7C15ULL | 1ULL, total_chunks - 1);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-34s chunk=%5d B %s  %.3f ms  %.1f GB/s written\n", name, CHUNK, NT ? "nt   " : "plain", best, bytes / best / 1e6);
}

int main() {
  const uint64_t bytes = 1ULL << 32;   // 4 GiB area, every byte written once per launch
  uint8_t* area;
  CK(hipMalloc(&area, bytes));
  CK(hipMemset(area, 0, bytes));
  run<16, false>(area, bytes, "unstaged 16-B tuples");
  run<16, true>(area, bytes, "unstaged 16-B tuples");
  run<64, false>(area, bytes, "half-line flushes");
  run<128, false>(area, bytes, "one-line flushes");
  run<128, true>(area, bytes, "one-line flushes");
  run<256, false>(area, bytes, "two-line flushes");
  run<256, true>(area, bytes, "two-line flushes");
  run<512, false>(area, bytes, "four-line flushes");
  run<1024, false>(area, bytes, "whole-wave 1 KB flushes");
  run<1024, true>(area, bytes, "whole-wave 1 KB flushes");
  return 0;
}
