// Probe (round 5): can a partition scatter write its tuples DIRECTLY — every lane a 4-byte store to the next free place of its bucket's
// stream — instead of sorting rounds in LDS and copying whole 128-byte lines out (pg_kernels_part.hip phases B..D, ~60 of the scatter's
// ~100 wave instructions per doc)?  Every wavefront owns NB append streams (contiguous regions); per doc: bucket = hash % NB, rank =
// returning LDS atomic on the wavefront's own counter, one dword store.  Successive stores of a stream land in the same 128-byte line
// until it is full: the question is whether L2 combines them or HBM sees partial lines (tools/probes/scatter_bw.hip: fully random 16-byte
// writes reach 0.35 TB/s).  Reports G tuples/s; under rocprofv3 --pmc WRITE_SIZE the bytes that reached memory.  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; return x ^ (x >> 16); }

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) append_kernel(uint32_t* __restrict__ area, const uint32_t* __restrict__ src, uint64_t n_docs, int nb, uint32_t cap) {
  extern __shared__ uint32_t cnt[];   // [WAVES][nb]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* my = cnt + wave * nb;
  for (int i = lane; i < nb; i += 64) my[i] = 0;
  const uint64_t gw = (uint64_t)blockIdx.x * WAVES + wave, n_waves = (uint64_t)gridDim.x * WAVES;
  uint32_t* base = area + gw * (uint64_t)nb * cap;
  // 16 docs per lane and round, as the scatter: 4 x 16-byte loads, 16 rank atomics back to back, 16 stores
  for (uint64_t r = gw; r * 1024 < n_docs; r += n_waves) {
    const uint4* p = reinterpret_cast<const uint4*>(src + r * 1024) + lane;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = p[k * 64];
    uint32_t x[16], b[16], rk[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { x[4 * k] = v[k].x; x[4 * k + 1] = v[k].y; x[4 * k + 2] = v[k].z; x[4 * k + 3] = v[k].w; }
#pragma unroll
    for (int j = 0; j < 16; j++) b[j] = (uint32_t)(((uint64_t)mix32(x[j]) * (uint32_t)nb) >> 32);
#pragma unroll
    for (int j = 0; j < 16; j++) rk[j] = atomicAdd(&my[b[j]], 1u);
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (rk[j] < cap) base[(uint64_t)b[j] * cap + rk[j]] = x[j];
  }
}

template <int WAVES>
void run(uint32_t* area, const uint32_t* src, uint64_t n_docs, int nb, int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu;
  const uint64_t n_waves = (uint64_t)grid * WAVES;
  const uint32_t cap = (uint32_t)((n_docs / (n_waves * nb)) * 5 / 4 + 64) & ~31u;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 3; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((append_kernel<WAVES>), dim3(grid), dim3(WAVES * 64), WAVES * nb * 4, 0, area, src, n_docs, nb, cap);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("direct append: buckets=%4d waves/wg=%d wgs/cu=%d (%5llu streams, %6u tuples each)  %.3f ms  %.1f G tuples/s  %.2f TB/s (4 B read + 4 B written)\n", nb, WAVES,
         wgs_per_cu, (unsigned long long)(n_waves * nb), cap, best, n_docs / best / 1e6, n_docs * 8.0 / best / 1e9);
}

int main() {
  const uint64_t n_docs = 200000000ULL / 1024 * 1024;
  uint32_t *area, *src;
  CK(hipMalloc(&area, n_docs * 4 * 2));
  CK(hipMalloc(&src, n_docs * 4));
  CK(hipMemset(area, 0, n_docs * 4 * 2));
  CK(hipMemset(src, 0x5A, n_docs * 4));
  {   // distinct values per doc (the bucket is a hash of them)
    uint32_t* h = (uint32_t*)malloc(n_docs * 4);
    for (uint64_t i = 0; i < n_docs; i++) h[i] = (uint32_t)i * 2654435761u;
    CK(hipMemcpy(src, h, n_docs * 4, hipMemcpyHostToDevice));
    free(h);
  }
  for (int nb : {20, 64, 80, 160, 256}) {
    run<4>(area, src, n_docs, nb, 4);
    run<4>(area, src, n_docs, nb, 2);
    run<16>(area, src, n_docs, nb, 1);
  }
  return 0;
}
