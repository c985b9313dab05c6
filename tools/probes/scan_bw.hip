// Probe: achievable HBM read rate of a range-predicate scan over a raw big-endian INT column, as a function of
// workgroup shape and loads in flight.  Dev tool (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)

template <int U, int BLOCK>
__global__ void __launch_bounds__(BLOCK) scan_kernel(const u32x4* __restrict__ data, long n_quads, int lo, unsigned span,
                                                     unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long n_waves = (long)gridDim.x * (BLOCK / 64);
  unsigned cnt = 0;
  // wave tile = U*64 quads
  const long n_wt = n_quads / (U * 64);
  for (long wt = wave; wt < n_wt; wt += n_waves) {
    const u32x4* p = data + wt * (U * 64) + lane;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
    for (int u = 0; u < U; u++) {
      cnt += (unsigned)(__builtin_bswap32(v[u].x) - lo) <= span;
      cnt += (unsigned)(__builtin_bswap32(v[u].y) - lo) <= span;
      cnt += (unsigned)(__builtin_bswap32(v[u].z) - lo) <= span;
      cnt += (unsigned)(__builtin_bswap32(v[u].w) - lo) <= span;
    }
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) atomicAdd(out, (unsigned long long)cnt);
}

template <int U, int BLOCK>
void run(const u32x4* d, long n_quads, int grid, unsigned long long* out, const char* name) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 6; it++) {
    CK(hipMemset(out, 0, 8));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((scan_kernel<U, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, d, n_quads, 250000, 499999u, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  unsigned long long h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
  printf("%-28s U=%d block=%4d grid=%5d  %.3f ms  %.1f GB/s  count=%llu\n", name, U, BLOCK, grid, best, n_quads * 16.0 / best / 1e6, h);
}

int main() {
  const long n = 1000000000L;
  const long n_quads = n / 4;
  u32x4* d; unsigned long long* out;
  CK(hipMalloc(&d, n_quads * 16)); CK(hipMalloc(&out, 8));
  // fill with pseudo-random big-endian ints in [0, 1e6)
  {
    std::vector<uint32_t> h(1 << 24);
    uint64_t s = 88172645463325252ULL;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = __builtin_bswap32((uint32_t)((s >> 32) * 1000000ULL >> 32)); }
    for (long off = 0; off < n; off += (1 << 24)) {
      long c = (n - off) < (1 << 24) ? (n - off) : (1 << 24);
      CK(hipMemcpy((uint32_t*)d + off, h.data(), c * 4, hipMemcpyHostToDevice));
    }
  }
  run<8, 1024>(d, n_quads, 256, out, "1wg/cu 16 waves");
  run<4, 1024>(d, n_quads, 256, out, "1wg/cu 16 waves");
  run<2, 1024>(d, n_quads, 256, out, "1wg/cu 16 waves");
  run<8, 512>(d, n_quads, 512, out, "2wg/cu 8 waves");
  run<8, 512>(d, n_quads, 1024, out, "4wg/cu 8 waves");
  run<8, 256>(d, n_quads, 2048, out, "8wg/cu 4 waves");
  run<4, 256>(d, n_quads, 2048, out, "8wg/cu 4 waves");
  run<2, 256>(d, n_quads, 2048, out, "8wg/cu 4 waves");
  run<1, 256>(d, n_quads, 2048, out, "8wg/cu 4 waves");
  run<4, 256>(d, n_quads, 4096, out, "16wg/cu(q) 4 waves");
  run<8, 256>(d, n_quads, 1024, out, "4wg/cu 4 waves");
  run<8, 256>(d, n_quads, 512, out, "2wg/cu 4 waves");
  run<8, 256>(d, n_quads, 256, out, "1wg/cu 4 waves");
  run<16, 256>(d, n_quads, 1024, out, "4wg/cu 4 waves");
  return 0;
}
