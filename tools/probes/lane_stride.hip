// Probe: HBM read rate when every lane of a wavefront owns S contiguous bytes of a sub-tile (S = 16: one fully coalesced dwordx4 per lane;
// S = 32 / 64: S/16 dwordx4 loads per lane, each instruction touching 16 of every S bytes — the "oct" layout's raw INT / LONG columns)
// against the fully coalesced order over the same bytes.  Dev tool (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)

// CH = chunks of 16 bytes per lane and sub-tile; U = sub-tiles requested before the first is consumed; OWNED: lane owns CH contiguous chunks
template <int CH, int U, bool OWNED, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k(const u32x4* __restrict__ data, long n_chunks, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long n_waves = (long)gridDim.x * (BLOCK / 64);
  const long per_tile = 64L * CH * U;
  const long n_t = n_chunks / per_tile;
  unsigned acc = 0;
  for (long t = wave; t < n_t; t += n_waves) {
    const u32x4* p = data + t * per_tile;
    u32x4 v[U][CH];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int c = 0; c < CH; c++)
        v[u][c] = __builtin_nontemporal_load(p + u * 64 * CH + (OWNED ? lane * CH + c : c * 64 + lane));
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int c = 0; c < CH; c++) acc += v[u][c].x ^ v[u][c].y ^ v[u][c].z ^ v[u][c].w;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) atomicAdd(out, (unsigned long long)acc);
}

template <int CH, int U, bool OWNED, int BLOCK>
void run(const u32x4* d, long n_chunks, int grid, unsigned long long* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int it = 0; it < 6; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<CH, U, OWNED, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, d, n_chunks, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-9s bytes/lane=%3d  in flight=%d x %d  block=%4d grid=%5d  %.3f ms  %7.1f GB/s\n", OWNED ? "owned" : "coalesced", CH * 16, U, CH, BLOCK, grid, best,
         n_chunks * 16.0 / best / 1e6);
}

int main() {
  const long bytes = 4L << 30;
  const long n_chunks = bytes / 16;
  u32x4* d; unsigned long long* out;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(d, 1, bytes)); CK(hipMemset(out, 0, 8));
  run<1, 8, true, 1024>(d, n_chunks, 256, out);
  run<2, 4, true, 1024>(d, n_chunks, 256, out);
  run<2, 4, false, 1024>(d, n_chunks, 256, out);
  run<4, 2, true, 1024>(d, n_chunks, 256, out);
  run<4, 2, false, 1024>(d, n_chunks, 256, out);
  run<4, 4, true, 1024>(d, n_chunks, 256, out);
  run<4, 4, false, 1024>(d, n_chunks, 256, out);
  run<2, 4, true, 512>(d, n_chunks, 512, out);
  run<2, 4, false, 512>(d, n_chunks, 512, out);
  run<4, 2, true, 512>(d, n_chunks, 512, out);
  run<4, 2, false, 512>(d, n_chunks, 512, out);
  run<4, 4, true, 512>(d, n_chunks, 512, out);
  run<4, 4, false, 512>(d, n_chunks, 512, out);
  run<2, 8, true, 512>(d, n_chunks, 512, out);
  run<4, 4, true, 256>(d, n_chunks, 1024, out);
  run<4, 4, false, 256>(d, n_chunks, 1024, out);
  return 0;
}
