// Probe: does ds_read_b32 / ds_read_b64 at a byte address that is not dword aligned return the bytes at that address on gfx950 (unaligned access
// mode), and what does it cost?  hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(uint32_t* out, uint64_t* out64, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)buf + (uint32_t)threadIdx.x * (uint32_t)stride + 1u;   // byte address, misaligned by (stride * lane + 1) & 3
  uint32_t v;
  uint64_t w;
  asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr));
  out[threadIdx.x] = v;
  out64[threadIdx.x] = w;
}
template <int UNALIGNED>
__global__ void bench(uint32_t* out, int iters, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = (uint8_t)i;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)buf + (uint32_t)(threadIdx.x & 63) * (uint32_t)stride + (UNALIGNED ? 1u : 0u);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(base), "n"(k * 4));
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
    for (int k = 0; k < 8; k++) acc += v[k];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  uint32_t* d; uint64_t* d64;
  hipMalloc(&d, 1 << 22); hipMalloc(&d64, 1 << 20);
  for (int stride : {20, 7, 5}) {
    probe<<<1, 64>>>(d, d64, stride);
    std::vector<uint32_t> h(64); std::vector<uint64_t> h64(64);
    hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost); hipMemcpy(h64.data(), d64, 512, hipMemcpyDeviceToHost);
    int bad32 = 0, bad64 = 0;
    for (int l = 0; l < 64; l++) {
      uint32_t e = 0; uint64_t e64 = 0;
      for (int b = 0; b < 8; b++) { const uint64_t byte = (uint8_t)((l * stride + 1 + b) * 7 + 3); if (b < 4) e |= (uint32_t)byte << (8 * b); e64 |= byte << (8 * b); }
      bad32 += h[l] != e; bad64 += h64[l] != e64;
    }
    printf("stride %2d: ds_read_b32 unaligned wrong lanes %d / 64, ds_read_b64 unaligned wrong lanes %d / 64\n", stride, bad32, bad64);
  }
  for (int stride : {20, 16, 7}) {
    for (int un = 0; un < 2; un++) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      const int iters = 20000;
      if (un) bench<1><<<256, 512>>>(d, 10, stride); else bench<0><<<256, 512>>>(d, 10, stride);
      hipEventRecord(a);
      if (un) bench<1><<<256, 512>>>(d, iters, stride); else bench<0><<<256, 512>>>(d, iters, stride);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      // per CU: 8 wavefronts x iters x 8 reads
      printf("stride %2d %s: %.3f ms -> %.2f ns per wave-level ds_read_b32 per CU (= %.1f cycles at 2.4 GHz)\n", stride, un ? "misaligned" : "aligned   ", ms,
             ms * 1e6 / (8.0 * iters * 8), ms * 1e6 / (8.0 * iters * 8) * 2.4);
    }
  }
  return 0;
}
