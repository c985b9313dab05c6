// Probe: what does a wave-level LDS atomic cost per CU under the group-by table's access pattern (slot = group x R + lane % R, random groups), by
// width (32 / 64 bit), operation (add / max), number of live lanes (EXEC-masked) and address stride?  16 wavefronts per CU, all hammering the table.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
enum { OP_ADD64, OP_MAX64, OP_ADD32, OP_MAX32, OP_ADD32_S4, OP_READ64, OP_READ32 };
template <int OP>
__global__ void __launch_bounds__(1024) bench(uint32_t* out, int iters, int live, int groups, int R) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tab[];
  for (int i = threadIdx.x; i < groups * R * 8 + 1024; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t addr[8];
  uint32_t h = (uint32_t)threadIdx.x * 2654435761u + 12345u;
  for (int k = 0; k < 8; k++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t g = (h >> 8) % (uint32_t)groups;
    const uint32_t slot = g * (uint32_t)R + ((uint32_t)threadIdx.x & (uint32_t)(R - 1));
    addr[k] = (uint32_t)(uintptr_t)tab + slot * (OP == OP_ADD32_S4 ? 4u : 8u);
  }
  uint32_t acc = 0;
  const uint64_t one = 1, val = (uint64_t)lane * 977u;
  const uint32_t val32 = (uint32_t)lane * 977u;
  if (lane < live) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (OP == OP_ADD64) asm volatile("ds_add_u64 %0, %1" :: "v"(addr[k]), "v"(one) : "memory");
        if (OP == OP_MAX64) asm volatile("ds_max_i64 %0, %1" :: "v"(addr[k]), "v"(val) : "memory");
        if (OP == OP_ADD32 || OP == OP_ADD32_S4) asm volatile("ds_add_u32 %0, %1" :: "v"(addr[k]), "v"(val32) : "memory");
        if (OP == OP_MAX32) asm volatile("ds_max_i32 %0, %1" :: "v"(addr[k]), "v"(val32) : "memory");
        if (OP == OP_READ64) { uint64_t w; asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(addr[k]) : "memory"); acc += (uint32_t)w; }
        if (OP == OP_READ32) { uint32_t w; asm volatile("ds_read_b32 %0, %1" : "=v"(w) : "v"(addr[k]) : "memory"); acc += w; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + tab[threadIdx.x];
}
template <int OP> void run(const char* name, uint32_t* d, int live, int groups, int R) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000, waves = 16;
  const size_t lds = (size_t)groups * R * 8 + 1024;
  hipFuncSetAttribute((const void*)bench<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  bench<OP><<<256, waves * 64, lds>>>(d, 10, live, groups, R);
  hipEventRecord(a);
  bench<OP><<<256, waves * 64, lds>>>(d, iters, live, groups, R);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / ((double)waves * iters * 8);
  printf("%-18s live %2d  groups %4d x R %2d: %6.2f ns per wave-level instruction per CU = %5.1f cycles at 2.4 GHz, %5.2f lane-ops / cycle\n", name, live, groups, R, ns, ns * 2.4,
         live / (ns * 2.4));
}
int main() {
  uint32_t* d; hipMalloc(&d, 1 << 22);
  for (int live : {64, 32, 16, 8, 1}) {
    run<OP_ADD64>("ds_add_u64", d, live, 100, 32);
    run<OP_MAX64>("ds_max_i64", d, live, 100, 32);
    run<OP_ADD32>("ds_add_u32 (8 B)", d, live, 100, 32);
    run<OP_MAX32>("ds_max_i32 (8 B)", d, live, 100, 32);
    run<OP_ADD32_S4>("ds_add_u32 (4 B)", d, live, 100, 32);
    run<OP_READ64>("ds_read_b64", d, live, 100, 32);
    run<OP_READ32>("ds_read_b32", d, live, 100, 32);
  }
  for (int R : {16, 64}) { run<OP_ADD64>("ds_add_u64", d, 64, 100, R); run<OP_ADD32>("ds_add_u32 (8 B)", d, 64, 100, R); }
  run<OP_ADD64>("ds_add_u64", d, 64, 1300, 8);
  run<OP_ADD64>("ds_add_u64", d, 64, 1300, 4);
  return 0;
}
