// Probe: throughput of global (HBM-table) atomics by scope and table layout on MI355X.
//   variants: agent-scope on one table | workgroup-scope on a per-XCD replica (HW_REG_XCC_ID) | agent-scope on per-XCD replicas
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip ; run: ./atomic_scope [slots] [updates_per_thread]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
template <int MODE>
__global__ void __launch_bounds__(1024) k(long long* table, unsigned slots, int per_thread, int* xcc_seen) {
  const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7;
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc;
  long long* t = MODE == 0 ? table : table + (size_t)xcc * slots;
  unsigned long long s = (unsigned long long)blockIdx.x * 1024 + threadIdx.x;
  for (int i = 0; i < per_thread; i++) {
    s = mix(s);
    const unsigned g = (unsigned)((s >> 32) * (unsigned long long)slots >> 32);
    if (MODE == 1) __hip_atomic_fetch_add(t + g, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(t + g, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main(int argc, char** argv) {
  unsigned slots = argc > 1 ? atoi(argv[1]) : 80000;
  int per_thread = argc > 2 ? atoi(argv[2]) : 256;
  const int grid = 256, block = 1024;
  long long* table; int* xs;
  CK(hipMalloc(&table, (size_t)slots * 8 * 8)); CK(hipMalloc(&xs, grid * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[3] = {"agent scope, one table", "workgroup scope, per-XCD replica", "agent scope, per-XCD replica"};
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9;
    long long total = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemset(table, 0, (size_t)slots * 8 * 8));
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(block), 0, 0, table, slots, per_thread, xs);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(block), 0, 0, table, slots, per_thread, xs);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(block), 0, 0, table, slots, per_thread, xs);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      std::vector<long long> h((size_t)slots * 8);
      CK(hipMemcpy(h.data(), table, h.size() * 8, hipMemcpyDeviceToHost));
      total = 0; for (long long v : h) total += v;
    }
    const double n = (double)grid * block * per_thread;
    printf("%-36s slots=%u  %.3f ms  %.2f G atomics/s  sum %s (%lld of %.0f)\n", names[mode], slots, best, n / best / 1e6,
           total == (long long)n ? "exact" : "LOST UPDATES", total, n);
  }
  std::vector<int> hx(grid); CK(hipMemcpy(hx.data(), xs, grid * 4, hipMemcpyDeviceToHost));
  printf("XCC_ID of workgroups 0..15:"); for (int i = 0; i < 16; i++) printf(" %d", hx[i]); printf("\n");
  return 0;
}
