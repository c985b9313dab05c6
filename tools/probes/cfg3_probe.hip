// Probe: hand-specialised cfg3 pipeline (2 posting leaves AND raw-INT range scan → GROUP BY 7-bit dict col, SUM/MAX of
// a raw INT metric) to find the achievable ceiling and the cost of each stage.  Dev tool (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)
#define DEVFN __device__ __forceinline__

__global__ void fill_random(uint32_t* p, long n, uint64_t seed, uint32_t range, int be, int thin = 0) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    uint32_t v = range ? (uint32_t)(((z >> 32) * range) >> 32) : (uint32_t)z;
    for (int j = 0; j < thin; j++) {
      uint64_t y = seed * 77 + j * 1315423911ULL + (uint64_t)i * 0xD6E8FEB86659FD93ULL;
      y = (y ^ (y >> 30)) * 0xBF58476D1CE4E5B9ULL; y = (y ^ (y >> 27)) * 0x94D049BB133111EBULL; y ^= y >> 31;
      v &= (uint32_t)y;
    }
    p[i] = be ? __builtin_bswap32(v) : v;
  }
}

DEVFN uint32_t lin_to_quad(uint32_t w, int lane) {
  uint32_t out = 0;
  const uint32_t sh = (uint32_t)(lane & 7) * 4u;
  const int src0 = lane >> 3;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t x = (uint32_t)__shfl((int)w, k * 8 + src0, 64);
    out |= ((x >> sh) & 0xFu) << (4 * k);
  }
  return out;
}

struct Args {
  const uint32_t* post[6];   // 4 bitmaps of leaf 1, 2 of leaf 2 (1 bit / doc)
  const u32x4* r_int;
  const uint32_t* g1;        // 7-bit packed, MSB-first big-endian
  const u32x4* m;
  long n_wt;
  int lo; unsigned span;
  unsigned long long* out;   // [0] matched, then partial tables
  long long* partials;       // [grid][2][100]
};

// MODE bit0: masked loads of g1/m (else unconditional, issued with the filter loads); bit1: skip aggregation atomics
#ifndef NT
#define NT 0
#endif
template <class T> DEVFN T ldg(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int BLOCK, int MODE>
__global__ void __launch_bounds__(BLOCK) cfg3_kernel(const Args a) {
  __shared__ long long tab[2][100 * 32];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 2 * 100 * 32; i += BLOCK) (&tab[0][0])[i] = i < 3200 ? 0 : (long long)0x8000000000000000LL;
  __syncthreads();
  const long wave = (long)blockIdx.x * (BLOCK / 64) + (t >> 6);
  const long n_waves = (long)gridDim.x * (BLOCK / 64);
  const uint32_t rep = t & 31;
  unsigned matched = 0;
  for (long wt = wave; wt < a.n_wt; wt += n_waves) {
    // ---- issue loads
    uint32_t pw[6];
#pragma unroll
    for (int i = 0; i < 6; i++) pw[i] = ldg(&a.post[i][wt * 64 + lane]);
    u32x4 rv[8];
    const u32x4* rp = a.r_int + wt * 512 + lane;
#pragma unroll
    for (int k = 0; k < 8; k++) rv[k] = ldg(&rp[k * 64]);
    u32x4 mv[8];
    u32x2 gv[8];
    const u32x4* mp = a.m + wt * 512 + lane;
    const uint32_t* gp = a.g1 + wt * (2048 * 7 / 32);
    if (!(MODE & 1)) {
#pragma unroll
      for (int k = 0; k < 8; k++) mv[k] = ldg(&mp[k * 64]);
#pragma unroll
      for (int k = 0; k < 8; k++) gv[k] = *(const u32x2_a4*)(gp + ((4u * (k * 64 + lane) * 7u) >> 5));
    }
    // ---- filter
    const uint32_t lin = (pw[0] | pw[1] | pw[2] | pw[3]) & (pw[4] | pw[5]);
    uint32_t cand = lin_to_quad(lin, lane);
    uint32_t mm = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint32_t r = 0;
      r |= (uint32_t)((unsigned)(__builtin_bswap32(rv[k].x) - a.lo) <= a.span) << 0;
      r |= (uint32_t)((unsigned)(__builtin_bswap32(rv[k].y) - a.lo) <= a.span) << 1;
      r |= (uint32_t)((unsigned)(__builtin_bswap32(rv[k].z) - a.lo) <= a.span) << 2;
      r |= (uint32_t)((unsigned)(__builtin_bswap32(rv[k].w) - a.lo) <= a.span) << 3;
      mm |= r << (4 * k);
    }
    mm &= cand;
    matched += __popc(mm);
    if (MODE & 1) {
#pragma unroll
      for (int k = 0; k < 8; k++) { mv[k] = u32x4{0, 0, 0, 0}; if ((mm >> (4 * k)) & 0xF) mv[k] = ldg(&mp[k * 64]); }
#pragma unroll
      for (int k = 0; k < 8; k++) { gv[k] = u32x2{0, 0}; if ((mm >> (4 * k)) & 0xF) gv[k] = *(const u32x2_a4*)(gp + ((4u * (k * 64 + lane) * 7u) >> 5)); }
    }
    // ---- aggregate
    if (!(MODE & 2)) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t nib = (mm >> (4 * k)) & 0xF;
        if (nib) {
          const uint32_t q = k * 64 + lane;
          const uint32_t sh = (4u * q * 7u) & 31u;
          const uint64_t win = ((uint64_t)__builtin_bswap32(gv[k].x) << 32) | __builtin_bswap32(gv[k].y);
          uint32_t x[4] = {__builtin_bswap32(mv[k].x), __builtin_bswap32(mv[k].y), __builtin_bswap32(mv[k].z), __builtin_bswap32(mv[k].w)};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            if ((nib >> i) & 1) {
              uint32_t d = (uint32_t)(win >> (64u - sh - (uint32_t)(i + 1) * 7u)) & 127u;
              if (d > 99) d = 99;
              const uint32_t slot = d * 32 + rep;
              atomicAdd((unsigned long long*)&tab[0][slot], (unsigned long long)x[i]);
              atomicMax(&tab[1][slot], (long long)x[i]);
            }
          }
        }
      }
    } else {
      // keep the loads alive
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) s += mv[k].x ^ mv[k].y ^ mv[k].z ^ mv[k].w ^ gv[k].x ^ gv[k].y;
      if (s == 0x12345678u) matched++;
    }
  }
  for (int off = 32; off > 0; off >>= 1) matched += __shfl_xor(matched, off, 64);
  if (lane == 0) atomicAdd(a.out, (unsigned long long)matched);
  __syncthreads();
  for (int i = t; i < 200; i += BLOCK) {
    const int o = i / 100, g = i % 100;
    long long acc = tab[o][g * 32];
    for (int r = 1; r < 32; r++) { long long v = tab[o][g * 32 + r]; acc = o == 0 ? acc + v : (v > acc ? v : acc); }
    a.partials[(long)blockIdx.x * 200 + i] = acc;
  }
}

template <int BLOCK, int MODE>
void run(const Args& a, int grid, long n, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 6; it++) {
    CK(hipMemset(a.out, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((cfg3_kernel<BLOCK, MODE>), dim3(grid), dim3(BLOCK), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned long long h; CK(hipMemcpy(&h, a.out, 8, hipMemcpyDeviceToHost));
  printf("%-34s block=%4d grid=%5d mode=%d  %.3f ms  %.1f GB/s (%.1f%% of 8TB/s) matched=%llu\n", name, BLOCK, grid, MODE, best,
         n * 9.625 / best / 1e6, n * 9.625 / best / 1e6 / 80.0, h);
}

int main() {
  const long n = 1000000000L / 2048 * 2048;
  Args a;
  uint32_t* post[6];
  for (int i = 0; i < 6; i++) {
    CK(hipMalloc(&post[i], n / 8));
    // leaf1: 4 bitmaps each 1/8 dense, disjoint-ish; just random words AND-ed to get ~12.5% density
    fill_random<<<4096, 256>>>(post[i], n / 32, 1000 + i, 0, 0, i < 4 ? 2 : 1);
    a.post[i] = post[i];
  }
  uint32_t *r_int, *g1, *m;
  CK(hipMalloc(&r_int, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&g1, n * 7 / 8 + 64));
  fill_random<<<8192, 256>>>(r_int, n, 1, 1000000, 1);
  fill_random<<<8192, 256>>>(m, n, 2, 1 << 20, 1);
  fill_random<<<8192, 256>>>(g1, n * 7 / 32, 3, 0, 0);
  CK(hipDeviceSynchronize());
  a.r_int = (const u32x4*)r_int; a.m = (const u32x4*)m; a.g1 = g1;
  a.n_wt = n / 2048; a.lo = 250000; a.span = 499999u;
  CK(hipMalloc(&a.out, 8));
  CK(hipMalloc(&a.partials, 4096 * 200 * 8));
  // random posting words have density 1/2 each: (OR of 4) & (OR of 2) = 0.94*0.75 → too dense; thin them: AND pairs on host? keep: probe only.
  run<256, 0>(a, 256, n, "uncond loads, 4 waves/cu");
  run<256, 0>(a, 512, n, "uncond loads, 8 waves/cu");
  run<512, 0>(a, 256, n, "uncond loads, 8 waves/cu (1 wg)");
  run<1024, 0>(a, 256, n, "uncond loads, 16 waves/cu (1 wg)");
  run<256, 2>(a, 256, n, "uncond, no atomics, 4 waves/cu");
  run<256, 2>(a, 512, n, "uncond, no atomics, 8 waves/cu");
  run<256, 1>(a, 256, n, "masked loads, 4 waves/cu");
  run<256, 1>(a, 512, n, "masked loads, 8 waves/cu");
  run<256, 1>(a, 1024, n, "masked loads, 16 waves/cu");
  run<1024, 1>(a, 256, n, "masked loads, 16 waves/cu (1 wg)");
  run<256, 3>(a, 512, n, "masked, no atomics, 8 waves/cu");
  return 0;
}
