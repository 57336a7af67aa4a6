// Variants of the 29-bit-limb Montgomery product, timed in isolation (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../plonk_amd/csrc/fr29.cuh"
using namespace plk;

#define PIN(x) asm volatile("" : "+v"(x))

// V1: one strictly sequential accumulation chain (no compiler re-association)
__device__ __forceinline__ F9 mont29_chain(const F9& a, const F9& b) {
  uint64_t acc = 0; uint32_t m[9]; F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; PIN(acc); }
#pragma unroll
    for (int i = 0; i < k; ++i) { acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i); PIN(acc); }
    m[k] = (0u - (uint32_t)acc) & M29;
    acc += m[k]; PIN(acc);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; PIN(acc); }
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i); PIN(acc); }
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}

// V2: sequential chain, carry between columns through 32-bit funnel shifts
__device__ __forceinline__ F9 mont29_chain32(const F9& a, const F9& b) {
  uint64_t acc = 0; uint32_t m[9]; F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; PIN(acc); }
#pragma unroll
    for (int i = 0; i < k; ++i) { acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i); PIN(acc); }
    m[k] = (0u - (uint32_t)acc) & M29;
    acc += (uint64_t)m[k] * 1u; PIN(acc);
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 29), nhi = hi >> 29;
    acc = ((uint64_t)nhi << 32) | nlo;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; PIN(acc); }
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i); PIN(acc); }
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    r.l[k - 9] = lo & M29;
    uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 29), nhi = hi >> 29;
    acc = ((uint64_t)nhi << 32) | nlo;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}

// V3: two products in lockstep, each a strictly sequential chain (dependent distance 2)
__device__ __forceinline__ void mont29_x2(const F9& a1, const F9& b1, const F9& a2, const F9& b2, F9& r1, F9& r2) {
  uint64_t acc1 = 0, acc2 = 0; uint32_t m1[9], m2[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc1 += (uint64_t)a1.l[i] * b1.l[k - i]; acc2 += (uint64_t)a2.l[i] * b2.l[k - i]; PIN(acc1); PIN(acc2); }
#pragma unroll
    for (int i = 0; i < k; ++i) { acc1 += (uint64_t)m1[i] * f9c::kp_limb29(1, k - i); acc2 += (uint64_t)m2[i] * f9c::kp_limb29(1, k - i); PIN(acc1); PIN(acc2); }
    m1[k] = (0u - (uint32_t)acc1) & M29; m2[k] = (0u - (uint32_t)acc2) & M29;
    acc1 += m1[k]; acc2 += m2[k]; PIN(acc1); PIN(acc2);
    acc1 >>= 29; acc2 >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc1 += (uint64_t)a1.l[i] * b1.l[k - i]; acc2 += (uint64_t)a2.l[i] * b2.l[k - i]; PIN(acc1); PIN(acc2); }
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc1 += (uint64_t)m1[i] * f9c::kp_limb29(1, k - i); acc2 += (uint64_t)m2[i] * f9c::kp_limb29(1, k - i); PIN(acc1); PIN(acc2); }
    r1.l[k - 9] = (uint32_t)acc1 & M29; r2.l[k - 9] = (uint32_t)acc2 & M29;
    acc1 >>= 29; acc2 >>= 29;
  }
  r1.l[8] = (uint32_t)acc1; r2.l[8] = (uint32_t)acc2;
}

template <int V>
__global__ __launch_bounds__(512) void mul_kernel(const uint32_t* tw, int reps, uint32_t* sink) {
  F9 x, w = f9_load_tw(tw, threadIdx.x & 1023);
  for (int i = 0; i < 9; ++i) x.l[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & M29;
  F9 x2 = x; x2.l[0] ^= 5;
  for (int r = 0; r < reps; ++r) {
    if (V == 3) { if (r & 1) continue; F9 o1, o2; mont29_x2(x, w, x2, w, o1, o2); x = o1; x2 = o2; }
    if (V == 0) x = mont29(x, w);
    if (V == 1) x = mont29_chain(x, w);
    if (V == 2) x = mont29_chain32(x, w);
  }
  uint32_t acc = 0;
  for (int i = 0; i < 9; ++i) acc ^= x.l[i] ^ (V == 3 ? x2.l[i] : 0);
  sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int V> int run(const char* name, const uint32_t* tw, uint32_t* sink, uint32_t* host, int wps) {
  const int blocks = 256 * wps / 2, reps = 256;   // 512 threads = 2 waves per SIMD per block
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mul_kernel<V>, dim3(blocks), dim3(512), 0, 0, tw, reps, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(mul_kernel<V>, dim3(blocks), dim3(512), 0, 0, tw, reps, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(host, sink, 512 * 4, hipMemcpyDeviceToHost));
  uint32_t h = 0; for (int i = 0; i < 512; ++i) h = h * 31 + host[i];
  const double muls = (double)blocks * 512 * reps;
  printf("%-28s waves/SIMD=%d  %8.3f ms  %7.1f G mul/s  checksum %08x\n", name, wps, ms, muls / (ms * 1e-3) * 1e-9, h);
  return 0;
}

int main() {
  uint32_t *tw, *sink; static uint32_t host[512];
  CHECK(hipMalloc(&tw, 1024 * 48));
  static uint32_t init[1024 * 12];
  for (int i = 0; i < 1024 * 12; ++i) init[i] = (i * 2654435761u) & M29;
  CHECK(hipMemcpy(tw, init, sizeof init, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&sink, 256 * 8 * 512 * 4));
  for (int wps : {2, 4, 8}) {
    if (run<0>("compiler schedule", tw, sink, host, wps)) return 1;
    if (run<1>("single chain", tw, sink, host, wps)) return 1;
    if (run<2>("single chain + 32-bit shifts", tw, sink, host, wps)) return 1;
    if (run<3>("two chains in lockstep", tw, sink, host, wps)) return 1;
  }
  return 0;
}
