// Ablation of one in-LDS transform stage (tools only): how much of a stage is the Montgomery
// product, the LDS traffic, the butterfly adds and the renormalisation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../plonk_amd/csrc/ntt.cuh"
using namespace plk;

template <int MODE>
__global__ __launch_bounds__(NTT_THREADS) void stage_kernel(const uint32_t* master, int log_r, int log_c, int reps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Tile t(lds_raw);
  for (int e = threadIdx.x; e < NTT_TILE; e += NTT_THREADS) {
    F9 v; for (int i = 0; i < 9; ++i) v.l[i] = (e * 2654435761u + i * 40503u + blockIdx.x) & M29;
    t.put(e, v);
  }
  __syncthreads();
  const int total = 1 << (log_r + log_c - 1);
  F9 keep; for (int i = 0; i < 9; ++i) keep.l[i] = threadIdx.x + i;
  for (int rep = 0; rep < reps; ++rep)
  for (int lh = 1; lh < log_r; ++lh) {
    const int h = 1 << lh;
    const bool renorm = (lh & 1) != 0;
    for (int b = threadIdx.x; b < total; b += NTT_THREADS) {
      const int c = b & ((1 << log_c) - 1);
      const int q = b >> log_c;
      const int j = q & (h - 1);
      const int i0 = ((q >> lh) << (lh + 1)) + j;
      const int e0 = (i0 << log_c) + c, e1 = ((i0 + h) << log_c) + c;
      F9 x, y;
      if (MODE & 1) { x = t.get(e0); y = t.get(e1); } else { x = keep; y = keep; y.l[0] += e1; }
      if (MODE & 2) y = mont29(y, f9_load_tw(master, (size_t)j << (NTT_MASTER_LOG - 1 - lh)));
      else { for (int i = 0; i < 9; ++i) y.l[i] &= M29; }
      F9 u = f9_add(x, y), v = f9_sub_q2(x, y);
      if ((MODE & 4) && renorm) { f9_normalize(u); f9_normalize(v); }
      if (!(MODE & 4)) { for (int i = 0; i < 9; ++i) { u.l[i] &= M29; v.l[i] &= M29; } }
      if (MODE & 1) { t.put(e0, u); t.put(e1, v); } else { for (int i = 0; i < 9; ++i) keep.l[i] ^= u.l[i] + v.l[i]; }
    }
    if (MODE & 8) __syncthreads();
  }
  uint32_t acc = 0;
  for (int i = 0; i < 9; ++i) acc ^= keep.l[i];
  if (MODE & 1) { F9 v = t.get(threadIdx.x); for (int i = 0; i < 9; ++i) acc ^= v.l[i]; }
  if (acc == 0x12345678u) sink[0] = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE> int run(const char* name, const uint32_t* master, uint32_t* sink, int log_r, int log_c) {
  CHECK(hipFuncSetAttribute((const void*)stage_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  const int blocks = 256 * 2 * 8, reps = 4;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stage_kernel<MODE>, dim3(blocks), dim3(NTT_THREADS), NTT_LDS_BYTES, 0, master, log_r, log_c, reps, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(stage_kernel<MODE>, dim3(blocks), dim3(NTT_THREADS), NTT_LDS_BYTES, 0, master, log_r, log_c, reps, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double stages = (double)blocks * reps * (log_r - 1);
  const double bfly = stages * (1 << (log_r + log_c - 1));
  printf("%-34s %8.3f ms  %7.2f ns per tile-stage  %6.1f G butterflies/s\n", name, ms, ms * 1e6 / stages * 512 / blocks * 1.0, bfly / (ms * 1e-3) * 1e-9);
  return 0;
}

int main() {
  uint32_t *master, *sink;
  CHECK(hipMalloc(&master, 1024 * TW_WORDS * 4)); CHECK(hipMemset(master, 0x11, 1024 * TW_WORDS * 4));
  CHECK(hipMalloc(&sink, 64));
  const int log_r = 7, log_c = 4;
  if (run<15>("full (lds+mul+renorm+barrier)", master, sink, log_r, log_c)) return 1;
  if (run<7>("no barrier", master, sink, log_r, log_c)) return 1;
  if (run<13>("no mul", master, sink, log_r, log_c)) return 1;
  if (run<11>("no renorm", master, sink, log_r, log_c)) return 1;
  if (run<14>("no lds (registers)", master, sink, log_r, log_c)) return 1;
  if (run<2>("mul only", master, sink, log_r, log_c)) return 1;
  if (run<9>("lds + add/sub only", master, sink, log_r, log_c)) return 1;
  if (run<0>("nothing (loop+addr)", master, sink, log_r, log_c)) return 1;
  return 0;
}
