// Which occupancy point serves the in-tile steps of the transform kernels best (tools only)?
//
//   A  the library's form: 2048-element tile, 512 threads, two stages per LDS round trip as a radix-4 step in
//      registers (4 elements per thread, the four products as two interleaved pairs, shoup29_x2), at most 128
//      registers: 4 waves per SIMD, 2 workgroups per CU.
//   B  1024-element tile, 512 threads, ONE stage per LDS round trip (2 elements per thread, one product, a single
//      multiply-add chain), at most 64 registers: 8 waves per SIMD, 4 workgroups per CU.
//   C  1024-element tile, 256 threads, the radix-4 step of A with its products one after another (single chains),
//      forced to 64 registers: 8 waves per SIMD, 8 workgroups per CU - does the compiler fit it without spilling?
//
// All three run the eight stages lh = 2 .. 9 of a 2^10-point transform `reps` times on a tile resident in LDS, with
// the twiddle pairs (w, floor(w 2^261 / p)) in LDS as in the library.  Figure of merit: element-stages per second
// (elements in the tile x stages / time); one product is spent per two element-stages in every form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../plonk_amd/csrc/ntt.cuh"
using namespace plk;

#define PIN(x) asm volatile("" : "+v"(x))

// one product by a constant with a precomputed quotient: shoup29_x2 of fr29.cuh with a single chain
__device__ __forceinline__ F9 shoup29_one(const F9& a, const F9& w, const F9& wq) {
  uint64_t acc = 0;
  uint32_t q[9];
#pragma unroll
  for (int i = 0; i <= 7; ++i) { acc += (uint64_t)a.l[i] * wq.l[7 - i]; PIN(acc); }
  acc >>= 29;
#pragma unroll
  for (int i = 0; i <= 8; ++i) { acc += (uint64_t)a.l[i] * wq.l[8 - i]; PIN(acc); }
  acc >>= 29;
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc += (uint64_t)a.l[i] * wq.l[k - i]; PIN(acc); }
    q[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  q[8] = (uint32_t)acc;
  acc = 0;
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (uint64_t)a.l[i] * w.l[k - i]; PIN(acc); }
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (uint64_t)q[i] * f9c::pbar_limb(k - i); PIN(acc); }
    r.l[k] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  return r;
}

#ifndef UB_PAIRS
#define UB_PAIRS 64
#endif
constexpr int PAIRS = UB_PAIRS;   // twiddle pairs in LDS: records 0..PAIRS-1 w, PAIRS..2 PAIRS-1 wq (32: four 1024-element workgroups fit a CU)

template <int ELEMS, int THREADS>
__device__ __forceinline__ void fill(const Tile& t, const Tile& tw, const uint32_t* master_sh) {
  for (int e = threadIdx.x; e < ELEMS; e += THREADS) {
    F9 v;
    for (int i = 0; i < 9; ++i) v.l[i] = (e * 2654435761u + i * 40503u + blockIdx.x) & M29;
    t.put(e, v);
  }
  for (int j = threadIdx.x; j < 2 * PAIRS; j += THREADS) tw.put((j & 1) * PAIRS + (j >> 1), f9_load_tw(master_sh, j));
  __syncthreads();
}

template <int ELEMS, int THREADS>
__device__ __forceinline__ void drain(const Tile& t, uint32_t* sink) {
  uint32_t acc = 0;
  for (int e = threadIdx.x; e < ELEMS; e += THREADS) { const F9 v = t.get(e); for (int i = 0; i < 9; ++i) acc ^= v.l[i]; }
  if (acc == 0x12345678u) sink[0] = acc;
}

// A: radix-4 paired steps, two interleaved products
__global__ __launch_bounds__(512, 4) void form_a(const uint32_t* master_sh, int reps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int ELEMS = 2048, LOG_C = 1, LOG_R = 10;
  const Tile t(lds_raw, ELEMS), tw(lds_raw + (size_t)ELEMS * 36, 2 * PAIRS);
  fill<ELEMS, 512>(t, tw, master_sh);
  const int groups = ELEMS / 4;
  for (int rep = 0; rep < reps; ++rep)
    for (int lh = 2; lh + 1 < LOG_R; lh += 2) {
      const int h = 1 << lh;
      for (int g = threadIdx.x; g < groups; g += 512) {
        const int c = g & ((1 << LOG_C) - 1), q = g >> LOG_C;
        const int j = q & (h - 1);
        const int i0 = ((q >> lh) << (lh + 2)) + j;
        const int e0 = (i0 << LOG_C) + c, step = h << LOG_C;
        const F9 x0 = t.get(e0), x1 = t.get(e0 + step), x2 = t.get(e0 + 2 * step), x3 = t.get(e0 + 3 * step);
        F9 t1, t3, s2, s3;
        { const int k = j & (PAIRS - 1); const F9 w = tw.get(k), wq = tw.get(PAIRS + k); shoup29_x2(x1, w, wq, x3, w, wq, t1, t3); }
        const F9 a0 = f9_add(x0, t1), a1 = f9_sub_q2(x0, t1), a2 = f9_add(x2, t3), a3 = f9_sub_q2(x2, t3);
        {
          const int k2 = (j * 3) & (PAIRS - 1), k3 = (j * 3 + h) & (PAIRS - 1);
          const F9 w2 = tw.get(k2), q2 = tw.get(PAIRS + k2), w3 = tw.get(k3), q3 = tw.get(PAIRS + k3);
          shoup29_x2(a2, w2, q2, a3, w3, q3, s2, s3);
        }
        F9 y0 = f9_add(a0, s2), y2 = f9_sub_q2(a0, s2), y1 = f9_add(a1, s3), y3 = f9_sub_q2(a1, s3);
        f9_normalize(y0); f9_normalize(y1); f9_normalize(y2); f9_normalize(y3);
        t.put(e0, y0); t.put(e0 + step, y1); t.put(e0 + 2 * step, y2); t.put(e0 + 3 * step, y3);
      }
      __syncthreads();
    }
  drain<ELEMS, 512>(t, sink);
}

// B: one stage per LDS round trip, one product per thread and stage
__global__ __launch_bounds__(512, 8) void form_b(const uint32_t* master_sh, int reps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int ELEMS = 1024, LOG_R = 10;
  const Tile t(lds_raw, ELEMS), tw(lds_raw + (size_t)ELEMS * 36, 2 * PAIRS);
  fill<ELEMS, 512>(t, tw, master_sh);
  for (int rep = 0; rep < reps; ++rep)
    for (int lh = 2; lh < LOG_R; ++lh) {
      const int h = 1 << lh;
      const int b = threadIdx.x;                  // 512 butterflies, one per thread
      const int j = b & (h - 1);
      const int i0 = ((b >> lh) << (lh + 1)) + j;
      const F9 x = t.get(i0), y = t.get(i0 + h);
      const int k = j & (PAIRS - 1);
      const F9 tt = shoup29_one(y, tw.get(k), tw.get(PAIRS + k));
      F9 u = f9_add(x, tt), v = f9_sub_q2(x, tt);
      if (lh & 1) { f9_normalize(u); f9_normalize(v); }     // every second stage, as in the library
      t.put(i0, u); t.put(i0 + h, v);
      __syncthreads();
    }
  drain<ELEMS, 512>(t, sink);
}

// C: the radix-4 step with its four products one after another, 64 registers
__global__ __launch_bounds__(256, 8) void form_c(const uint32_t* master_sh, int reps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int ELEMS = 1024, LOG_R = 10;
  const Tile t(lds_raw, ELEMS), tw(lds_raw + (size_t)ELEMS * 36, 2 * PAIRS);
  fill<ELEMS, 256>(t, tw, master_sh);
  for (int rep = 0; rep < reps; ++rep)
    for (int lh = 2; lh + 1 < LOG_R; lh += 2) {
      const int h = 1 << lh;
      const int q = threadIdx.x;                  // 256 groups of four, one per thread
      const int j = q & (h - 1);
      const int i0 = ((q >> lh) << (lh + 2)) + j;
      const int k = j & (PAIRS - 1), k2 = (j * 3) & (PAIRS - 1), k3 = (j * 3 + h) & (PAIRS - 1);
      F9 a0, a1, a2, a3;
      {
        const F9 x1 = t.get(i0 + h);
        const F9 t1 = shoup29_one(x1, tw.get(k), tw.get(PAIRS + k));
        const F9 x0 = t.get(i0);
        a0 = f9_add(x0, t1); a1 = f9_sub_q2(x0, t1);
      }
      {
        const F9 x3 = t.get(i0 + 3 * h);
        const F9 t3 = shoup29_one(x3, tw.get(k), tw.get(PAIRS + k));
        const F9 x2 = t.get(i0 + 2 * h);
        a2 = f9_add(x2, t3); a3 = f9_sub_q2(x2, t3);
      }
      {
        const F9 s2 = shoup29_one(a2, tw.get(k2), tw.get(PAIRS + k2));
        F9 y0 = f9_add(a0, s2), y2 = f9_sub_q2(a0, s2);
        f9_normalize(y0); f9_normalize(y2);
        t.put(i0, y0); t.put(i0 + 2 * h, y2);
      }
      {
        const F9 s3 = shoup29_one(a3, tw.get(k3), tw.get(PAIRS + k3));
        F9 y1 = f9_add(a1, s3), y3 = f9_sub_q2(a1, s3);
        f9_normalize(y1); f9_normalize(y3);
        t.put(i0 + h, y1); t.put(i0 + 3 * h, y3);
      }
      __syncthreads();
    }
  drain<ELEMS, 256>(t, sink);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class K>
int run(const char* name, K kernel, int threads, int elems, const uint32_t* master_sh, uint32_t* sink) {
  const size_t lds = (size_t)(elems + 2 * PAIRS) * 36;
  CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, (const void*)kernel));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
  const int blocks = 256 * 64 * (2048 / elems), reps = 8;       // the same number of elements for every form
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, master_sh, reps, sink);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, master_sh, reps, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double elem_stages = (double)blocks * elems * 8.0 * reps;
  printf("%-58s regs %3d  spill %4zu B  workgroups/CU %d  %7.3f ms  %7.1f G element-stages/s  %6.1f G products/s\n", name, fa.numRegs,
         (size_t)fa.localSizeBytes, per_cu, best, elem_stages / (best * 1e-3) * 1e-9, elem_stages / 2 / (best * 1e-3) * 1e-9);
  return 0;
}

int main() {
  uint32_t *master_sh, *sink;
  CHECK(hipMalloc(&master_sh, 2 * PAIRS * TW_WORDS * 4));
  CHECK(hipMemset(master_sh, 0x11, 2 * PAIRS * TW_WORDS * 4));
  CHECK(hipMalloc(&sink, 64));
  if (run("A 2048-element tile, radix-4 steps, 2 chains, <=128 regs", form_a, 512, 2048, master_sh, sink)) return 1;
  if (run("B 1024-element tile, radix-2 steps, 1 chain,  <= 64 regs", form_b, 512, 1024, master_sh, sink)) return 1;
  if (run("C 1024-element tile, radix-4 steps, 1 chain,  <= 64 regs", form_c, 256, 1024, master_sh, sink)) return 1;
  return 0;
}
