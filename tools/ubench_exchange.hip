// Exchange of the four elements a thread holds between two radix-4 register steps of the tile transform (tools only):
//   LDS     the form the library uses: four 36-byte stores, workgroup barrier, four loads at the next step's rows;
//   SWAP    a 4 x 4 transpose among lanes l, l+16, l+32, l+48 of one wave by the gfx950 lane-swap instructions
//           (v_permlane32_swap, v_permlane16_swap): 4 instructions per limb, no LDS, no barrier;
//   BPERM   the same transpose by ds_bpermute (__shfl) with per-lane source selection.
// Each repetition runs the four butterfly products of a radix-4 step (mont29_x2 twice) on the exchanged values,
// so the exchange is timed beside the arithmetic it has to hide behind.  NONE = products only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../plonk_amd/csrc/ntt.cuh"
using namespace plk;

enum { NONE = 0, LDS = 1, SWAP = 2, BPERM = 3 };

__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {   // upper 32 lanes of a <-> lower 32 lanes of b
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {   // odd rows of 16 lanes of a <-> even rows of b
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
}

template <int MODE>
__global__ __launch_bounds__(NTT_THREADS, 4) void exch_kernel(const uint32_t* master, int reps, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Tile t(lds_raw);
  F9 x[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 9; ++i) x[k].l[i] = (threadIdx.x * 2654435761u + (k * 9 + i) * 40503u + blockIdx.x) & M29;
  const F9 w1 = f9_load_tw(master, threadIdx.x & 63), w2 = f9_load_tw(master, 64 + (threadIdx.x & 63)), w3 = f9_load_tw(master, 128 + (threadIdx.x & 63));
  const int lane = threadIdx.x & 63;
  for (int rep = 0; rep < reps; ++rep) {
    F9 t1, t3, s2, s3;
    mont29_x2(x[1], w1, x[3], w1, t1, t3);
    const F9 a0 = f9_add(x[0], t1), a1 = f9_sub_q2(x[0], t1), a2 = f9_add(x[2], t3), a3 = f9_sub_q2(x[2], t3);
    mont29_x2(a2, w2, a3, w3, s2, s3);
    x[0] = f9_add(a0, s2); x[2] = f9_sub_q2(a0, s2); x[1] = f9_add(a1, s3); x[3] = f9_sub_q2(a1, s3);
    f9_normalize(x[0]); f9_normalize(x[1]); f9_normalize(x[2]); f9_normalize(x[3]);
    if (MODE == LDS) {
      // rows 4 g .. 4 g + 3 out, rows g, g + 128, g + 256, g + 384 back in (a 512-row, one-column-per-thread tile)
      const int e0 = threadIdx.x * 4;
      t.put(e0, x[0]); t.put(e0 + 1, x[1]); t.put(e0 + 2, x[2]); t.put(e0 + 3, x[3]);
      __syncthreads();
      x[0] = t.get(threadIdx.x); x[1] = t.get(threadIdx.x + 512); x[2] = t.get(threadIdx.x + 1024); x[3] = t.get(threadIdx.x + 1536);
      __syncthreads();
    } else if (MODE == SWAP) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        swap32(x[0].l[i], x[2].l[i]); swap32(x[1].l[i], x[3].l[i]);     // 2 x 2 blocks across lanes +-32
        swap16(x[0].l[i], x[1].l[i]); swap16(x[2].l[i], x[3].l[i]);     // 2 x 2 inside them across lanes +-16
      }
    } else if (MODE == BPERM) {
      const int k = lane >> 4;                                          // which of the four lanes of my group I am
#pragma unroll
      for (int d = 1; d < 4; ++d) {
        const int src = lane ^ (16 * d), slot = k ^ d;                  // I send element `slot`, and receive my partner's element k into it
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const uint32_t send = slot == 0 ? x[0].l[i] : (slot == 1 ? x[1].l[i] : (slot == 2 ? x[2].l[i] : x[3].l[i]));
          const uint32_t got = __shfl(send, src);
          x[0].l[i] = slot == 0 ? got : x[0].l[i]; x[1].l[i] = slot == 1 ? got : x[1].l[i];
          x[2].l[i] = slot == 2 ? got : x[2].l[i]; x[3].l[i] = slot == 3 ? got : x[3].l[i];
        }
      }
    }
  }
  uint32_t acc = 0;
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 9; ++i) acc ^= x[k].l[i];
  sink[blockIdx.x * NTT_THREADS + threadIdx.x] = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE> int run(const char* name, const uint32_t* tw, uint32_t* sink) {
  const int blocks = 512, reps = 128;      // two workgroups of 512 threads per CU, as the transform kernels
  CHECK(hipFuncSetAttribute((const void*)exch_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(exch_kernel<MODE>, dim3(blocks), dim3(NTT_THREADS), NTT_LDS_BYTES, 0, tw, reps, sink);
  CHECK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(exch_kernel<MODE>, dim3(blocks), dim3(NTT_THREADS), NTT_LDS_BYTES, 0, tw, reps, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double steps = (double)blocks * NTT_THREADS * reps;     // radix-4 register steps (4 elements, 4 products each)
  printf("%-44s %8.3f ms  %6.1f G products/s\n", name, best, 4.0 * steps / (best * 1e-3) * 1e-9);
  return 0;
}

int main() {
  uint32_t *tw, *sink;
  static uint32_t init[1024 * 12];
  for (int i = 0; i < 1024 * 12; ++i) init[i] = (i * 2654435761u) & M29;
  CHECK(hipMalloc(&tw, sizeof init));
  CHECK(hipMemcpy(tw, init, sizeof init, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&sink, 512 * NTT_THREADS * 4));
  if (run<NONE>("radix-4 step, no exchange", tw, sink)) return 1;
  if (run<LDS>("radix-4 step + exchange through LDS", tw, sink)) return 1;
  if (run<SWAP>("radix-4 step + transpose by lane swaps", tw, sink)) return 1;
  if (run<BPERM>("radix-4 step + transpose by ds_bpermute", tw, sink)) return 1;
  return 0;
}
