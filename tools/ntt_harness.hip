// Stand-alone timing of the two launches of the 17-column extension at n = 2^20 (tools only): ntt_pass_kernel<true, 10>
// as the first pass of three prescaled coset transforms per column and ntt_final_kernel<true> as their last pass, with
// the library's launch shapes and tables of the right sizes but arbitrary contents (the time does not depend on the
// values).  Compiles in a minute where the library takes several, so that variants of ntt.cuh (-D...) can be compared;
// prints the time per launch and a checksum of the outputs, which must be the same for variants that are supposed to
// compute the same thing.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-D...] -o tools/bin/ntt_harness_X tools/ntt_harness.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../plonk_amd/csrc/ntt.cuh"
using namespace plk;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(uint32_t* p, uint64_t words, uint32_t mask, uint32_t seed) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) p[i] = ((uint32_t)(i * 2654435761u) ^ seed ^ (uint32_t)(i >> 7)) & mask;
}

static int fill(void* p, size_t bytes, uint32_t mask, uint32_t seed) {
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((bytes / 4 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)p, (uint64_t)(bytes / 4), mask, seed);
  CHECK(hipGetLastError());
  return 0;
}

int main(int argc, char** argv) {
  const int log_n = 20, cols = argc > 1 ? atoi(argv[1]) : 17, cosets = 3, slots = cols * cosets;
  const uint64_t N = 1ull << log_n;
  uint32_t *master, *master_sh, *full, *prescale;
  CHECK(hipMalloc(&master, 1024 * TW_WORDS * 4)); CHECK(hipMalloc(&master_sh, 2048 * TW_WORDS * 4));
  CHECK(hipMalloc(&full, N * TW_WORDS * 4)); CHECK(hipMalloc(&prescale, 3 * N * TW_WORDS * 4));
  // twiddle records: nine 29-bit limbs (top limb small), three words of padding
  if (fill(master, 1024 * TW_WORDS * 4, 0x0fffffff, 1) || fill(master_sh, 2048 * TW_WORDS * 4, 0x0fffffff, 2) || fill(full, N * TW_WORDS * 4, 0x0fffffff, 3) ||
      fill(prescale, 3 * N * TW_WORDS * 4, 0x0fffffff, 4)) return 1;
  // argv[2]: bytes (a multiple of 128) by which consecutive columns and consecutive slots of the intermediate buffer are
  // skewed against each other, to see what equal offsets in equally aligned buffers cost in the caches
  const size_t pad = argc > 2 ? (size_t)atol(argv[2]) : 0;
  std::vector<uint64_t*> in(cols), out(cols);
  for (int b = 0; b < cols; ++b) {
    char* raw; CHECK(hipMalloc(&raw, N * 32 + cols * pad)); in[b] = (uint64_t*)(raw + b * pad);
    CHECK(hipMalloc(&out[b], 3 * N * 32));
    if (fill(in[b], N * 32, 0x3fffffff, 10 + b)) return 1;        // below 2^254 < p: canonical elements
  }
  uint64_t* tmp; CHECK(hipMalloc(&tmp, (size_t)slots * (N * 32 + pad)));
  CHECK(hipFuncSetAttribute((const void*)ntt_pass_kernel<true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CHECK(hipFuncSetAttribute((const void*)ntt_final_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));

  NttPassArgs a{};
  a.tab.master = master; a.tab.master_sh = master_sh; a.tab.tw_full[0] = full; a.tab.tw_hi = a.tab.tw_hi_scaled = a.tab.tw_lo = master; a.tab.lo_bits = 10;
  a.n_in = N; a.log_n = log_n; a.log_r = 10; a.log_inner = 10; a.log_c = 1; a.first = 1; a.use_scaled = 1; a.pass = 0; a.cosets = cosets; a.prescale = prescale;
  a.slots = (uint32_t)slots; a.log_group = 1;
  for (int b = 0; b < cols; ++b) { a.batch.in[b] = in[b]; a.batch.kind[b] = NTT_IN_FIELD; }
  a.batch.out[0] = tmp; a.batch.stride_out = N * 4 + pad / 8; a.batch.stride_in = 0;
  const unsigned tiles = 1u << (log_n - 10 - 1);

  NttFinalArgs f{};
  f.tab = a.tab; f.n_in = N; f.log_n = log_n; f.log_r = 10; f.log_r1 = 10; f.log_r2 = 0; f.log_c = 1; f.first = 0; f.cosets = cosets; f.prescale = prescale;
  f.coset_stride = N * 4;
  f.batch.in[0] = tmp; f.batch.stride_in = N * 4 + pad / 8; f.batch.kind[0] = NTT_IN_FIELD;
  for (int b = 0; b < cols; ++b) f.batch.out[b] = out[b];
  f.batch.stride_out = 0;

  hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  float sp = 0, sf = 0;
  const int reps = 6;
  for (int r = 0; r < reps + 2; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((ntt_pass_kernel<true, 10>), dim3(tiles * slots), dim3(NTT_THREADS), NTT_LDS_BYTES, 0, a);
    CHECK(hipEventRecord(e1, 0));
    hipLaunchKernelGGL(ntt_final_kernel<true>, dim3(tiles, slots), dim3(NTT_THREADS), NTT_LDS_BYTES, 0, f);
    CHECK(hipEventRecord(e2, 0)); CHECK(hipEventSynchronize(e2));
    float m1, m2; CHECK(hipEventElapsedTime(&m1, e0, e1)); CHECK(hipEventElapsedTime(&m2, e1, e2));
    if (r >= 2) { sp += m1; sf += m2; }
  }
  std::vector<uint64_t> h(3 * N * 4);
  uint64_t ck = 0;
  for (int b = 0; b < cols; b += 8) {
    CHECK(hipMemcpy(h.data(), out[b], 3 * N * 32, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); i += 97) ck = ck * 0x100000001b3ull ^ h[i];
  }
  hipFuncAttributes fa, fb; CHECK(hipFuncGetAttributes(&fa, (const void*)ntt_pass_kernel<true, 10>)); CHECK(hipFuncGetAttributes(&fb, (const void*)ntt_final_kernel<true>));
  const double elems = (double)slots * N;
  printf("%d columns x 3 cosets, 2^20 points: first pass %.4f ms (%.1f G products/s, regs %d, scratch %zu B)  last pass %.4f ms (%.1f G products/s, regs %d, scratch %zu B)  checksum %016llx\n",
         cols, sp / reps, 6.25 * elems / (sp / reps * 1e-3) * 1e-9, fa.numRegs, (size_t)fa.localSizeBytes, sf / reps, 4.25 * elems / (sf / reps * 1e-3) * 1e-9, fb.numRegs,
         (size_t)fb.localSizeBytes, (unsigned long long)ck);
  return 0;
}
