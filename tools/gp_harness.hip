// Stand-alone timing of the grand product's first kernel (tools only): compiles in seconds where the library takes
// minutes, so that variants of gp_local_z_kernel (-DPLK_GP_WAVES=..., -DPLK_GP_HOIST=1, ...) can be compared; prints
// the time per launch and a checksum of z, which must be the same for every variant.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-D...] -o tools/bin/gp_harness_X tools/gp_harness.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../plonk_amd/csrc/grand_product.cuh"
using namespace plk;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint64_t sm = 12345;
static uint64_t splitmix() { uint64_t z = (sm += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
  const int log_n = argc > 1 ? atoi(argv[1]) : 20;
  const uint64_t n = 1ull << log_n;
  std::vector<uint64_t> h(n * 4), hs(n);
  GpCols cols{};
  for (int k = 0; k < 4; ++k) {
    for (uint64_t i = 0; i < n; ++i) { for (int l = 0; l < 4; ++l) h[4 * i + l] = splitmix(); h[4 * i + 3] &= (1ull << 61) - 1; hs[i] = splitmix() % n; }
    uint64_t *dv, *ds;
    CHECK(hipMalloc(&dv, n * 32)); CHECK(hipMalloc(&ds, n * 8));
    CHECK(hipMemcpy(dv, h.data(), n * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(ds, hs.data(), n * 8, hipMemcpyHostToDevice));
    cols.v[k] = dv; cols.sigma[k] = ds;
  }
  Fe beta, gamma;
  for (int i = 0; i < 8; ++i) { beta.l[i] = (uint32_t)splitmix(); gamma.l[i] = (uint32_t)splitmix(); }
  beta.l[7] &= 0x0fffffff; gamma.l[7] &= 0x0fffffff;
  GpConsts k9;
  { Fe t = beta; for (int i = 0; i < 10; ++i) t = fe_double(t); k9.beta_sigma = f9_from_fe(t);
    Fe c = fe_r2(); for (int i = 0; i < 10; ++i) c = fe_double(c); k9.beta_index = f9_from_fe(fe_mul(beta, c));
    t = gamma; for (int i = 0; i < 5; ++i) t = fe_double(t); k9.gamma = f9_from_fe(t);
    t = fe_one(); for (int i = 0; i < 5; ++i) t = fe_double(t); k9.one = f9_from_fe(t); }
  const size_t nb = (n + GP_BLOCK_ROWS - 1) / GP_BLOCK_ROWS;
  uint64_t *z, *ratio; GpStatus* st;
  CHECK(hipMalloc(&z, (n + 1) * 32)); CHECK(hipMalloc(&ratio, nb * 32)); CHECK(hipMalloc(&st, sizeof(GpStatus)));
  CHECK(hipMemset(st, 0xff, sizeof(GpStatus)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0;
  const int reps = 20;
  for (int r = 0; r < reps + 3; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gp_local_z_kernel, dim3((unsigned)nb), dim3(GP_THREADS), 0, 0, cols, n, k9, z + 4, z, ratio, st);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 3) { sum += ms; if (ms < best) best = ms; }
  }
  std::vector<uint64_t> out((n + 1) * 4);
  CHECK(hipMemcpy(out.data(), z, (n + 1) * 32, hipMemcpyDeviceToHost));
  uint64_t ck = 0;
  for (size_t i = 0; i < out.size(); ++i) ck = ck * 0x100000001b3ull ^ out[i];
  hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)gp_local_z_kernel));
  printf("n=2^%d  gp_local_z_kernel: %.4f ms mean, %.4f ms best; regs %d, scratch %zu B; checksum %016llx\n", log_n, sum / reps, best, fa.numRegs,
         (size_t)fa.localSizeBytes, (unsigned long long)ck);
  return 0;
}
