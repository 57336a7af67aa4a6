// Probe of the mont256 compiler trap (fr29.cuh): the product of two freshly cut operands (top limbs known to the
// compiler to fit 24 bits) against fe_mul, with the empty asm fences of mont256 in place or, built with
// -DPLK_MONT256_NO_FENCE, without them.  Prints the number of differing results and the first few.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../plonk_amd/csrc/fr29.cuh"
using namespace plk;

__global__ void k(const uint64_t* a, const uint64_t* b, uint64_t* want, uint64_t* got, uint32_t* limbs, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe x = fe_load(a, i), y = fe_load(b, i);
  fe_store(want, i, fe_mul(x, y));
  const F9 r = mont256(f9_from_fe(x), f9_from_fe(y));
  for (int j = 0; j < 9; ++j) limbs[9 * i + j] = r.l[j];
  fe_store(got, i, fe_reduce_f9(r));
}

int main() {
  const int n = 1 << 14;
  std::vector<uint64_t> a(4 * n), b(4 * n);
  uint64_t s = 7;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return s; };
  for (int i = 0; i < n; ++i) for (int j = 0; j < 4; ++j) { a[4 * i + j] = rnd(); b[4 * i + j] = rnd(); if (j == 3) { a[4 * i + j] &= 0x3fffffffffffffffull; b[4 * i + j] &= 0x3fffffffffffffffull; } }
  uint64_t *da, *db, *dw, *dg; uint32_t* dl;
  hipMalloc(&da, 32 * n); hipMalloc(&db, 32 * n); hipMalloc(&dw, 32 * n); hipMalloc(&dg, 32 * n); hipMalloc(&dl, 36 * n);
  hipMemcpy(da, a.data(), 32 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 32 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dw, dg, dl, n);
  std::vector<uint64_t> w(4 * n), g(4 * n); std::vector<uint32_t> l(9 * n);
  hipMemcpy(w.data(), dw, 32 * n, hipMemcpyDeviceToHost); hipMemcpy(g.data(), dg, 32 * n, hipMemcpyDeviceToHost);
  hipMemcpy(l.data(), dl, 36 * n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    bool same = true;
    for (int j = 0; j < 4; ++j) same = same && w[4 * i + j] == g[4 * i + j];
    if (!same) {
      if (bad < 3) {
        // the same product on the host, limb by limb
        Fe x, y;
        for (int j = 0; j < 4; ++j) { x.l[2 * j] = (uint32_t)a[4 * i + j]; x.l[2 * j + 1] = (uint32_t)(a[4 * i + j] >> 32); y.l[2 * j] = (uint32_t)b[4 * i + j]; y.l[2 * j + 1] = (uint32_t)(b[4 * i + j] >> 32); }
        const F9 h = mont256(f9_from_fe(x), f9_from_fe(y));
        printf("element %d: limbs differing from the host's:", i);
        for (int j = 0; j < 9; ++j) if (h.l[j] != l[9 * i + j]) printf(" [%d] host %08x device %08x", j, h.l[j], l[9 * i + j]);
        printf("\n");
      }
      ++bad;
    }
  }
  printf("%d of %d products differ from fe_mul\n", bad, n);
  return 0;
}
