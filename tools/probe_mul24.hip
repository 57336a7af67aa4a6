// Probe: the 64-bit product of two values the compiler knows to fit 24 bits is compiled to
// v_mul_u32_u24 + v_mul_hi_u32_u24 on gfx950.  Prints how many of 2^16 such products differ from the host's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* x, uint64_t* out, uint32_t* hi, uint32_t* lo, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = x[2 * i] >> 8, b = x[2 * i + 1] >> 8;
  out[i] = (uint64_t)a * b;
  hi[i] = __umul24(a, b);            // low 32 bits by the 24-bit multiplier
  lo[i] = 0;
}
int main() {
  const int n = 1 << 16;
  std::vector<uint32_t> x(2 * n);
  uint64_t s = 99;
  for (auto& v : x) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)(s >> 32); }
  uint32_t *dx, *dh, *dl; uint64_t* dout;
  hipMalloc(&dx, 8 * n); hipMalloc(&dout, 8 * n); hipMalloc(&dh, 4 * n); hipMalloc(&dl, 4 * n);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, dh, dl, n);
  std::vector<uint64_t> out(n); std::vector<uint32_t> h(n);
  hipMemcpy(out.data(), dout, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h.data(), dh, 4 * n, hipMemcpyDeviceToHost);
  int bad = 0, bad_lo = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t want = (uint64_t)(x[2 * i] >> 8) * (x[2 * i + 1] >> 8);
    if (out[i] != want) { ++bad; if (shown++ < 3) std::printf("a %06x b %06x want %012llx got %012llx\n", x[2 * i] >> 8, x[2 * i + 1] >> 8, (unsigned long long)want, (unsigned long long)out[i]); }
    if (h[i] != (uint32_t)want) ++bad_lo;
  }
  std::printf("64-bit products of 24-bit factors: %d of %d wrong; low words by the 24-bit multiplier: %d wrong\n", bad, n, bad_lo);
  return 0;
}
