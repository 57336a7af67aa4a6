// What the FETCH_SIZE counter reports for 32-byte gathers: every lane reads one 32-byte element at a random index of a
// 128 MiB array (the access pattern of the grand product's v[sigma[j]]), or at its own index (streaming), and writes
// 32 bytes.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate runs) and compare the counted
// bytes with the bytes asked for:  ./ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void gather_kernel(const uint4* __restrict__ table, const uint64_t* __restrict__ idx, uint4* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t j = idx[i];
  const uint4 a = table[2 * j], b = table[2 * j + 1];
  out[2 * i] = make_uint4(a.x ^ b.x, a.y, a.z, a.w);
  out[2 * i + 1] = b;
}

int main() {
  const uint64_t n = 1ull << 22;                       // 2^22 elements of 32 bytes = 128 MiB
  uint4 *table, *out; uint64_t *idx_rand, *idx_own;
  CHECK(hipMalloc(&table, n * 32)); CHECK(hipMalloc(&out, n * 32));
  CHECK(hipMalloc(&idx_rand, n * 8)); CHECK(hipMalloc(&idx_own, n * 8));
  CHECK(hipMemset(table, 1, n * 32));
  std::vector<uint64_t> h(n);
  for (uint64_t i = 0; i < n; ++i) h[i] = i;
  CHECK(hipMemcpy(idx_own, h.data(), n * 8, hipMemcpyHostToDevice));
  uint64_t s = 88172645463325252ull;                   // Fisher-Yates with xorshift64
  for (uint64_t i = n - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint64_t j = s % (i + 1); const uint64_t t = h[i]; h[i] = h[j]; h[j] = t; }
  CHECK(hipMemcpy(idx_rand, h.data(), n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    const uint64_t* idx = which ? idx_rand : idx_own;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(gather_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, table, idx, out, n);
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%s: %.3f ms, asked for %.1f MB of elements + %.1f MB of indices in, %.1f MB out (launches %d-%d of gather_kernel)\n",
           which ? "random 32-byte gathers" : "own index (streaming)", best, n * 32 / 1e6, n * 8 / 1e6, n * 32 / 1e6, which * 3, which * 3 + 2);
  }
  return 0;
}
