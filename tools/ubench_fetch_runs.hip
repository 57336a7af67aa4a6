// What the FETCH_SIZE counter reports for reads in short contiguous runs at a large stride - the access pattern of a
// transform pass whose tile holds C neighbouring columns of a digit (C x 32 bytes per run, the next run 32 KiB
// further): every lane reads one 32-byte element, C neighbouring lanes form a run, the array (256 MiB) is read
// exactly once per launch.  Run under `rocprofv3 --pmc FETCH_SIZE` and compare the counted bytes with the bytes
// asked for:   ./ubench_fetch_runs        (launch k of runs_kernel has C = 1, 2, 4, 8, 32 for k = 0..4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void runs_kernel(const uint4* __restrict__ table, uint32_t* __restrict__ sink, uint64_t n, int log_c) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint64_t rows = n >> 10;                       // rows of 1024 elements = 32 KiB
  const uint64_t c_lo = g & ((1ull << log_c) - 1), r = (g >> log_c) % rows, c_hi = (g >> log_c) / rows;
  const uint64_t e = (r << 10) + (c_hi << log_c) + c_lo;
  const uint4 a = table[2 * e], b = table[2 * e + 1];
  if ((a.x ^ b.y) == 0x12345678u) sink[0] = a.x;       // (never true: the table holds ones)
}

int main() {
  const uint64_t n = 1ull << 23;                       // 2^23 elements of 32 bytes = 256 MiB
  uint4* table; uint32_t* sink;
  CHECK(hipMalloc(&table, n * 32)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(table, 1, n * 32));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int log_cs[5] = {0, 1, 2, 3, 5};
  for (int k = 0; k < 5; ++k) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(runs_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, table, sink, n, log_cs[k]);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("runs of %4d bytes at a 32 KiB stride: %.3f ms, asked for %.1f MB = %.0f KiB (launch %d of runs_kernel)\n", 32 << log_cs[k], ms, n * 32 / 1e6,
           n * 32 / 1024.0, k);
  }
  return 0;
}
