// Instruction-rate microbenchmark for gfx950 (MI355X).
//
// Measures the sustained issue rate of the VALU instructions a 256-bit
// Montgomery multiplier could be built from, so the multiplier design is chosen
// from measured rates.  Each kernel runs 8 independent dependency chains per
// lane, 256 lanes per workgroup, 8 waves per SIMD on every CU.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
// run  : ./tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define KERNEL_BEGIN(name) \
  __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) { \
    uint32_t a = threadIdx.x * 2654435761u + seed, b = (blockIdx.x + 1) * 40503u + seed;

#define KERNEL_END32(acc) \
    uint32_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= acc[c]; \
    if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = r; }

KERNEL_BEGIN(k_add_u32)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
  }
KERNEL_END32(acc)

KERNEL_BEGIN(k_mul_lo_u32)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
  }
KERNEL_END32(acc)

KERNEL_BEGIN(k_mul_hi_u32)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
  }
KERNEL_END32(acc)

KERNEL_BEGIN(k_mad_u32_u24)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(b));
  }
KERNEL_END32(acc)

KERNEL_BEGIN(k_mul_hi_u32_u24)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
  }
KERNEL_END32(acc)

KERNEL_BEGIN(k_mad_u32_u16)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(b));
  }
KERNEL_END32(acc)

KERNEL_BEGIN(k_fma_f32)
  float acc[CHAINS]; float fb = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
  for (int c = 0; c < CHAINS; ++c) acc[c] = (float)(a & 1023) + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(fb));
  }
  uint32_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= __float_as_uint(acc[c]);
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = r; }

KERNEL_BEGIN(k_mad_u64_u32)
  uint64_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= acc[c];
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r; }

// same, but the multiplicand changes every instruction through the chain (lo word feeds back)
KERNEL_BEGIN(k_mad_u64_u32_dep)
  uint64_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "s20", "s21");
  }
  uint64_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= acc[c];
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r; }

KERNEL_BEGIN(k_lshl_add_u64)
  uint64_t acc[CHAINS]; uint64_t bb = ((uint64_t)b << 32) | a;
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(bb));
  }
  uint64_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= acc[c];
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r; }

KERNEL_BEGIN(k_addc_chain)
  // add_co + addc_co pairs: 64-bit adds through VCC
  uint32_t lo[CHAINS], hi[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { lo[c] = a + c; hi[c] = b + c; }
  for (int i = 0; i < ITERS / 2; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                   : "+v"(lo[c]), "+v"(hi[c]) : "v"(a), "v"(b) : "vcc");
  }
  uint32_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= lo[c] ^ hi[c];
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = r; }

KERNEL_BEGIN(k_fma_f64)
  double acc[CHAINS]; double fb = 1.0 + (double)(b & 1023) * 1e-9;
  for (int c = 0; c < CHAINS; ++c) acc[c] = (double)(a & 1023) + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(fb));
  }
  uint64_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= (uint64_t)__double_as_longlong(acc[c]);
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r; }

KERNEL_BEGIN(k_add_f64)
  double acc[CHAINS]; double fb = 1.0 + (double)(b & 1023) * 1e-9;
  for (int c = 0; c < CHAINS; ++c) acc[c] = (double)(a & 1023) + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(fb));
  }
  uint64_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= (uint64_t)__double_as_longlong(acc[c]);
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r; }

KERNEL_BEGIN(k_pk_fma_f32)
  // packed f32 fma: two f32 per lane per instruction
  uint64_t acc[CHAINS]; uint64_t fb = 0x3f8000013f800001ull;
  for (int c = 0; c < CHAINS; ++c) acc[c] = 0x3f8000003f800000ull + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(acc[c]) : "v"(fb));
  }
  uint64_t r = 0; for (int c = 0; c < CHAINS; ++c) r ^= acc[c];
  if (r == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r; }

KERNEL_BEGIN(k_dot4_u32_u8)
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
  }
KERNEL_END32(acc)

typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double clk_ghz = prop.clockRate * 1e-6;
  printf("device: %s, CUs=%d, clock=%.3f GHz\n", prop.name, cus, clk_ghz);
  uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  struct K { const char* name; kern_t fn; double instr_per_iter; };
  std::vector<K> ks = {
    {"v_add_u32", k_add_u32, CHAINS}, {"v_mul_lo_u32", k_mul_lo_u32, CHAINS},
    {"v_mul_hi_u32", k_mul_hi_u32, CHAINS}, {"v_mad_u32_u24", k_mad_u32_u24, CHAINS},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, CHAINS}, {"v_mad_u32_u16", k_mad_u32_u16, CHAINS},
    {"v_fma_f32", k_fma_f32, CHAINS}, {"v_pk_fma_f32", k_pk_fma_f32, CHAINS},
    {"v_mad_u64_u32(vcc)", k_mad_u64_u32, CHAINS},
    {"v_mad_u64_u32(sgpr)", k_mad_u64_u32_dep, CHAINS}, {"v_lshl_add_u64", k_lshl_add_u64, CHAINS},
    {"add_co+addc_co", k_addc_chain, CHAINS}, {"v_fma_f64", k_fma_f64, CHAINS},
    {"v_add_f64", k_add_f64, CHAINS}, {"v_dot4_u32_u8", k_dot4_u32_u8, CHAINS},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {8, 4, 2, 1}) {   // waves per SIMD
    int blocks = cus * wps;          // 256 threads = 4 waves = 1 wave per SIMD per block
    printf("-- %d wave(s) per SIMD (%d blocks of 256)\n", wps, blocks);
    for (auto& k : ks) {
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      const int reps = 5;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 2u + r);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      double wave_instr = (double)blocks * 4 * ITERS * k.instr_per_iter;     // wave-instructions
      double per_simd_per_s = wave_instr / (cus * 4.0) / (ms * 1e-3);
      double cyc_per_instr = clk_ghz * 1e9 / per_simd_per_s;
      printf("%-22s %8.3f ms  %7.2f cycles/wave-instr/SIMD @%.2fGHz  (%.2f T lane-ops/s)\n", k.name, ms,
             cyc_per_instr, clk_ghz, wave_instr * 64 / (ms * 1e-3) * 1e-12);
    }
  }
  CHECK(hipFree(out));
  return 0;
}
