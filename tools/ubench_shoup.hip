// Product by a CONSTANT with a precomputed quotient (Shoup / Harvey style) on 9 x 29-bit limbs, against the
// Montgomery product mont29_x2 of fr29.cuh: isolated rate and a correctness check (tools only).
//
//   w  = the constant, canonical, as 9 limbs;  wq = floor(w * 2^261 / p)
//   q  = floor(a * wq / 2^261), from the product columns 7..16 only (the dropped columns change q by < 2^-20)
//   r  = a * w - q * p = low 261 bits of a * w + q * (2^261 - p),   0 <= r < 3 p for a < 2^261
// 143 multiply-adds (53 + 45 + 45) and ~36 shifts / masks, against 153 and ~80.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../plonk_amd/csrc/fr29.cuh"
using namespace plk;

#define PIN(x) asm volatile("" : "+v"(x))

namespace sh {
// limb i of 2^261 - p
constexpr uint32_t pbar_limb(int i) {
  // 2^261 - p = (2^261 - 1) - p + 1: limb-wise complement of p's limbs, plus one at the bottom
  // p's limb 0 is 1, so limb 0 of the result is 2^29 - 1 and no carry travels further
  return i == 0 ? M29 : (i < 8 ? (M29 - f9c::kp_limb29(1, i)) : (M29 - f9c::kp_limb29(1, 8)));
}
}  // namespace sh

__host__ __device__ __forceinline__ void shoup29(const F9& a, const F9& w, const F9& wq, F9& r) {
  uint64_t acc = 0;
  uint32_t q[9];
#pragma unroll
  for (int i = 0; i <= 7; ++i) acc += (uint64_t)a.l[i] * wq.l[7 - i];
  acc >>= 29;
#pragma unroll
  for (int i = 0; i <= 8; ++i) acc += (uint64_t)a.l[i] * wq.l[8 - i];
  acc >>= 29;
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a.l[i] * wq.l[k - i];
    q[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  q[8] = (uint32_t)acc;
  acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * w.l[k - i];
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)q[i] * sh::pbar_limb(k - i);
    r.l[k] = (uint32_t)acc & M29;
    acc >>= 29;
  }
}

__device__ __forceinline__ void shoup29_x2(const F9& a1, const F9& w1, const F9& wq1, const F9& a2, const F9& w2, const F9& wq2,
                                           F9& r1, F9& r2) {
  uint64_t acc1 = 0, acc2 = 0;
  uint32_t q1[9], q2[9];
#pragma unroll
  for (int i = 0; i <= 7; ++i) { acc1 += (uint64_t)a1.l[i] * wq1.l[7 - i]; acc2 += (uint64_t)a2.l[i] * wq2.l[7 - i]; PIN(acc1); PIN(acc2); }
  acc1 >>= 29; acc2 >>= 29;
#pragma unroll
  for (int i = 0; i <= 8; ++i) { acc1 += (uint64_t)a1.l[i] * wq1.l[8 - i]; acc2 += (uint64_t)a2.l[i] * wq2.l[8 - i]; PIN(acc1); PIN(acc2); }
  acc1 >>= 29; acc2 >>= 29;
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc1 += (uint64_t)a1.l[i] * wq1.l[k - i]; acc2 += (uint64_t)a2.l[i] * wq2.l[k - i]; PIN(acc1); PIN(acc2); }
    q1[k - 9] = (uint32_t)acc1 & M29; q2[k - 9] = (uint32_t)acc2 & M29;
    acc1 >>= 29; acc2 >>= 29;
  }
  q1[8] = (uint32_t)acc1; q2[8] = (uint32_t)acc2;
  acc1 = 0; acc2 = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc1 += (uint64_t)a1.l[i] * w1.l[k - i]; acc2 += (uint64_t)a2.l[i] * w2.l[k - i]; PIN(acc1); PIN(acc2); }
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc1 += (uint64_t)q1[i] * sh::pbar_limb(k - i); acc2 += (uint64_t)q2[i] * sh::pbar_limb(k - i); PIN(acc1); PIN(acc2); }
    r1.l[k] = (uint32_t)acc1 & M29; r2.l[k] = (uint32_t)acc2 & M29;
    acc1 >>= 29; acc2 >>= 29;
  }
}

// ---- host big integers (32-bit words, little endian) for the constants and the check
typedef std::vector<uint32_t> Big;
static int cmp(const Big& a, const Big& b) {
  for (int i = (int)a.size() - 1; i >= 0; --i) { uint32_t bi = i < (int)b.size() ? b[i] : 0; if (a[i] != bi) return a[i] < bi ? -1 : 1; }
  return 0;
}
static void sub_in(Big& a, const Big& b) {
  uint64_t bo = 0;
  for (size_t i = 0; i < a.size(); ++i) { uint64_t bi = i < b.size() ? b[i] : 0; uint64_t d = (uint64_t)a[i] - bi - bo; a[i] = (uint32_t)d; bo = (d >> 63) & 1; }
}
// floor(num / p), num % p by bitwise long division; num has `bits` bits
static void divmod(const Big& num, int bits, const Big& p, Big& quo, Big& rem) {
  quo.assign(num.size(), 0); rem.assign(p.size() + 1, 0);
  for (int b = bits - 1; b >= 0; --b) {
    uint32_t carry = (num[b >> 5] >> (b & 31)) & 1;
    for (size_t i = 0; i < rem.size(); ++i) { uint32_t nc = rem[i] >> 31; rem[i] = (rem[i] << 1) | carry; carry = nc; }
    if (cmp(rem, p) >= 0) { sub_in(rem, p); quo[b >> 5] |= 1u << (b & 31); }
  }
}
static F9 f9_from_big(const Big& v) {
  F9 r;
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, w = bit >> 5, o = bit & 31;
    uint64_t x = (w < (int)v.size() ? v[w] : 0);
    if (w + 1 < (int)v.size()) x |= (uint64_t)v[w + 1] << 32;
    r.l[i] = (uint32_t)(x >> o) & M29;
  }
  return r;
}

template <int V>
__global__ __launch_bounds__(512) void mul_kernel(const uint32_t* tw, int reps, uint32_t* sink) {
  // records: [0] = w * 2^261 mod p (Montgomery), [1] = w, [2] = wq
  const int t = threadIdx.x & 255;
  const F9 wm = f9_load_tw(tw, 3 * t), w = f9_load_tw(tw, 3 * t + 1), wq = f9_load_tw(tw, 3 * t + 2);
  F9 x;
  for (int i = 0; i < 9; ++i) x.l[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & M29;
  x.l[8] &= 0xffffff;
  F9 x2 = x; x2.l[0] ^= 5;
  for (int r = 0; r < reps; r += 2) {
    F9 o1, o2;
    if (V == 0) mont29_x2(x, wm, x2, wm, o1, o2);
    if (V == 1) shoup29_x2(x, w, wq, x2, w, wq, o1, o2);
    x = o1; x2 = o2;
  }
  uint32_t acc = 0;
  for (int i = 0; i < 9; ++i) acc ^= x.l[i] ^ x2.l[i];
  sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

// one product per thread by both methods, canonical results side by side
__global__ void check_kernel(const uint32_t* tw, const uint32_t* ain, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = i & 255;
  const F9 wm = f9_load_tw(tw, 3 * t), w = f9_load_tw(tw, 3 * t + 1), wq = f9_load_tw(tw, 3 * t + 2);
  F9 a;
  for (int k = 0; k < 9; ++k) a.l[k] = ain[9 * i + k];
  const F9 m = mont29(a, wm);
  F9 s, s2, d;
  shoup29(a, w, wq, s);
  shoup29_x2(a, w, wq, a, w, wq, s2, d);
  const Fe rm = fe_reduce_f9(m), rs = fe_reduce_f9(s), rs2 = fe_reduce_f9(s2);
  for (int k = 0; k < 8; ++k) { out[24 * i + k] = rm.l[k]; out[24 * i + 8 + k] = rs.l[k]; out[24 * i + 16 + k] = rs2.l[k]; }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int V> int run(const char* name, const uint32_t* tw, uint32_t* sink, int wps) {
  const int blocks = 256 * wps / 2, reps = 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mul_kernel<V>, dim3(blocks), dim3(512), 0, 0, tw, reps, sink);
  CHECK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(mul_kernel<V>, dim3(blocks), dim3(512), 0, 0, tw, reps, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double muls = (double)blocks * 512 * reps;
  printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f G products/s\n", name, wps, best, muls / (best * 1e-3) * 1e-9);
  return 0;
}

int main() {
  Big p(8);
  for (int i = 0; i < 8; ++i) p[i] = Field::P[i];
  // 256 constants w < p, their Montgomery records and quotients
  std::vector<uint32_t> rec(256 * 3 * 12, 0);
  srand(7);
  for (int t = 0; t < 256; ++t) {
    Big w(8);
    for (int i = 0; i < 8; ++i) w[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    w[7] &= 0x3fffffff;   // below 2^254 < p
    if (t == 0) { w.assign(8, 0); w[0] = 1; }
    if (t == 1) { w = p; w[0] -= 1; }   // p - 1
    Big num(17, 0);   // w << 261
    for (int i = 0; i < 8; ++i) { const int bit = 32 * i + 261; num[bit >> 5] |= w[i] << (bit & 31); if (bit & 31) num[(bit >> 5) + 1] |= w[i] >> (32 - (bit & 31)); }
    Big quo, rem;
    divmod(num, 17 * 32, p, quo, rem);
    const F9 wm = f9_from_big(rem), wl = f9_from_big(w), wq = f9_from_big(quo);
    for (int i = 0; i < 9; ++i) { rec[(3 * t) * 12 + i] = wm.l[i]; rec[(3 * t + 1) * 12 + i] = wl.l[i]; rec[(3 * t + 2) * 12 + i] = wq.l[i]; }
  }
  uint32_t *tw, *sink;
  CHECK(hipMalloc(&tw, rec.size() * 4));
  CHECK(hipMemcpy(tw, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&sink, 256 * 8 * 512 * 4));

  // correctness: lazy inputs at the bounds mont29 documents (limbs < 1.3 * 2^31, value < 64 p)
  const int n = 1 << 16;
  std::vector<uint32_t> a(9 * n);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 9; ++k) {
      uint32_t v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
      if (i % 3 == 0) v &= M29;                                   // normalised
      else if (i % 3 == 1) v = (uint32_t)((uint64_t)v * 5 / 8);   // < 1.25 * 2^31
      else v = (k & 1) ? 0xA6666665u : v % 0xA6666666u;           // at the limb bound
      a[9 * i + k] = v;
    }
    // value < 64 p: the top limb (weight 2^232) below 64 p / 2^232 ~ 2^28.86 minus what the lazy lower limbs add
    a[9 * i + 8] %= (i % 3 == 0) ? 0x1cfb69d0u : 0x1c000000u;
  }
  uint32_t *ad, *od;
  CHECK(hipMalloc(&ad, a.size() * 4)); CHECK(hipMalloc(&od, (size_t)n * 24 * 4));
  CHECK(hipMemcpy(ad, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, tw, ad, od, n);
  CHECK(hipDeviceSynchronize());
  std::vector<uint32_t> o((size_t)n * 24);
  CHECK(hipMemcpy(o.data(), od, o.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i)
    if (memcmp(&o[24 * i], &o[24 * i + 8], 32) || memcmp(&o[24 * i], &o[24 * i + 16], 32)) { if (bad < 5) printf("mismatch at %d\n", i); ++bad; }
  printf("check: %d products, %d mismatches between the Montgomery and the precomputed-quotient product\n", n, bad);
  if (bad) return 2;

  for (int wps : {2, 4, 8}) {
    if (run<0>("Montgomery, two chains in lockstep", tw, sink, wps)) return 1;
    if (run<1>("precomputed quotient, two chains", tw, sink, wps)) return 1;
  }
  return 0;
}
