// Coset transforms and quotient completion - the step after the two numerator vectors.
//
// The reference has no code for this step (README.md:15-17 lists it as next).  What is implemented
// is ark-poly 0.3's coset transform pair (`EvaluationDomain::coset_fft` / `coset_ifft`: coefficient j
// times g^j, then the plain transform; and back) and the textbook division by the vanishing
// polynomial Z_H(x) = x^n - 1 of the trace domain on the coset g * <omega_N>, where Z_H takes only
// N / n distinct values.
#pragma once
#include "field.cuh"

namespace plk {

constexpr int QT_LO_BITS = 10;          // g^j = hi[j >> 10] * lo[j & 1023]
constexpr int QT_MAX_NUMERATORS = 8;
constexpr int QT_MAX_RATIO = 16;        // N / n

// out[i] = base^(i << shift), i < count
__global__ __launch_bounds__(256) void pow_table_kernel(uint64_t* out, uint64_t count, Fe base, int shift) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  fe_store(out, i, fe_pow_u64(base, i << shift));
}

// out[j] = in[j] * g^j, j < count (out may alias in)
__global__ __launch_bounds__(256) void scale_powers_kernel(const uint64_t* in, uint64_t count, const uint64_t* lo, const uint64_t* hi,
                                                           uint64_t* out) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= count) return;
  const Fe g = fe_mul(fe_load(hi, j >> QT_LO_BITS), fe_load(lo, j & ((1u << QT_LO_BITS) - 1)));
  fe_store(out, j, fe_mul(fe_load(in, j), g));
}

struct QuotientArgs {
  const uint64_t* num[QT_MAX_NUMERATORS];
  int count;
  uint32_t ratio;               // N / n, a power of two
  Fe alpha;
  Fe zinv[QT_MAX_RATIO];        // 1 / Z_H at point j depends on j mod ratio only
};

// out[j] = (num_0[j] + alpha (num_1[j] + alpha (...))) / Z_H(point j)
__global__ __launch_bounds__(256) void quotient_combine_kernel(QuotientArgs a, uint64_t N, uint64_t* out) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  Fe acc = fe_load(a.num[a.count - 1], j);
  for (int i = a.count - 2; i >= 0; --i) acc = fe_add(fe_mul(acc, a.alpha), fe_load(a.num[i], j));
  const uint32_t r = (uint32_t)j & (a.ratio - 1);
  Fe zi = a.zinv[0];
#pragma unroll
  for (int k = 1; k < QT_MAX_RATIO; ++k) if (r == (uint32_t)k) zi = a.zinv[k];   // select without indexing kernel arguments
  fe_store(out, j, fe_mul(acc, zi));
}

}  // namespace plk
