// Multi-scalar multiplication in BLS12-381 G1 by the bucket method: sum_i s_i * P_i, the body of a
// KZG10 commitment (s = polynomial coefficients, P = powers of tau in G1).
//
// The reference only imports `ark_poly_commit::kzg10` (src/main.rs:9); commitments are listed as not
// implemented (README.md:15-17).  Steps, all on the device except the last:
//   1. every scalar (Montgomery form, as ark-ff keeps it) is made canonical and cut into W digits of c bits;
//      digit d != 0 of window w gives the pair (key = w * 2^c + d, value = scalar index);
//   2. the pairs are sorted by key (the stable radix sort of sigma_table.cuh);
//   3. one thread per bucket adds up its run of points (mixed additions);
//   4. per window, sum_d d * B_d: threads take chunks of MSM_CHUNK buckets (running sums), the chunk
//      results are added up by a tree per window;
//   5. the W window sums go to the host, which combines them (c doublings per window) and converts
//      the result to affine coordinates - a strictly serial tail of ~300 point operations.
#pragma once
#include "field.cuh"
#include "g1.cuh"
#include "sigma_table.cuh"

namespace plk {

constexpr int MSM_CHUNK = 16;           // buckets per thread in step 4
constexpr int MSM_REDUCE_THREADS = 256;

// step 1: keys[w * n + i], vals[w * n + i] for scalar i, window w; also the OR of all keys (for the sort)
__global__ __launch_bounds__(256) void msm_digits_kernel(const uint64_t* scalars, uint64_t n, int c, int windows, uint64_t* keys,
                                                         uint32_t* vals, uint64_t* or_bits) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t seen = 0;
  if (i < n) {
    Fe one = fe_zero(); one.l[0] = 1;
    const Fe s = fe_mul(fe_load(scalars, i), one);          // out of Montgomery form: the integer below p
    for (int w = 0; w < windows; ++w) {
      const int bit = w * c, limb = bit >> 5, off = bit & 31;
      uint64_t chunk = s.l[limb];
      if (limb + 1 < 8) chunk |= (uint64_t)s.l[limb + 1] << 32;
      const uint64_t d = (chunk >> off) & ((1ull << c) - 1);
      const uint64_t key = d ? (((uint64_t)w << c) | d) : SIG_NO_WIRE;
      keys[(uint64_t)w * n + i] = key;
      vals[(uint64_t)w * n + i] = (uint32_t)i;
      if (d) seen |= key;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) seen |= __shfl_xor(seen, d);
  if ((threadIdx.x & 63) == 0 && (seen & ~__atomic_load_n(or_bits, __ATOMIC_RELAXED)))
    atomicOr((unsigned long long*)or_bits, (unsigned long long)seen);
}

// run boundaries of every key in the sorted list: first[key], last[key] (both zero = empty bucket)
__global__ __launch_bounds__(256) void msm_bounds_kernel(SigPairs pairs, const uint64_t* or_bits, uint64_t total, uint32_t* first,
                                                         uint32_t* last) {
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  const uint64_t* keys = pairs.keys[sig_buffer_before(or_bits, 8)];
  const uint64_t key = keys[p];
  if (key == SIG_NO_WIRE) return;
  if (p == 0 || keys[p - 1] != key) first[key] = (uint32_t)p;
  if (p + 1 == total || keys[p + 1] != key) last[key] = (uint32_t)p + 1;
}

// step 3: bucket b = sum of bases[vals[p]] over its run
__global__ __launch_bounds__(256) void msm_bucket_sum_kernel(SigPairs pairs, const uint64_t* or_bits, const uint32_t* first,
                                                             const uint32_t* last, uint64_t nbuckets, const uint64_t* bases,
                                                             uint64_t* buckets) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= nbuckets) return;
  const uint32_t* vals = pairs.vals[sig_buffer_before(or_bits, 8)];
  G1Jac acc = g1_inf();
  for (uint32_t p = first[b]; p < last[b]; ++p) acc = g1_add_affine(acc, g1_load_affine(bases, vals[p]));
  g1_store_jac(buckets, b, acc);
}

// step 4a: thread t handles buckets [lo, lo + MSM_CHUNK) of one window (all inside it: 2^c is a multiple of the
// chunk): out[t] = sum_d d' * B_d with d' the digit value (bucket index within its window)
__global__ __launch_bounds__(256) void msm_chunk_kernel(const uint64_t* buckets, uint64_t nbuckets, int c, uint64_t* out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t lo = t * MSM_CHUNK;
  if (lo >= nbuckets) return;
  G1Jac run = g1_inf(), acc = g1_inf();
  for (int k = MSM_CHUNK - 1; k >= 1; --k) {          // acc = sum_k k * B_{lo+k}
    run = g1_add(run, g1_load_jac(buckets, lo + k));
    acc = g1_add(acc, run);
  }
  run = g1_add(run, g1_load_jac(buckets, lo));         // run = sum of the chunk
  const uint64_t digit0 = lo & ((1ull << c) - 1);      // digit value of the chunk's first bucket
  if (digit0) acc = g1_add(acc, g1_mul_u64(run, digit0));
  g1_store_jac(out, t, acc);
}

// step 4b: out[g] = sum of in[g * per_group .. (g + 1) * per_group), one workgroup per group
__global__ __launch_bounds__(MSM_REDUCE_THREADS) void msm_reduce_kernel(const uint64_t* in, uint64_t per_group, uint64_t* out) {
  __shared__ uint64_t lds[MSM_REDUCE_THREADS * 18];
  const uint64_t base = (uint64_t)blockIdx.x * per_group;
  G1Jac acc = g1_inf();
  for (uint64_t k = threadIdx.x; k < per_group; k += MSM_REDUCE_THREADS) acc = g1_add(acc, g1_load_jac(in, base + k));
  g1_store_jac(lds, threadIdx.x, acc);
  __syncthreads();
  for (int step = MSM_REDUCE_THREADS / 2; step >= 1; step >>= 1) {
    if ((int)threadIdx.x < step) {
      acc = g1_add(g1_load_jac(lds, threadIdx.x), g1_load_jac(lds, threadIdx.x + step));
      g1_store_jac(lds, threadIdx.x, acc);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) g1_store_jac(out, blockIdx.x, acc);
}

// elementwise helpers for setting up and checking bases: out[i] = (first + i) * base (affine, via one inversion each)
__global__ __launch_bounds__(256) void g1_multiples_kernel(G1Affine base, uint64_t first, uint64_t count, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  g1_store_affine(out, i, g1_to_affine(g1_mul_u64(g1_from_affine(base), first + i)));
}

}  // namespace plk
