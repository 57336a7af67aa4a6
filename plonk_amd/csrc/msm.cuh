// Multi-scalar multiplication in BLS12-381 G1 by the bucket method over a precomputed table:
// sum_i s_i * P_i, the body of a KZG10 commitment (s = polynomial coefficients, P = powers of tau in G1).
//
// The reference only imports `ark_poly_commit::kzg10` (src/main.rs:9); commitments are listed as not
// implemented (README.md:15-17).  The bases are fixed per circuit size, so when they are set the table
// T[w][i] = 2^(offset of window w) * P_i (affine, W = ceil(255 / c) windows of c or c - 1 bits) is built once; with it digit d of window w of
// scalar i contributes d * T[w][i], and ALL windows share one set of 2^c buckets:
//   1. every scalar (Montgomery form, as ark-ff keeps it) is made canonical and cut into W digits of c bits;
//      digit d of window w gives the 64-bit key (w * nb + i) << 16 | d;
//   2. the keys are sorted by digit (the stable radix sort of sigma_table.cuh, two byte passes);
//   3. a fixed number of adjacent lanes per bucket add up its run of table entries (mixed additions) and
//      combine their partial sums through lane shuffles;
//   4. sum_d d * B_d by folding: with d = hi * 2^h + lo, sum_d d B_d = sum_k k (2^h R_k + C_k), R_k / C_k the
//      row / column sums of B seen as a matrix - plain sums (trees), and the same problem on 2^ceil(c/2)
//      points; after about log2(c) folds two points are left and the second is the result;
//   5. the host converts that one point to affine coordinates (one base-field inversion).
#pragma once
#include "field.cuh"
#include "g1.cuh"
#include "fq28.cuh"
#include "sigma_table.cuh"

namespace plk {

constexpr int MSM_FOLD_THREADS = 512;
constexpr int MSM_MAX_WINDOW_BITS = 16;   // a fold handles a matrix of at most 256 x 256 buckets

// The 255 scalar bits are cut into `windows` digits of nearly equal width: the first `wide` of them have c bits, the
// others c - 1 (a narrow last window - 255 mod c bits - would put its digits into a few buckets with very long runs).
struct MsmWindows {
  int c, windows, wide;
  __host__ __device__ __forceinline__ int width(int w) const { return w < wide ? c : c - 1; }
  __host__ __device__ __forceinline__ int offset(int w) const { return w * (c - 1) + (w < wide ? w : wide); }
};
__host__ __device__ inline MsmWindows msm_windows_for(int max_bits) {
  MsmWindows r;
  r.windows = (255 + max_bits - 1) / max_bits;
  const int base = 255 / r.windows, rem = 255 % r.windows;
  r.c = base + (rem ? 1 : 0);
  r.wide = rem ? rem : r.windows;
  return r;
}

constexpr int MSM_KEY_DIGIT_BITS = 16;    // a pair is one 64-bit key: (table entry << 16) | digit; the sort looks at the digit only
static_assert(MSM_MAX_WINDOW_BITS <= MSM_KEY_DIGIT_BITS, "a digit must fit the low bits of a key");

// step 1: key[w * n + i] = (table entry w * nb + i) << 16 | digit w of scalar i; also the OR of all digits (the sort
// skips the byte passes it shows to be zero, which leaves the upper bits in their stable order)
__global__ __launch_bounds__(256) void msm_digits_kernel(const uint64_t* scalars, uint64_t n, uint64_t nb, MsmWindows win,
                                                         uint64_t* keys, uint64_t* or_bits) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t seen = 0;
  if (i < n) {
    Fe one = fe_zero(); one.l[0] = 1;
    const Fe s = fe_mul(fe_load(scalars, i), one);          // out of Montgomery form: the integer below p
    for (int w = 0; w < win.windows; ++w) {
      const int bit = win.offset(w), limb = bit >> 5, off = bit & 31;
      uint64_t chunk = s.l[limb];
      if (limb + 1 < 8) chunk |= (uint64_t)s.l[limb + 1] << 32;
      const uint64_t d = (chunk >> off) & ((1ull << win.width(w)) - 1);
      keys[(uint64_t)w * n + i] = (((uint64_t)w * nb + i) << MSM_KEY_DIGIT_BITS) | d;    // digit 0 lands in bucket 0, which is never read
      seen |= d;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) seen |= __shfl_xor(seen, d);
  if ((threadIdx.x & 63) == 0 && (seen & ~__atomic_load_n(or_bits, __ATOMIC_RELAXED)))
    atomicOr((unsigned long long*)or_bits, (unsigned long long)seen);
}

// run boundaries of every digit d >= 1 in the sorted list: first[d], last[d] (both zero = empty bucket)
__global__ __launch_bounds__(256) void msm_bounds_kernel(SigPairs pairs, const uint64_t* or_bits, uint64_t total, uint32_t* first,
                                                         uint32_t* last) {
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  constexpr uint64_t mask = (1ull << MSM_KEY_DIGIT_BITS) - 1;
  const uint64_t* keys = pairs.keys[sig_buffer_before(or_bits, 8)];
  const uint64_t d = keys[p] & mask;
  if (d == 0) return;
  if (p == 0 || (keys[p - 1] & mask) != d) first[d] = (uint32_t)p;
  if (p + 1 == total || (keys[p + 1] & mask) != d) last[d] = (uint32_t)p + 1;
}

// step 3: bucket b = sum of the table entries of its run.  A run is shared by ceil(length / per_lane) lanes (entry j,
// j + lanes, ... each), whatever its length: digits that crowd into a few buckets (equal scalars, small scalars)
// get as many lanes as they have entries for, and an empty bucket gets none.  Three kernels: the lanes of every
// bucket and their prefix sums; the additions (a kernel of its own because the mixed addition alone fits three waves
// per SIMD, with the full addition beside it two); and the sum of each bucket's partial sums - two lanes per bucket,
// or a whole workgroup for a bucket with more than MSM_BIG_BUCKET of them.
constexpr uint32_t MSM_BIG_BUCKET = 64;
constexpr int MSM_BIG_THREADS = 256;

__device__ __forceinline__ F14 f14_shfl_down(const F14& a, int delta) {
  F14 r;
#pragma unroll
  for (int k = 0; k < 14; ++k) r.l[k] = __shfl_down(a.l[k], delta);
  return r;
}

// off[b] = number of lanes of the buckets before b; off[nbuckets] = all lanes.  One workgroup.
__global__ __launch_bounds__(SIG_SCAN_THREADS) void msm_lanes_kernel(const uint32_t* first, const uint32_t* last, uint32_t nbuckets,
                                                                     uint32_t per_lane, uint32_t* off) {
  __shared__ uint32_t lds[SIG_SCAN_THREADS / 64];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nbuckets; base += SIG_SCAN_THREADS) {   // uniform trip count
    const uint32_t b = base + threadIdx.x;
    const uint32_t v = b < nbuckets ? (last[b] - first[b] + per_lane - 1) / per_lane : 0;
    uint32_t total;
    const uint32_t ex = sig_block_exclusive(v, lds, &total);
    if (b < nbuckets) off[b] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) off[nbuckets] = carry;
}

// one lane: partials[t] = sum of its share of its bucket's run (the grid is an upper bound of the lane count)
__global__ __launch_bounds__(256, 3) void msm_bucket_sum_kernel(SigPairs pairs, const uint64_t* or_bits, const uint32_t* first,
                                                                const uint32_t* last, uint32_t nbuckets, const uint32_t* off,
                                                                const uint32_t* table, uint32_t* partials) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= off[nbuckets]) return;
  const uint32_t t = (uint32_t)gid;
  uint32_t lo = 0, hi = nbuckets;                     // off[lo] <= t < off[hi]: the bucket is the b with off[b] <= t < off[b + 1]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= t) lo = mid; else hi = mid;
  }
  const uint32_t b = lo, lanes = off[b + 1] - off[b], j = t - off[b];
  const uint64_t* keys = pairs.keys[sig_buffer_before(or_bits, 8)];
  J28 acc = j28_inf();
  const uint32_t end = last[b];
  for (uint32_t p = first[b] + j; p < end; p += lanes) acc = j28_add_affine<true>(acc, a28_load(table, keys[p] >> MSM_KEY_DIGIT_BITS));
  j28_store(partials, t, acc);
}

// bucket b = sum of its partial sums, for the buckets with at most MSM_BIG_BUCKET of them: two lanes per bucket add
// up half of them each, one after the other, and combine by a shuffle
__global__ __launch_bounds__(256) void msm_bucket_combine_kernel(const uint32_t* partials, const uint32_t* off, uint32_t nbuckets, uint32_t* buckets) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // the grid covers whole waves: the shuffle needs every lane
  const uint32_t b = (uint32_t)(gid >> 1), h = (uint32_t)gid & 1;
  const bool in_range = b < nbuckets;
  const uint32_t at = in_range ? off[b] : 0, lanes = in_range ? off[b + 1] - at : 0;
  const bool mine_to_do = in_range && lanes <= MSM_BIG_BUCKET;
  J28 part = j28_inf();
  if (mine_to_do) {
    const uint32_t half = (lanes + 1) / 2, from = h ? half : 0, count = h ? lanes - half : half;
    for (uint32_t k = 0; k < count; ++k) part = k ? j28_add_inl(part, j28_load(partials, at + from + k)) : j28_load(partials, at + from);
  }
  J28 other;
  other.x = f14_shfl_down(part.x, 1); other.y = f14_shfl_down(part.y, 1); other.z = f14_shfl_down(part.z, 1);
  if (h) other.z = f14_zero();
  part = j28_add_inl(part, other);
  if (mine_to_do && h == 0) j28_store(buckets, b, part);
}

// the same for the buckets with more partial sums than that, a workgroup per bucket: every workgroup looks through its
// share of the buckets 256 at a time and adds up the few it finds (strided sums per thread, then an LDS tree)
__global__ __launch_bounds__(MSM_BIG_THREADS) void msm_bucket_combine_big_kernel(const uint32_t* partials, const uint32_t* off, uint32_t nbuckets,
                                                                                uint32_t* buckets) {
  __shared__ uint32_t lds[MSM_BIG_THREADS * J28_WORDS];
  __shared__ uint32_t big[MSM_BIG_THREADS];
  __shared__ uint32_t nbig;
  for (uint32_t base = blockIdx.x * MSM_BIG_THREADS; base < nbuckets; base += gridDim.x * MSM_BIG_THREADS) {   // uniform per workgroup
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
    const uint32_t mine = base + threadIdx.x;
    if (mine < nbuckets && off[mine + 1] - off[mine] > MSM_BIG_BUCKET) big[atomicAdd(&nbig, 1u)] = mine;
    __syncthreads();
    const uint32_t found = nbig;
    for (uint32_t q = 0; q < found; ++q) {
      const uint32_t b = big[q], at = off[b], lanes = off[b + 1] - at;
      J28 acc = j28_inf();
      for (uint32_t k = threadIdx.x; k < lanes; k += MSM_BIG_THREADS) acc = j28_add_inl(acc, j28_load(partials, at + k));
      j28_store(lds, threadIdx.x, acc);
      __syncthreads();
      for (uint32_t step = MSM_BIG_THREADS / 2; step >= 1; step >>= 1) {
        if (threadIdx.x < step) {
          acc = j28_add_inl(acc, j28_load(lds, threadIdx.x + step));
          j28_store(lds, threadIdx.x, acc);
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) j28_store(buckets, b, acc);
    }
    __syncthreads();
  }
}

// step 4, one fold: in = 2^c points B_d, d = hi * cols + lo with cols = 2^floor(c/2), rows = 2^ceil(c/2);
// out[k] = cols * (sum_lo B[k][lo]) + (k < cols ? sum_hi B[hi][k] : 0), k < rows, so that
// sum_d d * B_d = sum_k k * out[k].  One workgroup per k: its first half sums row k, its second half column k.
// With abi_out set (the last fold) the points are also written there in the 12 x 32-bit form of g1.cuh.
__global__ __launch_bounds__(MSM_FOLD_THREADS) void msm_fold_kernel(const uint32_t* in, int c, uint32_t* out, uint64_t* abi_out) {
  __shared__ uint32_t lds[MSM_FOLD_THREADS * J28_WORDS];
  const int lo_bits = c >> 1, hi_bits = c - lo_bits;
  const uint32_t rows = 1u << hi_bits, cols = 1u << lo_bits, k = blockIdx.x;
  const uint32_t t = threadIdx.x, u = t & 255;
  J28 acc = j28_inf();
  if (t < 256) { if (u < cols) acc = j28_load(in, (uint64_t)k * cols + u); }
  else if (k < cols && u < rows) acc = j28_load(in, (uint64_t)u * cols + k);
  j28_store(lds, t, acc);
  __syncthreads();
  for (uint32_t step = rows >> 1; step >= 1; step >>= 1) {      // rows >= cols: nothing lies beyond lane `rows` of a half
    if (u < step) {
      acc = j28_add_inl(acc, j28_load(lds, t + step));
      j28_store(lds, t, acc);
    }
    __syncthreads();
  }
  if (t == 0) {
    for (int d = 0; d < lo_bits; ++d) acc = j28_double_inl(acc);
    acc = j28_add_inl(acc, j28_load(lds, 256));
    j28_store(out, k, acc);
    if (abi_out) g1_store_jac(abi_out, k, j28_to_abi(acc));
  }
}

// Table set-up, one thread per base: table[w * nb + i] = 2^(offset of window w) * P_i as affine points in the working form of
// fq28.cuh (32 words each), w < windows.  Computed in the ABI form: the Jacobian X, Y are parked in the table slot
// itself and Z in `zs`; the running products of the Z's in `prods` turn the W conversions to affine into one
// inversion per base.  (An infinite base gives W infinite entries; G1 has odd order, so doubling a finite point
// never reaches infinity.)
__global__ __launch_bounds__(256) void msm_table_kernel(const uint64_t* bases, uint64_t nb, MsmWindows win, uint32_t* table,
                                                        uint64_t* zs, uint64_t* prods) {
  const int windows = win.windows;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nb) return;
  const G1Affine a = g1_load_affine(bases, i);
  if (g1_affine_is_inf(a)) {
    A28 inf; inf.x = f14_zero(); inf.y = f14_zero();
    for (int w = 0; w < windows; ++w) a28_store(table, (uint64_t)w * nb + i, inf);
    return;
  }
  G1Jac p = g1_from_affine(a);
  Fq run = fq_one();
  for (int w = 0; w < windows; ++w) {
    if (w) for (int d = 0; d < win.width(w - 1); ++d) p = g1_double(p);
    const uint64_t slot = (uint64_t)w * nb + i;
    uint64_t* park = reinterpret_cast<uint64_t*>(table + slot * A28_WORDS);     // 128 bytes: room for two 48-byte coordinates
    fq_store(park, 0, p.x);
    fq_store(park, 1, p.y);
    fq_store(zs, slot, p.z);
    run = fq_mul(run, p.z);
    fq_store(prods, slot, run);
  }
  Fq inv = fq_inverse(run);                       // 1 / (Z_0 ... Z_{W-1})
  for (int w = windows - 1; w >= 0; --w) {
    const uint64_t slot = (uint64_t)w * nb + i;
    const Fq zi = w ? fq_mul(inv, fq_load(prods, slot - nb)) : inv;       // 1 / Z_w
    inv = fq_mul(inv, fq_load(zs, slot));
    const Fq zi2 = fq_sqr(zi);
    const uint64_t* park = reinterpret_cast<const uint64_t*>(table + slot * A28_WORDS);
    G1Affine q;
    q.x = fq_mul(fq_load(park, 0), zi2);
    q.y = fq_mul(fq_load(park, 1), fq_mul(zi2, zi));
    a28_store(table, slot, a28_from_abi(q));
  }
}

// elementwise helpers for setting up and checking bases: out[i] = (first + i) * base (affine, via one inversion each)
__global__ __launch_bounds__(256) void g1_multiples_kernel(G1Affine base, uint64_t first, uint64_t count, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  g1_store_affine(out, i, g1_to_affine(g1_mul_u64(g1_from_affine(base), first + i)));
}

}  // namespace plk
