// Batched number-theoretic transforms over the two-adic subgroup, natural order in and out.
//
// Replaces ark-poly's `domain.ifft` (src/polynomial_utils.rs:29) and
// `evaluate_over_domain_by_ref` (src/execution_trace.rs:362-370,
// src/permutation_polynomial.rs:323-334): same root-of-unity convention
// (omega_N = TWO_ADIC_ROOT^(2^(s - log2 N))), output index i <-> point omega^i, inverse
// includes the 1/N scale, short coefficient vectors are zero-padded to the domain size.
//
// Decomposition.  N = R1 * R2 * R3 (one to three passes, each radix a power of two <= 2^11).
// Input index n = n1 * (N/R1) + n2 * (N/(R1 R2)) + n3, output index k = k1 + R1 k2 + R1 R2 k3.
// Pass p runs the size-Rp transform over digit p inside one workgroup (tile of Rp points x C
// neighbouring columns held in LDS), multiplies by the inter-pass twiddle
// omega_{N'}^(k_p * lower digits) and writes digit k_p back where n_p was; the last pass reads
// whole contiguous rows and scatters them to natural order in runs of C consecutive outputs.
// Every global access of a tile is therefore a run of C * 32 contiguous bytes.
#pragma once
#include "field.cuh"

namespace plk {

constexpr int NTT_TILE_LOG = 11;            // elements per workgroup tile (2^11 * 32 B = 64 KiB LDS)
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;
constexpr int NTT_THREADS = 512;
constexpr int NTT_MAX_BATCH = 24;           // columns per launch
constexpr int NTT_MASTER_LOG = 11;          // per-tile twiddles: omega_{2^11}^e, e < 2^10

enum NttInputKind : uint32_t { NTT_IN_FIELD = 0, NTT_IN_U64 = 1 };

struct NttBatch {
  const uint64_t* in[NTT_MAX_BATCH];
  uint64_t* out[NTT_MAX_BATCH];
  uint32_t kind[NTT_MAX_BATCH];   // NttInputKind, honoured by the first pass only
};

// Device-resident constants of one (log N, direction) transform.
struct NttTables {
  const uint64_t* master;  // omega_{2^11}^e (or its inverse), e < 2^10
  const uint64_t* tw_hi;   // omega_N^(e << lo_bits), e < 2^(logN - lo_bits)
  const uint64_t* tw_hi_scaled;  // same times 1/N (inverse, first inter-pass twiddle); == tw_hi for forward
  const uint64_t* tw_lo;   // omega_N^e, e < 2^lo_bits
  const uint64_t* scale;   // 1/N as one element, or nullptr
  int lo_bits;
};

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// LDS tile: element e -> low half at lds[e], high half at lds[NTT_TILE + e]
__device__ __forceinline__ Fe lds_get(const uint4* lds, int e) {
  const uint4 lo = lds[e], hi = lds[NTT_TILE + e];
  Fe r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}
__device__ __forceinline__ void lds_put(uint4* lds, int e, const Fe& v) {
  lds[e] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  lds[NTT_TILE + e] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// In-LDS decimation-in-frequency transform of size 2^log_r along the row axis of a
// [2^log_r][2^log_c] tile.  Leaves row i holding output digit bitrev(i).
__device__ __forceinline__ void tile_dif(uint4* lds, int log_r, int log_c, const uint64_t* master) {
  const int total = 1 << (log_r + log_c - 1);  // butterflies per stage
  for (int lh = log_r - 1; lh >= 0; --lh) {
    const int h = 1 << lh;
    for (int b = threadIdx.x; b < total; b += NTT_THREADS) {
      const int c = b & ((1 << log_c) - 1);
      const int t = b >> log_c;
      const int j = t & (h - 1);
      const int i0 = ((t >> lh) << (lh + 1)) + j;
      const int e0 = (i0 << log_c) + c, e1 = ((i0 + h) << log_c) + c;
      const Fe x = lds_get(lds, e0), y = lds_get(lds, e1);
      lds_put(lds, e0, fe_add(x, y));
      Fe d = fe_sub(x, y);
      if (lh > 0 && j != 0) d = fe_mul(d, fe_load(master, (size_t)j << (NTT_MASTER_LOG - 1 - lh)));
      lds_put(lds, e1, d);
    }
    __syncthreads();
  }
}

// omega_N^E from the two-level table
__device__ __forceinline__ Fe twiddle_two_level(const uint64_t* hi, const uint64_t* lo, int lo_bits, uint64_t E) {
  const Fe a = fe_load(hi, E >> lo_bits);
  const uint64_t el = E & ((1ull << lo_bits) - 1);
  if (el == 0) return a;
  return fe_mul(a, fe_load(lo, el));
}

struct NttPassArgs {
  NttBatch batch;
  NttTables tab;
  uint64_t n_in;     // first pass: number of valid input elements (rest read as zero)
  int log_n;         // log2 of the whole transform
  int log_r;         // this pass's radix
  int log_inner;     // log2(product of later radices)
  int log_c;         // columns per tile
  int first;         // 1 = reads the caller's input (kinds, zero padding)
  int use_scaled;    // 1 = take tw_hi_scaled for the inter-pass twiddle
};

// Non-final pass: digit transform + inter-pass twiddle; out may alias in when !first.
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(const NttPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int poly = blockIdx.y;
  const uint64_t* __restrict__ in = a.batch.in[poly];
  uint64_t* __restrict__ out = a.batch.out[poly];
  const int log_c = a.log_c, log_r = a.log_r, log_inner = a.log_inner;
  const int tiles_per_block = 1 << (log_inner - log_c);       // tiles per value of the outer digits
  const uint64_t o = blockIdx.x >> (log_inner - log_c);
  const uint64_t q0 = (uint64_t)(blockIdx.x & (tiles_per_block - 1)) << log_c;
  const uint64_t base = (o << (log_r + log_inner)) + q0;
  const int elems = 1 << (log_r + log_c);
  const uint32_t kind = a.first ? a.batch.kind[poly] : (uint32_t)NTT_IN_FIELD;

  for (int e = threadIdx.x; e < elems; e += NTT_THREADS) {
    const int r = e >> log_c, c = e & ((1 << log_c) - 1);
    const uint64_t idx = base + ((uint64_t)r << log_inner) + c;
    Fe v;
    if (a.first && idx >= a.n_in) v = fe_zero();
    else if (kind == NTT_IN_U64) v = fe_from_u64(in[idx]);
    else v = fe_load(in, idx);
    lds_put(lds, e, v);
  }
  __syncthreads();
  tile_dif(lds, log_r, log_c, a.tab.master);

  // inter-pass twiddle omega_{N'}^(k * q), N' = 2^(log_r + log_inner), as a power of omega_N
  const int shift = a.log_n - (log_r + log_inner);
  const uint64_t* hi = a.use_scaled ? a.tab.tw_hi_scaled : a.tab.tw_hi;
  for (int e = threadIdx.x; e < elems; e += NTT_THREADS) {
    const int r = e >> log_c, c = e & ((1 << log_c) - 1);
    const uint32_t k = bitrev((uint32_t)r, log_r);
    const uint64_t q = q0 + c;
    Fe v = lds_get(lds, e);
    v = fe_mul(v, twiddle_two_level(hi, a.tab.tw_lo, a.tab.lo_bits, ((uint64_t)k * q) << shift));
    fe_store(out, base + ((uint64_t)k << log_inner) + c, v);
  }
}

struct NttFinalArgs {
  NttBatch batch;
  NttTables tab;
  uint64_t n_in;
  int log_n;
  int log_r;      // this (last) pass's radix
  int log_r1;     // radix of pass 1 (0 if this is the only pass)
  int log_r2;     // radix of pass 2 (0 unless there are three passes)
  int log_c;      // rows (consecutive k1) per tile
  int first;
};

// Last pass: contiguous rows in, natural-order scatter out (never in place).
__global__ __launch_bounds__(NTT_THREADS) void ntt_final_kernel(const NttFinalArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int poly = blockIdx.y;
  const uint64_t* __restrict__ in = a.batch.in[poly];
  uint64_t* __restrict__ out = a.batch.out[poly];
  const int log_c = a.log_c, log_r = a.log_r;
  const uint64_t k2 = blockIdx.x & ((1u << a.log_r2) - 1);
  const uint64_t k1_0 = (uint64_t)(blockIdx.x >> a.log_r2) << log_c;
  const int elems = 1 << (log_r + log_c);
  const uint32_t kind = a.first ? a.batch.kind[poly] : (uint32_t)NTT_IN_FIELD;

  // tile[r][c] <- row o_c = (k1_0 + c) * R2 + k2, element r.  Threads run along r so that
  // each row is read as one contiguous stream.
  for (int e = threadIdx.x; e < elems; e += NTT_THREADS) {
    const int r = e & ((1 << log_r) - 1), c = e >> log_r;
    const uint64_t orow = ((k1_0 + c) << a.log_r2) + k2;
    const uint64_t idx = (orow << log_r) + r;
    Fe v;
    if (a.first && idx >= a.n_in) v = fe_zero();
    else if (kind == NTT_IN_U64) v = fe_from_u64(in[idx]);
    else v = fe_load(in, idx);
    lds_put(lds, (r << log_c) + c, v);
  }
  __syncthreads();
  tile_dif(lds, log_r, log_c, a.tab.master);

  const bool scaled = a.tab.scale != nullptr;
  Fe sc = fe_zero();
  if (scaled) sc = fe_load(a.tab.scale, 0);
  for (int e = threadIdx.x; e < elems; e += NTT_THREADS) {
    const int r = e >> log_c, c = e & ((1 << log_c) - 1);
    const uint64_t k = bitrev((uint32_t)r, log_r);
    Fe v = lds_get(lds, e);
    if (scaled) v = fe_mul(v, sc);
    const uint64_t dst = (k1_0 + c) + (k2 << a.log_r1) + (k << (a.log_r1 + a.log_r2));
    fe_store(out, dst, v);
  }
}

// ---- table construction (device side, run once per plan)

// out[i] = base^(i << shift) * mul, i < count ; base, mul given as elements
__global__ void ntt_table_kernel(uint64_t* out, uint64_t count, Fe base, int shift, Fe mul) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fe v = fe_mul(fe_pow_u64(base, i << shift), mul);
  fe_store(out, i, v);
}

}  // namespace plk
