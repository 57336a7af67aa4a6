// Permutation grand product Z as a device-wide parallel prefix product.
//
// Reference: build_grand_product_evaluation_polynomial, src/permutation_polynomial.rs:217-261.
//   z[0] = 1, z[j+1] = z[j] * num_j / den_j,
//   num_j = prod_k (v_k[j] + beta * v_k[sigma_k[j]] + gamma)   (:235-245, value at the permuted ROW)
//   den_j = prod_k (v_k[j] + beta * j + gamma)                 (:249-254, bare row index)
// The reference inverts den_j once per row (:258).  Here z[j+1] = P_j * S_{j+1} / D with
// P_j = prod_{i<=j} num_i, S_{j+1} = prod_{i>j} den_i, D = prod_i den_i: one prefix scan,
// one suffix scan and ONE field inversion for the whole column (Montgomery's batch-inversion
// identity applied to the running denominators).  The value is the same field element.
//
// A zero den_j makes the reference panic (`unwrap`, :258) and sigma_k[j] >= n is a slice
// panic (:235-238); both are recorded in `GpStatus` (smallest offending row).
#pragma once
#include "field.cuh"
#include "fr29.cuh"

namespace plk {

constexpr int GP_THREADS = 256;
constexpr int GP_ITEMS = 4;
constexpr int GP_BLOCK_ROWS = GP_THREADS * GP_ITEMS;
constexpr int GP_SCAN_THREADS = 1024;
constexpr int GP_SMALL_ROWS = 1 << 15;        // up to here: one row per thread (gp_local_z_small_kernel)

struct GpStatus {
  unsigned long long zero_den_row;   // ~0ull = none
  unsigned long long bad_sigma_row;  // ~0ull = none
};

struct GpCols {
  const uint64_t* v[4];      // wire value columns, n elements each
  const uint64_t* sigma[4];  // permutation index columns, n u64 each
};

__device__ __forceinline__ Fe fe_shfl(const Fe& v, int src_lane) {
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = __shfl(v.l[i], src_lane);
  return r;
}

// Inclusive scan (operator = field multiplication) of one value per thread across the
// workgroup, in thread order (REVERSE = false) or reversed thread order (REVERSE = true).
// `lds` holds one element per wave.  Returns the inclusive result; *total = product of all.
template <int THREADS, bool REVERSE>
__device__ __forceinline__ Fe block_scan_mul(Fe v, Fe* lds, Fe* total) {
  constexpr int WAVES = THREADS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int src = REVERSE ? lane + d : lane - d;
    const Fe o = fe_shfl(v, src & 63);
    if (REVERSE ? (src < 64) : (src >= 0)) v = fe_mul(v, o);
  }
  if (lane == (REVERSE ? 0 : 63)) lds[wave] = v;
  __syncthreads();
  // second level: the first wave scans the WAVES wave totals the same way, in place
  if (wave == 0) {
    Fe t = lane < WAVES ? lds[lane] : fe_one();
#pragma unroll
    for (int d = 1; d < WAVES; d <<= 1) {
      const int src = REVERSE ? lane + d : lane - d;
      const Fe o = fe_shfl(t, src & 63);
      if (REVERSE ? (src < WAVES) : (src >= 0)) t = fe_mul(t, o);
    }
    if (lane < WAVES) lds[lane] = t;
  }
  __syncthreads();
  *total = lds[REVERSE ? 0 : WAVES - 1];
  const int prev = REVERSE ? wave + 1 : wave - 1;
  const bool has_carry = REVERSE ? (prev < WAVES) : (prev >= 0);
  const Fe carry = lds[has_carry ? prev : 0];
  __syncthreads();
  return has_carry ? fe_mul(v, carry) : v;
}

// ---- whole-circuit form: one pass over the columns ------------------------------------------------------------
//
// z[j+1] = prod_{i<=j} num_i / den_i.  With the rows cut into blocks of GP_BLOCK_ROWS and r_b = (product of the
// block's numerators) / (product of its denominators),
//   z[j+1] = C_b * L_j,   C_b = prod_{b' < b} r_b',   L_j = prod_{i <= j, i in block b} num_i / den_i.
// gp_local_z_kernel reads the columns ONCE, keeps the per-row terms in registers, inverts ONE value per block
// (Montgomery's batch-inversion identity inside the block: 1 / prod_{i<=j} den_i = (1 / D_b) * prod_{i>j} den_i)
// and writes L_j where z[j+1] goes, plus r_b; gp_ratio_scan_kernel (one workgroup) turns the r_b into the C_b;
// gp_apply_kernel multiplies each block of z by its C_b.  Per row: 288 bytes in, 32 out, then 32 in and out again
// for the carry - against 288 + 64 out, 64 + 32 back in of the three-kernel form below (kept for row ranges on
// several GPUs, which need the products apart) - and no single-workgroup inversion on the critical path.

// The first kernel's arithmetic runs on 29-bit limbs with the transforms' product mont29 (a b 2^-261, fr29.cuh):
// ~195 vector instructions per product against ~390 for fe_mul, and its result is below (V_a V_b / 70.7 + 1) p, so
// chains of products stay small with no reduction in between.  For that the values live in the Montgomery form of
// radix R' = 2^261 = 32 R, in which mont29 is closed.  Getting there costs no product:
//   beta * v_sigma      = mont29(v_sigma R, beta 2^266)          (constant prepared on the host)
//   beta * F::from(j)   = mont29(j, beta 2^522)
//   v R'                = 32 (v R): a five-bit limb shift (f9_shl5), value below 32 p, well inside the lazy range
// and getting back costs none either: with inv = (1 / D) R in the library's own form,
//   z R = mont29(prefix R', mont29(suffix R', inv)),   r_b R = mont29(N_b R', inv),   D R = mont29(D R', R).
struct GpConsts {
  F9 beta_sigma;   // beta 2^266 mod p
  F9 beta_index;   // beta 2^522 mod p
  F9 gamma;        // gamma R' mod p
  F9 one;          // R' mod p
};

__device__ __forceinline__ F9 f9_shfl(const F9& v, int src_lane) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = __shfl(v.l[i], src_lane);
  return r;
}

// 32 a for a normalised a below 2^256: limbs below 2^29 + 2^5, top limb below 2^29
__device__ __forceinline__ F9 f9_shl5(const F9& a) {
  F9 r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r.l[i] = ((a.l[i] & ((1u << 24) - 1)) << 5) + c; c = a.l[i] >> 24; }
  r.l[8] = (a.l[8] << 5) + c;
  return r;
}

__device__ __forceinline__ F9 f9_index(uint64_t v) {
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = 0;
  r.l[0] = (uint32_t)v & M29; r.l[1] = (uint32_t)(v >> 29) & M29; r.l[2] = (uint32_t)(v >> 58);
  return r;
}

// block_scan_mul on 29-bit limbs (values in R' form, normalised, below 1.5 p after the first product)
template <int THREADS, bool REVERSE>
__device__ __forceinline__ F9 block_scan_mul9(F9 v, F9* lds, F9* total, const F9& one) {
  constexpr int WAVES = THREADS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int src = REVERSE ? lane + d : lane - d;
    const F9 o = f9_shfl(v, src & 63);
    if (REVERSE ? (src < 64) : (src >= 0)) v = mont29(v, o);
  }
  if (lane == (REVERSE ? 0 : 63)) lds[wave] = v;
  __syncthreads();
  if (wave == 0) {
    F9 t = lane < WAVES ? lds[lane] : one;
#pragma unroll
    for (int d = 1; d < WAVES; d <<= 1) {
      const int src = REVERSE ? lane + d : lane - d;
      const F9 o = f9_shfl(t, src & 63);
      if (REVERSE ? (src < WAVES) : (src >= 0)) t = mont29(t, o);
    }
    if (lane < WAVES) lds[lane] = t;
  }
  __syncthreads();
  *total = lds[REVERSE ? 0 : WAVES - 1];
  const int prev = REVERSE ? wave + 1 : wave - 1;
  const bool has_carry = REVERSE ? (prev < WAVES) : (prev >= 0);
  const F9 carry = lds[has_carry ? prev : 0];
  __syncthreads();
  return has_carry ? mont29(v, carry) : v;
}

// num_j and den_j of one row in R' form, normalised and below 5.5 p (R' itself for a row beyond the circuit);
// failures recorded with their row.  Bounds in units of p beside the lines.
__device__ __forceinline__ void gp_row_terms(const GpCols& cols, uint64_t n, uint64_t j, const GpConsts& k9, GpStatus* status, F9& nj, F9& dj) {
  nj = k9.one; dj = k9.one;
  if (j >= n) return;
  const F9 bj_g = f9_add(mont29(f9_index(j), k9.beta_index), k9.gamma);          // beta * F::from(j) + gamma    < 2.1
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const F9 v5 = f9_shl5(f9_from_fe(fe_load(cols.v[k], j)));                    // 32 v                          < 32
    uint64_t sg = cols.sigma[k][j];
    if (sg >= n) { atomicMin(&status->bad_sigma_row, (unsigned long long)j); sg = 0; }
    const F9 vs = f9_from_fe(fe_load(cols.v[k], sg));
    F9 tn = f9_add(f9_add(v5, mont29(vs, k9.beta_sigma)), k9.gamma);             // v + beta v_sigma + gamma      < 34.1
    F9 td = f9_add(v5, bj_g);                                                    // v + beta j + gamma            < 34.1
    if (k == 0) { f9_normalize(tn); f9_normalize(td); nj = tn; dj = td; }
    else { nj = mont29(tn, nj); dj = mont29(td, dj); }                           // < 17.5, 9.5, 5.6
  }
  if (fe_is_zero(fe_reduce_f9(dj))) atomicMin(&status->zero_den_row, (unsigned long long)j);
}

// (1 / D) R as limbs, from D R' (zero stays zero; the status word carries the error)
__device__ __forceinline__ F9 gp_invert(const F9& d_rprime) {
  const Fe d = fe_reduce_f9(mont29(d_rprime, f9_from_fe(fe_one())));   // D R, canonical
  return f9_from_fe(fe_inverse_gcd32(d));
}

// Kernel A: local quotient products L_j -> z_after[j] (= z[j+1] up to the block carry), block ratios r_b -> ratio[b].
#ifndef PLK_GP_WAVES
#define PLK_GP_WAVES 4
#endif
__global__ __launch_bounds__(GP_THREADS, PLK_GP_WAVES) void gp_local_z_kernel(GpCols cols, uint64_t n, GpConsts k9,
                                                                   uint64_t* z_after, uint64_t* z0, uint64_t* ratio, GpStatus* status) {
  __shared__ F9 lds[GP_THREADS / 64];
  __shared__ F9 edge[GP_THREADS / 64];
  __shared__ F9 inv_total;
  const uint64_t row0 = ((uint64_t)blockIdx.x * GP_THREADS + threadIdx.x) * GP_ITEMS;
  // the four rows are written out with named values (no array, no loop): the terms must stay in registers from
  // here to the stores at the end
  static_assert(GP_ITEMS == 4, "gp_local_z_kernel spells out four rows per thread");
  F9 n0, n1, n2, n3, d0, d1, d2, d3;
  gp_row_terms(cols, n, row0 + 0, k9, status, n0, d0);
  gp_row_terms(cols, n, row0 + 1, k9, status, n1, d1);
  gp_row_terms(cols, n, row0 + 2, k9, status, n2, d2);
  gp_row_terms(cols, n, row0 + 3, k9, status, n3, d3);
  // thread-local inclusive prefix of the numerators, inclusive suffix of the denominators (every product below 1.5 p)
  n1 = mont29(n0, n1); n2 = mont29(n1, n2); n3 = mont29(n2, n3);
  d2 = mont29(d3, d2); d1 = mont29(d2, d1); d0 = mont29(d1, d0);
  F9 tot_n, tot_d;
  const F9 inc_n = block_scan_mul9<GP_THREADS, false>(n3, lds, &tot_n, k9.one);
  const F9 inc_d = block_scan_mul9<GP_THREADS, true>(d0, lds, &tot_d, k9.one);
  if (threadIdx.x == 0) {
    const F9 inv = gp_invert(tot_d);
    inv_total = inv;
    fe_store(ratio, blockIdx.x, fe_reduce_f9(mont29(tot_n, inv)));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  F9 ex_n = f9_shfl(inc_n, (lane - 1) & 63);
  F9 ex_d = f9_shfl(inc_d, (lane + 1) & 63);
  if (lane == 63) lds[wave] = inc_n;
  if (lane == 0) edge[wave] = inc_d;
  __syncthreads();
  if (lane == 0) ex_n = wave == 0 ? k9.one : lds[wave - 1];
  if (lane == 63) ex_d = wave == GP_THREADS / 64 - 1 ? k9.one : edge[wave + 1];
  const F9 cd = mont29(ex_d, inv_total);          // (suffix of the later threads) / D, in the library's R form
  // L_j = (prefix of numerators up to j) * (suffix of denominators after j) / (all denominators of the block)
  if (row0 + 0 < n) fe_store(z_after, row0 + 0, fe_reduce_f9(mont29(mont29(ex_n, n0), mont29(d1, cd))));
  if (row0 + 1 < n) fe_store(z_after, row0 + 1, fe_reduce_f9(mont29(mont29(ex_n, n1), mont29(d2, cd))));
  if (row0 + 2 < n) fe_store(z_after, row0 + 2, fe_reduce_f9(mont29(mont29(ex_n, n2), mont29(d3, cd))));
  if (row0 + 3 < n) fe_store(z_after, row0 + 3, fe_reduce_f9(mont29(mont29(ex_n, n3), cd)));
  if (z0 && blockIdx.x == 0 && threadIdx.x == 0) fe_store(z0, 0, fe_one());
}

// Kernel A for small circuits: ONE row per thread (blocks of GP_THREADS rows).  A block's latency is what a small
// circuit waits for - its rows' terms, the two scans, the inversion, the finish, all dependent - and with one row
// per thread the terms and the finish are a quarter as long; the price, one inversion per 256 rows instead of per
// 1024, is paid in parallel while the GPU has lanes to spare.
__global__ __launch_bounds__(GP_THREADS) void gp_local_z_small_kernel(GpCols cols, uint64_t n, GpConsts k9,
                                                                      uint64_t* z_after, uint64_t* z0, uint64_t* ratio, GpStatus* status) {
  __shared__ F9 lds[GP_THREADS / 64];
  __shared__ F9 edge[GP_THREADS / 64];
  __shared__ F9 inv_total;
  const uint64_t j = (uint64_t)blockIdx.x * GP_THREADS + threadIdx.x;
  F9 nj, dj;
  gp_row_terms(cols, n, j, k9, status, nj, dj);
  F9 tot_n, tot_d;
  const F9 inc_d = block_scan_mul9<GP_THREADS, true>(dj, lds, &tot_d, k9.one);
  if (threadIdx.x == GP_THREADS - 1) inv_total = gp_invert(tot_d);
  const F9 inc_n = block_scan_mul9<GP_THREADS, false>(nj, lds, &tot_n, k9.one);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  F9 ex_d = f9_shfl(inc_d, (lane + 1) & 63);
  if (lane == 0) edge[wave] = inc_d;
  __syncthreads();
  if (lane == 63) ex_d = wave == GP_THREADS / 64 - 1 ? k9.one : edge[wave + 1];
  if (threadIdx.x == 0) fe_store(ratio, blockIdx.x, fe_reduce_f9(mont29(tot_n, inv_total)));
  // prefix of numerators * (suffix of denominators / all denominators)
  if (j < n) fe_store(z_after, j, fe_reduce_f9(mont29(inc_n, mont29(ex_d, inv_total))));
  if (z0 && blockIdx.x == 0 && threadIdx.x == 0) fe_store(z0, 0, fe_one());
}

// Kernel B (one workgroup): exclusive prefix products of the nb block ratios, in place.
__global__ __launch_bounds__(GP_SCAN_THREADS) void gp_ratio_scan_kernel(uint64_t* ratio, uint32_t nb) {
  __shared__ Fe lds[GP_SCAN_THREADS / 64];
  __shared__ Fe edge[GP_SCAN_THREADS / 64];
  const uint32_t per = (nb + GP_SCAN_THREADS - 1) / GP_SCAN_THREADS;
  const uint32_t lo = min(threadIdx.x * per, nb), hi = min(lo + per, nb);
  Fe prod = fe_one();
  for (uint32_t b = lo; b < hi; ++b) prod = fe_mul(prod, fe_load(ratio, b));
  Fe tot;
  const Fe inc = block_scan_mul<GP_SCAN_THREADS, false>(prod, lds, &tot);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Fe run = fe_shfl(inc, (lane - 1) & 63);
  if (lane == 63) edge[wave] = inc;
  __syncthreads();
  if (lane == 0) run = wave == 0 ? fe_one() : edge[wave - 1];
  for (uint32_t b = lo; b < hi; ++b) { const Fe t = fe_load(ratio, b); fe_store(ratio, b, run); run = fe_mul(run, t); }
}

// Kernel C: z_after[j] *= carry[block of j]; block 0's carry is one and its rows are left alone.
template <int ITEMS>
__global__ __launch_bounds__(GP_THREADS) void gp_apply_kernel(uint64_t* z_after, uint64_t n, const uint64_t* carry) {
  const uint32_t b = blockIdx.x + 1;
  const Fe c = fe_load(carry, b);
  const uint64_t base = (uint64_t)b * GP_THREADS * ITEMS;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const uint64_t j = base + (uint64_t)it * GP_THREADS + threadIdx.x;     // consecutive threads, consecutive rows
    if (j < n) fe_store(z_after, j, fe_mul(fe_load(z_after, j), c));
  }
}

// Kernels B and C in one for small circuits (blocks of GP_THREADS rows, at most GP_THREADS of them): every workgroup
// multiplies the ratios of the blocks before its own together by itself - a few dependent products - instead of
// waiting for a one-workgroup scan kernel in between.  `ratio` is only read.
__global__ __launch_bounds__(GP_THREADS) void gp_apply_small_kernel(uint64_t* z_after, uint64_t n, const uint64_t* ratio) {
  __shared__ Fe lds[GP_THREADS / 64];
  const uint32_t b = blockIdx.x + 1;
  Fe c;
  block_scan_mul<GP_THREADS, false>(threadIdx.x < b ? fe_load(ratio, threadIdx.x) : fe_one(), lds, &c);
  const uint64_t j = (uint64_t)b * GP_THREADS + threadIdx.x;
  if (j < n) fe_store(z_after, j, fe_mul(fe_load(z_after, j), c));
}

// The same for blocks of GP_THREADS * ITEMS rows, at most that many of them (2^20 rows at four items): a thread
// multiplies ITEMS of the earlier blocks' ratios, the workgroup the 256 partial products; about a dozen dependent
// products per workgroup, all workgroups at once, where the one-workgroup scan kernel took longer than the apply itself.
template <int ITEMS>
__global__ __launch_bounds__(GP_THREADS) void gp_apply_fused_kernel(uint64_t* z_after, uint64_t n, const uint64_t* ratio) {
  __shared__ Fe lds[GP_THREADS / 64];
  const uint32_t b = blockIdx.x + 1;
  Fe prod = fe_one();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const uint32_t i = (uint32_t)k * GP_THREADS + threadIdx.x;            // consecutive threads, consecutive ratios
    if (i < b) prod = (k == 0) ? fe_load(ratio, i) : fe_mul(prod, fe_load(ratio, i));
  }
  Fe c;
  block_scan_mul<GP_THREADS, false>(prod, lds, &c);
  const uint64_t base = (uint64_t)b * GP_THREADS * ITEMS;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const uint64_t j = base + (uint64_t)it * GP_THREADS + threadIdx.x;
    if (j < n) fe_store(z_after, j, fe_mul(fe_load(z_after, j), c));
  }
}

// ---- row-range form (rows split over several GPUs): terms, block products, carries, finish ---------------------

// Kernel 1: per-row numerator / denominator terms and per-block products, for the rows
// [first, first + rows) of an n-row circuit (the whole circuit: first = 0, rows = n).  num/den/block_*
// are indexed from the range's first row; the gathers reach any row of the columns.
__global__ __launch_bounds__(GP_THREADS) void gp_terms_kernel(GpCols cols, uint64_t n, uint64_t first, uint64_t rows,
                                                              Fe beta, Fe gamma, Fe beta_r2,
                                                              uint64_t* num, uint64_t* den,
                                                              uint64_t* block_num, uint64_t* block_den,
                                                              GpStatus* status) {
  __shared__ Fe lds[GP_THREADS / 64];
  const uint64_t row0 = ((uint64_t)blockIdx.x * GP_THREADS + threadIdx.x) * GP_ITEMS;
  Fe pn = fe_one(), pd = fe_one();
#pragma unroll 1
  for (int it = 0; it < GP_ITEMS; ++it) {
    const uint64_t local = row0 + it;
    if (local >= rows) break;
    const uint64_t j = first + local;
    // beta * F::from(j): j as a raw integer times beta * R^2 / R
    Fe jraw = fe_zero(); jraw.l[0] = (uint32_t)j; jraw.l[1] = (uint32_t)(j >> 32);
    const Fe bj_g = fe_add(fe_mul(jraw, beta_r2), gamma);
    Fe nj = fe_one(), dj = fe_one();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Fe v = fe_load(cols.v[k], j);
      uint64_t s = cols.sigma[k][j];
      if (s >= n) { atomicMin(&status->bad_sigma_row, (unsigned long long)j); s = 0; }
      const Fe vs = fe_load(cols.v[k], s);
      const Fe tn = fe_add(fe_add(v, fe_mul(beta, vs)), gamma);
      const Fe td = fe_add(v, bj_g);
      nj = (k == 0) ? tn : fe_mul(nj, tn);
      dj = (k == 0) ? td : fe_mul(dj, td);
    }
    if (fe_is_zero(dj)) atomicMin(&status->zero_den_row, (unsigned long long)j);
    fe_store(num, local, nj);
    fe_store(den, local, dj);
    pn = fe_mul(pn, nj);
    pd = fe_mul(pd, dj);
  }
  Fe tot_n, tot_d;
  block_scan_mul<GP_THREADS, false>(pn, lds, &tot_n);
  block_scan_mul<GP_THREADS, false>(pd, lds, &tot_d);
  if (threadIdx.x == 0) { fe_store(block_num, blockIdx.x, tot_n); fe_store(block_den, blockIdx.x, tot_d); }
}

// Kernel 2 (one workgroup): exclusive prefix of the block numerators, exclusive suffix of the
// block denominators, in place.  totals == nullptr: the range is the whole circuit and 1/D is
// folded into the prefix here.  Otherwise the range's two products go to totals[0], totals[1] and
// the inversion waits for gp_carry_kernel, which sees every range's totals.
__global__ __launch_bounds__(GP_SCAN_THREADS) void gp_block_scan_kernel(uint64_t* block_num, uint64_t* block_den, uint32_t nb,
                                                                        uint64_t* totals) {
  __shared__ Fe lds[GP_SCAN_THREADS / 64];
  __shared__ Fe inv_total;
  const uint32_t per = (nb + GP_SCAN_THREADS - 1) / GP_SCAN_THREADS;
  const uint32_t lo = threadIdx.x * per;
  const uint32_t hi = min(lo + per, nb);
  Fe pn = fe_one(), pd = fe_one();
  for (uint32_t b = lo; b < hi; ++b) { pn = fe_mul(pn, fe_load(block_num, b)); pd = fe_mul(pd, fe_load(block_den, b)); }
  Fe tot_n, tot_d;
  const Fe inc_n = block_scan_mul<GP_SCAN_THREADS, false>(pn, lds, &tot_n);
  const Fe inc_d = block_scan_mul<GP_SCAN_THREADS, true>(pd, lds, &tot_d);
  if (threadIdx.x == 0) {
    if (totals) { fe_store(totals, 0, tot_n); fe_store(totals, 1, tot_d); inv_total = fe_one(); }
    else inv_total = fe_inverse_gcd32(tot_d);               // zero stays zero; the status word carries the error
  }
  __syncthreads();
  // exclusive carries of this thread's chunk: inclusive result with the own chunk divided out is
  // not available (no division), so rebuild them from the neighbours' inclusive values
  Fe ex_n = fe_shfl(inc_n, (threadIdx.x - 1) & 63);
  Fe ex_d = fe_shfl(inc_d, (threadIdx.x + 1) & 63);
  // lane 0 / lane 63 need the neighbouring wave's value: pass through LDS
  __shared__ Fe edge_n[GP_SCAN_THREADS / 64], edge_d[GP_SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) edge_n[wave] = inc_n;
  if (lane == 0) edge_d[wave] = inc_d;
  __syncthreads();
  if (lane == 0) ex_n = wave == 0 ? fe_one() : edge_n[wave - 1];
  if (lane == 63) ex_d = wave == GP_SCAN_THREADS / 64 - 1 ? fe_one() : edge_d[wave + 1];
  Fe run = fe_mul(ex_n, inv_total);
  for (uint32_t b = lo; b < hi; ++b) { const Fe t = fe_load(block_num, b); fe_store(block_num, b, run); run = fe_mul(run, t); }
  run = ex_d;
  for (uint32_t b = hi; b > lo; --b) { const Fe t = fe_load(block_den, b - 1); fe_store(block_den, b - 1, run); run = fe_mul(run, t); }
}

// Kernel 3: in-block prefix / suffix scans combined with the block carries -> z[1..n]; z[0] = 1.
// z_after[local] = z[first + local + 1] for the range's rows; z0 (may be null) receives z[0] = 1.
__global__ __launch_bounds__(GP_THREADS) void gp_finish_kernel(const uint64_t* num, const uint64_t* den, uint64_t n,
                                                               const uint64_t* block_num_ex, const uint64_t* block_den_ex,
                                                               uint64_t* z_after, uint64_t* z0) {
  __shared__ Fe lds[GP_THREADS / 64];
  __shared__ Fe edge[GP_THREADS / 64];
  const uint64_t row0 = ((uint64_t)blockIdx.x * GP_THREADS + threadIdx.x) * GP_ITEMS;
  // named values, not arrays: the terms stay in registers (see gp_local_z_kernel)
  static_assert(GP_ITEMS == 4, "gp_finish_kernel spells out four rows per thread");
  const Fe one = fe_one();
  Fe n0 = row0 + 0 < n ? fe_load(num, row0 + 0) : one, d0 = row0 + 0 < n ? fe_load(den, row0 + 0) : one;
  Fe n1 = row0 + 1 < n ? fe_load(num, row0 + 1) : one, d1 = row0 + 1 < n ? fe_load(den, row0 + 1) : one;
  Fe n2 = row0 + 2 < n ? fe_load(num, row0 + 2) : one, d2 = row0 + 2 < n ? fe_load(den, row0 + 2) : one;
  Fe n3 = row0 + 3 < n ? fe_load(num, row0 + 3) : one, d3 = row0 + 3 < n ? fe_load(den, row0 + 3) : one;
  // thread-local inclusive prefix of num, inclusive suffix of den
  n1 = fe_mul(n0, n1); n2 = fe_mul(n1, n2); n3 = fe_mul(n2, n3);
  d2 = fe_mul(d3, d2); d1 = fe_mul(d2, d1); d0 = fe_mul(d1, d0);
  Fe tot;
  const Fe inc_n = block_scan_mul<GP_THREADS, false>(n3, lds, &tot);
  const Fe inc_d = block_scan_mul<GP_THREADS, true>(d0, lds, &tot);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // exclusive carries: previous thread's inclusive prefix, next thread's inclusive suffix
  Fe ex_n = fe_shfl(inc_n, (lane - 1) & 63);
  Fe ex_d = fe_shfl(inc_d, (lane + 1) & 63);
  if (lane == 63) lds[wave] = inc_n;
  if (lane == 0) edge[wave] = inc_d;
  __syncthreads();
  if (lane == 0) ex_n = wave == 0 ? fe_one() : lds[wave - 1];
  if (lane == 63) ex_d = wave == GP_THREADS / 64 - 1 ? fe_one() : edge[wave + 1];
  const Fe cn = fe_mul(ex_n, fe_load(block_num_ex, blockIdx.x));   // includes 1/D
  const Fe cd = fe_mul(ex_d, fe_load(block_den_ex, blockIdx.x));
  // z[j+1] = (cn * prefix_num_j) * (suffix_den_{j+1} * cd)
  if (row0 + 0 < n) fe_store(z_after, row0 + 0, fe_mul(fe_mul(cn, n0), fe_mul(d1, cd)));
  if (row0 + 1 < n) fe_store(z_after, row0 + 1, fe_mul(fe_mul(cn, n1), fe_mul(d2, cd)));
  if (row0 + 2 < n) fe_store(z_after, row0 + 2, fe_mul(fe_mul(cn, n2), fe_mul(d3, cd)));
  if (row0 + 3 < n) fe_store(z_after, row0 + 3, fe_mul(fe_mul(cn, n3), cd));
  if (z0 && blockIdx.x == 0 && threadIdx.x == 0) fe_store(z0, 0, fe_one());
}

// Row ranges on several GPUs: totals_all[2 g], totals_all[2 g + 1] = numerator and denominator
// product of range g (ranges in row order).  Range `mine` needs, on top of its own scans,
//   prefix carry = prod_{g < mine} numerator_g / D,   suffix carry = prod_{g > mine} denominator_g,
// D = prod_g denominator_g: the one inversion, done by every range's owner on the same inputs.
__global__ __launch_bounds__(GP_THREADS) void gp_carry_kernel(const uint64_t* totals_all, uint32_t ranges, uint32_t mine,
                                                              uint64_t* block_num_ex, uint64_t* block_den_ex, uint32_t nb) {
  __shared__ Fe carry[2];
  if (threadIdx.x == 0) {
    Fe cn = fe_one(), cd = fe_one(), all = fe_one();
    for (uint32_t g = 0; g < ranges; ++g) {
      const Fe d = fe_load(totals_all, 2 * g + 1);
      all = fe_mul(all, d);
      if (g < mine) cn = fe_mul(cn, fe_load(totals_all, 2 * g));
      if (g > mine) cd = fe_mul(cd, d);
    }
    carry[0] = fe_mul(cn, fe_inverse_gcd32(all));
    carry[1] = cd;
  }
  __syncthreads();
  const Fe cn = carry[0], cd = carry[1];
  for (uint32_t b = threadIdx.x; b < nb; b += GP_THREADS) {
    fe_store(block_num_ex, b, fe_mul(fe_load(block_num_ex, b), cn));
    fe_store(block_den_ex, b, fe_mul(fe_load(block_den_ex, b), cd));
  }
}

}  // namespace plk
