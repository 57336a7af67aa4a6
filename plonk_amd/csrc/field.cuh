// 256-bit prime-field arithmetic for gfx950, Montgomery form with R = 2^256 on 8 x 32-bit limbs.
//
// Memory format of an element is the one ark-ff 0.3 `Fp256` keeps (4 little-endian u64
// Montgomery limbs = 8 little-endian u32 limbs = 32 bytes), so buffers cross the C ABI
// without conversion.  Replaces the ark-ff operators the reference uses at
// src/permutation_polynomial.rs:242-258,354-372 and src/execution_trace.rs:396-426.
//
// The 32 x 32 -> 64 multiply-add (v_mad_u64_u32) issues at half the rate of a plain VALU op
// on MI355X and carries a free 64-bit add, so the multiplier is written to put every
// product through it; the compiler pattern-matches `(u64)a * b + c`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plk {

struct Fe { uint32_t l[8]; };

// BLS12-381 scalar field.  Constants are re-derived in tests/test_constants.py.
struct Bls12_381Fr {
  static constexpr uint32_t P[8]  = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                     0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  // R mod p (the field's one)
  static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                      0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
  // R^2 mod p
  static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                     0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
  static constexpr uint32_t INV32 = 0xffffffffu;  // -p^-1 mod 2^32
  static constexpr int TWO_ADICITY = 32;
  // GENERATOR^((p-1)/2^32), GENERATOR = 7, canonical (non-Montgomery) limbs
  static constexpr uint32_t ROOT[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                                       0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
};

using Field = Bls12_381Fr;

__host__ __device__ __forceinline__ Fe fe_zero() { Fe r; for (int i = 0; i < 8; ++i) r.l[i] = 0; return r; }
__host__ __device__ __forceinline__ Fe fe_one() { Fe r; for (int i = 0; i < 8; ++i) r.l[i] = Field::ONE[i]; return r; }
__host__ __device__ __forceinline__ Fe fe_r2() { Fe r; for (int i = 0; i < 8; ++i) r.l[i] = Field::R2[i]; return r; }

__host__ __device__ __forceinline__ bool fe_is_zero(const Fe& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.l[i];
  return o == 0;
}
__host__ __device__ __forceinline__ bool fe_eq(const Fe& a, const Fe& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i];
  return o == 0;
}

// t - p if t >= p (t < 2p, `top` is the 257th bit)
__host__ __device__ __forceinline__ Fe fe_reduce_once(const uint32_t t[8], uint32_t top) {
  uint32_t d[8], bo = 0, bi;
#pragma unroll
  for (int j = 0; j < 8; ++j) { d[j] = __builtin_subc(t[j], Field::P[j], bo, &bi); bo = bi; }
  const bool ge = (top != 0) | (bo == 0);
  Fe r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.l[j] = ge ? d[j] : t[j];
  return r;
}

__host__ __device__ __forceinline__ Fe fe_add(const Fe& a, const Fe& b) {
  uint32_t t[8], c = 0, ci;
#pragma unroll
  for (int j = 0; j < 8; ++j) { t[j] = __builtin_addc(a.l[j], b.l[j], c, &ci); c = ci; }
  return fe_reduce_once(t, c);  // p < 2^255, so c is always 0; kept for other 256-bit primes
}

__host__ __device__ __forceinline__ Fe fe_sub(const Fe& a, const Fe& b) {
  uint32_t d[8], bo = 0, bi;
#pragma unroll
  for (int j = 0; j < 8; ++j) { d[j] = __builtin_subc(a.l[j], b.l[j], bo, &bi); bo = bi; }
  const uint32_t mask = 0u - bo;  // borrow -> add p back
  Fe r; uint32_t c = 0, ci;
#pragma unroll
  for (int j = 0; j < 8; ++j) { r.l[j] = __builtin_addc(d[j], Field::P[j] & mask, c, &ci); c = ci; }
  return r;
}

__host__ __device__ __forceinline__ Fe fe_neg(const Fe& a) { return fe_sub(fe_zero(), a); }

__host__ __device__ __forceinline__ Fe fe_double(const Fe& a) { return fe_add(a, a); }

// Montgomery product a * b * 2^-256 mod p: coarsely integrated operand scanning, one
// multiplication row and one reduction row per limb of b.  Each row is 8 independent
// 32x32+32 multiply-adds followed by one carry chain that folds the high words in.
__host__ __device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b) {
  uint32_t t[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t pr[8];
    uint32_t c, ci;
#pragma unroll
    for (int j = 0; j < 8; ++j) pr[j] = (uint64_t)a.l[j] * b.l[i] + t[j];
    t[0] = (uint32_t)pr[0];
    c = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) { t[j] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), c, &ci); c = ci; }
    t[8] = __builtin_addc(t[8], (uint32_t)(pr[7] >> 32), c, &ci);
    t[9] = ci;
    const uint32_t m = t[0] * Field::INV32;
#pragma unroll
    for (int j = 0; j < 8; ++j) pr[j] = (uint64_t)m * Field::P[j] + t[j];
    c = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) { t[j - 1] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), c, &ci); c = ci; }
    t[7] = __builtin_addc(t[8], (uint32_t)(pr[7] >> 32), c, &ci);
    t[8] = t[9] + ci;
  }
  return fe_reduce_once(t, t[8]);
}

__host__ __device__ __forceinline__ Fe fe_sqr(const Fe& a) { return fe_mul(a, a); }

// F::from(u64): the integer x as a field element (Montgomery residue x * R mod p)
__host__ __device__ __forceinline__ Fe fe_from_u64(uint64_t x) {
  Fe v = fe_zero();
  v.l[0] = (uint32_t)x; v.l[1] = (uint32_t)(x >> 32);
  return fe_mul(v, fe_r2());
}

// a^e for a 64-bit exponent (host-side table construction and device-side small powers)
__host__ __device__ inline Fe fe_pow_u64(Fe base, uint64_t e) {
  Fe acc = fe_one();
  while (e) { if (e & 1) acc = fe_mul(acc, base); base = fe_sqr(base); e >>= 1; }
  return acc;
}

// a^(p-2): Fermat inversion (a != 0), 255 squarings.  Host-side table construction only.
__host__ __device__ inline Fe fe_inverse(const Fe& a) {
  // exponent p - 2, scanned from the top bit
  uint32_t e[8], borrow = 2;
  for (int i = 0; i < 8; ++i) { const uint32_t v = Field::P[i]; e[i] = v - borrow; borrow = v < borrow ? 1u : 0u; }
  Fe acc = fe_one();
  for (int i = 255; i >= 0; --i) {
    acc = fe_sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul(acc, a);
  }
  return acc;
}

// Inverse by the binary extended Euclidean algorithm on the Montgomery residue (a = xR -> x^-1 R).
// About 7x fewer instructions than the Fermat chain when a single lane has to do it, which is
// the case for the one inversion of the grand product.  Returns zero for zero.
__host__ __device__ inline Fe fe_inverse_eea(const Fe& a) {
  if (fe_is_zero(a)) return a;
  auto is_one = [](const uint32_t* x) {
    uint32_t o = x[0] ^ 1u;
    for (int i = 1; i < 8; ++i) o |= x[i];
    return o == 0;
  };
  auto halve = [](uint32_t* x) {
    for (int i = 0; i < 7; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
    x[7] >>= 1;
  };
  auto add_p = [](uint32_t* x) {   // x < p, so x + p < 2^256
    uint32_t c = 0, ci;
    for (int i = 0; i < 8; ++i) { x[i] = __builtin_addc(x[i], Field::P[i], c, &ci); c = ci; }
  };
  auto geq = [](const uint32_t* x, const uint32_t* y) {
    for (int i = 7; i >= 0; --i) if (x[i] != y[i]) return x[i] > y[i];
    return true;
  };
  auto sub_raw = [](uint32_t* x, const uint32_t* y) {
    uint32_t b = 0, bi;
    for (int i = 0; i < 8; ++i) { x[i] = __builtin_subc(x[i], y[i], b, &bi); b = bi; }
  };
  uint32_t u[8], v[8];
  Fe b = fe_r2(), c = fe_zero();
  for (int i = 0; i < 8; ++i) { u[i] = a.l[i]; v[i] = Field::P[i]; }
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) { halve(u); if (b.l[0] & 1) add_p(b.l); halve(b.l); }
    while (!(v[0] & 1)) { halve(v); if (c.l[0] & 1) add_p(c.l); halve(c.l); }
    if (geq(u, v)) { sub_raw(u, v); b = fe_sub(b, c); }
    else           { sub_raw(v, u); c = fe_sub(c, b); }
  }
  return is_one(u) ? b : c;
}

// ---- global / LDS access helpers: an element is two 16-byte halves
__device__ __forceinline__ Fe fe_load(const uint64_t* base, size_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * idx;
  const uint4 lo = p[0], hi = p[1];
  Fe r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}
__device__ __forceinline__ void fe_store(uint64_t* base, size_t idx, const Fe& v) {
  uint4* p = reinterpret_cast<uint4*>(base) + 2 * idx;
  p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

}  // namespace plk
