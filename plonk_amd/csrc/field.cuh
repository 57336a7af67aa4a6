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

// The fields the library is instantiated for.  Every constant below is printed by tools/derive_field_constants.py
// from the modulus alone (and re-derived by the tests); the transform arithmetic of fr29.cuh needs p < 2^255 and
// p = 1 mod 2^29.  QEST = (2^232 / p) (1 - 2^-20) rounded down to a float (fr29.cuh, fe_reduce_f9).

// BLS12-381 scalar field (field 0, the default).  Generator 7 and the root below are the constants arkworks
// publishes for this field.
struct Bls12_381Fr {
  static constexpr int ID = 0;
  static constexpr const char* NAME = "BLS12-381 Fr";
  static constexpr bool HAS_G1 = true;     // the commitments (g1.cuh, msm.cuh) are sums in this field's curve
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
  static constexpr float QEST = 0x1.1aa838p-23f;
};

// Scalar field of Tweedledee, 2^254 + 4707489544292117082687961190295928833 (field 1): the reference names
// `ark_tweedle::Fr` (src/main.rs:3), a crate its manifest does not list.  Generator 5 (the smallest quadratic
// non-residue) and the root derived from it are SELF-DERIVED: no arkworks source is available to check them
// against, so parity with ark-ff's constants for this field is unverifiable.
struct TweedleFr {
  static constexpr int ID = 1;
  static constexpr const char* NAME = "Tweedle Fr";
  static constexpr bool HAS_G1 = false;
  static constexpr uint32_t P[8]  = {0x00000001u, 0x842cafd4u, 0x696286c9u, 0x038aa127u,
                                     0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
  static constexpr uint32_t ONE[8] = {0xfffffffdu, 0x7379f083u, 0xc3d86ba3u, 0xf5601c89u,
                                      0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
  static constexpr uint32_t R2[8] = {0x00000010u, 0x8595fa80u, 0xc6895230u, 0x7e16a565u,
                                     0xb03aa0a2u, 0xf4c0e6fcu, 0x06886013u, 0x00c8ad91u};
  static constexpr uint32_t INV32 = 0xffffffffu;
  static constexpr int TWO_ADICITY = 34;
  // 5^((p-1)/2^34)
  static constexpr uint32_t ROOT[8] = {0x869d57ecu, 0x1cbd3234u, 0xb8084fbfu, 0x0a287dd1u,
                                       0x5a987293u, 0xf1dbcb64u, 0x0dc03c0bu, 0x113efc51u};
  static constexpr float QEST = 0x1.ffffdep-23f;
};

// One translation unit = one field: PLK_FIELD_ID picks it (capi.hip is compiled once per field).
#ifndef PLK_FIELD_ID
#define PLK_FIELD_ID 0
#endif
#if PLK_FIELD_ID == 0
using Field = Bls12_381Fr;
#elif PLK_FIELD_ID == 1
using Field = TweedleFr;
#else
#error "unknown PLK_FIELD_ID"
#endif
static_assert(Field::ID == PLK_FIELD_ID, "field table out of order");

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

// ---- inversion by the binary GCD with batched steps (Pornin, "Optimized Binary GCD for Modular
// Inversion", 2020): 31 steps at a time are decided on 64-bit approximations (top 33 bits and low
// 31 bits of the two running values), and their combined effect - two small signed factors per
// value - is applied to the full-width numbers once.  About a sixth of the instructions of the
// bit-at-a-time loop above; same result (the inverse is unique).  Takes and returns Montgomery form.
namespace bingcd {

__host__ __device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// r[0..4] = x[0..3] * m
__host__ __device__ __forceinline__ void mul_small(const uint64_t x[4], uint64_t m, uint64_t r[5]) {
  uint64_t carry = 0;
  for (int i = 0; i < 4; ++i) {
    uint64_t lo = x[i] * m, hi = mulhi64(x[i], m);
    lo += carry;
    hi += lo < carry;
    r[i] = lo;
    carry = hi;
  }
  r[4] = carry;
}

__host__ __device__ __forceinline__ void add5(uint64_t r[5], const uint64_t t[5]) {
  uint64_t c = 0;
  for (int i = 0; i < 5; ++i) {
    const uint64_t s = r[i] + t[i], s2 = s + c;
    c = (uint64_t)(s < r[i]) | (uint64_t)(s2 < s);
    r[i] = s2;
  }
}

__host__ __device__ __forceinline__ void neg5(uint64_t r[5]) {
  uint64_t c = 1;
  for (int i = 0; i < 5; ++i) { const uint64_t s = ~r[i] + c; c = (c && s == 0) ? 1 : 0; r[i] = s; }
}

__host__ __device__ __forceinline__ int bitlen4(const uint64_t x[4]) {
  for (int i = 3; i >= 0; --i)
    if (x[i]) return 64 * i + 64 - __builtin_clzll(x[i]);
  return 0;
}

// bits [s, s + 64) of x (zero beyond bit 255).  The limb is picked by selects, not by indexing: a run-time index
// would move the four-limb arrays of the caller from registers to scratch memory, hundreds of cycles per access.
__host__ __device__ __forceinline__ uint64_t bits_from(const uint64_t x[4], int s) {
  const int limb = s >> 6, off = s & 63;
  const uint64_t lo = limb == 0 ? x[0] : (limb == 1 ? x[1] : (limb == 2 ? x[2] : x[3]));
  const uint64_t hi = limb == 0 ? x[1] : (limb == 1 ? x[2] : (limb == 2 ? x[3] : 0));
  uint64_t v = lo >> off;
  if (off) v |= hi << (64 - off);
  return v;
}

// x <- |f x + g y| / 2^31 (the division is exact); returns true if the sign was flipped
__host__ __device__ __forceinline__ bool lincomb_div(uint64_t out[4], const uint64_t x[4], int64_t f, const uint64_t y[4], int64_t g) {
  uint64_t r[5], t[5];
  mul_small(x, (uint64_t)(f < 0 ? -f : f), r);
  if (f < 0) neg5(r);
  mul_small(y, (uint64_t)(g < 0 ? -g : g), t);
  if (g < 0) neg5(t);
  add5(r, t);
  const bool negative = (r[4] >> 63) != 0;
  if (negative) neg5(r);
  for (int i = 0; i < 4; ++i) out[i] = (r[i] >> 31) | (r[i + 1] << 33);
  return negative;
}

// out <- (f u + g v) / 2^31 mod p, u, v in [0, p); p = 1 mod 2^32, so the Montgomery factor is -t mod 2^31
__host__ __device__ __forceinline__ void lincomb_mod(uint64_t out[4], const uint64_t u[4], int64_t f, const uint64_t v[4], int64_t g,
                                                     const uint64_t p[4]) {
  uint64_t a[4], b[4], r[5], t[5];
  auto p_minus = [&](const uint64_t x[4], uint64_t o[4]) {   // p - x
    uint64_t bo = 0;
    for (int i = 0; i < 4; ++i) { const uint64_t d = p[i] - x[i], d2 = d - bo; bo = (uint64_t)(p[i] < x[i]) | (uint64_t)(d < bo); o[i] = d2; }
  };
  if (f < 0) p_minus(u, a); else for (int i = 0; i < 4; ++i) a[i] = u[i];
  if (g < 0) p_minus(v, b); else for (int i = 0; i < 4; ++i) b[i] = v[i];
  mul_small(a, (uint64_t)(f < 0 ? -f : f), r);
  mul_small(b, (uint64_t)(g < 0 ? -g : g), t);
  add5(r, t);
  mul_small(p, (0 - r[0]) & 0x7fffffffull, t);
  add5(r, t);                                       // now a multiple of 2^31, below 3 p 2^31
  uint64_t o[4];
  for (int i = 0; i < 4; ++i) o[i] = (r[i] >> 31) | (r[i + 1] << 33);
  for (int pass = 0; pass < 2; ++pass) {            // into [0, p)
    uint64_t d[4], bo = 0;
    for (int i = 0; i < 4; ++i) { const uint64_t e = o[i] - p[i], e2 = e - bo; bo = (uint64_t)(o[i] < p[i]) | (uint64_t)(e < bo); d[i] = e2; }
    if (!bo) for (int i = 0; i < 4; ++i) o[i] = d[i];
  }
  for (int i = 0; i < 4; ++i) out[i] = o[i];
}

}  // namespace bingcd

__host__ __device__ inline Fe fe_inverse_bingcd(const Fe& x) {
  using namespace bingcd;
  if (fe_is_zero(x)) return x;
  uint64_t a[4], b[4], u[4], v[4], p[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = (uint64_t)x.l[2 * i] | ((uint64_t)x.l[2 * i + 1] << 32);
    p[i] = (uint64_t)Field::P[2 * i] | ((uint64_t)Field::P[2 * i + 1] << 32);
    b[i] = p[i];
    u[i] = (uint64_t)Field::R2[2 * i] | ((uint64_t)Field::R2[2 * i + 1] << 32);   // x = a R: R^2 / x = R / a
    v[i] = 0;
  }
  // invariants: a = u x / R^2, b = v x / R^2 (mod p); gcd(a, b) = 1; ends with a = 0, b = 1
  for (int round = 0; round < 40 && (a[0] | a[1] | a[2] | a[3]); ++round) {
    const int la = bitlen4(a), lb = bitlen4(b), n = la > lb ? la : lb;
    uint64_t xa, xb;
    if (n <= 64) { xa = a[0]; xb = b[0]; }
    else {
      xa = (bits_from(a, n - 33) << 31) | (a[0] & 0x7fffffffull);
      xb = (bits_from(b, n - 33) << 31) | (b[0] & 0x7fffffffull);
    }
    int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    for (int i = 0; i < 31; ++i) {
      if (xa & 1) {
        if (xa < xb) {
          const uint64_t tx = xa; xa = xb; xb = tx;
          int64_t ts = f0; f0 = f1; f1 = ts;
          ts = g0; g0 = g1; g1 = ts;
        }
        xa -= xb; f0 -= f1; g0 -= g1;
      }
      xa >>= 1; f1 <<= 1; g1 <<= 1;
    }
    uint64_t na[4], nb[4], nu[4], nv[4];
    if (lincomb_div(na, a, f0, b, g0)) { f0 = -f0; g0 = -g0; }
    if (lincomb_div(nb, a, f1, b, g1)) { f1 = -f1; g1 = -g1; }
    lincomb_mod(nu, u, f0, v, g0, p);
    lincomb_mod(nv, u, f1, v, g1, p);
    for (int i = 0; i < 4; ++i) { a[i] = na[i]; b[i] = nb[i]; u[i] = nu[i]; v[i] = nv[i]; }
  }
  Fe r;
  for (int i = 0; i < 4; ++i) { r.l[2 * i] = (uint32_t)v[i]; r.l[2 * i + 1] = (uint32_t)(v[i] >> 32); }
  return r;
}

// ---- the same batched binary GCD on 32-bit limbs, the form the GPU runs fast ------------------------------------
// One lane computes one inverse, so its latency is what counts (a block of the grand product waits for it).  Against
// fe_inverse_bingcd: 32-bit limbs and factors (a 64-bit product on this GPU is ~10 instructions, a 32 x 32 + 64
// multiply-add is one), 30 steps per round so that the factors fit 32-bit integers, runs of even steps taken at once
// (count of trailing zeros), every array index a compile-time constant (nothing lives in scratch memory), one
// conditional subtraction after the Montgomery step.  Same result: the inverse is unique.
namespace gcd32 {

constexpr int K = 30;   // steps per round; |f| + |g| <= 2^30 for both rows of the step matrix

// r[0..8] = x[0..7] * m
__host__ __device__ __forceinline__ void mul_small(const uint32_t x[8], uint32_t m, uint32_t r[9]) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (uint64_t)x[i] * m; r[i] = (uint32_t)c; c >>= 32; }
  r[8] = (uint32_t)c;
}

// r += t (9 limbs), returns the carry out
__host__ __device__ __forceinline__ uint32_t add9(uint32_t r[9], const uint32_t t[9]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { const uint64_t s = (uint64_t)r[i] + t[i] + c; r[i] = (uint32_t)s; c = (uint32_t)(s >> 32); }
  return c;
}

// r -= t (9 limbs), returns the borrow out
__host__ __device__ __forceinline__ uint32_t sub9(uint32_t r[9], const uint32_t t[9]) {
  uint32_t b = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { const uint64_t d = (uint64_t)r[i] - t[i] - b; r[i] = (uint32_t)d; b = (uint32_t)(d >> 63); }
  return b;
}

__host__ __device__ __forceinline__ void neg9(uint32_t r[9]) {
  uint32_t c = 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) { const uint64_t s = (uint64_t)(~r[i]) + c; r[i] = (uint32_t)s; c = (uint32_t)(s >> 32); }
}

// out <- |f x + g y| / 2^K (exact); returns true if f x + g y was negative
__host__ __device__ __forceinline__ bool lincomb_div(uint32_t out[8], const uint32_t x[8], int32_t f, const uint32_t y[8], int32_t g) {
  uint32_t r[9], t[9];
  mul_small(x, (uint32_t)(f < 0 ? -f : f), r);
  mul_small(y, (uint32_t)(g < 0 ? -g : g), t);
  bool negative = f < 0;
  if ((f < 0) == (g < 0)) {
    add9(r, t);
  } else if (sub9(r, t)) {         // |f| x < |g| y: the sum has the sign of g
    neg9(r);
    negative = g < 0;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (r[i] >> K) | (r[i + 1] << (32 - K));
  return negative;
}

// out <- (f u + g v) / 2^K mod p for u, v in [0, p); p = 1 mod 2^32, so the Montgomery factor is -t mod 2^K
__host__ __device__ __forceinline__ void lincomb_mod(uint32_t out[8], const uint32_t u[8], int32_t f, const uint32_t v[8], int32_t g) {
  uint32_t a[8], b[8], r[9], t[9];
  {   // a negative factor: p - value instead
    uint32_t bo = 0, bo2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t d = (uint64_t)Field::P[i] - u[i] - bo; bo = (uint32_t)(d >> 63);
      a[i] = f < 0 ? (uint32_t)d : u[i];
      const uint64_t e = (uint64_t)Field::P[i] - v[i] - bo2; bo2 = (uint32_t)(e >> 63);
      b[i] = g < 0 ? (uint32_t)e : v[i];
    }
  }
  mul_small(a, (uint32_t)(f < 0 ? -f : f), r);
  mul_small(b, (uint32_t)(g < 0 ? -g : g), t);
  add9(r, t);                                              // below 2^K p
  const uint32_t m = (0u - r[0]) & ((1u << K) - 1);
  {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += (uint64_t)Field::P[i] * m; t[i] = (uint32_t)c; c >>= 32; }
    t[8] = (uint32_t)c;
  }
  add9(r, t);                                              // a multiple of 2^K, below 2^(K+1) p
  uint32_t o[8], d[8], bo = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (r[i] >> K) | (r[i + 1] << (32 - K));     // below 2 p
#pragma unroll
  for (int i = 0; i < 8; ++i) { const uint64_t e = (uint64_t)o[i] - Field::P[i] - bo; d[i] = (uint32_t)e; bo = (uint32_t)(e >> 63); }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = bo ? o[i] : d[i];
}

// the two 32-bit limbs that start at limb `limb` (limbs beyond the eighth read as zero), picked by selects
__host__ __device__ __forceinline__ uint64_t window(const uint32_t x[8], int limb) {
  uint32_t lo = x[0], hi = x[1];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    lo = limb == i ? x[i] : lo;
    hi = limb == i ? (i + 1 < 8 ? x[i + 1] : 0u) : hi;
  }
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

__host__ __device__ __forceinline__ int ctz64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffsll((unsigned long long)x) - 1;
#else
  return __builtin_ctzll(x);
#endif
}

}  // namespace gcd32

__host__ __device__ inline Fe fe_inverse_gcd32(const Fe& x) {
  using namespace gcd32;
  if (fe_is_zero(x)) return x;
  uint32_t a[8], b[8], u[8], v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = x.l[i]; b[i] = Field::P[i]; u[i] = Field::R2[i]; v[i] = 0; }   // x = a R: R^2 / x = R / a
  // invariants: a = u x / R^2, b = v x / R^2 (mod p); gcd(a, b) = 1; ends with a = 0, b = 1
  for (int round = 0; round < 40; ++round) {
    uint32_t any = 0, top = 0; int hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { any |= a[i]; const uint32_t c = a[i] | b[i]; if (c) { top = c; hi = i; } }
    if (!any) break;
    // n = bit length of max(a, b); approximations: the top 34 bits above the low K bits (exact values if n <= 64)
#if defined(__HIP_DEVICE_COMPILE__)
    const int n = 32 * hi + 32 - __clz((int)top);
#else
    const int n = 32 * hi + 32 - __builtin_clz(top);
#endif
    uint64_t xa, xb;
    if (n <= 64) {
      xa = (uint64_t)a[0] | ((uint64_t)a[1] << 32); xb = (uint64_t)b[0] | ((uint64_t)b[1] << 32);
    } else {
      const int s = n - 34, limb = s >> 5, off = s & 31;          // bits [s, s + 34): off + 34 <= 65, so a third limb may be touched
      const uint64_t wa = window(a, limb), wb = window(b, limb);
      uint64_t ta = wa >> off, tb = wb >> off;
      if (off > 30) {   // the 34 bits reach into limb + 2
        const uint64_t wa2 = window(a, limb + 1), wb2 = window(b, limb + 1);
        ta = (wa2 << (32 - off)) | (ta & ((1ull << (64 - off)) - 1)); tb = (wb2 << (32 - off)) | (tb & ((1ull << (64 - off)) - 1));
      }
      ta &= (1ull << 34) - 1; tb &= (1ull << 34) - 1;
      xa = (ta << K) | (a[0] & ((1u << K) - 1));
      xb = (tb << K) | (b[0] & ((1u << K) - 1));
    }
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    for (int done = 0; done < K;) {
      if (xa & 1) {
        if (xa < xb) {
          const uint64_t tx = xa; xa = xb; xb = tx;
          int32_t ts = f0; f0 = f1; f1 = ts;
          ts = g0; g0 = g1; g1 = ts;
        }
        xa -= xb; f0 -= f1; g0 -= g1;
      }
      // xa is even now: take the run of even steps at once (xa = 0 ends the round's useful work: shift all the rest)
      int t = xa ? ctz64(xa) : K;
      if (t > K - done) t = K - done;
      xa >>= t; f1 <<= t; g1 <<= t;
      done += t;
    }
    uint32_t na[8], nb[8], nu[8], nv[8];
    if (lincomb_div(na, a, f0, b, g0)) { f0 = -f0; g0 = -g0; }
    if (lincomb_div(nb, a, f1, b, g1)) { f1 = -f1; g1 = -g1; }
    lincomb_mod(nu, u, f0, v, g0);
    lincomb_mod(nv, u, f1, v, g1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = na[i]; b[i] = nb[i]; u[i] = nu[i]; v[i] = nv[i]; }
  }
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = v[i];
  return r;
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
// the same store marked non-temporal: the line is not kept in the caches it passes through
__device__ __forceinline__ void fe_store_nt(uint64_t* base, size_t idx, const Fe& v) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  v4u* p = reinterpret_cast<v4u*>(base) + 2 * idx;
  const v4u lo = {v.l[0], v.l[1], v.l[2], v.l[3]}, hi = {v.l[4], v.l[5], v.l[6], v.l[7]};
  __builtin_nontemporal_store(lo, p);
  __builtin_nontemporal_store(hi, p + 1);
}

}  // namespace plk
