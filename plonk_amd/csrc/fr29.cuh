// Redundant 9 x 29-bit-limb arithmetic for the transform kernels.
//
// Why a second representation: on MI355X v_mad_u64_u32 (32x32+64 -> 64) issues at the same
// rate as every carry-propagating add (v_add_co / v_addc_co), so with full 32-bit limbs half of
// a Montgomery product's instructions are carry handling.  With 29-bit limbs a whole column of
// partial products (up to 9 of a*b plus 8 of m*p) accumulates in one 64-bit register through
// chained multiply-adds with no carries at all: 153 multiply-adds + ~50 shifts/masks per
// product instead of ~120 multiply-adds + ~270 adds/moves.
//
// Conventions
//  * F9 = 9 limbs, value = sum l[i] * 2^(29 i).  "Normalised": l[0..7] < 2^29, l[8] holds the rest.
//    "Lazy": limbs may exceed 29 bits (bounds below), value may exceed p.
//  * Montgomery radix here is R' = 2^261.  mont29(a, w') with w' = w * 2^261 mod p returns
//    a * w mod p (times nothing): data stays in the library-wide R = 2^256 Montgomery form and
//    only the constants (twiddles) carry R'.
//  * mont29 input bounds: a lazy with limbs < 1.3 * 2^31 and value < 64 p; b normalised, < 1.5 p.
//    Output: normalised, value < (V_a / 70 + 1) p.
//  * Butterfly (decimation in time): T = mont29(Y, w); X' = X + T; Y' = X - T + Q2, Q2 = 2p written
//    with limbs in [2^29 - 1, 2^30) so the limb-wise subtraction never borrows.
#pragma once
#include "field.cuh"

namespace plk {

struct F9 { uint32_t l[9]; };

constexpr uint32_t M29 = (1u << 29) - 1;

namespace f9c {
// bits [29 i, 29 i + 29) of k * p  (k * p < 2^261)
constexpr uint32_t kp_limb29(int k, int i) {
  // k * p as 9 x 32-bit words
  uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t c = 0;
  for (int j = 0; j < 8; ++j) { c += (uint64_t)Field::P[j] * (uint32_t)k; w[j] = (uint32_t)c; c >>= 32; }
  w[8] = (uint32_t)c;
  const int bit = 29 * i, wi = bit >> 5, o = bit & 31;
  uint64_t v = (uint64_t)w[wi] >> o;
  if (wi + 1 < 9) v |= (uint64_t)w[wi + 1] << (32 - o);
  return (uint32_t)(i == 8 ? v : (v & M29));
}
constexpr uint32_t kp_word32(int k, int j) {
  uint64_t c = 0; uint32_t w = 0;
  for (int t = 0; t <= j && t < 8; ++t) { c += (uint64_t)Field::P[t] * (uint32_t)k; w = (uint32_t)c; c >>= 32; }
  return j < 8 ? w : (uint32_t)c;
}
// 2p with every low limb raised by 2^29 and the borrow taken from the limb above
constexpr uint32_t q2_limb(int i) {
  return i == 0 ? kp_limb29(2, 0) + (1u << 29) : (i < 8 ? kp_limb29(2, i) + (1u << 29) - 1 : kp_limb29(2, 8) - 1);
}
// 4p with every low limb raised by 2^30 (covers a subtrahend whose limbs are sums of two
// normalised limbs)
constexpr uint32_t q4_limb(int i) {
  return i == 0 ? kp_limb29(4, 0) + (1u << 30) : (i < 8 ? kp_limb29(4, i) + (1u << 30) - 2 : kp_limb29(4, 8) - 2);
}
}  // namespace f9c

// ---- conversions -------------------------------------------------------------------------

// 8 x 32-bit words (any value < 2^256) -> normalised F9
__host__ __device__ __forceinline__ F9 f9_from_fe(const Fe& a) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, w = bit >> 5, o = bit & 31;
    uint32_t v = a.l[w] >> o;
    if (o > 3 && w + 1 < 8) v |= a.l[w + 1] << (32 - o);
    r.l[i] = i == 8 ? v : (v & M29);
  }
  return r;
}

// carry propagation: limbs 0..7 below 2^29, limb 8 takes the rest (value unchanged)
__host__ __device__ __forceinline__ void f9_normalize(F9& a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { a.l[i + 1] += a.l[i] >> 29; a.l[i] &= M29; }
}

// normalised F9 with value < 2^256 -> 8 x 32-bit words (no reduction)
__host__ __device__ __forceinline__ Fe fe_pack_f9(const F9& a) {
  Fe r;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int bit = 32 * w, i = bit / 29, o = bit % 29;   // word w starts inside limb i at offset o
    uint32_t v = a.l[i] >> o;
    v |= a.l[i + 1] << (29 - o);
    if (29 - o + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - o);
    r.l[w] = v;
  }
  return r;
}

// lazy F9 with value < 32 p -> canonical element in [0, p) as 8 x 32-bit words, by conditional subtraction of
// 16 p ... p: the plain form, kept as the check of fe_reduce_f9 (tools/test_reduce.hip)
__host__ __device__ __forceinline__ Fe fe_reduce_f9_by_subtractions(F9 a) {
  f9_normalize(a);
  // 261-bit value as 9 x 32-bit words
  uint32_t w[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int bit = 32 * j, i = bit / 29, o = bit % 29;
    uint32_t v = a.l[i] >> o;
    if (i + 1 < 9) v |= a.l[i + 1] << (29 - o);
    if (29 - o + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - o);
    w[j] = v;
  }
  // conditional subtraction of 16p, 8p, 4p, 2p, p
#pragma unroll
  for (int k = 16; k >= 1; k >>= 1) {
    uint32_t d[9], bo = 0, bi;
#pragma unroll
    for (int j = 0; j < 9; ++j) { d[j] = __builtin_subc(w[j], f9c::kp_word32(k, j), bo, &bi); bo = bi; }
#pragma unroll
    for (int j = 0; j < 9; ++j) w[j] = bo ? w[j] : d[j];
  }
  Fe r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.l[j] = w[j];
  return r;
}

// lazy F9 with value v < 32 p -> canonical element in [0, p) as 8 x 32-bit words.
// The quotient floor(v / p) < 32 is estimated from the top limb alone: qh = floor(float(l8) * C) with
// C = (2^232 / p) (1 - 2^-20) rounded down to a float.  l8 <= v / 2^232 and the two float roundings raise the product
// by less than a relative 2^-22, so qh <= v / p; l8 > v / 2^232 - 1 and the factors lower it by less than 1e-4 in all,
// so qh >= floor(v / p) - 1.  Hence v - qh p lies in [0, 2 p): one signed carry chain over the limbs and one
// conditional subtraction, instead of five conditional subtractions.
__host__ __device__ __forceinline__ Fe fe_reduce_f9(F9 a) {
  f9_normalize(a);
  const uint32_t qh = (uint32_t)((float)a.l[8] * Field::QEST);
  int64_t acc = 0;
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    acc += (int64_t)a.l[i];
    acc -= (int64_t)((uint64_t)qh * f9c::kp_limb29(1, i));
    r.l[i] = i < 8 ? ((uint32_t)acc & M29) : (uint32_t)acc;
    acc >>= 29;                                   // arithmetic: a borrow travels as -1
  }
  const Fe w = fe_pack_f9(r);                     // below 2 p < 2^256
  return fe_reduce_once(w.l, 0);
}

// ---- arithmetic --------------------------------------------------------------------------

// a * b * 2^-261 mod p, product scanning with the reduction interleaved column by column.
// p = 1 mod 2^29 for every field with two-adicity >= 29, hence m_k = -t_k mod 2^29 and the p[0]
// term is the bare addition of m_k.  That addition is never made: a reduced column t_k + m_k is a
// multiple of 2^29, so its carry is (t_k + 2^29 - 1) >> 29 whatever m_k is.  The accumulator
// therefore carries the bias B = t + (2^29 - 1) through the nine reduced columns; then
// m_k = -(B - (2^29 - 1)) = ~B (mod 2^29) and the carry is B >> 29: two operations per column
// instead of four.
__device__ __forceinline__ F9 mont29(const F9& a, const F9& b) {
  static_assert(f9c::kp_limb29(1, 0) == 1, "modulus must be 1 mod 2^29");
  uint64_t acc = M29;
  uint32_t m[9];
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i);
    m[k] = ~(uint32_t)acc & M29;
    acc = (acc >> 29) + (k < 8 ? (uint64_t)M29 : 0);
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}

// Two independent products in lockstep.  Each accumulation is kept a strictly sequential chain
// (the empty asm stops the compiler from splitting it into partial sums that must be merged by
// 64-bit adds); interleaving the two chains keeps dependent multiply-adds two issues apart.
#define PLK_PIN(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ void mont29_x2(const F9& a1, const F9& b1, const F9& a2, const F9& b2, F9& r1, F9& r2) {
  uint64_t acc1 = M29, acc2 = M29;   // biased accumulators, as in mont29
  uint32_t m1[9], m2[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      acc1 += (uint64_t)a1.l[i] * b1.l[k - i]; acc2 += (uint64_t)a2.l[i] * b2.l[k - i];
      PLK_PIN(acc1); PLK_PIN(acc2);
    }
#pragma unroll
    for (int i = 0; i < k; ++i) {
      acc1 += (uint64_t)m1[i] * f9c::kp_limb29(1, k - i); acc2 += (uint64_t)m2[i] * f9c::kp_limb29(1, k - i);
      PLK_PIN(acc1); PLK_PIN(acc2);
    }
    m1[k] = ~(uint32_t)acc1 & M29; m2[k] = ~(uint32_t)acc2 & M29;
    acc1 = (acc1 >> 29) + (k < 8 ? (uint64_t)M29 : 0); acc2 = (acc2 >> 29) + (k < 8 ? (uint64_t)M29 : 0);
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) {
      acc1 += (uint64_t)a1.l[i] * b1.l[k - i]; acc2 += (uint64_t)a2.l[i] * b2.l[k - i];
      PLK_PIN(acc1); PLK_PIN(acc2);
    }
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) {
      acc1 += (uint64_t)m1[i] * f9c::kp_limb29(1, k - i); acc2 += (uint64_t)m2[i] * f9c::kp_limb29(1, k - i);
      PLK_PIN(acc1); PLK_PIN(acc2);
    }
    r1.l[k - 9] = (uint32_t)acc1 & M29; r2.l[k - 9] = (uint32_t)acc2 & M29;
    acc1 >>= 29; acc2 >>= 29;
  }
  r1.l[8] = (uint32_t)acc1; r2.l[8] = (uint32_t)acc2;
}
#undef PLK_PIN

// ---- product by a constant with a precomputed quotient (Shoup / Harvey style) ---------------------------------
// For a constant w < p the table holds w itself (canonical, 9 limbs) and wq = floor(w * 2^261 / p).  Then for any
// a < 2^261:   q = floor(a * wq / 2^261)  satisfies  a w / p - a / 2^261 - 1 < q <= a w / p,
// so r = a w - q p lies in [0, (1 + a / 2^261) p): below 1.91 p for a < 64 p, the bound mont29 gives as well.
// r is computed modulo 2^261 (it is below 2^261): the low nine columns of a * w + q * (2^261 - p).
// Of a * wq only columns 7..16 are formed: the dropped columns 0..6 hold at most 7 products each, below 2^60.4 at
// weights up to 2^174, less than 2^238 in all, which lowers q by less than 2^-23 (covered by the slack above:
// 64 p < 0.907 * 2^261).
// Cost: 53 + 45 + 45 = 143 multiply-adds against mont29's 153; measured alone 7 % more products per second at
// the transform kernels' occupancy (tools/ubench_shoup.hip).  The result is a * w with no Montgomery factor, so
// data in the library's R = 2^256 form stays in that form.
namespace f9c {
// limb i of 2^261 - p: the limb-wise complement of p plus one; p's limb 0 is 1, so no carry travels
constexpr uint32_t pbar_limb(int i) { return i == 0 ? M29 : M29 - kp_limb29(1, i); }
}  // namespace f9c

#define PLK_PIN(x) asm volatile("" : "+v"(x))
// Two independent products in lockstep (as mont29_x2): a1 * w1 -> r1, a2 * w2 -> r2; wq1 / wq2 the quotient records.
// Input bounds: a lazy with limbs < 1.3 * 2^31 and value < 64 p; w, wq normalised.  Output: normalised, < 1.91 p.
__device__ __forceinline__ void shoup29_x2(const F9& a1, const F9& w1, const F9& wq1, const F9& a2, const F9& w2, const F9& wq2,
                                           F9& r1, F9& r2) {
  uint64_t acc1 = 0, acc2 = 0;
  uint32_t q1[9], q2[9];
#pragma unroll
  for (int i = 0; i <= 7; ++i) { acc1 += (uint64_t)a1.l[i] * wq1.l[7 - i]; acc2 += (uint64_t)a2.l[i] * wq2.l[7 - i]; PLK_PIN(acc1); PLK_PIN(acc2); }
  acc1 >>= 29; acc2 >>= 29;
#pragma unroll
  for (int i = 0; i <= 8; ++i) { acc1 += (uint64_t)a1.l[i] * wq1.l[8 - i]; acc2 += (uint64_t)a2.l[i] * wq2.l[8 - i]; PLK_PIN(acc1); PLK_PIN(acc2); }
  acc1 >>= 29; acc2 >>= 29;
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) { acc1 += (uint64_t)a1.l[i] * wq1.l[k - i]; acc2 += (uint64_t)a2.l[i] * wq2.l[k - i]; PLK_PIN(acc1); PLK_PIN(acc2); }
    q1[k - 9] = (uint32_t)acc1 & M29; q2[k - 9] = (uint32_t)acc2 & M29;
    acc1 >>= 29; acc2 >>= 29;
  }
  q1[8] = (uint32_t)acc1; q2[8] = (uint32_t)acc2;
  acc1 = 0; acc2 = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc1 += (uint64_t)a1.l[i] * w1.l[k - i]; acc2 += (uint64_t)a2.l[i] * w2.l[k - i]; PLK_PIN(acc1); PLK_PIN(acc2); }
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc1 += (uint64_t)q1[i] * f9c::pbar_limb(k - i); acc2 += (uint64_t)q2[i] * f9c::pbar_limb(k - i); PLK_PIN(acc1); PLK_PIN(acc2); }
    r1.l[k] = (uint32_t)acc1 & M29; r2.l[k] = (uint32_t)acc2 & M29;
    acc1 >>= 29; acc2 >>= 29;
  }
}
#undef PLK_PIN

// host: the quotient record of a canonical constant, wq = floor(w * 2^261 / p), by bit-at-a-time long division
// (table set-up only: a thousand records per plan)
inline F9 f9_shoup_quotient_host(const Fe& w_canonical) {
  uint32_t rem[9] = {0}, quo[17] = {0};
  uint32_t p9[9];
  for (int i = 0; i < 8; ++i) p9[i] = Field::P[i];
  p9[8] = 0;
  for (int b = 255 + 261; b >= 0; --b) {
    const int src = b - 261;    // bit of w that sits at position b of w << 261
    uint32_t carry = (src >= 0) ? (w_canonical.l[src >> 5] >> (src & 31)) & 1u : 0u;
    for (int i = 0; i < 9; ++i) { const uint32_t nc = rem[i] >> 31; rem[i] = (rem[i] << 1) | carry; carry = nc; }
    bool ge = true;
    for (int i = 8; i >= 0; --i) if (rem[i] != p9[i]) { ge = rem[i] > p9[i]; break; }
    if (ge) {
      uint64_t bo = 0;
      for (int i = 0; i < 9; ++i) { const uint64_t d = (uint64_t)rem[i] - p9[i] - bo; rem[i] = (uint32_t)d; bo = (d >> 63) & 1; }
      quo[b >> 5] |= 1u << (b & 31);
    }
  }
  F9 r;
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, wd = bit >> 5, o = bit & 31;
    uint64_t x = quo[wd];
    if (wd + 1 < 17) x |= (uint64_t)quo[wd + 1] << 32;
    r.l[i] = (uint32_t)(x >> o) & M29;
  }
  return r;
}

// a * b * 2^-256 mod p: the library's own Montgomery product on 29-bit limbs, for point-wise kernels whose two
// operands are both data.  Eight columns are reduced as in mont29; of the ninth only the low 24 bits are
// cleared (8 * 29 + 24 = 256), and the columns above it are re-cut 24 bits higher.
// Input: a lazy (limbs < 1.3 * 2^31, value V_a p), b normalised (value V_b p).  Output: normalised, value
// below (0.4527 V_a V_b + 1) p  (p / 2^256 < 0.4527).
__host__ __device__ __forceinline__ F9 mont256(const F9& a_in, const F9& b_in) {
  F9 a = a_in, b = b_in;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PLK_MONT256_NO_FENCE)   // (the macro: tools/probe_mont256.hip only)
  // The compiler must not know how many bits a limb has.  Root cause of the wrong top limb seen without these
  // fences (ROCm 7.2 clang, gfx950; tools/probe_mont256.hip reproduces it, 16322 of 16384 products wrong): m[8] is
  // `x & 0xffffff` and p's top limb a 23-bit constant, so their product is first formed as a 24-bit multiply, whose
  // operands' upper 8 bits are don't-care - which lets the combiner drop the `and`; that product is then folded with
  // the neighbouring a[8] * b[8] pair (v_mul_u32_u24 + v_mul_hi_u32_u24, top limbs of freshly cut operands) into ONE
  // full-width multiply-add,
  //     v_mad_u64_u32 v[30:31], s[0:1], v51, s3, v[2:3]        ; v51 = 0 - acc, s3 = 0x73eda7
  // which reads all 32 bits of the unmasked v51, while the masked copy (v_and_b32 v18, 0xffffff, v51) feeds the
  // other products of m[8].  With the limb ranges hidden the same product is v_mad_u64_u32 of the masked register.
  // mont29, mont29_x2 and shoup29_x2 mask to 29 bits and multiply by 29-bit limbs, so no 24-bit multiply is ever
  // formed from a masked value there; tests/cpp/test_constraints_dev.hip checks them on freshly cut operands too.
#pragma unroll
  for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(a.l[i]), "+v"(b.l[i]));
#endif
  uint64_t acc = M29;
  uint32_t m[9], c[10];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i);
    m[k] = ~(uint32_t)acc & M29;
    acc = (acc >> 29) + (k < 7 ? (uint64_t)M29 : 0);
  }
#pragma unroll
  for (int i = 0; i <= 8; ++i) acc += (uint64_t)a.l[i] * b.l[8 - i];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += (uint64_t)m[i] * f9c::kp_limb29(1, 8 - i);
  m[8] = (0u - (uint32_t)acc) & ((1u << 24) - 1);
  acc += m[8];                                   // the p[0] = 1 term: the low 24 bits are now zero
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PLK_MONT256_NO_FENCE)
  asm volatile("" : "+v"(m[8]));
#endif
  c[0] = (uint32_t)acc & M29;
  acc >>= 29;
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * f9c::kp_limb29(1, k - i);
    c[k - 8] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  c[9] = (uint32_t)acc;
  F9 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = ((c[i] >> 24) | (c[i + 1] << 5)) & M29;
  r.l[8] = (c[8] >> 24) | (c[9] << 5);
  return r;
}

__host__ __device__ __forceinline__ F9 f9_add(const F9& a, const F9& b) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}

// a - t + 2p, limb-wise; t normalised with value < 2p
__host__ __device__ __forceinline__ F9 f9_sub_q2(const F9& a, const F9& t) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + (f9c::q2_limb(i) - t.l[i]);
  return r;
}

// a - t + 4p, limb-wise; t = sum of two normalised values (limbs <= 2^30 - 2, value < 4p - 2^233)
__host__ __device__ __forceinline__ F9 f9_sub_q4(const F9& a, const F9& t) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + (f9c::q4_limb(i) - t.l[i]);
  return r;
}

// a - t + K p, limb-wise; t normalised with value <= (K - 1) p
template <int K>
__host__ __device__ __forceinline__ F9 f9_sub_k(const F9& a, const F9& t) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    r.l[i] = a.l[i] + ((i == 0 ? f9c::kp_limb29(K, 0) + (1u << 29) : (i < 8 ? f9c::kp_limb29(K, i) + (1u << 29) - 1 : f9c::kp_limb29(K, 8) - 1)) - t.l[i]);
  return r;
}

// ---- twiddle records: 9 limbs padded to 12 words (three 16-byte loads)
__device__ __forceinline__ F9 f9_load_tw(const uint32_t* table, size_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(table) + 3 * idx;
  const uint4 a = p[0], b = p[1], c = p[2];
  F9 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x;
  return r;
}
__device__ __forceinline__ void f9_store_tw(uint32_t* table, size_t idx, const F9& v) {
  uint4* p = reinterpret_cast<uint4*>(table) + 3 * idx;
  p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  p[2] = make_uint4(v.l[8], 0, 0, 0);
}

// w (library Montgomery form, R = 2^256) -> twiddle record w * 2^261 mod p
__host__ __device__ __forceinline__ F9 f9_twiddle_from_fe(Fe w) {
#pragma unroll 1
  for (int i = 0; i < 5; ++i) w = fe_double(w);
  return f9_from_fe(w);
}

}  // namespace plk
