// BLS12-381 base field and G1 point arithmetic on 14 x 28-bit redundant limbs: the working form of the
// multi-scalar multiplication (msm.cuh).  g1.cuh keeps the 12 x 32-bit form of the ABI.
//
// Why: on MI355X a 32 x 32 + 64 multiply-add issues at the rate of a carry-propagating add, and every
// add-with-carry that follows another needs a wait state, so a 12 x 32-bit Montgomery product costs ~1300
// instructions of which 288 multiply.  With 28-bit limbs a whole column of partial products (14 of a * b
// and 14 of m * q) is summed in one 64-bit register by chained multiply-adds: ~500 instructions per product,
// and additions / subtractions are limb-wise with no carries at all.
//
// Conventions
//  * F14: value = sum l[i] * 2^(28 i).  "Normalised": l[0..12] < 2^28, l[13] holds the rest.
//    Values are residues mod q kept lazily: any non-negative representative, bounded as stated per function.
//  * Montgomery radix 2^392: an element x is stored as x * 2^392 mod q (plus a multiple of q).
//  * mul28(a, b): needs max limb(a) * max limb(b) <= 2^60 and value(a) * value(b) <= 1800 q^2; returns a
//    normalised value below 1.75 q.  (q / 2^392 < 1 / 2520, so a b / 2^392 + q < (1800 / 2520 + 1) q.)
//  * a - b is a + (k q - b) with k q written so that no limb goes negative: bias<K, S> = K q with limbs 0..12
//    raised by 2^S (and the limb above lowered to match), for b with limbs < 2^S and value <= (K - 1/4) q.
//  * Point coordinates are stored normalised with values below 7.75 q; the bounds of every intermediate are
//    checked by tests/test_fq28_bounds.py, which replays the formulas below on intervals.
#pragma once
#include "g1.cuh"

namespace plk {

struct F14 { uint32_t l[14]; };
constexpr uint32_t M28 = (1u << 28) - 1;

namespace fq28c {
constexpr uint32_t Q[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                            0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
constexpr uint32_t INV = 0xffcfffdu;     // -1 / q mod 2^28
constexpr uint32_t ONE[14] = {0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu,
                              0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au};           // 2^392 mod q
constexpr uint32_t TO_INTERNAL[14] = {0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                                      0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau};   // 2^400 mod q
constexpr uint32_t TO_ABI[14] = {0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                                 0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u};        // 2^384 mod q
// limb i of k * q (k < 2^4), limbs 0..12 of 28 bits, limb 13 the rest
constexpr uint32_t kq_limb(int k, int i) {
  uint64_t c = 0; uint32_t v = 0;
  for (int j = 0; j <= i; ++j) { c += (uint64_t)Q[j] * (uint32_t)k; v = j < 13 ? (uint32_t)(c & M28) : (uint32_t)c; c >>= 28; }
  return v;
}
constexpr uint32_t bias_limb(int k, int s, int i) {
  return i == 0 ? kq_limb(k, 0) + (1u << s) : (i < 13 ? kq_limb(k, i) + (1u << s) - (1u << (s - 28)) : kq_limb(k, 13) - (1u << (s - 28)));
}
}  // namespace fq28c

__host__ __device__ __forceinline__ F14 f14_zero() { F14 r; for (int i = 0; i < 14; ++i) r.l[i] = 0; return r; }
__host__ __device__ __forceinline__ F14 f14_one() { F14 r; for (int i = 0; i < 14; ++i) r.l[i] = fq28c::ONE[i]; return r; }
__host__ __device__ __forceinline__ bool f14_all_zero(const F14& a) { uint32_t o = 0; for (int i = 0; i < 14; ++i) o |= a.l[i]; return o == 0; }

// carry propagation: limbs 0..12 below 2^28, limb 13 takes the rest (value unchanged)
__host__ __device__ __forceinline__ void f14_normalize(F14& a) {
#pragma unroll
  for (int i = 0; i < 13; ++i) { a.l[i + 1] += a.l[i] >> 28; a.l[i] &= M28; }
}

__host__ __device__ __forceinline__ F14 f14_add(const F14& a, const F14& b) {
  F14 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}
// a * 2^S, limb-wise
template <int S>
__host__ __device__ __forceinline__ F14 f14_shl(const F14& a) {
  F14 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.l[i] = a.l[i] << S;
  return r;
}
// a - b + K q, limb-wise; b has limbs < 2^S and value <= (K - 1/4) q
template <int K, int S>
__host__ __device__ __forceinline__ F14 f14_sub(const F14& a, const F14& b) {
  F14 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.l[i] = a.l[i] + (fq28c::bias_limb(K, S, i) - b.l[i]);
  return r;
}

#ifdef __HIP_DEVICE_COMPILE__
#define PLK_PIN28(x) asm volatile("" : "+v"(x))     // keeps a column a single chain of multiply-adds
// Products are scheduled one after the other: interleaving independent products (the compiler's choice when left
// alone) needs more registers than a wave has and ends in hundreds of scratch reloads per point addition.
#define PLK_PRODUCT_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PLK_PIN28(x)
#define PLK_PRODUCT_FENCE()
#endif

// a * b / 2^392 mod q: product scanning with the reduction interleaved column by column.  A column is summed
// by two chains of multiply-adds issued alternately - the partial products a_i b_j on top of the carry, and the
// reduction terms m_i q_j - so that a multiply-add never waits for the one issued just before it.
__host__ __device__ __forceinline__ F14 mul28(const F14& a, const F14& b) {
  PLK_PRODUCT_FENCE();
  uint64_t carry = 0;
  uint32_t m[14];
  F14 r;
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    uint64_t t = 0, u = carry;
#pragma unroll
    for (int i = 0; i <= k; ++i) {      // the reduction terms newest m first: none of them is ready before this column
      t += (uint64_t)a.l[i] * b.l[k - i]; PLK_PIN28(t);
      if (i < k) { u += (uint64_t)m[k - 1 - i] * fq28c::Q[1 + i]; PLK_PIN28(u); }
    }
    u += t;
    m[k] = ((uint32_t)u * fq28c::INV) & M28;
    u += (uint64_t)m[k] * fq28c::Q[0];
    carry = u >> 28;
  }
#pragma unroll
  for (int k = 14; k < 27; ++k) {
    uint64_t t = 0, u = carry;
#pragma unroll
    for (int i = k - 13; i <= 13; ++i) {
      t += (uint64_t)a.l[i] * b.l[k - i]; PLK_PIN28(t);
      u += (uint64_t)m[k - i] * fq28c::Q[i]; PLK_PIN28(u);
    }
    u += t;
    r.l[k - 14] = (uint32_t)u & M28;
    carry = u >> 28;
  }
  r.l[13] = (uint32_t)carry;
  PLK_PRODUCT_FENCE();
  return r;
}

// a * a / 2^392 mod q: the symmetric partial products once, against the doubled operand (limbs(a) <= 2^30)
__host__ __device__ __forceinline__ F14 sqr28(const F14& a) {
  PLK_PRODUCT_FENCE();
  uint64_t carry = 0;
  uint32_t m[14];
  F14 r, a2;
#pragma unroll
  for (int i = 0; i < 14; ++i) a2.l[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    uint64_t t = 0, u = carry;
#pragma unroll
    for (int i = 0; i < k; ++i) {
      if (2 * i < k) { t += (uint64_t)a2.l[i] * a.l[k - i]; PLK_PIN28(t); }
      else if (2 * i == k) { t += (uint64_t)a.l[i] * a.l[i]; PLK_PIN28(t); }
      u += (uint64_t)m[k - 1 - i] * fq28c::Q[1 + i]; PLK_PIN28(u);
    }
    if (k == 0) t = (uint64_t)a.l[0] * a.l[0];
    u += t;
    m[k] = ((uint32_t)u * fq28c::INV) & M28;
    u += (uint64_t)m[k] * fq28c::Q[0];
    carry = u >> 28;
  }
#pragma unroll
  for (int k = 14; k < 27; ++k) {
    uint64_t t = 0, u = carry;
#pragma unroll
    for (int i = k - 13; i <= 13; ++i) {
      if (2 * i < k) { t += (uint64_t)a2.l[i] * a.l[k - i]; PLK_PIN28(t); }
      else if (2 * i == k) { t += (uint64_t)a.l[i] * a.l[i]; PLK_PIN28(t); }
      u += (uint64_t)m[k - i] * fq28c::Q[i]; PLK_PIN28(u);
    }
    u += t;
    r.l[k - 14] = (uint32_t)u & M28;
    carry = u >> 28;
  }
  r.l[13] = (uint32_t)carry;
  PLK_PRODUCT_FENCE();
  return r;
}

// is the value of a product (normalised, below 2 q) zero mod q?  It is then 0 or q itself.
__host__ __device__ __forceinline__ bool f14_product_is_zero(const F14& a) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) { z |= a.l[i]; e |= a.l[i] ^ fq28c::Q[i]; }
  return z == 0 || e == 0;
}

// ---- conversions to and from the ABI form (g1.cuh's Fq: 12 x 32-bit words, radix 2^384, canonical)
__host__ __device__ __forceinline__ F14 f14_from_fq(const Fq& a) {
  F14 s;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int bit = 28 * i, w = bit >> 5, o = bit & 31;
    uint32_t v = a.l[w] >> o;
    if (o > 4 && w + 1 < 12) v |= a.l[w + 1] << (32 - o);
    s.l[i] = i == 13 ? v : (v & M28);
  }
  F14 c;
#pragma unroll
  for (int i = 0; i < 14; ++i) c.l[i] = fq28c::TO_INTERNAL[i];
  return mul28(s, c);
}
// any a within mul28's input bounds
__host__ __device__ __forceinline__ Fq fq_from_f14(const F14& a) {
  F14 c;
#pragma unroll
  for (int i = 0; i < 14; ++i) c.l[i] = fq28c::TO_ABI[i];
  const F14 v = mul28(a, c);                 // normalised, below 2 q < 2^384
  uint32_t t[12];
#pragma unroll
  for (int w = 0; w < 12; ++w) {
    const int bit = 32 * w, i = bit / 28, o = bit % 28;     // word w starts inside limb i at offset o
    uint32_t x = v.l[i] >> o;
    if (i + 1 < 14) x |= v.l[i + 1] << (28 - o);
    if (28 - o + 28 < 32 && i + 2 < 14) x |= v.l[i + 2] << (56 - o);
    t[w] = x;
  }
  return fq_reduce_once(t);
}

// ---- memory: a coordinate takes 16 words (four 16-byte pieces, the last two words unused)
__device__ __forceinline__ F14 f14_load(const uint32_t* base, size_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 4 * idx;
  const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  F14 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w; r.l[12] = d.x; r.l[13] = d.y;
  return r;
}
__device__ __forceinline__ void f14_store(uint32_t* base, size_t idx, const F14& v) {
  uint4* p = reinterpret_cast<uint4*>(base) + 4 * idx;
  p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  p[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
  p[3] = make_uint4(v.l[12], v.l[13], 0, 0);
}

// ---- points.  Affine (x, y) with all limbs zero = infinity: 32 words; Jacobian (X, Y, Z), Z all zero = infinity
// (only ever produced as exact zeros, never as another multiple of q): 48 words.
struct A28 { F14 x, y; };
struct J28 { F14 x, y, z; };
constexpr int A28_WORDS = 32, J28_WORDS = 48;

__host__ __device__ __forceinline__ bool a28_is_inf(const A28& p) { return f14_all_zero(p.x) && f14_all_zero(p.y); }
__host__ __device__ __forceinline__ bool j28_is_inf(const J28& p) { return f14_all_zero(p.z); }
__host__ __device__ __forceinline__ J28 j28_inf() { J28 r; r.x = f14_one(); r.y = f14_one(); r.z = f14_zero(); return r; }
__host__ __device__ __forceinline__ J28 j28_from_affine(const A28& p) {
  if (a28_is_inf(p)) return j28_inf();
  J28 r; r.x = p.x; r.y = p.y; r.z = f14_one(); return r;
}
__host__ __device__ __forceinline__ A28 a28_from_abi(const G1Affine& p) {
  A28 r;
  if (g1_affine_is_inf(p)) { r.x = f14_zero(); r.y = f14_zero(); return r; }
  r.x = f14_from_fq(p.x); r.y = f14_from_fq(p.y);
  return r;
}
__host__ __device__ __forceinline__ G1Jac j28_to_abi(const J28& p) {
  if (j28_is_inf(p)) return g1_inf();
  G1Jac r; r.x = fq_from_f14(p.x); r.y = fq_from_f14(p.y); r.z = fq_from_f14(p.z);
  return r;
}

// 2 P on y^2 = x^3 + 4: A = X^2, B = Y^2, D = 4 X B, E = 3 A, X3 = E^2 - 2 D, Y3 = E (D - X3) - 8 B^2, Z3 = 2 Y Z
// (dbl-2009-l with its squarings of sums turned into products, which keeps every value small): 4M + 3S
__host__ __device__ __forceinline__ J28 j28_double_inl(const J28& p) {
  if (j28_is_inf(p)) return p;
  const F14 A = sqr28(p.x), B = sqr28(p.y), B4 = f14_shl<2>(B);
  const F14 D = mul28(p.x, B4);
  F14 E = f14_add(f14_shl<1>(A), A);
  f14_normalize(E);
  J28 r;
  r.x = f14_sub<4, 29>(sqr28(E), f14_shl<1>(D));
  f14_normalize(r.x);
  const F14 C8 = mul28(f14_shl<1>(B), B4);
  r.y = f14_sub<2, 28>(mul28(E, f14_sub<8, 28>(D, r.x)), C8);
  f14_normalize(r.y);
  r.z = f14_shl<1>(mul28(p.y, p.z));
  f14_normalize(r.z);
  return r;
}
// the same as a call; the copy makes the callee read its argument from memory once instead of at every use
__host__ __device__ __noinline__ J28 j28_double(const J28& p) { const J28 c = p; return j28_double_inl(c); }

// P + Q, Q affine: madd-2007-bl, 7M + 4S
// (INLINE_DOUBLING: the rare doubling branch is expanded in place instead of called, for kernels that must not
// contain calls - a callee's register need counts for the whole kernel)
template <bool INLINE_DOUBLING = false>
__host__ __device__ __forceinline__ J28 j28_add_affine(const J28& p, const A28& q) {
  if (a28_is_inf(q)) return p;
  if (j28_is_inf(p)) return j28_from_affine(q);
  const F14 Z1Z1 = sqr28(p.z), U2 = mul28(q.x, Z1Z1), S2 = mul28(mul28(q.y, p.z), Z1Z1);
  const F14 H = f14_sub<8, 28>(U2, p.x), HH = sqr28(H);
  F14 rr = f14_shl<1>(f14_sub<8, 28>(S2, p.y));
  f14_normalize(rr);
  if (f14_product_is_zero(HH)) {               // Q = +-P.  (The copy keeps p itself out of memory: j28_double is a call.)
    if (!f14_product_is_zero(mul28(rr, f14_one()))) return j28_inf();
    if (INLINE_DOUBLING) return j28_double_inl(p);
    const J28 copy = p;
    return j28_double(copy);
  }
  const F14 I = f14_shl<2>(HH), J = mul28(H, I), V = mul28(p.x, I);
  J28 r;
  r.x = f14_sub<4, 29>(f14_sub<2, 28>(sqr28(rr), J), f14_shl<1>(V));
  f14_normalize(r.x);
  r.y = f14_sub<4, 29>(mul28(rr, f14_sub<8, 28>(V, r.x)), f14_shl<1>(mul28(p.y, J)));
  f14_normalize(r.y);
  r.z = f14_sub<2, 28>(f14_sub<2, 28>(sqr28(f14_add(p.z, H)), Z1Z1), HH);
  f14_normalize(r.z);
  return r;
}

// P + Q, both Jacobian: add-2007-bl, 11M + 5S
__host__ __device__ __forceinline__ J28 j28_add_inl(const J28& p, const J28& q) {
  if (j28_is_inf(p)) return q;
  if (j28_is_inf(q)) return p;
  const F14 Z1Z1 = sqr28(p.z), Z2Z2 = sqr28(q.z);
  const F14 U1 = mul28(p.x, Z2Z2), U2 = mul28(q.x, Z1Z1);
  const F14 S1 = mul28(mul28(p.y, q.z), Z2Z2), S2 = mul28(mul28(q.y, p.z), Z1Z1);
  const F14 H = f14_sub<2, 28>(U2, U1), HH = sqr28(H);
  F14 rr = f14_shl<1>(f14_sub<2, 28>(S2, S1));
  f14_normalize(rr);
  if (f14_product_is_zero(HH)) {
    if (!f14_product_is_zero(mul28(rr, f14_one()))) return j28_inf();
    const J28 copy = p;
    return j28_double(copy);
  }
  const F14 I = f14_shl<2>(HH), J = mul28(H, I), V = mul28(U1, I);
  J28 r;
  r.x = f14_sub<4, 29>(f14_sub<2, 28>(sqr28(rr), J), f14_shl<1>(V));
  f14_normalize(r.x);
  r.y = f14_sub<4, 29>(mul28(rr, f14_sub<8, 28>(V, r.x)), f14_shl<1>(mul28(S1, J)));
  f14_normalize(r.y);
  r.z = mul28(f14_sub<2, 28>(f14_sub<2, 28>(sqr28(f14_add(p.z, q.z)), Z1Z1), Z2Z2), H);
  return r;
}
__host__ __device__ __noinline__ J28 j28_add(const J28& p, const J28& q) { const J28 a = p, b = q; return j28_add_inl(a, b); }

__device__ __forceinline__ A28 a28_load(const uint32_t* base, size_t idx) {
  A28 p; p.x = f14_load(base, 2 * idx); p.y = f14_load(base, 2 * idx + 1); return p;
}
__device__ __forceinline__ void a28_store(uint32_t* base, size_t idx, const A28& p) {
  f14_store(base, 2 * idx, p.x); f14_store(base, 2 * idx + 1, p.y);
}
__device__ __forceinline__ J28 j28_load(const uint32_t* base, size_t idx) {
  J28 p; p.x = f14_load(base, 3 * idx); p.y = f14_load(base, 3 * idx + 1); p.z = f14_load(base, 3 * idx + 2); return p;
}
__device__ __forceinline__ void j28_store(uint32_t* base, size_t idx, const J28& p) {
  f14_store(base, 3 * idx, p.x); f14_store(base, 3 * idx + 1, p.y); f14_store(base, 3 * idx + 2, p.z);
}

}  // namespace plk
