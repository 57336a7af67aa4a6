// BLS12-381 G1 for the commitment step: base-field arithmetic (381-bit prime, Montgomery form with
// R = 2^384 on 12 x 32-bit limbs = the memory of ark-ff 0.3 `Fp384`) and point arithmetic on
// y^2 = x^3 + 4 in Jacobian coordinates.
//
// The reference only imports `ark_poly_commit::kzg10` (src/main.rs:9) and lists commitments as not
// implemented (README.md:15-17); a KZG10 commitment is sum_i f_i * [tau^i] G1, i.e. this arithmetic
// under a multi-scalar multiplication (msm.cuh).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plk {

struct Fq { uint32_t l[12]; };

struct Bls12_381Fq {
  static constexpr uint32_t P[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                     0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                       0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
  static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                      0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
  static constexpr uint32_t INV32 = 0xfffcfffdu;   // -q^-1 mod 2^32
  // the standard generator of the prime-order subgroup, Montgomery form
  static constexpr uint32_t GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                                      0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
  static constexpr uint32_t GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                                      0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
};
using FqP = Bls12_381Fq;

__host__ __device__ __forceinline__ Fq fq_zero() { Fq r; for (int i = 0; i < 12; ++i) r.l[i] = 0; return r; }
__host__ __device__ __forceinline__ Fq fq_one() { Fq r; for (int i = 0; i < 12; ++i) r.l[i] = FqP::ONE[i]; return r; }
__host__ __device__ __forceinline__ bool fq_is_zero(const Fq& a) { uint32_t o = 0; for (int i = 0; i < 12; ++i) o |= a.l[i]; return o == 0; }
__host__ __device__ __forceinline__ bool fq_eq(const Fq& a, const Fq& b) { uint32_t o = 0; for (int i = 0; i < 12; ++i) o |= a.l[i] ^ b.l[i]; return o == 0; }

// t - q if t >= q (t < 2q < 2^384: q has 381 bits, so no 385th bit arises)
__host__ __device__ __forceinline__ Fq fq_reduce_once(const uint32_t t[12]) {
  uint32_t d[12], bo = 0, bi;
#pragma unroll
  for (int j = 0; j < 12; ++j) { d[j] = __builtin_subc(t[j], FqP::P[j], bo, &bi); bo = bi; }
  Fq r;
#pragma unroll
  for (int j = 0; j < 12; ++j) r.l[j] = bo ? t[j] : d[j];
  return r;
}

__host__ __device__ __forceinline__ Fq fq_add(const Fq& a, const Fq& b) {
  uint32_t t[12], c = 0, ci;
#pragma unroll
  for (int j = 0; j < 12; ++j) { t[j] = __builtin_addc(a.l[j], b.l[j], c, &ci); c = ci; }
  return fq_reduce_once(t);
}

__host__ __device__ __forceinline__ Fq fq_sub(const Fq& a, const Fq& b) {
  uint32_t d[12], bo = 0, bi;
#pragma unroll
  for (int j = 0; j < 12; ++j) { d[j] = __builtin_subc(a.l[j], b.l[j], bo, &bi); bo = bi; }
  const uint32_t mask = 0u - bo;
  Fq r; uint32_t c = 0, ci;
#pragma unroll
  for (int j = 0; j < 12; ++j) { r.l[j] = __builtin_addc(d[j], FqP::P[j] & mask, c, &ci); c = ci; }
  return r;
}

__host__ __device__ __forceinline__ Fq fq_neg(const Fq& a) { return fq_is_zero(a) ? a : fq_sub(fq_zero(), a); }
__host__ __device__ __forceinline__ Fq fq_double(const Fq& a) { return fq_add(a, a); }

// Montgomery product a * b * 2^-384 mod q (coarsely integrated operand scanning, as field.cuh's fe_mul)
__host__ __device__ __forceinline__ Fq fq_mul(const Fq& a, const Fq& b) {
  uint32_t t[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    uint64_t pr[12];
    uint32_t c, ci;
#pragma unroll
    for (int j = 0; j < 12; ++j) pr[j] = (uint64_t)a.l[j] * b.l[i] + t[j];
    t[0] = (uint32_t)pr[0];
    c = 0;
#pragma unroll
    for (int j = 1; j < 12; ++j) { t[j] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), c, &ci); c = ci; }
    t[12] = __builtin_addc(t[12], (uint32_t)(pr[11] >> 32), c, &ci);
    t[13] = ci;
    const uint32_t m = t[0] * FqP::INV32;
#pragma unroll
    for (int j = 0; j < 12; ++j) pr[j] = (uint64_t)m * FqP::P[j] + t[j];
    c = 0;
#pragma unroll
    for (int j = 1; j < 12; ++j) { t[j - 1] = __builtin_addc((uint32_t)pr[j], (uint32_t)(pr[j - 1] >> 32), c, &ci); c = ci; }
    t[11] = __builtin_addc(t[12], (uint32_t)(pr[11] >> 32), c, &ci);
    t[12] = t[13] + ci;
  }
  return fq_reduce_once(t);   // t < 2q
}
__host__ __device__ __forceinline__ Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }

// a^(q-2): the inverse (zero stays zero); host-side use and one-off device use
__host__ __device__ inline Fq fq_inverse(const Fq& a) {
  uint32_t e[12];
  for (int i = 0; i < 12; ++i) e[i] = FqP::P[i];
  e[0] -= 2;                                   // q ends in ...aaab: no borrow
  Fq acc = fq_one();
  for (int bit = 383; bit >= 0; --bit) {
    acc = fq_sqr(acc);
    if ((e[bit >> 5] >> (bit & 31)) & 1) acc = fq_mul(acc, a);
  }
  return acc;
}

// ---- memory format: 6 uint64_t per coordinate = three 16-byte pieces
__device__ __forceinline__ Fq fq_load(const uint64_t* base, size_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 3 * idx;
  const uint4 a = p[0], b = p[1], c = p[2];
  Fq r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
  return r;
}
__device__ __forceinline__ void fq_store(uint64_t* base, size_t idx, const Fq& v) {
  uint4* p = reinterpret_cast<uint4*>(base) + 3 * idx;
  p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  p[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}

// ---- points.  Affine points cross the ABI as (x, y): 12 uint64_t, with (0, 0) standing for the point at
// infinity (not on the curve, so unambiguous).  Jacobian (X, Y, Z), Z = 0 = infinity, is internal: 18 uint64_t.
struct G1Affine { Fq x, y; };
struct G1Jac { Fq x, y, z; };

__host__ __device__ __forceinline__ bool g1_affine_is_inf(const G1Affine& p) { return fq_is_zero(p.x) && fq_is_zero(p.y); }
__host__ __device__ __forceinline__ G1Jac g1_inf() { G1Jac r; r.x = fq_one(); r.y = fq_one(); r.z = fq_zero(); return r; }
__host__ __device__ __forceinline__ bool g1_is_inf(const G1Jac& p) { return fq_is_zero(p.z); }
__host__ __device__ __forceinline__ G1Jac g1_from_affine(const G1Affine& p) {
  if (g1_affine_is_inf(p)) return g1_inf();
  G1Jac r; r.x = p.x; r.y = p.y; r.z = fq_one(); return r;
}

// 2 P (a = 0): dbl-2009-l, 2M + 5S
__host__ __device__ inline G1Jac g1_double(const G1Jac& p) {
  if (g1_is_inf(p)) return p;
  const Fq A = fq_sqr(p.x), B = fq_sqr(p.y), C = fq_sqr(B);
  Fq D = fq_sub(fq_sub(fq_sqr(fq_add(p.x, B)), A), C);
  D = fq_double(D);
  const Fq E = fq_add(fq_double(A), A), F = fq_sqr(E);
  G1Jac r;
  r.x = fq_sub(F, fq_double(D));
  Fq C8 = fq_double(fq_double(fq_double(C)));
  r.y = fq_sub(fq_mul(E, fq_sub(D, r.x)), C8);
  r.z = fq_double(fq_mul(p.y, p.z));
  return r;
}

// P + Q, Q affine and finite: madd-2007-bl, 7M + 4S
__host__ __device__ inline G1Jac g1_add_affine(const G1Jac& p, const G1Affine& q) {
  if (g1_affine_is_inf(q)) return p;
  if (g1_is_inf(p)) return g1_from_affine(q);
  const Fq Z1Z1 = fq_sqr(p.z), U2 = fq_mul(q.x, Z1Z1), S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1);
  const Fq H = fq_sub(U2, p.x), rr0 = fq_sub(S2, p.y);
  if (fq_is_zero(H)) return fq_is_zero(rr0) ? g1_double(p) : g1_inf();
  const Fq HH = fq_sqr(H), I = fq_double(fq_double(HH)), J = fq_mul(H, I), rr = fq_double(rr0), V = fq_mul(p.x, I);
  G1Jac r;
  r.x = fq_sub(fq_sub(fq_sqr(rr), J), fq_double(V));
  r.y = fq_sub(fq_mul(rr, fq_sub(V, r.x)), fq_double(fq_mul(p.y, J)));
  r.z = fq_sub(fq_sub(fq_sqr(fq_add(p.z, H)), Z1Z1), HH);
  return r;
}

// P + Q, both Jacobian: add-2007-bl, 11M + 5S
__host__ __device__ inline G1Jac g1_add(const G1Jac& p, const G1Jac& q) {
  if (g1_is_inf(p)) return q;
  if (g1_is_inf(q)) return p;
  const Fq Z1Z1 = fq_sqr(p.z), Z2Z2 = fq_sqr(q.z);
  const Fq U1 = fq_mul(p.x, Z2Z2), U2 = fq_mul(q.x, Z1Z1);
  const Fq S1 = fq_mul(fq_mul(p.y, q.z), Z2Z2), S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1);
  const Fq H = fq_sub(U2, U1), rr0 = fq_sub(S2, S1);
  if (fq_is_zero(H)) return fq_is_zero(rr0) ? g1_double(p) : g1_inf();
  const Fq I = fq_sqr(fq_double(H)), J = fq_mul(H, I), rr = fq_double(rr0), V = fq_mul(U1, I);
  G1Jac r;
  r.x = fq_sub(fq_sub(fq_sqr(rr), J), fq_double(V));
  r.y = fq_sub(fq_mul(rr, fq_sub(V, r.x)), fq_double(fq_mul(S1, J)));
  r.z = fq_mul(fq_sub(fq_sub(fq_sqr(fq_add(p.z, q.z)), Z1Z1), Z2Z2), H);
  return r;
}

// k P for a small scalar (double-and-add from the top bit)
__host__ __device__ inline G1Jac g1_mul_u64(const G1Jac& p, uint64_t k) {
  G1Jac acc = g1_inf();
  for (int bit = 63; bit >= 0; --bit) {
    acc = g1_double(acc);
    if ((k >> bit) & 1) acc = g1_add(acc, p);
  }
  return acc;
}

__host__ __device__ inline G1Affine g1_to_affine(const G1Jac& p) {
  G1Affine r;
  if (g1_is_inf(p)) { r.x = fq_zero(); r.y = fq_zero(); return r; }
  const Fq zi = fq_inverse(p.z), zi2 = fq_sqr(zi);
  r.x = fq_mul(p.x, zi2);
  r.y = fq_mul(p.y, fq_mul(zi2, zi));
  return r;
}

__device__ __forceinline__ G1Affine g1_load_affine(const uint64_t* base, size_t idx) {
  G1Affine p; p.x = fq_load(base, 2 * idx); p.y = fq_load(base, 2 * idx + 1); return p;
}
__device__ __forceinline__ void g1_store_affine(uint64_t* base, size_t idx, const G1Affine& p) {
  fq_store(base, 2 * idx, p.x); fq_store(base, 2 * idx + 1, p.y);
}
__device__ __forceinline__ G1Jac g1_load_jac(const uint64_t* base, size_t idx) {
  G1Jac p; p.x = fq_load(base, 3 * idx); p.y = fq_load(base, 3 * idx + 1); p.z = fq_load(base, 3 * idx + 2); return p;
}
__device__ __forceinline__ void g1_store_jac(uint64_t* base, size_t idx, const G1Jac& p) {
  fq_store(base, 3 * idx, p.x); fq_store(base, 3 * idx + 1, p.y); fq_store(base, 3 * idx + 2, p.z);
}

}  // namespace plk
