/* C restatement of the commitment step for the parity tests at sizes Python cannot reach and for the timed CPU
 * baseline: BLS12-381 G1 and sum_i s_i * P_i by the bucket method.
 *
 * TEST INFRASTRUCTURE ONLY - PARITY UNPINNED: the reference imports `ark_poly_commit::kzg10` (src/main.rs:9) and
 * never calls it (README.md:15-17), so no reference output exists; the sum has one value whatever the order.
 * Execution model mirrors ark-ec 0.3 `VariableBaseMSM::multi_scalar_mul`: windows of c bits processed in parallel
 * (rayon there, OpenMP here), buckets per window, running-sum reduction, then doublings between windows.
 * Base field elements: 6 little-endian uint64 Montgomery limbs (R = 2^384), ark-ff `Fp384` memory. */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef struct { u64 l[6]; } fq;
typedef struct { fq x, y, z; } jac;   /* z = 0: infinity */

static const fq Q = {{0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};
static const fq Q_ONE = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};
static const u64 Q_INV64 = 0x89f3fffcfffcfffdull;   /* -q^-1 mod 2^64 */
/* the scalar field, to take scalars out of Montgomery form */
static const u64 R_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 R_INV64 = 0xfffffffeffffffffull;

static int fq_is_zero(const fq* a) { u64 o = 0; for (int i = 0; i < 6; ++i) o |= a->l[i]; return o == 0; }
static int fq_eq(const fq* a, const fq* b) { u64 o = 0; for (int i = 0; i < 6; ++i) o |= a->l[i] ^ b->l[i]; return o == 0; }
static int fq_geq_q(const u64* t) { for (int i = 5; i >= 0; --i) if (t[i] != Q.l[i]) return t[i] > Q.l[i]; return 1; }
static void fq_sub_q(u64* t) { u128 b = 0; for (int i = 0; i < 6; ++i) { u128 d = (u128)t[i] - Q.l[i] - b; t[i] = (u64)d; b = (d >> 64) & 1; } }

static fq fq_add(fq a, const fq* b) {
  u128 c = 0;
  for (int i = 0; i < 6; ++i) { c += (u128)a.l[i] + b->l[i]; a.l[i] = (u64)c; c >>= 64; }
  if (fq_geq_q(a.l)) fq_sub_q(a.l);   /* q < 2^381: no carry out */
  return a;
}
static fq fq_sub(fq a, const fq* b) {
  u128 bo = 0;
  for (int i = 0; i < 6; ++i) { u128 d = (u128)a.l[i] - b->l[i] - bo; a.l[i] = (u64)d; bo = (d >> 64) & 1; }
  if (bo) { u128 c = 0; for (int i = 0; i < 6; ++i) { c += (u128)a.l[i] + Q.l[i]; a.l[i] = (u64)c; c >>= 64; } }
  return a;
}
static fq fq_dbl(fq a) { return fq_add(a, &a); }
static fq fq_mul(const fq* a, const fq* b) {   /* CIOS */
  u64 t[8] = {0};
  for (int i = 0; i < 6; ++i) {
    u128 c = 0;
    for (int j = 0; j < 6; ++j) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    c += t[6]; t[6] = (u64)c; t[7] = (u64)(c >> 64);
    const u64 m = t[0] * Q_INV64;
    c = (u128)m * Q.l[0] + t[0]; c >>= 64;
    for (int j = 1; j < 6; ++j) { c += (u128)m * Q.l[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t[6]; t[5] = (u64)c; t[6] = t[7] + (u64)(c >> 64);
  }
  if (t[6] || fq_geq_q(t)) fq_sub_q(t);
  fq r; memcpy(r.l, t, 48); return r;
}
static fq fq_sqr(const fq* a) { return fq_mul(a, a); }
static fq fq_inv(const fq* a) {   /* a^(q-2) */
  fq e = Q; e.l[0] -= 2;
  fq acc = Q_ONE;
  for (int bit = 383; bit >= 0; --bit) {
    acc = fq_sqr(&acc);
    if ((e.l[bit >> 6] >> (bit & 63)) & 1) acc = fq_mul(&acc, a);
  }
  return acc;
}

static jac jac_inf(void) { jac r; r.x = Q_ONE; r.y = Q_ONE; memset(&r.z, 0, sizeof(fq)); return r; }
static jac jac_dbl(const jac* p) {   /* dbl-2009-l */
  if (fq_is_zero(&p->z)) return *p;
  fq A = fq_sqr(&p->x), B = fq_sqr(&p->y), C = fq_sqr(&B);
  fq xb = fq_add(p->x, &B), D = fq_sqr(&xb); D = fq_sub(D, &A); D = fq_sub(D, &C); D = fq_dbl(D);
  fq E = fq_add(fq_dbl(A), &A), F = fq_sqr(&E);
  jac r;
  fq D2 = fq_dbl(D); r.x = fq_sub(F, &D2);
  fq C8 = fq_dbl(fq_dbl(fq_dbl(C))), dx = fq_sub(D, &r.x), e = fq_mul(&E, &dx);
  r.y = fq_sub(e, &C8);
  fq yz = fq_mul(&p->y, &p->z); r.z = fq_dbl(yz);
  return r;
}
static jac jac_add(const jac* p, const jac* q) {   /* add-2007-bl */
  if (fq_is_zero(&p->z)) return *q;
  if (fq_is_zero(&q->z)) return *p;
  fq Z1Z1 = fq_sqr(&p->z), Z2Z2 = fq_sqr(&q->z);
  fq U1 = fq_mul(&p->x, &Z2Z2), U2 = fq_mul(&q->x, &Z1Z1);
  fq t1 = fq_mul(&p->y, &q->z), S1 = fq_mul(&t1, &Z2Z2), t2 = fq_mul(&q->y, &p->z), S2 = fq_mul(&t2, &Z1Z1);
  fq H = fq_sub(U2, &U1), r0 = fq_sub(S2, &S1);
  if (fq_is_zero(&H)) return fq_is_zero(&r0) ? jac_dbl(p) : jac_inf();
  fq H2 = fq_dbl(H), I = fq_sqr(&H2), J = fq_mul(&H, &I), rr = fq_dbl(r0), V = fq_mul(&U1, &I);
  jac r;
  fq rr2 = fq_sqr(&rr), V2 = fq_dbl(V);
  r.x = fq_sub(fq_sub(rr2, &J), &V2);
  fq vx = fq_sub(V, &r.x), a = fq_mul(&rr, &vx), sj = fq_mul(&S1, &J), sj2 = fq_dbl(sj);
  r.y = fq_sub(a, &sj2);
  fq zz = fq_add(p->z, &q->z), zz2 = fq_sqr(&zz); zz2 = fq_sub(zz2, &Z1Z1); zz2 = fq_sub(zz2, &Z2Z2);
  r.z = fq_mul(&zz2, &H);
  return r;
}
static jac jac_from_affine(const u64* xy) {
  jac r; memcpy(r.x.l, xy, 48); memcpy(r.y.l, xy + 6, 48);
  if (fq_is_zero(&r.x) && fq_is_zero(&r.y)) return jac_inf();   /* (0, 0) = infinity */
  r.z = Q_ONE; return r;
}

/* scalar out of Montgomery form: x * 1 * R^-1 mod r (one Montgomery reduction) */
static void fr_canonical(const u64* mont, u64* out) {
  u64 t[6] = {mont[0], mont[1], mont[2], mont[3], 0, 0};
  for (int i = 0; i < 4; ++i) {
    const u64 m = t[0] * R_INV64;
    u128 c = (u128)m * R_MOD[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * R_MOD[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t[4]; t[3] = (u64)c; t[4] = (u64)(c >> 64);
  }
  int ge = 1;
  for (int i = 3; i >= 0; --i) if (t[i] != R_MOD[i]) { ge = t[i] > R_MOD[i]; break; }
  if (ge) { u128 b = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)t[i] - R_MOD[i] - b; t[i] = (u64)d; b = (d >> 64) & 1; } }
  memcpy(out, t, 32);
}

/* sum_i scalars[i] * bases[i]; scalars: n x 4 limbs (Montgomery Fr), bases: n x 12 limbs affine; out_xy: 12 limbs */
int orc_g1_msm(const u64* scalars, const u64* bases, size_t n, u64* out_xy) {
  memset(out_xy, 0, 96);
  if (n == 0) return 0;
  int c = 3;
  while (((size_t)1 << (c + 4)) < n && c < 16) ++c;      /* about log2(n) - 3, as ark-ec chooses ln(n) + 2 */
  const int windows = (255 + c - 1) / c;
  u64* canon = (u64*)malloc(n * 32);
  jac* wsum = (jac*)malloc((size_t)windows * sizeof(jac));
  if (!canon || !wsum) { free(canon); free(wsum); return -8; }
  for (size_t i = 0; i < n; ++i) fr_canonical(scalars + 4 * i, canon + 4 * i);
  int failed = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int w = 0; w < windows; ++w) {
    jac* buckets = (jac*)malloc(((size_t)1 << c) * sizeof(jac));
    if (!buckets) { failed = 1; continue; }
    for (size_t b = 0; b < ((size_t)1 << c); ++b) buckets[b] = jac_inf();
    const int bit = w * c, limb = bit >> 6, off = bit & 63;
    for (size_t i = 0; i < n; ++i) {
      u64 v = canon[4 * i + limb] >> off;
      if (off && limb + 1 < 4) v |= canon[4 * i + limb + 1] << (64 - off);
      const u64 d = v & (((u64)1 << c) - 1);
      if (!d) continue;
      jac p = jac_from_affine(bases + 12 * i);
      buckets[d] = jac_add(&buckets[d], &p);
    }
    jac run = jac_inf(), acc = jac_inf();
    for (size_t b = ((size_t)1 << c) - 1; b >= 1; --b) { run = jac_add(&run, &buckets[b]); acc = jac_add(&acc, &run); }
    wsum[w] = acc;
    free(buckets);
  }
  if (failed) { free(canon); free(wsum); return -8; }
  jac total = jac_inf();
  for (int w = windows - 1; w >= 0; --w) {
    for (int k = 0; k < c; ++k) total = jac_dbl(&total);
    total = jac_add(&total, &wsum[w]);
  }
  free(canon); free(wsum);
  if (!fq_is_zero(&total.z)) {
    fq zi = fq_inv(&total.z), zi2 = fq_sqr(&zi), zi3 = fq_mul(&zi2, &zi);
    fq x = fq_mul(&total.x, &zi2), y = fq_mul(&total.y, &zi3);
    memcpy(out_xy, x.l, 48); memcpy(out_xy + 6, y.l, 48);
  }
  (void)fq_eq;
  return 0;
}
