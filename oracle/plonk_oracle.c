/* CPU restatement (plain C) of the PLONK polynomial hot path of wu-s-john/PLONK.
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never by the product library.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and does not build
 * (src/main.rs:3); its arithmetic is in un-vendored crates (ark-ff 0.3.0 Cargo.lock:50-53,
 * ark-poly 0.3.0 Cargo.lock:125-128).  This file restates their published algorithms
 * (4 x u64 Montgomery with R = 2^256, binary-Euclid inversion, radix-2 DFT over the
 * two-adic subgroup) under the reference's own loops, and is itself checked against the
 * big-integer oracle in plonk_oracle.py and the committed golden vectors.
 *
 * Execution model mirrors the reference: transforms run on all host threads (ark-poly's
 * `parallel` feature is on in the reference's lock file, Cargo.lock:135), the grand-product
 * and constraint loops are single-threaded exactly as written in the reference.
 *
 * Elements are 4 little-endian u64 Montgomery limbs (ark-ff Fp256 memory layout).
 * Field: BLS12-381 scalar field.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef struct { u64 l[4]; } fe;

/* The field: ORACLE_FIELD = 0 (default) the BLS12-381 scalar field, 1 the scalar field of Tweedledee
 * (2^254 + 4707489544292117082687961190295928833; liboracle_tweedle.so).  Constants as printed by
 * tools/derive_field_constants.py; the Tweedle generator (5) and root are self-derived, see there. */
#ifndef ORACLE_FIELD
#define ORACLE_FIELD 0
#endif
#if ORACLE_FIELD == 0
static const fe MODULUS = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
static const fe R_ONE   = {{0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full}};
static const fe R2      = {{0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull}};
/* GENERATOR^((p-1)/2^32) with GENERATOR = 7, canonical form (ark-ff TWO_ADIC_ROOT_OF_UNITY) */
static const fe ROOT_CANON = {{0x3829971f439f0d2bull, 0xb63683508c2280b9ull, 0xd09b681922c813b4ull, 0x16a2a19edfe81f20ull}};
#define INV64 0xfffffffeffffffffull /* -p^-1 mod 2^64 */
#define TWO_ADICITY 32
#elif ORACLE_FIELD == 1
static const fe MODULUS = {{0x842cafd400000001ull, 0x038aa127696286c9ull, 0x0000000000000000ull, 0x4000000000000000ull}};
static const fe R_ONE   = {{0x7379f083fffffffdull, 0xf5601c89c3d86ba3ull, 0xffffffffffffffffull, 0x3fffffffffffffffull}};
static const fe R2      = {{0x8595fa8000000010ull, 0x7e16a565c6895230ull, 0xf4c0e6fcb03aa0a2ull, 0x00c8ad9106886013ull}};
/* GENERATOR^((p-1)/2^34) with GENERATOR = 5 (self-derived), canonical form */
static const fe ROOT_CANON = {{0x1cbd3234869d57ecull, 0x0a287dd1b8084fbfull, 0xf1dbcb645a987293ull, 0x113efc510dc03c0bull}};
#define INV64 0x842cafd3ffffffffull
#define TWO_ADICITY 34
#else
#error "unknown ORACLE_FIELD"
#endif

enum { ORC_OK = 0, ORC_ERR_LENGTH = -1, ORC_ERR_NOT_POW2 = -2, ORC_ERR_TWO_ADICITY = -3,
       ORC_ERR_ZERO_DENOMINATOR = -4, ORC_ERR_SIGMA_RANGE = -5, ORC_ERR_NOMEM = -8 };

/* ------------------------------------------------------------------ field */

static inline int fe_geq(const fe* a, const fe* b) {
  for (int i = 3; i >= 0; --i) { if (a->l[i] != b->l[i]) return a->l[i] > b->l[i]; }
  return 1;
}
static inline void fe_sub_raw(fe* a, const fe* b) {
  u64 borrow = 0;
  for (int i = 0; i < 4; ++i) { u128 d = (u128)a->l[i] - b->l[i] - borrow; a->l[i] = (u64)d; borrow = (u64)(d >> 64) & 1; }
}
static inline void fe_add_raw(fe* a, const fe* b) {
  u64 carry = 0;
  for (int i = 0; i < 4; ++i) { u128 s = (u128)a->l[i] + b->l[i] + carry; a->l[i] = (u64)s; carry = (u64)(s >> 64); }
}
static inline fe fe_add(fe a, const fe b) { fe_add_raw(&a, &b); if (fe_geq(&a, &MODULUS)) fe_sub_raw(&a, &MODULUS); return a; }
static inline fe fe_sub(fe a, const fe b) { if (!fe_geq(&a, &b)) fe_add_raw(&a, &MODULUS); fe_sub_raw(&a, &b); return a; }
static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_is_one_raw(const fe* a) { return a->l[0] == 1 && (a->l[1] | a->l[2] | a->l[3]) == 0; }

/* Montgomery product a*b*2^-256 mod p (CIOS) */
static inline fe fe_mul(const fe a, const fe b) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u64 c = 0;
    for (int j = 0; j < 4; ++j) { u128 s = (u128)a.l[j] * b.l[i] + t[j] + c; t[j] = (u64)s; c = (u64)(s >> 64); }
    u128 s = (u128)t[4] + c; t[4] = (u64)s; t[5] = (u64)(s >> 64);
    u64 m = t[0] * INV64;
    u128 r = (u128)m * MODULUS.l[0] + t[0]; c = (u64)(r >> 64);
    for (int j = 1; j < 4; ++j) { r = (u128)m * MODULUS.l[j] + t[j] + c; t[j - 1] = (u64)r; c = (u64)(r >> 64); }
    r = (u128)t[4] + c; t[3] = (u64)r; t[4] = t[5] + (u64)(r >> 64);
  }
  fe out = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || fe_geq(&out, &MODULUS)) fe_sub_raw(&out, &MODULUS);
  return out;
}
static inline fe fe_from_u64(u64 x) { fe v = {{x, 0, 0, 0}}; return fe_mul(v, R2); } /* F::from(u64) */
static inline fe fe_to_mont(const fe canon) { return fe_mul(canon, R2); }

static inline void raw_div2(fe* a) {
  for (int i = 0; i < 3; ++i) a->l[i] = (a->l[i] >> 1) | (a->l[i + 1] << 63);
  a->l[3] >>= 1;
}
/* ark-ff 0.3 `Field::inverse` for Fp256: binary extended Euclid on the Montgomery residue,
 * started from b = R^2 so the result is again a Montgomery residue. Returns 0 for input 0. */
static int fe_inverse(const fe a, fe* out) {
  if (fe_is_zero(&a)) return 0;
  fe u = a, v = MODULUS, b = R2, c = {{0, 0, 0, 0}};
  while (!fe_is_one_raw(&u) && !fe_is_one_raw(&v)) {
    while (!(u.l[0] & 1)) { raw_div2(&u); if (b.l[0] & 1) fe_add_raw(&b, &MODULUS); raw_div2(&b); }
    while (!(v.l[0] & 1)) { raw_div2(&v); if (c.l[0] & 1) fe_add_raw(&c, &MODULUS); raw_div2(&c); }
    if (!fe_geq(&u, &v)) { fe_sub_raw(&v, &u); c = fe_sub(c, b); }
    else                 { fe_sub_raw(&u, &v); b = fe_sub(b, c); }
  }
  *out = fe_is_one_raw(&u) ? b : c;
  return 1;
}
static fe fe_pow_u64(fe base, u64 e) {
  fe acc = R_ONE;
  while (e) { if (e & 1) acc = fe_mul(acc, base); base = fe_mul(base, base); e >>= 1; }
  return acc;
}

/* exported single-element helpers (used by the tests to pin this file to plonk_oracle.py) */
void orc_fe_mul(const u64* a, const u64* b, u64* out) { fe x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); fe r = fe_mul(x, y); memcpy(out, &r, 32); }
void orc_fe_add(const u64* a, const u64* b, u64* out) { fe x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); fe r = fe_add(x, y); memcpy(out, &r, 32); }
void orc_fe_sub(const u64* a, const u64* b, u64* out) { fe x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); fe r = fe_sub(x, y); memcpy(out, &r, 32); }
int  orc_fe_inverse(const u64* a, u64* out) { fe x, r; memcpy(&x, a, 32); if (!fe_inverse(x, &r)) return 0; memcpy(out, &r, 32); return 1; }
void orc_fe_from_u64(u64 x, u64* out) { fe r = fe_from_u64(x); memcpy(out, &r, 32); }
/* canonical <-> Montgomery, n elements */
void orc_to_mont(const u64* in, size_t n, u64* out) { for (size_t i = 0; i < n; ++i) { fe x; memcpy(&x, in + 4 * i, 32); x = fe_to_mont(x); memcpy(out + 4 * i, &x, 32); } }
void orc_from_mont(const u64* in, size_t n, u64* out) { fe one = {{1, 0, 0, 0}}; for (size_t i = 0; i < n; ++i) { fe x; memcpy(&x, in + 4 * i, 32); x = fe_mul(x, one); memcpy(out + 4 * i, &x, 32); } }

/* ------------------------------------------------------------------ transforms */

static int log2_exact(size_t n) { if (!n || (n & (n - 1))) return -1; int l = 0; while (((size_t)1 << l) < n) ++l; return l; }

/* `FftField::get_root_of_unity(n)`: square the two-adic root down to order n */
static fe root_of_unity(int log_n) {
  fe w = fe_to_mont(ROOT_CANON);
  for (int i = 0; i < TWO_ADICITY - log_n; ++i) w = fe_mul(w, w);
  return w;
}

/* in-place radix-2 transform of a[0..n) with primitive root w, natural order in and out */
static int fft_in_place(fe* a, size_t n, int log_n, fe w) {
  for (size_t i = 1, j = 0; i < n; ++i) { /* bit reversal */
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j |= bit;
    if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  size_t half_n = n / 2 ? n / 2 : 1;
  fe* tw = (fe*)malloc(half_n * sizeof(fe));
  if (!tw) return ORC_ERR_NOMEM;
  tw[0] = R_ONE;
  for (size_t i = 1; i < half_n; ++i) tw[i] = fe_mul(tw[i - 1], w);
  for (int s = 1; s <= log_n; ++s) {
    size_t len = (size_t)1 << s, half = len >> 1, step = n >> s;
    long nb = (long)(n >> 1);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < nb; ++b) {
      size_t k = (size_t)b & (half - 1), start = ((size_t)b >> (s - 1)) << s;
      fe u = a[start + k], v = fe_mul(a[start + k + half], tw[k * step]);
      a[start + k] = fe_add(u, v);
      a[start + k + half] = fe_sub(u, v);
    }
  }
  free(tw);
  return ORC_OK;
}

/* `evaluate_over_domain_by_ref` (src/execution_trace.rs:362-370, src/permutation_polynomial.rs:323-334) */
int orc_ntt(const u64* coeffs, size_t n_coeffs, size_t N, u64* out) {
  int log_n = log2_exact(N);
  if (log_n < 0) return ORC_ERR_NOT_POW2;
  if (log_n > TWO_ADICITY) return ORC_ERR_TWO_ADICITY;
  if (n_coeffs > N) return ORC_ERR_LENGTH;
  memcpy(out, coeffs, n_coeffs * 32);
  memset(out + 4 * n_coeffs, 0, (N - n_coeffs) * 32);
  return fft_in_place((fe*)out, N, log_n, root_of_unity(log_n));
}

/* `evaluations_to_dense_polynomial` (src/polynomial_utils.rs:18-33): length assert, then domain.ifft */
int orc_intt(const u64* evals, size_t evals_len, size_t domain_size, u64* out) {
  int log_n = log2_exact(domain_size);
  if (log_n < 0) return ORC_ERR_NOT_POW2;
  if (log_n > TWO_ADICITY) return ORC_ERR_TWO_ADICITY;
  if (evals_len != domain_size) return ORC_ERR_LENGTH; /* assert_eq! :22-26 */
  fe w = root_of_unity(log_n), w_inv;
  fe_inverse(w, &w_inv);
  memcpy(out, evals, domain_size * 32);
  int rc = fft_in_place((fe*)out, domain_size, log_n, w_inv);
  if (rc) return rc;
  fe n_inv; fe_inverse(fe_from_u64((u64)domain_size), &n_inv);
  fe* o = (fe*)out;
  long n = (long)domain_size;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) o[i] = fe_mul(o[i], n_inv);
  return ORC_OK;
}

/* `gate_evaluation_table_to_gate_polynomial` (src/execution_trace.rs:318-344); sel[k] in
 * ASTKind order (src/ast.rs:56-66), NULL = selector absent */
int orc_gate_polys(const u64* const wires[4], const u64* const sel[9], size_t n, size_t domain_size,
                   u64* const wires_out[4], u64* const sel_out[9]) {
  for (int k = 0; k < 4; ++k) { int rc = orc_intt(wires[k], n, domain_size, wires_out[k]); if (rc) return rc; }
  for (int k = 0; k < 9; ++k) if (sel[k]) { int rc = orc_intt(sel[k], n, domain_size, sel_out[k]); if (rc) return rc; }
  return ORC_OK;
}

/* loop body of src/execution_trace.rs:379-428 on already evaluated columns */
static void gate_constraint_loop(const fe* const w[4], const fe* const s[9], size_t N, fe* out) {
  for (size_t i = 0; i < N; ++i) {
    fe a = w[0][i], b = w[1][i], c = w[2][i], o = w[3][i], acc = {{0, 0, 0, 0}};
    for (int k = 0; k < 9; ++k) {
      if (!s[k]) continue;
      fe sv = s[k][i], e;
      if (fe_is_zero(&sv)) continue;             /* :389-391 */
      switch (k) {
        case 0: case 1: continue;                /* Int, Bool: F::zero() :396-397 */
        case 2: e = fe_sub(o, fe_add(a, b)); break;            /* Add  :400 */
        case 3: e = fe_sub(o, fe_sub(a, b)); break;            /* Sub  :403 */
        case 4: e = fe_sub(o, fe_mul(a, b)); break;            /* Mult :406 */
        case 5: e = fe_sub(fe_mul(o, b), a); break;            /* Div  :409 */
        case 6: e = fe_mul(o, fe_sub(a, b)); break;            /* Eq   :412 */
        case 7: e = fe_sub(fe_add(o, a), R_ONE); break;        /* Not  :415 */
        default: {                                             /* If   :419-423 */
          fe one_minus_a = fe_sub(R_ONE, a);
          e = fe_sub(o, fe_add(fe_mul(a, b), fe_mul(one_minus_a, c)));
        }
      }
      acc = fe_add(acc, fe_mul(sv, e));          /* :426 */
    }
    out[i] = acc;
  }
}

/* `build_gate_constraint_evaluation_polynomial` (src/execution_trace.rs:357-431) */
int orc_gate_constraints(const u64* const wire_coeffs[4], const u64* const sel_coeffs[9], size_t n_coeffs, size_t N, u64* out) {
  fe* w[4] = {0}; fe* s[9] = {0}; int rc = ORC_OK;
  for (int k = 0; k < 4 && !rc; ++k) { w[k] = (fe*)malloc(N * 32); rc = w[k] ? orc_ntt(wire_coeffs[k], n_coeffs, N, (u64*)w[k]) : ORC_ERR_NOMEM; }
  for (int k = 0; k < 9 && !rc; ++k) if (sel_coeffs[k]) { s[k] = (fe*)malloc(N * 32); rc = s[k] ? orc_ntt(sel_coeffs[k], n_coeffs, N, (u64*)s[k]) : ORC_ERR_NOMEM; }
  if (!rc) gate_constraint_loop((const fe* const*)w, (const fe* const*)s, N, (fe*)out);
  for (int k = 0; k < 4; ++k) free(w[k]);
  for (int k = 0; k < 9; ++k) free(s[k]);
  return rc;
}

/* `build_grand_product_evaluation_polynomial` (src/permutation_polynomial.rs:217-261).
 * z_out holds n+1 elements. *bad_row receives the offending row on error. */
int orc_grand_product(const u64* const wires[4], const u64* const sigma[4], size_t n,
                      const u64* beta_, const u64* gamma_, u64* z_out, int64_t* bad_row) {
  fe beta, gamma; memcpy(&beta, beta_, 32); memcpy(&gamma, gamma_, 32);
  fe* z = (fe*)z_out;
  const fe* v[4] = {(const fe*)wires[0], (const fe*)wires[1], (const fe*)wires[2], (const fe*)wires[3]};
  z[0] = R_ONE;
  for (size_t j = 0; j < n; ++j) {
    fe num = R_ONE, den = R_ONE;
    fe bj = fe_mul(beta, fe_from_u64((u64)j));                 /* :249 */
    for (int k = 0; k < 4; ++k) {
      u64 sj = sigma[k][j];
      if (sj >= n) { if (bad_row) *bad_row = (int64_t)j; return ORC_ERR_SIGMA_RANGE; } /* slice index panic :235-238 */
      fe vj = v[k][j];
      num = fe_mul(num, fe_add(fe_add(vj, fe_mul(beta, v[k][sj])), gamma));  /* :242-245 */
      den = fe_mul(den, fe_add(fe_add(vj, bj), gamma));                      /* :251-254 */
    }
    fe inv;
    if (!fe_inverse(den, &inv)) { if (bad_row) *bad_row = (int64_t)j; return ORC_ERR_ZERO_DENOMINATOR; } /* unwrap :258 */
    z[j + 1] = fe_mul(fe_mul(z[j], num), inv);
  }
  return ORC_OK;
}

/* sigma index columns -> field columns, `F::from(x as u64)` (src/permutation_polynomial.rs:281) */
void orc_index_to_field(const u64* idx, size_t n, u64* out) {
  fe* o = (fe*)out;
  for (size_t i = 0; i < n; ++i) o[i] = fe_from_u64(idx[i]);
}

/* `build_permutation_polynomials` (src/permutation_polynomial.rs:272-298).  The reference passes
 * the n+1 grand-product values to the length-asserting interpolation (:296) and so panics whenever
 * n == domain_size; here z[0..n) is interpolated and z[n] is returned in z_last. */
int orc_perm_polys(const u64* const wires[4], const u64* const sigma[4], size_t n, size_t domain_size,
                   const u64* beta, const u64* gamma, u64* const sigma_coeffs_out[4], u64* z_coeffs_out,
                   u64* z_last, int64_t* bad_row) {
  if (n != domain_size) return ORC_ERR_LENGTH;
  u64* tmp = (u64*)malloc((n + 1) * 32);
  if (!tmp) return ORC_ERR_NOMEM;
  int rc = ORC_OK;
  for (int k = 0; k < 4 && !rc; ++k) { orc_index_to_field(sigma[k], n, tmp); rc = orc_intt(tmp, n, domain_size, sigma_coeffs_out[k]); }
  if (!rc) rc = orc_grand_product(wires, sigma, n, beta, gamma, tmp, bad_row);
  if (!rc) { if (z_last) memcpy(z_last, tmp + 4 * n, 32); rc = orc_intt(tmp, n, domain_size, z_coeffs_out); }
  free(tmp);
  return rc;
}

/* loop body of src/permutation_polynomial.rs:341-373 on already evaluated columns */
static void perm_constraint_loop(const fe* z, const fe* const s[4], const fe* const w[4], size_t N,
                                 fe beta, fe gamma, size_t z_shift, fe* out) {
  for (size_t i = 0; i < N; ++i) {
    fe z_next = z[(i + z_shift) % N], z_cur = z[i];           /* :347-348 */
    fe bi = fe_mul(beta, fe_from_u64((u64)i));                /* :351 */
    fe lhs = R_ONE, rhs = R_ONE;
    for (int k = 0; k < 4; ++k) {
      lhs = fe_mul(lhs, fe_add(fe_add(w[k][i], bi), gamma));                       /* :354-358 */
      rhs = fe_mul(rhs, fe_add(fe_add(w[k][i], fe_mul(beta, s[k][i])), gamma));    /* :365-369 */
    }
    out[i] = fe_sub(fe_mul(z_next, lhs), fe_mul(z_cur, rhs));  /* :372 */
  }
}

/* `build_permutation_constraint_evaluation_polynomial` (src/permutation_polynomial.rs:312-376) */
int orc_perm_constraints(const u64* z_coeffs, const u64* const sigma_coeffs[4], const u64* const wire_coeffs[4],
                         size_t n_coeffs, size_t N, const u64* beta_, const u64* gamma_, size_t z_shift, u64* out) {
  fe beta, gamma; memcpy(&beta, beta_, 32); memcpy(&gamma, gamma_, 32);
  fe *z = 0, *s[4] = {0}, *w[4] = {0}; int rc = ORC_OK;
  z = (fe*)malloc(N * 32); rc = z ? orc_ntt(z_coeffs, n_coeffs, N, (u64*)z) : ORC_ERR_NOMEM;
  for (int k = 0; k < 4 && !rc; ++k) { s[k] = (fe*)malloc(N * 32); rc = s[k] ? orc_ntt(sigma_coeffs[k], n_coeffs, N, (u64*)s[k]) : ORC_ERR_NOMEM; }
  for (int k = 0; k < 4 && !rc; ++k) { w[k] = (fe*)malloc(N * 32); rc = w[k] ? orc_ntt(wire_coeffs[k], n_coeffs, N, (u64*)w[k]) : ORC_ERR_NOMEM; }
  if (!rc) perm_constraint_loop(z, (const fe* const*)s, (const fe* const*)w, N, beta, gamma, z_shift, (fe*)out);
  free(z); for (int k = 0; k < 4; ++k) { free(s[k]); free(w[k]); }
  return rc;
}

/* Rounds 2+3 composed the way the reference's signatures imply: row 3 (gate polys), row 6
 * (permutation polys incl. grand product), row 4 (gate constraints), row 7 (permutation
 * constraints; the wires are evaluated a second time there, src/permutation_polynomial.rs:331-334).
 * Outputs: z (n+1), gate constraint vector (N), permutation constraint vector (N). */
int orc_round23(const u64* const wires[4], const u64* const sel[9], const u64* const sigma[4], size_t n, size_t N,
                const u64* beta, const u64* gamma, size_t z_shift,
                u64* z_out, u64* gate_out, u64* perm_out, int64_t* bad_row) {
  u64 *wc[4] = {0}, *sc[9] = {0}, *gc[4] = {0}, *zc = 0, *ztmp = 0; int rc = ORC_OK;
  for (int k = 0; k < 4; ++k) { wc[k] = (u64*)malloc(n * 32); gc[k] = (u64*)malloc(n * 32); if (!wc[k] || !gc[k]) rc = ORC_ERR_NOMEM; }
  for (int k = 0; k < 9; ++k) if (sel[k]) { sc[k] = (u64*)malloc(n * 32); if (!sc[k]) rc = ORC_ERR_NOMEM; }
  zc = (u64*)malloc(n * 32); ztmp = (u64*)malloc((n + 1) * 32); if (!zc || !ztmp) rc = ORC_ERR_NOMEM;
  if (!rc) rc = orc_gate_polys(wires, sel, n, n, wc, sc);
  if (!rc) rc = orc_grand_product(wires, sigma, n, beta, gamma, z_out, bad_row);
  for (int k = 0; k < 4 && !rc; ++k) { orc_index_to_field(sigma[k], n, ztmp); rc = orc_intt(ztmp, n, n, gc[k]); }
  if (!rc) rc = orc_intt(z_out, n, n, zc);
  if (!rc) rc = orc_gate_constraints((const u64* const*)wc, (const u64* const*)sc, n, N, gate_out);
  if (!rc) rc = orc_perm_constraints(zc, (const u64* const*)gc, (const u64* const*)wc, n, N, beta, gamma, z_shift, perm_out);
  for (int k = 0; k < 4; ++k) { free(wc[k]); free(gc[k]); }
  for (int k = 0; k < 9; ++k) free(sc[k]);
  free(zc); free(ztmp);
  return rc;
}

/* ---- the permutation (sigma) table from flattened node ids ----
 * build_permutation_trace + build_permutation_table, src/permutation_polynomial.rs:18-39, :79-202:
 * positions of one node id, in ascending (row, wire) order (collect_wire_positions,
 * src/execution_trace.rs:109-187), form a cycle; sigma[w][j] = row of the next position
 * (OffsetTable::to_index keeps the row only, src/offset_table.rs:61-64); unused wires (node id
 * UINT64_MAX) and single positions map to their own row. */
typedef struct { u64 node; u64 pos; } sig_pair;
static int sig_cmp(const void* a, const void* b) {
  const sig_pair *x = (const sig_pair*)a, *y = (const sig_pair*)b;
  if (x->node != y->node) return x->node < y->node ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
int orc_sigma_table(const u64* const node_ids[4], size_t n, u64* const sigma_out[4]) {
  sig_pair* v = (sig_pair*)malloc((4 * n + 1) * sizeof(sig_pair));
  if (!v) return -8;
  size_t m = 0;
  for (size_t j = 0; j < n; ++j)
    for (int w = 0; w < 4; ++w) {
      sigma_out[w][j] = j;
      if (node_ids[w][j] != UINT64_MAX) { v[m].node = node_ids[w][j]; v[m].pos = 4 * j + w; ++m; }
    }
  qsort(v, m, sizeof(sig_pair), sig_cmp);
  for (size_t i = 0, first = 0; i < m; ++i) {
    if (i > 0 && v[i].node != v[i - 1].node) first = i;
    const size_t next = (i + 1 < m && v[i + 1].node == v[i].node) ? i + 1 : first;
    sigma_out[v[i].pos & 3][v[i].pos >> 2] = v[next].pos >> 2;
  }
  free(v);
  return 0;
}

/* ---- gate evaluation table from the flattened trace ----
 * build_gate_evaluation_table, src/execution_trace.rs:248-302, completed for all three table maps:
 * row_ops[j] = ASTKind index | inputs << 4; wires_out[w][j] = node_values[node_ids[w][j]] when wire w
 * is an input the operation has or the output and the id is below n_values, else default_val
 * (:260-265, :286-289); sel_out[k][j] = one if kind == k else zero (:292-294). */
int orc_gate_table_fill(const uint8_t* row_ops, const u64* const node_ids[4], size_t n, const u64* node_values, size_t n_values,
                        const u64* default_val, u64* const wires_out[4], u64* const sel_out[9]) {
  const fe one = fe_to_mont((fe){{1, 0, 0, 0}}), zero = {{0, 0, 0, 0}};
  for (size_t j = 0; j < n; ++j) {
    const unsigned kind = row_ops[j] & 15u, inputs = row_ops[j] >> 4;
    for (int w = 0; w < 4; ++w) {
      const u64 id = node_ids[w][j];
      const int wired = (w == 3 || (unsigned)w < inputs) && id < n_values;
      memcpy(wires_out[w] + 4 * j, wired ? node_values + 4 * id : default_val, 32);
    }
    for (unsigned k = 0; k < 9; ++k)
      if (sel_out[k]) memcpy(sel_out[k] + 4 * j, kind == k ? &one : &zero, 32);
  }
  return 0;
}

/* ---- coset transforms and quotient completion ----
 * ark-poly 0.3 `coset_fft` / `coset_ifft`: coefficient j times shift^j, then the plain transform (and
 * back); the quotient is the textbook division by Z_H(x) = x^n - 1 on the coset shift * <omega_N>.
 * The reference has no code for this step (README.md:15-17). */
static void scale_by_powers(fe* v, size_t len, fe base) {
  /* serial powers per chunk start, parallel over chunks */
  long chunks = (long)((len + 4095) / 4096);
#pragma omp parallel for schedule(static)
  for (long c = 0; c < chunks; ++c) {
    size_t lo = (size_t)c * 4096, hi = lo + 4096 < len ? lo + 4096 : len;
    fe g = fe_pow_u64(base, (u64)lo);
    for (size_t j = lo; j < hi; ++j) { v[j] = fe_mul(v[j], g); g = fe_mul(g, base); }
  }
}
int orc_coset_ntt(const u64* coeffs, size_t n_coeffs, size_t N, const u64* shift_, u64* out) {
  int log_n = log2_exact(N);
  if (log_n < 0) return ORC_ERR_NOT_POW2;
  if (log_n > TWO_ADICITY) return ORC_ERR_TWO_ADICITY;
  if (n_coeffs > N) return ORC_ERR_LENGTH;
  fe shift; memcpy(&shift, shift_, 32);
  memcpy(out, coeffs, n_coeffs * 32);
  memset(out + 4 * n_coeffs, 0, (N - n_coeffs) * 32);
  scale_by_powers((fe*)out, n_coeffs, shift);
  return fft_in_place((fe*)out, N, log_n, root_of_unity(log_n));
}
int orc_coset_intt(const u64* evals, size_t N, const u64* shift_, u64* out) {
  int rc = orc_intt(evals, N, N, out);
  if (rc) return rc;
  fe shift, inv; memcpy(&shift, shift_, 32);
  if (!fe_inverse(shift, &inv)) return ORC_ERR_ZERO_DENOMINATOR;
  scale_by_powers((fe*)out, N, inv);
  return ORC_OK;
}
int orc_quotient(const u64* const nums[], size_t count, const u64* alpha_, size_t n, size_t N, const u64* shift_, u64* out) {
  int ln = log2_exact(n), lN = log2_exact(N);
  if (ln < 0 || lN < 0) return ORC_ERR_NOT_POW2;
  if (lN > TWO_ADICITY) return ORC_ERR_TWO_ADICITY;
  if (n > N) return ORC_ERR_LENGTH;
  fe alpha, shift; memcpy(&alpha, alpha_, 32); memcpy(&shift, shift_, 32);
  size_t ratio = N / n;
  fe* zinv = (fe*)malloc(ratio * sizeof(fe));
  fe* acc = (fe*)malloc(N * sizeof(fe));
  if (!zinv || !acc) { free(zinv); free(acc); return -8; }
  fe gn = fe_pow_u64(shift, (u64)n), w = root_of_unity(lN - ln), wi = R_ONE;
  for (size_t i = 0; i < ratio; ++i) {
    fe zh = fe_sub(fe_mul(gn, wi), R_ONE);
    if (!fe_inverse(zh, &zinv[i])) { free(zinv); free(acc); return ORC_ERR_ZERO_DENOMINATOR; }
    wi = fe_mul(wi, w);
  }
  long total = (long)N;
#pragma omp parallel for schedule(static)
  for (long j = 0; j < total; ++j) {
    fe a = {{0, 0, 0, 0}};
    for (size_t i = count; i-- > 0;) { fe v; memcpy(&v, nums[i] + 4 * j, 32); a = fe_add(fe_mul(a, alpha), v); }
    acc[j] = fe_mul(a, zinv[(size_t)j % ratio]);
  }
  int rc = orc_coset_intt((const u64*)acc, N, shift_, out);
  free(zinv); free(acc);
  return rc;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int t) {
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}
/* w^e for the order-n root, Montgomery limbs (test helper) */
void orc_root_pow(size_t n, u64 e, u64* out) { fe r = fe_pow_u64(root_of_unity(log2_exact(n)), e); memcpy(out, &r, 32); }
