// plonk_hip.hpp — C++ host-side mirror of the reference's polynomial interface on top of the C ABI
// (plonk_hip.h).  Header-only, C++17; link with -lplonk_hip.
//
// The functions carry the names, argument order and failure behaviour of the Rust functions of
// wu-s-john/PLONK they stand for (cited per declaration, paths relative to the reference tree):
// a Rust panic becomes a C++ exception.  Containers are the closest standard ones:
//   Vec<F>                      -> std::vector<Fr>
//   HashMap<ASTKind, Vec<F>>    -> std::map<ASTKind, std::vector<Fr>>
//   DensePolynomial<F>          -> DensePolynomial (coefficients, trailing zeros dropped)
//   GeneralEvaluationDomain<F>  -> GeneralEvaluationDomain (only the size matters)
//   PermutationTable<usize>     -> PermutationTable (std::vector<uint64_t> columns)
// Fr is the memory of ark-ff 0.3 `Fp256`: four little-endian uint64_t Montgomery limbs.
#ifndef PLONK_HIP_HPP
#define PLONK_HIP_HPP

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "plonk_hip.h"

namespace plonk_hip {

using Fr = std::array<uint64_t, 4>;

// src/ast.rs:56-66; the numeric value is the selector index of the C ABI
enum class ASTKind : int { Int = 0, Bool, Add, Sub, Mult, Div, Eq, Not, If };

// ---- failures -------------------------------------------------------------------------------

struct Error : std::runtime_error {
  int code;
  long long row;
  Error(int c, const std::string& what, long long r = -1) : std::runtime_error(what), code(c), row(r) {}
};
// assert_eq! on lengths — src/polynomial_utils.rs:22-26, :56-60
struct LengthMismatch : Error { using Error::Error; };
// `denominator.inverse().unwrap()` on zero — src/permutation_polynomial.rs:258
struct ZeroDenominator : Error { using Error::Error; };
// slice index panic — src/permutation_polynomial.rs:235-238
struct SigmaOutOfRange : Error { using Error::Error; };

// ---- the context: one per GPU, created on first use ------------------------------------------

class Context {
 public:
  // `field`: PLK_FIELD_BLS12_381_FR (the default) or PLK_FIELD_TWEEDLE_FR; every call on the context computes in it
  explicit Context(int device = 0, int field = PLK_FIELD_BLS12_381_FR) {
    const int rc = plk_ctx_create_field(device, field, &ctx_);
    if (rc != PLK_OK) throw Error(rc, std::string("plk_ctx_create_field: ") + plk_strerror(rc));
  }
  int field() const { return plk_ctx_field(ctx_); }
  ~Context() { plk_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  plk_ctx* get() const { return ctx_; }
  static Context& global() {
    static Context c(0);
    return c;
  }
  void check(int rc) const {
    if (rc == PLK_OK) return;
    std::string msg = plk_last_error(ctx_);
    if (msg.empty()) msg = plk_strerror(rc);
    const long long row = plk_last_error_row(ctx_);
    switch (rc) {
      case PLK_ERR_LENGTH: throw LengthMismatch(rc, msg, row);
      case PLK_ERR_ZERO_DENOMINATOR: throw ZeroDenominator(rc, msg, row);
      case PLK_ERR_SIGMA_RANGE: throw SigmaOutOfRange(rc, msg, row);
      default: throw Error(rc, msg, row);
    }
  }

 private:
  plk_ctx* ctx_ = nullptr;
};

// ---- the reference's types -------------------------------------------------------------------

// What GeneralEvaluationDomain::new(size) yields for this field: a radix-2 subgroup.
struct GeneralEvaluationDomain {
  size_t size_;
  explicit GeneralEvaluationDomain(size_t size) : size_(size) {}
  size_t size() const { return size_; }
};

// ark-poly's DensePolynomial: `from_coefficients_vec` drops trailing zero coefficients
// (src/polynomial_utils.rs:32).
struct DensePolynomial {
  std::vector<Fr> coeffs;
  static DensePolynomial from_coefficients_vec(std::vector<Fr> v) {
    while (!v.empty() && v.back() == Fr{0, 0, 0, 0}) v.pop_back();
    return DensePolynomial{std::move(v)};
  }
  // -1 for the zero polynomial
  long long degree() const { return (long long)coeffs.size() - 1; }
};

// src/execution_trace.rs:204-214
struct GateEvaluationTable {
  std::vector<Fr> input1, input2, input3, output;
  std::map<ASTKind, std::vector<Fr>> selectors;
};

// src/execution_trace.rs:307-314
struct GatePolynomials {
  DensePolynomial input1, input2, input3, output;
  std::map<ASTKind, DensePolynomial> selectors;
};

// src/permutation_polynomial.rs:51-61 with F = usize
struct PermutationTable {
  std::vector<uint64_t> sigma1, sigma2, sigma3, sigma_output;
};

// src/permutation_polynomial.rs:263-269.  z interpolates z[0..n); z_last = z[n] (the reference hands
// all n + 1 values to a length-n assert and panics, see build_permutation_polynomials below).
struct PermutationPolynomials {
  DensePolynomial sigma1, sigma2, sigma3, sigma_output, z;
  std::optional<Fr> z_last;
};

namespace detail {
inline const uint64_t* ptr(const std::vector<Fr>& v) { return v.empty() ? nullptr : v[0].data(); }
inline uint64_t* ptr(std::vector<Fr>& v) { return v.empty() ? nullptr : v[0].data(); }
// coefficient vector padded with zeros to `len` (the C ABI takes equal-length vectors)
inline std::vector<Fr> padded(const DensePolynomial& p, size_t len) {
  std::vector<Fr> v = p.coeffs;
  v.resize(len, Fr{0, 0, 0, 0});
  return v;
}
inline size_t max_len(std::initializer_list<const DensePolynomial*> ps, size_t floor = 1) {
  size_t m = floor;
  for (const DensePolynomial* p : ps) m = std::max(m, p->coeffs.size());
  return m;
}
}  // namespace detail

// ---- the seven functions -----------------------------------------------------------------------

// src/polynomial_utils.rs:18-33
inline DensePolynomial evaluations_to_dense_polynomial(const std::vector<Fr>& evals, const GeneralEvaluationDomain& domain,
                                                       Context& ctx = Context::global()) {
  std::vector<Fr> coeffs(domain.size());
  ctx.check(plk_intt(ctx.get(), detail::ptr(evals), evals.size(), domain.size(), detail::ptr(coeffs)));
  return DensePolynomial::from_coefficients_vec(std::move(coeffs));
}

// src/polynomial_utils.rs:47-62
inline DensePolynomial build_and_convert(const GeneralEvaluationDomain& domain, const std::function<std::vector<Fr>()>& builder,
                                         Context& ctx = Context::global()) {
  const std::vector<Fr> evals = builder();
  if (evals.size() != domain.size()) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));   // :56-60
  return evaluations_to_dense_polynomial(evals, domain, ctx);
}

// src/execution_trace.rs:318-344
inline GatePolynomials gate_evaluation_table_to_gate_polynomial(const GateEvaluationTable& table,
                                                                const GeneralEvaluationDomain& coset_domain,
                                                                Context& ctx = Context::global()) {
  const size_t d = coset_domain.size();
  const std::vector<Fr>* w[4] = {&table.input1, &table.input2, &table.input3, &table.output};
  const uint64_t* wires[4];
  const uint64_t* sel[PLK_NUM_SELECTORS] = {nullptr};
  std::vector<Fr> w_out[4], s_out[PLK_NUM_SELECTORS];
  uint64_t* wires_out[4];
  uint64_t* sel_out[PLK_NUM_SELECTORS] = {nullptr};
  for (int k = 0; k < 4; ++k) {
    if (w[k]->size() != d) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));   // per column, src/polynomial_utils.rs:22-26
    wires[k] = detail::ptr(*w[k]);
    w_out[k].resize(d);
    wires_out[k] = detail::ptr(w_out[k]);
  }
  for (const auto& kv : table.selectors) {
    if (kv.second.size() != d) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    const int k = (int)kv.first;
    sel[k] = detail::ptr(kv.second);
    s_out[k].resize(d);
    sel_out[k] = detail::ptr(s_out[k]);
  }
  ctx.check(plk_gate_polys(ctx.get(), wires, sel, d, d, wires_out, sel_out));
  GatePolynomials g;
  g.input1 = DensePolynomial::from_coefficients_vec(std::move(w_out[0]));
  g.input2 = DensePolynomial::from_coefficients_vec(std::move(w_out[1]));
  g.input3 = DensePolynomial::from_coefficients_vec(std::move(w_out[2]));
  g.output = DensePolynomial::from_coefficients_vec(std::move(w_out[3]));
  for (const auto& kv : table.selectors) g.selectors[kv.first] = DensePolynomial::from_coefficients_vec(std::move(s_out[(int)kv.first]));
  return g;
}

// src/execution_trace.rs:357-431
inline std::vector<Fr> build_gate_constraint_evaluation_polynomial(const GatePolynomials& gate_poly, const GeneralEvaluationDomain& domain,
                                                                   Context& ctx = Context::global()) {
  size_t len = detail::max_len({&gate_poly.input1, &gate_poly.input2, &gate_poly.input3, &gate_poly.output});
  for (const auto& kv : gate_poly.selectors) len = std::max(len, kv.second.coeffs.size());
  const std::vector<Fr> w[4] = {detail::padded(gate_poly.input1, len), detail::padded(gate_poly.input2, len),
                                detail::padded(gate_poly.input3, len), detail::padded(gate_poly.output, len)};
  std::vector<Fr> s[PLK_NUM_SELECTORS];
  const uint64_t* wires[4];
  const uint64_t* sel[PLK_NUM_SELECTORS] = {nullptr};
  for (int k = 0; k < 4; ++k) wires[k] = detail::ptr(w[k]);
  for (const auto& kv : gate_poly.selectors) {
    s[(int)kv.first] = detail::padded(kv.second, len);
    sel[(int)kv.first] = detail::ptr(s[(int)kv.first]);
  }
  std::vector<Fr> out(domain.size());
  ctx.check(plk_gate_constraints(ctx.get(), wires, sel, len, domain.size(), detail::ptr(out)));
  return out;
}

// src/permutation_polynomial.rs:217-261 — n + 1 values
inline std::vector<Fr> build_grand_product_evaluation_polynomial(const PermutationTable& permutation_table,
                                                                 const GateEvaluationTable& evaluation_table, const Fr& beta,
                                                                 const Fr& gamma, Context& ctx = Context::global()) {
  const size_t n = evaluation_table.input1.size();   // :223
  const std::vector<Fr>* w[4] = {&evaluation_table.input1, &evaluation_table.input2, &evaluation_table.input3, &evaluation_table.output};
  const std::vector<uint64_t>* sg[4] = {&permutation_table.sigma1, &permutation_table.sigma2, &permutation_table.sigma3,
                                        &permutation_table.sigma_output};
  const uint64_t *wires[4], *sigma[4];
  for (int k = 0; k < 4; ++k) {
    if (w[k]->size() < n || sg[k]->size() < n) throw SigmaOutOfRange(PLK_ERR_SIGMA_RANGE, "column shorter than input1");   // slice index panic
    wires[k] = detail::ptr(*w[k]);
    sigma[k] = sg[k]->data();
  }
  std::vector<Fr> z(n + 1);
  ctx.check(plk_grand_product(ctx.get(), wires, sigma, n, beta.data(), gamma.data(), detail::ptr(z)));
  return z;
}

// src/permutation_polynomial.rs:272-298.  The reference interpolates the n + 1 grand-product values on a
// size-n domain and panics on its own length assert; here z[0..n) is interpolated and z[n] returned apart.
inline PermutationPolynomials build_permutation_polynomials(const PermutationTable& permutation_table,
                                                            const GateEvaluationTable& evaluation_table, const Fr& beta, const Fr& gamma,
                                                            const GeneralEvaluationDomain& coset_domain,
                                                            Context& ctx = Context::global()) {
  const size_t n = evaluation_table.input1.size(), d = coset_domain.size();
  const std::vector<Fr>* w[4] = {&evaluation_table.input1, &evaluation_table.input2, &evaluation_table.input3, &evaluation_table.output};
  const std::vector<uint64_t>* sg[4] = {&permutation_table.sigma1, &permutation_table.sigma2, &permutation_table.sigma3,
                                        &permutation_table.sigma_output};
  const uint64_t *wires[4], *sigma[4];
  std::vector<Fr> s_out[4], z_out(d);
  uint64_t* sigma_out[4];
  for (int k = 0; k < 4; ++k) {
    if (sg[k]->size() != d || w[k]->size() != n) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    wires[k] = detail::ptr(*w[k]);
    sigma[k] = sg[k]->data();
    s_out[k].resize(d);
    sigma_out[k] = detail::ptr(s_out[k]);
  }
  Fr z_last{};
  ctx.check(plk_perm_polys(ctx.get(), wires, sigma, n, d, beta.data(), gamma.data(), sigma_out, detail::ptr(z_out), z_last.data()));
  PermutationPolynomials p;
  p.sigma1 = DensePolynomial::from_coefficients_vec(std::move(s_out[0]));
  p.sigma2 = DensePolynomial::from_coefficients_vec(std::move(s_out[1]));
  p.sigma3 = DensePolynomial::from_coefficients_vec(std::move(s_out[2]));
  p.sigma_output = DensePolynomial::from_coefficients_vec(std::move(s_out[3]));
  p.z = DensePolynomial::from_coefficients_vec(std::move(z_out));
  p.z_last = z_last;
  return p;
}

// src/permutation_polynomial.rs:312-376; z_shift = 1 is the reference (`z_evals[(i + 1) % n]`, :347)
inline std::vector<Fr> build_permutation_constraint_evaluation_polynomial(const PermutationPolynomials& polynomials,
                                                                          const GatePolynomials& gate_polynomials, const Fr& beta,
                                                                          const Fr& gamma, const GeneralEvaluationDomain& evaluation_domain,
                                                                          size_t z_shift = 1, Context& ctx = Context::global()) {
  const size_t len = detail::max_len({&polynomials.z, &polynomials.sigma1, &polynomials.sigma2, &polynomials.sigma3,
                                      &polynomials.sigma_output, &gate_polynomials.input1, &gate_polynomials.input2,
                                      &gate_polynomials.input3, &gate_polynomials.output});
  const std::vector<Fr> z = detail::padded(polynomials.z, len);
  const std::vector<Fr> s[4] = {detail::padded(polynomials.sigma1, len), detail::padded(polynomials.sigma2, len),
                                detail::padded(polynomials.sigma3, len), detail::padded(polynomials.sigma_output, len)};
  const std::vector<Fr> w[4] = {detail::padded(gate_polynomials.input1, len), detail::padded(gate_polynomials.input2, len),
                                detail::padded(gate_polynomials.input3, len), detail::padded(gate_polynomials.output, len)};
  const uint64_t *sc[4], *wc[4];
  for (int k = 0; k < 4; ++k) { sc[k] = detail::ptr(s[k]); wc[k] = detail::ptr(w[k]); }
  std::vector<Fr> out(evaluation_domain.size());
  ctx.check(plk_perm_constraints(ctx.get(), detail::ptr(z), sc, wc, len, evaluation_domain.size(), beta.data(), gamma.data(), z_shift,
                                 detail::ptr(out)));
  return out;
}

// ---- the fused call and the two table-building steps ---------------------------------------------

struct Round23 {
  std::vector<Fr> z, gate_constraints, permutation_constraints;   // n + 1, N, N values
};

// gate polys (:318) -> grand product (:217) -> sigma / Z interpolation (:272) -> gate constraints (:357) ->
// permutation constraints (:312) in one device-resident call
inline Round23 prove_round23(const GateEvaluationTable& table, const PermutationTable& perm, const Fr& beta, const Fr& gamma,
                             const GeneralEvaluationDomain& n_domain, const GeneralEvaluationDomain& evaluation_domain, size_t z_shift = 1,
                             Context& ctx = Context::global()) {
  const size_t n = n_domain.size(), N = evaluation_domain.size();
  const std::vector<Fr>* w[4] = {&table.input1, &table.input2, &table.input3, &table.output};
  const std::vector<uint64_t>* sg[4] = {&perm.sigma1, &perm.sigma2, &perm.sigma3, &perm.sigma_output};
  const uint64_t *wires[4], *sigma[4];
  const uint64_t* sel[PLK_NUM_SELECTORS] = {nullptr};
  for (int k = 0; k < 4; ++k) {
    if (w[k]->size() != n || sg[k]->size() != n) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    wires[k] = detail::ptr(*w[k]);
    sigma[k] = sg[k]->data();
  }
  for (const auto& kv : table.selectors) {
    if (kv.second.size() != n) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    sel[(int)kv.first] = detail::ptr(kv.second);
  }
  Round23 r;
  r.z.resize(n + 1);
  r.gate_constraints.resize(N);
  r.permutation_constraints.resize(N);
  ctx.check(plk_round23(ctx.get(), wires, sel, sigma, n, N, beta.data(), gamma.data(), z_shift, detail::ptr(r.z),
                        detail::ptr(r.gate_constraints), detail::ptr(r.permutation_constraints)));
  return r;
}

// ---- the step after the path ------------------------------------------------------------------------

// F::multiplicative_generator() of BLS12-381 Fr in Montgomery form (7 * R mod p): the shift ark-poly 0.3's
// coset_fft / coset_ifft use
inline Fr multiplicative_generator() {
  uint64_t modulus[4], one[4], r2[4];
  plk_field_constants(modulus, one, r2);
  // 7 * one mod p by repeated addition (one < p < 2^255, so sums stay below 2^256 before each reduction)
  auto add = [&](const Fr& a, const Fr& b) {
    Fr s; unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a[i] + b[i]; s[i] = (uint64_t)c; c >>= 64; }
    bool ge = true;
    for (int i = 3; i >= 0; --i) if (s[i] != modulus[i]) { ge = s[i] > modulus[i]; break; }
    if (ge) { unsigned __int128 bo = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)s[i] - modulus[i] - bo; s[i] = (uint64_t)d; bo = (d >> 64) & 1; } }
    return s;
  };
  const Fr o{one[0], one[1], one[2], one[3]};
  Fr g = o;
  for (int i = 1; i < 7; ++i) g = add(g, o);
  return g;
}

// ark-poly EvaluationDomain::coset_fft: evaluations of the polynomial at shift * omega^i
inline std::vector<Fr> coset_fft(const std::vector<Fr>& coeffs, const GeneralEvaluationDomain& domain, const Fr& shift = multiplicative_generator(),
                                 Context& ctx = Context::global()) {
  std::vector<Fr> out(domain.size());
  ctx.check(plk_coset_ntt(ctx.get(), detail::ptr(coeffs), coeffs.size(), domain.size(), shift.data(), 0, detail::ptr(out)));
  return out;
}

// ark-poly EvaluationDomain::coset_ifft
inline std::vector<Fr> coset_ifft(const std::vector<Fr>& evals, const GeneralEvaluationDomain& domain, const Fr& shift = multiplicative_generator(),
                                  Context& ctx = Context::global()) {
  std::vector<Fr> out(domain.size());
  ctx.check(plk_coset_ntt(ctx.get(), detail::ptr(evals), evals.size(), domain.size(), shift.data(), 1, detail::ptr(out)));
  return out;
}

// (sum_i alpha^i numerators[i]) / (x^n - 1) from evaluation vectors on the coset shift * <omega_N>; the reference
// stops before this step (README.md:15-17)
inline DensePolynomial quotient_polynomial(const std::vector<std::vector<Fr>>& numerators, const Fr& alpha,
                                           const GeneralEvaluationDomain& n_domain, const GeneralEvaluationDomain& evaluation_domain,
                                           const Fr& shift = multiplicative_generator(), Context& ctx = Context::global()) {
  std::vector<const uint64_t*> p;
  for (const auto& v : numerators) {
    if (v.size() != evaluation_domain.size()) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    p.push_back(detail::ptr(v));
  }
  std::vector<Fr> out(evaluation_domain.size());
  ctx.check(plk_quotient(ctx.get(), p.data(), p.size(), alpha.data(), n_domain.size(), evaluation_domain.size(), shift.data(), detail::ptr(out)));
  return DensePolynomial::from_coefficients_vec(std::move(out));
}

// ---- commitments.  ark-ec `GroupAffine<g1::Parameters>` of BLS12-381 as its memory: x then y, 6 Montgomery limbs each
// (`Fp384`), all zero for the point at infinity.  The reference imports ark_poly_commit::kzg10 (src/main.rs:9) and has no
// commitment code (README.md:15-17); `KZG10::commit` is multi_scalar_mul(powers_of_g, coefficients).
using G1Affine = std::array<uint64_t, 12>;

// the bases stay on the GPU with the context (the powers of tau are fixed per circuit size)
inline void msm_set_bases(const std::vector<G1Affine>& bases, Context& ctx = Context::global()) {
  ctx.check(plk_msm_set_bases(ctx.get(), bases.empty() ? nullptr : bases[0].data(), bases.size()));
}

// ark-ec VariableBaseMSM::multi_scalar_mul(bases, scalars) over the first scalars.size() bases set above
inline G1Affine multi_scalar_mul(const std::vector<Fr>& scalars, Context& ctx = Context::global()) {
  G1Affine out;
  ctx.check(plk_msm_g1(ctx.get(), detail::ptr(scalars), scalars.size(), out.data()));
  return out;
}

// the commitments of several polynomials of one length at once (one prover round), overlapped on the GPU
inline std::vector<G1Affine> multi_scalar_mul_batch(const std::vector<std::vector<Fr>>& scalar_columns, Context& ctx = Context::global()) {
  std::vector<G1Affine> out(scalar_columns.size());
  if (scalar_columns.empty()) return out;
  std::vector<const uint64_t*> p;
  for (const auto& v : scalar_columns) {
    if (v.size() != scalar_columns[0].size()) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    p.push_back(detail::ptr(v));
  }
  ctx.check(plk_msm_g1_batch(ctx.get(), p.data(), p.size(), scalar_columns[0].size(), out[0].data()));
  return out;
}

// KZG10::commit without hiding: the commitment of a polynomial under the bases set above
inline G1Affine commit(const DensePolynomial& polynomial, Context& ctx = Context::global()) { return multi_scalar_mul(polynomial.coeffs, ctx); }

// The execution trace flattened for the two table-building steps: global rows numbered as
// OffsetTable::build numbers them (src/offset_table.rs:22-48), one node id per wire (PLK_NO_WIRE where the
// operation lacks the wire), PLK_ROW_OP(kind, inputs) per row, and the evaluated value of every node.
struct FlatTrace {
  std::vector<uint8_t> row_ops;
  std::vector<uint64_t> node_ids[4];
  std::vector<Fr> node_values;
};

// build_permutation_trace + build_permutation_table — src/permutation_polynomial.rs:18-39, :79-202
inline PermutationTable build_permutation_table(const FlatTrace& trace, Context& ctx = Context::global()) {
  const size_t n = trace.node_ids[0].size();
  PermutationTable t;
  std::vector<uint64_t>* out[4] = {&t.sigma1, &t.sigma2, &t.sigma3, &t.sigma_output};
  const uint64_t* ids[4];
  uint64_t* sig[4];
  for (int k = 0; k < 4; ++k) {
    if (trace.node_ids[k].size() != n) throw LengthMismatch(PLK_ERR_LENGTH, "node id columns differ in length");
    ids[k] = trace.node_ids[k].data();
    out[k]->resize(n);
    sig[k] = out[k]->data();
  }
  ctx.check(plk_sigma_table(ctx.get(), ids, n, sig));
  return t;
}

// build_gate_evaluation_table — src/execution_trace.rs:248-302 (completed: see plk_gate_table_fill)
inline GateEvaluationTable build_gate_evaluation_table(const FlatTrace& trace, const Fr& default_val, Context& ctx = Context::global()) {
  const size_t n = trace.row_ops.size();
  GateEvaluationTable t;
  std::vector<Fr>* w[4] = {&t.input1, &t.input2, &t.input3, &t.output};
  const uint64_t* ids[4];
  uint64_t* wires[4];
  uint64_t* sel[PLK_NUM_SELECTORS] = {nullptr};
  for (int k = 0; k < 4; ++k) {
    if (trace.node_ids[k].size() != n) throw LengthMismatch(PLK_ERR_LENGTH, "node id columns differ in length");
    ids[k] = trace.node_ids[k].data();
    w[k]->resize(n);
    wires[k] = detail::ptr(*w[k]);
  }
  for (uint8_t op : trace.row_ops) {            // selector columns exist for the kinds that occur (:228-235)
    const ASTKind kind = (ASTKind)(op & 15);
    if (!t.selectors.count(kind)) t.selectors[kind].resize(n);
  }
  for (auto& kv : t.selectors) sel[(int)kv.first] = detail::ptr(kv.second);
  ctx.check(plk_gate_table_fill(ctx.get(), trace.row_ops.data(), ids, n, detail::ptr(trace.node_values), trace.node_values.size(),
                                default_val.data(), wires, sel));
  return t;
}

// ---- several GPUs behind the same interface (plk_node_*: no counterpart in the reference) -----------------------

// G ranks in this process, one GPU each.  `exchange`: PLK_NODE_RCCL (one GPU per rank) or PLK_NODE_COPY
// (device-to-device copies; several ranks may share a GPU).
class Node {
 public:
  Node(const std::vector<int>& devices, int exchange, int field = PLK_FIELD_BLS12_381_FR) {
    const int rc = plk_node_create_field(devices.data(), devices.size(), exchange, field, &node_);
    if (rc != PLK_OK) throw Error(rc, std::string("plk_node_create_field: ") + plk_strerror(rc));
  }
  ~Node() { plk_node_destroy(node_); }
  Node(const Node&) = delete;
  Node& operator=(const Node&) = delete;
  plk_node* get() const { return node_; }
  size_t ranks() const { return plk_node_ranks(node_); }
  void check(int rc) const {
    if (rc == PLK_OK) return;
    std::string msg = plk_node_last_error(node_);
    throw Error(rc, msg.empty() ? std::string(plk_strerror(rc)) : msg);
  }

 private:
  plk_node* node_ = nullptr;
};

// evaluations_to_dense_polynomial (src/polynomial_utils.rs:18-33) as ONE transform sharded over the node's GPUs
inline DensePolynomial evaluations_to_dense_polynomial(const std::vector<Fr>& evals, const GeneralEvaluationDomain& domain, Node& node) {
  if (evals.size() != domain.size()) throw LengthMismatch(PLK_ERR_LENGTH, "Evaluation length must match the domain size");
  std::vector<Fr> coeffs(domain.size());
  node.check(plk_node_ntt_host(node.get(), detail::ptr(evals), domain.size(), 1, detail::ptr(coeffs)));
  return DensePolynomial::from_coefficients_vec(std::move(coeffs));
}

// DensePolynomial::evaluate_over_domain_by_ref on the node (coefficients zero-padded to the domain size on the host)
inline std::vector<Fr> evaluate_over_domain(const DensePolynomial& p, const GeneralEvaluationDomain& domain, Node& node) {
  if (p.coeffs.size() > domain.size()) throw LengthMismatch(PLK_ERR_LENGTH, "polynomial degree exceeds the evaluation domain");
  std::vector<Fr> evals(domain.size());
  const std::vector<Fr> c = detail::padded(p, domain.size());
  node.check(plk_node_ntt_host(node.get(), detail::ptr(c), domain.size(), 0, detail::ptr(evals)));
  return evals;
}

// prove_round23 with the ONE circuit's columns dealt to the node's ranks (plk_node_round23_host): host vectors in and
// out.  Every rank's GPU receives the columns, rank g computes the points [g N / G, (g + 1) N / G) of both vectors.
inline Round23 prove_round23(const GateEvaluationTable& table, const PermutationTable& perm, const Fr& beta, const Fr& gamma,
                             const GeneralEvaluationDomain& n_domain, const GeneralEvaluationDomain& evaluation_domain, size_t z_shift, Node& node) {
  const size_t n = n_domain.size(), N = evaluation_domain.size();
  const std::vector<Fr>* w[4] = {&table.input1, &table.input2, &table.input3, &table.output};
  const std::vector<uint64_t>* sg[4] = {&perm.sigma1, &perm.sigma2, &perm.sigma3, &perm.sigma_output};
  const uint64_t *wires[4], *sigma[4];
  const uint64_t* sel[PLK_NUM_SELECTORS] = {nullptr};
  for (int k = 0; k < 4; ++k) {
    if (w[k]->size() != n || sg[k]->size() != n) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    wires[k] = detail::ptr(*w[k]);
    sigma[k] = sg[k]->data();
  }
  for (const auto& kv : table.selectors) {
    if (kv.second.size() != n) throw LengthMismatch(PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
    sel[(int)kv.first] = detail::ptr(kv.second);
  }
  Round23 r;
  r.z.resize(n + 1);
  r.gate_constraints.resize(N);
  r.permutation_constraints.resize(N);
  node.check(plk_node_round23_host(node.get(), wires, sel, sigma, n, N, beta.data(), gamma.data(), z_shift, detail::ptr(r.z),
                                   detail::ptr(r.gate_constraints), detail::ptr(r.permutation_constraints)));
  return r;
}

}  // namespace plonk_hip

#endif  // PLONK_HIP_HPP
