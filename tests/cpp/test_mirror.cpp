// Parity tests of the C++ host mirror (include/plonk_hip.hpp) on a GPU, written the way tests of the
// reference's own functions would read: build a table, call the function under its reference name,
// compare with the CPU oracle (oracle/plonk_oracle.c, the checker) bit for bit.
// Built by `make -C tests/cpp`; run by tests/test_gpu_cpp_mirror.py.  Exit code = number of failures.
#include <cinttypes>
#include <cstdio>
#include <cstring>

#include "plonk_hip.hpp"

using namespace plonk_hip;
using u64 = uint64_t;

extern "C" {
void orc_to_mont(const u64* in, size_t n, u64* out);
int orc_intt(const u64* evals, size_t evals_len, size_t domain_size, u64* out);
int orc_gate_polys(const u64* const wires[4], const u64* const sel[9], size_t n, size_t domain_size, u64* const wires_out[4],
                   u64* const sel_out[9]);
int orc_gate_constraints(const u64* const wire_coeffs[4], const u64* const sel_coeffs[9], size_t n_coeffs, size_t N, u64* out);
int orc_grand_product(const u64* const wires[4], const u64* const sigma[4], size_t n, const u64* beta, const u64* gamma, u64* z_out,
                      int64_t* bad_row);
int orc_perm_polys(const u64* const wires[4], const u64* const sigma[4], size_t n, size_t domain_size, const u64* beta, const u64* gamma,
                   u64* const sigma_coeffs_out[4], u64* z_coeffs_out, u64* z_last, int64_t* bad_row);
int orc_perm_constraints(const u64* z_coeffs, const u64* const sigma_coeffs[4], const u64* const wire_coeffs[4], size_t n_coeffs, size_t N,
                         const u64* beta, const u64* gamma, size_t z_shift, u64* out);
int orc_round23(const u64* const wires[4], const u64* const sel[9], const u64* const sigma[4], size_t n, size_t N, const u64* beta,
                const u64* gamma, size_t z_shift, u64* z_out, u64* gate_out, u64* perm_out, int64_t* bad_row);
int orc_sigma_table(const u64* const node_ids[4], size_t n, u64* const sigma_out[4]);
int orc_coset_ntt(const u64* coeffs, size_t n_coeffs, size_t N, const u64* shift, u64* out);
int orc_quotient(const u64* const nums[], size_t count, const u64* alpha, size_t n, size_t N, const u64* shift, u64* out);
int orc_g1_msm(const u64* scalars, const u64* bases, size_t n, u64* out_xy);
int orc_gate_table_fill(const uint8_t* row_ops, const u64* const node_ids[4], size_t n, const u64* node_values, size_t n_values,
                        const u64* default_val, u64* const wires_out[4], u64* const sel_out[9]);
}

static int failures = 0;
#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) { ++failures; std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

// ---- seeded inputs (splitmix64; any residue below the modulus is a valid element) -------------------
static u64 sm_state;
static u64 splitmix() {
  u64 z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static Fr random_fr() { return Fr{splitmix(), splitmix(), splitmix(), splitmix() >> 2}; }   // top limb below 2^62 < modulus' top limb
static Fr from_int(u64 x) { Fr c{x, 0, 0, 0}, m; orc_to_mont(c.data(), 1, m.data()); return m; }
static std::vector<Fr> random_column(size_t n) { std::vector<Fr> v(n); for (auto& e : v) e = random_fr(); return v; }

struct Circuit { GateEvaluationTable table; PermutationTable perm; Fr beta, gamma; };
static const ASTKind ALL_KINDS[9] = {ASTKind::Int, ASTKind::Bool, ASTKind::Add, ASTKind::Sub, ASTKind::Mult,
                                     ASTKind::Div, ASTKind::Eq, ASTKind::Not, ASTKind::If};

static Circuit make_circuit(size_t n, u64 seed, int n_kinds = 9) {
  sm_state = seed;
  Circuit c;
  c.table.input1 = random_column(n); c.table.input2 = random_column(n);
  c.table.input3 = random_column(n); c.table.output = random_column(n);
  const Fr one = from_int(1), zero{0, 0, 0, 0};
  for (int k = 0; k < n_kinds; ++k) c.table.selectors[ALL_KINDS[k]].assign(n, zero);
  for (size_t j = 0; j < n; ++j) c.table.selectors[ALL_KINDS[splitmix() % n_kinds]][j] = one;
  std::vector<u64>* sg[4] = {&c.perm.sigma1, &c.perm.sigma2, &c.perm.sigma3, &c.perm.sigma_output};
  for (auto* s : sg) {                       // Fisher-Yates
    s->resize(n);
    for (size_t j = 0; j < n; ++j) (*s)[j] = j;
    for (size_t j = n; j > 1; --j) std::swap((*s)[j - 1], (*s)[splitmix() % j]);
  }
  c.beta = random_fr(); c.gamma = random_fr();
  return c;
}

struct Ptrs {           // the oracle takes the same pointer tables as the C ABI
  const u64* wires[4]; const u64* sel[9]; const u64* sigma[4];
  explicit Ptrs(const Circuit& c) {
    const std::vector<Fr>* w[4] = {&c.table.input1, &c.table.input2, &c.table.input3, &c.table.output};
    const std::vector<u64>* s[4] = {&c.perm.sigma1, &c.perm.sigma2, &c.perm.sigma3, &c.perm.sigma_output};
    for (int k = 0; k < 4; ++k) { wires[k] = (*w[k])[0].data(); sigma[k] = s[k]->data(); }
    for (int k = 0; k < 9; ++k) sel[k] = nullptr;
    for (const auto& kv : c.table.selectors) sel[(int)kv.first] = kv.second[0].data();
  }
};

static std::vector<Fr> padded(const DensePolynomial& p, size_t n) { auto v = p.coeffs; v.resize(n, Fr{0, 0, 0, 0}); return v; }

// ---- tests ------------------------------------------------------------------------------------------

static void test_evaluations_to_dense_polynomial() {
  std::printf("test_evaluations_to_dense_polynomial\n");
  for (size_t n : {size_t(1), size_t(2), size_t(8), size_t(1) << 12}) {
    sm_state = 17 + n;
    const std::vector<Fr> evals = random_column(n);
    const DensePolynomial p = evaluations_to_dense_polynomial(evals, GeneralEvaluationDomain(n));
    std::vector<Fr> want(n);
    CHECK(orc_intt(evals[0].data(), n, n, want[0].data()) == 0);
    CHECK(padded(p, n) == want);
  }
  // a constant table interpolates to a degree-0 polynomial: trailing zeros are dropped (src/polynomial_utils.rs:32)
  const DensePolynomial c = evaluations_to_dense_polynomial(std::vector<Fr>(16, from_int(5)), GeneralEvaluationDomain(16));
  CHECK(c.coeffs.size() == 1 && c.coeffs[0] == from_int(5) && c.degree() == 0);
  // "Evaluation length must match the domain size" (src/polynomial_utils.rs:22-26)
  bool threw = false;
  try { evaluations_to_dense_polynomial(std::vector<Fr>(7, from_int(1)), GeneralEvaluationDomain(8)); } catch (const LengthMismatch&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { build_and_convert(GeneralEvaluationDomain(8), [] { return std::vector<Fr>(9, Fr{1, 0, 0, 0}); }); } catch (const LengthMismatch&) { threw = true; }
  CHECK(threw);
  const DensePolynomial b = build_and_convert(GeneralEvaluationDomain(4), [] { return std::vector<Fr>(4, Fr{0, 0, 0, 0}); });
  CHECK(b.coeffs.empty() && b.degree() == -1);
}

static void test_round_2_and_3_function_by_function(size_t n, size_t N, int n_kinds) {
  std::printf("test_round_2_and_3_function_by_function n=%zu N=%zu kinds=%d\n", n, N, n_kinds);
  const Circuit c = make_circuit(n, 1000 + n + N + n_kinds, n_kinds);
  const Ptrs p(c);
  const GeneralEvaluationDomain dn(n), dN(N);

  const GatePolynomials gp = gate_evaluation_table_to_gate_polynomial(c.table, dn);
  std::vector<Fr> w_ref[4], s_ref[9];
  u64 *w_out[4], *s_out[9];
  for (int k = 0; k < 4; ++k) { w_ref[k].resize(n); w_out[k] = w_ref[k][0].data(); }
  for (int k = 0; k < 9; ++k) { s_out[k] = nullptr; if (p.sel[k]) { s_ref[k].resize(n); s_out[k] = s_ref[k][0].data(); } }
  CHECK(orc_gate_polys(p.wires, p.sel, n, n, w_out, s_out) == 0);
  CHECK(padded(gp.input1, n) == w_ref[0] && padded(gp.input2, n) == w_ref[1] && padded(gp.input3, n) == w_ref[2] && padded(gp.output, n) == w_ref[3]);
  for (const auto& kv : gp.selectors) CHECK(padded(kv.second, n) == s_ref[(int)kv.first]);

  const std::vector<Fr> z = build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma);
  std::vector<Fr> z_ref(n + 1);
  int64_t bad = -1;
  CHECK(orc_grand_product(p.wires, p.sigma, n, c.beta.data(), c.gamma.data(), z_ref[0].data(), &bad) == 0);
  CHECK(z.size() == n + 1 && z == z_ref);

  const PermutationPolynomials pp = build_permutation_polynomials(c.perm, c.table, c.beta, c.gamma, dn);
  std::vector<Fr> sc_ref[4], zc_ref(n);
  u64* sc_out[4];
  Fr z_last_ref;
  for (int k = 0; k < 4; ++k) { sc_ref[k].resize(n); sc_out[k] = sc_ref[k][0].data(); }
  CHECK(orc_perm_polys(p.wires, p.sigma, n, n, c.beta.data(), c.gamma.data(), sc_out, zc_ref[0].data(), z_last_ref.data(), &bad) == 0);
  CHECK(padded(pp.z, n) == zc_ref && pp.z_last && *pp.z_last == z_last_ref && *pp.z_last == z[n]);
  CHECK(padded(pp.sigma1, n) == sc_ref[0] && padded(pp.sigma_output, n) == sc_ref[3]);

  const std::vector<Fr> g = build_gate_constraint_evaluation_polynomial(gp, dN);
  const u64* wc[4] = {w_ref[0][0].data(), w_ref[1][0].data(), w_ref[2][0].data(), w_ref[3][0].data()};
  const u64* scs[9];
  for (int k = 0; k < 9; ++k) scs[k] = p.sel[k] ? s_ref[k][0].data() : nullptr;
  std::vector<Fr> g_ref(N), e_ref(N);
  CHECK(orc_gate_constraints(wc, scs, n, N, g_ref[0].data()) == 0);
  CHECK(g == g_ref);

  const std::vector<Fr> e = build_permutation_constraint_evaluation_polynomial(pp, gp, c.beta, c.gamma, dN);
  const u64* sgc[4] = {sc_ref[0][0].data(), sc_ref[1][0].data(), sc_ref[2][0].data(), sc_ref[3][0].data()};
  CHECK(orc_perm_constraints(zc_ref[0].data(), sgc, wc, n, N, c.beta.data(), c.gamma.data(), 1, e_ref[0].data()) == 0);
  CHECK(e == e_ref);

  const Round23 r = prove_round23(c.table, c.perm, c.beta, c.gamma, dn, dN);
  CHECK(r.z == z_ref && r.gate_constraints == g_ref && r.permutation_constraints == e_ref);
}

static void test_panics_become_exceptions() {
  std::printf("test_panics_become_exceptions\n");
  Circuit c = make_circuit(64, 77);
  // sigma index past the end: slice index panic of src/permutation_polynomial.rs:235-238, reported with its row
  c.perm.sigma3[41] = 64;
  try { build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma); CHECK(!"no exception"); }
  catch (const SigmaOutOfRange& e) { CHECK(e.row == 41); }
  c.perm.sigma3[41] = 0;
  // zero denominator at row 5: v + beta * 5 + gamma = 0 for the first wire (`inverse().unwrap()`, :258).
  // With beta = 0 the term is v + gamma: take gamma = -v by setting v = 0 and gamma = 0.
  c.beta = Fr{0, 0, 0, 0}; c.gamma = Fr{0, 0, 0, 0}; c.table.input1[5] = Fr{0, 0, 0, 0};
  try { build_grand_product_evaluation_polynomial(c.perm, c.table, c.beta, c.gamma); CHECK(!"no exception"); }
  catch (const ZeroDenominator& e) { CHECK(e.row == 5); }
  // the context stays usable after a failure
  const Circuit ok = make_circuit(64, 78);
  CHECK(build_grand_product_evaluation_polynomial(ok.perm, ok.table, ok.beta, ok.gamma).size() == 65);
  // not a power of two
  bool threw = false;
  try { evaluations_to_dense_polynomial(std::vector<Fr>(12, Fr{0, 0, 0, 0}), GeneralEvaluationDomain(12)); } catch (const Error& e) { threw = e.code == PLK_ERR_NOT_POW2; }
  CHECK(threw);
}

static void test_tables_from_a_trace() {
  std::printf("test_tables_from_a_trace\n");
  // Add(Int 40, Int 2) (src/ast.rs:473-477): one double-input row; nodes: 0 = the sum, 1 = 40, 2 = 2
  FlatTrace ex;
  ex.row_ops = {PLK_ROW_OP((int)ASTKind::Add, 2)};
  ex.node_ids[0] = {1}; ex.node_ids[1] = {2}; ex.node_ids[2] = {PLK_NO_WIRE}; ex.node_ids[3] = {0};
  ex.node_values = {from_int(42), from_int(40), from_int(2)};
  const GateEvaluationTable t = build_gate_evaluation_table(ex, from_int(0));
  CHECK(t.input1[0] == from_int(40) && t.input2[0] == from_int(2) && t.input3[0] == from_int(0) && t.output[0] == from_int(42));
  CHECK(t.selectors.size() == 1 && t.selectors.at(ASTKind::Add)[0] == from_int(1));
  const PermutationTable s = build_permutation_table(ex);
  CHECK(s.sigma1[0] == 0 && s.sigma2[0] == 0 && s.sigma3[0] == 0 && s.sigma_output[0] == 0);

  // a seeded trace of 5000 rows: every row's output is a fresh node, inputs reuse earlier nodes
  const size_t n = 5000, leaves = 700;
  sm_state = 4242;
  FlatTrace tr;
  tr.row_ops.resize(n);
  for (auto& col : tr.node_ids) col.assign(n, PLK_NO_WIRE);
  for (size_t j = 0; j < n; ++j) {
    const int inputs = 1 + (int)(splitmix() % 3);
    const int kind = inputs == 1 ? (int)ASTKind::Not : inputs == 3 ? (int)ASTKind::If : 2 + (int)(splitmix() % 5);
    tr.row_ops[j] = PLK_ROW_OP(kind, inputs);
    for (int w = 0; w < inputs; ++w) tr.node_ids[w][j] = splitmix() % (leaves + j);
    tr.node_ids[3][j] = leaves + j;
  }
  tr.node_values = random_column(leaves + n - 100);          // the newest 100 nodes have no value: default applies
  const Fr dv = from_int(9);
  const GateEvaluationTable gt = build_gate_evaluation_table(tr, dv);
  const PermutationTable pt = build_permutation_table(tr);
  const u64* ids[4] = {tr.node_ids[0].data(), tr.node_ids[1].data(), tr.node_ids[2].data(), tr.node_ids[3].data()};
  std::vector<Fr> w_ref[4], s_ref[9];
  std::vector<u64> sg_ref[4];
  u64 *w_out[4], *s_out[9], *sg_out[4];
  for (int k = 0; k < 4; ++k) { w_ref[k].resize(n); w_out[k] = w_ref[k][0].data(); sg_ref[k].resize(n); sg_out[k] = sg_ref[k].data(); }
  for (int k = 0; k < 9; ++k) { s_ref[k].resize(n); s_out[k] = s_ref[k][0].data(); }
  CHECK(orc_gate_table_fill(tr.row_ops.data(), ids, n, tr.node_values[0].data(), tr.node_values.size(), dv.data(), w_out, s_out) == 0);
  CHECK(orc_sigma_table(ids, n, sg_out) == 0);
  CHECK(gt.input1 == w_ref[0] && gt.input2 == w_ref[1] && gt.input3 == w_ref[2] && gt.output == w_ref[3]);
  for (const auto& kv : gt.selectors) CHECK(kv.second == s_ref[(int)kv.first]);
  CHECK(gt.selectors.size() == 7);                            // Add..If occur; Int and Bool have no rows
  CHECK(pt.sigma1 == sg_ref[0] && pt.sigma2 == sg_ref[1] && pt.sigma3 == sg_ref[2] && pt.sigma_output == sg_ref[3]);
}

static void test_quotient_step() {
  std::printf("test_quotient_step\n");
  const size_t n = 256, N = 1024;
  const Fr g = multiplicative_generator();
  CHECK(g == from_int(7));
  sm_state = 606;
  const std::vector<Fr> f = random_column(3 * n);
  const std::vector<Fr> ev = coset_fft(f, GeneralEvaluationDomain(N));
  std::vector<Fr> ev_ref(N);
  CHECK(orc_coset_ntt(f[0].data(), f.size(), N, g.data(), ev_ref[0].data()) == 0);
  CHECK(ev == ev_ref);
  std::vector<Fr> back = coset_ifft(ev, GeneralEvaluationDomain(N));
  back.resize(f.size());
  CHECK(back == f);
  const std::vector<std::vector<Fr>> nums = {random_column(N), random_column(N)};
  const Fr alpha = random_fr();
  const DensePolynomial t = quotient_polynomial(nums, alpha, GeneralEvaluationDomain(n), GeneralEvaluationDomain(N));
  std::vector<Fr> t_ref(N);
  const u64* np[2] = {nums[0][0].data(), nums[1][0].data()};
  CHECK(orc_quotient(np, 2, alpha.data(), n, N, g.data(), t_ref[0].data()) == 0);
  CHECK(padded(t, N) == t_ref);
  bool threw = false;                      // x^n - 1 vanishes on the plain subgroup
  try { quotient_polynomial(nums, alpha, GeneralEvaluationDomain(n), GeneralEvaluationDomain(N), from_int(1)); } catch (const ZeroDenominator&) { threw = true; }
  CHECK(threw);
}

static void test_commitment() {
  std::printf("test_commitment\n");
  // the generator of BLS12-381 G1 in ark-ec's affine memory (Montgomery x, y)
  const G1Affine gen = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
  const size_t n = 300;
  std::vector<G1Affine> bases(n);                    // (i + 1)^2 * G, by the oracle, plus one point at infinity
  for (size_t i = 0; i < n; ++i) {
    const Fr k = from_int((i + 1) * (i + 1));
    CHECK(orc_g1_msm(k.data(), gen.data(), 1, bases[i].data()) == 0);
  }
  bases[17].fill(0);
  msm_set_bases(bases);
  sm_state = 707;
  DensePolynomial f = DensePolynomial::from_coefficients_vec(random_column(n));
  f.coeffs[3] = from_int(0); f.coeffs[4] = from_int(1);
  G1Affine want;
  CHECK(orc_g1_msm(f.coeffs[0].data(), bases[0].data(), n, want.data()) == 0);
  CHECK(commit(f) == want);
  const std::vector<Fr> few(f.coeffs.begin(), f.coeffs.begin() + 5);   // fewer scalars than bases
  CHECK(orc_g1_msm(few[0].data(), bases[0].data(), few.size(), want.data()) == 0);
  CHECK(multi_scalar_mul(few) == want);
  const std::vector<std::vector<Fr>> cols = {f.coeffs, random_column(n), random_column(n)};
  const std::vector<G1Affine> many = multi_scalar_mul_batch(cols);
  for (size_t k = 0; k < cols.size(); ++k) {
    CHECK(orc_g1_msm(cols[k][0].data(), bases[0].data(), n, want.data()) == 0);
    CHECK(many[k] == want);
  }
  bool threw = false;
  try { multi_scalar_mul(random_column(n + 1)); } catch (const Error&) { threw = true; }
  CHECK(threw);
}

// One transform sharded over four ranks of a node (the ranks share GPU 0 here and exchange by device-to-device
// copies), through the reference's own function shape, against the C restatement and the one-GPU call.
static void test_node_transform() {
  std::printf("test_node_transform\n");
  Node node({0, 0, 0, 0}, PLK_NODE_COPY);
  CHECK(node.ranks() == 4);
  for (size_t n : {size_t(1) << 10, size_t(1) << 16}) {
    sm_state = 99 + n;
    const std::vector<Fr> evals = random_column(n);
    const DensePolynomial p = evaluations_to_dense_polynomial(evals, GeneralEvaluationDomain(n), node);
    std::vector<Fr> want(n);
    CHECK(orc_intt(evals[0].data(), n, n, want[0].data()) == 0);
    CHECK(padded(p, n) == want);
    CHECK(padded(evaluations_to_dense_polynomial(evals, GeneralEvaluationDomain(n)), n) == want);
    CHECK(evaluate_over_domain(p, GeneralEvaluationDomain(n), node) == evals);
  }
  bool threw = false;
  try { evaluations_to_dense_polynomial(std::vector<Fr>(24, from_int(1)), GeneralEvaluationDomain(24), node); } catch (const Error& e) { threw = e.code == PLK_ERR_NOT_POW2; }
  CHECK(threw);
  threw = false;
  try { Node bad({0, 0}, PLK_NODE_RCCL); } catch (const Error& e) { threw = e.code == PLK_ERR_ARG; }   // RCCL wants one GPU per rank
  CHECK(threw);
}

// One proof over the four ranks of a node against the one-GPU call and, for Z, the C restatement.
static void test_node_round23() {
  std::printf("test_node_round23\n");
  Node node({0, 0, 0, 0}, PLK_NODE_COPY);
  const size_t n = 1 << 11, N = 4 * n;
  const Circuit c = make_circuit(n, 4242);
  const GeneralEvaluationDomain dn(n), dN(N);
  for (size_t z_shift : {size_t(1), size_t(4)}) {
    const Round23 one = prove_round23(c.table, c.perm, c.beta, c.gamma, dn, dN, z_shift);
    const Round23 many = prove_round23(c.table, c.perm, c.beta, c.gamma, dn, dN, z_shift, node);
    CHECK(many.z == one.z);
    CHECK(many.gate_constraints == one.gate_constraints);
    CHECK(many.permutation_constraints == one.permutation_constraints);
    Ptrs p(c);
    std::vector<Fr> z(n + 1);
    int64_t bad = -1;
    CHECK(orc_grand_product(p.wires, p.sigma, n, c.beta.data(), c.gamma.data(), z[0].data(), &bad) == 0);
    CHECK(many.z == z);
  }
  bool threw = false;
  try { prove_round23(c.table, c.perm, c.beta, c.gamma, dn, GeneralEvaluationDomain(2 * n), 1, node); } catch (const Error& e) { threw = e.code == PLK_ERR_ARG; }
  CHECK(threw);
}

// A context over the second field: the transform of the constant one is (n, 0, 0, ...) in any field, the one of
// (0, 1, 0, ...) is the powers of that field's root of unity - which differ between the fields - and the inverse
// transform returns the input.
static void test_second_field() {
  std::printf("test_second_field\n");
  Context tw(0, PLK_FIELD_TWEEDLE_FR);
  CHECK(tw.field() == PLK_FIELD_TWEEDLE_FR && Context::global().field() == PLK_FIELD_BLS12_381_FR);
  uint64_t one0[4], one1[4], p1[4];
  CHECK(plk_field_constants_of(PLK_FIELD_BLS12_381_FR, nullptr, one0, nullptr) == PLK_OK);
  CHECK(plk_field_constants_of(PLK_FIELD_TWEEDLE_FR, p1, one1, nullptr) == PLK_OK);
  CHECK(p1[3] == 0x4000000000000000ull && p1[0] == 0x842cafd400000001ull);
  const size_t n = 1 << 12;
  const Fr o1{one1[0], one1[1], one1[2], one1[3]}, zero{0, 0, 0, 0};
  std::vector<Fr> delta(n, zero); delta[1] = o1;
  DensePolynomial x; x.coeffs = delta;                       // the polynomial X
  const std::vector<Fr> pw = [&] { std::vector<Fr> e(n); tw.check(plk_ntt(tw.get(), plonk_hip::detail::ptr(delta), n, n, plonk_hip::detail::ptr(e))); return e; }();
  CHECK(pw[0] == o1 && !(pw[1] == o1) && !(pw[1] == zero));
  // omega^(n/2) = -1: pw[n/2] + 1 = 0 mod p, checked through the library's own product: (pw[n/2])^2 = 1
  std::vector<Fr> back(n);
  tw.check(plk_intt(tw.get(), plonk_hip::detail::ptr(pw), n, n, plonk_hip::detail::ptr(back)));
  CHECK(back == delta);
  // the same input on the first field's context gives other points (another modulus, another root)
  std::vector<Fr> d0(n, zero); d0[1] = Fr{one0[0], one0[1], one0[2], one0[3]};
  std::vector<Fr> pw0(n);
  Context::global().check(plk_ntt(Context::global().get(), plonk_hip::detail::ptr(d0), n, n, plonk_hip::detail::ptr(pw0)));
  CHECK(!(pw0[1] == pw[1]));
  // commitments live in BLS12-381 G1 only
  uint64_t xy[12];
  CHECK(plk_msm_g1(tw.get(), plonk_hip::detail::ptr(delta), 4, xy) == PLK_ERR_ARG);
}

int main() {
  std::setvbuf(stdout, nullptr, _IONBF, 0);        // the name of the running test is out before anything it does
  test_evaluations_to_dense_polynomial();
  test_round_2_and_3_function_by_function(4, 16, 9);
  test_round_2_and_3_function_by_function(1 << 10, 1 << 10, 9);     // "check" use: N = n
  test_round_2_and_3_function_by_function(1 << 10, 1 << 12, 9);
  test_round_2_and_3_function_by_function(1 << 13, 1 << 15, 4);     // a subset of the selector kinds
  test_panics_become_exceptions();
  test_tables_from_a_trace();
  test_quotient_step();
  test_commitment();
  test_node_transform();
  test_node_round23();
  test_second_field();
  std::printf("%d failure(s)\n", failures);
  return failures;
}
