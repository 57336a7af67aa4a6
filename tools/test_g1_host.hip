// Host-side check of g1.cuh and fq28.cuh (the functions are __host__ __device__): prints k*G for a few k and chains
// of additions, to be compared with oracle/g1_oracle.py by tests/test_g1_oracle.py.
#include <cstdio>
#include "../plonk_amd/csrc/fq28.cuh"
using namespace plk;
static void print_fq_canonical(const Fq& a) {
  Fq one = fq_zero(); one.l[0] = 1;
  const Fq c = fq_mul(a, one);               // out of Montgomery form
  for (int i = 11; i >= 0; --i) std::printf("%08x", c.l[i]);
}
static void print_point(const char* tag, const G1Jac& p) {
  const G1Affine a = g1_to_affine(p);
  std::printf("%s ", tag);
  if (g1_affine_is_inf(a)) { std::printf("inf\n"); return; }
  print_fq_canonical(a.x); std::printf(" "); print_fq_canonical(a.y); std::printf("\n");
}
int main() {
  G1Affine g; for (int i = 0; i < 12; ++i) { g.x.l[i] = FqP::GX[i]; g.y.l[i] = FqP::GY[i]; }
  const G1Jac G = g1_from_affine(g);
  print_point("1", G);
  print_point("2", g1_double(G));
  print_point("3", g1_add_affine(g1_double(G), g));
  print_point("5", g1_add(g1_double(G), g1_add_affine(g1_double(G), g)));
  print_point("dbl_via_add", g1_add(G, G));                         // equal inputs -> doubling branch
  print_point("dbl_via_madd", g1_add_affine(G, g));
  G1Affine ng = g; ng.y = fq_neg(g.y);
  print_point("zero_via_madd", g1_add_affine(G, ng));               // P + (-P)
  print_point("zero_via_add", g1_add(G, g1_from_affine(ng)));
  print_point("inf_plus_g", g1_add_affine(g1_inf(), g));
  print_point("k=0xdeadbeefcafe1234", g1_mul_u64(G, 0xdeadbeefcafe1234ull));
  G1Jac acc = g1_inf();
  for (int i = 1; i <= 100; ++i) acc = g1_add(acc, g1_mul_u64(G, (uint64_t)i * 7919));   // sum of 7919 i = 7919 * 5050
  print_point("sum", acc);

  // the same through the 28-bit-limb working form of the multi-scalar multiplication
  const A28 g28 = a28_from_abi(g);
  const J28 G28 = j28_from_affine(g28);
  print_point("w1", j28_to_abi(G28));
  print_point("w2", j28_to_abi(j28_double(G28)));
  print_point("w3", j28_to_abi(j28_add_affine(j28_double(G28), g28)));
  print_point("w5", j28_to_abi(j28_add(j28_double(G28), j28_add_affine(j28_double(G28), g28))));
  print_point("w_dbl_via_add", j28_to_abi(j28_add(G28, G28)));
  print_point("w_dbl_via_madd", j28_to_abi(j28_add_affine(G28, g28)));
  const A28 ng28 = a28_from_abi(ng);
  print_point("w_zero_via_madd", j28_to_abi(j28_add_affine(G28, ng28)));
  print_point("w_zero_via_add", j28_to_abi(j28_add(G28, j28_from_affine(ng28))));
  print_point("w_inf_plus_g", j28_to_abi(j28_add_affine(j28_inf(), g28)));
  {
    // 4 G + (-4 G) with the left operand a lazy Jacobian point (not Z = 1)
    G1Affine m4 = g1_to_affine(g1_mul_u64(G, 4)); m4.y = fq_neg(m4.y);
    print_point("w_zero_lazy", j28_to_abi(j28_add_affine(j28_double(j28_double(G28)), a28_from_abi(m4))));
  }
  J28 run = j28_inf(), tot = j28_inf();
  for (int i = 1; i <= 200; ++i) {                 // tot = sum_{i <= 200} i G = 20100 G, through mixed and full additions
    run = j28_add_affine(run, g28);
    tot = j28_add(tot, run);
  }
  print_point("w_sum", j28_to_abi(tot));
  J28 d = G28;
  for (int i = 0; i < 300; ++i) d = j28_double(d);   // 2^300 G: long chains keep the lazy values in bounds
  print_point("w_2^300", j28_to_abi(d));
  for (int i = 0; i < 50; ++i) d = j28_add(j28_double(d), j28_add_affine(d, g28));   // d -> 3 d + G, 50 times
  print_point("w_chain", j28_to_abi(d));
  return 0;
}
