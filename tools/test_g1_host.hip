// Host-side check of g1.cuh (the functions are __host__ __device__): prints k*G for a few k and a chain of
// additions, to be compared with oracle/g1_oracle.py by tests/test_g1_host.py.
#include <cstdio>
#include "../plonk_amd/csrc/g1.cuh"
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
  return 0;
}
