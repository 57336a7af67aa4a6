// Host-side check of fe_inverse_bingcd against the bit-at-a-time Euclid loop and against
// x * x^-1 = 1, on random and special values.  Build: hipcc -O2 -std=c++17 -o /tmp/test_inverse tools/test_inverse.hip
#include <cstdio>
#include <cstdlib>
#include "../plonk_amd/csrc/field.cuh"
using namespace plk;

static uint64_t sm_state = 0x1234567;
static uint64_t splitmix() {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static bool below_p(const Fe& a) {
  for (int i = 7; i >= 0; --i) if (a.l[i] != Field::P[i]) return a.l[i] < Field::P[i];
  return false;
}
static Fe random_fe(int kind) {
  Fe r;
  for (;;) {
    for (int i = 0; i < 4; ++i) { uint64_t w = splitmix(); r.l[2 * i] = (uint32_t)w; r.l[2 * i + 1] = (uint32_t)(w >> 32); }
    if (kind == 1) for (int i = 1 + (int)(splitmix() % 7); i < 8; ++i) r.l[i] = 0;   // short values
    if (kind == 2) { const int sh = (int)(splitmix() % 32); r.l[7] >>= sh; }
    r.l[7] &= 0x7fffffffu;
    if (below_p(r) && !fe_is_zero(r)) return r;
  }
}
int main() {
  long bad = 0, total = 0;
  auto check = [&](const Fe& x) {
    const Fe a = fe_inverse_bingcd(x), b = fe_inverse_eea(x), c = fe_inverse_gcd32(x);
    ++total;
    if (!fe_eq(a, b) || !fe_eq(a, c) || !fe_eq(fe_mul(a, x), fe_one()) || !below_p(a)) {
      if (bad++ < 5) { printf("MISMATCH x="); for (int i = 7; i >= 0; --i) printf("%08x", x.l[i]); printf("\n"); }
    }
  };
  Fe s = fe_one(); check(s);                       // field one
  s = fe_zero(); s.l[0] = 1; check(s);             // residue 1 (= R^-1)
  s.l[0] = 2; check(s);
  for (int i = 0; i < 8; ++i) s.l[i] = Field::P[i];
  s.l[0] -= 1; check(s);                           // p - 1
  s.l[0] -= 1; check(s);
  for (int b = 1; b < 255; ++b) { Fe t = fe_zero(); t.l[b / 32] = 1u << (b % 32); check(t); }   // powers of two
  for (long i = 0; i < 100000; ++i) check(random_fe((int)(i % 3)));
  const Fe z = fe_inverse_bingcd(fe_zero()), z2 = fe_inverse_gcd32(fe_zero());
  if (!fe_is_zero(z) || !fe_is_zero(z2)) { ++bad; printf("inverse of zero is not zero\n"); }
  printf("%ld values, %ld mismatches\n", total, bad);
  return bad != 0;
}
