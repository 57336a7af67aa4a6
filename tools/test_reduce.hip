// Host-side check of fe_reduce_f9 (quotient estimate) against fe_reduce_f9_by_subtractions on the values where an
// estimate can go wrong - k p - 1, k p, k p + 1 and neighbours for every k < 32, the largest lazy value - and on
// random lazy limb vectors.  Prints the number of mismatches.
#include <cstdio>
#include "../plonk_amd/csrc/fr29.cuh"
using namespace plk;
static bool same(const Fe& a, const Fe& b) { for (int i = 0; i < 8; ++i) if (a.l[i] != b.l[i]) return false; return true; }
// value -> normalised limbs; `spread` moves part of every limb into the one below to make the vector lazy
static F9 limbs_of(const uint32_t w[9], int spread) {
  F9 r;
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, wi = bit >> 5, o = bit & 31;
    uint64_t v = (uint64_t)w[wi] >> o;
    if (wi + 1 < 9) v |= (uint64_t)w[wi + 1] << (32 - o);
    r.l[i] = i == 8 ? (uint32_t)v : ((uint32_t)v & M29);
  }
  if (spread) for (int i = 8; i >= 1; --i) { const uint32_t take = r.l[i] < 3 ? r.l[i] : 3; r.l[i] -= take; r.l[i - 1] += take << 29; }
  return r;
}
int main() {
  long mismatches = 0, cases = 0;
  for (int k = 0; k < 32; ++k)
    for (int delta = -3; delta <= 3; ++delta)
      for (int spread = 0; spread < 2; ++spread) {
        if (k == 0 && delta < 0) continue;
        uint32_t w[9]; uint64_t c = 0;
        for (int j = 0; j < 8; ++j) { c += (uint64_t)Field::P[j] * (uint32_t)k; w[j] = (uint32_t)c; c >>= 32; }
        w[8] = (uint32_t)c;
        int64_t d = delta;                        // add delta with carry / borrow
        for (int j = 0; j < 9 && d; ++j) { const int64_t t = (int64_t)w[j] + d; w[j] = (uint32_t)t; d = t >> 32; }
        const F9 a = limbs_of(w, spread);
        ++cases;
        if (!same(fe_reduce_f9(a), fe_reduce_f9_by_subtractions(a))) { ++mismatches; std::printf("k %d delta %d spread %d\n", k, delta, spread); }
      }
  uint64_t s = 12345;
  for (long it = 0; it < 2000000; ++it) {
    F9 a;
    for (int i = 0; i < 9; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a.l[i] = (uint32_t)(s >> 33); }   // limbs < 2^31
    a.l[8] %= (31u * 7597479u);                    // keeps the value below 32 p whatever the lower limbs carry up
    ++cases;
    if (!same(fe_reduce_f9(a), fe_reduce_f9_by_subtractions(a))) { if (mismatches < 5) std::printf("random case %ld\n", it); ++mismatches; }
  }
  std::printf("%ld cases, %ld mismatches\n", cases, mismatches);
  return mismatches != 0;
}
