// Device-side check of the 29-bit-limb constraint arithmetic (PermPoint, gate_point) against the same formulas
// written with the 32-bit field functions, stage by stage.  Built by `make -C tests/cpp`; run by tests/test_gpu_cpp_mirror.py.
#include <cstdio>
#include <vector>
#include "../../plonk_amd/csrc/constraints.cuh"
using namespace plk;

__device__ Fe dev_rnd(uint64_t s, int small) {
  Fe a;
  if (small) { a = fe_from_u64(s % 50); return a; }
  for (int i = 0; i < 8; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a.l[i] = (uint32_t)(s >> 32); }
  a.l[7] &= 0x3fffffff;
  return a;
}

__global__ void check_kernel(int stage, int small, Fe beta, Fe gamma, Fe beta_r2, uint64_t* out_want, uint64_t* out_got, uint64_t n) {
  const uint64_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe w[4], sg[4];
  for (int k = 0; k < 4; ++k) { w[k] = dev_rnd(i * 16 + k, small); sg[k] = dev_rnd(i * 16 + 4 + k, small); }
  const Fe z_next = dev_rnd(i * 16 + 8, small), z_cur = dev_rnd(i * 16 + 9, small);
  Fe iraw = fe_zero(); iraw.l[0] = (uint32_t)i;
  const Fe bi_g = fe_add(fe_mul(iraw, beta_r2), gamma);
  Fe lhs, rhs;
  for (int k = 0; k < 4; ++k) {
    const Fe tl = fe_add(w[k], bi_g), tr = fe_add(fe_add(w[k], fe_mul(beta, sg[k])), gamma);
    lhs = k ? fe_mul(lhs, tl) : tl; rhs = k ? fe_mul(rhs, tr) : tr;
  }
  PermPoint pt(i, beta_r2, gamma);
#pragma unroll
  for (int k = 0; k < 4; ++k) pt.wire(k, f9_from_fe(w[k]), f9_from_fe(sg[k]), f9_from_fe(beta));
  if (stage == 0) { fe_store(out_want, i, fe_sub(fe_mul(z_next, lhs), fe_mul(z_cur, rhs))); fe_store(out_got, i, pt.finish(f9_from_fe(z_next), f9_from_fe(z_cur))); }
  if (stage == 1) { fe_store(out_want, i, fe_mul(w[0], sg[0])); fe_store(out_got, i, fe_reduce_f9(mont256(f9_from_fe(w[0]), f9_from_fe(sg[0])))); }
  if (stage == 2) { fe_store(out_want, i, bi_g); fe_store(out_got, i, fe_reduce_f9(pt.bi_g)); }
  if (stage == 3) { fe_store(out_want, i, lhs); fe_store(out_got, i, fe_reduce_f9(pt.lhs)); }
  if (stage == 4) { fe_store(out_want, i, rhs); fe_store(out_got, i, fe_reduce_f9(pt.rhs)); }
  if (stage == 5) { fe_store(out_want, i, fe_mul(z_next, lhs)); fe_store(out_got, i, fe_reduce_f9(mont256(pt.lhs, f9_from_fe(z_next)))); }
  // the transforms' products on operands cut in registers just before (top limbs known to the compiler to fit 24
  // bits, the situation in which mont256 needs its fences): mont29 returns a b 2^-261, i.e. fe_mul(a, b) / 32
  if (stage == 6 || stage == 7) {
    Fe r1, r2;
    if (stage == 6) { r1 = fe_reduce_f9(mont29(f9_from_fe(w[0]), f9_from_fe(sg[0]))); r2 = r1; }
    else { F9 o1, o2; mont29_x2(f9_from_fe(w[0]), f9_from_fe(sg[0]), f9_from_fe(w[1]), f9_from_fe(sg[1]), o1, o2); r1 = fe_reduce_f9(o1); r2 = fe_reduce_f9(o2); }
    for (int d = 0; d < 5; ++d) { r1 = fe_double(r1); r2 = fe_double(r2); }
    fe_store(out_want, i, stage == 6 ? fe_mul(w[0], sg[0]) : fe_add(fe_mul(w[0], sg[0]), fe_mul(w[1], sg[1])));
    fe_store(out_got, i, stage == 6 ? r1 : fe_add(r1, r2));
  }
}

int main() {
  const uint64_t n = 1 << 16;
  uint64_t *want, *got;
  hipMalloc(&want, n * 32); hipMalloc(&got, n * 32);
  Fe R2; for (int i = 0; i < 8; ++i) R2.l[i] = Field::R2[i];
  int bad_total = 0;
  for (int stage = 0; stage < 8; ++stage) for (int small = 0; small < 2; ++small) {
    const Fe beta = small ? fe_from_u64(5) : fe_from_u64(0x123456789abcdefull), gamma = fe_from_u64(7);
    hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, stage, small, beta, gamma, fe_mul(beta, R2), want, got, n);
    std::vector<uint64_t> a(n * 4), b(n * 4);
    hipMemcpy(a.data(), want, n * 32, hipMemcpyDeviceToHost); hipMemcpy(b.data(), got, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (uint64_t i = 0; i < n * 4; ++i) if (a[i] != b[i]) { if (bad < 1) std::printf("small %d mismatch at element %lu\n", small, i / 4); ++bad; }
    std::printf("stage %d small %d: %d mismatching words\n", stage, small, bad);
    bad_total += bad;
  }
  return bad_total != 0;
}
