// Point-wise constraint evaluation on an evaluation domain of N points.
//
// gate_constraint_kernel: the loop of build_gate_constraint_evaluation_polynomial,
//   src/execution_trace.rs:379-428:  G_i = sum_k s_k(i) * P_k(a_i, b_i, c_i, o_i).
// perm_constraint_kernel: the loop of build_permutation_constraint_evaluation_polynomial,
//   src/permutation_polynomial.rs:341-373:
//   E_i = z[(i + shift) mod N] * prod_k (w_k(i) + beta*i + gamma) - z[i] * prod_k (w_k(i) + beta*sigma_k(i) + gamma).
// One thread per point; every column is read once (plus the neighbouring z).
#pragma once
#include "field.cuh"

namespace plk {

constexpr int EW_THREADS = 256;

struct GateCols {
  const uint64_t* w[4];  // input1, input2, input3, output evaluations
  const uint64_t* s[9];  // selector evaluations in ASTKind order (src/ast.rs:56-66); nullptr = absent
};

__global__ __launch_bounds__(EW_THREADS) void gate_constraint_kernel(GateCols cols, uint64_t N, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= N) return;
  const Fe a = fe_load(cols.w[0], i), b = fe_load(cols.w[1], i), c = fe_load(cols.w[2], i), o = fe_load(cols.w[3], i);
  Fe acc = fe_zero();
  // Int (0) and Bool (1) contribute F::zero() (:396-397)
  if (cols.s[2]) acc = fe_add(acc, fe_mul(fe_load(cols.s[2], i), fe_sub(o, fe_add(a, b))));          // Add  :400
  if (cols.s[3]) acc = fe_add(acc, fe_mul(fe_load(cols.s[3], i), fe_sub(o, fe_sub(a, b))));          // Sub  :403
  if (cols.s[4] || cols.s[8]) {
    const Fe ab = fe_mul(a, b);
    if (cols.s[4]) acc = fe_add(acc, fe_mul(fe_load(cols.s[4], i), fe_sub(o, ab)));                  // Mult :406
    if (cols.s[8]) {                                                                                 // If   :419-423
      const Fe one_minus_a = fe_sub(fe_one(), a);
      acc = fe_add(acc, fe_mul(fe_load(cols.s[8], i), fe_sub(o, fe_add(ab, fe_mul(one_minus_a, c)))));
    }
  }
  if (cols.s[5]) acc = fe_add(acc, fe_mul(fe_load(cols.s[5], i), fe_sub(fe_mul(o, b), a)));          // Div  :409
  if (cols.s[6]) acc = fe_add(acc, fe_mul(fe_load(cols.s[6], i), fe_mul(o, fe_sub(a, b))));          // Eq   :412
  if (cols.s[7]) acc = fe_add(acc, fe_mul(fe_load(cols.s[7], i), fe_sub(fe_add(o, a), fe_one())));   // Not  :415
  fe_store(out, i, acc);
}

struct PermCols {
  const uint64_t* z;
  const uint64_t* s[4];
  const uint64_t* w[4];
};

// `count` points starting at domain index `index0`; z holds z_len elements and the neighbour
// index wraps at z_len (z_len = N for a whole domain; count + z_shift when a halo is appended).
__global__ __launch_bounds__(EW_THREADS) void perm_constraint_kernel(PermCols cols, uint64_t count, uint64_t index0, uint64_t z_len,
                                                                     Fe beta, Fe gamma, Fe beta_r2, uint64_t z_shift, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= count) return;
  uint64_t inext = i + z_shift;            // z_shift < z_len is checked on the host
  if (inext >= z_len) inext -= z_len;
  const uint64_t gi = index0 + i;
  Fe iraw = fe_zero(); iraw.l[0] = (uint32_t)gi; iraw.l[1] = (uint32_t)(gi >> 32);
  const Fe bi_g = fe_add(fe_mul(iraw, beta_r2), gamma);   // beta * F::from(i) + gamma  (:351)
  Fe lhs, rhs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Fe w = fe_load(cols.w[k], i);
    const Fe tl = fe_add(w, bi_g);                                                   // :354-358
    const Fe tr = fe_add(fe_add(w, fe_mul(beta, fe_load(cols.s[k], i))), gamma);    // :365-369
    lhs = (k == 0) ? tl : fe_mul(lhs, tl);
    rhs = (k == 0) ? tr : fe_mul(rhs, tr);
  }
  const Fe e = fe_sub(fe_mul(fe_load(cols.z, inext), lhs), fe_mul(fe_load(cols.z, i), rhs));   // :372
  fe_store(out, i, e);
}

// out[i] = F::from(idx[i])  (src/permutation_polynomial.rs:281)
__global__ __launch_bounds__(EW_THREADS) void index_to_field_kernel(const uint64_t* idx, uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  fe_store(out, i, fe_from_u64(idx[i]));
}

// out[i] = in[i] * factor  (Montgomery <-> canonical conversion helpers: factor = R^2 or 1)
__global__ __launch_bounds__(EW_THREADS) void scale_kernel(const uint64_t* in, uint64_t n, Fe factor, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  fe_store(out, i, fe_mul(fe_load(in, i), factor));
}

// out[i] = a[i] * b[i]
__global__ __launch_bounds__(EW_THREADS) void mul_kernel(const uint64_t* a, const uint64_t* b, uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  fe_store(out, i, fe_mul(fe_load(a, i), fe_load(b, i)));
}

// out[r * cols + c] = base^((row0 + r) * (col0 + c)) * mul : the twiddle block of a four-step
// transform owned by one rank
__global__ __launch_bounds__(EW_THREADS) void twiddle_block_kernel(uint64_t* out, uint64_t rows, uint64_t cols, uint64_t row0,
                                                                  uint64_t col0, Fe base, Fe mul) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= rows * cols) return;
  const uint64_t r = i / cols, c = i % cols;
  fe_store(out, i, fe_mul(fe_pow_u64(base, (row0 + r) * (col0 + c)), mul));
}

}  // namespace plk
