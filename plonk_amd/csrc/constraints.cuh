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

// The same sums for a 4n-point domain held coset-major: plane r (r = 0..3) of a column holds its
// values at the points omega_4n^(4k + r), k < n.  Plane 0 is the trace-domain column itself (the
// table the polynomial was interpolated from), planes 1..3 are consecutive n-element blocks.
// Thread k evaluates points 4k..4k+3 and writes them as one 128-byte run in natural order.
struct GateColsCoset {
  const uint64_t* w0[4];  const uint64_t* s0[9];   // plane 0
  const uint64_t* w[4];   const uint64_t* s[9];    // planes 1..3 at + (r - 1) * n elements
};

__device__ __forceinline__ Fe gate_point(const Fe& a, const Fe& b, const Fe& c, const Fe& o, const uint64_t* const s[9], uint64_t i) {
  Fe acc = fe_zero();
  if (s[2]) acc = fe_add(acc, fe_mul(fe_load(s[2], i), fe_sub(o, fe_add(a, b))));          // Add  :400
  if (s[3]) acc = fe_add(acc, fe_mul(fe_load(s[3], i), fe_sub(o, fe_sub(a, b))));          // Sub  :403
  if (s[4] || s[8]) {
    const Fe ab = fe_mul(a, b);
    if (s[4]) acc = fe_add(acc, fe_mul(fe_load(s[4], i), fe_sub(o, ab)));                  // Mult :406
    if (s[8]) {                                                                            // If   :419-423
      const Fe one_minus_a = fe_sub(fe_one(), a);
      acc = fe_add(acc, fe_mul(fe_load(s[8], i), fe_sub(o, fe_add(ab, fe_mul(one_minus_a, c)))));
    }
  }
  if (s[5]) acc = fe_add(acc, fe_mul(fe_load(s[5], i), fe_sub(fe_mul(o, b), a)));          // Div  :409
  if (s[6]) acc = fe_add(acc, fe_mul(fe_load(s[6], i), fe_mul(o, fe_sub(a, b))));          // Eq   :412
  if (s[7]) acc = fe_add(acc, fe_mul(fe_load(s[7], i), fe_sub(fe_add(o, a), fe_one())));   // Not  :415
  return acc;
}

// One thread per point i = 4k + r: lanes 4j..4j+3 share k and differ in the plane, so every plane
// is read in runs of 16 consecutive elements per wave and the output store is fully coalesced.
__global__ __launch_bounds__(EW_THREADS) void gate_constraint_coset_kernel(GateColsCoset cols, uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= 4 * n) return;
  const uint64_t k = i >> 2;
  const int r = (int)(i & 3);
  const uint64_t off = r ? (uint64_t)(r - 1) * n * 4 : 0;
  const uint64_t* s[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) s[j] = r ? (cols.s[j] ? cols.s[j] + off : nullptr) : cols.s0[j];
  const Fe a = fe_load(r ? cols.w[0] + off : cols.w0[0], k), b = fe_load(r ? cols.w[1] + off : cols.w0[1], k);
  const Fe c = fe_load(r ? cols.w[2] + off : cols.w0[2], k), o = fe_load(r ? cols.w[3] + off : cols.w0[3], k);
  fe_store(out, i, gate_point(a, b, c, o, s, k));
}

struct PermColsCoset {
  const uint64_t* z0;  const uint64_t* sig0[4];  const uint64_t* w0[4];   // plane 0: z[0..n), RAW u64 sigma indices, wire table
  const uint64_t* z;   const uint64_t* sig[4];   const uint64_t* w[4];    // planes 1..3
};

__global__ __launch_bounds__(EW_THREADS) void perm_constraint_coset_kernel(PermColsCoset cols, uint64_t n, Fe beta, Fe gamma, Fe beta_r2,
                                                                           uint64_t z_shift, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  const uint64_t N = 4 * n;
  if (i >= N) return;
  const uint64_t k = i >> 2;
  const int r = (int)(i & 3);
  uint64_t inext = i + z_shift;
  if (inext >= N) inext -= N;
  const uint64_t rn = inext & 3, kn = inext >> 2;
  const uint64_t off = r ? (uint64_t)(r - 1) * n * 4 : 0;
  const Fe z_cur = fe_load(r ? cols.z + off : cols.z0, k);
  const Fe z_next = fe_load(rn ? cols.z + (rn - 1) * n * 4 : cols.z0, kn);
  Fe iraw = fe_zero(); iraw.l[0] = (uint32_t)i; iraw.l[1] = (uint32_t)(i >> 32);
  const Fe bi_g = fe_add(fe_mul(iraw, beta_r2), gamma);   // beta * F::from(i) + gamma  (:351)
  Fe lhs, rhs;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const Fe w = fe_load(r ? cols.w[j] + off : cols.w0[j], k);
    // on the trace domain sigma_j(omega^k) IS F::from(sigma_j[k]): the raw index times beta * R^2;
    // elsewhere the evaluated sigma polynomial times beta.  One multiplication either way.
    Fe x, y;
    if (r) { x = fe_load(cols.sig[j] + off, k); y = beta; }
    else { const uint64_t sv = cols.sig0[j][k]; x = fe_zero(); x.l[0] = (uint32_t)sv; x.l[1] = (uint32_t)(sv >> 32); y = beta_r2; }
    const Fe tl = fe_add(w, bi_g);
    const Fe tr = fe_add(fe_add(w, fe_mul(x, y)), gamma);
    lhs = (j == 0) ? tl : fe_mul(lhs, tl);
    rhs = (j == 0) ? tr : fe_mul(rhs, tr);
  }
  fe_store(out, i, fe_sub(fe_mul(z_next, lhs), fe_mul(z_cur, rhs)));
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

// Natural-order evaluations on the 4n-point domain from the four coset tables of a column:
// out[4k + r] = plane_r[k].  Plane 0 is the column's own table (field elements, or raw u64 indices
// converted with F::from), planes 1..3 are consecutive n-element blocks at `planes`.
struct Interleave4Args {
  const uint64_t* plane0[24];
  uint32_t raw_index[24];      // 1 = plane 0 holds uint64_t indices
  uint64_t* out[24];
  const uint64_t* planes;      // column b: planes + b * 3n elements
  // Output layout: point i lands in chunk i / chunk at offset i % chunk; consecutive chunks of one
  // column are chunk_stride elements apart, and each chunk is followed by copies of the first
  // `halo` points of the next chunk (cyclically).  chunk = 4n, halo = 0 is the plain vector.
  uint64_t chunk, chunk_stride, halo;
};

__global__ __launch_bounds__(EW_THREADS) void interleave4_kernel(Interleave4Args a, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= 4 * n) return;
  const int b = blockIdx.y;
  const uint64_t k = i >> 2;
  const int r = (int)(i & 3);
  Fe v;
  if (r) v = fe_load(a.planes + ((uint64_t)b * 3 + (r - 1)) * n * 4, k);
  else v = a.raw_index[b] ? fe_from_u64(a.plane0[b][k]) : fe_load(a.plane0[b], k);
  const uint64_t c = i / a.chunk, j = i - c * a.chunk;
  fe_store(a.out[b], c * a.chunk_stride + j, v);
  if (j < a.halo) {
    const uint64_t chunks = 4 * n / a.chunk, prev = c ? c - 1 : chunks - 1;
    fe_store(a.out[b], prev * a.chunk_stride + a.chunk + j, v);
  }
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
