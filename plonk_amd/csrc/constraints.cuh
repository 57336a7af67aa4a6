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
#include "fr29.cuh"

namespace plk {

constexpr int EW_THREADS = 256;

struct GateCols {
  const uint64_t* w[4];  // input1, input2, input3, output evaluations
  const uint64_t* s[9];  // selector evaluations in ASTKind order (src/ast.rs:56-66); nullptr = absent
};

// The arithmetic of both kernels runs on 29-bit limbs (fr29.cuh): columns are cut into limbs once, sums and
// differences are limb-wise and lazy, products are mont256, and one full reduction at the end gives the canonical
// element the reference's field arithmetic would.  Values stay far below the 32 p that reduction accepts (the
// bounds are beside each line, in units of p; inputs are canonical, below p).
__device__ __forceinline__ F9 f9_load(const uint64_t* col, uint64_t i) { return f9_from_fe(fe_load(col, i)); }

// The sum is regrouped by wire so that it costs seven products instead of the eleven of the term-by-term form
// (field arithmetic is exact, so the value is the same element):
//   G = o (s_Add + s_Sub + s_Not + s_Mult + s_If + b s_Div + (a - b) s_Eq)
//     - a (s_Add + s_Sub - s_Not + s_Div + b (s_Mult + s_If) - c s_If)
//     - b (s_Add - s_Sub) - s_Not - c s_If.
// Int (0) and Bool (1) contribute F::zero() (:396-397).  An absent selector is a zero column.
__device__ __forceinline__ Fe gate_point(const F9& a, const F9& b, const F9& c, const F9& o, const uint64_t* const s[9], uint64_t i) {
  F9 zero;
#pragma unroll
  for (int k = 0; k < 9; ++k) zero.l[k] = 0;
  const F9 s_add = s[2] ? f9_load(s[2], i) : zero, s_sub = s[3] ? f9_load(s[3], i) : zero, s_mul = s[4] ? f9_load(s[4], i) : zero;
  const F9 s_div = s[5] ? f9_load(s[5], i) : zero, s_eq = s[6] ? f9_load(s[6], i) : zero, s_not = s[7] ? f9_load(s[7], i) : zero;
  const F9 s_if = s[8] ? f9_load(s[8], i) : zero;
  const F9 addsub = f9_add(s_add, s_sub);                                           // < 2, limbs < 2 * 2^29
  // coefficient of o:                                                                 value (units of p), limbs (units of 2^29)
  F9 co = f9_add(f9_add(addsub, s_not), f9_add(s_mul, s_if));                       // < 5, < 5
  if (s[5]) co = f9_add(co, mont256(b, s_div));                                     // + 1.46 -> < 6.5, < 6
  if (s[6]) co = f9_add(co, mont256(f9_sub_q2(a, b), s_eq));                        // a - b + 2p < 3: + 2.36 -> < 8.9, < 7
  f9_normalize(co);
  const F9 po = mont256(o, co);                                                     // < 0.4527 * 8.9 + 1 = 5.1
  // coefficient of a: s_Add + s_Sub + s_Div + b (s_Mult + s_If) + 4p - (s_Not + c s_If)
  F9 neg = s_not;                                                                   // what the sum subtracts: starts at s_Not
  F9 ca = f9_add(addsub, s_div);                                                    // < 3, < 3
  if (s[4] || s[8]) ca = f9_add(ca, mont256(f9_add(s_mul, s_if), b));               // + 1.91 -> < 4.91, < 4
  F9 cif = zero;
  if (s[8]) { cif = mont256(c, s_if); neg = f9_add(neg, cif); }                     // c s_If < 1.46; s_Not + c s_If < 2.46, limbs < 2
  ca = f9_sub_q4(ca, neg);                                                          // < 8.91, limbs < 4 + 3 = 7
  f9_normalize(ca);
  const F9 pa = mont256(a, ca);                                                     // < 0.4527 * 8.91 + 1 = 5.1
  // G = po + 16p - (pa + b (s_Add - s_Sub) + s_Not + c s_If)
  F9 sub = f9_add(pa, neg);                                                         // < 5.1 + 2.46 = 7.6, limbs < 3
  if (s[2] || s[3]) sub = f9_add(sub, mont256(f9_sub_q2(s_add, s_sub), b));         // (s_Add - s_Sub + 2p) b < 2.36 -> < 9.9, limbs < 4
  f9_normalize(sub);
  return fe_reduce_f9(f9_sub_k<16>(po, sub));                                       // < 5.1 + 16 = 21.1 < 32
}

__global__ __launch_bounds__(EW_THREADS) void gate_constraint_kernel(GateCols cols, uint64_t N, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= N) return;
  fe_store(out, i, gate_point(f9_load(cols.w[0], i), f9_load(cols.w[1], i), f9_load(cols.w[2], i), f9_load(cols.w[3], i), cols.s, i));
}

// The same sums for a 4n-point domain held coset-major: plane r (r = 0..3) of a column holds its
// values at the points omega_4n^(4k + r), k < n.  Plane 0 is the trace-domain column itself (the
// table the polynomial was interpolated from), planes 1..3 lie `plane` 64-bit words apart (n elements and the
// workspace's skew, PLANE_SKEW_WORDS).
// Thread k evaluates points 4k..4k+3 and writes them as one 128-byte run in natural order.
struct GateColsCoset {
  const uint64_t* w0[4];  const uint64_t* s0[9];   // plane 0
  const uint64_t* w[4];   const uint64_t* s[9];    // planes 1..3 at + (r - 1) * plane words
};

// One thread per point i = 4k + r: lanes 4j..4j+3 share k and differ in the plane, so every plane
// is read in runs of 16 consecutive elements per wave and the output store is fully coalesced.
__global__ __launch_bounds__(EW_THREADS) void gate_constraint_coset_kernel(GateColsCoset cols, uint64_t n, uint64_t plane, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= 4 * n) return;
  const uint64_t k = i >> 2;
  const int r = (int)(i & 3);
  const uint64_t off = r ? (uint64_t)(r - 1) * plane : 0;
  const uint64_t* s[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) s[j] = r ? (cols.s[j] ? cols.s[j] + off : nullptr) : cols.s0[j];
  const F9 a = f9_load(r ? cols.w[0] + off : cols.w0[0], k), b = f9_load(r ? cols.w[1] + off : cols.w0[1], k);
  const F9 c = f9_load(r ? cols.w[2] + off : cols.w0[2], k), o = f9_load(r ? cols.w[3] + off : cols.w0[3], k);
  fe_store(out, i, gate_point(a, b, c, o, s, k));
}

// E_i of one point on 29-bit limbs, as gate_point: lhs = prod (w + beta i + gamma), rhs = prod (w + beta sigma + gamma),
// E = z_next lhs - z_cur rhs.  Bounds in units of p beside each line.
__host__ __device__ __forceinline__ F9 f9_from_u64(uint64_t v) {
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = 0;
  r.l[0] = (uint32_t)v & M29; r.l[1] = (uint32_t)(v >> 29) & M29; r.l[2] = (uint32_t)(v >> 58);
  return r;
}
struct PermPoint {
  F9 bi_g, gamma, lhs, rhs;
  __host__ __device__ __forceinline__ PermPoint(uint64_t index, const Fe& beta_r2, const Fe& g) {
    gamma = f9_from_fe(g);
    bi_g = f9_add(mont256(f9_from_u64(index), f9_from_fe(beta_r2)), gamma);     // beta * F::from(i) + gamma (:351)   < 2.5
  }
  // x * y = beta * sigma(point), whichever of the two carries the conversion factor
  __host__ __device__ __forceinline__ void wire(int j, const F9& w, const F9& x, const F9& y) {
    F9 tl = f9_add(w, bi_g);                                  // < 3.5, limbs < 3 * 2^29
    F9 tr = f9_add(f9_add(w, mont256(x, y)), gamma);          // < 3.5
    if (j == 0) { f9_normalize(tl); f9_normalize(tr); lhs = tl; rhs = tr; }
    else { lhs = mont256(tl, lhs); rhs = mont256(tr, rhs); }   // < 6.5, 11.2, 18.5
  }
  // the same with beta * sigma(point) already at hand (a canonical element): no product for the term
  __host__ __device__ __forceinline__ void wire_scaled(int j, const F9& w, const F9& beta_sigma) {
    F9 tl = f9_add(w, bi_g);                                  // < 3.5, limbs < 3 * 2^29
    F9 tr = f9_add(f9_add(w, beta_sigma), gamma);             // < 3
    if (j == 0) { f9_normalize(tl); f9_normalize(tr); lhs = tl; rhs = tr; }
    else { lhs = mont256(tl, lhs); rhs = mont256(tr, rhs); }   // < 6.5, 11.2, 18.5
  }
  __host__ __device__ __forceinline__ Fe finish(const F9& z_next, const F9& z_cur) const {
    return fe_reduce_f9(f9_sub_k<16>(mont256(lhs, z_next), mont256(rhs, z_cur)));   // each < 9.4; difference + 16 p < 26
  }
};

// Every sigma plane holds beta * sigma_k at its points (the columns were transformed times beta; plane 0 is
// beta * F::from(sigma_k[j]), written by the interpolation's first pass), so a wire's right-hand term costs no product.
struct PermColsCoset {
  const uint64_t* z0;  const uint64_t* sig0[4];  const uint64_t* w0[4];   // plane 0: z[0..n), beta * sigma, wire table
  const uint64_t* z;   const uint64_t* sig[4];   const uint64_t* w[4];    // planes 1..3
};

__global__ __launch_bounds__(EW_THREADS) void perm_constraint_coset_kernel(PermColsCoset cols, uint64_t n, uint64_t plane, Fe gamma, Fe beta_r2,
                                                                           uint64_t z_shift, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  const uint64_t N = 4 * n;
  if (i >= N) return;
  const uint64_t k = i >> 2;
  const int r = (int)(i & 3);
  uint64_t inext = i + z_shift;
  if (inext >= N) inext -= N;
  const uint64_t rn = inext & 3, kn = inext >> 2;
  const uint64_t off = r ? (uint64_t)(r - 1) * plane : 0;
  PermPoint pt(i, beta_r2, gamma);
  auto wire = [&](int j) {          // called four times in so many words: the loop form is not unrolled (the products hold asm statements)
    pt.wire_scaled(j, f9_load(r ? cols.w[j] + off : cols.w0[j], k), f9_load(r ? cols.sig[j] + off : cols.sig0[j], k));
  };
  wire(0); wire(1); wire(2); wire(3);
  // (z is read last: fewer values are alive across the products above)
  fe_store(out, i, pt.finish(f9_load(rn ? cols.z + (rn - 1) * plane : cols.z0, kn), f9_load(r ? cols.z + off : cols.z0, k)));
}

// Both constraint vectors in one pass: the wire planes (a quarter of what the two kernels above read between them)
// are loaded once and serve the gate sum and the permutation products of the point.
#ifndef PLK_FUSED_CONSTRAINTS_WAVES
#define PLK_FUSED_CONSTRAINTS_WAVES 1
#endif
__global__ __launch_bounds__(EW_THREADS, PLK_FUSED_CONSTRAINTS_WAVES) void round23_constraints_coset_kernel(
    GateColsCoset gc, PermColsCoset pc, uint64_t n, uint64_t plane, Fe gamma, Fe beta_r2, uint64_t z_shift, uint64_t* gate_out, uint64_t* perm_out) {
  const uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  const uint64_t N = 4 * n;
  if (i >= N) return;
  const uint64_t k = i >> 2;
  const int r = (int)(i & 3);
  const uint64_t off = r ? (uint64_t)(r - 1) * plane : 0;
  const uint64_t* s[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) s[j] = r ? (gc.s[j] ? gc.s[j] + off : nullptr) : gc.s0[j];
  const F9 a = f9_load(r ? gc.w[0] + off : gc.w0[0], k), b = f9_load(r ? gc.w[1] + off : gc.w0[1], k);
  const F9 c = f9_load(r ? gc.w[2] + off : gc.w0[2], k), o = f9_load(r ? gc.w[3] + off : gc.w0[3], k);
  fe_store(gate_out, i, gate_point(a, b, c, o, s, k));
  uint64_t inext = i + z_shift;
  if (inext >= N) inext -= N;
  const uint64_t rn = inext & 3, kn = inext >> 2;
  PermPoint pt(i, beta_r2, gamma);
  pt.wire_scaled(0, a, f9_load(r ? pc.sig[0] + off : pc.sig0[0], k));
  pt.wire_scaled(1, b, f9_load(r ? pc.sig[1] + off : pc.sig0[1], k));
  pt.wire_scaled(2, c, f9_load(r ? pc.sig[2] + off : pc.sig0[2], k));
  pt.wire_scaled(3, o, f9_load(r ? pc.sig[3] + off : pc.sig0[3], k));
  fe_store(perm_out, i, pt.finish(f9_load(rn ? pc.z + (rn - 1) * plane : pc.z0, kn), f9_load(r ? pc.z + off : pc.z0, k)));
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
  PermPoint pt(gi, beta_r2, gamma);
  const F9 beta9 = f9_from_fe(beta);
  pt.wire(0, f9_load(cols.w[0], i), f9_load(cols.s[0], i), beta9);   // :354-369 (in so many words: the loop form is not unrolled)
  pt.wire(1, f9_load(cols.w[1], i), f9_load(cols.s[1], i), beta9);
  pt.wire(2, f9_load(cols.w[2], i), f9_load(cols.s[2], i), beta9);
  pt.wire(3, f9_load(cols.w[3], i), f9_load(cols.s[3], i), beta9);
  fe_store(out, i, pt.finish(f9_load(cols.z, inext), f9_load(cols.z, i)));                                 // :372
}

// Natural-order evaluations on the 4n-point domain from the four coset tables of a column:
// out[4k + r] = plane_r[k].  Plane 0 is the column's own table (field elements, or raw u64 indices
// converted with F::from), planes 1..3 are blocks `plane` 64-bit words apart at `planes`.
struct Interleave4Args {
  const uint64_t* plane0[24];
  uint32_t raw_index[24];      // 1 = plane 0 holds uint64_t indices
  uint64_t* out[24];
  const uint64_t* planes;      // column b: planes + b * 3 * plane words
  uint64_t plane;              // 64-bit words between consecutive planes
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
  if (r) v = fe_load(a.planes + ((uint64_t)b * 3 + (r - 1)) * a.plane, k);
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
// out[i] = v, i < n
__global__ void fe_fill_kernel(uint64_t* out, uint64_t n, Fe v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fe_store(out, i, v);
}

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
