// Batched number-theoretic transforms over the two-adic subgroup, natural order in and out.
//
// Replaces ark-poly's `domain.ifft` (src/polynomial_utils.rs:29) and
// `evaluate_over_domain_by_ref` (src/execution_trace.rs:362-370,
// src/permutation_polynomial.rs:323-334): same root-of-unity convention
// (omega_N = TWO_ADIC_ROOT^(2^(s - log2 N))), output index i <-> point omega^i, inverse
// includes the 1/N scale, short coefficient vectors are zero-padded to the domain size.
//
// Decomposition.  N = R1 * R2 * R3 (one to three passes, each radix a power of two <= 2^11).
// Input index n = n1 * (N/R1) + n2 * (N/(R1 R2)) + n3, output index k = k1 + R1 k2 + R1 R2 k3.
// Pass p runs the size-Rp transform over digit p inside one workgroup (tile of Rp points x C
// neighbouring columns held in LDS), multiplies by the inter-pass twiddle
// omega_{N'}^(k_p * lower digits) and writes digit k_p back where n_p was; the last pass reads
// whole contiguous rows and scatters them to natural order in runs of C consecutive outputs.
// Every global access of a tile is therefore a run of C * 32 contiguous bytes.
//
// Arithmetic inside a tile is the redundant 9 x 29-bit form of fr29.cuh: rows are written to
// LDS in bit-reversed order and combined by decimation-in-time butterflies
// (T = Y * w; X' = X + T; Y' = X - T + 2p) whose sums are left unreduced; limbs are
// renormalised every second stage and values reduced only when a pass stores its result.
// Between passes elements are 32-byte integers below 2^256 that may exceed p by < p/2.
#pragma once
#include "field.cuh"
#include "fr29.cuh"

namespace plk {

#ifndef PLK_TILE_GROUPING
#define PLK_TILE_GROUPING 1      // workgroup order of the non-final passes: see ntt_pass_kernel
#endif
#ifndef PLK_FINAL_XCD_ORDER
#define PLK_FINAL_XCD_ORDER 0    // measurement knob: XCD-contiguous workgroup order in the last pass
#endif
#ifndef PLK_COSET_MAJOR
#define PLK_COSET_MAJOR 0
#endif
#ifndef PLK_NT_PASS_STORE
#define PLK_NT_PASS_STORE 0      // measurement knob: non-temporal stores of a non-final pass's outputs
#endif
constexpr int TW_WORDS = 12;                // one twiddle record (9 limbs + padding)
constexpr int NTT_TILE_LOG = 11;            // elements per workgroup tile
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;
constexpr int NTT_THREADS = 512;
constexpr int NTT_MAX_BATCH = 24;           // columns per launch
constexpr int NTT_MASTER_LOG = 11;          // per-tile twiddles: omega_{2^11}^e, e < 2^10
constexpr int NTT_LDS_TW_LOG = 7;           // per-tile twiddles kept in LDS when the radix is <= 2^8
constexpr int NTT_LDS_TW_RECORDS = 192;     // the LDS twiddle area (tile_transform says how each radix fills it)
constexpr size_t NTT_LDS_BYTES = ((size_t)NTT_TILE + NTT_LDS_TW_RECORDS) * 36;   // 9 words per element; 2 workgroups per CU
static_assert(2 * ((NTT_LDS_BYTES + 511) / 512 * 512) <= 160 * 1024, "two workgroups of the transform kernels share a CU's LDS");

enum NttInputKind : uint32_t { NTT_IN_FIELD = 0, NTT_IN_U64 = 1 };

struct NttBatch {
  const uint64_t* in[NTT_MAX_BATCH];
  uint64_t* out[NTT_MAX_BATCH];
  uint32_t kind[NTT_MAX_BATCH];   // NttInputKind, honoured by the first pass only
  // strided mode (stride_* != 0): transform b reads in[0] + b * stride_in and writes
  // out[0] + b * stride_out (strides in uint64_t units), kind[0] applies to all
  uint64_t stride_in, stride_out;
  __device__ __forceinline__ const uint64_t* src(int b) const { return stride_in ? in[0] + (uint64_t)b * stride_in : in[b]; }
  __device__ __forceinline__ uint64_t* dst(int b) const { return stride_out ? out[0] + (uint64_t)b * stride_out : out[b]; }
  __device__ __forceinline__ uint32_t kind_of(int b) const { return stride_in ? kind[0] : kind[b]; }
};

// Device-resident constants of one (log N, direction) transform; every table holds twiddle
// records (value * 2^261 mod p as 9 limbs, fr29.cuh).
struct NttTables {
  const uint32_t* master;  // omega_{2^11}^e (or its inverse), e < 2^10
  // the same constants for the precomputed-quotient product (fr29.cuh, shoup29_x2): record 2e = omega_{2^11}^e as a
  // canonical integer, record 2e + 1 = floor(that * 2^261 / p); nullptr = not built
  const uint32_t* master_sh;
  const uint32_t* tw_hi;   // omega_N^(e << lo_bits), e < 2^(logN - lo_bits)
  const uint32_t* tw_hi_scaled;  // same times 1/N (inverse, first inter-pass twiddle); == tw_hi for forward
  const uint32_t* tw_lo;   // omega_N^e, e < 2^lo_bits
  const uint32_t* scale;   // 1/N as one record, or nullptr
  // full inter-pass tables (nullptr = use the two-level pair above): after pass p,
  // record[(k << log_inner) + q] = omega_{N'}^(k q), times 1/N in pass 0 of an inverse transform
  const uint32_t* tw_full[2];
  int lo_bits;
};

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// LDS tile: element e -> limbs 0-3 at A[e], limbs 4-7 at B[e] (16 bytes each), limb 8 at C[e]
struct Tile {
  uint4* A; uint4* B; uint32_t* C;
  __device__ __forceinline__ explicit Tile(void* base, int elems = NTT_TILE)
      : A(reinterpret_cast<uint4*>(base)), B(A + elems), C(reinterpret_cast<uint32_t*>(B + elems)) {}
  __device__ __forceinline__ F9 get(int e) const {
    const uint4 a = A[e], b = B[e];
    F9 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = C[e];
    return r;
  }
  __device__ __forceinline__ void put(int e, const F9& v) const {
    A[e] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    B[e] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    C[e] = v.l[8];
  }
};

// What a launch does with columns of raw 64-bit integers (NTT_IN_U64), passed by value with the kernel arguments.
struct NttU64 {
  // the conversion constant as a twiddle record: 2^(256+261) mod p turns the integer x into the element F::from(x);
  // c * 2^(256+261) mod p turns it into c * F::from(x) (a caller that wants c times the column transformed: the
  // transforms are linear)
  uint32_t rec[TW_WORDS];
  // where the converted inputs go as canonical elements (column b at conv_out + (b - conv_first) * conv_stride
  // uint64_t units), or nullptr
  uint64_t* conv_out;
  uint64_t conv_stride;
  int conv_first;
};

// One input element -> normalised F9 (value < 1.5 p).
__device__ __forceinline__ F9 load_input(const uint64_t* in, uint64_t idx, bool valid, uint32_t kind, const NttU64& u64) {
  F9 v;
  if (!valid) {
#pragma unroll
    for (int i = 0; i < 9; ++i) v.l[i] = 0;
  } else if (kind == NTT_IN_U64) {
    const uint64_t x = in[idx];
#pragma unroll
    for (int i = 0; i < 9; ++i) v.l[i] = 0;
    v.l[0] = (uint32_t)x & M29; v.l[1] = (uint32_t)(x >> 29) & M29; v.l[2] = (uint32_t)(x >> 58);
    F9 c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.l[i] = u64.rec[i];
    v = mont29(v, c);   // F::from(u64), times the caller's constant if there is one
  } else {
    v = f9_from_fe(fe_load(in, idx));
  }
  return v;
}

// Decimation-in-time transform of size 2^log_r along the row axis of a [2^log_r][2^log_c] tile,
// with both ends open: `load(r, c)` yields input row r (natural order) of column c as a normalised
// value, `store(k, c, v)` takes output row k.  Values pass through LDS only between steps:
//   * the first two stages are one radix-4 step on rows r0, r0 + R/2, r0 + R/4, r0 + 3R/4 taken
//     straight from `load` (their only non-trivial twiddle is omega_4: one product per four
//     elements; `quarter`: only rows below R/4 are non-zero - the first pass of an n -> 4n
//     extension - so the four outputs are copies of the one input);
//   * later stages go in pairs, as a radix-4 step in registers on rows i0, i0+h, i0+2h, i0+3h (half
//     the LDS traffic, address arithmetic and barriers of two separate stages, the same four
//     products, issued as two interleaved pairs); an odd stage left over runs two butterflies per thread;
//   * the last step hands its results to `store` instead of writing them back.
// Outputs are lazy: limbs < 1.3 * 2^31, value < (4 + 2 log_r) p.
//
// SH = true: the butterfly products of the later steps use the precomputed-quotient form (shoup29_x2, 143
// multiply-adds instead of 153) with twiddle pairs (w, wq) from `master_sh` kept in the LDS twiddle area.
//   radix <= 2^7:  the area holds the radix's whole table as pairs (at most 64: records 0-63 w, 64-127 wq);
//   radix >= 2^8:  32 pairs (records 0-31 w, 32-63 wq) serve the steps up to h = 16 (omega_{4h}^x with 4h <= 64 is a
//                  power of omega_64 whatever the radix), 128 plain twiddles (records 64-191, the powers of
//                  omega_256) serve the step h = 64, and the steps beyond take plain twiddles from memory.
// Same values, same bounds (both products return < 1.91 p).  See ntt_use_sh for who picks it.
struct TwPair { F9 w, q; };
// LR != 0: the radix is the compile-time constant 2^LR and the tile is full (2^(NTT_TILE_LOG - LR) columns): the steps
// are unrolled, each with the one product form it takes and its index arithmetic folded into constants.
template <bool SH, bool BIG, int LR, class Load, class Store>
__device__ __forceinline__ void tile_transform(const Tile& t, int log_r_arg, int log_c_arg, const uint32_t* master, const uint32_t* master_sh,
                                               bool quarter, Load load, Store store) {
  const int log_r = LR ? LR : log_r_arg, log_c = LR ? NTT_TILE_LOG - LR : log_c_arg;
  const int cmask = (1 << log_c) - 1;
  if (log_r == 0) {
    for (int c = threadIdx.x; c <= cmask; c += NTT_THREADS) store(0, c, load(0, c));
    return;
  }
  if (log_r == 1) {
    for (int c = threadIdx.x; c <= cmask; c += NTT_THREADS) {
      const F9 x = load(0, c), y = load(1, c);
      store(0, c, f9_add(x, y));
      store(1, c, f9_sub_q2(x, y));
    }
    return;
  }
  // the 2^(log_r - 1) twiddles of this radix, omega_R^j, copied next to the tile when they fit
  // (first used after the barrier that ends the first step)
  const Tile ltw(reinterpret_cast<unsigned char*>(t.A) + (size_t)NTT_TILE * 36, NTT_LDS_TW_RECORDS);
  const bool tw_lds = log_r - 1 <= NTT_LDS_TW_LOG;
  static_assert(SH || !BIG, "BIG is a form of SH");
  constexpr bool big = BIG;                                  // radix >= 2^8
  constexpr int SH_SLOTS = big ? 32 : 64;                    // pairs in the LDS twiddle area
  const int sh_shift = big ? log_r - 6 : 0;                  // radix index -> index into the pairs (powers of omega_64)
  constexpr int MID_BASE = 64, MID_LOG = 8;                  // big: plain powers of omega_256 behind the pairs
  if (SH) {
    if (log_r > 2) {
      const int count = min(1 << (log_r - 1), SH_SLOTS);
      for (int j = threadIdx.x; j < 2 * count; j += NTT_THREADS) {
        const int e = j >> 1, half = j & 1;
        ltw.put(half * SH_SLOTS + e, f9_load_tw(master_sh, (((size_t)e << (NTT_MASTER_LOG - log_r + sh_shift)) << 1) + half));
      }
      if (big)
        for (int j = threadIdx.x; j < (1 << (MID_LOG - 1)); j += NTT_THREADS)
          ltw.put(MID_BASE + j, f9_load_tw(master, (size_t)j << (NTT_MASTER_LOG - MID_LOG)));
    }
  } else if (tw_lds && log_r > 2) {
    const int count = 1 << (log_r - 1);
    for (int j = threadIdx.x; j < count; j += NTT_THREADS)
      ltw.put(j, f9_load_tw(master, (size_t)j << (NTT_MASTER_LOG - log_r)));
  }
  // plain twiddles: from LDS when this radix keeps them there (never together with the pairs), else from memory
  auto tw = [&](int idx) { return (!SH && tw_lds) ? ltw.get(idx) : f9_load_tw(master, (size_t)idx << (NTT_MASTER_LOG - log_r)); };
  auto tw_mid = [&](int idx) { return ltw.get(MID_BASE + (idx >> (log_r - MID_LOG))); };   // big, steps with 4h <= 256
  auto tw2 = [&](int idx) { TwPair r; r.w = ltw.get(idx >> sh_shift); r.q = ltw.get(SH_SLOTS + (idx >> sh_shift)); return r; };
  const int groups = 1 << (log_r + log_c - 2);
  const int quarter_rows = 1 << (log_r - 2);

  {  // first step
    const bool last = log_r == 2;
    const F9 w4 = f9_load_tw(master, (size_t)1 << (NTT_MASTER_LOG - 2));   // omega_4 (or its inverse)
    for (int g = threadIdx.x; g < groups; g += NTT_THREADS) {
      const int c = g & cmask, i = g >> log_c;
      const int r0 = (int)bitrev((uint32_t)i, log_r - 2);
      const F9 a0 = load(r0, c);
      F9 y0, y1, y2, y3;
      if (quarter) {
        y0 = y1 = y2 = y3 = a0;
      } else {
        const F9 a1 = load(r0 + 2 * quarter_rows, c), a2 = load(r0 + quarter_rows, c), a3 = load(r0 + 3 * quarter_rows, c);
        const F9 u0 = f9_add(a0, a1), u1 = f9_sub_q2(a0, a1), u2 = f9_add(a2, a3);
        const F9 u3 = mont29(f9_sub_q2(a2, a3), w4);
        y0 = f9_add(u0, u2); y2 = f9_sub_q4(u0, u2); y1 = f9_add(u1, u3); y3 = f9_sub_q2(u1, u3);
        f9_normalize(y0); f9_normalize(y1); f9_normalize(y2); f9_normalize(y3);
      }
      if (last) { store(4 * i, c, y0); store(4 * i + 1, c, y1); store(4 * i + 2, c, y2); store(4 * i + 3, c, y3); }
      else {
        const int e0 = ((4 * i) << log_c) + c, step = 1 << log_c;
        t.put(e0, y0); t.put(e0 + step, y1); t.put(e0 + 2 * step, y2); t.put(e0 + 3 * step, y3);
      }
    }
    if (last) return;
    __syncthreads();
  }

  auto paired_step = [&](const int lh) __attribute__((always_inline)) {   // two stages per step
    const int h = 1 << lh;
    const bool last = lh + 2 == log_r;
    const bool sh = SH && 2 * h <= SH_SLOTS;              // this step's twiddles are among the pairs
    const bool mid = SH && big && !sh && 4 * h <= (1 << MID_LOG);   // ... or among the plain ones behind them
    for (int g = threadIdx.x; g < groups; g += NTT_THREADS) {
      const int c = g & cmask, q = g >> log_c;
      const int j = q & (h - 1);
      const int i0 = ((q >> lh) << (lh + 2)) + j;
      const int e0 = (i0 << log_c) + c, step = h << log_c;
      const F9 x0 = t.get(e0), x1 = t.get(e0 + step), x2 = t.get(e0 + 2 * step), x3 = t.get(e0 + 3 * step);
      F9 t1, t3;
      if (SH && sh) { const TwPair w1 = tw2(j << (log_r - 1 - lh)); shoup29_x2(x1, w1.w, w1.q, x3, w1.w, w1.q, t1, t3); }   // omega_{2h}^j
      else { const int i1 = j << (log_r - 1 - lh); const F9 w1 = mid ? tw_mid(i1) : tw(i1); mont29_x2(x1, w1, x3, w1, t1, t3); }
      const F9 a0 = f9_add(x0, t1), a1 = f9_sub_q2(x0, t1), a2 = f9_add(x2, t3), a3 = f9_sub_q2(x2, t3);
      F9 s2, s3;                                                                          // omega_{4h}^j, omega_{4h}^(j+h)
      if (SH && sh) {
        const TwPair w2 = tw2(j << (log_r - 2 - lh)), w3 = tw2((j + h) << (log_r - 2 - lh));
        shoup29_x2(a2, w2.w, w2.q, a3, w3.w, w3.q, s2, s3);
      } else {
        const int i2 = j << (log_r - 2 - lh), i3 = (j + h) << (log_r - 2 - lh);
        const F9 w2 = mid ? tw_mid(i2) : tw(i2), w3 = mid ? tw_mid(i3) : tw(i3);
        mont29_x2(a2, w2, a3, w3, s2, s3);
      }
      F9 y0 = f9_add(a0, s2), y2 = f9_sub_q2(a0, s2), y1 = f9_add(a1, s3), y3 = f9_sub_q2(a1, s3);
      f9_normalize(y0); f9_normalize(y1); f9_normalize(y2); f9_normalize(y3);
      if (last) { store(i0, c, y0); store(i0 + h, c, y1); store(i0 + 2 * h, c, y2); store(i0 + 3 * h, c, y3); }
      else { t.put(e0, y0); t.put(e0 + step, y1); t.put(e0 + 2 * step, y2); t.put(e0 + 3 * step, y3); }
    }
  };
  int lh = 2;
  if constexpr (LR != 0) {
#pragma unroll
    for (int l = 2; l + 1 < LR; l += 2) {
      paired_step(l);
      if (l + 2 == LR) return;
      __syncthreads();
    }
    lh = 2 + 2 * ((LR - 2) / 2);
  } else {
    for (; lh + 1 < log_r; lh += 2) {
      paired_step(lh);
      if (lh + 2 == log_r) return;
      __syncthreads();
    }
  }

  {  // one stage left (lh = log_r - 1): two butterflies per thread, their products interleaved
    const int h = 1 << lh, total = 1 << (log_r + log_c - 1);
    for (int b = threadIdx.x; b < total; b += 2 * NTT_THREADS) {
      const int bb[2] = {b, b + NTT_THREADS};
      const bool second = bb[1] < total;
      int row[2], col[2], jj[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int bu = (u == 1 && !second) ? b : bb[u];
        col[u] = bu & cmask;
        const int q = bu >> log_c;
        jj[u] = q & (h - 1);
        row[u] = ((q >> lh) << (lh + 1)) + jj[u];
      }
      const F9 xa = t.get((row[0] << log_c) + col[0]), xb = t.get((row[1] << log_c) + col[1]);
      const F9 ya = t.get(((row[0] + h) << log_c) + col[0]), yb = t.get(((row[1] + h) << log_c) + col[1]);
      F9 ta, tb;
      if (SH && h <= SH_SLOTS) {
        const TwPair wa = tw2(jj[0] << (log_r - 1 - lh)), wb = tw2(jj[1] << (log_r - 1 - lh));
        shoup29_x2(ya, wa.w, wa.q, yb, wb.w, wb.q, ta, tb);
      } else {
        mont29_x2(ya, tw(jj[0] << (log_r - 1 - lh)), yb, tw(jj[1] << (log_r - 1 - lh)), ta, tb);
      }
      store(row[0], col[0], f9_add(xa, ta)); store(row[0] + h, col[0], f9_sub_q2(xa, ta));
      if (second) { store(row[1], col[1], f9_add(xb, tb)); store(row[1] + h, col[1], f9_sub_q2(xb, tb)); }
    }
  }
}

// Which form a pass of radix 2^log_r takes: pairs in LDS for every step (radix <= 2^7) or for the steps that fit
// (radix >= 2^8, the rest from LDS and memory as plain twiddles: BIG, a kernel instantiation of its own, so that the
// common radices keep their register budget; for the radix 2^8 itself that is the steps h = 4 and h = 16 on pairs and
// the step h = 64 on its 128 plain twiddles: direct transforms of 2^22 and 2^24 points 1.5-2 % faster than with
// plain twiddles throughout).
__host__ __device__ __forceinline__ bool ntt_big_radix(const uint32_t* master_sh, int log_r) { return master_sh && log_r >= NTT_LDS_TW_LOG + 1; }

template <bool BIG, int LR = 0, class Load, class Store>
__device__ __forceinline__ void tile_transform_any(const Tile& t, int log_r, int log_c, const uint32_t* master, const uint32_t* master_sh,
                                                   bool quarter, Load load, Store store) {
  static_assert(LR == 0 || (BIG && LR > NTT_LDS_TW_LOG), "the fixed-radix form is instantiated for large radices only");
  if constexpr (BIG) tile_transform<true, true, LR>(t, log_r, log_c, master, master_sh, quarter, load, store);
  else if (master_sh && log_r >= 3 && log_r <= NTT_LDS_TW_LOG) tile_transform<true, false, 0>(t, log_r, log_c, master, master_sh, quarter, load, store);
  else tile_transform<false, false, 0>(t, log_r, log_c, master, master_sh, quarter, load, store);
}

// omega_N^E (twiddle record) from the two-level table
__device__ __forceinline__ F9 twiddle_two_level(const uint32_t* hi, const uint32_t* lo, int lo_bits, uint64_t E) {
  const F9 a = f9_load_tw(hi, E >> lo_bits);
  const uint64_t el = E & ((1ull << lo_bits) - 1);
  if (el == 0) return a;
  return mont29(a, f9_load_tw(lo, el));
}

struct NttPassArgs {
  NttBatch batch;
  NttTables tab;
  uint64_t n_in;     // first pass: number of valid input elements (rest read as zero)
  int log_n;         // log2 of the whole transform
  int log_r;         // this pass's radix
  int log_inner;     // log2(product of later radices)
  int log_c;         // columns per tile
  int first;         // 1 = reads the caller's input (kinds, zero padding)
  int use_scaled;    // 1 = take tw_hi_scaled for the inter-pass twiddle
  int pass;          // index of this pass (selects tw_full)
  // coset mode (cosets > 0): slot blockIdx.y = column * cosets + r evaluates column `column` on
  // the coset omega_M^(r+1) * <omega_N> (M = blowup * N) by scaling input j with
  // prescale[r * N + j] = omega_M^((r+1) j) before the transform
  int cosets;
  const uint32_t* prescale;
  uint32_t slots;    // transforms in this launch; the grid is 1-D with tiles * slots workgroups
  int log_group;     // log2 of the neighbouring tile positions kept together in the workgroup order (0 = none)
  NttU64 u64;        // first pass only
};

// Non-final pass: digit transform + inter-pass twiddle; out may alias in when !first.
//
// Workgroup order: hardware hands consecutive workgroup ids to the 8 XCDs in turn, each with its
// own L2.  The id is first un-swizzled so that one XCD walks a contiguous range of logical ids,
// and a logical id is tile-major / slot-minor: the slots of one tile position (all columns and
// cosets) run on one XCD at about the same time, so the inter-pass twiddle and prescale records
// of that position, and in coset mode the input tile shared by a column's cosets, come from L2
// instead of once per slot from memory.
template <bool BIG, int LR = 0>
__global__ __launch_bounds__(NTT_THREADS, 4) void ntt_pass_kernel(const NttPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Tile tile(lds_raw);
  uint32_t wg = blockIdx.x;
  if ((gridDim.x & 7u) == 0) wg = (wg & 7u) * (gridDim.x >> 3) + (wg >> 3);
#if PLK_TILE_GROUPING
  // Tiles narrower than a 128-byte line (fewer than 4 columns) leave part of every line they touch to the tile
  // positions next to them: 2^a.log_group neighbouring positions are therefore made neighbours in the order as well
  // (position-group major, slot, position within the group minor), so that the pieces of a line are asked for - and
  // written - by workgroups that run side by side on one XCD, and meet in its L2.
  const uint32_t in_group = wg & ((1u << a.log_group) - 1u);
  const uint32_t gs = wg >> a.log_group;
  const uint32_t group_id = gs / a.slots;
  int slot = (int)(gs - group_id * a.slots);
  const uint32_t tile_id = (group_id << a.log_group) + in_group;
#if PLK_COSET_MAJOR
  // measurement knob: the cosets of one column far apart in the order (coset-major) instead of side by side
  if (a.first && a.cosets) { const int b = (int)a.slots / a.cosets; slot = (slot % b) * a.cosets + slot / b; }
#endif
#else
  const uint32_t tile_id = wg / a.slots;
  const int slot = (int)(wg - tile_id * a.slots);
#endif
  const int poly = (a.first && a.cosets) ? slot / a.cosets : slot;
  const uint64_t* __restrict__ in = a.batch.src(poly);
  uint64_t* __restrict__ out = a.batch.dst(slot);
  const int log_c = LR ? NTT_TILE_LOG - LR : a.log_c, log_r = LR ? LR : a.log_r, log_inner = a.log_inner;
  const int tiles_per_block = 1 << (log_inner - log_c);       // tiles per value of the outer digits
  const uint64_t o = tile_id >> (log_inner - log_c);
  const uint64_t q0 = (uint64_t)(tile_id & (tiles_per_block - 1)) << log_c;
  const uint64_t base = (o << (log_r + log_inner)) + q0;
  const int elems = 1 << (log_r + log_c);
  const uint32_t kind = a.first ? a.batch.kind_of(poly) : (uint32_t)NTT_IN_FIELD;

  // inter-pass twiddle omega_{N'}^(k * q), N' = 2^(log_r + log_inner), as a power of omega_N
  const int shift = a.log_n - (log_r + log_inner);
  const uint32_t* hi = a.use_scaled ? a.tab.tw_hi_scaled : a.tab.tw_hi;
  const uint32_t* full = a.tab.tw_full[a.pass];
  const int coset = a.cosets ? slot % a.cosets : 0;
  (void)elems;
  const bool quarter = a.first && log_r >= 2 && (a.n_in << 2) <= (1ull << a.log_n);
  auto load =
      [&](int r, int c) {   // row r of the digit this pass transforms: a run of 2^log_c consecutive elements per row
        const uint64_t idx = base + ((uint64_t)r << log_inner) + c;
        const bool valid = !(a.first && idx >= a.n_in);
        F9 v = load_input(in, idx, valid, kind, a.u64);
        if (kind == NTT_IN_U64 && a.u64.conv_out && valid) fe_store(a.u64.conv_out + (uint64_t)(poly - a.u64.conv_first) * a.u64.conv_stride, idx, fe_reduce_f9(v));
        if (a.first && a.cosets) v = mont29(v, f9_load_tw(a.prescale, ((uint64_t)coset << a.log_n) + idx));
        return v;
      };
  auto store =
      [&](int k, int c, const F9& v) {
        const uint64_t q = q0 + c;
        const F9 tw = full ? f9_load_tw(full, ((uint64_t)k << log_inner) + q)
                           : twiddle_two_level(hi, a.tab.tw_lo, a.tab.lo_bits, ((uint64_t)k * q) << shift);
#if PLK_NT_PASS_STORE
        fe_store_nt(out, base + ((uint64_t)k << log_inner) + c, fe_pack_f9(mont29(v, tw)));
#else
        fe_store(out, base + ((uint64_t)k << log_inner) + c, fe_pack_f9(mont29(v, tw)));
#endif
      };
  tile_transform_any<BIG, LR>(tile, log_r, log_c, a.tab.master, a.tab.master_sh, quarter, load, store);
}

struct NttFinalArgs {
  NttBatch batch;
  NttTables tab;
  uint64_t n_in;
  int log_n;
  int log_r;      // this (last) pass's radix
  int log_r1;     // radix of pass 1 (0 if this is the only pass)
  int log_r2;     // radix of pass 2 (0 unless there are three passes)
  int log_c;      // rows (consecutive k1) per tile
  int first;
  // coset mode, as in NttPassArgs; slot r of column b is written at out[b] + r * coset_stride
  int cosets;
  const uint32_t* prescale;
  uint64_t coset_stride;   // in uint64_t units
  NttU64 u64;              // when this is the only pass
};

// Last pass: contiguous rows in, natural-order scatter out (never in place), canonical values.
// Row o_c = (k1_0 + c) * R2 + k2 of the pass-major buffer feeds column c of the tile; a thread of the
// first step reads its four elements of that row straight from memory (32-byte pieces; the row's
// lines are shared by neighbouring waves of the workgroup through the vector cache).
template <bool BIG, int LR = 0>
__global__ __launch_bounds__(NTT_THREADS, 4) void ntt_final_kernel(const NttFinalArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Tile tile(lds_raw);
#if PLK_FINAL_XCD_ORDER
  // Hardware deals consecutive workgroup ids to the 8 XCDs in turn; un-swizzled, one XCD walks a contiguous range of
  // (slot, tile) pairs, so the neighbouring tiles that write the two halves of one 128-byte output line (tiles of 2
  // columns) run side by side on one XCD and the halves leave its L2 as one line.
  uint32_t wg = blockIdx.x + gridDim.x * blockIdx.y;
  const uint32_t total = gridDim.x * gridDim.y;
  if ((total & 7u) == 0) wg = (wg & 7u) * (total >> 3) + (wg >> 3);
  const int slot = (int)(wg / gridDim.x);
  const uint32_t tile_x = wg - (uint32_t)slot * gridDim.x;
#else
  const int slot = blockIdx.y;   // slots share nothing here, so the plain 2-D order is kept
  const uint32_t tile_x = blockIdx.x;
#endif
  const int col = a.cosets ? slot / a.cosets : slot;
  const int poly = a.first ? col : slot;
  const uint64_t* __restrict__ in = a.batch.src(poly);
  uint64_t* __restrict__ out = a.batch.dst(col) + (a.cosets ? (uint64_t)(slot % a.cosets) * a.coset_stride : 0);
  const int log_c = LR ? NTT_TILE_LOG - LR : a.log_c, log_r = LR ? LR : a.log_r;
  const uint64_t k2 = tile_x & ((1u << a.log_r2) - 1);
  const uint64_t k1_0 = (uint64_t)(tile_x >> a.log_r2) << log_c;
  const uint32_t kind = a.first ? a.batch.kind_of(poly) : (uint32_t)NTT_IN_FIELD;

  const bool scaled = a.tab.scale != nullptr;   // single-pass inverse transforms only: the record is read where it is used, so that
                                               // the larger transforms do not hold nine registers for it across the tile transform
  const bool quarter = a.first && log_r >= 2 && (a.n_in << 2) <= (1ull << a.log_n);
  auto load =
      [&](int r, int c) {
        const uint64_t orow = ((k1_0 + c) << a.log_r2) + k2;
        const uint64_t idx = (orow << log_r) + r;
        const bool valid = !(a.first && idx >= a.n_in);
        F9 v = load_input(in, idx, valid, kind, a.u64);
        if (kind == NTT_IN_U64 && a.u64.conv_out && valid) fe_store(a.u64.conv_out + (uint64_t)(poly - a.u64.conv_first) * a.u64.conv_stride, idx, fe_reduce_f9(v));
        if (a.first && a.cosets) v = mont29(v, f9_load_tw(a.prescale, ((uint64_t)(slot % a.cosets) << a.log_n) + idx));
        return v;
      };
  auto store =
      [&](int k, int c, const F9& v) {
        const uint64_t dst = (k1_0 + c) + (k2 << a.log_r1) + ((uint64_t)k << (a.log_r1 + a.log_r2));
        fe_store(out, dst, fe_reduce_f9(scaled ? mont29(v, f9_load_tw(a.tab.scale, 0)) : v));
      };
  tile_transform_any<BIG, LR>(tile, log_r, log_c, a.tab.master, a.tab.master_sh, quarter, load, store);
}

// ---- the local phases of a four-step transform sharded over G ranks ---------------------------------------------
//
// N = N1 * N2, n = n1 * N2 + n2, k = k1 + N1 * k2, rank g of G holds the contiguous inputs n1 in [g r1, (g + 1) r1),
// r1 = N1 / G, and after the exchanges the columns n2 in [g c2, (g + 1) c2), c2 = N2 / G.  Both local phases run on
// the buffers exactly as the all-to-all exchanges deliver them and write exactly what the next exchange sends, so no
// pack or unpack copy stands between a collective and a kernel:
//   phase A  (column transforms, size N1 = 2^(la + lb), n1 = a 2^lb + b):
//     in   recv1 = [n1][c2]            (rank h's block of rows is what h sent: already row-major in n1)
//     A_FIRST  digit a, times omega_N1^(k_a b)                     -> tmp  [k_a 2^lb + b][c2]
//     A_LAST   digit b, times omega_N^(k1 (g c2 + col)) (and 1/N)  -> send2 [k1][c2], k1 = k_a + 2^la k_b
//              (rows in k1 order: the rows of destination h are one contiguous block)
//   phase B  (row transforms, size N2 = 2^(lA + lB), n2 = A 2^lB + B; recv2 = [h][k1l][c2], n2 = h c2 + j):
//     B_FIRST  digit A, times omega_N2^(k_A B)                     -> tmp, same positions
//     B_LAST   digit B -> canonical values, k2 = k_A + 2^lA k_B, written either
//              transposed   [k1l][k2]                   (no further exchange), or
//              send3 = [h][k2l][k1l], k2 = h c2 + k2l   (exchange 3 then delivers [g][k2l][k1l] to the owner of k2).
// A sub-transform of at most 2^11 points is a single pass (the LAST kernel alone, lb = 0 or lB = 0).
// The twiddle tables are records as everywhere else: tw_sub[(k << log_lo) + q] = omega_M^(k q),
// tw_block[k1 * c2 + col] = omega_N^(k1 (col0 + col)) (times 1/N for the inverse), or the two-level pair if no block.
enum FsMode : int { FS_A_FIRST = 0, FS_A_LAST = 1, FS_B_FIRST = 2, FS_B_LAST_NATURAL = 3, FS_B_LAST_TRANSPOSED = 4 };

struct FsArgs {
  const uint64_t* in;
  uint64_t* out;
  const uint32_t* master;
  const uint32_t* master_sh;
  const uint32_t* tw_sub;     // FIRST modes
  const uint32_t* tw_block;   // A_LAST: full block, or nullptr
  const uint32_t* tw_hi;      // A_LAST without a block: omega_N^(e << lo_bits) (times 1/N for the inverse) ...
  const uint32_t* tw_lo;      // ... and omega_N^e
  int lo_bits;
  int log_r;       // this pass's radix
  int log_lo;      // bits of the lower digit of the sub-transform (lb / lB; 0 for a single pass)
  int log_hi;      // bits of the higher digit (la / lA; 0 for a single pass)
  int log_c;       // columns per tile
  int log_c2;      // log2(N2 / G)
  int log_r1;      // log2(N1 / G)
  int log_n2;      // log2(N2)
  uint64_t col0;   // rank * c2
};

template <int MODE>
__global__ __launch_bounds__(NTT_THREADS, 4) void ntt_fs_kernel(const FsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const Tile tile(lds_raw);
  const int log_r = a.log_r, log_c = a.log_c, log_c2 = a.log_c2;
  const uint64_t* __restrict__ in = a.in;
  uint64_t* __restrict__ out = a.out;
  const uint64_t cmask2 = (1ull << log_c2) - 1;
  // phase B address of (k1l, n2) in recv2 = [h][k1l][c2]
  auto addr_b = [&](uint64_t k1l, uint64_t n2) { return ((n2 >> log_c2) << (a.log_r1 + log_c2)) + (k1l << log_c2) + (n2 & cmask2); };
  const uint32_t wg = blockIdx.x;
  auto run = [&](auto load, auto store) {
    tile_transform_any<false>(tile, log_r, log_c, a.master, a.master_sh, false, load, store);   // (sub-transform radices stay below 2^9)
  };
  if constexpr (MODE == FS_A_FIRST) {
    const uint64_t colblocks = 1ull << (log_c2 - log_c);
    const uint64_t b = wg / colblocks, colbase = (wg % colblocks) << log_c;
    run([&](int r, int c) { return f9_from_fe(fe_load(in, (((((uint64_t)r << a.log_lo) + b)) << log_c2) + colbase + c)); },
        [&](int k, int c, const F9& v) {
          const F9 tw = f9_load_tw(a.tw_sub, ((uint64_t)k << a.log_lo) + b);
          fe_store(out, ((((uint64_t)k << a.log_lo) + b) << log_c2) + colbase + c, fe_pack_f9(mont29(v, tw)));
        });
  } else if constexpr (MODE == FS_A_LAST) {
    const uint64_t colblocks = 1ull << (log_c2 - log_c);
    const uint64_t ka = wg / colblocks, colbase = (wg % colblocks) << log_c;
    run([&](int r, int c) { return f9_from_fe(fe_load(in, (((ka << log_r) + (uint64_t)r) << log_c2) + colbase + c)); },
        [&](int k, int c, const F9& v) {
          const uint64_t k1 = ka + ((uint64_t)k << a.log_hi), col = colbase + c;
          const F9 tw = a.tw_block ? f9_load_tw(a.tw_block, (k1 << log_c2) + col)
                                   : twiddle_two_level(a.tw_hi, a.tw_lo, a.lo_bits, k1 * (a.col0 + col));
          fe_store(out, (k1 << log_c2) + col, fe_pack_f9(mont29(v, tw)));
        });
  } else if constexpr (MODE == FS_B_FIRST) {
    const uint64_t bblocks = 1ull << (a.log_lo - log_c);
    const uint64_t k1l = wg / bblocks, b0 = (wg % bblocks) << log_c;
    run([&](int r, int c) { return f9_from_fe(fe_load(in, addr_b(k1l, ((uint64_t)r << a.log_lo) + b0 + c))); },
        [&](int k, int c, const F9& v) {
          const F9 tw = f9_load_tw(a.tw_sub, ((uint64_t)k << a.log_lo) + b0 + c);
          fe_store(out, addr_b(k1l, ((uint64_t)k << a.log_lo) + b0 + c), fe_pack_f9(mont29(v, tw)));
        });
  } else if constexpr (MODE == FS_B_LAST_NATURAL) {   // columns = neighbouring k1l
    const uint64_t kblocks = 1ull << (a.log_r1 - log_c);
    const uint64_t kA = wg / kblocks, k1l0 = (wg % kblocks) << log_c;
    run([&](int r, int c) { return f9_from_fe(fe_load(in, addr_b(k1l0 + c, (kA << log_r) + (uint64_t)r))); },
        [&](int k, int c, const F9& v) {
          const uint64_t k2 = kA + ((uint64_t)k << a.log_hi);
          fe_store(out, ((k2 >> log_c2) << (log_c2 + a.log_r1)) + ((k2 & cmask2) << a.log_r1) + k1l0 + c, fe_reduce_f9(v));
        });
  } else {                  // FS_B_LAST_TRANSPOSED, columns = neighbouring k_A
    const uint64_t ablocks = 1ull << (a.log_hi - log_c);
    const uint64_t k1l = wg / ablocks, kA0 = (wg % ablocks) << log_c;
    run([&](int r, int c) { return f9_from_fe(fe_load(in, addr_b(k1l, ((kA0 + c) << log_r) + (uint64_t)r))); },
        [&](int k, int c, const F9& v) {
          const uint64_t k2 = kA0 + c + ((uint64_t)k << a.log_hi);
          fe_store(out, (k1l << a.log_n2) + k2, fe_reduce_f9(v));
        });
  }
}

// record[k1 * c2 + col] = base^(k1 (col0 + col)) * mul, k1 < rows, col < 2^log_c2
__global__ void ntt_block_table_kernel(uint32_t* out, uint64_t rows, int log_c2, uint64_t col0, Fe base, Fe mul) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (rows << log_c2)) return;
  const uint64_t k1 = i >> log_c2, col = i & ((1ull << log_c2) - 1);
  f9_store_tw(out, i, f9_twiddle_from_fe(fe_mul(fe_pow_u64(base, k1 * (col0 + col)), mul)));
}

// ---- table construction (device side, run once per plan)

// record[i] = (base^(i << shift) * mul) as a twiddle record, i < count; base, mul in the
// library's Montgomery form
__global__ void ntt_table_kernel(uint32_t* out, uint64_t count, Fe base, int shift, Fe mul) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Fe v = fe_mul(fe_pow_u64(base, i << shift), mul);
  f9_store_tw(out, i, f9_twiddle_from_fe(v));
}

// record[(r << log_n) + j] = base^((r + 1) j), r < cosets, j < 2^log_n (base = omega_{blowup * N})
__global__ void ntt_prescale_table_kernel(uint32_t* out, int log_n, int cosets, Fe base) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((uint64_t)cosets << log_n)) return;
  const uint64_t r = i >> log_n, j = i & ((1ull << log_n) - 1);
  f9_store_tw(out, i, f9_twiddle_from_fe(fe_pow_u64(base, (r + 1) * j)));
}

// record[(k << log_inner) + q] = base^(k q) * mul, k < 2^log_r, q < 2^log_inner
__global__ void ntt_full_table_kernel(uint32_t* out, int log_r, int log_inner, Fe base, Fe mul) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> (log_r + log_inner)) return;
  const uint64_t k = i >> log_inner, q = i & ((1ull << log_inner) - 1);
  const Fe v = fe_mul(fe_pow_u64(base, k * q), mul);
  f9_store_tw(out, i, f9_twiddle_from_fe(v));
}

}  // namespace plk
