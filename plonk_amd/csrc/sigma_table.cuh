// Permutation (sigma) table from per-wire node ids: group wire positions by node id and link
// each group into a cycle.
//
// Reference: build_permutation_trace, src/permutation_polynomial.rs:18-39 (cycle p0 -> p1 -> ...
// -> p0 per node id; a single position maps to itself), over the groups of
// group_wire_positions_by_node_id, src/execution_trace.rs:191-197, then build_permutation_table,
// src/permutation_polynomial.rs:79-202, which stores OffsetTable::to_index (src/offset_table.rs:61-64)
// of the next position — the global ROW only, whatever wire the next position is on — into the
// sigma column of the current position's wire; unused wires point at their own row (:121-122, :157).
//
// collect_wire_positions (src/execution_trace.rs:109-187) walks the operation tables in the order
// the offset table numbers them and, within a row, input 0, 1, 2, then the output, so a group's
// positions are in ascending (global row, wire) order.  That order is reproduced with a STABLE
// least-significant-digit radix sort of (node id, position) pairs by node id, positions starting
// out ascending; the cycle links then come from neighbouring entries of the sorted list.
//
// Flattened input: node_ids[w][j] = node id on wire w (0..2 inputs, 3 output) of global row j, or
// SIG_NO_WIRE where the operation of that row does not use the wire.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plk {

constexpr uint64_t SIG_NO_WIRE = ~0ull;
constexpr int SIG_CHUNK = 2048;        // pairs per wave (one wave per workgroup)
constexpr int SIG_BINS = 257;          // 256 digit values + one bin that keeps unused wires last
constexpr int SIG_SCAN_THREADS = 1024;
constexpr int SIG_SCAN_ITEMS = 4;
constexpr int SIG_SCAN_TILE = SIG_SCAN_THREADS * SIG_SCAN_ITEMS;   // histogram entries per scan workgroup

struct SigCols {
  const uint64_t* node[4];
  uint64_t* sigma[4];
};

__device__ __forceinline__ uint32_t sig_digit(uint64_t key, int shift) {
  return key == SIG_NO_WIRE ? 256u : (uint32_t)(key >> shift) & 0xffu;
}

// pass `p` (digit bits 8p..8p+7) can be skipped when that digit is zero in every used key:
// the stable pass would be the identity (unused wires already sit at the end after pass 0)
__device__ __forceinline__ bool sig_pass_is_identity(const uint64_t* or_bits, int p) {
  return p > 0 && ((*or_bits >> (8 * p)) & 0xffu) == 0;
}

// The pairs ping-pong between two buffers; only real passes move them, so the buffer holding the
// pairs before pass `p` (p = 8: after the last pass) is the number of real passes before p, mod 2.
__device__ __forceinline__ int sig_buffer_before(const uint64_t* or_bits, int p) {
  int real = 0;
  for (int q = 0; q < p; ++q) real += sig_pass_is_identity(or_bits, q) ? 0 : 1;
  return real & 1;
}

struct SigPairs {
  uint64_t* keys[2];
  uint32_t* vals[2];     // null when the keys carry their payload in the bits the sort does not look at
};

// pairs in position order: position = 4 * row + wire; also the OR of all used node ids
__global__ __launch_bounds__(256) void sig_keys_kernel(SigCols cols, uint64_t n, uint64_t* keys, uint32_t* vals, uint64_t* or_bits) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;   // one row per thread
  uint64_t key = 0;
  if (j < n) {
    uint64_t k[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { k[w] = cols.node[w][j]; if (k[w] != SIG_NO_WIRE) key |= k[w]; }
    uint4* kd = reinterpret_cast<uint4*>(keys + 4 * j);          // 32-byte aligned: 4 keys per row
    kd[0] = make_uint4((uint32_t)k[0], (uint32_t)(k[0] >> 32), (uint32_t)k[1], (uint32_t)(k[1] >> 32));
    kd[1] = make_uint4((uint32_t)k[2], (uint32_t)(k[2] >> 32), (uint32_t)k[3], (uint32_t)(k[3] >> 32));
    const uint32_t p = (uint32_t)(4 * j);
    *reinterpret_cast<uint4*>(vals + 4 * j) = make_uint4(p, p + 1, p + 2, p + 3);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) key |= __shfl_xor(key, d);
  // one atomic per wave, and only while it still adds bits (nearly every wave finds them all set)
  if ((threadIdx.x & 63) == 0 && (key & ~__atomic_load_n(or_bits, __ATOMIC_RELAXED)))
    atomicOr((unsigned long long*)or_bits, (unsigned long long)key);
}

// per-wave digit counts: hist[bin * waves + wave]
__global__ __launch_bounds__(64) void sig_hist_kernel(SigPairs pairs, uint64_t total, int pass, const uint64_t* or_bits,
                                                      uint32_t* hist, uint32_t waves) {
  if (sig_pass_is_identity(or_bits, pass)) return;
  const uint64_t* keys = pairs.keys[sig_buffer_before(or_bits, pass)];
  __shared__ uint32_t cnt[SIG_BINS];
  for (int b = threadIdx.x; b < SIG_BINS; b += 64) cnt[b] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * SIG_CHUNK;
  for (int it = 0; it < SIG_CHUNK / 64; ++it) {
    const uint64_t i = base + it * 64 + threadIdx.x;
    if (i < total) atomicAdd(&cnt[sig_digit(keys[i], 8 * pass)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < SIG_BINS; b += 64) hist[(uint64_t)b * waves + blockIdx.x] = cnt[b];
}

// Exclusive scan of the whole histogram in bin-major order, in two steps: every workgroup scans
// SIG_SCAN_TILE consecutive entries in place and leaves its total in sums[]; one workgroup then
// turns sums[] into exclusive offsets.  The scatter kernel adds the two.
__device__ __forceinline__ uint32_t sig_block_exclusive(uint32_t v, uint32_t* lds, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t carry = 0, all = 0;
  for (int w = 0; w < SIG_SCAN_THREADS / 64; ++w) {
    const uint32_t t = lds[w];
    if (w < wave) carry += t;
    all += t;
  }
  __syncthreads();
  *total = all;
  return carry + inc - v;
}

__global__ __launch_bounds__(SIG_SCAN_THREADS) void sig_scan_tiles_kernel(uint32_t* hist, uint64_t entries, uint32_t* sums, int pass,
                                                                          const uint64_t* or_bits) {
  if (sig_pass_is_identity(or_bits, pass)) return;
  __shared__ uint32_t lds[SIG_SCAN_THREADS / 64];
  const uint64_t e0 = (uint64_t)blockIdx.x * SIG_SCAN_TILE + (uint64_t)threadIdx.x * SIG_SCAN_ITEMS;
  uint32_t c[SIG_SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < SIG_SCAN_ITEMS; ++k) { c[k] = e0 + k < entries ? hist[e0 + k] : 0; sum += c[k]; }
  uint32_t total;
  uint32_t run = sig_block_exclusive(sum, lds, &total);
#pragma unroll
  for (int k = 0; k < SIG_SCAN_ITEMS; ++k) { if (e0 + k < entries) hist[e0 + k] = run; run += c[k]; }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SIG_SCAN_THREADS) void sig_scan_sums_kernel(uint32_t* sums, uint32_t count, int pass, const uint64_t* or_bits) {
  if (sig_pass_is_identity(or_bits, pass)) return;
  __shared__ uint32_t lds[SIG_SCAN_THREADS / 64];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < count; base += SIG_SCAN_THREADS) {   // uniform trip count
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < count ? sums[i] : 0;
    uint32_t total;
    const uint32_t ex = sig_block_exclusive(v, lds, &total);
    if (i < count) sums[i] = carry + ex;
    carry += total;
  }
}

// stable scatter of one wave's chunk into the other buffer; an identity pass moves nothing
__global__ __launch_bounds__(64) void sig_scatter_kernel(SigPairs pairs, uint64_t total, int pass, const uint64_t* or_bits,
                                                         const uint32_t* hist, const uint32_t* sums, uint32_t waves) {
  if (sig_pass_is_identity(or_bits, pass)) return;
  const int src = sig_buffer_before(or_bits, pass);
  const uint64_t* keys_in = pairs.keys[src];
  const uint32_t* vals_in = pairs.vals[src];
  uint64_t* keys_out = pairs.keys[src ^ 1];
  uint32_t* vals_out = pairs.vals[src ^ 1];
  const uint64_t base = (uint64_t)blockIdx.x * SIG_CHUNK;
  __shared__ uint32_t off[SIG_BINS];
  for (int b = threadIdx.x; b < SIG_BINS; b += 64) {
    const uint64_t e = (uint64_t)b * waves + blockIdx.x;
    off[b] = hist[e] + sums[e / SIG_SCAN_TILE];
  }
  __syncthreads();
  const uint64_t lane_lt = (1ull << threadIdx.x) - 1;
  // the chunk is read in groups of SIG_PREFETCH rounds before any of them is ranked: ranking is a chain of ballots,
  // LDS updates and barriers per round, and a load issued inside it would be waited for every time
  constexpr int SIG_PREFETCH = 8;
  for (int it0 = 0; it0 < SIG_CHUNK / 64; it0 += SIG_PREFETCH) {
    uint64_t kk[SIG_PREFETCH];
    uint32_t vv[SIG_PREFETCH];
#pragma unroll
    for (int u = 0; u < SIG_PREFETCH; ++u) {
      const uint64_t i = base + (uint64_t)(it0 + u) * 64 + threadIdx.x;
      kk[u] = i < total ? keys_in[i] : 0;
      vv[u] = (i < total && vals_in) ? vals_in[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < SIG_PREFETCH; ++u) {
      const uint64_t i = base + (uint64_t)(it0 + u) * 64 + threadIdx.x;
      const bool live = i < total;
      const uint64_t key = kk[u];
      const uint32_t val = vv[u];
      const uint32_t d = live ? sig_digit(key, 8 * pass) : 0;
      // lanes holding the same bin as this one: nine ballots, one per bit of the bin number
      uint64_t same = __ballot(live);
#pragma unroll
      for (int b = 0; b < 9; ++b) {
        const uint64_t m = __ballot(live && ((d >> b) & 1u));
        same &= ((d >> b) & 1u) ? m : ~m;
      }
      const uint32_t rank = (uint32_t)__popcll(same & lane_lt);
      uint32_t dst = 0;
      if (live) dst = off[d] + rank;
      __syncthreads();                                   // every lane has read its offset
      if (live && rank == 0) off[d] += (uint32_t)__popcll(same);
      __syncthreads();
      if (live) { keys_out[dst] = key; if (vals_out) vals_out[dst] = val; }
    }
  }
}

// sigma of every position from the sorted pairs
__global__ __launch_bounds__(256) void sig_link_kernel(SigPairs pairs, const uint64_t* or_bits, uint64_t total, SigCols cols) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int buf = sig_buffer_before(or_bits, 8);
  const uint64_t* keys = pairs.keys[buf];
  const uint32_t* vals = pairs.vals[buf];
  const uint64_t key = keys[i];
  const uint32_t pos = vals[i];
  uint32_t next;
  if (key == SIG_NO_WIRE) {
    next = pos;                                        // unused wire: its own row
  } else if (i + 1 < total && keys[i + 1] == key) {
    next = vals[i + 1];
  } else {                                             // last of its group: back to the group's first entry
    uint64_t lo = 0, hi = i;                           // first index whose key is >= key
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    next = vals[lo];
  }
  cols.sigma[pos & 3][pos >> 2] = (uint64_t)(next >> 2);
}

}  // namespace plk
