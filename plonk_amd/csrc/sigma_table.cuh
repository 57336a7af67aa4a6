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

// pairs in position order: position = 4 * row + wire; also the OR of all used node ids
__global__ __launch_bounds__(256) void sig_keys_kernel(SigCols cols, uint64_t n, uint64_t* keys, uint32_t* vals, uint64_t* or_bits) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t key = 0;
  if (i < 4 * n) {
    key = cols.node[i & 3][i >> 2];
    keys[i] = key;
    vals[i] = (uint32_t)i;
    if (key == SIG_NO_WIRE) key = 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) key |= __shfl_xor(key, d);
  if ((threadIdx.x & 63) == 0 && key) atomicOr((unsigned long long*)or_bits, (unsigned long long)key);
}

// per-wave digit counts: hist[bin * waves + wave]
__global__ __launch_bounds__(64) void sig_hist_kernel(const uint64_t* keys, uint64_t total, int pass, const uint64_t* or_bits,
                                                      uint32_t* hist, uint32_t waves) {
  if (sig_pass_is_identity(or_bits, pass)) return;
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

// exclusive scan of the whole histogram in bin-major order (one workgroup)
__global__ __launch_bounds__(SIG_SCAN_THREADS) void sig_scan_kernel(uint32_t* hist, uint64_t entries, int pass, const uint64_t* or_bits) {
  if (sig_pass_is_identity(or_bits, pass)) return;
  __shared__ uint32_t part[SIG_SCAN_THREADS];
  const uint64_t per = (entries + SIG_SCAN_THREADS - 1) / SIG_SCAN_THREADS;
  const uint64_t lo = min((uint64_t)threadIdx.x * per, entries), hi = min(lo + per, entries);
  uint32_t sum = 0;
  for (uint64_t e = lo; e < hi; ++e) sum += hist[e];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < SIG_SCAN_THREADS; d <<= 1) {   // Hillis-Steele inclusive scan
    const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  for (uint64_t e = lo; e < hi; ++e) { const uint32_t c = hist[e]; hist[e] = run; run += c; }
}

// stable scatter of one wave's chunk; an identity pass copies instead
__global__ __launch_bounds__(64) void sig_scatter_kernel(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t total, int pass,
                                                         const uint64_t* or_bits, const uint32_t* hist, uint32_t waves,
                                                         uint64_t* keys_out, uint32_t* vals_out) {
  const uint64_t base = (uint64_t)blockIdx.x * SIG_CHUNK;
  if (sig_pass_is_identity(or_bits, pass)) {
    for (int it = 0; it < SIG_CHUNK / 64; ++it) {
      const uint64_t i = base + it * 64 + threadIdx.x;
      if (i < total) { keys_out[i] = keys_in[i]; vals_out[i] = vals_in[i]; }
    }
    return;
  }
  __shared__ uint32_t off[SIG_BINS];
  for (int b = threadIdx.x; b < SIG_BINS; b += 64) off[b] = hist[(uint64_t)b * waves + blockIdx.x];
  __syncthreads();
  const uint64_t lane_lt = (1ull << threadIdx.x) - 1;
  for (int it = 0; it < SIG_CHUNK / 64; ++it) {
    const uint64_t i = base + it * 64 + threadIdx.x;
    const bool live = i < total;
    uint64_t key = 0;
    uint32_t val = 0, d = 0;
    if (live) { key = keys_in[i]; val = vals_in[i]; d = sig_digit(key, 8 * pass); }
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
    if (live) { keys_out[dst] = key; vals_out[dst] = val; }
  }
}

// sigma of every position from the sorted pairs
__global__ __launch_bounds__(256) void sig_link_kernel(const uint64_t* keys, const uint32_t* vals, uint64_t total, SigCols cols) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
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
